"""tiler_amd -- MI355X (gfx950) implementation of the TileMotion encoder's per-frame tile pipeline.

The compute path is libtilemotion.so (hand-written HIP, C ABI in include/tilemotion.h).  This package holds
only the thin host-side mirror of the reference's TTilingEncoder interface (tilingencoder.pas:486-568) and the
ctypes binding; PyTorch is used for device memory and torch.distributed plumbing only.  There is no CPU path:
importing works anywhere, calling a compute entry point without the library or a GPU raises.
"""
from ._lib import lib, TileMotionError, lib_path, check  # noqa: F401
from . import stages  # noqa: F401
