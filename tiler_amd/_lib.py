"""ctypes binding of libtilemotion.so (include/tilemotion.h).  Fails loudly when the library is missing."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path():
    variant = os.environ.get("TM_LIB_VARIANT")  # development builds of tools/build_variant.sh
    if variant:
        return os.path.join(_HERE, "lib", "variants", "libtilemotion_%s.so" % variant)
    return os.path.join(_HERE, "lib", "libtilemotion.so")


class TileMotionError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libtilemotion error {code}: {msg}")
        self.code = code


_lib = None

c_void_p, c_int, c_int64, c_double, c_char_p = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double, ctypes.c_char_p

# name -> (restype, argtypes); mirrors include/tilemotion.h one to one
SIGNATURES = {
    "tm_last_error": (c_char_p, []),
    "tm_device_count": (c_int, []),
    "tm_version": (c_char_p, []),
    "tm_probe_mfma_i8": (c_int, [c_double, ctypes.POINTER(c_double)]),
    "tm_probe_hbm_triad": (c_int, [c_int64, ctypes.POINTER(c_double)]),
    "tm_stage_load": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "tm_stage_features_rgb": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "tm_stage_features_pal": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "tm_stage_features_cluster": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "tm_stage_window_dcts": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "tm_stage_motion_search": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "tm_stage_knn_topk": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "tm_stage_epu_rerank": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                    c_void_p, c_void_p]),
    "tm_stage_knn": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "tm_knn_index_create": (c_void_p, [c_void_p, c_int64, c_void_p]),
    "tm_knn_index_destroy": (None, [c_void_p]),
    "tm_knn_index_search": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "tm_knn_index_last_stats": (c_int, [c_void_p, ctypes.POINTER(c_double), ctypes.POINTER(c_int), ctypes.POINTER(c_int64)]),
    "tm_knn_last_plan": (c_int, [ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int64)]),
    "tm_stage_dedup": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_int64), c_void_p]),
    "tm_stage_kmeans": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, ctypes.POINTER(c_int),
                                ctypes.POINTER(c_int), c_void_p]),
    "tm_stage_kmeans_seeded": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, ctypes.POINTER(c_int),
                                       ctypes.POINTER(c_int), c_void_p]),
    "tm_stage_quantize_palettes": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p]),
    "tm_stage_palettize": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "tm_stage_kmodes": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(c_int), c_void_p]),
    "tm_stage_dither": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
}


def lib():
    """The loaded library; raises if it has not been built (python __graft_entry__.py / tiler_amd/csrc/build.sh)."""
    global _lib
    if _lib is None:
        p = lib_path()
        if not os.path.exists(p):
            raise TileMotionError(-2, f"{p} not found: build it with tiler_amd/csrc/build.sh (no CPU fallback exists)")
        L = ctypes.CDLL(p)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name, None)
            if fn is None:
                continue  # declared-but-not-yet-built symbols are caught by tests/test_abi.py
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise TileMotionError(rc, lib().tm_last_error().decode("utf-8", "replace"))
