"""One process per GPU orchestration of TTilingEncoder.Run(esAll) (tilingencoder.pas:5529-5554).

Where the path shards: the frame tiles are independent queries of the KNN branch of
TFrame.Reconstruct (DoXY, tilingencoder.pas:1464-1659), >95 % of the work.  Every rank therefore runs Load..Dither on
the whole clip (small, deterministic, bit-identical on all ranks -- no collective needed to agree on the global tile
set, palettes or dithered tiles), matches only ITS frame range against the full database, and the per-frame results
are merged with one all-reduce(MAX) (other ranks hold -1) over RCCL/xGMI: 2 x Q x 4 bytes.  Reindex then runs
everywhere on the merged tile maps.  The collective calls go through `torch.distributed`, so the same code is
exercised on CPU with gloo in tests/test_distributed_cpu.py (there with an oracle-backed stand-in for the encoder).
"""
import torch.distributed as dist


def frame_shard(nframes, rank, world):
    """contiguous frame range of a rank: (first, count); earlier ranks take the remainder frames"""
    base, rem = divmod(nframes, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def run_all(enc, nframes, rank=0, world=1, group=None):
    """Run(esAll) over `world` processes.  `enc` needs Run/SetQueryShard/DeviceArray/SyncTileMap (TilingEncoder or a stand-in)."""
    from .encoder import TEncoderStep as S
    for step in (S.esLoad, S.esPredictMotion, S.esReduce, S.esPreparePalettes, S.esDither):
        enc.Run(step)
    first, count = frame_shard(nframes, rank, world)
    enc.SetQueryShard(first, count)
    enc.Run(S.esReconstruct)
    if world > 1:
        for which in (0, 1):  # TileIdx, KNN error: owner holds values >= 0 (errors < 2^31), everyone else -1
            t = enc.DeviceArray(which)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        enc.SyncTileMap()
    enc.Run(S.esReindex)
