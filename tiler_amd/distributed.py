"""One process per GPU orchestration of TTilingEncoder.Run(esAll) (tilingencoder.pas:5529-5554).

Where the path shards: the frame tiles are independent queries of the KNN branch of
TFrame.Reconstruct (DoXY, tilingencoder.pas:1464-1659), >95 % of the work.  Every rank therefore runs Load..Dither on
the whole clip (small, deterministic, bit-identical on all ranks -- no collective needed to agree on the global tile
set, palettes or dithered tiles), matches only ITS frame range against the full database, and the per-frame results
are merged with one all-reduce(MAX) (other ranks hold -1) over RCCL/xGMI: 2 x Q x 4 bytes (3 x with the extended-palette
re-rank, whose PalIdx is per item).  Reindex then runs
everywhere on the merged tile maps.  Dither is sharded by global tile (its tiles are independent: every rank dithers a
contiguous share, one all-reduce(SUM) of the 64-byte index tiles puts the whole set on every rank).  With motion prediction on, PredictMotion is sharded by frame too (merged with
all-reduce(SUM), other ranks hold 0) and Reconstruct by whole key-frame groups, the unit that chains (1496).  The collective calls go through `torch.distributed`, so the same code is
exercised on CPU with gloo in tests/test_distributed_cpu.py (there with an oracle-backed stand-in for the encoder).
"""
import torch.distributed as dist


def frame_shard(nframes, rank, world):
    """contiguous frame range of a rank: (first, count); earlier ranks take the remainder frames"""
    base, rem = divmod(nframes, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def keyframe_shard(keyframes, nframes, rank, world):
    """contiguous range of whole key-frame groups: boundaries of frame_shard snapped to the nearest key-frame start
    (frames chain inside a key frame once motion prediction is on, tilingencoder.pas:1496); a rank may get no frame"""
    kf = sorted(int(k) for k in keyframes)
    cuts = [0]
    for r in range(1, world):
        ideal = frame_shard(nframes, r, world)[0]
        snap = min(kf, key=lambda k: (abs(k - ideal), k))
        cuts.append(max(snap, cuts[-1]))
    cuts.append(nframes)
    return cuts[rank], cuts[rank + 1] - cuts[rank]


def describe(world):
    """what bench.py prints as config.parallelism"""
    if world <= 1:
        return "1 GPU"
    return ("%d ranks, one per GPU: PredictMotion / Dither / Reconstruct sharded (frames, global tiles, frames); merges by all-reduce over RCCL; "
            "Load, Reduce, PreparePalettes and the database side of Reconstruct replicated") % world


def run_all(enc, nframes, rank=0, world=1, group=None):
    """Run(esAll) over `world` processes.  `enc` needs Run/SetQueryShard/DeviceArray/SyncTileMap/KeyFrames and the
    MotionPredictRadius setting (TilingEncoder or a stand-in)."""
    from .encoder import TEncoderStep as S
    enc.Run(S.esLoad)
    motion = int(enc.MotionPredictRadius) > 0
    first, count = frame_shard(nframes, rank, world)
    enc.SetQueryShard(first, count)
    enc.Run(S.esPredictMotion)  # frames are independent (each is searched in the source pixels of its neighbour)
    if world > 1 and motion:
        for which in (6, 4, 5):  # best error, PredictedX, PredictedY: owner holds the value, everyone else 0
            dist.all_reduce(enc.DeviceArray(which), op=dist.ReduceOp.SUM, group=group)
    for step in (S.esReduce, S.esPreparePalettes):
        enc.Run(step)
    # DitherTile is independent per global tile (2690): every rank dithers a contiguous share of the tiles, the others stay 0
    enc.SetDitherShard(rank, world)
    enc.Run(S.esDither)
    if world > 1:
        dist.all_reduce(enc.DeviceArray(7), op=dist.ReduceOp.SUM, group=group)
    if motion:
        first, count = keyframe_shard(enc.KeyFrames(), nframes, rank, world)
        enc.SetQueryShard(first, count)
    enc.Run(S.esReconstruct)
    if world > 1:
        # TileIdx, error: owner holds values >= 0 (errors < 2^31) or -1 (perfect prediction), everyone else -1.  With
        # FrameTilingExtendedPaletteUsage the item's palette is the re-rank's own choice (1591-1608), not the tile's: merged the same way
        epu = bool(getattr(enc, "FrameTilingExtendedPaletteUsage", False))
        for which in (0, 1, 2) if epu else (0, 1):
            dist.all_reduce(enc.DeviceArray(which), op=dist.ReduceOp.MAX, group=group)
        if motion:
            for which in (3, 4, 5):  # IsPredicted, PredictedX/Y of the redo: everyone else 0
                dist.all_reduce(enc.DeviceArray(which), op=dist.ReduceOp.SUM, group=group)
        enc.SyncTileMap()
    enc.Run(S.esReindex)
