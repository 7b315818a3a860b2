"""One process per GPU orchestration of TTilingEncoder.Run(esAll) (tilingencoder.pas:5529-5554).

Where the path shards, with `world` processes on one node and RCCL over xGMI underneath torch.distributed:

  Load            frames are independent (1293-1411): every process loads its own contiguous frame range (+ the frame before it, for the
                  first correlation); mirror flags and Pearson sums are merged with all-reduce(SUM) (owner holds the value, others 0).
  Reduce          exact dedup of the process's own frame tiles, ALL-GATHER of every process's distinct tiles (tile, use count, mirror flags),
                  exact dedup of the union on every process (identical result everywhere).
  PreparePalettes tile -> palette clustering as data-parallel Lloyd over each process's share of the global tiles: farthest-first picks
                  settled by an all-gather of one candidate per process, one all-reduce(SUM) of the exact integer sums + counts per
                  iteration; palette colours by palette (independent tasks, 1864), assembled with an all-reduce(SUM); the host search of
                  OptimizePalettes is replicated (P x 16 colours).
  Dither          by share of the global tiles (one DitherTile per tile, 2690), all-reduce(SUM) of the 64-byte index tiles.
  Reconstruct     database rows (int16 features of the dithered tiles) built per share and ALL-GATHERED (T x 384 bytes: the north star's
                  all-gather); every process then matches ITS frames against the whole database; the items are merged by all-reduce.
  Reindex         replicated on the merged tile maps (small).

The merges live INSIDE the library's steps (tm_set_collective): this module hands the encoder a callback that runs each collective
with torch.distributed and keeps the shard arithmetic.  With motion prediction on, Load and Reduce stay replicated (the threshold
search of Reduce needs every frame's prediction error), PredictMotion shards by frame and Reconstruct by whole key-frame groups,
the unit that chains (1496).  The same code runs under gloo on CPU tensors in tests/test_distributed_cpu.py (there with an
oracle-backed stand-in for the encoder).
"""
import contextlib
import ctypes
import os

import torch
import torch.distributed as dist

KIND_SUM_I32, KIND_MAX_I32, KIND_SUM_I64, KIND_ALLGATHER = 0, 1, 2, 3
_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64)


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
    return ("%d ranks, one per GPU: Load / Reduce / PreparePalettes / Dither / Reconstruct sharded (frames; local dedup + all-gather of distinct tiles; "
            "data-parallel Lloyd with an all-reduce per iteration + palette-parallel quantisation; global tiles; all-gather of the database rows + "
            "query frames); Reindex and the host search of OptimizePalettes replicated") % world


def _backend(group):
    """'nccl' / 'gloo' / '' (an in-process stand-in for torch.distributed has no notion of one)"""
    try:
        return str(dist.get_backend(group))
    except (AttributeError, RuntimeError, ValueError):
        return ""


class Collective:
    """The collectives the library asks for, over a torch.distributed group (NCCL = RCCL on ROCm, or gloo on CPU tensors)."""

    def __init__(self, rank, world, group=None):
        self.rank, self.world, self.group = rank, world, group
        self.calls = {KIND_SUM_I32: 0, KIND_MAX_I32: 0, KIND_SUM_I64: 0, KIND_ALLGATHER: 0}
        self.bytes = 0
        self.log = [] if os.environ.get("TM_COLL_DEBUG") else None  # (kind, bytes) of every call the library made
        self._cb = _CB(self._from_library)  # kept alive with the object
        self._stream = None  # the encoder's stream as a torch stream, when the collectives are queued on it (bind_stream)

    def bind_stream(self, stream_ptr):
        """Called by TilingEncoder.SetCollective with the library's HIP stream.  With RCCL underneath (backend nccl) every collective
        is issued with that stream current: torch orders the RCCL kernel after what the library queued and the library's next
        kernel after the RCCL kernel, and nobody blocks the host.  Returns whether that mode is on (gloo stages through the host
        and keeps the blocking contract; TM_COLL_BLOCKING=1 forces it)."""
        self._stream = None
        if os.environ.get("TM_COLL_BLOCKING") == "1" or not stream_ptr or _backend(self.group) != "nccl":
            return False
        self._stream = torch.cuda.ExternalStream(int(stream_ptr))
        return True

    # ---- tensor level (also used by the CPU stand-in of the tests)
    def allreduce_sum(self, t):
        self.calls[KIND_SUM_I64 if t.dtype == torch.int64 else KIND_SUM_I32] += 1
        self.bytes += t.numel() * t.element_size()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def allreduce_max(self, t):
        self.calls[KIND_MAX_I32] += 1
        self.bytes += t.numel() * t.element_size()
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return t

    def allgather(self, send, recv=None):
        """equal-sized pieces: recv = world x send"""
        self.calls[KIND_ALLGATHER] += 1
        if recv is None:
            recv = torch.empty((self.world * send.numel(),), dtype=send.dtype, device=send.device)
        self.bytes += recv.numel() * recv.element_size()
        # as bytes: what travels is opaque to the collective (and gloo has no 16-bit integer types)
        rb, sb = recv.reshape(-1).view(torch.uint8), send.contiguous().reshape(-1).view(torch.uint8)
        if sb.is_cuda and _backend(self.group) == "gloo":  # gloo gathers host tensors only (rehearsals of the GPU path without RCCL)
            host = torch.empty(rb.shape, dtype=torch.uint8)
            dist.all_gather_into_tensor(host, sb.cpu(), group=self.group)
            rb.copy_(host)
        else:
            dist.all_gather_into_tensor(rb, sb, group=self.group)
        return recv

    def allgather_var(self, send):
        """pieces of different lengths along dim 0 -> their concatenation in rank order (what the library's gather_var does)"""
        n = torch.tensor([send.shape[0]], dtype=torch.int64, device=send.device)
        counts = self.allgather(n).tolist()
        mx = max(counts)
        if mx == 0:
            return send[:0], counts
        pad = torch.zeros((mx,) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        pad[: send.shape[0]] = send
        got = self.allgather(pad).reshape((self.world, mx) + tuple(send.shape[1:]))
        return torch.cat([got[r, : counts[r]] for r in range(self.world)]), counts

    # ---- pointer level: what tm_set_collective calls
    @staticmethod
    def _view(ptr, count, typestr):
        class _V:
            __cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(ptr), False), "version": 2}
        return torch.as_tensor(_V(), device="cuda")

    def _from_library(self, user, kind, buf, recv, count):
        if self.log is not None:
            self.log.append((kind, int(count) * (self.world if kind == KIND_ALLGATHER else (8 if kind == KIND_SUM_I64 else 4))))
        try:
            with torch.cuda.stream(self._stream) if self._stream is not None else contextlib.nullcontext():
                if kind == KIND_SUM_I32:
                    self.allreduce_sum(self._view(buf, count, "<i4"))
                elif kind == KIND_MAX_I32:
                    self.allreduce_max(self._view(buf, count, "<i4"))
                elif kind == KIND_SUM_I64:
                    self.allreduce_sum(self._view(buf, count, "<i8"))
                elif kind == KIND_ALLGATHER:
                    self.allgather(self._view(buf, count, "|u1"), self._view(recv, count * self.world, "|u1"))
                else:
                    return -1
                if self._stream is None:
                    # blocking contract: torch's stream is not the library's, the result must be in place before the library's next kernel reads it
                    torch.cuda.current_stream().synchronize()
            return 0
        except Exception as exc:  # noqa: BLE001  (must not unwind through the C frame)
            import traceback
            traceback.print_exc()
            self.error = exc
            if self.world > 1 and _backend(self.group) in ("nccl", "gloo") and os.environ.get("TM_COLL_KEEP_ALIVE") != "1":
                # the other ranks are inside (or about to enter) this very collective and would wait for this rank until the group's
                # timeout: leaving at once breaks their connections, which they report as errors of their own
                import sys
                sys.stderr.write("[tiler_amd.distributed] rank %d: collective failed, leaving the job\n" % self.rank)
                sys.stderr.flush()
                os._exit(70)
            return -2

    @property
    def callback(self):
        return self._cb


def run_all(enc, nframes, rank=0, world=1, group=None, before_reindex=None):
    """Run(esAll) over `world` processes.  `enc` needs Run/SetCollective/SetQueryShard/KeyFrames and the MotionPredictRadius
    setting (TilingEncoder or a stand-in).  before_reindex (optional): called on every rank between Reconstruct and Reindex, when the
    tile maps still index the dithered global tiles (bench.py's parity gate)."""
    from .encoder import TEncoderStep as S
    coll = getattr(enc, "_collective", None)
    native = getattr(enc, "_native_comm", None)  # TilingEncoder.CommInit: the library's own RCCL communicator carries the merges
    if native is not None:
        assert native == (rank, world), "the encoder's native communicator is rank %d of %d" % native
    elif coll is None or coll.world != world or coll.rank != rank or coll.group is not group:
        coll = Collective(rank, world, group)
        enc.SetCollective(rank, world, coll)  # world == 1: plain single-process run
        enc._collective = coll
    motion = int(enc.MotionPredictRadius) > 0
    first, count = frame_shard(nframes, rank, world)
    enc.SetQueryShard(first, count)
    stats = getattr(enc, "CollectiveStats", None) if world > 1 else None
    by_step = getattr(enc, "_coll_bytes_by_step", None)
    if stats is not None and by_step is None:
        by_step = enc._coll_bytes_by_step = {}

    def run(step):
        before = stats()["bytes"] if stats is not None else 0
        enc.Run(step)
        if stats is not None:
            by_step[step.name] = by_step.get(step.name, 0) + stats()["bytes"] - before

    run(S.esLoad)
    run(S.esPredictMotion)  # frames are independent (each is searched in the source pixels of its neighbour)
    run(S.esReduce)
    run(S.esPreparePalettes)
    run(S.esDither)
    if motion:
        first, count = keyframe_shard(enc.KeyFrames(), nframes, rank, world)
        enc.SetQueryShard(first, count)
    run(S.esReconstruct)
    if before_reindex is not None:
        before_reindex()
    run(S.esReindex)
