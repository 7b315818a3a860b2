"""N > 1 orchestration on CPU: two gloo ranks drive tiler_amd.distributed.run_all with an oracle-backed stand-in for the
encoder (same method surface as TilingEncoder), and the merged result must equal the single-process oracle pipeline."""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleEncoder:
    """Stand-in with TilingEncoder's Run/SetQueryShard/DeviceArray/SyncTileMap surface, computing with the CPU oracle."""

    def __init__(self, oracle, frames, palette_count):
        self.o, self.frames, self.pc = oracle, frames, palette_count
        self.shard = (0, frames.shape[0])
        self.st = {}

    def SetQueryShard(self, first, count):
        self.shard = (first, count)

    def Run(self, step):
        from tests import oracle_pipeline
        step = int(step)
        if step == 4:  # after Dither everything up to the dithered tiles exists; compute it once
            self.st = oracle_pipeline.run(self.o, self.frames, palette_count=self.pc, stop_after=None)
            per = self.st["per"]
            q = self.frames.shape[0] * per
            self.tm_tile = torch.full((q,), -1, dtype=torch.int32)
            self.tm_err = torch.full((q,), -1, dtype=torch.int32)
        elif step == 5:  # Reconstruct: only this rank's frames
            per = self.st["per"]
            f0, n = self.shard
            sl = slice(f0 * per, (f0 + n) * per)
            db = self.o.features_pal(self.st["pal_px"], self.st["pal_idx"], self.st["palettes"], 1)
            qf = self.o.features_rgb(self.st["tiles"][sl], None, 1, False)
            idx, err = self.o.knn1(qf, db)
            self.tm_tile[sl] = torch.from_numpy(idx)
            self.tm_err[sl] = torch.from_numpy(err.view(np.int32))
        elif step == 6:
            hist = np.bincount(self.tm_tile.numpy(), minlength=self.st["T"]).astype(np.uint32)
            nu, rep, order, use, remap = self.o.dedup(self.st["pal_px"], hist)
            self.final = dict(T=nu, use=use, tm=remap[self.tm_tile.numpy()].astype(np.int32))

    def DeviceArray(self, which):
        return self.tm_tile if which == 0 else self.tm_err

    def SyncTileMap(self):
        pass


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from tests.oracle_binding import Oracle
    from tiler_amd import synth, distributed
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle = Oracle(os.path.join(ROOT, "oracle", "libtm_oracle.so"))
    frames = synth.video(5, 48, 32)
    enc = OracleEncoder(oracle, frames, 2)
    distributed.run_all(enc, frames.shape[0], rank, world)
    q.put((rank, enc.final["T"], enc.final["use"], enc.final["tm"]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one(oracle):
    import torch.multiprocessing as mp
    from tests import oracle_pipeline
    from tiler_amd import synth
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp = oracle_pipeline.run(oracle, synth.video(5, 48, 32), palette_count=2)
    for rank, T, use, tm in res:
        assert T == exp["final_T"]
        assert np.array_equal(use, exp["final_use"])
        assert np.array_equal(tm, exp["final_tm_tile"])
