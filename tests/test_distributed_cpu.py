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
    """Stand-in with TilingEncoder's Run/SetQueryShard/DeviceArray/SyncTileMap/KeyFrames surface, computing with the CPU
    oracle.  With motion prediction the whole clip is computed once and each rank only EXPOSES its shard (zeros / -1
    elsewhere, like the encoder), so what is tested is the orchestration: shard choice, merges, their order."""

    def __init__(self, oracle, frames, palette_count, motion_radius=0, min_s=1.0, epu=False):
        self.o, self.frames, self.pc = oracle, frames, palette_count
        self.MotionPredictRadius = motion_radius
        self.FrameTilingExtendedPaletteUsage = epu
        self.min_s = min_s
        self.shard = (0, frames.shape[0])
        self.st = {}
        self.arr = {}

    def SetQueryShard(self, first, count):
        self.shard = (first, count)

    def SetDitherShard(self, rank, world):
        self.dshard = (rank, world)

    def KeyFrames(self):
        return self.st["keyframes"]

    def _mask(self, a, fill):
        per = self.st["per"]
        f0, n = self.shard
        out = np.full_like(a, fill)
        out[f0 * per:(f0 + n) * per] = a[f0 * per:(f0 + n) * per]
        return torch.from_numpy(out)

    def Run(self, step):
        from tests import oracle_pipeline
        step = int(step)
        mp = self.MotionPredictRadius > 0
        if step == 0:
            self.st = oracle_pipeline.run(self.o, self.frames, palette_count=self.pc, min_s=self.min_s, motion_radius=self.MotionPredictRadius,
                                           epu=self.FrameTilingExtendedPaletteUsage)
        elif step == 1 and mp:
            self.arr[6] = self._mask(self.st["pm_err"].view(np.int32), 0)
            self.arr[4] = self._mask(self.st["pm_x"], 0)
            self.arr[5] = self._mask(self.st["pm_y"], 0)
        elif step == 2 and mp:  # Reduce needs the merged PredictMotion results on every rank
            assert np.array_equal(self.arr[6].numpy().view(np.uint32), self.st["pm_err"])
            assert np.array_equal(self.arr[4].numpy(), self.st["pm_x"]) and np.array_equal(self.arr[5].numpy(), self.st["pm_y"])
        elif step == 4:  # Dither: only this rank's share of the global tiles, as 32-bit words; the others 0
            px = self.st["pal_px"]
            r, w = self.dshard
            t0, t1 = px.shape[0] * r // w, px.shape[0] * (r + 1) // w
            own = np.zeros_like(px)
            own[t0:t1] = px[t0:t1]
            self.arr[7] = torch.from_numpy(own.reshape(-1).view(np.int32).copy())
        elif step == 5:  # Reconstruct: only this rank's frames
            assert np.array_equal(self.arr[7].numpy().view(np.uint8).reshape(self.st["pal_px"].shape), self.st["pal_px"]), "dither shards not merged"
            per = self.st["per"]
            f0, n = self.shard
            sl = slice(f0 * per, (f0 + n) * per)
            q = self.frames.shape[0] * per
            if mp or self.FrameTilingExtendedPaletteUsage:
                assert not mp or n == 0 or f0 in self.st["keyframes"], "a shard must start on a key frame"
                self.arr[2] = self._mask(self.st["tm_pal"].astype(np.int32), -1)
                self.arr[0] = self._mask(self.st["tm_tile_recon"], -1)
                self.arr[1] = self._mask(self.st["tm_err"].view(np.int32), -1)
                if mp:
                    self.arr[3] = self._mask(self.st["is_predicted"].astype(np.uint8), 0)
                    self.arr[4] = self._mask(self.st["pred_x"], 0)
                    self.arr[5] = self._mask(self.st["pred_y"], 0)
            else:
                self.arr[0] = torch.full((q,), -1, dtype=torch.int32)
                self.arr[1] = torch.full((q,), -1, dtype=torch.int32)
                db = self.o.features_pal(self.st["pal_px"], self.st["pal_idx"], self.st["palettes"], 1)
                qf = self.o.features_rgb(self.st["tiles"][sl], None, 1, False)
                idx, err = self.o.knn1(qf, db)
                self.arr[0][sl] = torch.from_numpy(idx)
                self.arr[1][sl] = torch.from_numpy(err.view(np.int32))
        elif step == 6:
            tm = self.arr[0].numpy()
            hist = np.bincount(tm[tm >= 0], minlength=self.st["T"]).astype(np.uint32)
            nu, rep, order, use, remap = self.o.dedup(self.st["pal_px"], hist)
            self.final = dict(T=nu, use=use, tm=np.where(tm >= 0, remap[np.maximum(tm, 0)], -1).astype(np.int32),
                              pred=self.arr[3].numpy().copy() if mp else None, px=self.arr[4].numpy().copy() if mp else None,
                              py=self.arr[5].numpy().copy() if mp else None, err=self.arr[1].numpy().copy(),
                              pal=self.arr[2].numpy().copy() if self.FrameTilingExtendedPaletteUsage else None)

    def DeviceArray(self, which):
        return self.arr[int(which)]

    def SyncTileMap(self):
        pass


def _worker(rank, world, port, q, motion_radius=0, epu=False):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from tests.oracle_binding import Oracle
    from tiler_amd import synth, distributed
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle = Oracle(os.path.join(ROOT, "oracle", "libtm_oracle.so"))
    frames = synth.video(9, 48, 32, cut=3) if motion_radius else synth.video(5, 48, 32)
    enc = OracleEncoder(oracle, frames, 2, motion_radius, 0.1 if motion_radius else 1.0, epu)
    distributed.run_all(enc, frames.shape[0], rank, world)
    q.put((rank, enc.final))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("motion_radius,epu", [(0, False), (8, False), (0, True), (8, True)])
def test_two_ranks_equal_one(oracle, motion_radius, epu):
    import torch.multiprocessing as mp
    from tests import oracle_pipeline
    from tiler_amd import synth
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, motion_radius, epu)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if motion_radius:
        exp = oracle_pipeline.run(oracle, synth.video(9, 48, 32, cut=3), palette_count=2, min_s=0.1, motion_radius=motion_radius, epu=epu)
        assert len(exp["keyframes"]) >= 3 and exp["is_predicted"].any()
    else:
        exp = oracle_pipeline.run(oracle, synth.video(5, 48, 32), palette_count=2, epu=epu)
    for rank, fin in res:
        assert fin["T"] == exp["final_T"]
        assert np.array_equal(fin["use"], exp["final_use"])
        assert np.array_equal(fin["tm"], exp["final_tm_tile"])
        assert np.array_equal(fin["err"].view(np.uint32), exp["tm_err"])
        if epu:
            assert np.array_equal(fin["pal"], exp["tm_pal"])
        if motion_radius:
            assert np.array_equal(fin["pred"].astype(bool), exp["is_predicted"])
            assert np.array_equal(fin["px"], exp["pred_x"]) and np.array_equal(fin["py"], exp["pred_y"])


def test_keyframe_shards_cover_the_clip():
    from tiler_amd.distributed import keyframe_shard, frame_shard
    for kf, n, world in [([0, 4, 8], 9, 2), ([0], 10, 4), ([0, 100, 200], 300, 8), ([0, 1, 2, 3], 4, 3)]:
        got = [keyframe_shard(kf, n, r, world) for r in range(world)]
        assert got[0][0] == 0 and sum(c for _, c in got) == n
        for (a, c), (b, _) in zip(got, got[1:]):
            assert a + c == b
        assert all(c == 0 or a in kf for a, c in got)
    assert [frame_shard(10, r, 4) for r in range(4)] == [(0, 3), (3, 3), (6, 2), (8, 2)]
