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
    """Stand-in with TilingEncoder's Run/SetCollective/SetQueryShard/KeyFrames surface, computing with the CPU oracle and merging with
    the SAME collectives, in the same places, as the library's steps do (tm_encoder.hip: step_load, step_reduce, step_prepare_palettes,
    step_dither, step_reconstruct).  The whole clip is computed once per rank for reference; what a rank EXPOSES or feeds into a
    collective is only its shard, so what is tested is the orchestration: shard choice, the four collective kinds, their order."""

    def __init__(self, oracle, frames, palette_count, motion_radius=0, min_s=1.0, epu=False):
        self.o, self.frames, self.pc = oracle, frames, palette_count
        self.MotionPredictRadius = motion_radius
        self.FrameTilingExtendedPaletteUsage = epu
        self.min_s = min_s
        self.shard = (0, frames.shape[0])
        self.rank, self.world, self.coll = 0, 1, None
        self.st = {}
        self.arr = {}

    def SetCollective(self, rank, world, coll):
        self.rank, self.world, self.coll = rank, world, coll

    def SetQueryShard(self, first, count):
        self.shard = (first, count)

    def KeyFrames(self):
        return self.st["keyframes"]

    def _own(self, a, fill, shard=None):
        per = self.st["per"]
        f0, n = shard or self.shard
        out = np.full_like(a, fill)
        out[f0 * per:(f0 + n) * per] = a[f0 * per:(f0 + n) * per]
        return torch.from_numpy(out)

    def _share(self, n):
        base, rem = divmod(n, self.world)
        lo = self.rank * base + min(self.rank, rem)
        return lo, lo + base + (1 if self.rank < rem else 0)

    def Run(self, step):
        from tests import oracle_pipeline
        step = int(step)
        mp = self.MotionPredictRadius > 0
        st, co, o = self.st, self.coll, self.o
        if step == 0:
            self.st = st = oracle_pipeline.run(o, self.frames, palette_count=self.pc, min_s=self.min_s, motion_radius=self.MotionPredictRadius,
                                               epu=self.FrameTilingExtendedPaletteUsage)
            if self.world > 1 and not mp:  # sharded Load: own frames' correlation and mirror flags, merged with SUM (others hold 0)
                f0, n = self.shard
                correl = np.zeros_like(st["correl"])
                correl[f0:f0 + n] = st["correl"][f0:f0 + n]
                got = co.allreduce_sum(torch.from_numpy(correl.view(np.int32).copy())).numpy().view(np.float32)
                assert np.array_equal(got, st["correl"])
                flags = self._own(np.pad(st["flags"], (0, (-len(st["flags"])) % 4)), 0, (f0, n))  # bytes merged as 32-bit words
                flags = flags if len(st["flags"]) % 4 == 0 else torch.from_numpy(np.pad(self._own(st["flags"], 0).numpy(), (0, (-len(st["flags"])) % 4)))
                merged = co.allreduce_sum(torch.from_numpy(flags.numpy().view(np.int32).copy())).numpy().view(np.uint8)[: len(st["flags"])]
                assert np.array_equal(merged, st["flags"])
        elif step == 1 and mp:
            self.arr[6] = self._own(st["pm_err"].view(np.int32), 0)
            self.arr[4] = self._own(st["pm_x"], 0)
            self.arr[5] = self._own(st["pm_y"], 0)
            if self.world > 1:
                co.allreduce_sum(self.arr[6])
                for k in (4, 5):  # int8 arrays go as 32-bit words
                    a = np.pad(self.arr[k].numpy(), (0, (-self.arr[k].numel()) % 4))
                    self.arr[k] = torch.from_numpy(co.allreduce_sum(torch.from_numpy(a.view(np.int32).copy())).numpy().view(np.int8)[: len(st["pm_x"])].copy())
        elif step == 2:
            if mp:  # Reduce needs the merged PredictMotion results on every rank
                assert np.array_equal(self.arr[6].numpy().view(np.uint32), st["pm_err"])
                assert np.array_equal(self.arr[4].numpy(), st["pm_x"]) and np.array_equal(self.arr[5].numpy(), st["pm_y"])
            elif self.world > 1:
                # sharded Reduce: dedup of the own frames' tiles, all-gather of the distinct ones (tile | use | flags), dedup of the union
                per = st["per"]
                f0, n = self.shard
                sl = slice(f0 * per, (f0 + n) * per)
                lnu, _, lorder, luse, lremap = o.dedup(st["tiles"][sl], None)
                rec = np.zeros((lnu, 66), np.uint32)
                rec[:, :64] = st["tiles"][sl][lorder]
                rec[:, 64] = luse
                rec[:, 65] = st["flags"][sl][lorder]
                union, counts = co.allgather_var(torch.from_numpy(rec.view(np.int32)))
                union = union.numpy().view(np.uint32)
                nu, _, order, use, remap = o.dedup(np.ascontiguousarray(union[:, :64]), np.ascontiguousarray(union[:, 64]))
                T = min(nu, st["T"]) if nu >= st["T"] else nu
                assert T == st["T"]
                assert np.array_equal(union[order[:T], :64], st["gtiles"]) and np.array_equal(use[:T], st["guse"])
                assert np.array_equal(union[order[:T], 65].astype(np.uint8), st["gflags"])
                off = sum(counts[: self.rank])
                mine = remap[off + lremap]
                assert np.array_equal(np.where(mine < T, mine, -1), st["tm_tile_reduce"][sl])
        elif step == 3 and self.world > 1:
            # PreparePalettes: palette indices of the own share of the tiles all-gathered; palette rows of the own palettes merged with SUM.
            # (The data-parallel Lloyd itself runs in the library; here the shares of its RESULT travel the same way.)
            t0, t1 = self._share(st["T"])
            allidx, _ = co.allgather_var(torch.from_numpy(st["pal_idx"][t0:t1].astype(np.int32)))
            assert np.array_equal(allidx.numpy(), st["pal_idx"])
            rows = st["palettes"].copy()
            rows[[p for p in range(rows.shape[0]) if p % self.world != self.rank]] = 0
            assert np.array_equal(co.allreduce_sum(torch.from_numpy(rows.reshape(-1))).numpy().reshape(rows.shape), st["palettes"])
            cnt = torch.from_numpy(np.bincount(st["pal_idx"][t0:t1], minlength=self.pc).astype(np.int64))
            assert np.array_equal(co.allreduce_sum(cnt).numpy(), np.bincount(st["pal_idx"], minlength=self.pc))  # the int64 kind
        elif step == 4:  # Dither: only this rank's share of the global tiles, as 32-bit words; the others 0
            px = st["pal_px"]
            t0, t1 = self._share(px.shape[0])
            own = np.zeros_like(px)
            own[t0:t1] = px[t0:t1]
            self.arr[7] = torch.from_numpy(own.reshape(-1).view(np.int32).copy())
            if self.world > 1:
                co.allreduce_sum(self.arr[7])
        elif step == 5:  # Reconstruct: database rows per share + all-gather; only this rank's frames matched; merges
            assert np.array_equal(self.arr[7].numpy().view(np.uint8).reshape(st["pal_px"].shape), st["pal_px"]), "dither shards not merged"
            per = st["per"]
            f0, n = self.shard
            sl = slice(f0 * per, (f0 + n) * per)
            q = self.frames.shape[0] * per
            db_full = o.features_pal(st["pal_px"], st["pal_idx"], st["palettes"], 1)
            if self.world > 1:
                t0, t1 = self._share(st["T"])
                part = o.features_pal(st["pal_px"][t0:t1], st["pal_idx"][t0:t1], st["palettes"], 1)
                db, _ = co.allgather_var(torch.from_numpy(part))
                db = db.numpy()
                assert np.array_equal(db, db_full)
            else:
                db = db_full
            if mp or self.FrameTilingExtendedPaletteUsage:
                assert not mp or n == 0 or f0 in st["keyframes"], "a shard must start on a key frame"
                self.arr[2] = self._own(st["tm_pal"].astype(np.int32), -1)
                self.arr[0] = self._own(st["tm_tile_recon"], -1)
                self.arr[1] = self._own(st["tm_err"].view(np.int32), 0)
                if mp:
                    self.arr[3] = self._own(st["is_predicted"].astype(np.uint8), 0)
                    self.arr[4] = self._own(st["pred_x"], 0)
                    self.arr[5] = self._own(st["pred_y"], 0)
            else:
                self.arr[0] = torch.full((q,), -1, dtype=torch.int32)
                self.arr[1] = torch.zeros((q,), dtype=torch.int32)
                qf = o.features_rgb(st["tiles"][sl], None, 1, False)
                idx, err = o.knn1(qf, db)
                self.arr[0][sl] = torch.from_numpy(idx)
                self.arr[1][sl] = torch.from_numpy(err.view(np.int32))
            if self.world > 1:
                co.allreduce_max(self.arr[0])
                co.allreduce_sum(self.arr[1])   # errors are arbitrary 32-bit patterns: owner's value + zeros
                if self.FrameTilingExtendedPaletteUsage:
                    co.allreduce_max(self.arr[2])
                if mp:
                    for k in (3, 4, 5):
                        a = np.pad(self.arr[k].numpy(), (0, (-self.arr[k].numel()) % 4))
                        dt = a.dtype
                        self.arr[k] = torch.from_numpy(co.allreduce_sum(torch.from_numpy(a.view(np.int32).copy())).numpy().view(dt)[:q].copy())
        elif step == 6:
            tm = self.arr[0].numpy()
            hist = np.bincount(tm[tm >= 0], minlength=st["T"]).astype(np.uint32)
            nu, rep, order, use, remap = o.dedup(st["pal_px"], hist)
            self.final = dict(T=nu, use=use, tm=np.where(tm >= 0, remap[np.maximum(tm, 0)], -1).astype(np.int32),
                              pred=self.arr[3].numpy().copy() if mp else None, px=self.arr[4].numpy().copy() if mp else None,
                              py=self.arr[5].numpy().copy() if mp else None, err=self.arr[1].numpy().copy(),
                              pal=self.arr[2].numpy().copy() if self.FrameTilingExtendedPaletteUsage else None,
                              calls=dict(self.coll.calls) if self.coll else {})


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
        if not motion_radius:  # every collective kind was exercised: SUM i32, MAX i32, SUM i64, all-gather
            assert all(fin["calls"][k] > 0 for k in (0, 1, 2, 3)), fin["calls"]
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


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the driver's command form) must start two rank processes itself and hand their
    outcome back: here, without a GPU, both ranks refuse loudly (libtilemotion has no CPU path), the launcher reports the failure and the
    parent exits non-zero with no JSON line on stdout."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("covered on the GPU by tests/test_gpu_rehearsal.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode != 0
    assert p.stdout.strip() == ""
    # (a refusal per rank, unless the launcher has already stopped the second rank when the first one failed)
    assert p.stderr.count("bench.py needs an MI355X") >= 1, p.stderr[-2000:]
    assert "torch.distributed" in p.stderr or "ChildFailedError" in p.stderr or "exitcode" in p.stderr, p.stderr[-2000:]
