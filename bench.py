#!/usr/bin/env python3
"""bench.py -- BASELINE.json headline metric on MI355X: encoded frames/sec (+ tiles-matched/sec) of the TileMotion
per-frame tile pipeline on the 720p, 300-frame, 16-palette synthetic clip (configs[1]), generated exactly as SURVEY.md 8(d) writes it.

A step = one TTilingEncoder.Run(esAll) pass (Load -> Reduce -> PreparePalettes -> Dither -> Reconstruct -> Reindex)
over the whole clip.  `value` is measured with the RGB frames already resident in HBM; `with_h2d_d2h` repeats the same K steps with
the clip in page-locked HOST memory and the results read back (tile maps of every frame, tiles, palettes), so that every step also
moves the clip across PCIe and its results back (SURVEY.md 8d's "H2D/D2H included" reading); `with_h2d_overlapped_d2h` does the same
with the next clip's upload queued beside this clip's steps.
N > 1: one process per GPU (torch.distributed, RCCL); the clip is ONE job split over the ranks (strong scaling), see
tiler_amd/distributed.py.  `python bench.py --gpus N` with no launcher around it starts the N ranks itself (torch.distributed.run as a
child process, before this process touches a GPU) and relays rank 0's line.  Prints one JSON line.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

I8_DENSE_PEAK_TOPS = 5000.0  # MI355X_MICROARCH.md: bf16 dense ~2.5 PF, i8 MFMA = 2x bf16 per clock
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s (6.3 TB/s achievable)
INT32_VALU_PEAK_TOPS = 256 * 4 * 32 * 2.4e9 / 1e12  # one int32 operation per lane and clock: 78.6 Tops/s
STAGES = ["load", "predict_motion", "reduce", "prepare_palettes", "dither", "reconstruct", "reindex", "save"]


def _traffic_from_profiles(kernel_build):
    """HBM-side bytes per launch of the dominant kernel, from the committed PMC passes (profiles/r*_pmc_knn_traffic.json, newest round
    first), only when they were taken on this build of the kernel; null otherwise -- the figure is never a constant of this file."""
    import glob
    seen = []
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_knn_traffic.json")), reverse=True):
        try:
            with open(p) as f:
                t = json.load(f)
        except (OSError, ValueError):
            continue
        if t.get("kernel_build") == kernel_build:
            _traffic_from_profiles.extra = {k: t[k] for k in ("mfma_busy_frac", "mfma_busy_note", "tcc_hit_rate_pruned") if k in t}
            return t["traffic_bytes"], "2 x FETCH_SIZE + WRITE_SIZE of one pruned launch, separate --pmc passes (%s)" % t.get("source", "profiles/")
        seen.append("%s: build %r" % (os.path.basename(p), t.get("kernel_build")))
    return None, "no PMC pass committed for this build %r (%s)" % (kernel_build, "; ".join(seen) or "none in profiles/")


_traffic_from_profiles.extra = {}


def _host_threads(visible):
    """threads of the all-cores legs = what this process may really use: TM_BENCH_THREADS if set, else the cgroup's CPU quota if there
    is one, else the affinity mask capped at 16 (a one-GPU box of the pool exposes every CPU of the node in its mask but owns a
    16-CPU share of it)"""
    if os.environ.get("TM_BENCH_THREADS"):
        return max(1, int(os.environ["TM_BENCH_THREADS"]))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            return max(1, min(visible, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(visible, 16))


def _collective_counts(enc, nsteps):
    """calls and bytes of this rank's collectives per step, counted by the library (tm_get_collective_stats) on whichever path carries
    them: its own RCCL communicator (tm_comm_init) or the host's callback (tm_set_collective)"""
    st = enc.CollectiveStats()
    out = {k: v / float(nsteps) for k, v in st.items()}
    by_step = getattr(enc, "_coll_bytes_by_step", None)
    if by_step:
        out["bytes_by_stage"] = {k: v / float(nsteps) for k, v in by_step.items() if v}
    out["path"] = "native RCCL inside libtilemotion (tm_comm_init)" if getattr(enc, "_native_comm", None) else "host callback (tm_set_collective) over torch.distributed"
    coll = getattr(enc, "_collective", None)
    if coll is not None and coll.log is not None:
        per = len(coll.log) // nsteps
        out["last_step_calls"] = coll.log[-per:]
    return out


def _spawn_ranks(n):
    """`--gpus N` without a launcher: N fresh rank processes through torch.distributed.run, started from this process BEFORE it has
    made any GPU call (never an exec: a process that has initialised the GPU must not be replaced).  Rank 0's JSON line is relayed on
    stdout, everything else goes to stderr; the exit code is the launcher's (non-zero if any rank failed)."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver of this pool only does dmabuf IPC
    # the launcher picks the rendezvous port itself (c10d store on 127.0.0.1, port 0 = any free one): a port found here by bind-and-close
    # could be taken by another process before the launcher binds it
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n, "--rdzv-backend=c10d", "--rdzv-endpoint=127.0.0.1:0",
           "--local-addr", "127.0.0.1", os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in proc.stdout:
        if ln.startswith('{"metric"'):
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    return rc if rc != 0 else (0 if line is not None else 1)


def cpu_baseline(width, height, nframes, palette_count, t_global, t_distinct, seconds_budget=24.0, threads=None):
    """The oracle (CPU restatement, kind 'port') on a bounded sample of the same workload, scaled to frames/s.  Three legs:
    1 thread with the brute-force search; every host core (one oracle call per thread: ctypes drops the GIL) with the brute force;
    every host core with the exact kd-tree (bucket 32) the reference searches with.  `value` is the fastest of them."""
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    so = os.path.join(ROOT, "oracle", "libtm_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libtm_oracle.so"])
    from tests.oracle_binding import Oracle
    from tiler_amd import synth
    o = Oracle(so)
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = threads or _host_threads(visible)
    tm_w, tm_h = (width - 1) // 8 + 1, (height - 1) // 8 + 1
    per = tm_w * tm_h
    fr = synth.video(2, width, height, freeze=False)
    rng = np.random.default_rng(0)

    def par(fn, parts, threads):
        if threads == 1:
            t0 = time.time()
            out = [fn(p) for p in parts]
            return time.time() - t0, out
        with ThreadPoolExecutor(threads) as ex:
            t0 = time.time()
            out = list(ex.map(fn, parts))
            return time.time() - t0, out

    # one frame: tile scatter + Lab means + mirrors (per frame, serial per frame in the reference: one thread per frame)
    t0 = time.time()
    tiles = o.load_from_image(fr[0], tm_w, tm_h)
    o.inter_frame_data(tiles)
    canon, flags = o.canonicalise(tiles)
    t_load = time.time() - t0
    # query features of one frame's tiles
    feat_parts = np.array_split(canon[: per // 4], max(cores, 1))
    t_feat1, _ = par(lambda p: o.features_rgb(p, None, 1, False), [canon[: per // 16]], 1)
    t_feat1 *= 16
    t_featn, qparts = par(lambda p: o.features_rgb(p, None, 1, False), feat_parts, cores)
    t_featn *= 4
    qf = np.concatenate(qparts)
    # global-tile stages, paid once per clip for T tiles: cluster features + Thomas-Knoll dither + database features
    gs = canon[:512]
    pal_idx = rng.integers(0, palette_count, size=gs.shape[0]).astype(np.int32)
    palettes = rng.integers(0, 1 << 24, size=(palette_count, 16)).astype(np.int32)

    def global_stage(sl):
        o.features_cluster(gs[sl], 4)
        pp = o.dither(gs[sl], flags[:512][sl], pal_idx[sl], palettes, True)
        return o.features_pal(pp, pal_idx[sl], palettes, 1)
    t_g1, _ = par(global_stage, [slice(0, 64)], 1)
    t_g1 = t_g1 / 64
    gparts = [slice(i * 512 // cores, (i + 1) * 512 // cores) for i in range(cores)]
    t_gn, dbparts = par(global_stage, gparts, cores)
    t_gn = t_gn / 512
    db_small = np.concatenate(dbparts)
    # exact dedup of the clip's frame tiles: comparison sort of 256-byte keys, single-threaded in the reference (TFPList.Sort)
    ds = np.concatenate([canon, canon[: per // 2]])
    t0 = time.time()
    o.dedup(ds, None)
    n_s, n_full = ds.shape[0], per * nframes
    t_dedup_clip = (time.time() - t0) * (n_full * np.log2(n_full)) / (n_s * np.log2(n_s))
    # KNN against the distinct database rows (what both the reference's tree and the GPU scan search)
    db = rng.integers(-300, 300, size=(t_distinct, 192)).astype(np.int16)
    db[:, 0] = rng.integers(0, 13000, size=t_distinct)
    db[: db_small.shape[0]] = db_small
    nq1 = 16
    t_knn1, _ = par(lambda q: o.knn1(q, db), [qf[:nq1]], 1)
    nqn = 16 * cores
    t_knnn, _ = par(lambda q: o.knn1(q, db), np.array_split(qf[:nqn], cores), cores)
    t0 = time.time()
    tree = o.kdtree_build(db, 32)
    t_tree_build = time.time() - t0
    t_kd, res = par(lambda q: o.kdtree_search1(tree, q), np.array_split(qf[:nqn], cores), cores)
    visited = sum(r[2] for r in res)
    o.kdtree_free(tree)

    def fps(load, feat, knn_per_query, glob_per_tile, extra_clip=0.0):
        return 1.0 / (load + feat + knn_per_query * per + (glob_per_tile * t_global + t_dedup_clip + extra_clip) / nframes)
    one = fps(t_load, t_feat1, t_knn1 / nq1, t_g1)
    allb = fps(t_load / cores, t_featn, t_knnn / nqn, t_gn)          # frames load on separate threads in the reference (1326)
    allk = fps(t_load / cores, t_featn, t_kd / nqn, t_gn, t_tree_build)
    best = max(allb, allk)
    return {
        "value": best, "unit": "frames/s", "cores": cores, "kind": "port", "host_cpus_visible": visible,
        "cores_rule": "threads of the all-cores legs = TM_BENCH_THREADS if set, else the cgroup CPU quota of this process, else min(CPUs in the affinity mask, 16): "
                      "a one-GPU box of the pool shows every CPU of its 8-GPU host in the mask but owns a 16-CPU share of it",
        "sample": (f"oracle (C restatement, gcc -O3) timed on: 1 frame load+Lab+mirrors, {per // 4} query feature vectors, 512 global tiles "
                   f"(cluster features + Thomas-Knoll dither + database features), exact dedup of {n_s} tiles (scaled n log n, one thread as "
                   f"TFPList.Sort), KNN of {nqn} queries x {t_distinct} distinct rows; scaled linearly to {per} tiles/frame and {t_global} "
                   f"global tiles; k-means stages excluded (favours the CPU)"),
        "legs": {"one_thread_brute_force": one, "all_cores_brute_force": allb, "all_cores_kdtree_bucket32": allk},
        "kdtree": {"build_s": t_tree_build, "rows_visited_fraction": visited / float(nqn * t_distinct),
                   "note": "exact kd-tree (ANN's published standard split and search, eps 0): in 192 dimensions it degenerates towards a brute force"},
        "tiles_matched_per_sec": nqn / min(t_knnn, t_kd),
    }


def parity_gate(enc, frames, nframes, tm_w, tm_h, oracle_so):
    """BASELINE.md section 3's gate, on the clip and the encoder state the timed steps ran on, called between Reconstruct and Reindex of one
    more (untimed) step: (1) 4 096 queries of four frames against an exact fp64 scan of the whole database (integer-valued doubles: the matmul
    is exact) -- tile index (lowest among equals) and error; (2) with the CPU baseline's oracle at hand, 192 global tiles' dither indices against
    the oracle's Thomas-Knoll.  Returns a dict; "passed" is False on the first mismatch."""
    import torch
    from tiler_amd import stages
    res = {"passed": True, "knn_sample": None, "dither_sample": None}
    per = tm_w * tm_h
    hdr, pal_px, rgb = enc.Tiles()
    pals = enc.Palettes()
    pal_idx = hdr["PalIdx_Initial"].astype(np.int32)
    T = pal_px.shape[0]
    gen = torch.Generator(device="cuda").manual_seed(20251005)
    db = stages.features_pal(torch.from_numpy(pal_px).cuda(), torch.from_numpy(pal_idx).cuda(), torch.from_numpy(pals).cuda(), 1)
    dbd = db.to(torch.float64)
    dn = (dbd * dbd).sum(1)
    checked = wrong = 0
    kinds = {"tile_index": 0, "psnr": 0, "palette_of_tile": 0}
    first_bad = None
    for f in torch.randint(0, nframes, (4,), generator=gen, device="cuda").tolist():
        ft, _, _ = stages.load(frames[f:f + 1], tm_w, tm_h)
        qf = stages.features_rgb(ft, None, 1, False)
        pick = torch.randperm(per, generator=gen, device="cuda")[:1024]
        tmap = enc.TileMap(f)
        got_idx = torch.from_numpy(tmap["TileIdx"].astype(np.int64)).cuda()[pick]
        got_psnr = tmap["PSNR"][pick.cpu().numpy()]
        for s0 in range(0, pick.shape[0], 256):
            q = qf[pick[s0:s0 + 256]].to(torch.float64)
            d = (q * q).sum(1)[:, None] + dn[None, :] - 2.0 * (q @ dbd.T)
            m = d.min(1).values
            first = (d == m[:, None]).to(torch.uint8).argmax(1)  # lowest index among the minima
            bad_i = first != got_idx[s0:s0 + 256]
            kinds["tile_index"] += int(bad_i.sum().item())
            if first_bad is None and bool(bad_i.any()):
                j = int(torch.nonzero(bad_i)[0].item())
                gi = int(got_idx[s0 + j].item())
                first_bad = {"frame": f, "position": int(pick[s0 + j].item()), "got_tile": gi, "want_tile": int(first[j].item()), "want_err": float(m[j].item()),
                             "err_of_got_tile": float(d[j, gi].item()) if 0 <= gi < T else None}
            # EuclideanToPSNR, utils.pas:1074-1078: the mean error as a Single, the logarithm in double, the result a Single
            r32 = (m.cpu().numpy() * (1.0 / 192)).astype(np.float32)
            want = (10.0 * np.log10(255.0 * 255.0 / np.maximum(np.float32(0.5), r32).astype(np.float64))).astype(np.float32)
            kinds["psnr"] += int((~np.isclose(got_psnr[s0:s0 + 256], want, rtol=1e-6)).sum())
            checked += q.shape[0]
        kinds["palette_of_tile"] += int((tmap["PalIdx"] != pal_idx[tmap["TileIdx"]]).sum())
    wrong = sum(kinds.values())
    res["knn_sample"] = {"queries": checked, "database_rows": int(T), "mismatches": wrong, "by_kind": kinds, "first_wrong_tile": first_bad,
                         "against": "exact fp64 scan (torch matmul on integer-valued doubles), lowest index among equals"}
    res["passed"] &= wrong == 0
    del db, dbd, dn
    if oracle_so is not None:
        from tests.oracle_binding import Oracle
        o = Oracle(oracle_so)
        sample = torch.randint(0, T, (192,), generator=gen, device="cuda").cpu().numpy()
        gflags = ((hdr["Flags"] >> 3) & 3).astype(np.uint8)
        want = o.dither(rgb[sample], gflags[sample], pal_idx[sample], pals, True)
        bad = int((pal_px[sample] != want).any(axis=1).sum())
        res["dither_sample"] = {"tiles": 192, "mismatches": bad, "against": "the oracle's Thomas-Knoll (DeviseBestMixingPlanThomasKnoll, tilingencoder.pas:2565-2612) on the same tiles and palettes"}
        res["passed"] &= bad == 0
    else:
        res["dither_sample"] = "skipped: the oracle is only loaded where the CPU baseline runs (rank 0 of a one-GPU run without --no-cpu-baseline)"
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--palettes", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--motion-radius", type=int, default=0,
                    help="MotionPredictRadius for the timed steps; 0 (default) = the headline definition of SURVEY.md 8(d): motion prediction excluded")
    ap.add_argument("--no-motion-extra", action="store_true", help="skip the untimed extra pass with MotionPredictRadius=32")
    ap.add_argument("--no-defaults-extra", action="store_true",
                    help="skip the untimed extra passes with the extended palette usage on (alone, and with motion prediction)")
    ap.add_argument("--no-h2d-extra", action="store_true", help="skip the timed regions with the clip in host memory")
    ap.add_argument("--no-kmodes-extra", action="store_true", help="skip the k-modes operator's roofline line (A17, config 5's shape)")
    ap.add_argument("--no-frozen-extra", action="store_true",
                    help="skip the extra pass on the clip with frozen tile columns (this repo's addition to the generator: exact inter-frame duplicates)")
    ap.add_argument("--frozen-columns", action="store_true",
                    help="time `value` on the clip with frozen tile columns instead of SURVEY.md 8(d)'s literal generator (A/B runs against earlier rounds)")
    ap.add_argument("--no-dense-extra", action="store_true",
                    help="skip the dense diagnostic launch of the KNN kernel (under rocprofv3 --stats it would share the kernel's row with the pruned launches)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(_spawn_ranks(args.gpus))

    import torch
    import torch.distributed as dist
    from tiler_amd import synth, distributed, stages
    from tiler_amd import lib as _lib_fn
    from tiler_amd._lib import check as _check
    from tiler_amd.encoder import TilingEncoder, TEncoderStep

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libtilemotion has no CPU path")
    # rehearsal on a one-GPU box (TM_BENCH_REHEARSE=1): every rank on cuda:0 over gloo -- the whole multi-process path except RCCL itself
    rehearse = os.environ.get("TM_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        import datetime
        tmo = datetime.timedelta(seconds=int(os.environ.get("TM_BENCH_DIST_TIMEOUT", "600")))  # a rank that dies must not leave the others waiting for ever
        if rehearse:
            dist.init_process_group("gloo", timeout=tmo)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=tmo)

    W, H, F = args.width, args.height, args.frames
    # synthetic clip, generated on the host into page-locked memory, then parked in HBM before any timing
    host_frames = torch.empty((F, H, W), dtype=torch.int32, pin_memory=True)
    rng = np.random.Generator(np.random.PCG64(synth.SEED))
    hv = host_frames.numpy()
    for f in range(F):
        hv[f] = synth.frame(f, W, H, rng, freeze=args.frozen_columns).view(np.int32)
    frames = host_frames.cuda()
    torch.cuda.synchronize()

    enc = TilingEncoder()
    enc.LoadDefaultSettings()
    enc.PaletteCount = args.palettes
    enc.PaletteSize = 16
    enc.FrameTilingExtendedPaletteUsage = False  # headline KNN number: EPU off
    enc.MotionPredictRadius = args.motion_radius
    enc.SetVideo(W, H, 24.0, F)
    enc.SetFramesDevice(frames)

    if world > 1 and not rehearse and os.environ.get("TM_BENCH_NATIVE") == "1":
        # Opt-in (the default is the host callback over torch.distributed's RCCL process group, the path every multi-process test has run):
        # RCCL linked into libtilemotion, one communicator per encoder, the collectives on the encoder's own stream.  It has only ever run
        # with one rank (tests/c/native_comm.c), so the ranks first agree that each of them is READY to enter the collective init (library
        # exports tm_comm_*, device chosen, id received) -- a rank that fails before that never leaves the others inside it -- and
        # tm_comm_init itself is non-blocking with a time limit (TM_COMM_TIMEOUT_S).
        box = [TilingEncoder.CommUniqueId() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        ready = 1 if (box[0] is not None and hasattr(_lib_fn(), "tm_comm_init")) else 0
        agreed = torch.tensor([ready], dtype=torch.int32, device="cuda")
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
        if int(agreed.item()) == 1:
            ok = 1
            try:
                enc.CommInit(box[0], rank, world)
            except Exception as ex:  # noqa: BLE001
                ok = 0
                print("[bench] rank %d: tm_comm_init failed (%s); every rank falls back to the host callback" % (rank, ex), file=sys.stderr)
            agreed = torch.tensor([ok], dtype=torch.int32, device="cuda")
            dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
            if int(agreed.item()) == 0 and ok:
                enc.CommDestroy()
        else:
            print("[bench] rank %d: not every rank is ready for tm_comm_init; the host callback carries the collectives" % rank, file=sys.stderr)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        distributed.run_all(enc, F, rank, world)

    def timed(nsteps):
        barrier()
        t0 = time.perf_counter()
        knn = dict(ms=0.0, pairs=0, launches=0, seed_ms=0.0, lists_ms=0.0, consume_ms=0.0, seed_pairs=0, consume_pairs=0, consume_mfma=0)
        stage_ms, stage_max, step_max, tl = np.zeros(8), np.zeros(8), 0.0, t0
        for _ in range(nsteps):
            step()
            ks = enc.KnnStats()
            knn["ms"] += ks["kernel_ms"]; knn["pairs"] += ks["pairs"]; knn["launches"] += ks["launches"]
            for k in ("seed_ms", "lists_ms", "consume_ms", "seed_pairs", "consume_pairs", "consume_mfma"):
                knn[k] += ks[k]
            sm = enc.StageMs()
            stage_ms += sm
            stage_max = np.maximum(stage_max, sm)
            tn = time.perf_counter()  # (Run is blocking: the step's results are on the host side of the call when it returns)
            step_max, tl = max(step_max, tn - tl), tn
        barrier()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        knn["stage_max"], knn["step_max_ms"] = stage_max, step_max * 1e3
        return float(t.item()), knn, stage_ms

    for _ in range(args.warmup):
        step()
    dt, knn, stage_ms = timed(args.steps)

    c = enc.counts()
    ks = enc.KnnStats()
    # the dominant kernel = the scan's consume kernel (k_knn_consume); its seed and list kernels are reported beside it
    nl = max(knn["launches"], 1)
    per_launch_ms = knn["consume_ms"] / nl
    alg_ops_per_launch = 384.0 * knn["consume_pairs"] / nl  # 2*192 integer ops per (query, tile) pair
    achieved = alg_ops_per_launch / (per_launch_ms * 1e-3) / 1e12 if per_launch_ms > 0 else 0.0
    # what the matrix pipe really executed: the consume kernel's 32x32x32 int8 instructions (65 536 ops each), zero chunks skipped
    mfma_tops = knn["consume_mfma"] / nl * 65536.0 / (per_launch_ms * 1e-3) / 1e12 if per_launch_ms > 0 else 0.0
    scan_ms = knn["ms"] / nl
    achieved_scan = 384.0 * knn["pairs"] / nl / (scan_ms * 1e-3) / 1e12 if scan_ms > 0 else 0.0
    ms_per_step = dt / args.steps * 1e3
    per = c["tm_w"] * c["tm_h"]
    q_total = F * per
    T = int(enc.GlobalTilingTileCount)
    kernel_build = _lib_fn().tm_version().decode()
    traffic, traffic_note = _traffic_from_profiles(kernel_build) if (W, H, F, args.palettes) == (1280, 720, 300, 16) else (None, "not the profiled workload")
    st = {n: float(v) / args.steps for n, v in zip(STAGES, stage_ms)}
    out = {
        "metric": "encoded frames/sec + tiles-matched/sec, 720p 8x8 tiles, 1/2/4/8 MI355X",
        "value": F * args.steps / dt,
        "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "i8",
        "data": "synthetic",
        "config": {"workload": f"{W}x{H} {F}-frame synthetic noise+gradients, 8x8 tiles, {args.palettes} palettes x 16 colours, "
                               f"Thomas-Knoll dither, KNN k=1 (EPU off), MotionPredictRadius={args.motion_radius}" + (" (motion prediction excluded, SURVEY.md 8d)" if args.motion_radius == 0 else "")
                               + ("; generator = SURVEY.md 8(d) as written (tiler_amd.synth.frame(freeze=False)): %.1f %% of the frame tiles are exact duplicates of another "
                                  "frame tile; `with_frozen_columns` is the same step on the clip earlier rounds were quoted on" if not args.frozen_columns else
                                  "; generator = SURVEY.md 8(d) plus ONE addition (--frozen-columns): the +f drift is frozen on every third tile column, which makes "
                                  "%.1f %% of the frame tiles exact duplicates of another frame tile") % (100.0 * (1.0 - float(ks.get("queries", q_total)) / q_total)),
                   "frames": F, "tiles_per_frame": per, "query_tiles": q_total, "global_tiles_T": T,
                   "distinct_database_rows": int(ks["db_rows"]), "knn_queries": int(ks.get("queries", q_total)),
                   "duplicate_frame_tiles_fraction": 1.0 - float(ks.get("queries", q_total)) / q_total,
                   "value_is": "device-resident: RGB frames in HBM when the timed region starts; the transfer-inclusive figures of SURVEY.md 8(d) are "
                               "`with_h2d_d2h` (upload inside Load, tile maps / tiles / palettes read back after the step) and `with_h2d_overlapped_d2h` "
                               "(upload of the next clip beside this clip's steps), details under `transfers`",
                   "final_tiles_after_reindex": int(c["tiles"]),
                   "input": "RGB frames resident in HBM when the timed region starts (with_h2d_d2h: in page-locked host memory)",
                   "parallelism": distributed.describe(world)},
        "collectives_per_step": _collective_counts(enc, args.steps + args.warmup) if world > 1 else None,
        "tiles_matched_per_sec": q_total / (st["reconstruct"] * 1e-3) if st["reconstruct"] > 0 else None,
        "stage_ms": {n: round(v, 3) for n, v in st.items()},
        "stage_ms_max": {n: round(float(v), 3) for n, v in zip(STAGES, knn["stage_max"])},
        "step_ms_max": round(knn["step_max_ms"], 3),
        "nominal_pairs": float(q_total) * float(T),
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": I8_DENSE_PEAK_TOPS, "unit": "TFLOP/s", "frac": achieved / I8_DENSE_PEAK_TOPS,
                     "traffic": traffic, "traffic_note": traffic_note, "kernel": "k_knn_consume", "kernel_build": kernel_build,
                     "launch_ms": per_launch_ms, "k_bytes": ks["k_bytes"],
                     "pairs_per_launch": knn["consume_pairs"] / nl,
                     "scan": {"kernels_ms": {"k_knn_seed": knn["seed_ms"] / nl, "k_knn_lists": knn["lists_ms"] / nl, "k_knn_consume": per_launch_ms},
                              "ms": scan_ms, "pairs": knn["pairs"] / nl, "achieved": achieved_scan, "frac": achieved_scan / I8_DENSE_PEAK_TOPS,
                              "note": "the whole search = three kernels back to back (seeds, tile lists, consume); `frac` here prices all evaluated pairs "
                                      "against the three kernels' time, the roofline line above the dominant kernel alone"},
                     "mfma_pipe_frac": mfma_tops / I8_DENSE_PEAK_TOPS, "mfma_instructions_per_launch": knn["consume_mfma"] / nl,
                     "note": "int8 ops; algorithmic = 384 ops per evaluated (query, distinct database row) pair (exact count: padding rows are not pairs); "
                             "the kernel's chain has k_bytes / 32 matrix instructions per 32 x 32 block and skips those whose high-digit chunk is all zero: mfma_pipe_frac prices the instructions it issued"},
    }
    if traffic is not None:  # what the committed PMC passes of this kernel build say beside the traffic: matrix pipe busy time, L2 hit rate
        out["roofline"].update({"pmc_" + k: v for k, v in _traffic_from_profiles.extra.items()})
    # ---- parity gate, outside the timed region: one more step on the very clip and encoder `value` was timed on, checked
    # between its Reconstruct and its Reindex.  Every rank runs the step (its collectives need them all); rank 0 checks.  A mismatch is fatal.
    gate_box = {}

    def _gate():
        if rank == 0 and args.motion_radius == 0:
            oracle_so = None
            if world == 1 and not args.no_cpu_baseline:
                import subprocess
                oracle_so = os.path.join(ROOT, "oracle", "libtm_oracle.so")
                if not os.path.exists(oracle_so):
                    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libtm_oracle.so"])
            gate_box.update(parity_gate(enc, frames, F, c["tm_w"], c["tm_h"], oracle_so))
    distributed.run_all(enc, F, rank, world, before_reindex=_gate)
    if rank == 0:
        if args.motion_radius != 0:
            out["parity_gate"] = "not run: the gate checks the KNN-only pipeline of the headline (MotionPredictRadius=0)"
        else:
            out["parity_gate"] = "passed" if gate_box.get("passed") else "FAILED"
            out["parity_gate_detail"] = {k: v for k, v in gate_box.items() if k != "passed"}
            if not gate_box.get("passed"):
                print(json.dumps(out), flush=True)
                print("[bench] PARITY GATE FAILED: %r" % (gate_box,), file=sys.stderr)
                raise SystemExit(3)
    if world == 1 and not args.no_h2d_extra:
        # SURVEY.md 8(d)'s metric as it is defined, "H2D/D2H included": the same K steps with the clip in page-locked host memory -- every
        # step is handed the clip anew (tm_set_frames_host lends it until that step's Load has returned) and its Load moves 4*W*H*F bytes
        # across PCIe, chunks beside its own kernel -- and with what the reference's consumers read after Run (tilingencoder.pas:486-568:
        # Frames[i].TileMap of every frame, Tiles, Palettes) copied back into page-locked host buffers inside the timed region.
        tm_host = torch.empty(F * per * 18, dtype=torch.uint8, pin_memory=True)
        cap_t = int(T)
        hdr_host = torch.empty(cap_t * 20, dtype=torch.uint8, pin_memory=True)
        pal_host = torch.empty(cap_t * 64, dtype=torch.uint8, pin_memory=True)
        rgb_host = torch.empty(cap_t * 256, dtype=torch.uint8, pin_memory=True)
        Lc = _lib_fn()

        def read_back():
            enc.TileMaps(0, F, out=tm_host)
            nt_ = enc.counts()["tiles"]
            _check(Lc.tm_get_tiles(ctypes.c_void_p(enc._h), 0, nt_, ctypes.c_void_p(hdr_host.data_ptr()), ctypes.c_void_p(pal_host.data_ptr()),
                                   ctypes.c_void_p(rgb_host.data_ptr())))
            enc.Palettes()
            return F * per * 18 + nt_ * 340 + args.palettes * 16 * 4

        def timed_host(nsteps, overlapped):
            if overlapped:
                enc.PrefetchFramesHost(host_frames)  # the first timed step's clip: its upload belongs to the step before it
            barrier()
            t0 = time.perf_counter()
            sm, d2h_s, d2h_b = np.zeros(8), 0.0, 0
            for _ in range(nsteps):
                enc.SetFramesHost(host_frames)
                if overlapped:
                    enc.PrefetchFramesHost(host_frames)  # the NEXT clip starts crossing PCIe now, beside this clip's steps
                step()
                sm += enc.StageMs()
                t1 = time.perf_counter()
                d2h_b += read_back()
                d2h_s += time.perf_counter() - t1
            barrier()
            return time.perf_counter() - t0, sm, d2h_s, d2h_b
        enc.SetFramesHost(host_frames)
        step()
        read_back()
        dth, sth, dh, db = timed_host(args.steps, False)
        dto, sto, doh, dob = timed_host(args.steps, True)
        out["with_h2d_d2h"] = F * args.steps / dth
        out["with_h2d_overlapped_d2h"] = F * args.steps / dto
        out["transfers"] = {
            "unit": "frames/s", "h2d_bytes_per_step": 4 * W * H * F, "d2h_bytes_per_step": db // args.steps,
            "with_h2d_d2h": {"value": F * args.steps / dth, "ms_per_step": dth / args.steps * 1e3, "load_ms": float(sth[0]) / args.steps,
                             "d2h_ms": dh / args.steps * 1e3, "d2h_gb_s": db / dh / 1e9 if dh > 0 else None,
                             "pcie_gb_s_in_load": 4.0 * W * H * F / (float(sth[0]) / args.steps * 1e-3) / 1e9 if sth[0] > 0 else None,
                             "note": "tm_set_frames_host before every step: pinned host memory -> HBM inside Load; tm_get_tilemaps (all frames, one copy), "
                                     "tm_get_tiles, tm_get_palette after it; one clip on its own; never `value`"},
            "with_h2d_overlapped_d2h": {"value": F * args.steps / dto, "ms_per_step": dto / args.steps * 1e3, "load_ms": float(sto[0]) / args.steps,
                                        "d2h_ms": doh / args.steps * 1e3,
                                        "note": "clips back to back: tm_prefetch_frames_host moves clip n+1 into a second device buffer on the copy stream "
                                                "while clip n's steps run, and its Load adopts the copies -- every timed step still moves one whole clip across "
                                                "PCIe (the one issued in the last step is drained inside the timed region) and reads its results back; never `value`"}}
        torch.cuda.synchronize()
        enc.SetFramesDevice(frames)
        del tm_host, hdr_host, pal_host, rgb_host
    if world == 1 and rank == 0:
        # the peaks measured on this very device: a bare loop of the kernel's MFMA instruction and an HBM stream triad
        tops, gbs = ctypes.c_double(), ctypes.c_double()
        _check(_lib_fn().tm_probe_mfma_i8(0.3, ctypes.byref(tops)))
        _check(_lib_fn().tm_probe_hbm_triad(1 << 30, ctypes.byref(gbs)))
        out["measured_peaks"] = {"mfma_i8_tops": tops.value, "hbm_triad_gb_s": gbs.value,
                                 "note": "bare v_mfma_i32_32x32x32_i8 loop (two waves per SIMD, operands in registers) and a = b + s*c over 3 x 1 GiB; "
                                         "roofline.frac stays against the 5000 TOP/s vendor figure"}
        out["roofline"]["frac_of_measured_peak"] = achieved / tops.value if tops.value > 0 else None
        out["roofline"]["mfma_pipe_frac_of_measured_peak"] = mfma_tops / tops.value if tops.value > 0 else None
    if world == 1:
        # per-stage rooflines: the streaming kernels timed on their own (torch events on the stream the stage seam
        # launches on) on the clip's own data; Dither and the k-means stage from the step's wall time
        def ev_time(fn, reps=5):  # the median of `reps` separately timed calls (a first call's allocations and a stray stall stay out of it)
            r = fn()
            times = []
            for _ in range(reps):
                del r
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = fn()
                e1.record()
                torch.cuda.synchronize()
                times.append(e0.elapsed_time(e1))
            return sorted(times)[len(times) // 2], r
        sr = {}
        ms, (tiles, flags, lab) = ev_time(lambda: stages.load(frames, c["tm_w"], c["tm_h"]))
        b = q_total * (256 + 256 + 1 + 12)
        sr["load"] = {"bound": "hbm", "kernel": "k_load_tiles", "achieved": b / ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": b / ms / 1e6 / HBM_PEAK_GBS,
                      "ms": ms, "algorithmic_bytes": b, "note": "256 B pixels in, 256 B canonical tile + 1 B mirror flags + 12 B Lab means out per tile"}
        del lab
        ms, feats = ev_time(lambda: stages.features_rgb(tiles, None, 1, False))
        b = q_total * 640
        sr["features"] = {"bound": "hbm", "kernel": "k_features_tiles8", "achieved": b / ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": b / ms / 1e6 / HBM_PEAK_GBS,
                          "ms": ms, "algorithmic_bytes": b, "flops": q_total * 24576.0, "tflops": q_total * 24576.0 / ms / 1e9,
                          "note": "640 B per tile; `flops` is the nominal 24 576 per tile of the reference's 64-term sums -- the kernel takes a separable first look (two 8-point fast DCTs a row / column) "
                                  "and sums in the summation order DCTInner_asm fixes only the coefficients whose rounding is in doubt"}
        del feats
        ms, _ = ev_time(lambda: stages.dedup(tiles), reps=3)
        b = q_total * 260
        sr["dedup"] = {"bound": "hbm", "kernel": "run_dedup (hash table, full compare against the representative, radix sort of the distinct rows' prefixes + ties by whole rows)", "achieved": b / ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": b / ms / 1e6 / HBM_PEAK_GBS, "ms": ms, "algorithmic_bytes": b, "note": "one 256 B key read + one 4 B index written per tile; blocking call (host reads the distinct count)"}
        del tiles, flags
        # 64 error-feedback steps x PaletteSize colours per planned pixel, 9 int32 / fp32 operations per compare; the stage plans every DISTINCT
        # (palette, colour) pair once (tm_get_dither_pairs) and the pixels look their pair up, so the executed work is the pairs', not the pixels'
        pairs = enc.DitherPairs() if hasattr(enc, "DitherPairs") else 0
        planned = pairs if pairs > 0 else T * 64
        ops = planned * 64.0 * 16.0 * 9.0
        sr["dither"] = {"bound": "int32 valu", "kernel": "k_dither_tk_fast<true> + k_dd_mark / k_dd_lookup" if pairs > 0 else "k_dither_tk_fast<false>",
                        "achieved": ops / st["dither"] / 1e9, "peak": INT32_VALU_PEAK_TOPS, "unit": "Tops/s",
                        "frac": ops / st["dither"] / 1e9 / INT32_VALU_PEAK_TOPS, "ms": st["dither"], "pixels": T * 64, "distinct_pairs": pairs, "executed_ops": ops,
                        "nominal_ops": T * 64 * 64.0 * 16.0 * 9.0, "hbm_gb_s": T * 320 / st["dither"] / 1e6,
                        "note": "executed = planned pixels x 64 steps x PaletteSize compares x 9 operations (DESIGN.md section 5); `ms` is the stage's wall time, "
                                "most of which is now Reconstruct's query-feature kernel running beside it on the second stream"}
        it = enc.KmeansIters()
        if it["tile_iters"] > 0:
            n_it = it["tile_iters"] + it["pixel_iters"]
            sr["kmeans"] = {"bound": "latency (dependent iterations)", "kernel": "k_assign192 / k_h_bounds / k_assign192_list4 / k_h_update (192-D), k_kmeans3_persistent (colours)",
                            "ms": st["prepare_palettes"], "dependent_iterations": n_it, "us_per_iteration": st["prepare_palettes"] * 1e3 / max(n_it, 1), **it,
                            "nominal_point_bytes": 3.0 * it["pixel_points"] * it["pixel_iters"] + 768.0 * it["tile_points"] * it["tile_iters"],
                            "note": "SURVEY.md 8(d) prices k-means at 3 B per pixel and iteration (768 B per tile and iteration for the 192-D clustering); the "
                                    "kernels never move those bytes -- the colours are clustered as distinct (colour, count) points held in LDS and the tiles' "
                                    "iterations skip what provably cannot change -- so no HBM fraction is claimed: the stage is its dependent iterations "
                                    "(tile clustering, then the slowest palette's colours) times the microseconds each takes"}
        if not args.no_kmodes_extra:
            # A17's operator at config 5's shape (4K x 600: T = 1 618 022 rows of 80 bytes, 64 clusters): SURVEY.md 8(d) prices it at 80 B per
            # point and iteration against HBM; what bounds it is KModesIter's bin-serial rule (960 points, then the modes move)
            gk = torch.Generator(device="cuda").manual_seed(5)
            nk, kk_ = 1618022, 64
            proto = torch.randint(0, 48, (40, 80), generator=gk, device="cuda", dtype=torch.uint8)
            rows_k = proto[torch.randint(0, 40, (nk,), generator=gk, device="cuda")]
            noise = torch.rand((nk, 80), generator=gk, device="cuda") < 0.2
            rows_k = torch.where(noise, torch.randint(0, 48, (nk, 80), generator=gk, device="cuda", dtype=torch.uint8), rows_k).contiguous()
            del noise
            stages.kmodes_dev(rows_k, kk_, 0, 48, 1)  # warm-up: pool growth
            torch.cuda.synchronize()
            d1 = d5 = float("inf")
            for _rep in range(2):  # (two runs of each, the faster one: the initialisation's 64 host round trips vary by milliseconds between runs)
                t1 = time.perf_counter()
                _, _, cost1, _, pit1 = stages.kmodes_dev(rows_k, kk_, 0, 48, 1)
                torch.cuda.synchronize()
                d1 = min(d1, time.perf_counter() - t1)
                t1 = time.perf_counter()
                _, _, cost5, _, pit5 = stages.kmodes_dev(rows_k, kk_, 0, 48, 5)
                torch.cuda.synchronize()
                d5 = min(d5, time.perf_counter() - t1)
            per_iter = (d5 - d1) / max(1, (pit5 - pit1) // nk)
            bk = 80.0 * nk
            sr["kmodes"] = {"bound": "hbm", "kernel": "k_kmodes_owner + k_kmodes_walker per bin of 960 points (one graph per iteration); from the second iteration on k_kmodes_argmin over all remaining points + k_kmodes_fast (one workgroup, bin after bin) for as long as no mode changes", "achieved": bk / per_iter / 1e9, "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": bk / per_iter / 1e9 / HBM_PEAK_GBS, "ms_per_iteration": per_iter * 1e3, "rows": nk, "clusters": kk_,
                            "init_and_first_iteration_ms": d1 * 1e3, "algorithmic_bytes_per_iteration": bk,
                            "note": "tm_stage_kmodes_dev (TKModes.ComputeKModes, kmodes.pas:923-1094) on device pointers, 80 B per point and iteration; "
                                    "a bin of 960 points is two dependent launches (owners: previous moves into the histograms + scores; walker: the moves in order) where the modes keep changing -- the first iteration, most of the second --, which is what the time is: bound \"hbm\" prices the bytes, the chain of 2 x 1 686 launches is the limit; "
                                    "iterations in which no mode changes (the third and fourth here) score every point once and walk the bins in one launch; the figure is the mean over the iterations after the first"}
            del rows_k
        sr["knn"] = {k: out["roofline"][k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "launch_ms", "scan")}
        out["stage_rooflines"] = sr
    if world == 1 and not args.no_dense_extra:
        # diagnostic, outside the timed region: the same kernel with pruning off = a dense Q x T_distinct distance GEMM (BASELINE config 3)
        os.environ["TM_KNN_NOPRUNE"] = "1"
        for s_ in (TEncoderStep.esLoad, TEncoderStep.esPredictMotion, TEncoderStep.esReduce, TEncoderStep.esPreparePalettes,
                   TEncoderStep.esDither, TEncoderStep.esReconstruct):
            enc.Run(s_)
        del os.environ["TM_KNN_NOPRUNE"]
        kd = enc.KnnStats()
        dense_mfma = kd["consume_mfma"] * 65536.0 / (kd["consume_ms"] * 1e-3) / 1e12 if kd["consume_ms"] > 0 else 0.0
        dense = 384.0 * kd["pairs"] / max(kd["launches"], 1) / (kd["kernel_ms"] / max(kd["launches"], 1) * 1e-3) / 1e12  # (dense: the consume kernel is the whole search)
        out["roofline_dense"] = {"bound": "mfma", "achieved": dense, "peak": I8_DENSE_PEAK_TOPS, "unit": "TFLOP/s", "frac": dense / I8_DENSE_PEAK_TOPS,
                                 "mfma_pipe_frac": dense_mfma / I8_DENSE_PEAK_TOPS,
                                 "mfma_pipe_frac_of_measured_peak": (dense_mfma / out["measured_peaks"]["mfma_i8_tops"]) if "measured_peaks" in out else None,
                                 "launch_ms": kd["kernel_ms"] / max(kd["launches"], 1), "pairs_per_launch": kd["pairs"] / max(kd["launches"], 1),
                                 "pairs_expected": float(kd.get("queries", q_total)) * float(kd["db_rows"]),
                                 "note": "same kernel, pruning disabled (TM_KNN_NOPRUNE=1): every (query, distinct row) pair evaluated; parity-tested in tests/test_gpu_parity.py::test_knn_dense_mode*"}
    psnr = enc.PSNR() if hasattr(enc, "PSNR") else None
    if psnr:
        out["quality"] = psnr

    def extra(label, warm=True, **settings):
        saved = {k: getattr(enc, k) for k in settings}
        for k, v in settings.items():
            setattr(enc, k, v)
        if warm:
            enc.Run()  # untimed: the first pass of a configuration grows the memory pool (hipMalloc: 40-800 ms, by box)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        enc.Run()
        torch.cuda.synchronize()
        dt1 = time.perf_counter() - t1
        sm = enc.StageMs()
        r = {"value": F / dt1, "unit": "frames/s", "ms": dt1 * 1e3, "settings": settings,
             "stage_ms": {n: round(float(v), 3) for n, v in zip(STAGES, sm)}, "global_tiles": int(enc.counts()["tiles"])}
        if hasattr(enc, "PSNR"):
            r["quality"] = enc.PSNR()
        out[label] = r
        for k, v in saved.items():
            setattr(enc, k, v)
        return r

    if world == 1 and args.motion_radius == 0 and not args.no_motion_extra:
        r = extra("with_motion_prediction", warm=False, MotionPredictRadius=32)
        r["note"] = "the headline settings plus MotionPredictRadius=32 (PredictMotion + the motion branch of Reconstruct); not part of `value`"
    if world == 1 and args.motion_radius == 0 and not args.no_defaults_extra:
        r = extra("with_extended_palette_usage", FrameTilingExtendedPaletteUsage=True)
        r["tiles_matched_per_sec"] = q_total / (r["stage_ms"]["reconstruct"] * 1e-3) if r["stage_ms"]["reconstruct"] > 0 else None
        r["note"] = "SURVEY.md 8d's second number: k = 64 nearest rows + the tile x palette re-rank, motion prediction excluded"
        r = extra("with_motion_and_extended_palette_usage", MotionPredictRadius=32, FrameTilingExtendedPaletteUsage=True)
        r["note"] = ("the reference's default code paths (motion prediction radius 32 + extended palette usage) at the benchmark's %d palettes; "
                     "the reference's default PaletteCount is 1024 (tilingencoder.pas:3826)" % args.palettes)
    if world == 1 and args.motion_radius == 0 and not args.no_frozen_extra and not args.frozen_columns:
        # the clip rounds 1-3 were quoted on: the generator plus frozen tile columns (exact inter-frame duplicates, so Reconstruct searches
        # once per DISTINCT frame tile: 3.24 M of 4.32 M)
        rng2 = np.random.Generator(np.random.PCG64(synth.SEED))
        for f in range(F):
            hv[f] = synth.frame(f, W, H, rng2, freeze=True).view(np.int32)
        frames.copy_(host_frames)
        torch.cuda.synchronize()
        enc.SetFramesDevice(frames)
        enc.Run()  # untimed: pool sizes
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        nrep = max(1, min(args.steps, 3))
        sm = np.zeros(8)
        for _ in range(nrep):
            enc.Run()
            sm += enc.StageMs()
        torch.cuda.synchronize()
        dt1 = (time.perf_counter() - t1) / nrep
        k2 = enc.KnnStats()
        out["with_frozen_columns"] = {"value": F / dt1, "unit": "frames/s", "ms_per_step": dt1 * 1e3, "steps": nrep,
                                      "stage_ms": {n: round(float(v) / nrep, 3) for n, v in zip(STAGES, sm)},
                                      "knn_queries": int(k2.get("queries", q_total)), "distinct_database_rows": int(k2["db_rows"]),
                                      "knn_kernels_ms": {"k_knn_seed": k2["seed_ms"] / max(k2["launches"], 1), "k_knn_lists": k2["lists_ms"] / max(k2["launches"], 1),
                                                         "k_knn_consume": k2["consume_ms"] / max(k2["launches"], 1)},
                                      "knn_pairs_per_launch": k2["pairs"] / max(k2["launches"], 1),
                                      "final_tiles_after_reindex": int(enc.counts()["tiles"]),
                                      "note": "the headline step on the generator PLUS this repo's frozen tile columns (tiler_amd.synth.frame(freeze=True)), the clip "
                                              "rounds 1-3 quoted `value` on: device-resident like `value`, not part of it"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(W, H, F, args.palettes, T, int(ks["db_rows"]))
        visible = out["cpu_baseline"]["host_cpus_visible"]
        t32 = min(visible, 32)
        if t32 != out["cpu_baseline"]["cores"] and not os.environ.get("TM_BENCH_THREADS"):
            # the reference's demo streams were encoded with MaxThreadCount=32 (their embedded settings): the same legs with that many
            # threads, whatever share of the host this process owns
            c32 = cpu_baseline(W, H, F, args.palettes, T, int(ks["db_rows"]), threads=t32)
            out["cpu_baseline"]["with_32_threads"] = {"value": c32["value"], "cores": c32["cores"], "legs": c32["legs"],
                                                      "note": "min(visible CPUs, 32) threads, as the reference's demo settings (MaxThreadCount=32)"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    enc.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
