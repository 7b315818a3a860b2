#!/usr/bin/env python3
"""bench.py -- BASELINE.json headline metric on MI355X: encoded frames/sec (+ tiles-matched/sec) of the TileMotion
per-frame tile pipeline on the 720p, 300-frame, 16-palette synthetic clip (configs[1]).

A step = one TTilingEncoder.Run(esAll) pass (Load -> Reduce -> PreparePalettes -> Dither -> Reconstruct -> Reindex)
over the whole clip with the RGB frames already resident in HBM.  N > 1: one process per GPU (torch.distributed, RCCL);
the clip is ONE job split over the ranks (strong scaling), see tiler_amd/distributed.py.  Prints one JSON line.
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
# HBM-side bytes per launch of the dominant kernel on the default workload, from the separate rocprofv3 --pmc passes of
# profiles/r01_pmc_knn.md: 2 x FETCH_SIZE (gfx950 counts half of a 16-B/lane stream) + WRITE_SIZE, in bytes
PMC_TRAFFIC_BYTES = 2 * 33385770 * 1024 + 36914 * 1024


def cpu_baseline(width, height, nframes, palette_count, t_global, seconds_budget=20.0):
    """The oracle (CPU restatement, kind 'port', 1 thread) on a bounded sample, scaled to frames/s of the same workload."""
    import subprocess
    so = os.path.join(ROOT, "oracle", "libtm_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libtm_oracle.so"])
    from tests.oracle_binding import Oracle
    from tiler_amd import synth
    o = Oracle(so)
    tm_w, tm_h = (width - 1) // 8 + 1, (height - 1) // 8 + 1
    per = tm_w * tm_h
    fr = synth.video(2, width, height)
    t0 = time.time()
    tiles = o.load_from_image(fr[0], tm_w, tm_h)
    o.inter_frame_data(tiles)
    canon, flags = o.canonicalise(tiles)
    t_load = time.time() - t0
    sample = canon[: per // 8]
    t0 = time.time()
    qf = o.features_rgb(sample, None, 1, False)
    t_feat = (time.time() - t0) * per / sample.shape[0]
    # global-tile stages are paid once per clip for T tiles: amortised per frame = cost(T) / nframes
    gs = canon[:256]
    rng = np.random.default_rng(0)
    pal_idx = rng.integers(0, palette_count, size=gs.shape[0]).astype(np.int32)
    palettes = rng.integers(0, 1 << 24, size=(palette_count, 16)).astype(np.int32)
    t0 = time.time()
    o.features_cluster(gs, 4)
    pp = o.dither(gs, flags[:256], pal_idx, palettes, True)
    db_small = o.features_pal(pp, pal_idx, palettes, 1)
    t_global_per_tile = (time.time() - t0) / gs.shape[0]
    # exact dedup: sort of 256-byte keys, n log n; sample then scale by n log n
    ds = np.concatenate([canon, canon[: per // 2]])
    t0 = time.time()
    o.dedup(ds, None)
    n_s, n_full = ds.shape[0], per * nframes
    t_dedup_clip = (time.time() - t0) * (n_full * np.log2(n_full)) / (n_s * np.log2(n_s))
    # KNN: the scalar SSD loop of utils.pas:541-557 over the full database for a handful of queries
    db = rng.integers(-300, 300, size=(t_global, 192)).astype(np.int16)
    db[: db_small.shape[0]] = db_small
    nq = 8
    t0 = time.time()
    o.knn1(qf[:nq], db)
    dt = time.time() - t0
    while dt < seconds_budget / 4 and nq < 512:
        nq *= 4
        t0 = time.time()
        o.knn1(qf[:nq], db)
        dt = time.time() - t0
    t_knn_frame = dt * per / nq
    sec_per_frame = t_load + t_feat + t_knn_frame + (t_global_per_tile * t_global + t_dedup_clip) / nframes
    return {
        "value": 1.0 / sec_per_frame, "unit": "frames/s", "cores": 1, "kind": "port",
        "sample": (f"oracle (C restatement, gcc -O3, 1 thread) timed on: 1 frame load+Lab+mirrors, {sample.shape[0]} query feature "
                   f"vectors, 256 global tiles (cluster features + Thomas-Knoll dither + database features), exact dedup of "
                   f"{n_s} tiles (scaled n log n), brute-force KNN of {nq} queries x full {t_global}-tile database; scaled linearly to "
                   f"{per} tiles/frame; k-means stages excluded (favours the CPU)"),
        "tiles_matched_per_sec": per / t_knn_frame,
    }


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
                    help="skip the untimed extra pass with the reference's default settings (motion prediction + extended palette usage)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from tiler_amd import synth, distributed
    from tiler_amd.encoder import TilingEncoder

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libtilemotion has no CPU path")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    W, H, F = args.width, args.height, args.frames
    # synthetic clip, generated on the host in slabs and parked in HBM before any timing
    frames = torch.empty((F, H, W), dtype=torch.int32, device="cuda")
    rng = np.random.Generator(np.random.PCG64(synth.SEED))
    for f in range(F):
        frames[f] = torch.from_numpy(synth.frame(f, W, H, rng).view(np.int32)).cuda()
    torch.cuda.synchronize()

    enc = TilingEncoder()
    enc.LoadDefaultSettings()
    enc.PaletteCount = args.palettes
    enc.PaletteSize = 16
    enc.FrameTilingExtendedPaletteUsage = False  # headline KNN number: EPU off
    enc.MotionPredictRadius = args.motion_radius
    enc.SetVideo(W, H, 24.0, F)
    enc.SetFramesDevice(frames)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        distributed.run_all(enc, F, rank, world)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    knn_ms = knn_pairs = knn_launches = 0
    stage_ms = np.zeros(8)
    for _ in range(args.steps):
        step()
        ks = enc.KnnStats()
        knn_ms += ks["kernel_ms"]; knn_pairs += ks["pairs"]; knn_launches += ks["launches"]
        stage_ms += enc.StageMs()
    barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    c = enc.counts()
    ks = enc.KnnStats()
    per_launch_ms = knn_ms / max(knn_launches, 1)
    alg_ops_per_launch = 384.0 * knn_pairs / max(knn_launches, 1)  # 2*192 integer ops per (query, tile) pair
    achieved = alg_ops_per_launch / (per_launch_ms * 1e-3) / 1e12 if per_launch_ms > 0 else 0.0
    ms_per_step = dt / args.steps * 1e3
    q_total = F * c["tm_w"] * c["tm_h"]
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
                               f"Thomas-Knoll dither, KNN k=1 (EPU off), MotionPredictRadius={args.motion_radius}" + (" (motion prediction excluded, SURVEY.md 8d)" if args.motion_radius == 0 else ""),
                   "frames": F, "tiles_per_frame": c["tm_w"] * c["tm_h"], "query_tiles": q_total, "global_tiles_T": int(enc.GlobalTilingTileCount),
                   "distinct_database_rows": int(ks["db_rows"]), "final_tiles_after_reindex": int(c["tiles"]),
                   "parallelism": f"frames sharded over {world} GPU(s) for Reconstruct; other steps replicated"},
        "tiles_matched_per_sec": q_total / (float(stage_ms[5]) / args.steps * 1e-3) if stage_ms[5] > 0 else None,
        "stage_ms": {n: round(float(v) / args.steps, 3) for n, v in zip(["load", "predict_motion", "reduce", "prepare_palettes", "dither", "reconstruct", "reindex", "save"], stage_ms)},
        "nominal_pairs": float(q_total) * float(enc.GlobalTilingTileCount),
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": I8_DENSE_PEAK_TOPS, "unit": "TFLOP/s", "frac": achieved / I8_DENSE_PEAK_TOPS,
                     "traffic": PMC_TRAFFIC_BYTES if (W, H, F, args.palettes) == (1280, 720, 300, 16) else None, "kernel": "k_knn_mfma", "launch_ms": per_launch_ms, "k_bytes": ks["k_bytes"],
                     "pairs_per_launch": knn_pairs / max(knn_launches, 1),
                     "note": "int8 ops; algorithmic = 384 ops per evaluated (query, distinct database row) pair; the kernel executes 2*k_bytes ops per pair on the MFMA pipe"},
    }
    if world == 1 and rank == 0:
        # the peaks measured on this very device: a bare loop of the kernel's MFMA instruction and an HBM stream triad
        import ctypes
        from tiler_amd import lib as _lib_fn
        from tiler_amd._lib import check as _check
        tops, gbs = ctypes.c_double(), ctypes.c_double()
        _check(_lib_fn().tm_probe_mfma_i8(0.3, ctypes.byref(tops)))
        _check(_lib_fn().tm_probe_hbm_triad(1 << 30, ctypes.byref(gbs)))
        out["measured_peaks"] = {"mfma_i8_tops": tops.value, "hbm_triad_gb_s": gbs.value,
                                 "note": "bare v_mfma_i32_32x32x32_i8 loop (two waves per SIMD, operands in registers) and a = b + s*c over 3 x 1 GiB; "
                                         "roofline.frac stays against the 5000 TOP/s vendor figure"}
        out["roofline"]["frac_of_measured_peak"] = achieved / tops.value if tops.value > 0 else None
        out["roofline"]["mfma_pipe_frac_of_measured_peak"] = achieved * (2 * ks["k_bytes"] / 384.0) / tops.value if tops.value > 0 else None
    if world == 1:
        # diagnostic, outside the timed region: the same kernel with pruning off = a dense Q x T_distinct distance GEMM,
        # which is what the MFMA roofline is really about (the shipped path skips >95 % of it)
        os.environ["TM_KNN_NOPRUNE"] = "1"
        from tiler_amd.encoder import TEncoderStep
        for st in (TEncoderStep.esLoad, TEncoderStep.esPredictMotion, TEncoderStep.esReduce, TEncoderStep.esPreparePalettes,
                   TEncoderStep.esDither, TEncoderStep.esReconstruct):
            enc.Run(st)
        del os.environ["TM_KNN_NOPRUNE"]
        kd = enc.KnnStats()
        dense = 384.0 * kd["pairs"] / max(kd["launches"], 1) / (kd["kernel_ms"] / max(kd["launches"], 1) * 1e-3) / 1e12
        out["roofline_dense"] = {"bound": "mfma", "achieved": dense, "peak": I8_DENSE_PEAK_TOPS, "unit": "TFLOP/s", "frac": dense / I8_DENSE_PEAK_TOPS,
                                 "mfma_pipe_frac": dense * (2 * kd["k_bytes"] / 384.0) / I8_DENSE_PEAK_TOPS,
                                 "mfma_pipe_frac_of_measured_peak": (dense * (2 * kd["k_bytes"] / 384.0) / out["measured_peaks"]["mfma_i8_tops"]) if "measured_peaks" in out else None,
                                 "launch_ms": kd["kernel_ms"] / max(kd["launches"], 1), "pairs_per_launch": kd["pairs"] / max(kd["launches"], 1),
                                 "note": "same kernel, pruning disabled (TM_KNN_NOPRUNE=1): every (query, distinct row) pair evaluated"}
    if world == 1 and args.motion_radius == 0 and not args.no_motion_extra:
        # second number, outside the timed region: the reference's default path with motion prediction (radius 32)
        enc.MotionPredictRadius = 32
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        enc.Run()
        torch.cuda.synchronize()
        dt1 = time.perf_counter() - t1
        sm = enc.StageMs()
        pred = 0
        for f in range(0, F, max(1, F // 10)):
            pred += int(((enc.TileMap(f)["Flags"] >> 2) & 1).sum())
        out["with_motion_prediction"] = {"value": F / dt1, "unit": "frames/s", "ms": dt1 * 1e3, "radius": 32,
                                         "stage_ms": {n: round(float(v), 3) for n, v in zip(["load", "predict_motion", "reduce", "prepare_palettes", "dither", "reconstruct", "reindex", "save"], sm)},
                                         "global_tiles": int(enc.counts()["tiles"]),
                                         "predicted_fraction_sampled": pred / float(len(range(0, F, max(1, F // 10))) * c["tm_w"] * c["tm_h"])}
        enc.MotionPredictRadius = 0
    if world == 1 and args.motion_radius == 0 and not args.no_defaults_extra:
        # third number, outside the timed region: the reference's own defaults (MotionPredictRadius 32, FrameTilingExtendedPaletteUsage on)
        enc.MotionPredictRadius = 32
        enc.FrameTilingExtendedPaletteUsage = True
        enc.Run()  # untimed: the first pass of this configuration grows the memory pool by ~10 GB (hipMalloc: 40-800 ms, by box)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        enc.Run()
        torch.cuda.synchronize()
        dt1 = time.perf_counter() - t1
        sm = enc.StageMs()
        moved = 0
        hdr_e, _, _ = enc.Tiles()
        for f in range(0, F, max(1, F // 10)):
            tm_e = enc.TileMap(f)
            ok = tm_e["TileIdx"] >= 0
            moved += int((tm_e["PalIdx"][ok] != hdr_e["PalIdx_Initial"][tm_e["TileIdx"][ok]]).sum())
        out["with_reference_defaults"] = {"value": F / dt1, "unit": "frames/s", "ms": dt1 * 1e3,
                                          "stage_ms": {n: round(float(v), 3) for n, v in zip(["load", "predict_motion", "reduce", "prepare_palettes", "dither", "reconstruct", "reindex", "save"], sm)},
                                          "items_on_another_palette_sampled": moved}
        enc.MotionPredictRadius = 0
        enc.FrameTilingExtendedPaletteUsage = False
    if world == 1 and args.motion_radius == 0 and not args.no_defaults_extra:
        # SURVEY.md 8d's "second number": the headline configuration with FrameTilingExtendedPaletteUsage on (k = 64 + re-rank), motion prediction excluded
        enc.FrameTilingExtendedPaletteUsage = True
        enc.Run()  # untimed warm-up of this configuration (grows the memory pool)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        enc.Run()
        torch.cuda.synchronize()
        dt1 = time.perf_counter() - t1
        sm = enc.StageMs()
        out["with_extended_palette_usage"] = {"value": F / dt1, "unit": "frames/s", "ms": dt1 * 1e3,
                                              "tiles_matched_per_sec": q_total / (float(sm[5]) * 1e-3) if sm[5] > 0 else None,
                                              "stage_ms": {n: round(float(v), 3) for n, v in zip(["load", "predict_motion", "reduce", "prepare_palettes", "dither", "reconstruct", "reindex", "save"], sm)}}
        enc.FrameTilingExtendedPaletteUsage = False
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(W, H, F, args.palettes, int(enc.GlobalTilingTileCount))
        out["config"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(out), flush=True)
    enc.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
