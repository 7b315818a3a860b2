"""End-to-end CPU pipeline built from the oracle's functions, in the order TTilingEncoder.Run(esAll) walks them
(tilingencoder.pas:5529-5554), with the build's documented scope (no motion prediction, KNN k=1, no OptimizePalettes).
Test infrastructure: the checker for tests/test_gpu_encoder.py and the CPU baseline of bench.py."""
import numpy as np


def run(oracle, frames, fps=24.0, palette_size=16, palette_count=1, dithering_mode=4, quality_tc=7.0, tile_count=0,
        max_s=15.0, min_s=1.0, lo=0.8, stop_after=None, timings=None):
    import time
    nf, h, w = frames.shape
    tm_w, tm_h = (w - 1) // 8 + 1, (h - 1) // 8 + 1
    per = tm_w * tm_h
    out = {}
    t0 = time.time()
    tiles, flags, labs = [], [], []
    for f in range(nf):
        t = oracle.load_from_image(frames[f], tm_w, tm_h)
        labs.append(oracle.inter_frame_data(t))
        c, fl = oracle.canonicalise(t)
        tiles.append(c)
        flags.append(fl)
    tiles = np.concatenate(tiles)
    flags = np.concatenate(flags)
    correl = np.zeros(nf, np.float32)
    for f in range(1, nf):
        correl[f] = oracle.pearson(labs[f - 1], labs[f])
    kf, nkf = oracle.find_keyframes(correl, fps, max_s, min_s, lo)
    out.update(tiles=tiles, flags=flags, correl=correl, keyframes=np.nonzero(kf)[0].astype(np.int32))
    if timings is not None:
        timings["load"] = time.time() - t0
    if stop_after == "load":
        return out
    # Reduce
    t0 = time.time()
    q = tiles.shape[0]
    nu, rep, order, use, remap = oracle.dedup(tiles, None)
    if tile_count <= 0:
        eqtc = oracle.L.tmo_equal_quality_tile_count(__import__("ctypes").c_double(float(q)))
        tile_count = min(int(np.rint(quality_tc * eqtc)), q)
    T = min(nu, tile_count)
    gtiles = tiles[order[:T]]
    gflags = flags[order[:T]]
    guse = use[:T]
    tm_tile = np.where(remap < T, remap, -1).astype(np.int32)
    out.update(T=T, gtiles=gtiles, gflags=gflags, guse=guse, tm_tile_reduce=tm_tile)
    if timings is not None:
        timings["reduce"] = time.time() - t0
    if stop_after == "reduce":
        return out
    # PreparePalettes
    t0 = time.time()
    feat = oracle.features_cluster(gtiles, dithering_mode)
    pal_idx = oracle.palettize(feat, guse, palette_count)
    palettes = np.stack([oracle.quantize_palette(gtiles[pal_idx == p].ravel(), palette_size) for p in range(palette_count)])
    import ctypes
    palettes = np.ascontiguousarray(palettes, np.int32)
    oracle.L.tmo_optimize_palettes(palettes.ctypes.data_as(ctypes.c_void_p), palette_count, palette_size)  # OptimizePalettes
    out.update(pal_idx=pal_idx, palettes=palettes)
    if timings is not None:
        timings["palettes"] = time.time() - t0
    # Dither
    t0 = time.time()
    pal_px = oracle.dither(gtiles, gflags, pal_idx, palettes, True)
    out.update(pal_px=pal_px)
    if timings is not None:
        timings["dither"] = time.time() - t0
    # Reconstruct (KNN branch)
    t0 = time.time()
    db = oracle.features_pal(pal_px, pal_idx, palettes, 1)
    qf = oracle.features_rgb(tiles, None, 1, False)
    idx, err = oracle.knn1(qf, db)
    out.update(knn_idx=idx, knn_err=err, tm_pal=pal_idx[idx])
    if timings is not None:
        timings["reconstruct"] = time.time() - t0
    # Reindex
    t0 = time.time()
    hist = np.bincount(idx, minlength=T).astype(np.uint32)
    nu2, rep2, order2, use2, remap2 = oracle.dedup(pal_px, hist)
    out.update(final_T=nu2, final_pal_px=pal_px[order2], final_rgb=gtiles[order2], final_use=use2, final_pal_idx=pal_idx[order2],
               final_tm_tile=remap2[idx].astype(np.int32), per=per)
    if timings is not None:
        timings["reindex"] = time.time() - t0
    return out
