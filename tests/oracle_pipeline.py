"""End-to-end CPU pipeline built from the oracle's functions, in the order TTilingEncoder.Run(esAll) walks them
(tilingencoder.pas:5529-5554): Load, PredictMotion (motion_radius > 0), Reduce, PreparePalettes (+OptimizePalettes),
Dither, Reconstruct (KNN k = 1; motion redo when motion_radius > 0), Reindex.
Test infrastructure: the checker for tests/test_gpu_encoder.py and the CPU baseline of bench.py."""
import ctypes

import numpy as np

HIGH = 0xFFFFFFFF


def _to_screen(tiles_u32, tm_w, tm_h):
    """[tm_h*tm_w][64] tiles (original orientation) -> [tm_h*8][tm_w*8] frame buffer"""
    return np.ascontiguousarray(tiles_u32.reshape(tm_h, tm_w, 8, 8).transpose(0, 2, 1, 3).reshape(tm_h * 8, tm_w * 8))


def run(oracle, frames, fps=24.0, palette_size=16, palette_count=1, dithering_mode=4, quality_tc=7.0, tile_count=0,
        max_s=15.0, min_s=1.0, lo=0.8, stop_after=None, timings=None, motion_radius=0, epu=False):
    import time
    nf, h, w = frames.shape
    tm_w, tm_h = (w - 1) // 8 + 1, (h - 1) // 8 + 1
    per = tm_w * tm_h
    out = {}
    t0 = time.time()
    tiles, flags, labs, screens = [], [], [], []
    for f in range(nf):
        t = oracle.load_from_image(frames[f], tm_w, tm_h)
        labs.append(oracle.inter_frame_data(t))
        if motion_radius > 0:
            screens.append(_to_screen(t, tm_w, tm_h))
        c, fl = oracle.canonicalise(t)
        tiles.append(c)
        flags.append(fl)
    tiles = np.concatenate(tiles)
    flags = np.concatenate(flags)
    correl = np.zeros(nf, np.float32)
    for f in range(1, nf):
        correl[f] = oracle.pearson(labs[f - 1], labs[f])
    kf, nkf = oracle.find_keyframes(correl, fps, max_s, min_s, lo)
    keyframes = np.nonzero(kf)[0].astype(np.int32)
    out.update(tiles=tiles, flags=flags, correl=correl, keyframes=keyframes, per=per)
    if timings is not None:
        timings["load"] = time.time() - t0
    if stop_after == "load":
        return out
    q = tiles.shape[0]

    # PredictMotion (1964-1991): frame 0 against frame 1, frame f against the SOURCE pixels of frame f-1
    mp = motion_radius > 0
    if mp:
        t0 = time.time()
        cur_all = oracle.features_rgb(tiles, flags, 1, False)  # features of the tiles in their original orientation
        pm_err = np.zeros(q, np.uint32)
        pm_x, pm_y = np.zeros(q, np.int8), np.zeros(q, np.int8)
        for f in range(nf):
            back = screens[f - 1] if f >= 1 else (screens[1] if nf > 1 else np.zeros_like(screens[0]))
            win = oracle.window_dcts(back)
            sl = slice(f * per, (f + 1) * per)
            pm_err[sl], pm_x[sl], pm_y[sl] = oracle.motion_search(cur_all[sl], tm_w, tm_h, win, motion_radius)
        pm_psnr = oracle.psnr(pm_err)
        out.update(pm_err=pm_err, pm_x=pm_x, pm_y=pm_y, pm_psnr=pm_psnr)
        if timings is not None:
            timings["predict_motion"] = time.time() - t0
        if stop_after == "predict_motion":
            return out

    # Reduce
    t0 = time.time()
    if tile_count <= 0:
        eqtc = oracle.L.tmo_equal_quality_tile_count(ctypes.c_double(float(q)))
        tile_count = min(int(np.rint(quality_tc * eqtc)), q)
    if mp:
        # SolveTileCount (4043): golden-section search of the PSNR threshold; state = the last probe's
        frame_of = np.arange(q) // per
        eff = np.where(np.isin(frame_of, keyframes), pm_psnr.astype(np.float64) / 10.0, pm_psnr.astype(np.float64))  # 4028-4031
        _, rep, _, _, _ = oracle.dedup(tiles, None)
        gmin = np.full(q, np.inf)
        np.minimum.at(gmin, rep, eff)
        x, probes = oracle.solve_tile_count(np.sort(gmin[np.isfinite(gmin)]), tile_count)
        assert probes > 0
        predicted = eff > x
        sel = np.nonzero(~predicted)[0]  # TransferTiles (4048-4103) in frame-major order
        nu, rep2, order, use, remap = oracle.dedup(tiles[sel], None)
        T = nu
        gtiles = tiles[sel][order]
        gflags = flags[sel][order]
        guse = use
        tm_tile = np.full(q, -1, np.int32)
        tm_tile[sel] = remap
        out.update(threshold=x, probes=probes, predicted_reduce=predicted)
    else:
        nu, rep, order, use, remap = oracle.dedup(tiles, None)
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
    # Reconstruct
    t0 = time.time()
    db = oracle.features_pal(pal_px, pal_idx, palettes, 1)
    qf = oracle.features_rgb(tiles, None, 1, False)
    if epu:  # FrameTilingExtendedPaletteUsage (1559-1610): 64 nearest rows, every unique tile x every unique palette of the list
        idx64, _ = oracle.knnk(qf, db, 64)
        idx, epu_pal, err = oracle.epu_rerank(qf, idx64, pal_px, pal_idx, palettes)
        out.update(knn_idx64=idx64)
    else:
        idx, err = oracle.knn1(qf, db)
        epu_pal = pal_idx[idx]
    tm_tile_r = idx.astype(np.int32).copy()
    tm_pal = epu_pal.astype(np.int32).copy()
    tm_err = err.copy()
    is_pred = np.zeros(q, bool)
    px = pm_x.copy() if mp else np.zeros(q, np.int8)
    py = pm_y.copy() if mp else np.zeros(q, np.int8)
    if mp:
        # frames in order, each against the previous RECONSTRUCTED frame (1950-1956, 1496-1532, 1612-1654)
        pals_u = palettes.view(np.uint32)
        front = np.zeros((tm_h * 8, tm_w * 8), np.uint32)
        back = np.zeros_like(front)
        for f in range(nf):
            sl = slice(f * per, (f + 1) * per)
            mp_err = np.full(per, HIGH, np.uint32)
            if f not in keyframes:
                win = oracle.window_dcts(back)
                mp_err, px[sl], py[sl] = oracle.motion_search(cur_all[sl], tm_w, tm_h, win, motion_radius)
            perfect = mp_err <= 192                                   # IsZero(mpErr, cTileDCTSize), 1534
            knn_e = np.where(perfect, HIGH, err[sl]).astype(np.int64)
            mpe = mp_err.astype(np.int64)
            knn_best = (np.abs(knn_e - mpe) > 192) & (knn_e < mpe)    # CompareValue(knnErr, mpErr, 192) = LessThanValue, 1614
            tm_tile_r[sl] = np.where(perfect, -1, tm_tile_r[sl])
            tm_pal[sl] = np.where(perfect, -1, tm_pal[sl])
            is_pred[sl] = ~knn_best
            tm_err[sl] = np.where(knn_best, err[sl], mp_err)
            for i in range(per):
                sy, sx = divmod(i, tm_w)
                dy, dx = sy * 8, sx * 8
                if knn_best[i]:
                    t = pal_px[tm_tile_r[sl][i]].reshape(8, 8)
                    fl = flags[f * per + i]
                    if fl & 1:
                        t = t[:, ::-1]
                    if fl & 2:
                        t = t[::-1, :]
                    front[dy:dy + 8, dx:dx + 8] = pals_u[tm_pal[sl][i]][t]
                else:
                    oy, ox = dy + int(py[sl][i]), dx + int(px[sl][i])
                    front[dy:dy + 8, dx:dx + 8] = back[oy:oy + 8, ox:ox + 8]
            front, back = back, front
        out.update(recon_last=back.copy())
    out.update(knn_idx=idx, knn_err=err, tm_pal=tm_pal, tm_tile_recon=tm_tile_r, tm_err=tm_err, is_predicted=is_pred, pred_x=px, pred_y=py)
    if timings is not None:
        timings["reconstruct"] = time.time() - t0
    # Reindex
    t0 = time.time()
    hist = np.bincount(tm_tile_r[tm_tile_r >= 0], minlength=T).astype(np.uint32)
    nu2, rep2, order2, use2, remap2 = oracle.dedup(pal_px, hist)
    final_tm = np.where(tm_tile_r >= 0, remap2[np.maximum(tm_tile_r, 0)], -1).astype(np.int32)
    out.update(final_T=nu2, final_pal_px=pal_px[order2], final_rgb=gtiles[order2], final_use=use2, final_pal_idx=pal_idx[order2],
               final_tm_tile=final_tm)
    if timings is not None:
        timings["reindex"] = time.time() - t0
    return out
