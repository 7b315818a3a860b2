"""Stage-level host wrappers over the C ABI for torch device tensors (tm_stage_*, include/tilemotion.h).

Each function names the reference routine it stands for; tensors must live on the current CUDA(HIP) device.
"""
import ctypes

import torch

from ._lib import lib, check


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def load(frames, tm_w, tm_h):
    """TFrame.LoadFromImage + PrepareInterFrameData + mirror canonicalisation (tilingencoder.pas:1293-1411).
    frames: uint32-as-int32 [F][H][W] RGB32 -> (tiles int32 [F*tm_w*tm_h][64], flags uint8, lab_means float32 [.,3])"""
    assert frames.is_cuda and frames.dtype == torch.int32 and frames.is_contiguous()
    f, h, w = frames.shape
    n = f * tm_w * tm_h
    tiles = torch.empty((n, 64), dtype=torch.int32, device=frames.device)
    flags = torch.empty((n,), dtype=torch.uint8, device=frames.device)
    lab = torch.empty((n, 3), dtype=torch.float32, device=frames.device)
    check(lib().tm_stage_load(_p(frames), f, w, h, tm_w, tm_h, _p(tiles), _p(flags), _p(lab), _stream()))
    return tiles, flags, lab


def rgb_to_lab(rgb):
    """RGBToLAB (utils.pas:374-410) of colours 0x00RRGGBB (int32 [n]) -> float32 [n][3]"""
    assert rgb.is_cuda and rgb.dtype == torch.int32 and rgb.is_contiguous()
    out = torch.empty((rgb.shape[0], 3), dtype=torch.float32, device=rgb.device)
    check(lib().tm_stage_rgb_to_lab(_p(rgb), rgb.shape[0], _p(out), _stream()))
    return out


def features_rgb(tiles, mirror_flags=None, mode=1, use_lab=False):
    """ConvertToCpnPixels + ComputeCpnPixelsPsyVisFeatures (tilingencoder.pas:3049-3131) -> int16 [n][192]"""
    assert tiles.is_cuda and tiles.dtype == torch.int32 and tiles.is_contiguous()
    n = tiles.shape[0]
    out = torch.empty((n, 192), dtype=torch.int16, device=tiles.device)
    check(lib().tm_stage_features_rgb(_p(tiles), n, _p(mirror_flags), mode, int(use_lab), _p(out), _stream()))
    return out


def features_pal(pal_px, pal_idx, palettes, mode=1):
    """PrepareReconstruct.DoPsyV (tilingencoder.pas:4570-4583) -> int16 [n][192]"""
    n = pal_px.shape[0]
    out = torch.empty((n, 192), dtype=torch.int16, device=pal_px.device)
    check(lib().tm_stage_features_pal(_p(pal_px), _p(pal_idx), n, _p(palettes), palettes.shape[1], mode, _p(out), _stream()))
    return out


def features_cluster(tiles, mode=4):
    """ComputeTilePsyVisFeatures as DoPalettization calls it (tilingencoder.pas:4126), Round()ed -> int32 [n][192]"""
    n = tiles.shape[0]
    out = torch.empty((n, 192), dtype=torch.int32, device=tiles.device)
    check(lib().tm_stage_features_cluster(_p(tiles), n, mode, _p(out), _stream()))
    return out


def window_dcts(frame_buffer):
    """PredictMotion.DoDCTs / Reconstruct.DoDCTs (tilingencoder.pas:1157-1182): int32 [H][W] 0x00BBGGRR -> int16 [(H-7)*(W-7)][192]"""
    h, w = frame_buffer.shape
    out = torch.empty(((h - 7) * (w - 7), 192), dtype=torch.int16, device=frame_buffer.device)
    check(lib().tm_stage_window_dcts(_p(frame_buffer), w, h, _p(out), _stream()))
    return out


def motion_search(cur, tm_w, tm_h, win, radius):
    """PredictMotion.DoXY search (tilingencoder.pas:1209-1253) -> (err int32-as-uint32, px int8, py int8), one per tile"""
    n = tm_w * tm_h
    err = torch.empty((n,), dtype=torch.int32, device=cur.device)
    px = torch.empty((n,), dtype=torch.int8, device=cur.device)
    py = torch.empty((n,), dtype=torch.int8, device=cur.device)
    check(lib().tm_stage_motion_search(_p(cur), tm_w, tm_h, _p(win), radius, _p(err), _p(px), _p(py), _stream()))
    return err, px, py


def knn_topk(queries, db, k=64):
    """ann_kdtree_short_search_multi (tilingencoder.pas:1563) for every query -> (idx int32 [nq][k], err int32-as-uint32 [nq][k])"""
    nq = queries.shape[0]
    idx = torch.empty((nq, k), dtype=torch.int32, device=queries.device)
    err = torch.empty((nq, k), dtype=torch.int32, device=queries.device)
    check(lib().tm_stage_knn_topk(_p(queries), nq, _p(db), db.shape[0], k, _p(idx), _p(err), _stream()))
    return idx, err


def epu_rerank(queries, knn_idx, pal_px, tile_pal_idx, palettes):
    """FrameTilingExtendedPaletteUsage re-rank (tilingencoder.pas:1576-1610) -> (tile int32, pal int32, err int32-as-uint32)"""
    nq = queries.shape[0]
    t = torch.empty((nq,), dtype=torch.int32, device=queries.device)
    p = torch.empty((nq,), dtype=torch.int32, device=queries.device)
    e = torch.empty((nq,), dtype=torch.int32, device=queries.device)
    check(lib().tm_stage_epu_rerank(_p(queries), nq, _p(knn_idx), knn_idx.shape[1], _p(pal_px), _p(tile_pal_idx), pal_px.shape[0],
                                    _p(palettes), palettes.shape[0], palettes.shape[1], _p(t), _p(p), _p(e), _stream()))
    return t, p, e


def knn(queries, db):
    """ann_kdtree_short_search(eps=0) for every query (tilingencoder.pas:1547) -> (idx int32, err int32-as-uint32)"""
    nq, nt = queries.shape[0], db.shape[0]
    idx = torch.empty((nq,), dtype=torch.int32, device=queries.device)
    err = torch.empty((nq,), dtype=torch.int32, device=queries.device)
    check(lib().tm_stage_knn(_p(queries), nq, _p(db), nt, _p(idx), _p(err), _stream()))
    return idx, err


def knn_last_plan():
    """(ht, hq, topk, arena_retries): the digit plan and mode of this thread's last scan, and the process's count of repeated scans (tests)"""
    ht, hq, tk, r = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int64()
    check(lib().tm_knn_last_plan(ctypes.byref(ht), ctypes.byref(hq), ctypes.byref(tk), ctypes.byref(r)))
    return ht.value, hq.value, tk.value, r.value


class KnnIndex:
    """ann_kdtree_short_create analogue: the database is packed once and searched by many query batches."""

    def __init__(self, db):
        self.db = db  # borrowed for the index lifetime, like the reference's DS.Dataset (tilingencoder.pas:4600)
        self.h = lib().tm_knn_index_create(_p(db), db.shape[0], _stream())
        if not self.h:
            check(-3)

    def search(self, queries):
        nq = queries.shape[0]
        idx = torch.empty((nq,), dtype=torch.int32, device=queries.device)
        err = torch.empty((nq,), dtype=torch.int32, device=queries.device)
        check(lib().tm_knn_index_search(ctypes.c_void_p(self.h), _p(queries), nq, _p(idx), _p(err), _stream()))
        return idx, err

    def last_stats(self):
        ms, kb, pairs = ctypes.c_double(), ctypes.c_int(), ctypes.c_int64()
        check(lib().tm_knn_index_last_stats(ctypes.c_void_p(self.h), ctypes.byref(ms), ctypes.byref(kb), ctypes.byref(pairs)))
        return ms.value, kb.value, pairs.value

    def close(self):
        if self.h:
            lib().tm_knn_index_destroy(ctypes.c_void_p(self.h))
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def dither(tiles, flags, pal_idx, palettes, use_thomas_knoll=True, y2_mixed=4):
    """Dither = PreparePlan + DitherTile per tile (tilingencoder.pas:1873-1907) -> uint8 [n][64]"""
    n = tiles.shape[0]
    out = torch.empty((n, 64), dtype=torch.uint8, device=tiles.device)
    check(lib().tm_stage_dither(_p(tiles), _p(flags), _p(pal_idx), n, _p(palettes), palettes.shape[0], palettes.shape[1],
                                int(use_thomas_knoll), y2_mixed, _p(out), _stream()))
    return out


def dedup(rows, use_in=None):
    """MakeTilesUnique + ReindexTiles (tilingencoder.pas:4720-4781, 4626-4700).  rows: int32 [n][64] (RGB tiles) or
    uint8 [n][64] (palette-index tiles).  -> (n_unique, remap int32 [n], order int32 [n_unique], use uint32-as-int32 [n_unique])"""
    n = rows.shape[0]
    row_bytes = rows.shape[1] * rows.element_size()
    remap = torch.empty((n,), dtype=torch.int32, device=rows.device)
    order = torch.empty((n,), dtype=torch.int32, device=rows.device)
    use = torch.empty((n,), dtype=torch.int32, device=rows.device)
    nu = ctypes.c_int64()
    check(lib().tm_stage_dedup(_p(rows), n, row_bytes, _p(use_in), _p(remap), _p(order), _p(use), ctypes.byref(nu), _stream()))
    return nu.value, remap, order[: nu.value], use[: nu.value]


def kmeans(pts, weights, k, max_iter=300):
    """the build's deterministic k-means: pts int32 [n][d] -> (live_k, assign int32 [n], centroids float64 [k][d], iters)"""
    n, d = pts.shape
    assign = torch.empty((n,), dtype=torch.int32, device=pts.device)
    cent = torch.zeros((k, d), dtype=torch.float64, device=pts.device)
    hk, hi = ctypes.c_int(), ctypes.c_int()
    check(lib().tm_stage_kmeans(_p(pts), _p(weights), n, d, k, max_iter, _p(assign), _p(cent), ctypes.byref(hk), ctypes.byref(hi), _stream()))
    return hk.value, assign, cent, hi.value


def kmeans_seeded(pts, weights, k, init_idx, max_iter=300):
    """the same Lloyd iterations from the caller's own initial centres (init_idx: k point indices, -1 = none)"""
    import numpy as np
    n, d = pts.shape
    assign = torch.empty((n,), dtype=torch.int32, device=pts.device)
    cent = torch.zeros((k, d), dtype=torch.float64, device=pts.device)
    idx = np.full(k, -1, np.int64)
    idx[: len(init_idx)] = np.asarray(init_idx, np.int64)[:k]
    hk, hi = ctypes.c_int(), ctypes.c_int()
    check(lib().tm_stage_kmeans_seeded(_p(pts), _p(weights), n, d, k, idx.ctypes.data_as(ctypes.c_void_p), max_iter, _p(assign), _p(cent),
                                       ctypes.byref(hk), ctypes.byref(hi), _stream()))
    return hk.value, assign, cent, hi.value


def quantize_palettes(tiles, pal_idx, npal, pal_size, max_iter=300):
    """QuantizeUsingYakmo + DoQuantization for every palette (tilingencoder.pas:4434-4564) -> int32 [npal][pal_size]"""
    out = torch.empty((npal, pal_size), dtype=torch.int32, device=tiles.device)
    check(lib().tm_stage_quantize_palettes(_p(tiles), _p(pal_idx), tiles.shape[0], npal, pal_size, max_iter, _p(out), _stream()))
    return out


def palettize(feat, use, npal, max_iter=300):
    """DoPalettization (tilingencoder.pas:4105-4245): cluster features -> PalIdx_Initial int32 [n], palettes ranked by tile count"""
    out = torch.empty((feat.shape[0],), dtype=torch.int32, device=feat.device)
    check(lib().tm_stage_palettize(_p(feat), _p(use), feat.shape[0], npal, max_iter, _p(out), _stream()))
    return out


def kmodes(rows, num_clusters, num_init=0, num_modalities=256, max_iter=-1):
    """TKModes.ComputeKModes (kmodes.pas:923-1094): rows uint8 numpy [n][80] (host, like the Pascal arrays) ->
    (labels int32 [n], centroids uint8 [k][80], cost, iterations of the best run)"""
    import numpy as np
    rows = np.ascontiguousarray(rows, np.uint8)
    assert rows.ndim == 2 and rows.shape[1] == 80
    labels = np.zeros(rows.shape[0], np.int32)
    cent = np.zeros((num_clusters, 80), np.uint8)
    cost, iters = ctypes.c_uint64(), ctypes.c_int()
    check(lib().tm_stage_kmodes(rows.ctypes.data_as(ctypes.c_void_p), rows.shape[0], num_clusters, num_init, num_modalities, max_iter,
                                labels.ctypes.data_as(ctypes.c_void_p), cent.ctypes.data_as(ctypes.c_void_p), ctypes.byref(cost), ctypes.byref(iters), _stream()))
    return labels, cent, cost.value, iters.value


def dl3quant(rgb, quant_to, lookup_bpc):
    """dl3quant (dlquant/quantizer.c:437-455): rgb = torch uint8 CUDA tensor [n][3] (R, G, B) -> (palette uint8 CUDA [3][quant_to] planar,
    number of colours left)"""
    import torch
    assert rgb.is_cuda and rgb.dtype == torch.uint8 and rgb.ndim == 2 and rgb.shape[1] == 3
    rgb = rgb.contiguous()
    pal = torch.zeros((3, quant_to), dtype=torch.uint8, device=rgb.device)
    n = ctypes.c_int()
    L = lib()
    L.tm_stage_dl3quant.restype = ctypes.c_int
    L.tm_stage_dl3quant.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]
    check(L.tm_stage_dl3quant(ctypes.c_void_p(rgb.data_ptr()), rgb.shape[0], int(quant_to), int(lookup_bpc), ctypes.c_void_p(pal.data_ptr()), ctypes.byref(n), _stream()))
    return pal, n.value


def kmodes_dev(rows, num_clusters, num_init=0, num_modalities=256, max_iter=-1):
    """TKModes.ComputeKModes on device memory: rows torch uint8 CUDA [n][80] -> (labels int32 CUDA [n], centroids uint8 CUDA [k][80], cost,
    iterations of the best run, points x iterations of all runs)"""
    import torch
    assert rows.is_cuda and rows.dtype == torch.uint8 and rows.ndim == 2 and rows.shape[1] == 80
    rows = rows.contiguous()
    labels = torch.zeros(rows.shape[0], dtype=torch.int32, device=rows.device)
    cent = torch.zeros((num_clusters, 80), dtype=torch.uint8, device=rows.device)
    cost, iters, pit = ctypes.c_uint64(), ctypes.c_int(), ctypes.c_int64()
    L = lib()
    L.tm_stage_kmodes_dev.restype = ctypes.c_int
    L.tm_stage_kmodes_dev.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int64), ctypes.c_void_p]
    check(L.tm_stage_kmodes_dev(ctypes.c_void_p(rows.data_ptr()), rows.shape[0], num_clusters, num_init, num_modalities, max_iter, ctypes.c_void_p(labels.data_ptr()),
                                ctypes.c_void_p(cent.data_ptr()), ctypes.byref(cost), ctypes.byref(iters), ctypes.byref(pit), _stream()))
    return labels, cent, cost.value, iters.value, pit.value
