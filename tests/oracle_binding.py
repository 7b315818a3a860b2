"""numpy/ctypes binding of the CPU oracle (oracle/tm_oracle.h).  Test infrastructure only."""
import ctypes

import numpy as np


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


class Oracle:
    def __init__(self, so):
        self.L = ctypes.CDLL(so)
        L = self.L
        L.tmo_ssd_i16.restype = ctypes.c_uint32
        L.tmo_ssd_i16_sse_quirk.restype = ctypes.c_uint32
        L.tmo_euclidean_to_psnr.restype = ctypes.c_float
        L.tmo_euclidean_to_psnr.argtypes = [ctypes.c_uint32]
        L.tmo_lab_to_rgb.restype = ctypes.c_int32
        L.tmo_lab_to_rgb.argtypes = [ctypes.c_float] * 3
        L.tmo_yuv_to_rgb.restype = ctypes.c_int32
        L.tmo_yuv_to_rgb.argtypes = [ctypes.c_float] * 3
        L.tmo_cbrt_det.restype = ctypes.c_double
        L.tmo_cbrt_det.argtypes = [ctypes.c_double]
        L.tmo_pearson.restype = ctypes.c_float
        L.tmo_dedup_u32.restype = ctypes.c_int64
        L.tmo_dedup_u8.restype = ctypes.c_int64
        L.tmo_find_keyframes.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                         ctypes.c_double, ctypes.c_void_p]
        L.tmo_knn1.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
        L.tmo_knnk.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p,
                               ctypes.c_void_p]
        L.tmo_dither_tiles.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.tmo_dedup_u32.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int] + [ctypes.c_void_p] * 5
        L.tmo_dedup_u8.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int] + [ctypes.c_void_p] * 5
        L.tmo_kmeans_i32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.tmo_quantize_palette.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.tmo_palettize_tiles.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.tmo_kmeans_pp_seeds.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.tmo_kmeans_pp_seeds.restype = ctypes.c_int

    # ---- colour
    def rgb_to_lab(self, r, g, b, det=False):
        l, a, bb = ctypes.c_float(), ctypes.c_float(), ctypes.c_float()
        (self.L.tmo_rgb_to_lab_det if det else self.L.tmo_rgb_to_lab)(r, g, b, ctypes.byref(l), ctypes.byref(a), ctypes.byref(bb))
        return l.value, a.value, bb.value

    def rgb_to_lab_array(self, rgb, det=True):
        """RGBToLAB of colours 0x00RRGGBB (uint32 [n]) -> float32 [n][3]"""
        rgb = np.ascontiguousarray(rgb, np.uint32)
        out = np.empty((rgb.shape[0], 3), np.float32)
        self.L.tmo_rgb_to_lab_array.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
        self.L.tmo_rgb_to_lab_array.restype = None
        self.L.tmo_rgb_to_lab_array(rgb.ctypes.data, rgb.shape[0], 1 if det else 0, out.ctypes.data)
        return out

    def rgb_to_yuv(self, r, g, b):
        y, u, v = ctypes.c_float(), ctypes.c_float(), ctypes.c_float()
        self.L.tmo_rgb_to_yuv(r, g, b, ctypes.byref(y), ctypes.byref(u), ctypes.byref(v))
        return y.value, u.value, v.value

    def lab_to_rgb(self, l, a, b):
        return self.L.tmo_lab_to_rgb(l, a, b) & 0xFFFFFFFF

    def yuv_to_rgb(self, y, u, v):
        return self.L.tmo_yuv_to_rgb(y, u, v) & 0xFFFFFFFF

    # ---- load side
    def load_from_image(self, img, tm_w, tm_h):
        img = np.ascontiguousarray(img, dtype=np.uint32)
        tiles = np.zeros((tm_w * tm_h, 64), np.uint32)
        self.L.tmo_load_from_image(_p(img), img.shape[1], img.shape[0], tm_w, tm_h, _p(tiles))
        return tiles

    def inter_frame_data(self, tiles):
        tiles = np.ascontiguousarray(tiles, dtype=np.uint32)
        out = np.zeros((tiles.shape[0], 3), np.float32)
        self.L.tmo_inter_frame_data(_p(tiles), tiles.shape[0], _p(out))
        return out

    def pearson(self, x, y):
        x = np.ascontiguousarray(x, np.float32).ravel()
        y = np.ascontiguousarray(y, np.float32).ravel()
        return float(self.L.tmo_pearson(_p(x), _p(y), x.size))

    def canonicalise(self, tiles):
        t = np.array(tiles, dtype=np.uint32, copy=True)
        flags = np.zeros(t.shape[0], np.uint8)
        self.L.tmo_canonicalise_tiles(_p(t), t.shape[0], _p(flags))
        return t, flags

    def find_keyframes(self, correl, fps, max_s=15.0, min_s=1.0, lo=0.8):
        c = np.ascontiguousarray(correl, np.float32)
        kf = np.zeros(c.size, np.uint8)
        n = self.L.tmo_find_keyframes(_p(c), c.size, fps, max_s, min_s, lo, _p(kf))
        return kf, n

    # ---- features
    def features_rgb(self, tiles, flags=None, mode=1, use_lab=False):
        tiles = np.ascontiguousarray(tiles, np.uint32)
        out = np.zeros((tiles.shape[0], 192), np.int16)
        f = np.ascontiguousarray(flags, np.uint8) if flags is not None else None
        self.L.tmo_tiles_features_i16(_p(tiles), tiles.shape[0], _p(f), mode, int(use_lab), _p(out))
        return out

    def features_pal(self, pal_px, pal_idx, palettes, mode=1):
        pal_px = np.ascontiguousarray(pal_px, np.uint8)
        pal_idx = np.ascontiguousarray(pal_idx, np.int32)
        palettes = np.ascontiguousarray(palettes, np.int32)
        out = np.zeros((pal_px.shape[0], 192), np.int16)
        self.L.tmo_paltiles_features_i16(_p(pal_px), _p(pal_idx), pal_px.shape[0], _p(palettes), palettes.shape[1], mode, _p(out))
        return out

    def features_cluster(self, tiles, mode=4):
        tiles = np.ascontiguousarray(tiles, np.uint32)
        out = np.zeros((tiles.shape[0], 192), np.int32)
        self.L.tmo_tiles_features_cluster_i32(_p(tiles), tiles.shape[0], mode, _p(out))
        return out

    def features_f64(self, tile, mode, use_lab=False):
        tile = np.ascontiguousarray(tile, np.uint32)
        cpn = np.zeros(192, np.float32)
        out = np.zeros(192, np.float64)
        self.L.tmo_cpn_from_rgb(_p(tile), int(use_lab), 0, 0, _p(cpn))
        self.L.tmo_features_f64(_p(cpn), mode, _p(out))
        return out

    def inv_features_f64(self, dct, mode, use_lab=False):
        dct = np.ascontiguousarray(dct, np.float64)
        out = np.zeros(64, np.uint32)
        self.L.tmo_inv_features_f64(_p(dct), mode, int(use_lab), _p(out))
        return out

    # ---- distances / knn
    def ssd(self, a, b):
        a = np.ascontiguousarray(a, np.int16)
        b = np.ascontiguousarray(b, np.int16)
        return int(self.L.tmo_ssd_i16(_p(a), _p(b)))

    def ssd_sse_quirk(self, a, b):
        a = np.ascontiguousarray(a, np.int16)
        b = np.ascontiguousarray(b, np.int16)
        return int(self.L.tmo_ssd_i16_sse_quirk(_p(a), _p(b)))

    def knn1(self, q, db):
        q = np.ascontiguousarray(q, np.int16)
        db = np.ascontiguousarray(db, np.int16)
        idx = np.zeros(q.shape[0], np.int32)
        err = np.zeros(q.shape[0], np.uint32)
        self.L.tmo_knn1(_p(q), q.shape[0], _p(db), db.shape[0], _p(idx), _p(err))
        return idx, err

    def kmodes(self, rows, k, num_init=0, nmod=256, max_iter=-1):
        rows = np.ascontiguousarray(rows, np.uint8)
        labels = np.zeros(rows.shape[0], np.int32)
        cent = np.zeros((k, 80), np.uint8)
        cost, iters = ctypes.c_uint64(), ctypes.c_int()
        self.L.tmo_kmodes(_p(rows), ctypes.c_int64(rows.shape[0]), k, num_init, nmod, max_iter, _p(labels), _p(cent), ctypes.byref(cost), ctypes.byref(iters))
        return labels, cent, cost.value, iters.value

    def dl3quant(self, rgb, quant_to, lookup_bpc):
        """dl3quant (dlquant/quantizer.c:437-455): rgb uint8 [n][3] -> (palette uint8 [3][quant_to] planar, colours left)"""
        rgb = np.ascontiguousarray(rgb, np.uint8)
        pal = np.zeros((3, quant_to), np.uint8)
        self.L.tmo_dl3quant.restype = ctypes.c_int
        n = self.L.tmo_dl3quant(_p(rgb), ctypes.c_int64(rgb.shape[0]), ctypes.c_int(quant_to), ctypes.c_int(lookup_bpc), _p(pal))
        return pal, n

    def kdtree_build(self, db, bucket=32):
        """exact kd-tree over db (kept alive by the returned handle's reference to the array)"""
        db = np.ascontiguousarray(db, np.int16)
        self.L.tmo_kdtree_build.restype = ctypes.c_void_p
        h = self.L.tmo_kdtree_build(_p(db), ctypes.c_int64(db.shape[0]), ctypes.c_int(bucket))
        return (ctypes.c_void_p(h), db)

    def kdtree_search1(self, tree, q):
        q = np.ascontiguousarray(q, np.int16)
        idx = np.zeros(q.shape[0], np.int32)
        err = np.zeros(q.shape[0], np.uint32)
        self.L.tmo_kdtree_search1.restype = ctypes.c_int64
        visited = self.L.tmo_kdtree_search1(tree[0], _p(q), ctypes.c_int64(q.shape[0]), _p(idx), _p(err))
        return idx, err, visited

    def kdtree_free(self, tree):
        self.L.tmo_kdtree_free(tree[0])

    def knnk(self, q, db, k):
        q = np.ascontiguousarray(q, np.int16)
        db = np.ascontiguousarray(db, np.int16)
        idx = np.zeros((q.shape[0], k), np.int32)
        err = np.zeros((q.shape[0], k), np.uint32)
        self.L.tmo_knnk(_p(q), q.shape[0], _p(db), db.shape[0], k, _p(idx), _p(err))
        return idx, err

    # ---- dither
    def dither(self, tiles, flags, pal_idx, palettes, use_tk=True, y2_mixed=4):
        tiles = np.ascontiguousarray(tiles, np.uint32)
        f = np.ascontiguousarray(flags, np.uint8) if flags is not None else None
        pal_idx = np.ascontiguousarray(pal_idx, np.int32)
        palettes = np.ascontiguousarray(palettes, np.int32)
        out = np.zeros((tiles.shape[0], 64), np.uint8)
        self.L.tmo_dither_tiles(_p(tiles), _p(f), _p(pal_idx), tiles.shape[0], _p(palettes), palettes.shape[1], int(use_tk), y2_mixed,
                                _p(out))
        return out

    # ---- dedup
    def dedup(self, rows, use_in=None):
        rows = np.ascontiguousarray(rows)
        n = rows.shape[0]
        rep = np.zeros(n, np.int64)
        order = np.zeros(n, np.int64)
        use_out = np.zeros(n, np.uint32)
        remap = np.zeros(n, np.int64)
        u = np.ascontiguousarray(use_in, np.uint32) if use_in is not None else None
        if rows.dtype == np.uint8:
            nu = self.L.tmo_dedup_u8(_p(rows), n, rows.shape[1], _p(u), _p(rep), _p(order), _p(use_out), _p(remap))
        else:
            rows = np.ascontiguousarray(rows, np.uint32)
            nu = self.L.tmo_dedup_u32(_p(rows), n, rows.shape[1], _p(u), _p(rep), _p(order), _p(use_out), _p(remap))
        return int(nu), rep, order[:nu].copy(), use_out[:nu].copy(), remap

    def epu_rerank(self, q, knn_idx, pal_px, tile_pal_idx, palettes):
        """FrameTilingExtendedPaletteUsage (tilingencoder.pas:1559-1610) -> (tile, pal, err) per query"""
        q = np.ascontiguousarray(q, np.int16)
        knn_idx = np.ascontiguousarray(knn_idx, np.int32)
        pal_px = np.ascontiguousarray(pal_px, np.uint8)
        tile_pal_idx = np.ascontiguousarray(tile_pal_idx, np.int32)
        palettes = np.ascontiguousarray(palettes, np.int32)
        nq = q.shape[0]
        t, p, e = np.zeros(nq, np.int32), np.zeros(nq, np.int32), np.zeros(nq, np.uint32)
        self.L.tmo_epu_rerank_batch.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        self.L.tmo_epu_rerank_batch(_p(q), nq, _p(knn_idx), knn_idx.shape[1], _p(pal_px), _p(tile_pal_idx), pal_px.shape[0], _p(palettes),
                                    palettes.shape[1], _p(t), _p(p), _p(e))
        return t, p, e

    # ---- motion prediction
    def window_dcts(self, fb):
        fb = np.ascontiguousarray(fb, np.uint32)
        h, w = fb.shape
        out = np.zeros(((h - 7) * (w - 7), 192), np.int16)
        self.L.tmo_window_dcts(_p(fb), w, h, _p(out))
        return out

    def motion_search(self, cur, tm_w, tm_h, win, radius):
        cur = np.ascontiguousarray(cur, np.int16)
        win = np.ascontiguousarray(win, np.int16)
        n = tm_w * tm_h
        err, px, py = np.zeros(n, np.uint32), np.zeros(n, np.int8), np.zeros(n, np.int8)
        self.L.tmo_motion_search(_p(cur), tm_w, tm_h, _p(win), radius, _p(err), _p(px), _p(py))
        return err, px, py

    def solve_tile_count(self, sorted_min_psnr, target):
        a = np.ascontiguousarray(sorted_min_psnr, np.float64)
        probes = ctypes.c_int()
        self.L.tmo_solve_tile_count.restype = ctypes.c_double
        self.L.tmo_solve_tile_count.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_double, ctypes.c_void_p]
        x = self.L.tmo_solve_tile_count(_p(a), a.size, float(target), ctypes.byref(probes))
        return x, probes.value

    def psnr(self, err):
        return np.array([self.L.tmo_euclidean_to_psnr(int(e)) for e in np.asarray(err).ravel()], np.float32).reshape(np.shape(err))

    # ---- k-means family
    def kmeans(self, pts, weights, k, max_iter=300):
        pts = np.ascontiguousarray(pts, np.int32)
        n, d = pts.shape
        w = np.ascontiguousarray(weights, np.uint32) if weights is not None else None
        assign = np.zeros(n, np.int32)
        cent = np.zeros((k, d), np.float64)
        iters = ctypes.c_int()
        kk = self.L.tmo_kmeans_i32(_p(pts), _p(w), n, d, k, max_iter, _p(assign), _p(cent), ctypes.byref(iters))
        return kk, assign, cent, iters.value

    def quantize_palette(self, pixels, pal_size, max_iter=300):
        pixels = np.ascontiguousarray(pixels, np.uint32).ravel()
        out = np.zeros(pal_size, np.int32)
        self.L.tmo_quantize_palette(_p(pixels), pixels.size, pal_size, max_iter, _p(out))
        return out

    def kmeans_pp_seeds(self, pts, weights, k):
        """the build's deterministic D^2 seeding: indices of the picked points (fewer than k when no distinct point is left)"""
        pts = np.ascontiguousarray(pts, np.int32)
        w = np.ascontiguousarray(weights, np.uint32) if weights is not None else None
        seeds = np.full(k, -1, np.int64)
        kk = self.L.tmo_kmeans_pp_seeds(_p(pts), _p(w), pts.shape[0], pts.shape[1], k, _p(seeds))
        return seeds[:kk]

    def palettize(self, feat, use, pal_count, max_iter=300):
        feat = np.ascontiguousarray(feat, np.int32)
        u = np.ascontiguousarray(use, np.uint32) if use is not None else None
        out = np.zeros(feat.shape[0], np.int32)
        self.L.tmo_palettize_tiles(_p(feat), _p(u), feat.shape[0], pal_count, max_iter, _p(out))
        return out
