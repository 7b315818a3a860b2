"""Host-side logic that needs no GPU: synthetic generator, shard maths, oracle k-means/dither invariants, encoder mirror."""
import ctypes

import numpy as np


def test_synth_is_seeded_and_has_duplicates():
    from tiler_amd import synth
    a = synth.video(3, 64, 48)
    b = synth.video(3, 64, 48)
    assert np.array_equal(a, b) and a.dtype == np.uint32 and a.shape == (3, 48, 64)
    assert (a >> 24 == 0xFF).all()
    assert not np.array_equal(a[0], a[1])
    # frozen tile columns: some 8x8 blocks repeat exactly between frames (unless hit by noise)
    same = sum(np.array_equal(a[0, y:y + 8, x:x + 8], a[1, y:y + 8, x:x + 8]) for y in range(0, 48, 8) for x in range(0, 64, 8))
    assert same >= 4


def test_frame_shard_covers_everything():
    from tiler_amd.distributed import frame_shard
    for n in (1, 7, 300, 1000):
        for w in (1, 2, 3, 4, 8):
            spans = [frame_shard(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_oracle_kmeans_invariants(oracle):
    rng = np.random.default_rng(0)
    pts = rng.integers(0, 256, size=(400, 3)).astype(np.int32)
    w = rng.integers(1, 9, size=400).astype(np.uint32)
    kk, assign, cent, iters = oracle.kmeans(pts, w, 8)
    assert kk == 8 and 0 < iters < 300
    for c in range(kk):  # fixed point: centroid = weighted mean of its members
        m = assign == c
        assert np.allclose(cent[c], (pts[m] * w[m, None]).sum(0) / w[m].sum())
    d = ((pts[:, None, :] - cent[None, :kk, :]) ** 2).sum(-1)
    assert np.array_equal(d.argmin(1), assign)
    # fewer distinct points than k: farthest-first stops early
    kk2, *_ = oracle.kmeans(np.array([[1, 2, 3]] * 5 + [[9, 9, 9]], np.int32), None, 4)
    assert kk2 == 2


def test_oracle_dither_picks_palette_colours(oracle):
    rng = np.random.default_rng(1)
    pal = rng.integers(0, 1 << 24, size=(1, 16)).astype(np.int32)
    tiles = np.repeat(pal[0, :4].astype(np.uint32)[:, None], 64, axis=1)  # flat tiles of exact palette colours
    out = oracle.dither(tiles, np.zeros(4, np.uint8), np.zeros(4, np.int32), pal, True)
    for k in range(4):
        assert (pal[0, out[k]] == pal[0, k]).all()  # zero error: every pick is that colour (or an identical duplicate)
    # mirror flags: dithering a flipped tile with its flags gives the flipped result of the natural tile
    t = rng.integers(0, 1 << 24, size=(1, 64)).astype(np.uint32)
    nat = oracle.dither(t, np.zeros(1, np.uint8), np.zeros(1, np.int32), pal, True)[0].reshape(8, 8)
    hm = t.reshape(8, 8)[:, ::-1].reshape(1, 64).copy()
    got = oracle.dither(hm, np.ones(1, np.uint8), np.zeros(1, np.int32), pal, True)[0].reshape(8, 8)
    assert np.array_equal(got[:, ::-1], nat)


def test_keyframes_rule(oracle):
    correl = np.ones(100, np.float32)
    correl[40] = 0.5
    correl[45] = 0.5  # within ShotTransMinSecondsPerKF of the previous keyframe at 24 fps: suppressed
    kf, n = oracle.find_keyframes(correl, 24.0)
    assert list(np.nonzero(kf)[0]) == [0, 40] and n == 2
    kf, n = oracle.find_keyframes(np.ones(800, np.float32), 24.0)
    assert list(np.nonzero(kf)[0]) == [0, 360, 720]  # ShotTransMaxSecondsPerKF = 15 s


def test_encoder_mirror_names():
    from tiler_amd.encoder import TEncoderStep, TPsyVisMode, TILEMAP_ITEM, TILE_HDR
    assert [s.name for s in TEncoderStep] == ["esAll", "esLoad", "esPredictMotion", "esReduce", "esPreparePalettes", "esDither",
                                              "esReconstruct", "esReindex", "esSave"]
    assert TEncoderStep.esAll == -1 and TEncoderStep.esSave == 7 and TPsyVisMode.pvsWeightedSpeDCT == 4
    assert TILEMAP_ITEM.itemsize == 18 and TILE_HDR.itemsize == 20


# ---- (f)#1 motion prediction: the oracle's restatements checked against independent statements ----------------------
def test_window_dcts_are_features_of_the_windows(oracle):
    rng = np.random.default_rng(3)
    fb = rng.integers(0, 1 << 24, (19, 26), dtype=np.uint32)
    win = oracle.window_dcts(fb)
    assert win.shape == (12 * 19, 192)
    for (y, x) in [(0, 0), (11, 18), (5, 7)]:
        tile = np.ascontiguousarray(fb[y:y + 8, x:x + 8]).reshape(1, 64)
        assert np.array_equal(win[y * 19 + x], oracle.features_rgb(tile, None, 1, False)[0])


def test_motion_search_finds_a_pure_shift(oracle):
    """a smooth image shifted by (+3, -2): interior tiles point back at (-3, +2); and the error of a perfect match is NOT 0
    (CompareEuclideanDCTPtr_asm subtracts block 5 of b from block 6, utils.pas:604-605) but what the quirk leaves"""
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, (3, 12, 14)).astype(np.float64)
    big = np.kron(base, np.ones((8, 8)))  # 96 x 112, piecewise constant -> blur a little
    for _ in range(3):
        big = (big + np.roll(big, 1, 1) + np.roll(big, 1, 2) + np.roll(big, 3, 1) + np.roll(big, 5, 2)) / 5
    img = (big.astype(np.uint32)[0] | (big.astype(np.uint32)[1] << 8) | (big.astype(np.uint32)[2] << 16))
    prev = np.ascontiguousarray(img[8:72, 8:88])             # 64 x 80 -> tile map 10 x 8
    cur = np.ascontiguousarray(img[8 - 2:72 - 2, 8 + 3:88 + 3])  # content moved by (-3, +2) in the new frame
    tm_w, tm_h = 10, 8
    tiles = np.ascontiguousarray(cur.reshape(tm_h, 8, tm_w, 8).transpose(0, 2, 1, 3).reshape(-1, 64))
    feats = oracle.features_rgb(tiles, None, 1, False)
    win = oracle.window_dcts(prev)
    err, px, py = oracle.motion_search(feats, tm_w, tm_h, win, 8)
    inner = np.zeros((tm_h, tm_w), bool)
    inner[1:-1, 1:-1] = True
    inner = inner.ravel()
    assert np.mean((px[inner] == 3) & (py[inner] == -2)) > 0.9
    i = int(np.nonzero(inner & (px == 3) & (py == -2))[0][0])
    sy, sx = divmod(i, tm_w)
    hit = win[(sy * 8 - 2) * (tm_w * 8 - 7) + sx * 8 + 3]
    assert np.array_equal(hit, feats[i])  # the very same vector ...
    b = hit.astype(np.int64)
    quirk = 0
    for half in (0, 96):  # ... leaves sum(b5^2) per half (block 6 term = (a6 - b5 - b6)^2), block 5 itself drops out
        quirk += int((b[half + 40:half + 48] ** 2).sum())
    assert int(err[i]) == quirk + 5 and oracle.L.tmo_ssd_i16_sse_quirk(ctypes.c_void_p(hit.ctypes.data), ctypes.c_void_p(hit.ctypes.data)) == quirk


def test_solve_tile_count_follows_golden_section(oracle):
    """independent restatement of GoldenRatioSearch (utils.pas:1044-1072) with a brute-force count"""
    rng = np.random.default_rng(9)
    mins = np.sort(rng.uniform(5, 51, 5000))
    inv_phi = 2 / (1 + 5 ** 0.5)
    for target in (100, 2500, 4999, 10 ** 6):
        mn, mx, last, n = 0.0, 10 * np.log(255 * 255 / 0.5) / np.log(10), None, 0
        while abs(mn - mx) > 1e-6:
            x = mn + (mx - mn) * (1 - inv_phi)
            y = int((mins <= x).sum())
            last, n = x, n + 1
            if abs(y - target) <= 0.5:
                break
            if y < target:
                mn = x
            else:
                mx = x
        x, probes = oracle.solve_tile_count(mins, target)
        assert probes == n and abs(x - last) < 1e-12


def test_dither_kernel_arithmetic_shortcuts():
    """the counting Thomas-Knoll kernel (tm_dither.hip) replaces two integer divisions by float arithmetic; both are exact over
    the whole range the kernel can meet, checked value by value here"""
    import numpy as np
    e = np.arange(-16400, 16401, dtype=np.int64)  # fed-back error, |e| <= 64 * 255
    ref = np.sign(e) * ((np.abs(e) * 9) // 100)  # Pascal div truncates toward zero (tilingencoder.pas:2589)
    got = np.trunc(e.astype(np.float32) * np.float32(0.09)).astype(np.int64)
    assert np.array_equal(ref, got)
    a = np.arange(0, 1 << 22, dtype=np.int64)  # |luma difference| before the div 1000 of ColorCompare (2323-2337)
    got = np.trunc(a.astype(np.float32) * np.float32(0.001) + np.float32(0.0005))  # numpy rounds the product, the kernel fuses: see below
    fused = np.trunc((a.astype(np.float64) * np.float64(np.float32(0.001)) + np.float64(np.float32(0.0005))).astype(np.float32))
    assert np.array_equal(fused.astype(np.int64), a // 1000)
    assert np.array_equal(got.astype(np.int64), a // 1000)
