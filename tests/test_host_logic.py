"""Host-side logic that needs no GPU: synthetic generator, shard maths, oracle k-means/dither invariants, encoder mirror."""
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
