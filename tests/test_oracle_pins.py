"""Pins the CPU oracle to everything the reference itself asserts for this path: the properties of TTilingEncoder.Test
(tilingencoder.pas:3847-3902) -- the only known-answer checks in the reference tree -- plus table/structure checks derived
from utils.pas constants.  Runs on CPU."""
import numpy as np
import pytest


def test_colour_round_trips(oracle):
    """3857-3867: 10 001 random colours survive RGB->LAB->RGB and RGB->YUV->RGB exactly"""
    rng = np.random.default_rng(20241218)
    for _ in range(10001):
        c = int(rng.integers(0, (1 << 24) - 1))
        r, g, b = c & 255, (c >> 8) & 255, (c >> 16) & 255
        assert oracle.lab_to_rgb(*oracle.rgb_to_lab(r, g, b)) == c
        assert oracle.yuv_to_rgb(*oracle.rgb_to_yuv(r, g, b)) == c
        assert oracle.lab_to_rgb(*oracle.rgb_to_lab(r, g, b, det=True)) == c  # the build's +,-,*,/ cube root too


def _test_tile():
    t = np.zeros(64, np.uint32)
    for i in range(8):
        for j in range(8):
            t[i * 8 + j] = (((i * j) & 255) << 16) | (((j * 32) & 255) << 8) | (i * 8)  # ToRGB(i*8, j*32, i*j), 3872-3874
    return t


@pytest.mark.parametrize("mode", [0, 1])  # pvsDCT, pvsWeightedDCT (3876-3893)
def test_dct_forward_inverse_exact(oracle, mode):
    t = _test_tile()
    dct = oracle.features_f64(t, mode)
    assert np.array_equal(oracle.inv_features_f64(dct, mode), t)


def test_det_cbrt_is_correctly_rounded_enough(oracle):
    xs = np.linspace(0.008856, 1.3, 5001)
    got = np.array([oracle.L.tmo_cbrt_det(float(x)) for x in xs])
    assert np.max(np.abs(got - np.cbrt(xs)) / np.cbrt(xs)) < 3e-16
    # Lab through the deterministic root equals Lab through libm pow after narrowing to Single, on a dense colour sample
    rng = np.random.default_rng(1)
    for c in rng.integers(0, 1 << 24, size=4000):
        r, g, b = int(c) & 255, (int(c) >> 8) & 255, (int(c) >> 16) & 255
        assert oracle.rgb_to_lab(r, g, b) == oracle.rgb_to_lab(r, g, b, det=True)


def test_lab_forms_agree_on_every_colour(oracle):
    """the whole domain, not a sample: over all 2^24 colours RGBToLAB through libm pow (the reference's power(), utils.pas:403), through
    the deterministic Newton root (the oracle's definition) and through the division-free form the kernels run give the same Singles"""
    import ctypes
    out = (ctypes.c_int64 * 2)()
    oracle.L.tmo_lab_domain_check(out)
    assert list(out) == [0, 0]


def test_tables(oracle):
    import ctypes
    snake = np.frombuffer((ctypes.c_uint8 * 64).in_dll(oracle.L, "tmo_dct_snake"), np.uint8)
    dmap = np.frombuffer((ctypes.c_uint8 * 64).in_dll(oracle.L, "tmo_dithering_map"), np.uint8)
    assert sorted(snake) == list(range(64)) and sorted(dmap) == list(range(64))  # both are permutations
    assert list(snake[:8]) == [0, 1, 5, 6, 14, 15, 27, 28] and snake[63] == 63  # utils.pas:60, :67
    w = np.frombuffer((ctypes.c_double * 192).in_dll(oracle.L, "tmo_dct_weights"), np.float64).reshape(3, 8, 8)
    assert w[0, 0, 0] == 1.6193873005 and w[2, 7, 7] == 0.285345396658
    assert np.allclose(w[0], w[0].T) and np.allclose(w[1], w[1].T)  # CSF tables are symmetric


def test_features_i16_follow_f64(oracle):
    """the int16 path (DCTInner_asm order) rounds the same transform the double path computes: |diff| <= 1"""
    rng = np.random.default_rng(5)
    tiles = rng.integers(0, 1 << 24, size=(50, 64), dtype=np.uint32)
    f16 = oracle.features_rgb(tiles, None, 1, False)
    for k in range(50):
        f64 = oracle.features_f64(tiles[k], 1)
        assert np.max(np.abs(f16[k] - f64)) <= 0.5 + 1e-3
    assert np.abs(f16).max() <= 13215  # SURVEY.md A.3 bound for pvsWeightedDCT


def test_ssd_and_sse_quirk(oracle):
    rng = np.random.default_rng(2)
    a = rng.integers(-13000, 13000, size=192).astype(np.int16)
    b = rng.integers(-13000, 13000, size=192).astype(np.int16)
    exact = int(((a.astype(np.int64) - b) ** 2).sum()) & 0xFFFFFFFF
    assert oracle.ssd(a, b) == exact
    # the asm twin (utils.pas:559-725) equals true L2 when blocks 5/6 of both halves agree and nothing saturates
    a2, b2 = (a // 8).astype(np.int16), (b // 8).astype(np.int16)
    for h in (0, 96):
        a2[h + 40:h + 56] = 0
        b2[h + 40:h + 56] = 0
    q = oracle.ssd_sse_quirk(a2, b2)
    blk7 = [(int(a2[56 + 2 * k]) - int(b2[56 + 2 * k])) ** 2 + (int(a2[57 + 2 * k]) - int(b2[57 + 2 * k])) ** 2 for k in range(4)]
    garbage = sum(((v & 0xFFFF) - (0x10000 if v & 0x8000 else 0)) ** 2 + (v >> 16) ** 2 for v in blk7)
    assert q == (oracle.ssd(a2, b2) + garbage) & 0xFFFFFFFF  # half 2 re-squares half 1's block-7 partial sums (xmm7)


def test_quicksort_matches_reference_semantics(oracle):
    """extern.pas:370-418 on bytes: result is sorted by key; equal keys may be permuted (unstable) but deterministically"""
    import ctypes
    rng = np.random.default_rng(3)
    luma = rng.integers(0, 1000, size=16).astype(np.int32)
    luma[3] = luma[7]
    lst = rng.integers(0, 16, size=64).astype(np.uint8)
    CMP = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p)

    def cmp(pa, pb, user):
        x = luma[ctypes.cast(pa, ctypes.POINTER(ctypes.c_uint8))[0]]
        y = luma[ctypes.cast(pb, ctypes.POINTER(ctypes.c_uint8))[0]]
        return int(x > y) - int(x < y)

    work = lst.copy()
    oracle.L.tmo_quicksort.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, CMP, ctypes.c_void_p]
    oracle.L.tmo_quicksort(work.ctypes.data_as(ctypes.c_void_p), 0, 63, 1, CMP(cmp), None)
    assert sorted(work) == sorted(lst)
    assert all(luma[work[i]] <= luma[work[i + 1]] for i in range(63))


def test_dedup_rules(oracle):
    rows = np.array([[5] * 64, [1] * 64, [5] * 64, [9] * 64, [1] * 64, [1] * 64], np.uint32)
    nu, rep, order, use, remap = oracle.dedup(rows, None)
    assert nu == 3 and list(order) == [1, 0, 3] and list(use) == [3, 2, 1]  # use desc, then content asc
    assert list(rep) == [0, 1, 0, 3, 1, 1] and list(remap) == [1, 0, 1, 2, 0, 0]
    rows[3, 63] = 0x80000000 | 9
    rows[0, 63] = 0x7FFFFFFF  # CompareDWord is unsigned: 0x80000009 sorts after 0x7FFFFFFF
    _, _, order, use, _ = oracle.dedup(rows[[0, 3]], None)
    assert list(order) == [0, 1]


def test_equal_quality_tile_count(oracle):
    import ctypes
    f = oracle.L.tmo_equal_quality_tile_count
    f.argtypes = [ctypes.c_double]
    assert 7 * f(4320000.0) == 320705  # 720p x 300
    assert 7 * f(32400000.0) == 994105  # 1080p x 1000
    assert min(7 * f(640.0), 640) == 640  # 64x64 x 10


def test_kdtree_equals_brute_force(oracle):
    """the baseline's kd-tree (ANN's published standard split and search, eps 0, bucket 32) returns what the brute force returns,
    lowest index among equal distances included"""
    rng = np.random.default_rng(7)
    db = rng.integers(-300, 301, size=(3000, 192)).astype(np.int16)
    db[:, 0] = rng.integers(0, 13000, size=3000)
    db[1500] = db[7]
    db[2999] = db[7]
    q = rng.integers(-300, 301, size=(200, 192)).astype(np.int16)
    q[:, 0] = rng.integers(0, 13000, size=200)
    q[0] = db[7]
    q[1] = db[2999]
    tree = oracle.kdtree_build(db, 32)
    idx, err, visited = oracle.kdtree_search1(tree, q)
    oracle.kdtree_free(tree)
    eidx, eerr = oracle.knn1(q, db)
    assert np.array_equal(idx, eidx) and np.array_equal(err, eerr)
    assert idx[0] == 7 and idx[1] == 7 and err[0] == 0
    assert 0 < visited <= q.shape[0] * db.shape[0]


def test_kmodes_restatement_properties(oracle):
    """A17 has no vector in the reference (parity unpinned); what can be checked of the restatement on the CPU: it recovers planted
    prototypes exactly when the noise is small, every point sits with its nearest mode of the final modes by the reference's
    dissimilarity (sum |a-b| + 2048 per differing byte, kmodes.pas:248-259), and the run is reproducible (the LCG is seeded, :933)"""
    rng = np.random.default_rng(5)
    proto = np.stack([np.full(80, v, np.uint8) for v in (3, 90, 200)])
    truth = rng.integers(0, 3, size=900)
    rows = proto[truth].copy()
    flip = rng.random(rows.shape) < 0.05
    rows[flip] = rng.integers(0, 256, size=int(flip.sum()))
    labels, cent, cost, iters = oracle.kmodes(rows, 3, 0, 256, -1)
    assert sorted(map(bytes, cent)) == sorted(map(bytes, proto))
    d = (np.abs(rows[:, None, :].astype(np.int64) - cent[None].astype(np.int64)).sum(2) + 2048 * (rows[:, None, :] != cent[None]).sum(2))
    assert np.array_equal(d.min(1), d[np.arange(900), labels])
    assert len(set(zip(truth.tolist(), labels.tolist()))) == 3
    again = oracle.kmodes(rows, 3, 0, 256, -1)
    assert np.array_equal(again[0], labels) and again[2] == cost and iters >= 1


def test_kmeans_pp_seeding_follows_its_stated_rule(oracle):
    """tmo_kmeans_pp_seeds against an independent big-integer restatement of the rule in DESIGN.md section 6: 64-bit LCG (MMIX
    constants) from 0x42381337, r = floor(x * total / 2^64) over the masses weight * nearest squared distance (weight alone for the
    first pick), first point whose running sum exceeds r; a zero total ends the seeding."""
    rng = np.random.default_rng(5)
    for n, d, k, wmax in ((1, 4, 3, 1), (7, 3, 4, 5), (300, 192, 16, 4_000_000), (50, 8, 60, 2 ** 32 - 1)):
        pts = rng.integers(-40000, 40000, size=(n, d)).astype(np.int32)
        pts[n // 2:] = pts[: n - n // 2]  # duplicates: fewer distinct points than k in the last case
        w = rng.integers(1, wmax + 1, size=n, dtype=np.uint64).astype(np.uint32)
        got = oracle.kmeans_pp_seeds(pts, w, k)
        state, seeds, mind = 0x42381337, [], None
        P = pts.astype(object)
        for _ in range(k):
            mass = [int(w[i]) * (1 if mind is None else mind[i]) for i in range(n)]
            total = sum(mass)
            if total == 0:
                break
            state = (state * 6364136223846793005 + 1442695040888963407) % (1 << 64)
            r = (state * total) >> 64
            run, pick = 0, None
            for i in range(n):
                run += mass[i]
                if run > r:
                    pick = i
                    break
            seeds.append(pick)
            dist = [int(sum((int(a) - int(b)) ** 2 for a, b in zip(P[i], P[pick]))) for i in range(n)]
            mind = dist if mind is None else [min(a, b) for a, b in zip(mind, dist)]
        assert list(got) == seeds
        assert len(set(seeds)) == len(seeds)


def test_powell_bracket_and_brent_against_scipy(oracle):
    """A11's scalar pieces against the code powell.pas says it was taken from: the fixture (tests/golden/powell_scipy.json, made by
    tests/golden/make_powell_fixtures.py) holds what scipy.optimize's own `bracket` and `Brent` compute on eight scalar functions, with
    powell.pas' exact golden-ratio constants in place of scipy's rounded literals and scipy's tolerance made absolute.
      * Bracket (powell.pas:56-147): the three points and the number of function evaluations, bit for bit, in every case.
      * Brent (:149-266): one evaluation more than scipy (the re-evaluation of the bracket's middle point, :262) and identical iterates
        wherever the fixture's run met none of the documented differences in the tolerance rule (`exact`); everywhere the minimiser
        within 4 xtol of scipy's and the value no worse than scipy's up to the function's change over that distance."""
    import ctypes
    import json
    import os
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "powell_scipy.json")))
    L = oracle.L
    L.tmo_test_scalar_fn.restype = ctypes.c_double
    L.tmo_test_scalar_fn.argtypes = [ctypes.c_int, ctypes.c_double]
    L.tmo_test_bracket.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.POINTER(ctypes.c_double)]
    L.tmo_test_brent.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    assert len(fx["bracket"]) >= 24 and len(fx["brent"]) >= 20
    for c in fx["bracket"]:
        r = (ctypes.c_double * 4)()
        L.tmo_test_bracket(c["fn"], c["xa"], c["xb"], r)
        assert sorted([r[0], r[2]]) == c["ends"] and r[1] == c["mid"] and int(r[3]) == c["calls"], (c, list(r))
    n_exact = 0
    for c in fx["brent"]:
        r = (ctypes.c_double * 3)()
        L.tmo_test_brent(c["fn"], c["xtol"], 100, r)
        x, fmin, calls = r[0], r[1], int(r[2])
        assert fmin == L.tmo_test_scalar_fn(c["fn"], x)
        if c["exact"]:
            n_exact += 1
            assert x == c["x"] and fmin == c["fx"] and calls == c["calls"] + 1, (c, list(r))
        assert abs(x - c["x"]) <= 4 * c["xtol"], (c, list(r))
        slack = max(abs(L.tmo_test_scalar_fn(c["fn"], c["x"] + s * 4 * c["xtol"]) - c["fx"]) for s in (-1.0, 1.0))
        assert fmin <= c["fx"] + slack, (c, list(r))
    assert n_exact >= 16
