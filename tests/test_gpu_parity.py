"""GPU parity: every HIP stage, called through the C ABI, against the CPU oracle on the same seeded inputs.
Bit-exact everywhere (integer/byte/index outputs, and floats because the build only uses IEEE +,-,*,/ in a fixed order)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu


def _dev(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint32:
        a = a.view(np.int32)
    return torch.from_numpy(a).cuda()


def _host_u32(t):
    return t.cpu().numpy().view(np.uint32)


@pytest.fixture(scope="module")
def frames():
    from tiler_amd import synth
    return synth.video(3, 100, 60)  # not a multiple of 8: exercises the crop/zero-fill edge (tile map 13x8)


@pytest.fixture(scope="module")
def tiles_flags(frames, oracle):
    tm_w, tm_h = 13, 8
    all_tiles, all_flags = [], []
    for f in range(frames.shape[0]):
        t = oracle.load_from_image(frames[f], tm_w, tm_h)
        c, fl = oracle.canonicalise(t)
        all_tiles.append(c)
        all_flags.append(fl)
    return np.concatenate(all_tiles), np.concatenate(all_flags)


def test_load_stage(frames, oracle):
    from tiler_amd import stages
    tm_w, tm_h = 13, 8
    tiles, flags, lab = stages.load(_dev(frames), tm_w, tm_h)
    torch.cuda.synchronize()
    exp_tiles, exp_flags, exp_lab = [], [], []
    for f in range(frames.shape[0]):
        t = oracle.load_from_image(frames[f], tm_w, tm_h)
        exp_lab.append(oracle.inter_frame_data(t))
        c, fl = oracle.canonicalise(t)
        exp_tiles.append(c)
        exp_flags.append(fl)
    assert np.array_equal(_host_u32(tiles), np.concatenate(exp_tiles))
    assert np.array_equal(flags.cpu().numpy(), np.concatenate(exp_flags))
    got = lab.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), np.concatenate(exp_lab).view(np.uint32)), "Lab tile means must match bit for bit"


@pytest.mark.parametrize("mode,use_lab,mirrors", [(1, False, False), (1, False, True), (0, False, False), (3, False, True),
                                                  (4, True, False), (4, False, True)])
def test_features_rgb(tiles_flags, oracle, mode, use_lab, mirrors):
    from tiler_amd import stages
    tiles, flags = tiles_flags
    exp = oracle.features_rgb(tiles, flags if mirrors else None, mode, use_lab)
    got = stages.features_rgb(_dev(tiles), _dev(flags) if mirrors else None, mode, use_lab).cpu().numpy()
    assert np.array_equal(got, exp)


def test_features_rgb_extremes(oracle):
    from tiler_amd import stages
    rng = np.random.default_rng(7)
    t = rng.integers(0, 1 << 24, size=(257, 64), dtype=np.uint32)  # odd count: ragged last workgroup
    t[0] = 0
    t[1] = 0xFFFFFF
    t[2, ::2] = 0xFFFFFF
    t[2, 1::2] = 0
    exp = oracle.features_rgb(t, None, 1, False)
    got = stages.features_rgb(_dev(t), None, 1, False).cpu().numpy()
    assert np.array_equal(got, exp)
    assert stages.features_rgb(_dev(t[:0]), None, 1, False).shape == (0, 192)


@pytest.mark.parametrize("mode,use_lab", [(1, False), (0, False), (3, False), (4, True), (4, False), (3, True)])
def test_features_first_look_agrees_with_reference_order(mode, use_lab, monkeypatch):
    """The int16 features take a separable first look at every coefficient and sum it in the reference's order only when the rounding
    is in doubt (tm_features.hip): over 300 000 tiles of every kind -- noise, flat, two-colour, ramps, near-black -- both forms give the
    same 57.6 M integers (TM_FEATURES_PLAIN=1 sums every coefficient the reference's way; that form is the one checked against the
    oracle above, at the oracle's sizes)."""
    from tiler_amd import stages
    g = torch.Generator(device="cuda").manual_seed(1234 + mode)
    n = 300000
    t = torch.randint(0, 1 << 24, (n, 64), generator=g, device="cuda", dtype=torch.int32)
    k = n // 6
    t[k:2 * k] = t[k:2 * k, :1]                                         # flat
    two = torch.randint(0, 1 << 24, (k, 2), generator=g, device="cuda", dtype=torch.int32)
    pick = torch.randint(0, 2, (k, 64), generator=g, device="cuda")
    t[2 * k:3 * k] = torch.gather(two, 1, pick)                         # two colours
    ramp = (torch.arange(64, device="cuda", dtype=torch.int32) % 8)[None, :] * torch.randint(0, 32, (k, 1), generator=g, device="cuda", dtype=torch.int32)
    base = torch.randint(0, 32, (k, 1), generator=g, device="cuda", dtype=torch.int32)
    v = (ramp + base).clamp_(0, 255)
    t[3 * k:4 * k] = v | (v << 8) | (v << 16)                           # grey ramps
    t[4 * k:5 * k] &= 0x070707                                          # near black
    flags = torch.randint(0, 4, (n,), generator=g, device="cuda", dtype=torch.uint8)
    monkeypatch.delenv("TM_FEATURES_PLAIN", raising=False)
    fast = stages.features_rgb(t, flags, mode, use_lab)
    monkeypatch.setenv("TM_FEATURES_PLAIN", "1")
    plain = stages.features_rgb(t, flags, mode, use_lab)
    monkeypatch.delenv("TM_FEATURES_PLAIN", raising=False)
    diff = (fast != plain)
    assert not bool(diff.any()), "%d of %d coefficients differ, first at %s" % (int(diff.sum()), diff.numel(), torch.nonzero(diff)[0].tolist())
    if mode == 4 and use_lab:  # the clustering's int32 features (k_features_cluster_i32) take the same first look
        cfast = stages.features_cluster(t, mode)
        monkeypatch.setenv("TM_FEATURES_PLAIN", "1")
        cplain = stages.features_cluster(t, mode)
        monkeypatch.delenv("TM_FEATURES_PLAIN", raising=False)
        assert torch.equal(cfast, cplain)


def _tile_kinds(n, seed):
    """noise, flat, two-colour, grey ramps, near-black and plain noise again: the kinds test_features_first_look_agrees_with_reference_order builds"""
    rng = np.random.default_rng(seed)
    t = rng.integers(0, 1 << 24, size=(n, 64), dtype=np.uint32)
    k = n // 6
    t[k:2 * k] = t[k:2 * k, :1]
    two = rng.integers(0, 1 << 24, size=(k, 2), dtype=np.uint32)
    t[2 * k:3 * k] = np.take_along_axis(two, rng.integers(0, 2, size=(k, 64)), axis=1)
    v = np.clip((np.arange(64) % 8)[None, :] * rng.integers(0, 32, size=(k, 1)) + rng.integers(0, 32, size=(k, 1)), 0, 255).astype(np.uint32)
    t[3 * k:4 * k] = v | (v << 8) | (v << 16)
    t[4 * k:5 * k] &= 0x070707
    return t, rng.integers(0, 4, size=n).astype(np.uint8)


def test_features_default_path_against_the_oracle_at_scale(oracle):
    """the shipped (separable first look) int16 features of 120 000 tiles of every kind against the ORACLE's reference-order sums, for the
    two forms the pipeline runs: pvsWeightedDCT on YUV with mirrors (the search's features, A5 / A13 / A14) and the clustering's int32
    features (pvsWeightedSpeDCT on Lab, A6).  The in-doubt branch fires for a few coefficients per thousand: ~70 000 of them here."""
    from concurrent.futures import ThreadPoolExecutor
    from tiler_amd import stages
    n = 120000
    t, flags = _tile_kinds(n, 2025)
    parts = [slice(i * n // 8, (i + 1) * n // 8) for i in range(8)]
    with ThreadPoolExecutor(8) as ex:  # (ctypes drops the GIL)
        exp = np.concatenate(list(ex.map(lambda sl: oracle.features_rgb(t[sl], flags[sl], 1, False), parts)))
        expc = np.concatenate(list(ex.map(lambda sl: oracle.features_cluster(t[sl], 4), parts)))
    got = stages.features_rgb(_dev(t), _dev(flags), 1, False).cpu().numpy()
    bad = np.argwhere(got != exp)
    assert bad.size == 0, "%d coefficients differ, first at tile %d coefficient %d" % (bad.shape[0], bad[0][0], bad[0][1])
    gotc = stages.features_cluster(_dev(t), 4).cpu().numpy()
    assert np.array_equal(gotc, expc)


def test_features_first_look_in_every_source_instantiation(monkeypatch):
    """the first look's margin in the kernel's other sources: palette-indexed tiles (k_features_i16<1>: small palettes, extreme
    colours), all 8 x 8 windows of a frame buffer (<2>), the (tile, palette) table (<3>) and pairs (<4>) through the k-nearest re-rank's
    two paths -- default against TM_FEATURES_PLAIN=1 (every coefficient in the reference's order)"""
    from tiler_amd import stages
    g = torch.Generator(device="cuda").manual_seed(808)
    n, npal = 200000, 6
    palettes = torch.randint(0, 1 << 24, (npal, 16), generator=g, device="cuda", dtype=torch.int32)
    palettes[0, :] = torch.tensor([0, 0xFFFFFF] * 8, device="cuda", dtype=torch.int32)          # extremes
    palettes[1, :] = torch.arange(16, device="cuda", dtype=torch.int32) * 0x010101               # near black
    palettes[2, 2:] = palettes[2, :1]                                                              # three live colours
    pal_idx = torch.randint(0, npal, (n,), generator=g, device="cuda", dtype=torch.int32)
    pal_px = torch.randint(0, 16, (n, 64), generator=g, device="cuda", dtype=torch.uint8)
    pal_px[: n // 4] = pal_px[: n // 4] & 1                                                        # two-colour tiles
    fb = torch.randint(0, 1 << 24, (136, 264), generator=g, device="cuda", dtype=torch.int32)
    fb[:, :64] = fb[:1, :64]
    fb[40:80] &= 0x0F0F0F
    q = stages.features_pal(pal_px[:3000], pal_idx[:3000], palettes, 1)
    knn_idx, _ = stages.knn_topk(q, stages.features_pal(pal_px[:20000], pal_idx[:20000], palettes, 1), 64)

    def run():
        a = stages.features_pal(pal_px, pal_idx, palettes, 1)
        b = stages.window_dcts(fb)
        c = stages.epu_rerank(q, knn_idx, pal_px[:20000], pal_idx[:20000], palettes)
        return a, b, c
    monkeypatch.delenv("TM_FEATURES_PLAIN", raising=False)
    monkeypatch.delenv("TM_EPU_TABLE_GIB", raising=False)
    fa, fb_, fc = run()
    monkeypatch.setenv("TM_EPU_TABLE_GIB", "0")   # the pairs asked for, not the table
    _, _, fd = run()
    monkeypatch.setenv("TM_FEATURES_PLAIN", "1")
    _, _, pd = run()
    monkeypatch.delenv("TM_EPU_TABLE_GIB")
    pa, pb, pc = run()
    assert torch.equal(fa, pa) and torch.equal(fb_, pb)
    for x, y in zip(fc + fd, pc + pd):
        assert torch.equal(x, y)
    # the windows by strips (k_window_dcts: conversions and row transforms shared between windows) against the window-at-a-time kernel
    monkeypatch.delenv("TM_FEATURES_PLAIN")
    monkeypatch.setenv("TM_WINDOW_DCTS_BY_TILE", "1")
    assert torch.equal(stages.window_dcts(fb), fb_)
    for hh, ww_ in ((8, 8), (9, 41), (17, 8), (23, 77)):  # ragged strips, a single window
        small = fb[:hh, :ww_].contiguous()
        by_tile = stages.window_dcts(small)
        monkeypatch.delenv("TM_WINDOW_DCTS_BY_TILE")
        assert torch.equal(stages.window_dcts(small), by_tile)
        monkeypatch.setenv("TM_WINDOW_DCTS_BY_TILE", "1")


@pytest.mark.parametrize("mode,use_lab", [(1, False), (0, False), (1, True)])
def test_features_eight_tiles_a_wave_against_a_tile_at_a_time(monkeypatch, mode, use_lab):
    """k_features_tiles8 (a lane per tile row / column, fast 8-point DCTs, the default for RGB tiles in the plain DCT modes) against
    k_features_i16<0> (TM_FEATURES_BY_TILE=1) and against every coefficient in the reference's order (TM_FEATURES_PLAIN=1), over every
    kind of tile, with and without mirror flags, at counts that leave the last wave ragged"""
    from tiler_amd import stages
    t, flags = _tile_kinds(60011, 77 + mode)
    td, fd = _dev(t), _dev(flags)
    for n in (60011, 4099, 9, 8, 7, 1):
        for fl in (None, fd[:n].contiguous()):
            monkeypatch.delenv("TM_FEATURES_BY_TILE", raising=False)
            monkeypatch.delenv("TM_FEATURES_PLAIN", raising=False)
            got = stages.features_rgb(td[:n].contiguous(), fl, mode, use_lab)
            monkeypatch.setenv("TM_FEATURES_BY_TILE", "1")
            by_tile = stages.features_rgb(td[:n].contiguous(), fl, mode, use_lab)
            assert torch.equal(got, by_tile), (n, fl is not None)
            if n <= 4099:
                monkeypatch.delenv("TM_FEATURES_BY_TILE")
                monkeypatch.setenv("TM_FEATURES_PLAIN", "1")
                assert torch.equal(stages.features_rgb(td[:n].contiguous(), fl, mode, use_lab), got), (n, fl is not None)


def test_features_pal_and_cluster(tiles_flags, oracle):
    from tiler_amd import stages
    tiles, _ = tiles_flags
    rng = np.random.default_rng(3)
    n = tiles.shape[0]
    palettes = rng.integers(0, 1 << 24, size=(5, 16), dtype=np.int32)
    pal_idx = rng.integers(0, 5, size=n, dtype=np.int32)
    pal_px = rng.integers(0, 16, size=(n, 64), dtype=np.uint8)
    exp = oracle.features_pal(pal_px, pal_idx, palettes, 1)
    got = stages.features_pal(_dev(pal_px), _dev(pal_idx), _dev(palettes), 1).cpu().numpy()
    assert np.array_equal(got, exp)
    expc = oracle.features_cluster(tiles, 4)
    gotc = stages.features_cluster(_dev(tiles), 4).cpu().numpy()
    assert np.array_equal(gotc, expc)


def _rand_features(rng, n, spread):
    """int16[192] rows shaped like real features: a few wide (DC-like) columns, the rest narrow"""
    f = rng.integers(-spread, spread + 1, size=(n, 192)).astype(np.int32)
    f[:, 0] = rng.integers(0, 13216, size=n)  # Y DC is non-negative (SURVEY.md A.3 bounds keep every SSD < 2^31)
    f[:, 64] = rng.integers(-6500, 6501, size=n)
    f[:, 128] = rng.integers(-9000, 9001, size=n)
    f[:, 1:6] = rng.integers(-3000, 3001, size=(n, 5))
    return f.astype(np.int16)


@pytest.mark.parametrize("nq,nt,spread", [(70, 100, 90), (1, 1, 90), (33, 31, 1200), (300, 1000, 600), (64, 2049, 100), (700, 5000, 300)])
def test_knn_exact(oracle, nq, nt, spread):
    from tiler_amd import stages
    rng = np.random.default_rng(nq * 1000 + nt)
    db = _rand_features(rng, nt, spread)
    q = _rand_features(rng, nq, spread)
    if nt > 10:  # plant exact duplicates so the lowest-index tie rule is exercised
        db[7] = db[3]
        q[0] = db[3]
        db[nt - 1] = db[nt - 2]
        q[nq - 1] = db[nt - 1]
    eidx, eerr = oracle.knn1(q, db)
    idx, err = stages.knn(_dev(q), _dev(db))
    assert np.array_equal(err.cpu().numpy().view(np.uint32), eerr)
    assert np.array_equal(idx.cpu().numpy(), eidx)


def _torch_nn(q, db):
    """exact nearest row by (SSD, index) in int64 on the GPU, 2048 queries at a time"""
    dq, dd = torch.from_numpy(q).cuda().to(torch.float64), torch.from_numpy(db).cuda().to(torch.float64)  # |values| < 2^15: products and sums exact
    nd = (dd * dd).sum(1)
    idx, err = [], []
    for a in range(0, q.shape[0], 2048):
        x = dq[a:a + 2048]
        d = (x * x).sum(1)[:, None] + nd[None, :] - 2.0 * (x @ dd.T)
        e, i = torch.min(d, dim=1)  # first minimum = lowest index
        first = (d == e[:, None]).to(torch.uint8).argmax(dim=1)
        idx.append(first.cpu().numpy())
        err.append(e.cpu().numpy())
    return np.concatenate(idx).astype(np.int32), np.concatenate(err).astype(np.uint64).astype(np.uint32)


@pytest.mark.parametrize("case", ["near-domain-limit", "norm-shells", "all-equal", "one-hot-columns"])
def test_knn_radial_bound_on_hostile_data(case):
    """the radial box dimension (norm over the non-box columns, rounded outwards) must never exclude the true neighbour: data built to
    stress its rounding -- norms next to the 2^31 domain limit, rows on a few thin norm shells (queries between them), a database of
    identical rows (R range of width zero), and rows whose energy sits in single columns"""
    from tiler_amd import stages
    rng = np.random.default_rng(len(case))
    nq, nt = 6000, 9000
    if case == "near-domain-limit":  # per-column range 2 * 2350: 192 * 4700^2 = 4.2e9 would be refused, so 120 wide columns + 72 narrow
        wide = rng.integers(-2100, 2101, size=(nt + nq, 120))
        narrow = rng.integers(-40, 41, size=(nt + nq, 72))
        allv = np.concatenate([wide, narrow], axis=1).astype(np.int16)
    elif case == "norm-shells":
        dirs = rng.normal(size=(nt + nq, 192))
        dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        radii = np.concatenate([rng.choice([500.0, 1500.0, 1501.0, 4000.0], size=nt), rng.choice([499.0, 1000.0, 1500.5, 2750.0, 4001.0], size=nq)])
        allv = np.rint(dirs * radii[:, None]).astype(np.int16)
    elif case == "all-equal":
        allv = np.tile(rng.integers(-300, 301, size=(1, 192)), (nt + nq, 1)).astype(np.int16)
        allv[nt:] += rng.integers(-3, 4, size=(nq, 192)).astype(np.int16)
    else:
        allv = np.zeros((nt + nq, 192), np.int16)
        allv[np.arange(nt + nq), rng.integers(0, 192, size=nt + nq)] = rng.integers(-1600, 1601, size=nt + nq).astype(np.int16)
    db, q = np.ascontiguousarray(allv[:nt]), np.ascontiguousarray(allv[nt:])
    eidx, eerr = _torch_nn(q, db)
    idx, err = stages.knn(_dev(q), _dev(db))
    assert np.array_equal(err.cpu().numpy().view(np.uint32), eerr)
    assert np.array_equal(idx.cpu().numpy(), eidx)


def _hostile(case, rng, nq, nt):
    if case == "near-domain-limit":
        wide = rng.integers(-2100, 2101, size=(nt + nq, 120))
        narrow = rng.integers(-40, 41, size=(nt + nq, 72))
        allv = np.concatenate([wide, narrow], axis=1).astype(np.int16)
    elif case == "norm-shells":
        dirs = rng.normal(size=(nt + nq, 192))
        dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        radii = np.concatenate([rng.choice([500.0, 1500.0, 1501.0, 4000.0], size=nt), rng.choice([499.0, 1000.0, 1500.5, 2750.0, 4001.0], size=nq)])
        allv = np.rint(dirs * radii[:, None]).astype(np.int16)
    elif case == "all-equal":
        allv = np.tile(rng.integers(-300, 301, size=(1, 192)), (nt + nq, 1)).astype(np.int16)
        allv[nt:] += rng.integers(-3, 4, size=(nq, 192)).astype(np.int16)
    elif case == "one-hot-columns":
        allv = np.zeros((nt + nq, 192), np.int16)
        allv[np.arange(nt + nq), rng.integers(0, 192, size=nt + nq)] = rng.integers(-1600, 1601, size=nt + nq).astype(np.int16)
    else:  # "features": rows shaped like tile features, with planted duplicates on both sides
        allv = _rand_features(rng, nt + nq, 250)
        allv[nt // 2] = allv[3]
        allv[nt - 1] = allv[3]
        allv[nt + 5] = allv[3]
        allv[nt + nq - 1] = allv[nt - 2]
    return np.ascontiguousarray(allv[:nt]), np.ascontiguousarray(allv[nt:])


@pytest.mark.parametrize("case", ["features", "near-domain-limit", "norm-shells", "all-equal", "one-hot-columns"])
def test_knn_dense_mode(case, monkeypatch):
    """BASELINE config 3: the scan with pruning off (TM_KNN_NOPRUNE=1) is a dense Q x T distance GEMM on the int8 MFMA pipe.  Same
    answers as the pruned scan and as an exact fp64 brute force, and it reports exactly nq x nt evaluated pairs (padding rows of the
    last tiles are not pairs)."""
    from tiler_amd import stages
    monkeypatch.setenv("TM_KNN_NOPRUNE", "1")
    rng = np.random.default_rng(len(case) + 77)
    nq, nt = 5003, 3001  # neither a multiple of 32 nor of a group's size
    db, q = _hostile(case, rng, nq, nt)
    eidx, eerr = _torch_nn(q, db)
    ix = stages.KnnIndex(_dev(db))
    idx, err = ix.search(_dev(q))
    _, _, pairs = ix.last_stats()
    ix.close()
    assert np.array_equal(err.cpu().numpy().view(np.uint32), eerr)
    assert np.array_equal(idx.cpu().numpy(), eidx)
    assert pairs == nq * nt


def test_knn_dense_mode_large(monkeypatch):
    """dense mode over >= 60 k database rows (several list rounds per workgroup), against the pruned scan and the exact brute force"""
    from tiler_amd import stages
    rng = np.random.default_rng(4242)
    nq, nt = 4100, 66000
    db, q = _hostile("features", rng, nq, nt)
    eidx, eerr = _torch_nn(q, db)
    pidx, perr = stages.knn(_dev(q), _dev(db))
    monkeypatch.setenv("TM_KNN_NOPRUNE", "1")
    ix = stages.KnnIndex(_dev(db))
    idx, err = ix.search(_dev(q))
    _, _, pairs = ix.last_stats()
    ix.close()
    assert np.array_equal(err.cpu().numpy().view(np.uint32), eerr) and np.array_equal(idx.cpu().numpy(), eidx)
    assert np.array_equal(perr.cpu().numpy().view(np.uint32), eerr) and np.array_equal(pidx.cpu().numpy(), eidx)
    assert pairs == nq * nt


def _plan_data(rng, nq, nt, ct, cq):
    """rows whose first ct (database) / cq (query) columns need a second int8 digit and whose other columns fit one: the digit plan becomes
    HT = ceil(ct / 32), HQ = ceil(cq / 32) (tm_knn.hip: make_plan_scaled)"""
    db = rng.integers(-25, 26, size=(nt, 192)).astype(np.int32)
    q = rng.integers(-25, 26, size=(nq, 192)).astype(np.int32)
    if ct:
        db[:, :ct] = rng.integers(-400, 401, size=(nt, ct))
    if cq:
        q[:, :cq] = rng.integers(-400, 401, size=(nq, cq))
    db[nt // 2:] = db[: nt - nt // 2]  # every row twice, far apart in index: ties everywhere, the lower index must win
    return db.astype(np.int16), q.astype(np.int16)


_SEEN_PLANS = set()


@pytest.mark.parametrize("hq", range(7))
@pytest.mark.parametrize("ht", range(7))
def test_knn_every_digit_plan(ht, hq):
    """every instantiation <HT, HQ> of the scan's kernels (7 x 7 digit plans: k_knn_seed, k_knn_consume and, below, its collection mode) is
    driven once and named by what the library says ran (tm_knn_last_plan): nearest neighbour against the fp64 brute force, pruned and dense.
    (Round 4's fault sat in one instantiation, <6, 6, collection>, that no test named.)"""
    from tiler_amd import stages
    rng = np.random.default_rng(1000 + ht * 7 + hq)
    nq, nt = 700, 4200
    db, q = _plan_data(rng, nq, nt, 32 * ht - (5 if ht else 0), 32 * hq - (5 if hq else 0))
    eidx, eerr = _torch_nn(q, db)
    idx, err = stages.knn(_dev(q), _dev(db))
    got = stages.knn_last_plan()
    assert got[:3] == (ht, hq, 0), "the data was built for <%d, %d>, the library planned %r" % (ht, hq, got)
    _SEEN_PLANS.add((ht, hq, 0))
    assert np.array_equal(err.cpu().numpy().view(np.uint32), eerr) and np.array_equal(idx.cpu().numpy(), eidx)


@pytest.mark.parametrize("ht,hq", [(6, 6), (0, 0), (6, 0), (0, 6), (3, 5), (5, 3), (1, 1), (2, 6), (6, 2), (4, 4), (1, 6), (6, 1), (2, 3), (3, 2), (5, 5), (4, 1), (1, 4)])
def test_knn_topk_digit_plans(ht, hq, monkeypatch):
    """the collection mode of the consume kernel (k = 64) on named digit plans, <6, 6> first, against the VALU brute force's exact top 64"""
    from tiler_amd import stages
    monkeypatch.delenv("TM_TOPK_BRUTE", raising=False)
    rng = np.random.default_rng(5000 + ht * 7 + hq)
    nq, nt, k = 300, 6000, 64
    db, q = _plan_data(rng, nq, nt, 32 * ht - (7 if ht else 0), 32 * hq - (7 if hq else 0))
    idx, err = stages.knn_topk(_dev(q), _dev(db), k)
    got = stages.knn_last_plan()
    assert got[:3] == (ht, hq, 1), "the data was built for <%d, %d, collection>, the library planned %r" % (ht, hq, got)
    _SEEN_PLANS.add((ht, hq, 1))
    monkeypatch.setenv("TM_TOPK_BRUTE", "1")
    bidx, berr = stages.knn_topk(_dev(q), _dev(db), k)
    assert torch.equal(err, berr) and torch.equal(idx, bidx)


def test_knn_plans_covered():
    """(runs after the two tests above in file order) every nearest-neighbour instantiation and the listed collection ones were named"""
    if not _SEEN_PLANS:
        pytest.skip("the plan tests did not run in this session")
    assert {(a, b, 0) for a in range(7) for b in range(7)} <= _SEEN_PLANS
    assert (6, 6, 1) in _SEEN_PLANS


@pytest.mark.parametrize("mode", ["nearest", "topk"])
def test_knn_list_arena_overflow_is_repeated_with_the_counted_size(mode, monkeypatch):
    """TM_KNN_ARENA_ENTRIES=64: the tile lists of the first attempt cannot fit, the scan reads the counted size back and runs again
    (knn_index_search / the collection pass): the result is the brute force's, and the retry counter moved.  (The arena's floor and the
    process-wide experience otherwise make this path depend on test order: ADVICE r04.)"""
    from tiler_amd import stages
    rng = np.random.default_rng(77)
    centres = _rand_features(rng, 30, 500).astype(np.int32)
    nt, nq = 30000, 6000
    db = (centres[rng.integers(0, 30, size=nt)] + rng.integers(-80, 81, size=(nt, 192))).astype(np.int16)
    q = (centres[rng.integers(0, 30, size=nq)] + rng.integers(-80, 81, size=(nq, 192))).astype(np.int16)
    monkeypatch.setenv("TM_KNN_ARENA_ENTRIES", "64")
    before = stages.knn_last_plan()[3]
    if mode == "nearest":
        eidx, eerr = _torch_nn(q, db)
        idx, err = stages.knn(_dev(q), _dev(db))
        assert np.array_equal(err.cpu().numpy().view(np.uint32), eerr) and np.array_equal(idx.cpu().numpy(), eidx)
    else:
        k = 64
        idx, err = stages.knn_topk(_dev(q[:1500]), _dev(db), k)
        monkeypatch.delenv("TM_KNN_ARENA_ENTRIES")
        monkeypatch.setenv("TM_TOPK_BRUTE", "1")
        bidx, berr = stages.knn_topk(_dev(q[:1500]), _dev(db), k)
        assert torch.equal(err, berr) and torch.equal(idx, bidx)
    assert stages.knn_last_plan()[3] > before, "the search was not repeated: the arena did not overflow"


def test_knn_high_chunk_masks_with_many_tiles():
    """list entries are tile << 8 | mask of the tile's non-zero high-digit chunks: more than 65 536 database tiles (2.2 M rows) whose high chunks
    are non-zero on most tiles, pruned scan against the exact fp64 scan on a sample; the guard word the consume loop leaves (a tile index or a
    segment out of range) is checked by the library on every search and would fail it (round 4's aperture fault: tile | mask << 24)"""
    from tiler_amd import stages
    g = torch.Generator(device="cuda").manual_seed(4242)
    nt, nq = 2_200_000, 20_000
    centres = torch.randint(-1500, 1501, (64, 192), generator=g, device="cuda", dtype=torch.int32)
    centres[:, 40:] //= 15  # (the wide columns are a prefix of 40: every possible SSD stays below 2^31)
    db = (centres[torch.randint(0, 64, (nt,), generator=g, device="cuda")] + torch.randint(-300, 301, (nt, 192), generator=g, device="cuda", dtype=torch.int32)).to(torch.int16)
    q = (centres[torch.randint(0, 64, (nq,), generator=g, device="cuda")] + torch.randint(-300, 301, (nq, 192), generator=g, device="cuda", dtype=torch.int32)).to(torch.int16)
    idx, err = stages.knn(q, db)
    ht, hq, _, _ = stages.knn_last_plan()
    assert ht >= 1 and hq >= 1 and nt // 32 > 65536
    dbd = db.to(torch.float64)
    dn = (dbd * dbd).sum(1)
    for s0 in range(0, 2048, 256):
        qq = q[s0:s0 + 256].to(torch.float64)
        d = (qq * qq).sum(1)[:, None] + dn[None, :] - 2.0 * (qq @ dbd.T)
        m = d.min(1).values
        first = (d == m[:, None]).to(torch.uint8).argmax(1)
        assert torch.equal(first.to(torch.int32), idx[s0:s0 + 256]) and torch.equal(m.to(torch.int64), err[s0:s0 + 256].to(torch.int64) & 0xFFFFFFFF)


def test_knn_many_groups_clustered():
    """the shipped (pruned) scan on clustered rows: many query groups, long tile lists, ties between tiles (identical rows far apart
    in index), queries that are database rows"""
    from tiler_amd import stages
    rng = np.random.default_rng(31337)
    centres = _rand_features(rng, 40, 600).astype(np.int32)
    nt, nq = 50000, 20011
    db = (centres[rng.integers(0, 40, size=nt)] + rng.integers(-60, 61, size=(nt, 192))).astype(np.int16)
    q = (centres[rng.integers(0, 40, size=nq)] + rng.integers(-60, 61, size=(nq, 192))).astype(np.int16)
    db[40000:40100] = db[100:200]  # identical rows in far-apart tiles of the sorted order? (same content -> same key -> neighbours; still ties)
    q[:300] = db[rng.integers(0, nt, size=300)]
    eidx, eerr = _torch_nn(q, db)
    idx, err = stages.knn(_dev(q), _dev(db))
    assert np.array_equal(err.cpu().numpy().view(np.uint32), eerr)
    assert np.array_equal(idx.cpu().numpy(), eidx)


def test_knn_refuses_out_of_domain_data():
    """arbitrary int16 rows can reach SSD >= 2^31, where mod-2^32 arithmetic stops being exact: refuse loudly"""
    from tiler_amd import stages, TileMotionError
    rng = np.random.default_rng(0)
    f = rng.integers(-20000, 20001, size=(64, 192)).astype(np.int16)
    with pytest.raises(TileMotionError):
        stages.knn(_dev(f), _dev(f))


def test_knn_real_features(tiles_flags, oracle):
    from tiler_amd import stages
    tiles, flags = tiles_flags
    feats = oracle.features_rgb(tiles, None, 1, False)
    db, q = feats[:150], feats
    eidx, eerr = oracle.knn1(q, db)
    idx, err = stages.knn(_dev(q), _dev(db))
    assert np.array_equal(idx.cpu().numpy(), eidx)
    assert np.array_equal(err.cpu().numpy().view(np.uint32), eerr)


def test_dither_thomas_knoll(tiles_flags, oracle):
    from tiler_amd import stages
    tiles, flags = tiles_flags
    tiles, flags = tiles[:120], flags[:120]
    rng = np.random.default_rng(11)
    palettes = rng.integers(0, 1 << 24, size=(4, 16), dtype=np.int32)
    palettes[1, 5:] = -65281  # cDitheringNullColor $FFFF00FF as int32: short palette
    palettes[2, 3] = palettes[2, 9]  # duplicate colour
    # two DIFFERENT colours with equal luma (299*15 - 587*9 + 114*7 = 0): the unstable-sort tie case of extern.pas:370
    r, g, b = 100, 120, 60
    palettes[3, 0] = (b << 16) | (g << 8) | r
    palettes[3, 1] = ((b + 7) << 16) | ((g - 9) << 8) | (r + 15)
    palettes[3, 2] = ((b + 14) << 16) | ((g - 18) << 8) | (r + 30)
    pal_idx = rng.integers(0, 4, size=tiles.shape[0], dtype=np.int32)
    exp = oracle.dither(tiles, flags, pal_idx, palettes, True)
    got = stages.dither(_dev(tiles), _dev(flags), _dev(pal_idx), _dev(palettes), True).cpu().numpy()
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("pal_size", [2, 4, 16, 17, 32, 64])
def test_dither_palette_sizes(tiles_flags, oracle, pal_size):
    """<= 16 live colours with distinct lumas take the counting kernel, everything else the literal-sort kernel"""
    from tiler_amd import stages
    tiles, flags = tiles_flags
    tiles, flags = tiles[:64], flags[:64]
    rng = np.random.default_rng(pal_size)
    palettes = rng.integers(0, 1 << 24, size=(3, pal_size), dtype=np.int32)
    palettes[2, pal_size // 2:] = -65281
    pal_idx = rng.integers(0, 3, size=tiles.shape[0], dtype=np.int32)
    exp = oracle.dither(tiles, flags, pal_idx, palettes, True)
    got = stages.dither(_dev(tiles), _dev(flags), _dev(pal_idx), _dev(palettes), True).cpu().numpy()
    assert np.array_equal(got, exp)


def test_dither_thomas_knoll_at_the_error_bound(oracle):
    """palettes that cannot reach the pixels (dark colours for white tiles, bright ones for black tiles, one channel only): the fed-back
    error grows by up to 255 a step, to the bound the counting kernel's int32 comparison is sized for (|t| <= 1700 per channel)"""
    from tiler_amd import stages
    rng = np.random.default_rng(3)
    tiles = np.empty((96, 64), np.int32)
    tiles[:24] = 0xFFFFFF
    tiles[24:48] = 0
    tiles[48:72] = rng.choice(np.array([0, 0xFFFFFF, 0xFF, 0xFF00, 0xFF0000, 0xFFFF], np.int32), size=(24, 64))
    tiles[72:] = rng.integers(0, 1 << 24, size=(24, 64), dtype=np.int32)
    flags = rng.integers(0, 4, size=96, dtype=np.uint8)
    palettes = np.empty((6, 16), np.int32)
    palettes[0] = np.arange(16) * 0x010101  # 16 greys 0..15
    palettes[1] = 0xFFFFFF - np.arange(16) * 0x010101  # 16 greys 240..255
    palettes[2] = np.arange(16)  # reds 0..15 only
    palettes[3] = (255 - np.arange(16)) << 16  # blues 240..255 only
    palettes[4] = [0, 0xFFFFFF] + [-65281] * 14  # two colours
    palettes[5] = [0x000001] + [-65281] * 15  # one colour
    pal_idx = (np.arange(96) % 6).astype(np.int32)
    exp = oracle.dither(tiles, flags, pal_idx, palettes, True)
    got = stages.dither(_dev(tiles), _dev(flags), _dev(pal_idx), _dev(palettes), True).cpu().numpy()
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("flavour", ["few-colours", "smooth", "no-duplicates"])
def test_dither_thomas_knoll_distinct_pairs_path(oracle, flavour):
    """enough tiles for launch_dither to plan every distinct (palette, colour) pair once and look the pixels up (tm_dither.hip, k_dd_*):
    few colours (almost everything is a duplicate), smooth gradients (neighbouring B values share bitmap words), and random pixels (no
    duplicates: the call falls back to the per-pixel kernel).  Palettes: full, short, one with a luma tie (literal-sort kernel, not
    part of the pairs path), a one-colour palette and one that no tile uses"""
    from tiler_amd import stages
    rng = np.random.default_rng(17)
    n = 3000
    if flavour == "few-colours":
        pool = rng.integers(0, 1 << 24, size=40, dtype=np.int32)
        tiles = pool[rng.integers(0, 40, size=(n, 64))]
    elif flavour == "smooth":
        base = rng.integers(0, 200, size=(n, 1, 3))
        ramp = (np.arange(64) // 8 + np.arange(64) % 8).reshape(1, 64, 1) * rng.integers(0, 4, size=(n, 1, 3))
        px = np.clip(base + ramp, 0, 255).astype(np.int64)
        tiles = (px[..., 0] | (px[..., 1] << 8) | (px[..., 2] << 16)).astype(np.int32)
    else:
        tiles = rng.integers(0, 1 << 24, size=(n, 64), dtype=np.int32)
    tiles = tiles | np.int32(rng.integers(0, 128)) << 24  # the top byte is not colour
    flags = rng.integers(0, 4, size=n, dtype=np.uint8)
    palettes = rng.integers(0, 1 << 24, size=(6, 16), dtype=np.int32)
    palettes[1, 6:] = -65281
    r, g, b = 100, 120, 60  # equal lumas, different colours: 299*15 - 587*9 + 114*7 = 0
    palettes[2, 0] = (b << 16) | (g << 8) | r
    palettes[2, 1] = ((b + 7) << 16) | ((g - 9) << 8) | (r + 15)
    palettes[3, 1:] = -65281
    pal_idx = rng.choice(np.array([0, 1, 2, 3, 5], np.int32), size=n)
    exp = oracle.dither(tiles, flags, pal_idx, palettes, True)
    got = stages.dither(_dev(tiles), _dev(flags), _dev(pal_idx), _dev(palettes), True).cpu().numpy()
    assert np.array_equal(got, exp)


def test_dither_distinct_pairs_with_palette_indices_out_of_range(oracle):
    """tiles that name no palette (-1, or one past the last) inside a call large enough for the distinct-pairs path: they come back as zeros,
    as from the per-pixel kernels, and do not disturb their neighbours"""
    from tiler_amd import stages
    rng = np.random.default_rng(29)
    n = 2048
    pool = rng.integers(0, 1 << 24, size=300, dtype=np.int32)
    tiles = pool[rng.integers(0, 300, size=(n, 64))]
    flags = rng.integers(0, 4, size=n, dtype=np.uint8)
    palettes = rng.integers(0, 1 << 24, size=(4, 16), dtype=np.int32)
    pal_idx = rng.integers(0, 4, size=n, dtype=np.int32)
    bad = rng.choice(n, size=40, replace=False)
    pal_idx[bad[:20]] = -1
    pal_idx[bad[20:]] = 4
    got = stages.dither(_dev(tiles), _dev(flags), _dev(pal_idx), _dev(palettes), True).cpu().numpy()
    good = np.ones(n, bool)
    good[bad] = False
    exp = oracle.dither(tiles[good], flags[good], pal_idx[good], palettes, True)
    assert np.array_equal(got[good], exp)
    assert not got[bad].any()


def test_dither_distinct_pairs_with_many_palettes(monkeypatch):
    """300 palettes over 80 000 tiles (the table of the distinct-pairs path: 19.7 M entries, 630 MB): the same bytes as a plan per pixel
    (TM_DITHER_NO_DEDUP), which the other dither tests hold against the oracle"""
    from tiler_amd import stages
    rng = np.random.default_rng(23)
    n, npal = 80000, 300
    pool = rng.integers(0, 1 << 24, size=5000, dtype=np.int32)
    tiles = pool[rng.integers(0, 5000, size=(n, 64))]
    flags = rng.integers(0, 4, size=n, dtype=np.uint8)
    palettes = rng.integers(0, 1 << 24, size=(npal, 16), dtype=np.int32)
    palettes[7, 9:] = -65281
    pal_idx = rng.integers(0, npal, size=n, dtype=np.int32)
    args = (_dev(tiles), _dev(flags), _dev(pal_idx), _dev(palettes), True)
    got = stages.dither(*args).cpu().numpy()
    monkeypatch.setenv("TM_DITHER_NO_DEDUP", "1")
    ref = stages.dither(*args).cpu().numpy()
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("path", ["hash", "hash-collisions", "hash-sort", "hash-sort-collisions", "plain"])
@pytest.mark.parametrize("kind", ["rgb", "pal"])
def test_dedup_reindex(tiles_flags, oracle, kind, path, monkeypatch):
    """the ways through the kernel: rows grouped by their 64-bit hash in a hash table (the default) or by a radix sort of the hashes
    (TM_DEDUP_SORT); either with the hash cut to 2 bits, so that different rows collide and the call has to notice and take the plain path;
    the plain path (merge sort of all rows) asked for outright"""
    from tiler_amd import stages
    for name in ("TM_DEDUP_DEGRADE_HASH", "TM_DEDUP_PLAIN", "TM_DEDUP_SORT"):
        monkeypatch.delenv(name, raising=False)
    if path.endswith("collisions"):
        monkeypatch.setenv("TM_DEDUP_DEGRADE_HASH", "1")
    if path.startswith("hash-sort"):
        monkeypatch.setenv("TM_DEDUP_SORT", "1")
    if path == "plain":
        monkeypatch.setenv("TM_DEDUP_PLAIN", "1")
    tiles, _ = tiles_flags
    rng = np.random.default_rng(5)
    if kind == "rgb":
        rows = np.concatenate([tiles, tiles[::3], tiles[5:40]])  # planted duplicates
        rows[-1, 63] ^= 0x80000000  # differs only in the last dword's top bit: unsigned compare matters
        use = None
    else:
        rows = rng.integers(0, 4, size=(700, 64), dtype=np.uint8)
        rows[:, :60] = 1  # long common prefix
        rows[100:200] = rows[0:100]
        rows[5, 0] = 200  # > 127: CompareByte is unsigned
        use = rng.integers(0, 4, size=700).astype(np.uint32)  # zero-use rows must drop out (ReindexTiles packs UseCount>0)
    nu, rep, order, use_out, remap = oracle.dedup(rows, use)
    g_nu, g_remap, g_order, g_use = stages.dedup(_dev(rows), _dev(use) if use is not None else None)
    assert g_nu == nu
    assert np.array_equal(g_order.cpu().numpy().astype(np.int64), order)
    assert np.array_equal(g_use.cpu().numpy().view(np.uint32), use_out)
    assert np.array_equal(g_remap.cpu().numpy().astype(np.int64), remap)


@pytest.mark.parametrize("longest", [8, 9])
@pytest.mark.parametrize("kind", ["rgb", "pal"])
def test_dedup_runs_of_equal_prefixes(oracle, kind, longest, monkeypatch):
    """the content sort's two legs: from TM_DEDUP_RADIX_MIN distinct rows on they go by a radix sort of their 8-byte prefixes, rows that share
    a prefix are put in order by whole-row compares inside their run (runs of 2..8 rows here, differing first in every possible later dword,
    top bits included), and a run of 9 sends the call to the comparator merge sort; every case against the oracle, and the merge sort
    outright (the default at this size) against the same"""
    from tiler_amd import stages
    rng = np.random.default_rng(77 + longest)
    if kind == "rgb":
        rows = rng.integers(0, 1 << 24, size=(3000, 64), dtype=np.int64).astype(np.uint32)
        at = 100
        for run in range(2, longest + 1):
            for rep in range(6):
                base = rows[at].copy()
                for j in range(run):
                    rows[at + j] = base
                    d = int(rng.integers(2, 64))  # the first dword that differs (the prefix is dwords 0 and 1)
                    rows[at + j, d:] = rng.integers(0, 1 << 24, size=64 - d)
                    if j & 1:
                        rows[at + j, d] |= 0x80000000  # unsigned compare
                at += run
        rows = rows.view(np.int32)
        use = None
    else:
        rows = rng.integers(0, 16, size=(3000, 64), dtype=np.uint8)
        at = 100
        for run in range(2, longest + 1):
            for rep in range(6):
                base = rows[at].copy()
                for j in range(run):
                    rows[at + j] = base
                    d = int(rng.integers(8, 64))  # (the prefix is the first 8 bytes)
                    rows[at + j, d:] = rng.integers(0, 256, size=64 - d)
                at += run
        use = rng.integers(1, 4, size=3000).astype(np.uint32)
    nu, rep_, order, use_out, remap = oracle.dedup(rows, use)
    for radix in (True, False):
        monkeypatch.delenv("TM_DEDUP_RADIX_MIN", raising=False)
        if radix:
            monkeypatch.setenv("TM_DEDUP_RADIX_MIN", "1")
        g_nu, g_remap, g_order, g_use = stages.dedup(_dev(rows), _dev(use) if use is not None else None)
        assert g_nu == nu
        assert np.array_equal(g_order.cpu().numpy().astype(np.int64), order)
        assert np.array_equal(g_use.cpu().numpy().view(np.uint32), use_out)
        assert np.array_equal(g_remap.cpu().numpy().astype(np.int64), remap)


@pytest.mark.parametrize("kind", ["rgb", "pal-use"])
def test_dedup_table_equals_the_hash_sort_at_scale(kind, monkeypatch):
    """the hash table against the front end it replaced (TM_DEDUP_SORT) on 2.4 M rows of which a third are duplicates, in runs from two to
    hundreds of copies scattered over the whole index range: distinct count, order, merged use counts and remap equal; the RGB case has more
    than 2^20 distinct rows, so its content sort is the radix sort of prefixes with whole-row ties (a planted family of rows that share
    their first two dwords included), the other carries use counts, zeros among them"""
    from tiler_amd import stages
    g = torch.Generator(device="cuda").manual_seed(2024)
    n, nd = 2_400_000, 1_600_000
    if kind == "rgb":
        base = torch.randint(0, 1 << 24, (nd, 64), generator=g, device="cuda", dtype=torch.int32)
        base[1000:1006, :2] = base[1000, :2]   # six distinct rows under one 8-byte prefix
        use = None
    else:
        base = torch.randint(0, 16, (nd, 64), generator=g, device="cuda", dtype=torch.int32).to(torch.uint8)
        use = torch.randint(0, 5, (n,), generator=g, device="cuda", dtype=torch.int32)
    src = torch.randint(0, nd, (n,), generator=g, device="cuda")
    src[: nd] = torch.arange(nd, device="cuda")            # every distinct row at least once
    src[nd: nd + 300] = 7                                  # one long run
    src = src[torch.randperm(n, generator=g, device="cuda")]
    rows = base[src].contiguous()
    monkeypatch.delenv("TM_DEDUP_SORT", raising=False)
    a = stages.dedup(rows, use)
    monkeypatch.setenv("TM_DEDUP_SORT", "1")
    b = stages.dedup(rows, use)
    torch.cuda.synchronize()
    assert a[0] == b[0] and a[0] <= nd
    for x, y in zip(a[1:], b[1:]):
        assert torch.equal(x, y)
    if kind == "rgb":
        assert a[0] == nd
        order = a[2].long()
        first = rows[order[:-1]].long() & 0xFFFFFFFF
        second = rows[order[1:]].long() & 0xFFFFFFFF
        neq = first != second
        col = neq.int().argmax(1)
        idx = torch.arange(order.numel() - 1, device="cuda")
        u = a[3].long()
        assert bool((u[:-1] >= u[1:]).all())                                       # use count descending ...
        same = u[:-1] == u[1:]
        assert bool((first[idx, col] < second[idx, col])[same].all())              # ... content strictly ascending among equal counts (CompareDWord: unsigned)


def test_lab_of_every_colour(oracle):
    """RGBToLAB (utils.pas:374-410) on the device over the WHOLE domain: all 2^24 colours through tm_stage_rgb_to_lab against the oracle's
    deterministic form (which tests/test_oracle_pins.py proves equal to the reference's libm power() on the same domain).  The kernels
    seed their cube root with the hardware's log2 / exp2, which no host restatement can follow bit for bit -- so the proof that only
    the narrowed Singles matter, and that they are the oracle's, is this exhaustive comparison on the device itself."""
    from tiler_amd import stages
    rgb = torch.arange(1 << 24, dtype=torch.int32, device="cuda")
    got = stages.rgb_to_lab(rgb).cpu().numpy()
    want = oracle.rgb_to_lab_array(np.arange(1 << 24, dtype=np.uint32), det=True)
    bad = np.flatnonzero((got.view(np.uint32) != want.view(np.uint32)).any(axis=1))
    assert bad.size == 0, (bad.size, [hex(int(c)) for c in bad[:5]], got[bad[:5]], want[bad[:5]])


@pytest.fixture(params=["resident", "launches", "resident-gave-up"])
def km_path(request, monkeypatch):
    """the tile k-means' skipping iterations: one resident launch for all of them (k_h_resident, <= 16 centroids), three launches each, or the
    resident launch treated as if its barrier had given up (the clustering is then repeated from its seeds through the launches)"""
    monkeypatch.delenv("TM_KM_LAUNCHES", raising=False)
    monkeypatch.delenv("TM_KM_RESIDENT_FAIL", raising=False)
    if request.param == "launches":
        monkeypatch.setenv("TM_KM_LAUNCHES", "1")
    elif request.param == "resident-gave-up":
        monkeypatch.setenv("TM_KM_RESIDENT_FAIL", "1")
    return request.param


@pytest.mark.parametrize("n,d,k", [(500, 3, 16), (40, 3, 64), (300, 192, 8), (1, 3, 4), (2000, 192, 40), (5000, 192, 16), (1500, 192, 3)])
def test_kmeans(oracle, n, d, k, km_path):
    from tiler_amd import stages
    if km_path != "resident" and not (d == 192 and k <= 16):
        pytest.skip("one path only")
    rng = np.random.default_rng(n + d + k)
    if d == 3:
        pts = rng.integers(0, 256, size=(n, 3)).astype(np.int32)
        if n > 10:
            pts[n // 2:] = pts[: n - n // 2] // 8 * 8  # repeated colours -> fewer distinct points than k in the small case
    else:
        centres = rng.integers(-3000, 3000, size=(5, d))
        pts = (centres[rng.integers(0, 5, size=n)] + rng.integers(-200, 200, size=(n, d))).astype(np.int32)
    w = rng.integers(1, 50, size=n).astype(np.uint32)
    kk, assign, cent, iters = oracle.kmeans(pts, w, k)
    g_kk, g_assign, g_cent, g_iters = stages.kmeans(_dev(pts), _dev(w), k)
    assert g_kk == kk and g_iters == iters
    assert np.array_equal(g_assign.cpu().numpy(), assign)
    assert np.array_equal(g_cent.cpu().numpy()[:kk].view(np.uint64), cent[:kk].view(np.uint64)), "centroids must match bit for bit"


def test_kmeans_192_list_kernel_switches_shape_inside_a_clustering(oracle, km_path):
    """(resident: 59 workgroups with a barrier between them, lists of several passes early on)  60 000 points in loose clusters, 16 centroids: the first skipping iterations list more than 8 192 unproven points (the list kernel's
    four-lanes-per-point passes), the later ones fewer (its lane-per-pair passes of 16 points): both shapes and the switch between them
    inside one clustering, against the oracle's plain Lloyd"""
    from tiler_amd import stages
    rng = np.random.default_rng(99)
    n, d, k = 60000, 192, 16
    centres = rng.integers(-1500, 1500, size=(40, d))
    pts = (centres[rng.integers(0, 40, size=n)] + rng.integers(-700, 701, size=(n, d))).astype(np.int32)
    w = rng.integers(1, 9, size=n).astype(np.uint32)
    kk, assign, cent, iters = oracle.kmeans(pts, w, k)
    g_kk, g_assign, g_cent, g_iters = stages.kmeans(_dev(pts), _dev(w), k)
    assert g_kk == kk and g_iters == iters
    assert np.array_equal(g_assign.cpu().numpy(), assign)
    assert np.array_equal(g_cent.cpu().numpy()[:kk].view(np.uint64), cent[:kk].view(np.uint64))


@pytest.mark.parametrize("k", [16, 40])
@pytest.mark.parametrize("case", ["lattice-ties", "tight-clusters", "one-cluster-far"])
def test_kmeans_192_skipping_iterations_are_exact(oracle, case, k, km_path):
    """(k = 16: through the resident launch and through the three launches per iteration; k = 40: the launches only)  the iterations that skip points whose bounds prove their assignment (k_h_bounds, k_assign192_list4) against the oracle's plain Lloyd
    on data that stresses them: lattice points with many EXACTLY equal distances (ties go to the lowest centroid), clusters tighter than
    the margins of any sloppy bound, and one far cluster (large centroid displacements in the first iterations); 40 centroids take three
    passes of 16 through the list kernel, whose four lanes per point each score four centroids of a pass"""
    from tiler_amd import stages
    if km_path != "resident" and k > 16:
        pytest.skip("one path only")
    rng = np.random.default_rng(len(case))
    n, d = 6000, 192
    if case == "lattice-ties":
        pts = (rng.integers(-1, 2, size=(n, d)) * 100).astype(np.int32)
        pts[:, 8:] = 0  # 3^8 lattice points, each many times: equal distances everywhere
    elif case == "tight-clusters":
        centres = rng.integers(-2000, 2000, size=(24, d))
        pts = (centres[rng.integers(0, 24, size=n)] + rng.integers(-2, 3, size=(n, d))).astype(np.int32)
    else:
        pts = rng.integers(-50, 51, size=(n, d)).astype(np.int32)
        pts[:40] += 20000
    w = rng.integers(1, 9, size=n).astype(np.uint32)
    kk, assign, cent, iters = oracle.kmeans(pts, w, k)
    g_kk, g_assign, g_cent, g_iters = stages.kmeans(_dev(pts), _dev(w), k)
    assert g_kk == kk and g_iters == iters
    assert np.array_equal(g_assign.cpu().numpy(), assign)
    assert np.array_equal(g_cent.cpu().numpy()[:kk].view(np.uint64), cent[:kk].view(np.uint64))


def test_quantize_and_palettize(tiles_flags, oracle):
    from tiler_amd import stages
    tiles, _ = tiles_flags
    tiles = tiles[:200]
    rng = np.random.default_rng(9)
    npal, pal_size = 5, 16
    pal_idx = rng.integers(0, npal - 1, size=tiles.shape[0], dtype=np.int32)  # palette 4 owns no tile: all null colours
    exp = np.stack([oracle.quantize_palette(tiles[pal_idx == p].ravel(), pal_size) for p in range(npal)])
    got = stages.quantize_palettes(_dev(tiles), _dev(pal_idx), npal, pal_size).cpu().numpy()
    assert np.array_equal(got, exp)
    assert (got[4] == -65281).all()  # cDitheringNullColor
    feat = oracle.features_cluster(tiles, 4)
    use = rng.integers(1, 9, size=tiles.shape[0]).astype(np.uint32)
    exp_idx = oracle.palettize(feat, use, 6)
    got_idx = stages.palettize(_dev(feat), _dev(use), 6).cpu().numpy()
    assert np.array_equal(got_idx, exp_idx)


@pytest.mark.parametrize("mixed", [1, 4, 16])
def test_dither_yliluoma(tiles_flags, oracle, mixed):
    """DitheringUseThomasKnoll=0: the live SSE4.1 restatement (tilingencoder.pas:2417-2504) incl. its 32-bit wrap"""
    from tiler_amd import stages
    tiles, flags = tiles_flags
    tiles, flags = tiles[:96], flags[:96]
    rng = np.random.default_rng(13 + mixed)
    palettes = rng.integers(0, 1 << 24, size=(3, 16), dtype=np.int32)
    palettes[1, 7:] = -65281
    pal_idx = rng.integers(0, 3, size=tiles.shape[0], dtype=np.int32)
    exp = oracle.dither(tiles, flags, pal_idx, palettes, False, mixed)
    got = stages.dither(_dev(tiles), _dev(flags), _dev(pal_idx), _dev(palettes), False, mixed).cpu().numpy()
    assert np.array_equal(got, exp)


def test_fine_seam_twins(oracle):
    """ann_kdtree_short_* / yakmo_* / bico_* called with host pointers exactly as extern.pas:182-223 declares them"""
    import ctypes
    from tiler_amd import lib
    L = lib()
    rng = np.random.default_rng(21)
    # ANN: rows as an array of row pointers (PPSmallint)
    db = _rand_features(rng, 300, 200)
    q = _rand_features(rng, 5, 200)
    rows = (ctypes.c_void_p * 300)(*[db[i].ctypes.data for i in range(300)])
    L.ann_kdtree_short_create.restype = ctypes.c_void_p
    tree = L.ann_kdtree_short_create(rows, 300, 192, 32, 0)
    assert tree
    eidx, eerr = oracle.knn1(q, db)
    err = ctypes.c_uint32()
    L.ann_kdtree_short_search.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
    for i in range(5):
        assert L.ann_kdtree_short_search(tree, q[i].ctypes.data, 0, ctypes.byref(err)) == eidx[i] and err.value == eerr[i]
    kidx, kerr = oracle.knnk(q[:1], db, 64)
    gi, ge = np.zeros(64, np.int32), np.zeros(64, np.uint32)
    L.ann_kdtree_short_search_multi.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint32]
    L.ann_kdtree_short_search_multi(tree, gi.ctypes.data, ge.ctypes.data, 64, q[0].ctypes.data, 0)
    assert np.array_equal(gi, kidx[0]) and np.array_equal(ge, kerr[0])
    L.ann_kdtree_short_destroy.argtypes = [ctypes.c_void_p]
    L.ann_kdtree_short_destroy(tree)
    # yakmo: rows of doubles (PPDouble), integer valued like QuantizeUsingYakmo's pixel dataset
    px = rng.integers(0, 256, size=(400, 3)).astype(np.float64)
    prow = (ctypes.c_void_p * 400)(*[px[i].ctypes.data for i in range(400)])
    L.yakmo_create.restype = ctypes.c_void_p
    y = L.yakmo_create(8, 1, 300, 1, 0, 0, 0)
    L.yakmo_load_train_data.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
    L.yakmo_load_train_data(y, 400, 3, prow)
    assign = np.zeros(400, np.int32)
    L.yakmo_train_on_data.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.yakmo_train_on_data(y, assign.ctypes.data)
    cent = np.zeros((8, 3), np.float64)
    crow = (ctypes.c_void_p * 8)(*[cent[i].ctypes.data for i in range(8)])
    L.yakmo_get_centroids.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.yakmo_get_centroids(y, crow)
    kk, eassign, ecent, _ = oracle.kmeans(px.astype(np.int32), None, 8)
    assert kk == 8 and np.array_equal(assign, eassign) and np.array_equal(cent, ecent)
    L.yakmo_destroy.argtypes = [ctypes.c_void_p]
    L.yakmo_destroy(y)
    # bico: weighted lines in, centres + weights out
    L.bico_create.restype = ctypes.c_void_p
    L.bico_create.argtypes = [ctypes.c_int64] * 5 + [ctypes.c_int]
    b = L.bico_create(3, 400, 2, 32, 16, 0x42381337)
    L.bico_insert_line.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double]
    w = rng.integers(1, 9, size=400)
    for i in range(400):
        L.bico_insert_line(b, px[i].ctypes.data, float(w[i]))
    bc, bw = np.zeros((16, 3)), np.zeros(16)
    L.bico_get_results.restype = ctypes.c_int64
    L.bico_get_results.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    n = L.bico_get_results(b, bc.ctypes.data, bw.ctypes.data)
    kk, eassign, ecent, _ = oracle.kmeans(px.astype(np.int32), w.astype(np.uint32), 16)
    assert n == kk and np.array_equal(bc[:kk], ecent[:kk]) and bw.sum() == w.sum()
    L.bico_destroy.argtypes = [ctypes.c_void_p]
    L.bico_destroy(b)


# ---- (f)#1 motion prediction -----------------------------------------------------------------------------------------
def _screen(tiles, flags, tm_w, tm_h):
    """un-mirror canonical tiles and lay them out as the frame buffer PredictMotion draws (tilingencoder.pas:1255-1260)"""
    t = tiles.reshape(-1, 8, 8).copy()
    for i, f in enumerate(flags):
        if f & 1:
            t[i] = t[i][:, ::-1]
        if f & 2:
            t[i] = t[i][::-1, :]
    return np.ascontiguousarray(t.reshape(tm_h, tm_w, 8, 8).transpose(0, 2, 1, 3).reshape(tm_h * 8, tm_w * 8))


def test_window_dcts(tiles_flags, oracle):
    from tiler_amd import stages
    tiles, flags = tiles_flags
    fb = _screen(tiles[:104], flags[:104], 13, 8)
    got = stages.window_dcts(_dev(fb))
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), oracle.window_dcts(fb))


@pytest.mark.parametrize("radius", [1, 2, 9, 32, 128])
def test_motion_search(tiles_flags, oracle, radius):
    """frame 1's tiles searched in frame 0's buffer: error (the asm's SSD quirks + manhattan penalty), offsets, first-minimum rule"""
    from tiler_amd import stages
    tiles, flags = tiles_flags
    tm_w, tm_h = 13, 8
    fb = _screen(tiles[:104], flags[:104], tm_w, tm_h)
    win = oracle.window_dcts(fb)
    cur = oracle.features_rgb(tiles[104:208], flags[104:208], 1, False)
    e_err, e_px, e_py = oracle.motion_search(cur, tm_w, tm_h, win, radius)
    err, px, py = stages.motion_search(_dev(cur), tm_w, tm_h, _dev(win), radius)
    torch.cuda.synchronize()
    assert np.array_equal(_host_u32(err), e_err)
    assert np.array_equal(px.cpu().numpy(), e_px) and np.array_equal(py.cpu().numpy(), e_py)


def test_motion_search_quirks_and_ties(oracle):
    """crafted vectors: saturating differences, the dropped block 5, the double subtraction in block 6, the re-squared
    block-7 pair sums, and many equal candidates (a constant buffer: the manhattan penalty then decides)"""
    from tiler_amd import stages
    rng = np.random.default_rng(11)
    tm_w, tm_h = 5, 4
    nwin = (tm_w * 8 - 7) * (tm_h * 8 - 7)
    win = rng.integers(-30000, 30000, (nwin, 192)).astype(np.int16)  # big values: psubsw saturates, sums wrap mod 2^32
    cur = rng.integers(-30000, 30000, (tm_w * tm_h, 192)).astype(np.int16)
    win[::7] = win[3]            # repeated vectors: equal errors up to the penalty
    cur[0] = win[3]              # an exact hit exists for tile 0 ... except the quirks make it non-zero
    for radius in (3, 16):
        e = oracle.motion_search(cur, tm_w, tm_h, win, radius)
        g = stages.motion_search(_dev(cur), tm_w, tm_h, _dev(win), radius)
        torch.cuda.synchronize()
        assert np.array_equal(_host_u32(g[0]), e[0]) and np.array_equal(g[1].cpu().numpy(), e[1]) and np.array_equal(g[2].cpu().numpy(), e[2])
    const = np.tile(rng.integers(-200, 200, (1, 192)).astype(np.int16), (nwin, 1))
    e = oracle.motion_search(cur, tm_w, tm_h, const, 32)
    g = stages.motion_search(_dev(cur), tm_w, tm_h, _dev(const), 32)
    torch.cuda.synchronize()
    assert np.array_equal(_host_u32(g[0]), e[0]) and np.array_equal(g[1].cpu().numpy(), e[1]) and np.array_equal(g[2].cpu().numpy(), e[2])
    assert np.all(e[1] == 0) and np.all(e[2] == 0)  # all candidates equal: the tile's own position wins


@pytest.mark.parametrize("tm_w,tm_h,radius,amp", [(19, 7, 32, 9000), (8, 4, 5, 16383), (33, 9, 17, 3000), (9, 13, 32, 10922), (12, 5, 32, 16384), (10, 6, 32, 12000)])
def test_motion_search_matrix_path(oracle, monkeypatch, tm_w, tm_h, radius, amp):
    """The matrix-core search (k_mo_search_mfma) against the oracle AND against the VALU kernel on the same data, on tile grids that are
    not multiples of its 4 x 8 groups, with coefficients up to the bounds under which it runs (+-16383 in the plain blocks, +-10922 in
    blocks 5 and 6: sums then wrap mod 2^32 like paddd) and just beyond them (the frame falls back to the VALU kernel on the device);
    repeated windows make equal candidates, which the raster-order rule settles"""
    from tiler_amd import stages
    rng = np.random.default_rng(tm_w * 1000 + tm_h * 10 + radius)
    nwin = (tm_w * 8 - 7) * (tm_h * 8 - 7)
    win = rng.integers(-amp, amp + 1, (nwin, 192)).astype(np.int16)
    cur = rng.integers(-amp, amp + 1, (tm_w * tm_h, 192)).astype(np.int16)
    q = min(amp, 10922) if amp != 12000 else 12000  # blocks 5 / 6 of both halves (amp 12000: beyond their bound -> fallback)
    for hf in (0, 96):
        win[:, hf + 40:hf + 56] = rng.integers(-q, q + 1, (nwin, 16))
        cur[:, hf + 40:hf + 56] = rng.integers(-q, q + 1, (tm_w * tm_h, 16))
    win[::11] = win[5]
    cur[3] = win[5]
    e = oracle.motion_search(cur, tm_w, tm_h, win, radius)
    g = stages.motion_search(_dev(cur), tm_w, tm_h, _dev(win), radius)
    torch.cuda.synchronize()
    assert np.array_equal(_host_u32(g[0]), e[0]) and np.array_equal(g[1].cpu().numpy(), e[1]) and np.array_equal(g[2].cpu().numpy(), e[2])
    monkeypatch.setenv("TM_MOTION_VALU", "1")
    v = stages.motion_search(_dev(cur), tm_w, tm_h, _dev(win), radius)
    torch.cuda.synchronize()
    assert np.array_equal(_host_u32(v[0]), e[0]) and np.array_equal(v[1].cpu().numpy(), e[1]) and np.array_equal(v[2].cpu().numpy(), e[2])


# ---- (f)#3 FrameTilingExtendedPaletteUsage -------------------------------------------------------------------------
@pytest.fixture(params=["by-size", "sampled", "bound"])
def topk_thresholds(request, monkeypatch):
    """where the k-nearest search's first thresholds come from: the shipped rule (a sample of the database for many queries against a database
    of some size, tm_knn.hip: knn_index_search_topk), the sample whenever the database has rows enough for one (TM_TOPK_ESTIMATE=1: queries that
    find fewer than k rows within their estimate are searched again), never (=0: the curve window's bound)"""
    monkeypatch.delenv("TM_TOPK_ESTIMATE", raising=False)
    if request.param != "by-size":
        monkeypatch.setenv("TM_TOPK_ESTIMATE", "1" if request.param == "sampled" else "0")
    return request.param


@pytest.mark.parametrize("nt,k", [(700, 64), (40, 64), (300, 5), (3000, 64)])
def test_knn_topk(oracle, nt, k, topk_thresholds):
    from tiler_amd import stages
    rng = np.random.default_rng(nt + k)
    db = _rand_features(rng, nt, 300)
    db[5] = db[2]; db[9] = db[2]  # equal distances: (distance, index) order
    q = _rand_features(rng, 150, 300)
    q[0] = db[2]
    eidx, eerr = oracle.knnk(q, db, k)
    idx, err = stages.knn_topk(_dev(q), _dev(db), k)
    torch.cuda.synchronize()
    assert np.array_equal(idx.cpu().numpy(), eidx)
    assert np.array_equal(_host_u32(err), eerr)


def test_knn_topk_on_norm_shells(oracle, topk_thresholds):
    """k nearest rows where the radial box dimension is most selective (rows on thin norm shells, queries between them)"""
    from tiler_amd import stages
    rng = np.random.default_rng(64)
    nt, nq = 4000, 300
    dirs = rng.normal(size=(nt + nq, 192))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    radii = np.concatenate([rng.choice([500.0, 1500.0, 1501.0, 4000.0], size=nt), rng.choice([499.0, 1000.0, 1500.5, 2750.0, 4001.0], size=nq)])
    allv = np.rint(dirs * radii[:, None]).astype(np.int16)
    db, q = np.ascontiguousarray(allv[:nt]), np.ascontiguousarray(allv[nt:])
    eidx, eerr = oracle.knnk(q, db, 64)
    idx, err = stages.knn_topk(_dev(q), _dev(db), 64)
    torch.cuda.synchronize()
    assert np.array_equal(idx.cpu().numpy(), eidx)
    assert np.array_equal(_host_u32(err), eerr)


def test_epu_rerank(tiles_flags, oracle):
    from tiler_amd import stages
    tiles, flags = tiles_flags
    rng = np.random.default_rng(77)
    nt, npal = 150, 5
    palettes = rng.integers(0, 1 << 24, size=(npal, 16), dtype=np.int32)
    pal_px = rng.integers(0, 16, size=(nt, 64), dtype=np.uint8)
    pal_px[40] = pal_px[3]                      # same pixels under different palettes
    tile_pal = rng.integers(0, npal, size=nt, dtype=np.int32)
    db = oracle.features_pal(pal_px, tile_pal, palettes, 1)
    q = oracle.features_rgb(tiles[:200], None, 1, False)
    idx64, _ = oracle.knnk(q, db, 64)
    idx64[7, 10:] = -1                           # a padded list
    idx64[8, :] = idx64[8, 0]                    # a single unique tile
    et, ep, ee = oracle.epu_rerank(q, idx64, pal_px, tile_pal, palettes)
    t, p, e = stages.epu_rerank(_dev(q), _dev(idx64), _dev(pal_px), _dev(tile_pal), _dev(palettes))
    torch.cuda.synchronize()
    assert np.array_equal(t.cpu().numpy(), et) and np.array_equal(p.cpu().numpy(), ep)
    assert np.array_equal(_host_u32(e), ee)
    assert (ep != tile_pal[et]).any()            # the re-rank does move tiles to other palettes


def test_epu_rerank_without_the_table(tiles_flags, oracle, monkeypatch):
    """the same re-rank when the table of every tile under every palette does not fit (TM_EPU_TABLE_GIB=0 forces it; PaletteCount = 1024,
    the reference's default, needs it): only the (tile, palette) pairs the queries name get a feature row, through a sorted pair list"""
    from tiler_amd import stages
    monkeypatch.setenv("TM_EPU_TABLE_GIB", "0")
    tiles, flags = tiles_flags
    rng = np.random.default_rng(78)
    nt, npal = 400, 37
    palettes = rng.integers(0, 1 << 24, size=(npal, 16), dtype=np.int32)
    pal_px = rng.integers(0, 16, size=(nt, 64), dtype=np.uint8)
    pal_px[41] = pal_px[5]
    tile_pal = rng.integers(0, npal, size=nt, dtype=np.int32)
    db = oracle.features_pal(pal_px, tile_pal, palettes, 1)
    q = oracle.features_rgb(tiles[:250], None, 1, False)
    idx64, _ = oracle.knnk(q, db, 64)
    idx64[7, 10:] = -1
    idx64[8, :] = idx64[8, 0]
    idx64[9, :] = -1                             # nothing at all: TileIdx / PalIdx -1, error $FFFFFFFF
    et, ep, ee = oracle.epu_rerank(q, idx64, pal_px, tile_pal, palettes)
    t, p, e = stages.epu_rerank(_dev(q), _dev(idx64), _dev(pal_px), _dev(tile_pal), _dev(palettes))
    torch.cuda.synchronize()
    assert np.array_equal(t.cpu().numpy(), et) and np.array_equal(p.cpu().numpy(), ep)
    assert np.array_equal(_host_u32(e), ee)


def test_knn_topk_matches_brute_force_at_scale(topk_thresholds):
    """the pruned MFMA collection scan against the exact VALU brute force (TM_TOPK_BRUTE=1) on clustered data with many
    duplicates and near-duplicates: overflow re-scans and the (distance, index) order at the 64th place get exercised"""
    import os
    from tiler_amd import stages
    rng = np.random.default_rng(123)
    centres = _rand_features(rng, 300, 400)
    db = (centres[rng.integers(0, 300, 60000)].astype(np.int32) + rng.integers(-3, 4, (60000, 192))).astype(np.int16)
    db[1000:1900] = db[999]          # 900 identical rows: more ties than any candidate list holds
    q = (centres[rng.integers(0, 300, 5000)].astype(np.int32) + rng.integers(-3, 4, (5000, 192))).astype(np.int16)
    q[:50] = db[999]
    idx, err = stages.knn_topk(_dev(q), _dev(db), 64)
    os.environ["TM_TOPK_BRUTE"] = "1"
    try:
        bidx, berr = stages.knn_topk(_dev(q), _dev(db), 64)
    finally:
        del os.environ["TM_TOPK_BRUTE"]
    torch.cuda.synchronize()
    assert torch.equal(err, berr) and torch.equal(idx, bidx)
    assert int(idx[0, 0]) == 999 and int(err[0, 0]) == 0


def _kmodes_rows(case, rng):
    if case == "clusters":  # six well separated prototypes + per-byte noise: the usual shape
        proto = rng.integers(0, 32, size=(6, 80))
        rows = proto[rng.integers(0, 6, size=6000)].copy()
        noise = rng.random(rows.shape) < 0.15
        rows[noise] = rng.integers(0, 32, size=int(noise.sum()))
        return rows.astype(np.uint8), 8, 32
    if case == "uniform":   # no structure: many moves, long runs, modes change in almost every bin
        return rng.integers(0, 256, size=(3000, 80)).astype(np.uint8), 12, 256
    if case == "identical":  # every row the same: k - 1 clusters start empty -> RandInt modes, empty-cluster repairs
        return np.tile(rng.integers(0, 8, size=(1, 80)), (2500, 1)).astype(np.uint8), 5, 8
    if case == "few-points":  # fewer points than clusters: the farthest-first pick runs out of unused points
        return rng.integers(0, 4, size=(3, 80)).astype(np.uint8), 6, 4
    raise ValueError(case)


@pytest.mark.parametrize("path", ["fast-leg", "fast-leg-always", "binwise"])
@pytest.mark.parametrize("num_init", [0, -7, 3])
@pytest.mark.parametrize("case", ["clusters", "uniform", "identical", "few-points"])
def test_kmodes(oracle, case, num_init, path, monkeypatch):
    """A17: TKModes.ComputeKModes (kmodes.pas:923-1094) -- labels, modes, cost and the best run's iteration count equal the oracle's
    restatement on structured, structureless and degenerate inputs, from point 0, from point 7 and over three spread starting points;
    with the later iterations' fast leg (all remaining points scored at once, the bins walked by one launch until a mode changes: it
    runs to the end on "clusters", stops at once on "uniform" and hands over to the bin-by-bin launches in the middle of the iteration when
    TM_KMODES_FAST_ALWAYS has it tried there, meets the empty-cluster repairs on "identical") and with every iteration bin by bin
    (TM_KMODES_BINWISE)"""
    from tiler_amd import stages
    monkeypatch.delenv("TM_KMODES_BINWISE", raising=False)
    monkeypatch.delenv("TM_KMODES_FAST_ALWAYS", raising=False)
    if path == "binwise":
        monkeypatch.setenv("TM_KMODES_BINWISE", "1")
    if path == "fast-leg-always":  # (by default the leg is not tried behind an iteration that moved more than 16 points a bin)
        monkeypatch.setenv("TM_KMODES_FAST_ALWAYS", "1")
    rng = np.random.default_rng(len(case) * 10 + 3)
    rows, k, nmod = _kmodes_rows(case, rng)
    if num_init < 0 and rows.shape[0] <= -num_init:
        pytest.skip("starting point outside the data")
    exp_labels, exp_cent, exp_cost, exp_iters = oracle.kmodes(rows, k, num_init, nmod, 60)
    labels, cent, cost, iters = stages.kmodes(rows, k, num_init, nmod, 60)
    assert cost == exp_cost and iters == exp_iters
    assert np.array_equal(cent, exp_cent)
    assert np.array_equal(labels, exp_labels)


def test_kmodes_on_device_pointers_at_the_4k_clip_shape(oracle):
    """config 5's shape: a million rows of 80 bytes, 64 clusters, on device pointers -- labels, modes and cost equal
    the oracle's after a farthest-first initialisation from point 0 and three iterations of KModesIter (the oracle's CPU time bounds the
    iteration count here, not the GPU's); the data have structure (prototypes + noise) so that early bins move most of their points,
    modes change in almost every bin and the late ones settle"""
    from tiler_amd import stages
    rng = np.random.default_rng(5)
    n, k = 1_000_000, 64
    proto = rng.integers(0, 48, size=(40, 80))
    rows = proto[rng.integers(0, 40, size=n)].copy()
    noise = rng.random(rows.shape) < 0.2
    rows[noise] = rng.integers(0, 48, size=int(noise.sum()))
    rows = rows.astype(np.uint8)
    exp_labels, exp_cent, exp_cost, exp_iters = oracle.kmodes(rows, k, 0, 48, 3)
    import os
    drows = torch.from_numpy(rows).cuda()
    for always in ("0", "1"):  # the later iterations' fast leg as shipped, and tried in the second iteration too (stops, then the hand-over)
        os.environ["TM_KMODES_FAST_ALWAYS"] = always
        try:
            labels, cent, cost, iters, point_iters = stages.kmodes_dev(drows, k, 0, 48, 3)
        finally:
            del os.environ["TM_KMODES_FAST_ALWAYS"]
        assert cost == exp_cost and iters == exp_iters and point_iters == 3 * n
        assert np.array_equal(cent.cpu().numpy(), exp_cent)
        assert np.array_equal(labels.cpu().numpy(), exp_labels)


def _dl3_image(case, rng):
    if case == "photo":      # smooth gradients + noise: a few thousand occupied cells, sums far from wrapping
        y, x = np.mgrid[0:240, 0:320]
        img = np.stack([(x * 255 // 320 + rng.integers(-20, 21, x.shape)) % 256, (y * 255 // 240 + rng.integers(-20, 21, x.shape)) % 256,
                        ((x + y) * 255 // 560 + rng.integers(-9, 10, x.shape)) % 256], axis=-1)
        return img.reshape(-1, 3).astype(np.uint8)
    if case == "uniform":    # every cell occupied, counts nearly equal: many near-ties in the error
        return rng.integers(0, 256, size=(150000, 3)).astype(np.uint8)
    if case == "flat":       # a handful of colours with equal counts: exact ties -> the first index must win everywhere
        base = np.array([[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [128, 128, 128], [64, 64, 64], [192, 192, 192],
                         [255, 255, 0], [0, 255, 255]], np.uint8)
        return np.repeat(base, 1000, axis=0)
    if case == "one":        # fewer colours than asked for: nothing to merge
        return np.tile(np.array([[17, 99, 230]], np.uint8), (500, 1))
    raise ValueError(case)


@pytest.mark.parametrize("case,quant_to,bpc", [("photo", 16, 4), ("photo", 64, 5), ("uniform", 16, 3), ("uniform", 256, 4), ("flat", 4, 4), ("flat", 1, 2),
                                               ("one", 16, 4), ("photo", 2, 1)])
def test_dl3quant(oracle, case, quant_to, bpc):
    """A17: dl3quant (dlquant/quantizer.c:437-455) -- the palette and the number of colours left equal the oracle's restatement of
    build_table3 / reduce_table3 / set_palette3 (parity unpinned: the reference holds no output of it), on smooth, structureless,
    exactly tied and degenerate inputs; through the stage seam and through the import's own signature (extern.pas:196)"""
    import ctypes
    from tiler_amd import stages, lib
    rng = np.random.default_rng(quant_to * 7 + bpc)
    img = _dl3_image(case, rng)
    exp_pal, exp_n = oracle.dl3quant(img, quant_to, bpc)
    pal, n = stages.dl3quant(torch.from_numpy(img).cuda(), quant_to, bpc)
    assert n == exp_n and np.array_equal(pal.cpu().numpy(), exp_pal)
    # the fine seam: host pointers, userpal[3][65536]
    L = lib()
    L.dl3quant.restype = ctypes.c_int
    L.dl3quant.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    userpal = np.zeros((3, 65536), np.uint8)
    h = 100 if img.shape[0] % 100 == 0 else 1
    assert L.dl3quant(img.ctypes.data, img.shape[0] // h, h, quant_to, bpc, userpal.ctypes.data) == 0
    assert np.array_equal(userpal[:, :exp_n], exp_pal[:, :exp_n])


def test_fine_seam_ann_double():
    """ANN.dll's own entry points (extern.pas:178-180) as DoPalettization calls them (tilingencoder.pas:4128, 4183-4187): an array of row
    pointers to double[192] centroids, one double query per call, the squared distance back through *err.  Checked against numpy's
    float64 arithmetic in the same summation order; ties -> lowest index."""
    import ctypes
    from tiler_amd import lib
    L = lib()
    rng = np.random.default_rng(21)
    n, dd = 700, 192
    cent = rng.normal(0, 300, size=(n, dd))
    cent[500] = cent[20]  # an exact duplicate: the lower index must win
    q = np.concatenate([cent[[20, 3]] + rng.normal(0, 1e-3, size=(2, dd)), rng.normal(0, 300, size=(30, dd)), cent[[20]]])
    rows = (ctypes.c_void_p * n)(*[cent[i].ctypes.data for i in range(n)])
    L.ann_kdtree_create.restype = ctypes.c_void_p
    L.ann_kdtree_create.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    tree = L.ann_kdtree_create(rows, n, dd, 32, 0)
    assert tree
    L.ann_kdtree_search.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.POINTER(ctypes.c_double)]
    L.ann_kdtree_search.restype = ctypes.c_int

    def ref(x):
        d = np.zeros(n)
        for j in range(dd):  # one subtraction, one multiplication, one addition per dimension, in order
            t = x[j] - cent[:, j]
            d = d + t * t
        i = int(np.argmin(d))  # first minimum = lowest index
        return i, d[i]
    for i in range(q.shape[0]):
        err = ctypes.c_double()
        got = L.ann_kdtree_search(tree, q[i].ctypes.data, 0.0, ctypes.byref(err))
        ei, ed = ref(q[i])
        assert got == ei and err.value == ed
    idx = np.zeros(q.shape[0], np.int32)
    errs = np.zeros(q.shape[0], np.float64)
    L.ann_kdtree_search_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    assert L.ann_kdtree_search_batch(tree, q.ctypes.data, q.shape[0], idx.ctypes.data, errs.ctypes.data) == 0
    assert idx[0] == 20 and idx[-1] == 20 and errs[-1] == 0.0 and idx[1] == 3
    L.ann_kdtree_destroy.argtypes = [ctypes.c_void_p]
    L.ann_kdtree_destroy(tree)
