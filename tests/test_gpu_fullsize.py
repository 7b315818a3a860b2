"""Parity at BASELINE.json's full sizes through size-independent properties (the oracle cannot finish these sizes):
configs[1]/[2] 1280x720x300 P=16, configs[3] 1920x1080x1000 P=16 (single GPU: the whole corpus on one card), configs[4]
3840x2160x600 P=64.  Every check is either an exact independent recomputation on the GPU in torch (fp64 matmul is
exact for these integer ranges) or the CPU oracle on a random sample of tiles.

The clip is synthesised on the device (gradients + tile noise + scene cuts like tiler_amd/synth.py, but from torch's
generator: these tests do not need the PCG64 stream)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

SIZES = {
    "720p300": (1280, 720, 300, 16),
    "1080p1000": (1920, 1080, 1000, 16),
    "4k600": (3840, 2160, 600, 64),
}


def device_video(w, h, nf, seed=1234, noise=8, rho=0.25, cut=100, freeze=True):
    """freeze=True: every third tile column keeps its frame-0 gradient (exact inter-frame duplicates, the clip of rounds 1-3);
    freeze=False: SURVEY.md 8(d)'s generator as written -- the clip bench.py quotes `value` on (no duplicate frame tiles)"""
    g = torch.Generator(device="cuda").manual_seed(seed)
    y, x = torch.meshgrid(torch.arange(h, device="cuda"), torch.arange(w, device="cuda"), indexing="ij")
    frozen = ((x >> 3) % 3 == 0) if freeze else torch.zeros_like(x, dtype=torch.bool)
    out = torch.empty((nf, h, w), dtype=torch.int32, device="cuda")
    th, tw = (h + 7) // 8, (w + 7) // 8
    for f in range(nf):
        drift = torch.where(frozen, 0, f)
        r = (x * 255 // w + 2 * drift) % 256
        gch = (y * 255 // h + drift) % 256
        b = ((x + y) * 255 // (w + h) + 3 * drift) % 256
        noisy = (torch.rand((th, tw), generator=g, device="cuda") < rho).repeat_interleave(8, 0).repeat_interleave(8, 1)[:h, :w]
        n = torch.randint(-noise, noise + 1, (3, h, w), generator=g, device="cuda")
        r = (r + n[0] * noisy).clamp(0, 255)
        gch = (gch + n[1] * noisy).clamp(0, 255)
        b = (b + n[2] * noisy).clamp(0, 255)
        rot = (f // cut) % 3
        if rot == 1:
            r, gch, b = gch, b, r
        elif rot == 2:
            r, gch, b = b, r, gch
        out[f] = ((0xFF << 24) | (r << 16) | (gch << 8) | b).to(torch.int64).to(torch.int32)
    return out


def exact_nn(qf, db, block=256):
    """lowest-index exact nearest row by SSD, in fp64 (exact: |v| < 2^15, 192 terms)"""
    dbd = db.to(torch.float64)
    dn = (dbd * dbd).sum(1)
    idx = torch.empty(qf.shape[0], dtype=torch.int64, device=qf.device)
    err = torch.empty(qf.shape[0], dtype=torch.float64, device=qf.device)
    for s in range(0, qf.shape[0], block):
        q = qf[s:s + block].to(torch.float64)
        d = (q * q).sum(1)[:, None] + dn[None, :] - 2.0 * (q @ dbd.T)
        m = d.min(1).values
        first = (d == m[:, None]).to(torch.uint8).argmax(1)  # lowest index among the minima
        idx[s:s + block] = first
        err[s:s + block] = m
    return idx, err


@pytest.mark.parametrize("name", ["720p300", "720p300-literal", "1080p1000", "4k600"])
def test_full_size_properties(oracle, name, tmp_path):
    from tiler_amd import stages
    from tiler_amd.encoder import TilingEncoder, TEncoderStep as S
    literal = name.endswith("-literal")  # the generator as SURVEY.md 8(d) writes it: what bench.py's `value` is timed on
    name = name.split("-")[0]
    w, h, nf, npal = SIZES[name]
    free, _ = torch.cuda.mem_get_info()
    need = nf * h * w * 4 * 4.5
    if free < need:
        pytest.skip(f"needs ~{need / 2**30:.0f} GiB of HBM")
    frames = device_video(w, h, nf, freeze=not literal)
    tm_w, tm_h = w // 8, h // 8
    per, q = tm_w * tm_h, nf * tm_w * tm_h
    enc = TilingEncoder()
    enc.LoadDefaultSettings()
    enc.PaletteCount = npal
    enc.FrameTilingExtendedPaletteUsage = False
    enc.MotionPredictRadius = 0  # the KNN-only pipeline; motion prediction has its own full-size test below
    enc.SetVideo(w, h, 24.0, nf)
    enc.SetFramesDevice(frames)
    gen = torch.Generator(device="cuda").manual_seed(99)

    # ---- Load: canonical tiles un-mirror to the frame's own 8x8 blocks; keyframes obey the 1 s / 15 s rule
    enc.Run(S.esLoad)
    tiles_f, flags_f, _ = stages.load(frames[:2], tm_w, tm_h)  # stage seam on two frames = the encoder's frame tiles
    blk = frames[:2].reshape(2, tm_h, 8, tm_w, 8).permute(0, 1, 3, 2, 4).reshape(-1, 8, 8)
    sw = ((blk & 0xFF) << 16) | (blk & 0xFF00) | ((blk >> 16) & 0xFF)  # SwapRB, alpha dropped
    t = tiles_f.reshape(-1, 8, 8)
    t = torch.where((flags_f & 1).bool()[:, None, None], t.flip(2), t)
    t = torch.where((flags_f & 2).bool()[:, None, None], t.flip(1), t)
    assert torch.equal(t, sw)
    kf = enc.KeyFrames()
    assert kf[0] == 0 and np.all(np.diff(kf) > 0) and np.all(np.diff(np.r_[kf, nf]) <= 15 * 24 + 1)
    assert set(kf) >= set(range(0, nf, 100))  # every scene cut of the clip is a keyframe

    # ---- Reduce: global tiles are distinct, budgeted, sorted by use; tile map points at byte-identical tiles
    enc.Run(S.esPredictMotion)
    enc.Run(S.esReduce)
    c = enc.counts()
    T = c["tiles"]
    assert T == min(int(enc.GlobalTilingTileCount), T) and T > 0
    hdr, _, rgb = enc.Tiles()
    use = hdr["UseCount"].astype(np.int64)
    assert np.all(np.diff(use) <= 0) and use.min() >= 1
    rgb_d = torch.from_numpy(rgb.view(np.int32)).cuda()
    assert torch.unique(rgb_d, dim=0).shape[0] == T
    for f in (0, nf // 2, nf - 1):
        tmap = enc.TileMap(f)
        ti = torch.from_numpy(tmap["TileIdx"].astype(np.int64)).cuda()
        ft, _, _ = stages.load(frames[f:f + 1], tm_w, tm_h)
        ok = ti >= 0
        assert torch.equal(rgb_d[ti[ok]], ft[ok].reshape(-1, 64))
    del rgb_d

    # ---- PreparePalettes + Dither: indices inside the palette, palettes ordered by (Val, Sat, Hue) live colours first;
    #      a random sample of tiles against the oracle's Thomas-Knoll
    enc.Run(S.esPreparePalettes)
    enc.Run(S.esDither)
    pals = enc.Palettes()
    assert pals.shape == (npal, 16)
    hdr, pal_px, rgb = enc.Tiles()
    pal_idx = hdr["PalIdx_Initial"]
    assert pal_idx.min() >= 0 and pal_idx.max() < npal and pal_px.max() < 16
    counts = np.bincount(pal_idx, minlength=npal)
    assert np.all(np.diff(counts) <= 0)  # palettes ranked by tile count (4229-4234)
    sample = torch.randint(0, T, (192,), generator=gen, device="cuda").cpu().numpy()
    gflags = ((hdr["Flags"] >> 3) & 3).astype(np.uint8)
    want = oracle.dither(rgb[sample], gflags[sample], pal_idx[sample], pals, True)
    assert np.array_equal(pal_px[sample], want)
    live = pals[pal_idx[sample][:, None], pal_px[sample]]
    assert np.all(live != -65281)  # never a null colour

    # ---- Reconstruct: a random sample of queries against an exact fp64 scan of the whole database
    enc.Run(S.esReconstruct)
    db = stages.features_pal(torch.from_numpy(pal_px).cuda(), torch.from_numpy(pal_idx.astype(np.int32)).cuda(),
                             torch.from_numpy(pals).cuda(), 1)
    nq_s = 4096
    for f in torch.randint(0, nf, (3,), generator=gen, device="cuda").tolist():
        ft, ffl, _ = stages.load(frames[f:f + 1], tm_w, tm_h)
        qf = stages.features_rgb(ft, None, 1, False)
        pick = torch.randperm(per, generator=gen, device="cuda")[:nq_s]
        idx, err = exact_nn(qf[pick], db)
        tmap = enc.TileMap(f)
        got_idx = torch.from_numpy(tmap["TileIdx"].astype(np.int64)).cuda()[pick]
        assert torch.equal(got_idx, idx)
        psnr = np.array([oracle.L.tmo_euclidean_to_psnr(int(e)) for e in err.cpu().numpy()[:64]], np.float32)
        assert np.allclose(tmap["PSNR"][pick.cpu().numpy()[:64]], psnr, rtol=1e-6)
        assert np.array_equal(tmap["PalIdx"], pal_idx[tmap["TileIdx"]])
    ks = enc.KnnStats()
    assert 0 < ks["pairs"] <= q * ks["db_rows"] and ks["db_rows"] <= T
    del db

    # ---- Reindex: use counts are the tile-map histogram, order is (use desc, content asc), content is distinct
    pre_px = pal_px
    pre_map = [enc.TileMap(f)["TileIdx"] for f in (0, nf - 1)]
    enc.Run(S.esReindex)
    hdr2, px2, _ = enc.Tiles()
    T2 = px2.shape[0]
    assert hdr2["UseCount"].astype(np.int64).sum() == q and T2 <= T
    u2 = hdr2["UseCount"].astype(np.int64)
    assert np.all(np.diff(u2) <= 0)
    px2_d = torch.from_numpy(px2).cuda()
    assert torch.unique(px2_d, dim=0).shape[0] == T2
    key = torch.from_numpy(px2.copy()).cuda().to(torch.int64)
    same_use = torch.from_numpy((np.diff(u2) == 0)).cuda()
    diff = key[1:] - key[:-1]
    first_nz = (diff != 0).to(torch.uint8).argmax(1)
    asc = diff.gather(1, first_nz[:, None])[:, 0] > 0
    assert bool((asc | ~same_use).all())  # equal use counts: ascending byte content (CompareTileUseCountRev, 584-599)
    for fi, f in enumerate((0, nf - 1)):
        t2 = enc.TileMap(f)["TileIdx"]
        assert np.array_equal(px2[t2], pre_px[pre_map[fi]])  # remap preserves what every position shows

    # ---- Save (720p only: the Python player is slow): the file plays back the first frames the tables describe
    if name == "720p300":
        from tests import gtm_reader
        path = str(tmp_path / "full.gtm")
        enc.Save(path)
        data = open(path, "rb").read()
        hdr_g, raws = gtm_reader.unpack(oracle, data)
        assert hdr_g["frame_count"] == nf and hdr_g["kf_count"] == len(kf) and hdr_g["width"] == w
        pl = gtm_reader.Player(max_frames=2)
        pl.feed(raws[0])
        tms = np.stack([enc.TileMap(0), enc.TileMap(1)])
        want = gtm_reader.render_expected(px2, pals, tms, tm_w, tm_h)
        assert np.array_equal(np.stack(pl.frames), want)
    enc.close()


@pytest.mark.parametrize("n,k,sep", [(320705, 16, 0), (524288, 16, 120), (262145, 7, 300)])
def test_kmeans_resident_launch_equals_the_launches_per_iteration(n, k, sep, monkeypatch):
    """the tile -> palette clustering at the bench clip's size (320 705 points: 256 resident workgroups, two rounds of points each, one barrier
    per iteration) on loosely clustered points that keep moving for many iterations: k_h_resident against the three launches per iteration
    (TM_KM_LAUNCHES=1) -- assignments, centroids bit for bit, iteration count"""
    from tiler_amd import stages
    g = torch.Generator(device="cuda").manual_seed(n + k)
    centres = torch.randint(-sep, sep + 1, (k * 3, 192), generator=g, device="cuda", dtype=torch.int32)  # (sep 0: structureless noise, the slowest to settle)
    pts = centres[torch.randint(0, k * 3, (n,), generator=g, device="cuda")] + torch.randint(-1000, 1001, (n, 192), generator=g, device="cuda", dtype=torch.int32)
    w = torch.randint(1, 9, (n,), generator=g, device="cuda", dtype=torch.int32)
    monkeypatch.delenv("TM_KM_LAUNCHES", raising=False)
    kk, assign, cent, iters = stages.kmeans(pts, w, k, 300)
    monkeypatch.setenv("TM_KM_LAUNCHES", "1")
    kk2, assign2, cent2, iters2 = stages.kmeans(pts, w, k, 300)
    assert (kk, iters) == (kk2, iters2) and iters > 20
    assert torch.equal(assign, assign2)
    assert torch.equal(cent.view(torch.int64), cent2.view(torch.int64))


@pytest.mark.parametrize("n,d,k", [(320705, 192, 16), (994105, 192, 16), (1618022, 192, 64), (20_000_000, 3, 16)])
def test_kmeans_fixed_point_at_full_size(n, d, k):
    """Lloyd's fixed point at the sizes of configs[3]/[4]: every point sits with its nearest centroid (lowest index on
    ties, IEEE double in dimension order) and every centroid is the exact weighted mean of its points"""
    from tiler_amd import stages
    g = torch.Generator(device="cuda").manual_seed(5)
    centres = torch.randint(-400, 400, (k * 2, d), generator=g, device="cuda", dtype=torch.int32)
    pts = centres[torch.randint(0, k * 2, (n,), generator=g, device="cuda")] + torch.randint(-60, 60, (n, d), generator=g, device="cuda",
                                                                                                dtype=torch.int32)
    if d == 3:
        pts = pts.clamp(0, 255)
    w = torch.randint(1, 5, (n,), generator=g, device="cuda", dtype=torch.int32)
    kk, assign, cent, iters = stages.kmeans(pts, w, k, 300)
    assert kk == k and 1 <= iters <= 300
    a64 = assign.to(torch.int64)
    wsum = torch.zeros(k, dtype=torch.float64, device="cuda").index_add_(0, a64, w.to(torch.float64))
    sums = torch.zeros((k, d), dtype=torch.float64, device="cuda")
    for s in range(0, n, 1 << 20):
        sums.index_add_(0, a64[s:s + (1 << 20)], pts[s:s + (1 << 20)].to(torch.float64) * w[s:s + (1 << 20), None].to(torch.float64))
    if iters < 300:  # converged: centroids are the exact means of the final assignment, which is a fixed point of the distance rule
        assert torch.equal(cent, sums / wsum[:, None])  # integer sums < 2^53: fp64 accumulation is exact
        for s in range(0, n, 1 << 18):
            p = pts[s:s + (1 << 18)].to(torch.float64)
            dist = torch.zeros((p.shape[0], k), dtype=torch.float64, device="cuda")
            for j in range(d):  # dimension order matters for the rounding of the running sum; the build's rule fuses multiply and add
                diff = p[:, j:j + 1] - cent[None, :, j]
                dist = torch.addcmul(dist, diff, diff)
            best = dist.min(1).values
            first = (dist == best[:, None]).to(torch.uint8).argmax(1)
            assert torch.equal(first, a64[s:s + (1 << 18)])


def test_motion_prediction_at_720p(oracle, tmp_path):
    """configs[1] with the reference's default MotionPredictRadius = 32: PredictMotion against the oracle on crops (the
    search is position-relative: a tile whose whole +-32 window lies inside a crop gets the same answer from the crop),
    the tile budget the threshold search lands on, and the .gtm chain played back"""
    from tiler_amd import stages
    from tiler_amd.encoder import TilingEncoder, TEncoderStep as S
    from tests import gtm_reader
    w, h, nf, npal = SIZES["720p300"]
    frames = device_video(w, h, nf)
    tm_w, tm_h = w // 8, h // 8
    per = tm_w * tm_h
    enc = TilingEncoder()
    enc.LoadDefaultSettings()
    enc.PaletteCount = npal
    enc.FrameTilingExtendedPaletteUsage = False
    assert enc.MotionPredictRadius == 32
    enc.SetVideo(w, h, 24.0, nf)
    enc.SetFramesDevice(frames)
    enc.Run(S.esLoad)
    enc.Run(S.esPredictMotion)
    ct = 17  # crop of 17 x 17 tiles: tiles 5..11 keep their full window (33 px up/left, 32 + 7 px down/right)
    for f, ty0, tx0 in [(0, 0, 0), (1, 20, 60), (150, 73, 143), (299, 40, 3)]:
        src = f - 1 if f >= 1 else 1
        crop = lambda k: np.ascontiguousarray(frames[k, ty0 * 8:(ty0 + ct) * 8, tx0 * 8:(tx0 + ct) * 8].cpu().numpy().view(np.uint32))
        prev_t = oracle.load_from_image(crop(src), ct, ct)
        cur_t = oracle.load_from_image(crop(f), ct, ct)
        fb = np.ascontiguousarray(prev_t.reshape(ct, ct, 8, 8).transpose(0, 2, 1, 3).reshape(ct * 8, ct * 8))
        err, px, py = oracle.motion_search(oracle.features_rgb(cur_t, None, 1, False), ct, ct, oracle.window_dcts(fb), 32)
        tmap = enc.TileMap(f).reshape(tm_h, tm_w)[ty0:ty0 + ct, tx0:tx0 + ct]
        # interior of the crop, unless the crop touches the frame border (there the clamped window is the same on both sides)
        lo_y, hi_y = (0 if ty0 == 0 else 5), (ct if ty0 + ct == tm_h else 12)
        lo_x, hi_x = (0 if tx0 == 0 else 5), (ct if tx0 + ct == tm_w else 12)
        sel = np.zeros((ct, ct), bool)
        sel[lo_y:hi_y, lo_x:hi_x] = True
        assert np.array_equal(tmap["PredictedX"][sel], px.reshape(ct, ct)[sel])
        assert np.array_equal(tmap["PredictedY"][sel], py.reshape(ct, ct)[sel])
        assert np.allclose(tmap["PSNR"][sel], oracle.psnr(err).reshape(ct, ct)[sel], rtol=1e-6)
    for st in (S.esReduce, S.esPreparePalettes, S.esDither, S.esReconstruct):
        enc.Run(st)
    target = int(enc.GlobalTilingTileCount)
    T = enc.counts()["tiles"]
    assert 0.9 * target <= T <= 1.1 * target, (T, target)  # the search stops within half a tile of the target unless PSNR ties block it
    kf = enc.KeyFrames()
    t0 = enc.TileMap(int(kf[1]))
    assert not ((t0["Flags"] >> 2) & 1).any()          # a key frame's first frame has no motion candidate (1496)
    t1 = enc.TileMap(int(kf[1]) + 1)
    pred = ((t1["Flags"] >> 2) & 1).astype(bool)
    assert 0.05 < pred.mean() < 0.99
    assert np.all(np.abs(t1["PredictedX"].astype(int)) <= 32) and np.all(np.abs(t1["PredictedY"].astype(int)) <= 32)
    enc.Run(S.esReindex)
    path = str(tmp_path / "mp.gtm")
    enc.Save(path)
    hdr, raws = gtm_reader.unpack(oracle, open(path, "rb").read())
    pl = gtm_reader.Player(max_frames=3)
    pl.feed(raws[0])
    kinds = {it[0] for fr in pl.items for it in fr}
    assert "ps" in kinds or "skip" in kinds
    hdr_t, px2, _ = enc.Tiles()
    want0 = gtm_reader.render_expected(px2, enc.Palettes(), np.stack([enc.TileMap(0)]), tm_w, tm_h)
    assert np.array_equal(pl.frames[0], want0[0])  # frame 0 is all KNN
    enc.close()


def test_extended_palette_usage_at_720p(oracle):
    """configs[1] with FrameTilingExtendedPaletteUsage on (the reference default): for sampled queries the 64 nearest rows come from
    an exact fp64 scan of the whole database in torch, ordered by (distance, index); the oracle's re-rank of that list must be what
    the encoder chose (tile and palette), through the distinct-row scan, duplicate expansion and threshold re-scans"""
    from tiler_amd import stages
    from tiler_amd.encoder import TilingEncoder, TEncoderStep as S
    w, h, nf, npal = SIZES["720p300"]
    frames = device_video(w, h, nf)
    tm_w, tm_h = w // 8, h // 8
    per = tm_w * tm_h
    enc = TilingEncoder()
    enc.LoadDefaultSettings()
    enc.PaletteCount = npal
    enc.MotionPredictRadius = 0
    assert enc.FrameTilingExtendedPaletteUsage
    enc.SetVideo(w, h, 24.0, nf)
    enc.SetFramesDevice(frames)
    for st in (S.esLoad, S.esPredictMotion, S.esReduce, S.esPreparePalettes, S.esDither, S.esReconstruct):
        enc.Run(st)
    pals = enc.Palettes()
    hdr, pal_px, _ = enc.Tiles()
    pal_idx = hdr["PalIdx_Initial"].astype(np.int32)
    db = stages.features_pal(torch.from_numpy(pal_px).cuda(), torch.from_numpy(pal_idx).cuda(), torch.from_numpy(pals).cuda(), 1)
    dbd = db.to(torch.float64)
    dn = (dbd * dbd).sum(1)
    gen = torch.Generator(device="cuda").manual_seed(7)
    moved = 0
    for f in torch.randint(0, nf, (3,), generator=gen, device="cuda").tolist():
        ft, _, _ = stages.load(frames[f:f + 1], tm_w, tm_h)
        qf = stages.features_rgb(ft, None, 1, False)
        pick = torch.randperm(per, generator=gen, device="cuda")[:256]
        q = qf[pick].to(torch.float64)
        d = (q * q).sum(1)[:, None] + dn[None, :] - 2.0 * (q @ dbd.T)  # exact: integers below 2^53
        key = d * float(1 << 22) + torch.arange(db.shape[0], device="cuda", dtype=torch.float64)[None, :]  # (distance, index): d < 2^31, index < 2^22
        idx64 = torch.topk(key, 64, dim=1, largest=False, sorted=True).indices.to(torch.int32).cpu().numpy()
        et, ep, ee = oracle.epu_rerank(qf[pick].cpu().numpy(), idx64, pal_px, pal_idx, pals)
        tmap = enc.TileMap(f)
        sel = pick.cpu().numpy()
        assert np.array_equal(tmap["TileIdx"][sel], et) and np.array_equal(tmap["PalIdx"][sel], ep)
        assert np.allclose(tmap["PSNR"][sel], oracle.psnr(ee), rtol=1e-6)
        moved += int((ep != pal_idx[et]).sum())
    assert moved > 0
    enc.close()
