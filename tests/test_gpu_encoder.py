"""End-to-end: TTilingEncoder.Run(esAll) through the coarse C ABI against the oracle pipeline (BASELINE config 1:
64x64, 10 frames, 8x8 tiles, 1 palette; plus a multi-palette, non-multiple-of-8 case)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _run_encoder(frames, **settings):
    from tiler_amd.encoder import TilingEncoder
    enc = TilingEncoder()
    enc.LoadDefaultSettings()
    for k, v in settings.items():
        setattr(enc, k, v)
    nf, h, w = frames.shape
    enc.SetVideo(w, h, 24.0, nf)
    for f in range(nf):
        enc.PushFrame(f, frames[f])
    enc.Run()
    return enc


@pytest.mark.parametrize("shape,pc,radius,tc,epu", [((10, 64, 64), 1, 0, 0, False), ((6, 52, 100), 3, 0, 0, False), ((10, 64, 64), 1, 32, 0, False),
                                                      ((10, 64, 64), 2, 32, 150, False), ((7, 52, 100), 3, 5, 300, False),
                                                      ((1, 32, 32), 1, 32, 0, False), ((6, 52, 100), 3, 0, 0, True),
                                                      ((10, 64, 64), 4, 32, 150, True), ((3, 24, 24), 2, 32, 0, True)])
def test_run_all_matches_oracle(oracle, shape, pc, radius, tc, epu):
    """radius 0 = motion prediction off (the build's switch); radius > 0 = the reference's default path: PredictMotion,
    the PSNR threshold search of Reduce (tc = a GlobalTilingTileCount small enough to make the search bite), the motion
    redo and KNN-vs-motion decision of Reconstruct; epu = FrameTilingExtendedPaletteUsage (k = 64 x palettes re-rank, the
    (3, 24, 24) case has fewer than 64 global tiles: the padded list)"""
    from tiler_amd import synth
    from tests import oracle_pipeline
    frames = synth.video(*shape[:1], shape[2], shape[1], cut=4)
    exp = oracle_pipeline.run(oracle, frames, palette_count=pc, min_s=0.1, motion_radius=radius, tile_count=tc, epu=epu)
    kw = dict(GlobalTilingTileCount=tc) if tc else {}
    enc = _run_encoder(frames, PaletteCount=pc, ShotTransMinSecondsPerKF=0.1, MotionPredictRadius=radius, FrameTilingExtendedPaletteUsage=epu, **kw)
    c = enc.counts()
    assert np.array_equal(enc.FrameCorrelations().view(np.uint32), exp["correl"].view(np.uint32))
    assert np.array_equal(enc.KeyFrames(), exp["keyframes"])
    assert c["tiles"] == exp["final_T"]
    hdr, pal, rgb = enc.Tiles()
    assert np.array_equal(pal, exp["final_pal_px"])
    assert np.array_equal(hdr["UseCount"], exp["final_use"])
    assert np.array_equal(hdr["PalIdx_Initial"], exp["final_pal_idx"])
    assert np.array_equal(rgb, exp["final_rgb"])
    assert np.array_equal(enc.Palettes(), exp["palettes"])
    per = exp["per"]
    for f in range(shape[0]):
        tm = enc.TileMap(f)
        sl = slice(f * per, (f + 1) * per)
        assert np.array_equal(tm["TileIdx"], exp["final_tm_tile"][sl])
        assert np.array_equal(tm["PalIdx"], exp["tm_pal"][sl])
        assert np.array_equal(tm["Flags"] & 3, exp["flags"][sl])
        assert np.array_equal((tm["Flags"] >> 2) & 1, exp["is_predicted"][sl])
        assert np.array_equal(tm["PredictedX"], exp["pred_x"][sl]) and np.array_equal(tm["PredictedY"], exp["pred_y"][sl])
        psnr = np.array([oracle.L.tmo_euclidean_to_psnr(int(e)) for e in exp["tm_err"][sl]], np.float32)
        assert np.allclose(tm["PSNR"], psnr, rtol=1e-6)  # PSNR goes through log10: tolerance 1e-6 relative
    # TKeyFrame.LogPSNR (1006-1028): mean PSNR by tile per key frame and over the clip
    q = enc.PSNR()
    allp = np.array([oracle.L.tmo_euclidean_to_psnr(int(e)) for e in exp["tm_err"]], np.float64)
    kf = list(exp["keyframes"]) + [shape[0]]
    assert np.allclose(q["per_keyframe"], [allp[a * per:b * per].mean() for a, b in zip(kf[:-1], kf[1:])], rtol=1e-6)
    assert np.isclose(q["global"], allp.mean(), rtol=1e-6)
    enc.close()


def test_reference_default_palette_count_with_extended_palette_usage(oracle):
    """LoadDefaultSettings + Run on a small clip: PaletteCount = 1024 and FrameTilingExtendedPaletteUsage = True (tilingencoder.pas:3826,
    3840).  The table of every tile under every palette is out of the question there (T x 1024 rows), so the re-rank builds the rows its
    queries name; the result must be the oracle's."""
    from tiler_amd import synth
    from tests import oracle_pipeline
    frames = synth.video(5, 64, 48, cut=3)
    exp = oracle_pipeline.run(oracle, frames, palette_count=1024, min_s=0.1, motion_radius=0, epu=True)
    import os
    os.environ["TM_EPU_TABLE_GIB"] = "0.01"  # (240 tiles x 1024 palettes would still fit a real budget)
    try:
        enc = _run_encoder(frames, ShotTransMinSecondsPerKF=0.1, MotionPredictRadius=0)  # PaletteCount / EPU: the defaults
    finally:
        del os.environ["TM_EPU_TABLE_GIB"]
    assert enc.PaletteCount == 1024 and enc.FrameTilingExtendedPaletteUsage
    assert np.array_equal(enc.Palettes(), exp["palettes"])
    per = exp["per"]
    for f in range(5):
        tm = enc.TileMap(f)
        sl = slice(f * per, (f + 1) * per)
        assert np.array_equal(tm["TileIdx"], exp["final_tm_tile"][sl]) and np.array_equal(tm["PalIdx"], exp["tm_pal"][sl])
    enc.close()


def test_steps_after_reload_without_load_fail_cleanly(oracle, tmp_path):
    """ReloadGTM brings palette indices, palettes and tile maps, not the frame tiles or the RGB pixels: steps that compute from those
    must say so (TM_E_INVAL) instead of launching kernels on buffers that do not exist"""
    from tiler_amd import synth, TileMotionError
    from tiler_amd.encoder import TilingEncoder, TEncoderStep
    frames = synth.video(4, 32, 32)
    a = str(tmp_path / "a.gtm")
    enc = _run_encoder(frames, PaletteCount=2, MotionPredictRadius=0, FrameTilingExtendedPaletteUsage=False, OutputFileName=a)
    enc.close()
    enc2 = TilingEncoder()
    enc2.LoadDefaultSettings()
    enc2.PaletteCount = 2
    enc2.SetVideo(32, 32, 24.0, 4)
    enc2.ReloadGTM(a)
    for step in (TEncoderStep.esPredictMotion, TEncoderStep.esReduce, TEncoderStep.esPreparePalettes, TEncoderStep.esDither, TEncoderStep.esReconstruct):
        with pytest.raises(TileMotionError) as ei:
            enc2.Run(step)
        assert ei.value.code == -1
    enc2.Run(TEncoderStep.esReindex)  # works on what the stream holds
    for f in range(4):
        enc2.PushFrame(f, frames[f])
    enc2.MotionPredictRadius = 0
    enc2.FrameTilingExtendedPaletteUsage = False
    enc2.Run()  # with the frames in, everything runs again
    enc2.close()


def test_step_order_and_errors():
    from tiler_amd.encoder import TilingEncoder, TEncoderStep
    from tiler_amd import TileMotionError
    enc = TilingEncoder()
    with pytest.raises(TileMotionError):
        enc.Run(TEncoderStep.esLoad)  # no video yet
    enc.SetVideo(16, 16, 24.0, 1)
    enc.PushFrame(0, np.zeros((16, 16), np.uint32))
    with pytest.raises(TileMotionError):
        enc.Run(TEncoderStep.esDither)  # Reduce/PreparePalettes not run
    enc.PaletteSize = 1000
    assert enc.PaletteSize == 64  # clamp of SetPaletteSize, tilingencoder.pas:2965
    enc.PaletteCount = 0
    assert enc.PaletteCount == 1
    enc.PaletteCount = 1
    enc.Run()  # a single black frame: 4 identical tiles -> 1 global tile
    assert enc.counts()["tiles"] == 1
    with pytest.raises(TileMotionError):
        enc.Run(TEncoderStep.esSave)
    enc.close()


@pytest.mark.parametrize("radius", [0, 32])
def test_save_gtm_matches_host_writer_on_oracle_tables(oracle, tmp_path, radius):
    """Run(esAll) with an OutputFileName ends in Save (tilingencoder.pas:5551): the file is byte-identical to the one the
    host writer makes from the ORACLE pipeline's tables, and the player semantics show the encoder's frames"""
    import ctypes
    import os
    from tiler_amd import synth
    from tests import gtm_reader, oracle_pipeline, test_gtm
    frames = synth.video(10, 64, 48, cut=5)
    path = str(tmp_path / "enc.gtm")
    enc = _run_encoder(frames, PaletteCount=2, ShotTransMinSecondsPerKF=0.1, OutputFileName=path, MotionPredictRadius=radius,
                       GlobalTilingTileCount=120, FrameTilingExtendedPaletteUsage=False)
    data = open(path, "rb").read()
    hdr, pl = gtm_reader.play(oracle, data)
    assert "PaletteCount=2" in pl.settings and "[Dither]" in pl.settings
    exp = oracle_pipeline.run(oracle, frames, palette_count=2, min_s=0.1, motion_radius=radius, tile_count=120)
    per = exp["per"]
    tm = np.zeros((10, per), test_gtm.TMI)
    tm["TileIdx"] = exp["final_tm_tile"].reshape(10, per)
    tm["PalIdx"] = exp["tm_pal"].reshape(10, per)
    tm["Flags"] = (exp["flags"].reshape(10, per) & 3) | (exp["is_predicted"].reshape(10, per).astype(np.uint32) << 2)
    tm["PredictedX"] = exp["pred_x"].reshape(10, per)
    tm["PredictedY"] = exp["pred_y"].reshape(10, per)
    L = ctypes.CDLL(test_gtm.LIB)
    L.tm_write_gtm_host.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_void_p,
                                    ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int,
                                    ctypes.c_int, ctypes.c_void_p, ctypes.c_char_p]
    L.tm_last_error.restype = ctypes.c_char_p
    want = test_gtm.write(L, str(tmp_path / "host.gtm"), 8, 6, 24.0, exp["keyframes"], exp["final_pal_px"], exp["final_use"],
                          exp["palettes"], tm, pl.settings)
    assert data == want
    if radius == 0:
        assert np.array_equal(np.stack(pl.frames), gtm_reader.render_expected(exp["final_pal_px"], exp["palettes"], tm, 8, 6))
    else:  # the player's last frame is the encoder's last reconstructed frame buffer (what the next frame was searched in)
        assert exp["is_predicted"].any() and "ps" in {it[0] for fr in pl.items for it in fr}
        assert np.array_equal(pl.frames[-1] & 0xFFFFFF, exp["recon_last"] & 0xFFFFFF)
    enc.Save(str(tmp_path / "again.gtm"))  # Save on its own, explicit path
    assert open(str(tmp_path / "again.gtm"), "rb").read() == data
    enc.close()


@pytest.mark.parametrize("sampled", [False, True])
def test_extended_palette_usage_scan_matches_brute_force(monkeypatch, sampled):
    """a clip large enough for duplicate-heavy databases, candidate overflows and re-scans: the pruned MFMA k-nearest scan with
    duplicate expansion gives the same encoder output as the VALU brute force over all rows (TM_TOPK_BRUTE=1); sampled: the scan's first
    thresholds from a sample of the database (TM_TOPK_ESTIMATE=1: what a full-size clip gets by itself), with member lists on the full database"""
    import os
    from tiler_amd import synth
    frames = synth.video(12, 320, 176, cut=6)
    if sampled:
        monkeypatch.setenv("TM_TOPK_ESTIMATE", "1")
    outs = []
    for brute in (False, True):
        if brute:
            os.environ["TM_TOPK_BRUTE"] = "1"
        try:
            enc = _run_encoder(frames, PaletteCount=8, ShotTransMinSecondsPerKF=0.1, MotionPredictRadius=0, FrameTilingExtendedPaletteUsage=True)
        finally:
            os.environ.pop("TM_TOPK_BRUTE", None)
        outs.append((np.stack([enc.TileMap(f) for f in range(12)]), enc.Tiles()[1]))
        enc.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    moved = outs[0][0]["PalIdx"] >= 0
    assert moved.any()


@pytest.mark.parametrize("epu", [False, True])
def test_distinct_query_groups_equal_per_item_queries(monkeypatch, epu):
    """Reconstruct searches once per DISTINCT frame tile (Reduce's exact groups) and expands the answers; with TM_NO_QUERY_GROUPS it
    searches every item: same tile maps and tiles, fewer queries"""
    from tiler_amd import synth
    frames = synth.video(12, 320, 176, cut=6)
    outs = []
    for plain in (False, True):
        if plain:
            monkeypatch.setenv("TM_NO_QUERY_GROUPS", "1")
        enc = _run_encoder(frames, PaletteCount=8, ShotTransMinSecondsPerKF=0.1, MotionPredictRadius=0, FrameTilingExtendedPaletteUsage=epu)
        outs.append((np.stack([enc.TileMap(f) for f in range(12)]), enc.Tiles()[1], enc.KnnStats()["queries"]))
        enc.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert outs[1][2] == 12 * 40 * 22
    assert 0 < outs[0][2] < outs[1][2]


def test_dither_plans_distinct_pairs_from_the_quantisation_keys(monkeypatch):
    """a clip with enough global tiles for Dither's distinct-pair path: with PreparePalettes' sorted pixel keys handed over (the
    default), with Dither marking the pairs from the pixels itself (TM_DITHER_OWN_KEYS) and with a plan per pixel
    (TM_DITHER_NO_DEDUP is read once per process, so that leg is the stage test's: test_dither_thomas_knoll_distinct_pairs_path) the
    encoder's output is the same; the pair count is reported and is at most the pixel count"""
    from tiler_amd import synth
    frames = synth.video(12, 320, 176, cut=6)
    outs = []
    for own in (False, True):
        if own:
            monkeypatch.setenv("TM_DITHER_OWN_KEYS", "1")
        enc = _run_encoder(frames, PaletteCount=8, ShotTransMinSecondsPerKF=0.1, MotionPredictRadius=0)
        outs.append((np.stack([enc.TileMap(f) for f in range(12)]), enc.Tiles()[1], enc.DitherPairs()))
        enc.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert outs[0][2] == outs[1][2]


@pytest.mark.parametrize("radius", [0, 8])
def test_reload_gtm_round_trip(oracle, tmp_path, radius):
    """Save -> ReloadGTM into a fresh encoder (LoadStream, tilingencoder.pas:4880): palettes, tile pixels, tile maps and key
    frames come back (single-use tiles renumbered in order of appearance, as LoadStream does), and saving again gives the
    same file"""
    from tiler_amd import synth
    from tiler_amd.encoder import TilingEncoder
    from tiler_amd import TileMotionError
    frames = synth.video(9, 64, 48, cut=3)
    a = str(tmp_path / "a.gtm")
    enc = _run_encoder(frames, PaletteCount=3, ShotTransMinSecondsPerKF=0.1, MotionPredictRadius=radius, FrameTilingExtendedPaletteUsage=True,
                       GlobalTilingTileCount=100, OutputFileName=a)
    per = 8 * 6
    enc2 = TilingEncoder()
    enc2.LoadDefaultSettings()
    for k in ("PaletteCount", "MotionPredictRadius", "GlobalTilingTileCount", "ShotTransMinSecondsPerKF"):
        setattr(enc2, k, getattr(enc, k))
    enc2.OutputFileName = a
    enc2.SetVideo(64, 48, 24.0, 9)
    enc2.ReloadGTM(a)
    assert np.array_equal(enc2.KeyFrames(), enc.KeyFrames())
    pal = enc.Palettes().copy()
    pal[pal == -65281] = 0xFFFFFF  # the stream stores the null colour as white (5284-5285)
    assert np.array_equal(enc2.Palettes(), pal)
    h1, px1, _ = enc.Tiles()
    h2, px2, _ = enc2.Tiles()
    assert px1.shape == px2.shape
    for f in range(9):
        t1, t2 = enc.TileMap(f), enc2.TileMap(f)
        pred = ((t1["Flags"] >> 2) & 1).astype(bool)
        assert np.array_equal(pred, ((t2["Flags"] >> 2) & 1).astype(bool))
        assert np.array_equal(t1["Flags"][~pred], t2["Flags"][~pred])  # a predicted item's mirrors are not in the stream
        assert np.array_equal(t1["PredictedX"][pred], t2["PredictedX"][pred]) and np.array_equal(t1["PredictedY"][pred], t2["PredictedY"][pred])
        ok = ~pred
        assert np.array_equal(t1["PalIdx"][ok], t2["PalIdx"][ok])
        assert np.array_equal(px1[t1["TileIdx"][ok]], px2[t2["TileIdx"][ok]])  # same pixels, possibly under another index
    b = str(tmp_path / "b.gtm")
    if radius:
        # a predicted item keeps the KNN's tile index in the encoder and Reindex counts it (2030), but the stream does not carry it:
        # reloaded use counts are lower and no longer sorted, and SaveStream's TileSet rule (5296-5305) needs them sorted -> Reindex
        from tiler_amd.encoder import TEncoderStep
        enc2.Run(TEncoderStep.esReindex)
    enc2.Save(b)
    if radius == 0:
        assert open(a, "rb").read() == open(b, "rb").read()
    else:
        from tests import gtm_reader  # some tiles are now single-use and travel as IntraTile: same pictures either way
        _, pa = gtm_reader.play(oracle, open(a, "rb").read())
        _, pb = gtm_reader.play(oracle, open(b, "rb").read())
        assert np.array_equal(np.stack(pa.frames), np.stack(pb.frames))
    enc3 = TilingEncoder()
    enc3.SetVideo(64, 48, 24.0, 8)  # one frame short: "Mismatch between GTM and loaded video!" (5021-5032)
    with pytest.raises(TileMotionError):
        enc3.ReloadGTM(a)
    for e in (enc, enc2, enc3):
        e.close()


def test_device_array_aliases_encoder_memory():
    """the multi-GPU merge all-reduces these views in place (tiler_amd/distributed.py): they must alias the encoder's arrays"""
    from tiler_amd import synth
    frames = synth.video(3, 32, 32, cut=2)
    enc = _run_encoder(frames, PaletteCount=1, MotionPredictRadius=4, FrameTilingExtendedPaletteUsage=False)
    t = enc.DeviceArray(0)
    assert t.dtype == torch.int32 and t.numel() == 3 * 16
    before = enc.TileMap(1)["TileIdx"].copy()
    t[16:32] += 1000
    torch.cuda.synchronize()
    assert np.array_equal(enc.TileMap(1)["TileIdx"], before + 1000)
    p = enc.DeviceArray(3)
    assert p.dtype == torch.uint8 and p.numel() == 3 * 16
    x = enc.DeviceArray(4)
    assert x.dtype == torch.int8
    x[16:32] = -7
    torch.cuda.synchronize()
    assert np.all(enc.TileMap(1)["PredictedX"] == -7)
    # a 1-rank "all_reduce" through the same code path as the bench
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29617", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(x, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()
    assert np.array_equal(enc.TileMap(1)["TileIdx"], before + 1000)
    dist.destroy_process_group()
    enc.close()


class _FakeDist:
    """torch.distributed stand-in for ranks that are threads of one process sharing one GPU: all_reduce with MAX / SUM over the
    tensors the ranks pass in the same call order (what RCCL would do across GPUs)"""

    class ReduceOp:
        MAX, SUM = "max", "sum"

    def __init__(self, world):
        import threading
        self.world = world
        self.bar = threading.Barrier(world)
        self.slots = [None] * world
        self.local = threading.local()

    def all_reduce(self, t, op=None, group=None):
        r = self.local.rank
        self.slots[r] = t
        self.bar.wait()
        if r == 0:
            acc = self.slots[0].clone()
            for o in self.slots[1:]:
                acc = torch.maximum(acc, o) if op == "max" else acc + o
            for o in self.slots:
                o.copy_(acc)
            torch.cuda.synchronize()
        self.bar.wait()

    def all_gather_into_tensor(self, recv, send, group=None):
        r = self.local.rank
        self.slots[r] = (recv, send)
        self.bar.wait()
        if r == 0:
            cat = torch.cat([s for _, s in self.slots])
            for o, _ in self.slots:
                o.copy_(cat)
            torch.cuda.synchronize()
        self.bar.wait()


@pytest.mark.parametrize("radius,epu,world,pp_sharded", [(0, False, 2, False), (0, True, 2, True), (8, False, 2, True), (8, True, 2, False), (8, True, 3, True),
                                                         (8, False, 6, False), (0, False, 5, True), (0, False, 4, False)])
def test_sharded_ranks_merge_to_the_single_run(monkeypatch, radius, epu, world, pp_sharded):
    """tiler_amd.distributed.run_all with REAL encoders: the ranks (threads, one GPU) shard Load / Reduce / PreparePalettes (data-parallel
    Lloyd, palette-parallel quantisation) / Dither / Reconstruct -- with motion prediction on: PredictMotion and whole key-frame groups --
    through the library's collective callback, and must end with exactly the single-process result.  With 6 ranks on 4 key frames some
    ranks own no frame at all; with 5 ranks on 3 palettes some own no palette."""
    import threading
    from tiler_amd import synth, distributed
    from tiler_amd.encoder import TilingEncoder
    # pp_sharded: the tile -> palette clustering data-parallel (run_palettize_dist: an all-reduce per Lloyd iteration) instead of run whole by
    # every rank (one resident launch each; the ranks are threads here: the launches take their turns)
    if pp_sharded:
        monkeypatch.setenv("TM_PP_SHARDED", "1")
    else:
        monkeypatch.delenv("TM_PP_SHARDED", raising=False)
    frames = synth.video(12, 64, 48, cut=3)
    kw = dict(PaletteCount=3, ShotTransMinSecondsPerKF=0.1, MotionPredictRadius=radius, FrameTilingExtendedPaletteUsage=epu, GlobalTilingTileCount=150)
    ref = _run_encoder(frames, **kw)
    want = (np.stack([ref.TileMap(f) for f in range(12)]), ref.Tiles())
    ref.close()
    fake = _FakeDist(world)
    monkeypatch.setattr(distributed, "dist", fake)
    out, errs = [None] * world, []

    def rank_main(r):
        try:
            fake.local.rank = r
            torch.cuda.set_device(0)
            enc = TilingEncoder()
            enc.LoadDefaultSettings()
            for k, v in kw.items():
                setattr(enc, k, v)
            enc.SetVideo(64, 48, 24.0, 12)
            for f in range(12):
                enc.PushFrame(f, frames[f])
            distributed.run_all(enc, 12, r, world)
            out[r] = (np.stack([enc.TileMap(f) for f in range(12)]), enc.Tiles())
            enc.close()
        except Exception as e:  # noqa: BLE001
            errs.append(e)
            fake.bar.abort()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not errs, errs
    for r in range(world):
        assert np.array_equal(out[r][0], want[0])
        for a, b in zip(out[r][1], want[1]):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("mode", ["keys", "keys-colliding-hashes"])
def test_sharded_reduce_moves_only_what_can_survive(monkeypatch, mode):
    """Reduce over three processes on a clip whose static tile columns repeat across the shard borders and whose tile budget bites (1 500 of
    ~7 000 distinct tiles): the processes exchange 16-byte keys, choose the tiles that can be among the first 1 500 of the merged order and
    all-gather only those; the result is the single run's, also when every hash collides (TM_DEDUP_DEGRADE_HASH:
    four hash values for all tiles -- everything is then a candidate); and the key path moves a fraction of the bytes a copy of every
    distinct tile would (what rounds 1-2 gathered)."""
    import threading
    from tiler_amd import synth, distributed
    from tiler_amd.encoder import TilingEncoder, TEncoderStep
    nf, w, h, world = 40, 160, 96, 3
    frames = synth.video(nf, w, h, cut=17)
    kw = dict(PaletteCount=3, ShotTransMinSecondsPerKF=0.1, MotionPredictRadius=0, FrameTilingExtendedPaletteUsage=False, GlobalTilingTileCount=1500)
    ref = _run_encoder(frames, **kw)
    want = (np.stack([ref.TileMap(f) for f in range(nf)]), ref.Tiles())
    ref.close()
    if mode == "keys-colliding-hashes":
        monkeypatch.setenv("TM_DEDUP_DEGRADE_HASH", "1")
    fake = _FakeDist(world)
    monkeypatch.setattr(distributed, "dist", fake)
    out, nbytes, errs = [None] * world, [0] * world, []

    def rank_main(r):
        try:
            fake.local.rank = r
            torch.cuda.set_device(0)
            enc = TilingEncoder()
            enc.LoadDefaultSettings()
            for k, v in kw.items():
                setattr(enc, k, v)
            enc.SetVideo(w, h, 24.0, nf)
            for f in range(nf):
                enc.PushFrame(f, frames[f])
            coll = distributed.Collective(r, world)
            enc.SetCollective(r, world, coll)
            enc._collective = coll
            first, count = distributed.frame_shard(nf, r, world)
            enc.SetQueryShard(first, count)
            enc.Run(TEncoderStep.esLoad)
            enc.CollectiveStats(reset=True)
            enc.Run(TEncoderStep.esReduce)
            nbytes[r] = enc.CollectiveStats()["bytes"]
            for st in (TEncoderStep.esPreparePalettes, TEncoderStep.esDither, TEncoderStep.esReconstruct, TEncoderStep.esReindex):
                enc.Run(st)
            out[r] = (np.stack([enc.TileMap(f) for f in range(nf)]), enc.Tiles())
            enc.close()
        except Exception as e:  # noqa: BLE001
            errs.append(e)
            fake.bar.abort()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=180)
    assert not errs, errs
    for r in range(world):
        assert np.array_equal(out[r][0], want[0])
        for a, b in zip(out[r][1], want[1]):
            assert np.array_equal(a, b)
    distinct = len(np.unique(np.concatenate([synth_tiles for synth_tiles in [frames.reshape(nf, h // 8, 8, w // 8, 8).transpose(0, 1, 3, 2, 4).reshape(-1, 64)]]), axis=0))
    if mode == "keys":
        assert distinct > 3 * 1500  # the budget bites
        assert nbytes[0] < 0.5 * distinct * 264 * 1  # far below one copy of every distinct tile (gathering them all moves world x that)


def test_motion_search_forms_agree(monkeypatch):
    """the encoder's motion search -- window features made in the matrix search's layout at once (k_window_dcts<true>) -- against the two-pass form
    (int16 window features, k_mo_pack_win; TM_MOTION_PACK_SEPARATE=1) and against the fallback of a frame beyond the matrix form's exact range
    (int16 windows + the VALU search, TM_MOTION_FORCE_FLAG=1), on frames whose window rows end inside a block of 32 and whose last strip is ragged"""
    from tiler_amd import synth
    frames = synth.video(5, 232, 144, cut=3, noise=40)
    got = {}
    for env in (None, "TM_MOTION_PACK_SEPARATE", "TM_MOTION_FORCE_FLAG"):
        for k in ("TM_MOTION_PACK_SEPARATE", "TM_MOTION_FORCE_FLAG"):
            monkeypatch.delenv(k, raising=False)
        if env:
            monkeypatch.setenv(env, "1")
        enc = _run_encoder(frames, PaletteCount=2, ShotTransMinSecondsPerKF=0.1, MotionPredictRadius=32, FrameTilingExtendedPaletteUsage=False, GlobalTilingTileCount=900)
        maps = [enc.TileMap(f) for f in range(5)]
        got[env] = [np.concatenate([np.asarray(m[k]).astype(np.int64).ravel() for m in maps]) for k in ("TileIdx", "PalIdx", "Flags", "PredictedX", "PredictedY")] + \
                   [np.concatenate([np.asarray(m["PSNR"]).view(np.uint32).astype(np.int64).ravel() for m in maps])]
    assert sum(int(((m >> 2) & 1).sum()) for m in [got[None][2]]) > 0  # some items are motion predicted
    for env in ("TM_MOTION_PACK_SEPARATE", "TM_MOTION_FORCE_FLAG"):
        for a, b in zip(got[None], got[env]):
            assert np.array_equal(a, b), env


@pytest.mark.parametrize("radius", [0, 8])
def test_y4m_and_png_export_show_what_the_player_shows(oracle, tmp_path, radius):
    """GenerateY4M / GeneratePNGs (tilingencoder.pas:2126-2199, 2075-2124): the rendered output frames must be the pictures the reference's
    player (gtm.player.js semantics, tests/gtm_reader.py, pinned by the demo streams) decodes from the saved .gtm -- as RGB in the PNGs,
    as RGBToYUV (utils.pas:478-490) rounded planes in the .y4m; the input export gives back the source frames"""
    import zlib
    from tiler_amd import synth
    from tests import gtm_reader
    frames = synth.video(6, 64, 48, cut=3)
    out = str(tmp_path / "clip.gtm")
    enc = _run_encoder(frames, PaletteCount=3, ShotTransMinSecondsPerKF=0.1, MotionPredictRadius=radius, FrameTilingExtendedPaletteUsage=False,
                       GlobalTilingTileCount=120, OutputFileName=out)
    _, player = gtm_reader.play(oracle, open(out, "rb").read())
    shown = [np.asarray(f) & 0xFFFFFF for f in player.frames]
    assert len(shown) == 6
    if radius:
        assert any(((enc.TileMap(f)["Flags"] >> 2) & 1).any() for f in range(6))  # some items are motion predicted

    def png_rgb(path):
        b = open(path, "rb").read()
        assert b[:8] == b"\x89PNG\r\n\x1a\n"
        pos, idat, w, h = 8, b"", 0, 0
        while pos < len(b):
            n = int.from_bytes(b[pos:pos + 4], "big")
            typ, data = b[pos + 4:pos + 8], b[pos + 8:pos + 8 + n]
            assert zlib.crc32(typ + data) == int.from_bytes(b[pos + 8 + n:pos + 12 + n], "big")
            if typ == b"IHDR":
                w, h = int.from_bytes(data[:4], "big"), int.from_bytes(data[4:8], "big")
                assert data[8:] == bytes([8, 2, 0, 0, 0])
            if typ == b"IDAT":
                idat += data
            pos += 12 + n
        raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * 3)
        assert (raw[:, 0] == 0).all()
        px = raw[:, 1:].reshape(h, w, 3).astype(np.uint32)
        return px[:, :, 0] | (px[:, :, 1] << 8) | (px[:, :, 2] << 16)

    enc.GeneratePNGs(False)
    for f in range(6):
        assert np.array_equal(png_rgb(str(tmp_path / ("clip_%04d.png" % f))), shown[f])
    pal_lines = open(str(tmp_path / "clip.txt")).read().split()
    assert pal_lines == ["%08X" % (0xFF000000 | (int(c) & 0xFFFFFFFF)) for c in enc.Palettes().reshape(-1)]

    def y4m_frames(path):
        b = open(path, "rb").read()
        head, rest = b.split(b"\n", 1)
        assert head == b"YUV4MPEG2 W64 H48 F24000000:1000000 Ip C444"
        out_f, n = [], 64 * 48 * 3
        while rest:
            assert rest[:7] == b"FRAME \n"
            out_f.append(np.frombuffer(rest[7:7 + n], np.uint8).reshape(3, 48, 64))
            rest = rest[7 + n:]
        return out_f

    def to_yuv(img):
        import ctypes
        o = np.zeros((3,) + img.shape, np.uint8)
        y, u, v = ctypes.c_float(), ctypes.c_float(), ctypes.c_float()
        for (yy, xx), c in np.ndenumerate(img):
            oracle.L.tmo_rgb_to_yuv(int(c) & 255, (int(c) >> 8) & 255, (int(c) >> 16) & 255, ctypes.byref(y), ctypes.byref(u), ctypes.byref(v))
            for k, val in enumerate((y.value, u.value + np.float32(128), v.value + np.float32(128))):
                o[k, yy, xx] = min(255, max(0, int(np.rint(np.float32(val)))))
        return o

    y4 = str(tmp_path / "out.y4m")
    enc.GenerateY4M(y4, False)
    got = y4m_frames(y4)
    assert len(got) == 6
    for f in range(6):
        assert np.array_equal(got[f], to_yuv(shown[f]))
    y4i = str(tmp_path / "in.y4m")
    enc.GenerateY4M(y4i, True)
    src = [((frames[f] & 0xFF) << 16) | (frames[f] & 0xFF00) | ((frames[f] >> 16) & 0xFF) for f in range(6)]  # RGB32 -> 0x00BBGGRR
    got = y4m_frames(y4i)
    for f in range(6):
        assert np.array_equal(got[f], to_yuv(np.asarray(src[f], np.uint32)))
    enc.close()


@pytest.mark.parametrize("tc,full", [(40, False), (151, False), (400, False), (151, True)])
def test_tile_budget_without_motion_prediction_keeps_the_first_tiles_of_the_order(oracle, monkeypatch, tc, full):
    """MotionPredictRadius = 0 with a GlobalTilingTileCount below the number of distinct tiles: Reduce keeps the first `tc` tiles of
    ReindexTiles' order (use count descending, content ascending).  The library orders only the rows that can be among them (a histogram
    of the use counts, one of the leading dword among the rows at the cut-off's count, a sort of those candidates: VERDICT r02 item 8);
    the oracle sorts everything.  Budgets that cut inside the single-use tiles, inside the multiply used ones, and the full sort forced
    (TM_DEDUP_FULL_ORDER) all give the oracle's tiles, palettes and tile maps."""
    from tiler_amd import synth
    from tests import oracle_pipeline
    if full:
        monkeypatch.setenv("TM_DEDUP_FULL_ORDER", "1")
    frames = synth.video(9, 104, 56, cut=4)
    exp = oracle_pipeline.run(oracle, frames, palette_count=2, min_s=0.1, motion_radius=0, tile_count=tc)
    enc = _run_encoder(frames, PaletteCount=2, ShotTransMinSecondsPerKF=0.1, MotionPredictRadius=0, FrameTilingExtendedPaletteUsage=False, GlobalTilingTileCount=tc)
    assert enc.counts()["tiles"] == exp["final_T"]
    hdr, pal, rgb = enc.Tiles()
    assert np.array_equal(pal, exp["final_pal_px"]) and np.array_equal(hdr["UseCount"], exp["final_use"]) and np.array_equal(rgb, exp["final_rgb"])
    assert np.array_equal(enc.Palettes(), exp["palettes"])
    per = exp["per"]
    for f in range(9):
        tm = enc.TileMap(f)
        sl = slice(f * per, (f + 1) * per)
        assert np.array_equal(tm["TileIdx"], exp["final_tm_tile"][sl]) and np.array_equal(tm["PalIdx"], exp["tm_pal"][sl])
    enc.close()


def _state(enc, nframes):
    hdr, pal, rgb = enc.Tiles()
    maps = [enc.TileMap(f) for f in range(nframes)]
    return dict(use=hdr["UseCount"].copy(), pal=pal.copy(), rgb=rgb.copy(), palettes=enc.Palettes().copy(), keyframes=np.array(enc.KeyFrames()),
                correl=enc.FrameCorrelations().view(np.uint32).copy(), tile=np.concatenate([m["TileIdx"] for m in maps]),
                flags=np.concatenate([m["Flags"] for m in maps]), mpal=np.concatenate([m["PalIdx"] for m in maps]))


def _same(a, b):
    return all(np.array_equal(a[k], b[k]) for k in a)


@pytest.mark.parametrize("pinned", [True, False])
def test_frames_from_host_memory_equal_frames_on_the_device(pinned):
    """tm_set_frames_host: a clip of several 48 MB chunks -- 200 frames of 320x200 go as 187 + 13 -- read
    from page-locked and from pageable host memory gives what the same clip resident in HBM gives; the host clip is lent only until
    the Load that reads it has returned (it is overwritten afterwards and Load re-run on the encoder's own copy); and a clip moved
    by tm_prefetch_frames_host beside the previous clip's steps is adopted by its Load."""
    from tiler_amd import synth
    from tiler_amd.encoder import TilingEncoder, TEncoderStep
    nf, h, w = 200, 200, 320
    clip_a = synth.video(nf, w, h, cut=50)
    clip_b = synth.video(nf, w, h, seed=7, cut=70)

    def host(arr):
        t = torch.from_numpy(arr.view(np.int32).copy())
        return t.pin_memory() if pinned else t

    def fresh():
        enc = TilingEncoder()
        enc.LoadDefaultSettings()
        enc.PaletteCount = 4
        enc.MotionPredictRadius = 0
        enc.FrameTilingExtendedPaletteUsage = False
        enc.SetVideo(w, h, 24.0, nf)
        return enc

    ref = {}
    for name, clip in (("a", clip_a), ("b", clip_b)):
        enc = fresh()
        dev = torch.from_numpy(clip.view(np.int32)).cuda()
        enc.SetFramesDevice(dev)
        enc.Run()
        ref[name] = _state(enc, nf)
        enc.close()
    assert not _same(ref["a"], ref["b"])

    enc = fresh()
    ha, hb = host(clip_a), host(clip_b)
    enc.SetFramesHost(ha)
    enc.Run()
    assert _same(_state(enc, nf), ref["a"])
    ha.zero_()                       # the loan ended with that Load: the memory is the host's again
    enc.Run()                        # Load again, without new frames: the encoder's own device copy
    assert _same(_state(enc, nf), ref["a"])
    ha.copy_(torch.from_numpy(clip_a.view(np.int32)))
    # clips back to back: b crosses PCIe while a's steps run, then a again while b's run
    enc.PrefetchFramesHost(ha)
    enc.SetFramesHost(ha)
    enc.PrefetchFramesHost(hb)
    enc.Run()
    assert _same(_state(enc, nf), ref["a"])
    enc.SetFramesHost(hb)
    enc.PrefetchFramesHost(ha)
    enc.Run()
    assert _same(_state(enc, nf), ref["b"])
    with pytest.raises(Exception):   # two clips already wait beside the one in flight?  No: one waits (a) -- a second prefetch takes the
        enc.PrefetchFramesHost(hb)   # last Load's buffer, a third has nowhere to go
        enc.PrefetchFramesHost(hb)
    enc.SetFramesHost(ha)
    enc.Run()
    assert _same(_state(enc, nf), ref["a"])
    enc.close()
