"""Host-side mirror of the reference's TTilingEncoder (tilingencoder.pas:308-568) over the coarse C ABI.

Same names and argument meaning as the Pascal class: LoadDefaultSettings / LoadSettings, the settings properties
(INI key names, tilingencoder.pas:3745-3770), Run(step) with TEncoderStep values, Tiles / TileMap / Palettes views.
Errors that the reference raises as exceptions/assertions surface as TileMotionError.
"""
import ctypes
import os
import enum

import numpy as np

from ._lib import lib, check, TileMotionError, c_void_p, c_int, c_int64, c_double, c_char_p


class TEncoderStep(enum.IntEnum):  # tilingencoder.pas:18
    esAll = -1
    esLoad = 0
    esPredictMotion = 1
    esReduce = 2
    esPreparePalettes = 3
    esDither = 4
    esReconstruct = 5
    esReindex = 6
    esSave = 7


class TPsyVisMode(enum.IntEnum):  # tilingencoder.pas:21
    pvsDCT = 0
    pvsWeightedDCT = 1
    pvsWavelets = 2
    pvsSpeDCT = 3
    pvsWeightedSpeDCT = 4


TILE_HDR = np.dtype([("UseCount", "<u4"), ("TmpIndex", "<i4"), ("MergeIndex", "<i4"), ("PalIdx_Initial", "<i4"), ("Flags", "<u4")])
TILEMAP_ITEM = np.dtype([("TileIdx", "<i4"), ("PalIdx", "<i4"), ("PredictedX", "i1"), ("PredictedY", "i1"), ("PSNR", "<f4"),
                         ("Flags", "<u4")])  # packed, 18 bytes (tilingencoder.pas:178-184)
assert TILE_HDR.itemsize == 20 and TILEMAP_ITEM.itemsize == 18

_ENC_SIGS = {
    "tm_create": (c_void_p, []),
    "tm_destroy": (None, [c_void_p]),
    "tm_set_device": (c_int, [c_void_p, c_int]),
    "tm_load_default_settings": (c_int, [c_void_p]),
    "tm_load_settings_ini": (c_int, [c_void_p, c_char_p]),
    "tm_set_int": (c_int, [c_void_p, c_char_p, c_int64]),
    "tm_set_float": (c_int, [c_void_p, c_char_p, c_double]),
    "tm_set_bool": (c_int, [c_void_p, c_char_p, c_int]),
    "tm_set_str": (c_int, [c_void_p, c_char_p, c_char_p]),
    "tm_get_int": (c_int, [c_void_p, c_char_p, ctypes.POINTER(c_int64)]),
    "tm_get_float": (c_int, [c_void_p, c_char_p, ctypes.POINTER(c_double)]),
    "tm_set_progress_cb": (c_int, [c_void_p, c_void_p, c_void_p]),
    "tm_set_video": (c_int, [c_void_p, c_int, c_int, c_double, c_int]),
    "tm_push_frame_rgb32": (c_int, [c_void_p, c_int, c_void_p, c_int]),
    "tm_set_frames_device": (c_int, [c_void_p, c_void_p]),
    "tm_set_frames_host": (c_int, [c_void_p, c_void_p]),
    "tm_prefetch_frames_host": (c_int, [c_void_p, c_void_p]),
    "tm_run": (c_int, [c_void_p, c_int]),
    "tm_get_counts": (c_int, [c_void_p, ctypes.POINTER(c_int64)] + [ctypes.POINTER(c_int)] * 5),
    "tm_get_tile": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "tm_get_tiles": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "tm_get_tilemap": (c_int, [c_void_p, c_int, c_void_p]),
    "tm_get_tilemaps": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "tm_get_palette": (c_int, [c_void_p, c_int, c_void_p]),
    "tm_get_keyframes": (c_int, [c_void_p, c_void_p]),
    "tm_get_frame_correlations": (c_int, [c_void_p, c_void_p]),
    "tm_get_stage_ms": (c_int, [c_void_p, c_void_p]),
    "tm_get_psnr": (c_int, [c_void_p, c_void_p, ctypes.POINTER(c_double)]),
    "tm_save_gtm": (c_int, [c_void_p, c_char_p]),
    "tm_reload_gtm": (c_int, [c_void_p, c_char_p]),
    "tm_generate_y4m": (c_int, [c_void_p, c_char_p, c_int]),
    "tm_generate_pngs": (c_int, [c_void_p, c_int]),
    "tm_set_query_shard": (c_int, [c_void_p, c_int, c_int]),
    "tm_set_dither_shard": (c_int, [c_void_p, c_int, c_int]),
    "tm_set_collective": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "tm_get_stream": (c_void_p, [c_void_p]),
    "tm_set_collective_mode": (c_int, [c_void_p, c_int]),
    "tm_get_device_array": (c_int, [c_void_p, c_int, ctypes.POINTER(c_void_p), ctypes.POINTER(c_int64)]),
    "tm_sync_tilemap": (c_int, [c_void_p]),
    "tm_get_knn_kernel_split": (c_int, [c_void_p, ctypes.POINTER(c_double), ctypes.POINTER(c_int64)]),
    "tm_get_knn_stats": (c_int, [c_void_p, ctypes.POINTER(c_double), ctypes.POINTER(c_int64), ctypes.POINTER(c_int), ctypes.POINTER(c_int),
                         ctypes.POINTER(c_int64)]),
}

_INT_KEYS = ["StartFrame", "FrameCount", "MotionPredictRadius", "GlobalTilingTileCount", "PaletteSize", "PaletteCount", "DitheringMode",
             "DitheringYliluoma2MixedColors", "MaxThreadCount"]
_BOOL_KEYS = ["GlobalTilingUseTargetPSNR", "DitheringUseThomasKnoll", "FrameTilingExtendedPaletteUsage"]
_FLOAT_KEYS = ["Scaling", "GlobalTilingTargetPSNR", "GlobalTilingQualityBasedTileCount", "ShotTransMaxSecondsPerKF",
               "ShotTransMinSecondsPerKF", "ShotTransCorrelLoThres"]


def _bind():
    L = lib()
    if not getattr(L, "_enc_bound", False):
        for name, (res, args) in _ENC_SIGS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        L._enc_bound = True
    return L


class TilingEncoder:
    """TTilingEncoder: Create -> settings -> SetVideo/PushFrame (the FFMPEG callback's contract) -> Run(step)."""

    def __init__(self):
        self._L = _bind()
        self._h = self._L.tm_create()
        if not self._h:
            check(-2)
        self._frames_ref = None

    _h = None  # class-level default: close() / __del__ are safe on an instance whose __init__ failed early
    _L = None

    def close(self):
        h = self.__dict__.get("_h")
        if h:
            self._L.tm_destroy(c_void_p(h))
            object.__setattr__(self, "_h", None)

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 (interpreter shutdown)
            pass

    # -- settings (properties named like the Pascal ones)
    def __getattr__(self, name):
        if name in _INT_KEYS or name in _BOOL_KEYS:
            v = c_int64()
            check(self._L.tm_get_int(c_void_p(self._h), name.encode(), ctypes.byref(v)))
            return bool(v.value) if name in _BOOL_KEYS else v.value
        if name in _FLOAT_KEYS:
            v = c_double()
            check(self._L.tm_get_float(c_void_p(self._h), name.encode(), ctypes.byref(v)))
            return v.value
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name in _INT_KEYS:
            check(self._L.tm_set_int(c_void_p(self._h), name.encode(), int(value)))
        elif name in _BOOL_KEYS:
            check(self._L.tm_set_bool(c_void_p(self._h), name.encode(), int(bool(value))))
        elif name in _FLOAT_KEYS:
            check(self._L.tm_set_float(c_void_p(self._h), name.encode(), float(value)))
        elif name in ("InputFileName", "OutputFileName"):
            check(self._L.tm_set_str(c_void_p(self._h), name.encode(), str(value).encode()))
        else:
            object.__setattr__(self, name, value)

    def LoadDefaultSettings(self):
        check(self._L.tm_load_default_settings(c_void_p(self._h)))

    def LoadSettings(self, path):
        check(self._L.tm_load_settings_ini(c_void_p(self._h), str(path).encode()))

    # -- video
    def SetVideo(self, width, height, fps, frame_count):
        check(self._L.tm_set_video(c_void_p(self._h), width, height, float(fps), frame_count))

    def PushFrame(self, index, pixels):
        """pixels: numpy uint32 [height][width] RGB32 (AV_PIX_FMT_RGB32), host memory, read during the call only"""
        a = np.ascontiguousarray(pixels, dtype=np.uint32)
        check(self._L.tm_push_frame_rgb32(c_void_p(self._h), index, a.ctypes.data_as(c_void_p), a.shape[1]))

    def SetFramesDevice(self, frames):
        """frames: torch int32 CUDA tensor [F][H][W] holding RGB32; borrowed until the encoder is closed"""
        self._frames_ref = frames
        check(self._L.tm_set_frames_device(c_void_p(self._h), c_void_p(frames.data_ptr())))

    def SetFramesHost(self, frames):
        """frames: torch int32 CPU tensor (ideally pinned) or numpy uint32 array [F][H][W] holding RGB32; borrowed: Load copies it to
        the device in chunks beside its own kernel"""
        self._frames_ref = frames
        ptr = frames.data_ptr() if hasattr(frames, "data_ptr") else frames.ctypes.data
        check(self._L.tm_set_frames_host(c_void_p(self._h), c_void_p(ptr)))

    def SaveSettings(self, path):
        self._L.tm_save_settings_ini.restype = c_int
        self._L.tm_save_settings_ini.argtypes = [c_void_p, c_char_p]
        check(self._L.tm_save_settings_ini(c_void_p(self._h), str(path).encode()))

    def PrefetchFramesHost(self, frames):
        """queues the upload of the clip the NEXT Load will read (same argument forms as SetFramesHost) beside the current clip's steps;
        call SetFramesHost with the same clip before the Run that is to adopt it.  Borrowed until that Load has returned."""
        self._prefetch_ref = frames
        ptr = frames.data_ptr() if hasattr(frames, "data_ptr") else frames.ctypes.data
        check(self._L.tm_prefetch_frames_host(c_void_p(self._h), c_void_p(ptr)))

    def Run(self, step=TEncoderStep.esAll):
        check(self._L.tm_run(c_void_p(self._h), int(step)))

    # -- read-back
    def counts(self):
        t = c_int64()
        v = [c_int() for _ in range(5)]
        check(self._L.tm_get_counts(c_void_p(self._h), ctypes.byref(t), *[ctypes.byref(x) for x in v]))
        return dict(tiles=t.value, frames=v[0].value, palettes=v[1].value, tm_w=v[2].value, tm_h=v[3].value, keyframes=v[4].value)

    def Tiles(self, first=0, count=None):
        n = self.counts()["tiles"]
        count = n - first if count is None else count
        hdr = np.zeros(count, TILE_HDR)
        pal = np.zeros((count, 64), np.uint8)
        rgb = np.zeros((count, 64), np.uint32)
        check(self._L.tm_get_tiles(c_void_p(self._h), first, count, hdr.ctypes.data_as(c_void_p), pal.ctypes.data_as(c_void_p),
                                   rgb.ctypes.data_as(c_void_p)))
        return hdr, pal, rgb

    def TileMap(self, frame):
        c = self.counts()
        items = np.zeros(c["tm_w"] * c["tm_h"], TILEMAP_ITEM)
        check(self._L.tm_get_tilemap(c_void_p(self._h), frame, items.ctypes.data_as(c_void_p)))
        return items

    def TileMaps(self, first=0, count=None, out=None):
        """Frames[first .. first+count-1].TileMap in one read-back; `out` (optional) = a uint8 array / tensor of count*tm_w*tm_h*18 bytes to
        fill (page-locked memory makes the copy one DMA)"""
        c = self.counts()
        count = c["frames"] - first if count is None else count
        per = c["tm_w"] * c["tm_h"]
        if out is None:
            items = np.zeros(count * per, TILEMAP_ITEM)
            check(self._L.tm_get_tilemaps(c_void_p(self._h), first, count, items.ctypes.data_as(c_void_p)))
            return items.reshape(count, per)
        ptr = out.data_ptr() if hasattr(out, "data_ptr") else out.ctypes.data
        check(self._L.tm_get_tilemaps(c_void_p(self._h), first, count, c_void_p(ptr)))
        return out

    def Palettes(self):
        c = self.counts()
        out = np.zeros((c["palettes"], self.PaletteSize), np.int32)
        for i in range(c["palettes"]):
            check(self._L.tm_get_palette(c_void_p(self._h), i, out[i].ctypes.data_as(c_void_p)))
        return out

    def KeyFrames(self):
        n = self.counts()["keyframes"]
        out = np.zeros(n, np.int32)
        check(self._L.tm_get_keyframes(c_void_p(self._h), out.ctypes.data_as(c_void_p)))
        return out

    def FrameCorrelations(self):
        out = np.zeros(self.counts()["frames"], np.float32)
        check(self._L.tm_get_frame_correlations(c_void_p(self._h), out.ctypes.data_as(c_void_p)))
        return out

    def PSNR(self):
        """TKeyFrame.LogPSNR (tilingencoder.pas:1006-1028): {"per_keyframe": [...], "global": mean PSNR-HVS by tile}"""
        n = self.counts()["keyframes"]
        per = np.zeros(max(n, 1), np.float64)
        g = c_double()
        check(self._L.tm_get_psnr(c_void_p(self._h), per.ctypes.data_as(c_void_p), ctypes.byref(g)))
        return {"per_keyframe": [float(v) for v in per[:n]], "global": g.value, "unit": "dB, PSNR-HVS by tile (LogPSNR)"}

    def StageMs(self):
        out = np.zeros(8, np.float64)
        check(self._L.tm_get_stage_ms(c_void_p(self._h), out.ctypes.data_as(c_void_p)))
        return out

    # -- multi-GPU plumbing (one process per GPU; collectives stay in the host, see tiler_amd/distributed.py)
    def SetQueryShard(self, first_frame, frame_count):
        check(self._L.tm_set_query_shard(c_void_p(self._h), first_frame, frame_count))

    # -- the native multi-process path: RCCL inside the library (tm_comm_*)
    @staticmethod
    def CommUniqueId():
        """128 bytes from ncclGetUniqueId: one process makes them, every process of the job passes them to CommInit"""
        from ._lib import lib
        buf = (ctypes.c_uint8 * 128)()
        L = lib()
        L.tm_comm_unique_id.restype = c_int
        L.tm_comm_unique_id.argtypes = [c_void_p]
        check(L.tm_comm_unique_id(buf))
        return bytes(buf)

    def CommInit(self, comm_id, rank, world):
        """collective: returns once all `world` processes have called it.  From then on Run(step) shards and merges by itself."""
        assert len(comm_id) == 128
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(comm_id)
        self._L.tm_comm_init.restype = c_int
        self._L.tm_comm_init.argtypes = [c_void_p, c_void_p, c_int, c_int]
        check(self._L.tm_comm_init(c_void_p(self._h), buf, int(rank), int(world)))
        self._native_comm = (int(rank), int(world))

    def CommDestroy(self):
        self._L.tm_comm_destroy.restype = c_int
        self._L.tm_comm_destroy.argtypes = [c_void_p]
        check(self._L.tm_comm_destroy(c_void_p(self._h)))
        self._native_comm = None

    def CollectiveStats(self, reset=False):
        calls = (c_int64 * 4)()
        nbytes = c_int64()
        self._L.tm_get_collective_stats.restype = c_int
        self._L.tm_get_collective_stats.argtypes = [c_void_p, c_void_p, ctypes.POINTER(c_int64), c_int]
        check(self._L.tm_get_collective_stats(c_void_p(self._h), calls, ctypes.byref(nbytes), 1 if reset else 0))
        return dict(all_reduce_sum_i32=calls[0], all_reduce_max_i32=calls[1], all_reduce_sum_i64=calls[2], all_gather=calls[3], bytes=nbytes.value)

    def SetCollective(self, rank, world, coll):
        """one process per GPU: `coll` (tiler_amd.distributed.Collective) runs the collectives the steps ask for; world == 1 clears it"""
        self._coll_ref = coll  # the ctypes callback must outlive the encoder's use of it
        cb = ctypes.cast(coll.callback, c_void_p) if (coll is not None and world > 1) else None
        check(self._L.tm_set_collective(c_void_p(self._h), int(rank), int(world), cb, None))
        # a communicator that can queue its work on the encoder's own stream (RCCL through torch.distributed) is used stream-ordered
        ordered = bool(coll is not None and world > 1 and getattr(coll, "bind_stream", None) and coll.bind_stream(self._L.tm_get_stream(c_void_p(self._h))))
        check(self._L.tm_set_collective_mode(c_void_p(self._h), 1 if ordered else 0))

    def SetDitherShard(self, rank, world):
        """Dither only tiles [T * rank / world, T * (rank + 1) / world); the others stay 0 in DeviceArray(7) for an all-reduce(SUM)"""
        check(self._L.tm_set_dither_shard(c_void_p(self._h), rank, world))

    def DeviceArray(self, which):
        """torch view of an encoder-owned device array, no copy: 0 TileIdx, 1 error, 2 PalIdx (int32); with motion
        prediction 3 IsPredicted (uint8), 4/5 PredictedX/Y (int8), 6 PredictMotion's best error (int32); 7 the dithered
        tiles' palette indices, seen as int32 words (64 bytes = 16 words per tile) so that any backend can add them"""
        import torch
        ptr, cnt = c_void_p(), c_int64()
        check(self._L.tm_get_device_array(c_void_p(self._h), which, ctypes.byref(ptr), ctypes.byref(cnt)))
        typestr = {3: "|u1", 4: "|i1", 5: "|i1"}.get(int(which), "<i4")
        n = cnt.value // 4 if int(which) == 7 else cnt.value

        class _View:
            __cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr.value, False), "version": 2}

        return torch.as_tensor(_View(), device="cuda")

    def Save(self, path=None):
        """Save (tilingencoder.pas:2040) -> SaveStream (:5177): writes the .gtm; path defaults to OutputFileName"""
        if path is None:
            self.Run(TEncoderStep.esSave)
        else:
            check(self._L.tm_save_gtm(c_void_p(self._h), os.fsencode(path)))

    def GenerateY4M(self, path, input=False):
        """GenerateY4M (tilingencoder.pas:2126): the rendered output frames (or the source frames) as a C444 .y4m"""
        check(self._L.tm_generate_y4m(c_void_p(self._h), os.fsencode(path), int(bool(input))))

    def GeneratePNGs(self, input=False):
        """GeneratePNGs (tilingencoder.pas:2075): <OutputFileName>_NNNN.png per frame + the palettes as <OutputFileName>.txt"""
        check(self._L.tm_generate_pngs(c_void_p(self._h), int(bool(input))))

    def ReloadGTM(self, path):
        """ReloadGTM (tilingencoder.pas:2059) -> LoadStream (:4880): tiles, palettes, tile maps, key frames from a .gtm"""
        check(self._L.tm_reload_gtm(c_void_p(self._h), os.fsencode(path)))

    def SyncTileMap(self):
        check(self._L.tm_sync_tilemap(c_void_p(self._h)))

    def KmeansIters(self):
        """iterations and points of the last PreparePalettes' two clusterings (tm_get_kmeans_iters)"""
        ti, pi = c_int(), c_int()
        tp, pc, px, pci = c_int64(), c_int64(), c_int64(), c_int64()
        self._L.tm_get_kmeans_iters.restype = c_int
        self._L.tm_get_kmeans_iters.argtypes = [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int64), ctypes.POINTER(c_int), ctypes.POINTER(c_int64), ctypes.POINTER(c_int64),
                                                ctypes.POINTER(c_int64)]
        check(self._L.tm_get_kmeans_iters(c_void_p(self._h), ctypes.byref(ti), ctypes.byref(tp), ctypes.byref(pi), ctypes.byref(pc), ctypes.byref(px), ctypes.byref(pci)))
        return dict(tile_iters=ti.value, tile_points=tp.value, pixel_iters=pi.value, pixel_colours=pc.value, pixel_points=px.value, pixel_colour_iters=pci.value)

    def DitherPairs(self):
        """distinct (palette, colour) pairs the last Dither planned once each; 0 = every pixel planned on its own"""
        self._L.tm_get_dither_pairs.restype = c_int64
        self._L.tm_get_dither_pairs.argtypes = [c_void_p]
        return int(self._L.tm_get_dither_pairs(c_void_p(self._h)))

    def KnnStats(self):
        ms, pairs, launches, kb, rows = c_double(), c_int64(), c_int(), c_int(), c_int64()
        check(self._L.tm_get_knn_stats(c_void_p(self._h), ctypes.byref(ms), ctypes.byref(pairs), ctypes.byref(launches), ctypes.byref(kb),
                                       ctypes.byref(rows)))
        self._L.tm_get_knn_queries.restype = c_int64
        self._L.tm_get_knn_queries.argtypes = [c_void_p]
        sm, sp = (c_double * 3)(), (c_int64 * 3)()
        check(self._L.tm_get_knn_kernel_split(c_void_p(self._h), sm, sp))
        return dict(kernel_ms=ms.value, pairs=pairs.value, launches=launches.value, k_bytes=kb.value, db_rows=rows.value,
                    queries=int(self._L.tm_get_knn_queries(c_void_p(self._h))),
                    seed_ms=sm[0], lists_ms=sm[1], consume_ms=sm[2], seed_pairs=sp[0], consume_pairs=sp[1], consume_mfma=sp[2])
