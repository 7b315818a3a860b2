"""Reader for .gtm files: header + per-keyframe LZMA streams + command parser and a renderer with the semantics of the
reference's HTML/JS player (decoders/htmljs/gtm.player.js:195-227 parseHeader, :365-515 decodeFrame, :276-331 draw*).
Test infrastructure: reads back what tm_write_gtm_host / tm_save_gtm wrote."""
import ctypes
import struct

import numpy as np

CMD_PRED_SHORT, CMD_PRED_LONG, CMD_SHORT_SHORT, CMD_LONG_SHORT, CMD_LONG_LONG, CMD_INTRA, CMD_SKIP = range(7)
CMD_FRAME_END, CMD_LOAD_PALETTE, CMD_TILE_SET, CMD_SET_DIMENSIONS, CMD_EXTENDED = 11, 12, 13, 14, 15


def lzma_decode(oracle, blob, cap):
    """-> (bytes, consumed, (props, dict_size, size_field)); oracle.L.tmo_lzma_decode restates lzma.js"""
    src = np.frombuffer(blob, np.uint8)
    dst = np.zeros(cap, np.uint8)
    consumed = ctypes.c_size_t()
    props = (ctypes.c_int * 3)()
    f = oracle.L.tmo_lzma_decode
    f.restype = ctypes.c_int64
    f.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
    n = f(src.ctypes.data, src.size, dst.ctypes.data, cap, ctypes.byref(consumed), props)
    assert n >= 0, "corrupt LZMA stream"
    return dst[:n].tobytes(), consumed.value, tuple(props)


def read_header(data):
    fcc, riff, whole, version, pw, ph, kfc, fc, avg, kfmax = struct.unpack_from("<4s9I", data, 0)
    assert fcc == b"GTMv" and riff == 32
    hdr = dict(whole=whole, version=version, width=pw, height=ph, kf_count=kfc, frame_count=fc, avg_bps=avg, kf_max_bps=kfmax, kf=[])
    for k in range(kfc):
        fcc, riff, idx, frame, raw, comp, ms = struct.unpack_from("<4s6I", data, 40 + 28 * k)
        assert fcc == b"GTMk" and riff == 20 and idx == k
        hdr["kf"].append(dict(frame=frame, raw=raw, comp=comp, ms=ms))
    assert whole == 40 + 28 * kfc
    return hdr


def unpack(oracle, data):
    """-> header, [raw command bytes per keyframe]"""
    hdr = read_header(data)
    pos = hdr["whole"]
    raws = []
    for kf in hdr["kf"]:
        raw, consumed, props = lzma_decode(oracle, data[pos:pos + kf["comp"]], kf["raw"] + 16)
        assert props == (0x62, 1 << 22, -1), props           # lc 8 / lp 0 / pb 2, 4 MiB, unknown size (extern.pas:427-436)
        assert len(raw) == kf["raw"] and consumed == kf["comp"]  # the player decodes stream after stream (wlzma.wrk.js:48)
        raws.append(raw)
        pos += kf["comp"]
    assert pos == len(data)
    return hdr, raws


class Player:
    """gtm.player.js state machine; frames come out as [H][W] uint32 0xAABBGGRR like the canvas' RGBA bytes."""

    def __init__(self, max_frames=None, render=True):
        self.max_frames = max_frames
        self.render = render  # False: walk the commands only (dimensions may then be preset by the caller)
        self.done = False
        self.w = self.h = 0
        self.tile_count = 0
        self.pal_size = 0
        self.palettes = {}
        self.tiles = None
        self.cur_intra = 0
        self.settings = None
        self.frame_ns = 0
        self.buf = [None, None]
        self.dbl = 0
        self.pos = 0
        self.frames = []
        self.items = []       # per frame: list of tuples describing each command that fills tile-map positions
        self._cur_items = []
        self.kf_ends = []
        self.tileset_ranges = []  # (first, last) of every TileSet command

    def _draw(self, idx, attrs):
        if not self.render:
            self.pos += 1
            return
        pal = self.palettes[attrs >> 2]
        t = self.tiles[idx]
        if attrs & 1:
            t = t[:, ::-1]
        if attrs & 2:
            t = t[::-1, :]
        x, y = (self.pos % self.w) * 8, (self.pos // self.w) * 8
        self.buf[self.dbl][y:y + 8, x:x + 8] = pal[t]
        self.pos += 1

    def _draw_pred(self, ox, oy):
        if not self.render:
            self.pos += 1
            return
        x, y = (self.pos % self.w) * 8, (self.pos // self.w) * 8
        self.buf[self.dbl][y:y + 8, x:x + 8] = self.buf[1 - self.dbl][y + oy:y + oy + 8, x + ox:x + ox + 8]
        self.pos += 1

    def feed(self, raw):
        p = 0
        u16 = lambda: struct.unpack_from("<H", raw, p)[0]
        u32 = lambda: struct.unpack_from("<I", raw, p)[0]
        while p < len(raw):
            word = u16(); p += 2
            cmd, arg = word & 15, word >> 4
            if cmd == CMD_SET_DIMENSIONS:
                self.w, self.h = struct.unpack_from("<HH", raw, p); p += 4
                self.frame_ns = u32(); p += 4
                self.tile_count = u32(); p += 4
                self.cur_intra = self.tile_count
                self.tiles = np.zeros((self.tile_count + self.w * self.h * 2, 8, 8), np.uint8)
                self.buf = [np.zeros((self.h * 8, self.w * 8), np.uint32) + 0xFF000000 for _ in range(2)]
            elif cmd == CMD_TILE_SET:
                a = u32(); p += 4
                b = u32(); p += 4
                self.pal_size = arg
                self.tileset_ranges.append((a, b))
                n = b - a + 1
                self.tiles[a:b + 1] = np.frombuffer(raw, np.uint8, n * 64, p).reshape(n, 8, 8); p += n * 64
            elif cmd == CMD_FRAME_END:
                assert self.pos == self.w * self.h, "incomplete tile map"
                self.pos = 0
                self.frames.append(self.buf[self.dbl].copy() if self.render else None)
                self.items.append(self._cur_items)
                self._cur_items = []
                self.kf_ends.append(arg & 1)
                self.dbl = 1 - self.dbl
                if self.max_frames is not None and len(self.frames) >= self.max_frames:
                    self.done = True
                    return
            elif cmd == CMD_SKIP:
                self._cur_items.append(("skip", arg + 1))
                for _ in range(arg + 1):
                    self._draw_pred(0, 0)
            elif cmd == CMD_SHORT_SHORT:
                t = u16(); p += 2
                self._cur_items.append(("ss", t, arg >> 2, arg & 3))
                self._draw(t, arg)
            elif cmd == CMD_LONG_SHORT:
                t = u32(); p += 4
                self._cur_items.append(("ls", t, arg >> 2, arg & 3))
                self._draw(t, arg)
            elif cmd == CMD_LONG_LONG:
                pal = u16(); p += 2
                t = u32(); p += 4
                self._cur_items.append(("ll", t, pal, arg & 3))
                self._draw(t, arg | (pal << 2))
            elif cmd == CMD_LOAD_PALETTE:
                idx = u16(); p += 2
                assert arg == 0
                self.palettes[idx] = np.frombuffer(raw, "<u4", self.pal_size, p).copy(); p += 4 * self.pal_size
            elif cmd == CMD_PRED_SHORT:
                ox, oy = (arg & 31) - (arg & 32), ((arg >> 6) & 31) - ((arg >> 6) & 32)
                self._cur_items.append(("ps", ox, oy))
                self._draw_pred(ox, oy)
            elif cmd == CMD_PRED_LONG:
                bx, by = raw[p], raw[p + 1]; p += 2
                ox, oy = (bx & 127) - (bx & 128), (by & 127) - (by & 128)
                self._cur_items.append(("pl", ox, oy))
                self._draw_pred(ox, oy)
            elif cmd == CMD_INTRA:
                pal = u16(); p += 2
                if self.tiles is not None:
                    self.tiles[self.cur_intra] = np.frombuffer(raw, np.uint8, 64, p).reshape(8, 8)
                p += 64
                self._cur_items.append(("intra", bytes(raw[p - 64:p]), pal, arg & 3))
                self._draw(self.cur_intra, arg | (pal << 2))
                self.cur_intra += 1
                if self.cur_intra >= self.tile_count + self.w * self.h * 2:
                    self.cur_intra = self.tile_count
            elif cmd == CMD_EXTENDED:
                n = u32(); p += 4
                text = raw[p:p + n]; p += n
                if arg == 0:
                    self.settings = text.decode("latin-1")
            else:
                raise AssertionError(f"undecoded command {cmd} @ {p}")
        assert p == len(raw)


def play(oracle, data):
    hdr, raws = unpack(oracle, data)
    pl = Player()
    for raw in raws:
        pl.feed(raw)
    return hdr, pl


def render_expected(pal_px, palettes, tilemaps, tm_w, tm_h):
    """frames straight from the encoder's tables: what TTilingEncoder.Render draws for non-predicted items
    (tilingencoder.pas:3949-4090: palette lookup of the mirrored palette-index tile)."""
    nf = tilemaps["TileIdx"].shape[0]
    out = np.zeros((nf, tm_h * 8, tm_w * 8), np.uint32)
    pals = np.where(palettes == np.int32(-65281), 0xFFFFFF, palettes).astype(np.uint32) | np.uint32(0xFF000000)
    for f in range(nf):
        t = pal_px[tilemaps["TileIdx"][f]].reshape(-1, 8, 8)
        fl = tilemaps["Flags"][f]
        t = np.where((fl & 1)[:, None, None] != 0, t[:, :, ::-1], t)
        t = np.where((fl & 2)[:, None, None] != 0, t[:, ::-1, :], t)
        px = np.take_along_axis(pals[tilemaps["PalIdx"][f]], t.reshape(-1, 64).astype(np.int64), axis=1).reshape(tm_h, tm_w, 8, 8)
        out[f] = px.transpose(0, 2, 1, 3).reshape(tm_h * 8, tm_w * 8)
    return out
