"""Derives the .gtm reader's pins from the reference's demo files (docs/demo/*.gtm -- data, not source) and writes
tests/golden/gtm_demo_pins.json + football_cif_kf1.lzma (the second keyframe's compressed stream, 352 KB, verbatim
data).  Run here (needs /root/reference; node optional: when present the reference's lzma.js must agree on every raw
stream).  Usage: python tests/golden/make_gtm_fixtures.py"""
import collections
import hashlib

import numpy as np
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import gtm_reader, oracle_binding  # noqa: E402

DEMO = "/root/reference/docs/demo"
REF_JS = "/root/reference/decoders/htmljs"
MAX_FRAMES = 40


def pins_for(oracle, name):
    data = open(os.path.join(DEMO, name + ".gtm"), "rb").read()
    hdr, raws = gtm_reader.unpack(oracle, data)
    out = dict(header={k: v for k, v in hdr.items() if k != "kf"}, kf=hdr["kf"], raw_sha256=[hashlib.sha256(r).hexdigest() for r in raws])
    if shutil.which("node"):
        with tempfile.TemporaryDirectory() as td:
            res = subprocess.run(["node", os.path.join(ROOT, "tests", "node_unpack_gtm.js"), REF_JS, os.path.join(DEMO, name + ".gtm"),
                                 os.path.join(td, "raw")], capture_output=True, text=True, check=True)
            assert open(os.path.join(td, "raw"), "rb").read() == b"".join(raws), "lzma.js disagrees with the oracle's decoder"
            assert res.stdout.split() == [str(len(r)) for r in raws]
        out["lzma_js_agrees"] = True
    pl = gtm_reader.Player(max_frames=MAX_FRAMES)
    for raw in raws:
        pl.feed(raw)
        if pl.done:
            break
    hist = collections.Counter(it[0] for fr in pl.items for it in fr)
    out.update(frames_played=len(pl.frames), tm_w=pl.w, tm_h=pl.h, frame_ns=pl.frame_ns, tile_count=pl.tile_count, pal_size=pl.pal_size,
               palettes=len(pl.palettes), settings_sha256=hashlib.sha256(pl.settings.encode("latin-1")).hexdigest(),
               settings_head=pl.settings[:40], item_histogram=dict(sorted(hist.items())),
               frame_sha256={str(f): hashlib.sha256(pl.frames[f].tobytes()).hexdigest() for f in (0, 1, MAX_FRAMES - 1)})
    # What the stream says about the encoder steps behind it: walk ALL key frames without rendering
    full = gtm_reader.Player(render=False)
    for raw in raws:
        full.feed(raw)
    n_ts = sum(b - a + 1 for a, b in full.tileset_ranges)
    ts = np.concatenate([full.tiles[a:b + 1] for a, b in full.tileset_ranges]).reshape(-1, 64)
    intra = np.frombuffer(b"".join(it[1] for fr in full.items for it in fr if it[0] == "intra"), np.uint8).reshape(-1, 64)
    both = np.concatenate([ts, intra])
    refs = np.array([it[1] for fr in full.items for it in fr if it[0] in ("ss", "ls", "ll")], np.int64)
    out["tiles"] = dict(
        declared=int(full.tile_count), tileset=int(n_ts), tileset_ranges=[list(r) for r in full.tileset_ranges], intra=int(intra.shape[0]),
        # MakeTilesUnique(False) (tilingencoder.pas:4702-4718, 1993-2038) keys tiles by their 64 palette-index bytes alone, whatever
        # their palette: no two tiles of a finished stream may share them.  Counted here with numpy, re-counted with the oracle's dedup in the test.
        distinct_index_tiles=int(np.unique(both, axis=0).shape[0]),
        # DoTMI (5208-5268): a tile used once travels inside its one item (intra); everything addressed by index lives in the TileSet,
        # which holds the tiles before the first UseCount = 1 one (5236, 5292-5316)
        max_explicit_reference=int(refs.max()), explicit_references=int(refs.shape[0]),
        never_visibly_referenced=int(full.tile_count - n_ts - intra.shape[0]))
    out["settings_text"] = pl.settings
    if len(raws) > 1:  # command walk of the second keyframe alone (what tests can redo from football_cif_kf1.lzma)
        w = gtm_reader.Player(render=False)
        w.w, w.h, w.tile_count = pl.w, pl.h, pl.tile_count
        w.feed(raws[1])
        out["kf1_walk"] = dict(frames=len(w.frames), item_histogram=dict(sorted(collections.Counter(it[0] for fr in w.items for it in fr).items())),
                               kf_ends=sum(w.kf_ends))
    return out, data, hdr


def main():
    oracle = oracle_binding.Oracle(os.path.join(ROOT, "oracle", "libtm_oracle.so"))
    pins = {}
    for name in ("city_cif", "football_cif"):
        pins[name], data, hdr = pins_for(oracle, name)
        if name == "football_cif":
            pos = hdr["whole"] + hdr["kf"][0]["comp"]
            with open(os.path.join(ROOT, "tests", "golden", "football_cif_kf1.lzma"), "wb") as f:
                f.write(data[pos:pos + hdr["kf"][1]["comp"]])
    with open(os.path.join(ROOT, "tests", "golden", "gtm_demo_pins.json"), "w") as f:
        json.dump(pins, f, indent=1, sort_keys=True)
    print(json.dumps(pins, indent=1, sort_keys=True)[:3000])


if __name__ == "__main__":
    main()
