"""Runs the reference-defaults configuration (motion prediction + extended palette usage) a few times on the bench clip and prints the
stage times of each run (development aid: run-to-run spread of the k-nearest passes)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tiler_amd import synth  # noqa: E402
from tiler_amd.encoder import TilingEncoder  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
radius = int(sys.argv[2]) if len(sys.argv) > 2 else 32
F, W, H = 300, 1280, 720
frames = torch.from_numpy(synth.video(F, W, H, freeze=os.environ.get("TM_PROBE_LITERAL") != "1").view(np.int32)).cuda()  # TM_PROBE_LITERAL=1: the generator as written
enc = TilingEncoder()
enc.LoadDefaultSettings()
enc.PaletteCount = 16
enc.MotionPredictRadius = radius
enc.FrameTilingExtendedPaletteUsage = (int(sys.argv[3]) != 0) if len(sys.argv) > 3 else True
enc.SetVideo(W, H, 24.0, F)
enc.SetFramesDevice(frames)
for r in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    enc.Run()
    torch.cuda.synchronize()
    print("run %d: %.1f ms, stages %s" % (r, (time.perf_counter() - t0) * 1e3, [round(float(v), 1) for v in enc.StageMs()]), flush=True)
