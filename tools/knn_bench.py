"""KNN kernel micro-benchmark on synthetic-video features (not the judged bench; used for A/B of kernel variants)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tiler_amd import stages, synth  # noqa: E402

nframes = int(sys.argv[1]) if len(sys.argv) > 1 else 30
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 320705
W, H = 1280, 720
v = synth.video(nframes, W, H)
fr = torch.from_numpy(v.view(np.int32)).cuda()
tiles, flags, lab = stages.load(fr, W // 8, H // 8)
q = stages.features_rgb(tiles, None, 1, False)
nt = min(nt, q.shape[0])
perm = torch.randperm(q.shape[0], device="cuda")[:nt]
db = q[perm].contiguous()
ix = stages.KnnIndex(db)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.time()
    idx, err = ix.search(q)
    torch.cuda.synchronize()
    dt = time.time() - t0
    ms, kb, pairs = ix.last_stats()
    print(f"variant={os.environ.get('TM_KNN_VARIANT','-')} nq={q.shape[0]} nt={nt} K={kb} kernel_ms={ms:.2f} wall_ms={dt*1e3:.2f} "
          f"alg_TOPS={pairs*384/ms/1e9:.1f} mfma_TOPS={pairs*2*kb/ms/1e9:.1f}", flush=True)
# self-check: queries that are database rows must come back with err 0
print("zero-err fraction among planted:", float((err[perm].cpu() == 0).float().mean()))
