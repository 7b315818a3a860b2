"""development aid: one small exact-KNN call through the stage seam against a torch brute force (stderr visible, no pytest capture)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tiler_amd import stages
nq, nt, spread = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (70, 100, 90)
rng = np.random.default_rng(nq * 1000 + nt)
def feats(n):
    f = rng.integers(-spread, spread + 1, size=(n, 192)).astype(np.int32)
    f[:, 0] = rng.integers(0, 13216, size=n); f[:, 64] = rng.integers(-6500, 6501, size=n); f[:, 128] = rng.integers(-9000, 9001, size=n)
    f[:, 1:6] = rng.integers(-3000, 3001, size=(n, 5))
    return f.astype(np.int16)
db, q = feats(nt), feats(nq)
idx, err = stages.knn(torch.from_numpy(q).cuda(), torch.from_numpy(db).cuda())
d = ((q.astype(np.int64)[:, None, :] - db.astype(np.int64)[None, :, :]) ** 2).sum(2)
print("ok", np.array_equal(d.argmin(1), idx.cpu().numpy()), np.array_equal(d.min(1), err.cpu().numpy().view(np.uint32).astype(np.int64)), flush=True)
