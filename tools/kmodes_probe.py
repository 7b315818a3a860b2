"""development aid: the k-modes operator at config 5's shape (1 618 022 rows x 80 B, 64 clusters): init + first iteration, and ms per later iteration"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tiler_amd import stages
gk = torch.Generator(device="cuda").manual_seed(5)
nk, kk = int(sys.argv[1]) if len(sys.argv) > 1 else 1618022, 64
proto = torch.randint(0, 48, (40, 80), generator=gk, device="cuda", dtype=torch.uint8)
rows = proto[torch.randint(0, 40, (nk,), generator=gk, device="cuda")]
noise = torch.rand((nk, 80), generator=gk, device="cuda") < 0.2
rows = torch.where(noise, torch.randint(0, 48, (nk, 80), generator=gk, device="cuda", dtype=torch.uint8), rows).contiguous()
stages.kmodes_dev(rows, kk, 0, 48, 1)
torch.cuda.synchronize()
t = time.perf_counter(); _, _, c1, _, p1 = stages.kmodes_dev(rows, kk, 0, 48, 1); torch.cuda.synchronize(); d1 = time.perf_counter() - t
t = time.perf_counter(); _, _, c5, _, p5 = stages.kmodes_dev(rows, kk, 0, 48, 5); torch.cuda.synchronize(); d5 = time.perf_counter() - t
print("kmodes n=%d: init + first iteration %.1f ms, %.2f ms per later iteration (cost %d -> %d)" % (nk, d1 * 1e3, (d5 - d1) / max(1, (p5 - p1) // nk) * 1e3, c1, c5), flush=True)
