"""development aid: device time of the sliding-window features of one 720p frame (k_window_dcts against k_features_i16<2>)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tiler_amd import stages  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(1)
fb = torch.randint(0, 1 << 24, (720, 1280), generator=g, device="cuda", dtype=torch.int32)
fb[:, :640] = (torch.arange(640, device="cuda", dtype=torch.int32) // 5)[None, :] * 0x010101
for env in ("", "1"):
    if env:
        os.environ["TM_WINDOW_DCTS_BY_TILE"] = env
    else:
        os.environ.pop("TM_WINDOW_DCTS_BY_TILE", None)
    out = stages.window_dcts(fb)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        out = stages.window_dcts(fb)
    e1.record()
    torch.cuda.synchronize()
    print("by tile" if env else "by strips", "%.3f ms per frame" % (e0.elapsed_time(e1) / 10), int(out.sum().item()))
