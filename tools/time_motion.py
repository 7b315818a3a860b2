"""development aid: one 720p frame's motion search at the stage seam (int16 window features -> k_mo_pack_win -> k_mo_search_mfma), timed; also the
target of tools/pmc_script.sh for the search kernel's counters"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from tiler_amd import stages  # noqa: E402
from test_gpu_fullsize import device_video  # noqa: E402

frames = device_video(1280, 720, 2, freeze=False)
tiles, flags, _ = stages.load(frames, 160, 90)
cur = stages.features_rgb(tiles[14400:].contiguous(), flags[14400:].contiguous(), 1, False)
win = stages.window_dcts((frames[0] & 0xFFFFFF).contiguous())
reps = int(os.environ.get("TM_TIME_REPS", "10"))
out = stages.motion_search(cur, 160, 90, win, 32)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    out = stages.motion_search(cur, 160, 90, win, 32)
e1.record()
torch.cuda.synchronize()
print("motion search (pack + search) %.3f ms per frame" % (e0.elapsed_time(e1) / reps), int(out[0].to(torch.int64).sum().item()))
