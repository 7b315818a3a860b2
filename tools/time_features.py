"""development aid: device time of the int16 features of a 720p x 300 clip's tiles (k_features_tiles8 against k_features_i16<0>)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from tiler_amd import stages  # noqa: E402
from test_gpu_fullsize import device_video  # noqa: E402

nframes = int(os.environ.get("TM_TIME_FRAMES", "300"))
frames = device_video(1280, 720, nframes, freeze=False)
tiles_all, flags_all, lab = stages.load(frames, 160, 90)
for ntiles in (14400, 144000, tiles_all.shape[0]):
  tiles, flags = tiles_all[:ntiles].contiguous(), flags_all[:ntiles].contiguous()
  for env in ("", "1"):
      if env:
          os.environ["TM_FEATURES_BY_TILE"] = env
      else:
          os.environ.pop("TM_FEATURES_BY_TILE", None)
      for fl in (None, flags):
          out = stages.features_rgb(tiles, fl, 1, False)
          e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
          e0.record()
          for _ in range(20):
              out = stages.features_rgb(tiles, fl, 1, False)
          e1.record()
          torch.cuda.synchronize()
          print(ntiles, "by tile " if env else "8 a wave", "flags" if fl is not None else "no flags", "%.3f ms" % (e0.elapsed_time(e1) / 20), int(out.to(torch.int64).sum().item()))
