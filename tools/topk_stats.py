"""development aid: how the k = 64 nearest rows of the bench clip's queries lie -- the ratios that decide how a first threshold should be estimated"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tiler_amd import stages, synth  # noqa: E402
from tiler_amd.encoder import TilingEncoder  # noqa: E402

F, W, H = 300, 1280, 720
frames = torch.from_numpy(synth.video(F, W, H, freeze=os.environ.get("TM_PROBE_LITERAL") != "1").view(np.int32)).cuda()
enc = TilingEncoder()
enc.LoadDefaultSettings()
enc.PaletteCount = 16
enc.MotionPredictRadius = 0
enc.FrameTilingExtendedPaletteUsage = False
enc.SetVideo(W, H, 24.0, F)
enc.SetFramesDevice(frames)
enc.Run()
hdr, pal_px, rgb = enc.Tiles()
pals = enc.Palettes()
pal_idx = hdr["PalIdx_Initial"].astype(np.int32)
T = pal_px.shape[0]
db = stages.features_pal(torch.from_numpy(pal_px).cuda(), torch.from_numpy(pal_idx).cuda(), torch.from_numpy(pals).cuda(), 1)
dbd = db.to(torch.float64)
dn = (dbd * dbd).sum(1)
gen = torch.Generator(device="cuda").manual_seed(7)
qs = []
for f in torch.randint(0, F, (4,), generator=gen, device="cuda").tolist():
    ft, _, _ = stages.load(frames[f:f + 1], 160, 90)
    qf = stages.features_rgb(ft, None, 1, False)
    qs.append(qf[torch.randperm(14400, generator=gen, device="cuda")[:512]])
q = torch.cat(qs).to(torch.float64)
print("T", T, "queries", q.shape[0])
sub16 = torch.randperm(T, generator=gen, device="cuda")[: T // 16]
res = {k: [] for k in ("d1", "d64", "d512", "c2", "est8", "cnt8", "est16", "cnt16", "cnt4")}
for s0 in range(0, q.shape[0], 256):
    qq = q[s0:s0 + 256]
    d = (qq * qq).sum(1)[:, None] + dn[None, :] - 2.0 * (qq @ dbd.T)
    srt = d.sort(1).values
    res["d1"].append(srt[:, 0]); res["d64"].append(srt[:, 63]); res["d512"].append(srt[:, 511])
    res["c2"].append((d <= 2.0 * srt[:, 63:64]).sum(1))
    ds = d[:, sub16].sort(1).values
    for kk in (4, 8, 16):
        est = ds[:, kk - 1]
        res["cnt%d" % kk].append((d <= est[:, None]).sum(1))
r = {k: torch.cat(v).double().cpu().numpy() for k, v in res.items() if v}
pc = lambda x: np.percentile(x, [5, 25, 50, 75, 95]).round(3)
print("d1/d64          ", pc(r["d1"] / r["d64"]))
print("d512/d64        ", pc(r["d512"] / r["d64"]))
print("count(<= 2 d64) ", pc(r["c2"]))
for kk in (4, 8, 16):
    c = r["cnt%d" % kk]
    print("sample 1/16, %2d-th: count in full" % kk, pc(c), " <64: %.3f  >512: %.3f  >1024: %.3f" % ((c < 64).mean(), (c > 512).mean(), (c > 1024).mean()))
