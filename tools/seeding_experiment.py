"""Quality of the k-means seeding rule: the build's farthest-first initialisation against a k-means++-style
D^2-weighted seeding (own seeded RNG), on the bench clip, through the stage seam -- same Lloyd iterations, same everything downstream.
Reports LogPSNR's mean "PSNR-HVS by tile" (tilingencoder.pas:1006-1028) and the final tile count after Reindex for both.

    python tools/seeding_experiment.py [frames] [palettes] > gpurun_out/seeding.json
"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tiler_amd import lib, stages, synth  # noqa: E402
from tiler_amd._lib import check  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 300
P = int(sys.argv[2]) if len(sys.argv) > 2 else 16
W, H, S = 1280, 720, 16


def kmeanspp(pts, weights, k, rng):
    """D^2 sampling weighted by the points' weights; pts int32 [n][d] on the GPU; returns k point indices"""
    x = pts.to(torch.float64)
    w = weights.to(torch.float64)
    first = int(np.searchsorted(np.cumsum(w.cpu().numpy()), rng.random() * float(w.sum())))
    idx = [min(first, x.shape[0] - 1)]
    mind = ((x - x[idx[0]]) ** 2).sum(1)
    for _ in range(1, k):
        prob = (mind * w).cpu().numpy()
        tot = prob.sum()
        if tot <= 0:
            break
        i = int(np.searchsorted(np.cumsum(prob), rng.random() * tot))
        i = min(i, x.shape[0] - 1)
        idx.append(i)
        mind = torch.minimum(mind, ((x - x[i]) ** 2).sum(1))
    return idx


def muldiv(a, b, c):
    p = a * b
    q, cc = abs(p), abs(c)
    r = (q + cc // 2) // cc
    return -r if (p < 0) != (c < 0) else r


def hsv(r, g, b):  # RGBToHSV, utils.pas:278-325
    mx, mn = max(r, g, b), min(r, g, b)
    h = s = 0
    if mx != mn:
        d = mx - mn
        s = muldiv(d, 255, mx)
        if r == mx:
            h = muldiv(42, g - b, d)
        elif g == mx:
            h = muldiv(42, b - r, d) + 84
        else:
            h = muldiv(42, r - g, d) + 168
        h = int(np.fmod(h, 252))
    return h & 255, s & 255, mx & 255


def palette_from_centroids(cent, kk):
    items = []
    for i in range(kk):
        r, g, b = [int(min(255, max(0, np.rint(v)))) for v in cent[i]]
        h, s, v = hsv(r, g, b)
        items.append((v, s, h, r, g, b, i))
    items.sort()
    out = np.full(S, -65281, np.int32)  # cDitheringNullColor
    for i, it in enumerate(items):
        out[i] = (it[5] << 16) | (it[4] << 8) | it[3]
    return out


def run(policy, tiles, flags, gtiles, gflags, guse, rng):
    feat = stages.features_cluster(gtiles, 4)
    tiles_pp = policy.startswith("k-means++") or "tiles:k-means++" in policy
    colours_pp = policy.startswith("k-means++") or "colours:k-means++" in policy
    if not tiles_pp:
        pal_idx = stages.palettize(feat, guse, P)
    else:
        kk, assign, _, _ = stages.kmeans_seeded(feat, guse, P, kmeanspp(feat, guse, P, rng))
        cnt = torch.bincount(assign.long(), minlength=P).cpu().numpy()
        order = np.argsort(-cnt, kind="stable")
        lut = np.empty(P, np.int64)
        lut[order] = np.arange(P)
        pal_idx = torch.from_numpy(lut).cuda()[assign.long()].to(torch.int32)
    if not colours_pp:
        palettes = stages.quantize_palettes(gtiles, pal_idx, P, S).cpu().numpy()
    else:
        palettes = np.full((P, S), -65281, np.int32)
        for p in range(P):
            px = gtiles[pal_idx == p].reshape(-1)
            if px.numel() == 0:
                continue
            r, g, b = px & 255, (px >> 8) & 255, (px >> 16) & 255
            key = (g.long() << 16) | (r.long() << 8) | b.long()
            uk, uc = torch.unique(key, return_counts=True)  # sorted by (G, R, B): CompareDSPixel
            pts = torch.stack([(uk >> 8) & 255, (uk >> 16) & 255, uk & 255], 1).to(torch.int32).contiguous()
            wts = uc.to(torch.int32).contiguous()
            k3, _, cent, _ = stages.kmeans_seeded(pts, wts, S, kmeanspp(pts, wts, S, rng))
            palettes[p] = palette_from_centroids(cent.cpu().numpy(), k3)
    palettes = np.ascontiguousarray(palettes, np.int32)
    sweeps = ctypes.c_int()
    lib().tm_optimize_palettes_host.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    check(lib().tm_optimize_palettes_host(palettes.ctypes.data_as(ctypes.c_void_p), P, S, ctypes.byref(sweeps)))
    dpal = torch.from_numpy(palettes).cuda()
    pal_px = stages.dither(gtiles, gflags, pal_idx, dpal, True)
    db = stages.features_pal(pal_px, pal_idx, dpal, 1)
    psnr_sum, n, hist = 0.0, 0, torch.zeros(gtiles.shape[0], dtype=torch.int64, device="cuda")
    chunk = 14400 * 50
    for a in range(0, tiles.shape[0], chunk):
        qf = stages.features_rgb(tiles[a:a + chunk].contiguous(), None, 1, False)
        idx, err = stages.knn(qf, db)
        e = err.to(torch.int64) & 0xFFFFFFFF
        r = (e.to(torch.float64) / 192.0).to(torch.float32)
        psnr = (10.0 * torch.log10(255.0 * 255.0 / torch.clamp(r, min=0.5).to(torch.float64))).to(torch.float32)
        psnr_sum += float(psnr.to(torch.float64).sum())
        n += psnr.numel()
        hist += torch.bincount(idx.long(), minlength=gtiles.shape[0])
    nu2, _, _, _ = stages.dedup(pal_px, hist.to(torch.int32))
    return {"policy": policy, "mean_psnr_hvs_by_tile": psnr_sum / n, "final_tiles_after_reindex": int(nu2),
            "tiles_per_palette": torch.bincount(pal_idx.long(), minlength=P).cpu().tolist()}


def main():
    assert torch.cuda.is_available()
    frames = torch.from_numpy(synth.video(F, W, H).view(np.int32)).cuda()
    tm_w, tm_h = W // 8, H // 8
    tiles, flags, _ = stages.load(frames, tm_w, tm_h)
    del frames
    nu, remap, order, use = stages.dedup(tiles)
    q = tiles.shape[0]
    target = min(int(round(7.0 * round(np.sqrt(np.float32(q)) * np.log2(1 + np.float32(q))))), q)
    T = min(nu, target)
    sel = order[:T].long()
    gtiles, gflags, guse = tiles[sel].contiguous(), flags[sel].contiguous(), use[:T].contiguous()
    out = {"clip": f"{W}x{H} x {F} synthetic (tiler_amd.synth), {P} palettes x {S} colours", "global_tiles_T": int(T), "runs": []}
    out["runs"].append(run("farthest-first", tiles, flags, gtiles, gflags, guse, None))
    only = os.environ.get("TM_SEEDING_ONLY")  # "tiles": just the mixed policy the build could adopt, over more seeds
    if only == "tiles":
        for seed in range(1, 11):
            out["runs"].append(run("mixed tiles:k-means++ colours:farthest-first (seed %d)" % seed, tiles, flags, gtiles, gflags, guse, np.random.Generator(np.random.PCG64(seed))))
        print(json.dumps(out, indent=1))
        return
    for seed in (1, 2, 3):
        r = run("k-means++ (D^2 sampling, PCG64 seed %d)" % seed, tiles, flags, gtiles, gflags, guse, np.random.Generator(np.random.PCG64(seed)))
        out["runs"].append(r)
    for seed in (1, 2):
        out["runs"].append(run("mixed tiles:farthest-first colours:k-means++ (seed %d)" % seed, tiles, flags, gtiles, gflags, guse, np.random.Generator(np.random.PCG64(seed))))
        out["runs"].append(run("mixed tiles:k-means++ colours:farthest-first (seed %d)" % seed, tiles, flags, gtiles, gflags, guse, np.random.Generator(np.random.PCG64(seed))))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
