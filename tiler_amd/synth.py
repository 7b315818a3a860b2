"""Seeded synthetic video of SURVEY.md section 8(d): smooth gradients + per-tile noise + scene cuts.

frame f, pixel (x, y):  R = (x*255//W + 2f) mod 256, G = (y*255//H + f) mod 256, B = ((x+y)*255//(W+H) + 3f) mod 256;
uniform noise in [-N, N] (N = 8) on a fraction rho (0.25) of the 8x8 tiles, chosen per frame; every `cut` frames the
channels rotate (R,G,B) -> (G,B,R) so keyframe detection has scene cuts to find.  Frames are RGB32 (0xAARRGGBB, the
AV_PIX_FMT_RGB32 layout TFFMPEGFrameCallback hands over, extern.pas:149).

An addition to SURVEY.md 8(d)'s generator, on by default (`freeze=True`): the `+f` drift is frozen on every third tile column, so
that a third of the picture is static between scene cuts and exact inter-frame duplicate tiles occur (a quarter of all frame tiles
on the 720p clip: the static columns' tiles that carry no noise).  `freeze=False` is the literal generator of 8(d), in which
every tile changes every frame; bench.py reports both.
"""
import numpy as np

SEED = 0x42381337  # echoes CRandomSeed, extern.pas:226


def frame(f, width, height, rng, noise=8, rho=0.25, cut=100, freeze=True):
    y, x = np.mgrid[0:height, 0:width].astype(np.int64)
    tx = x >> 3
    drift = np.where(tx % 3 == 0, 0, f) if freeze else f  # frozen columns of tiles: exact inter-frame duplicates
    r = (x * 255 // width + 2 * drift) % 256
    g = (y * 255 // height + drift) % 256
    b = ((x + y) * 255 // (width + height) + 3 * drift) % 256
    tw, th = (width + 7) // 8, (height + 7) // 8
    noisy = rng.random((th, tw)) < rho
    mask = np.repeat(np.repeat(noisy, 8, axis=0), 8, axis=1)[:height, :width]
    n = rng.integers(-noise, noise + 1, size=(3, height, width))
    r = np.clip(r + n[0] * mask, 0, 255)
    g = np.clip(g + n[1] * mask, 0, 255)
    b = np.clip(b + n[2] * mask, 0, 255)
    rot = (f // cut) % 3
    if rot == 1:
        r, g, b = g, b, r
    elif rot == 2:
        r, g, b = b, r, g
    return ((0xFF << 24) | (r << 16) | (g << 8) | b).astype(np.uint32)


def video(nframes, width, height, seed=SEED, **kw):
    """uint32 [nframes][height][width] RGB32"""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = np.empty((nframes, height, width), np.uint32)
    for f in range(nframes):
        out[f] = frame(f, width, height, rng, **kw)
    return out
