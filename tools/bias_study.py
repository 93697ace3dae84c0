"""Seed-averaged difference between the fast and the parity-order kernel, with its standard error (development aid).
    python tools/bias_study.py [scene=synthetic|default|glass5] [seeds=512] [depth=5] [w=128] [h=72] [spp=16]
Fast and parity-order kernels trace the same random streams at a seed, so the per-seed difference of the frame means is
a paired sample: its mean over seeds is the bias of the fast arithmetic, its spread over seeds gives the standard
error.  Prints the relative bias per channel with the error, overall and split by what the pixel's centre ray hits
first (plane / sphere / cube ...)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import pathtracer_amd as pt
opt = dict(a.split("=") for a in sys.argv[1:])
name = opt.get("scene", "synthetic"); seeds = int(opt.get("seeds", 512)); depth = int(opt.get("depth", 5))
W, H, SPP = int(opt.get("w", 128)), int(opt.get("h", 72)), int(opt.get("spp", 16))
data = pt.synthetic_scene(5000, 4989, 12345) if name == "synthetic" else (pt.default_scene() if name == "default" else pt.glass_grid_scene(int(name[5:])))
sc = pt.Scene(*data, general_kinds=int(opt.get("general", 0)))
# what the centre ray of each pixel hits first (parity order): type of the primitive
cfg0 = pt.make_config(W, H, 1)
ii, jj = np.meshgrid(np.arange(W), np.arange(H))
vh = 5.0; vw = vh * W / H; f = cfg0.focal_length
px = (-vw / 2 + (ii + 0.5) * vw / W); py = (vh / 2 - (H - jj + 0.5) * vh / H)
d = np.stack([px, py, np.full_like(px, -f)], axis=-1).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
o = np.zeros_like(d)
hit = sc.nearest(torch.tensor(o.reshape(-1, 3)).cuda(), torch.tensor(d.reshape(-1, 3)).cuda())
prim = hit["prim"].cpu().numpy().reshape(H, W)
types = np.array([p.type for p in data[0][: data[1]]])
cls = np.where(prim >= 0, types[np.maximum(prim, 0)], -1)
names = {-1: "miss", 0: "plane", 1: "cube", 2: "cylinder", 3: "sphere"}
diff_means = {k: [] for k in ["all"] + sorted(set(cls.ravel().tolist()))}
par_means = {k: 0.0 for k in diff_means}
for seed in range(seeds):
    fa = sc.render(pt.make_config(W, H, SPP, n_bounces=depth, seed=seed, math_mode=pt.PT_MATH_FAST)).double().cpu().numpy()
    pa = sc.render(pt.make_config(W, H, SPP, n_bounces=depth, seed=seed, math_mode=pt.PT_MATH_PARITY)).double().cpu().numpy()
    for k in diff_means:
        m = np.ones((H, W), bool) if k == "all" else (cls == k)
        diff_means[k].append((fa - pa)[m].mean(axis=0))
        par_means[k] += pa[m].mean(axis=0) / seeds
    if seed % 64 == 63: print("seed", seed + 1, flush=True)
print("%s, %dx%d @ %d spp, depth %d, %d seeds: (fast - parity) / parity per channel, +- one standard error" % (name, W, H, SPP, depth, seeds))
for k, v in diff_means.items():
    v = np.stack(v)
    share = 1.0 if k == "all" else float((cls == k).mean())
    bias = v.mean(axis=0) / par_means[k]; err = v.std(axis=0, ddof=1) / np.sqrt(seeds) / par_means[k]
    print("  %-8s (%5.1f %% of pixels): %s" % (k if k == "all" else names[k], 100 * share, "  ".join("%+.4f %% +- %.4f" % (100 * b, 100 * e) for b, e in zip(bias, err))))
