"""Hierarchy split rule sweep on scenes with a hierarchy, 1080p@16 (development aid): time per frame and whether the
image equals the median-split one bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pathtracer_amd as pt

def timed(sc, cfg, out):
    ts = []
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); sc.render(cfg, out=out); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts)

SCENES = {"synthetic 10000": lambda: pt.synthetic_scene(5000, 4989, 12345),
          "synthetic 1000": lambda: pt.synthetic_scene(500, 489, 7),
          "glass grid 10x10 depth 31": lambda: pt.glass_grid_scene(10)}
for name, make in SCENES.items():
    cfg = pt.make_config(1920, 1080, 16, n_bounces=31 if "glass" in name else 5)
    ref = None
    for split in (1, 2, 3, 4, 5):
        sc = pt.Scene(*make(), bvh_min_prims=2, bvh_split=split)
        out = sc.render(cfg); torch.cuda.synchronize()
        t = timed(sc, cfg, out)
        if ref is None: ref = out.clone()
        print("%-28s split %d: %.2f ms  bitwise equal to median: %s" % (name, split, t, bool(torch.equal(out, ref))), flush=True)
        sc.close()
