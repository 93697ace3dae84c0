"""Hierarchy vs the loop over all primitives on the synthetic scenes: image difference and time.  (development aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pathtracer_amd as pt

def timed(sc, cfg, reps=3):
    out = sc.render(cfg); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); sc.render(cfg, out=out); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return out, min(ts)

for (ns, nc, w, h, spp) in ((500, 489, 320, 180, 16), (5000, 4989, 480, 270, 4), (5000, 4989, 1920, 1080, 4)):
    sc = pt.Scene(*pt.synthetic_scene(ns, nc, 12345))
    a, ta = timed(sc, pt.make_config(w, h, spp, math_mode=pt.PT_MATH_FAST))
    b, tb = timed(sc, pt.make_config(w, h, spp, math_mode=pt.PT_MATH_FAST_SCALAR), reps=1)
    diff = (a != b).any(dim=2)
    print("%d prims, %d nodes, %dx%d@%d: hierarchy %.3f ms, loop %.3f ms (x%.1f); pixels that differ: %d of %d, max abs %.3g"
          % (11 + ns + nc, sc.bvh_nodes, w, h, spp, ta, tb, tb / ta, int(diff.sum()), diff.numel(), float((a - b).abs().max())), flush=True)
sc = pt.Scene(*pt.synthetic_scene(5000, 4989, 12345))
_, t = timed(sc, pt.make_config(1920, 1080, 64))
print("10 000 prims 1920x1080@64 hierarchy: %.2f ms -> %.1f Msamples/s" % (t, 1920 * 1080 * 64 / t / 1e3))
