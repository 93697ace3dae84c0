"""Parity-order render kernel: the loop with camera batches against the plain loop on the 1080p frame (development aid).
    python tools/parity_ab.py [spp]
Prints both kernel times and whether the two frames are equal bit for bit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pathtracer_amd as pt
from pathtracer_amd import binding as b
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for name, data, depth in (("default", pt.default_scene(), 5), ("glass grid 5x5 depth 31", pt.glass_grid_scene(5), 31)):
    sc = pt.Scene(*data)
    cfg = pt.make_config(1920, 1080, spp, n_bounces=depth, math_mode=pt.PT_MATH_PARITY)
    imgs = {}
    for how, q in (("plain loop", -1), ("camera batches + queue", 0)):
        sc.tune(b.PT_TUNE_GLASS_QUEUE, q)
        fb = torch.zeros((1080, 1920, 3), device="cuda")
        ts = []
        for rep in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); sc.render(cfg, out=fb); e1.record(); torch.cuda.synchronize()
            if rep: ts.append(e0.elapsed_time(e1))
        imgs[how] = fb
        print("%-26s %-24s min %.3f median %.3f ms" % (name, how, min(ts), sorted(ts)[len(ts) // 2]), flush=True)
    print("%-26s bitwise equal: %s" % (name, bool(torch.equal(imgs["plain loop"], imgs["camera batches + queue"]))), flush=True)
