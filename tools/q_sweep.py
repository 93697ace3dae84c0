"""Glass-queue sweep on the 1080p frame (development aid): time and bit-equality against the plain loop.
    python tools/q_sweep.py [scene]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pathtracer_amd as pt
from pathtracer_amd import binding as b

name = sys.argv[1] if len(sys.argv) > 1 else "default"
depth = 31 if name.startswith("glass") else 5
prims, n, mats, nm = pt.glass_grid_scene(int(name[5:] or 5)) if name.startswith("glass") else pt.default_scene()
sc = pt.Scene(prims, n, mats, nm)
cfg = pt.make_config(1920, 1080, 64, n_bounces=depth)
ref = sc.render(pt.make_config(1920, 1080, 64, n_bounces=depth, math_mode=pt.PT_MATH_FAST_SCALAR))

def timed():
    out = sc.render(cfg)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); sc.render(cfg, out=out); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts), bool(torch.equal(out, ref))

sc.tune(b.PT_TUNE_GLASS_QUEUE, -1)
print("plain loop           %.3f ms equal %s" % timed(), flush=True)
sc.tune(b.PT_TUNE_GLASS_QUEUE, 0)
print("automatic            %.3f ms equal %s" % timed(), flush=True)
for cap in (16, 24, 32, 40, 48):
    for gate in sorted({cap // 2, (3 * cap) // 4, (7 * cap) // 8}):
        sc.tune(b.PT_TUNE_GLASS_QUEUE, cap).tune(b.PT_TUNE_GLASS_GATE, gate)
        print("cap %2d gate %2d       %.3f ms equal %s" % ((cap, gate) + timed()), flush=True)
