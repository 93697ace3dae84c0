"""Frame time of the default scene with the general record kinds switched on by shape class (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pathtracer_amd as pt
cfg = pt.make_config(1920, 1080, 64)
for name, bits in (("all short kinds", 0), ("general spheres", 4), ("general planes", 1), ("general cubes", 2), ("general everything", 7)):
    sc = pt.Scene(*pt.default_scene(), general_kinds=bits)
    out = sc.render(cfg); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); sc.render(cfg, out=out); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("%-20s %.3f ms  mean %s" % (name, min(ts), [round(x, 5) for x in out.mean(dim=(0, 1)).tolist()]), flush=True)
