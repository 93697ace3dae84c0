"""Walk budget / shade gate sweep on the 10 000-primitive scene (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pathtracer_amd as pt

sc = pt.Scene(*pt.synthetic_scene(5000, 4989, 12345))
cfg = pt.make_config(1920, 1080, 16)
out = sc.render(cfg); torch.cuda.synchronize()
def timed():
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); sc.render(cfg, out=out); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts)
for walk in ("fat", "skip"):
    sc.tune(pt.binding.PT_TUNE_BVH_WALK, 1 if walk == "skip" else 0)
    for budget in (4, 8, 16, 32, 64, 1000000):
        row = []
        for gate in (1, 8, 16, 32, 64):
            sc.tune(pt.binding.PT_TUNE_WALK_BUDGET, budget).tune(pt.binding.PT_TUNE_SHADE_GATE, gate)
            row.append("%.2f" % timed())
        print(walk, "budget", budget, "gate 1/8/16/32/64:", " ".join(row), flush=True)
