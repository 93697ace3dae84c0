"""Walk budget / shade gate / leaf gate sweep on the 10 000-primitive scene, 1080p@16 (development aid)."""
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
for walk, leaf_gates in (("fat", (4, 8, 16)),):
    sc.tune(pt.binding.PT_TUNE_BVH_WALK, 1 if walk == "skip" else 0)
    for leaf_gate in leaf_gates:
        sc.tune(pt.binding.PT_TUNE_LEAF_GATE, leaf_gate)
        for budget in (8, 12, 16, 24, 32):
            row = []
            for gate in (1, 8, 16, 24):
                sc.tune(pt.binding.PT_TUNE_WALK_BUDGET, budget).tune(pt.binding.PT_TUNE_SHADE_GATE, gate)
                row.append("%.2f" % timed())
            print(walk, "leaf gate", leaf_gate, "budget", budget, "shade gate 1/8/16/24:", " ".join(row), flush=True)
