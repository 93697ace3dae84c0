"""Where the hierarchy overtakes the loop over all primitives: small scattered shapes and grids of large glass spheres
(development aid)."""
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

cases = [("synthetic %d+%d" % (a, b), lambda a=a, b=b: pt.synthetic_scene(a, b, 12345), 5) for a, b in ((16, 16), (24, 24), (32, 32), (52, 52), (100, 100))]
cases += [("glass grid %dx%d depth 31" % (g, g), lambda g=g: pt.glass_grid_scene(g), 31) for g in (6, 8, 10, 12)]
for name, build, depth in cases:
    with_h = pt.Scene(*build(), bvh_min_prims=2)
    without = pt.Scene(*build(), bvh_min_prims=1000000)
    cfg = pt.make_config(1920, 1080, 16, n_bounces=depth)
    a, ta = timed(with_h, cfg)
    b, tb = timed(without, cfg)
    print("%-28s hierarchy (%5d nodes) %7.3f ms   loop %7.3f ms   equal %s" % (name, with_h.bvh_nodes, ta, tb, bool(torch.equal(a, b))), flush=True)
