"""Upload time and hierarchy-vs-loop equality on large synthetic scenes (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pathtracer_amd as pt

for n in (20000, 100000, 400000):
    t0 = time.time()
    made = pt.synthetic_scene(n // 2, n // 2, 99)
    t1 = time.time()
    sc = pt.Scene(*made)
    t2 = time.time()
    cfg = pt.make_config(96, 54, 2)
    a = sc.render(cfg); torch.cuda.synchronize()
    t3 = time.time()
    b = sc.render(pt.make_config(96, 54, 2, math_mode=pt.PT_MATH_FAST_SCALAR)); torch.cuda.synchronize()
    t4 = time.time()
    big = pt.make_config(1920, 1080, 4)
    out = sc.render(big); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); sc.render(big, out=out); e1.record(); torch.cuda.synchronize()
    print("%7d primitives: scene made in %.2f s, uploaded (hierarchy built) in %.2f s, %d nodes; 96x54@2 through the hierarchy %.3f s, "
          "through the loop %.3f s, equal %s; 1080p@4 %.1f ms" % (n, t1 - t0, t2 - t1, sc.bvh_nodes, t3 - t2, t4 - t3, bool(torch.equal(a, b)), e0.elapsed_time(e1)), flush=True)
    sc.close()
