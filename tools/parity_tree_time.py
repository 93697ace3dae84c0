"""Parity-order render of the 10 000-primitive scene through the hierarchy (development aid): time per frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pathtracer_amd as pt
sc = pt.Scene(*pt.synthetic_scene(5000, 4989, 12345))
for w, h, spp in ((480, 270, 4), (1920, 1080, 4), (1920, 1080, 64)):
    cfg = pt.make_config(w, h, spp, math_mode=pt.PT_MATH_PARITY)
    t0 = time.time(); out = sc.render(cfg); torch.cuda.synchronize(); t1 = time.time()
    print("%dx%d@%d parity order through the hierarchy: %.3f s, mean %.5f" % (w, h, spp, t1 - t0, float(out.mean())), flush=True)
