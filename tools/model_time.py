"""Frame time of the headline frame with the second shading model and with the plain loop (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pathtracer_amd as pt
from pathtracer_amd import binding as b
def t(sc, cfg):
    out = sc.render(cfg); torch.cuda.synchronize(); ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); sc.render(cfg, out=out); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts)
sc = pt.Scene(*pt.default_scene())
print("bsdf.h model, batched loop      %.3f ms" % t(sc, pt.make_config(1920, 1080, 64)))
print("class-based model (plain loop)  %.3f ms" % t(sc, pt.make_config(1920, 1080, 64, bsdf_model=pt.PT_BSDF_CLASSES)))
sc.tune(b.PT_TUNE_GLASS_QUEUE, -1)
print("bsdf.h model, plain loop        %.3f ms" % t(sc, pt.make_config(1920, 1080, 64)))
print("run table in scene order        %.3f ms" % t(sc, pt.make_config(1920, 1080, 64, math_mode=pt.PT_MATH_FAST_SCALAR)))
