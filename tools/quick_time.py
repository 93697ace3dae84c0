"""Quick timing of pt_render on the GPU box (development aid).
    python tools/quick_time.py [scene] [width height spp depth]   scene: default | glassN | synthetic"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pathtracer_amd as pt

def get_scene(name):
    if name == "default":
        return pt.default_scene()
    if name.startswith("glass"):
        return pt.glass_grid_scene(int(name[5:] or 5))
    if name == "synthetic":
        return pt.synthetic_scene(5000, 4989, 12345)
    raise SystemExit("unknown scene")

def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "default"
    w, h, spp, depth = (int(x) for x in (sys.argv[2:6] if len(sys.argv) > 5 else (1920, 1080, 64, 5)))
    modes = sys.argv[6].split(",") if len(sys.argv) > 6 else ["parity", "fast"]
    prims, n, mats, nm = get_scene(name)
    sc = pt.Scene(prims, n, mats, nm)
    for mname in modes:
        mm = {"parity": pt.PT_MATH_PARITY, "fast": pt.PT_MATH_FAST, "scalar": pt.PT_MATH_FAST_SCALAR, "lds": pt.PT_MATH_FAST_LDS}[mname]
        cfg = pt.make_config(w, h, spp, n_bounces=depth, math_mode=mm)
        out = sc.render(cfg)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); sc.render(cfg, out=out); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = min(ts)
        print("%s %s (%d prims) %dx%d@%d depth %d: min %.3f ms median %.3f ms -> %.1f Msamples/s, mean %s" % (
            name, mname, n, w, h, spp, depth, ms, sorted(ts)[2], w * h * spp / ms / 1e3, [round(x, 5) for x in out.mean(dim=(0, 1)).tolist()]), flush=True)

main()
