"""Quick timing of pt_render on the GPU box (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pathtracer_amd as pt

def main():
    w, h, spp = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (1920, 1080, 64)))
    prims, n, mats, nm = pt.default_scene()
    sc = pt.Scene(prims, n, mats, nm)
    for name, mm in (("parity", pt.PT_MATH_PARITY), ("fast", pt.PT_MATH_FAST)):
        cfg = pt.make_config(w, h, spp, math_mode=mm)
        out = sc.render(cfg)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); sc.render(cfg, out=out); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = min(ts)
        print("%s %dx%d@%d: min %.3f ms median %.3f ms -> %.1f Msamples/s, mean %s" % (name, w, h, spp, ms, sorted(ts)[2], w*h*spp/ms/1e3, out.mean(dim=(0,1)).tolist()), flush=True)

main()
