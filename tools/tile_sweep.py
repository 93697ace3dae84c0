"""Times rank 0's interleaved row set of the 1080p frame for N = 1..16 GPUs at every pixels-per-wave setting
(PT_TILE_LOG2): which tile size should pt_render pick for a launch of a given size?  (development aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pathtracer_amd as pt

W, H, SPP = 1920, 1080, 64
sc = pt.Scene(*pt.default_scene())
out = torch.zeros((H, W, 3), device="cuda")

def time_cfg(cfg):
    sc.render(cfg, out=out); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); sc.render(cfg, out=out); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts)

for n in (1, 2, 3, 4, 6, 8, 16):
    cfg = pt.make_config(W, H, SPP, row_begin=0, row_step=n, fb_compact=1)
    line = []
    for lg in (6, 5, 4, 3):
        sc.tune(pt.binding.PT_TUNE_TILE_LOG2, lg)
        line.append("P=%d %.3f" % (1 << lg, time_cfg(cfg)))
    print("N=%2d (%7d px): %s" % (n, len(range(0, H, n)) * W, "  ".join(line)), flush=True)
