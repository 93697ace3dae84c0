"""Times each of N contiguous row tiles (and N interleaved row-block sets) of the 1080p frame on one GPU: how
unevenly would N GPUs be loaded?  (development aid)"""
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
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); sc.render(cfg, out=out); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts)

full = time_cfg(pt.make_config(W, H, SPP))
print("full frame %.3f ms" % full)
for n in (2, 4, 8):
    rows = H // n
    ts = [time_cfg(pt.make_config(W, H, SPP, row_begin=r * rows, row_end=(r + 1) * rows)) for r in range(n)]
    print("N=%d contiguous tiles: %s  max %.3f  ideal %.3f  efficiency %.1f %%" % (n, ["%.3f" % t for t in ts], max(ts), full / n, 100 * full / n / max(ts)))
    ts = [time_cfg(pt.make_config(W, H, SPP, row_begin=r, row_step=n, fb_compact=1)) for r in range(n)]
    print("N=%d interleaved rows:  %s  max %.3f  ideal %.3f  efficiency %.1f %%" % (n, ["%.3f" % t for t in ts], max(ts), full / n, 100 * full / n / max(ts)))
