"""Sample rate of the headline frame against samples per pixel and tile size (development aid): what the end of the
launch and each wave's start and drain cost."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pathtracer_amd as pt
sc = pt.Scene(*pt.default_scene())
for spp in (16, 32, 64, 128, 256, 1024):
    row = []
    for tl in (4, 5, 6):
        cfg = pt.make_config(1920, 1080, spp, tile_log2=tl)
        out = sc.render(cfg); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); sc.render(cfg, out=out); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        t = min(ts)
        row.append("%d px/wave %.3f ms %.1f Gs/s" % (1 << tl, t, 1920 * 1080 * spp / t / 1e6))
    print("1080p @ %4d spp: %s" % (spp, " | ".join(row)), flush=True)
