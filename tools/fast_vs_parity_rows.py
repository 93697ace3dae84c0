"""Fast against parity-order kernel on every `step`-th row of a frame (development aid): mean radiance per channel,
RMSE, pixels off by more than 0.05.   python3 tools/fast_vs_parity_rows.py synthetic 1920 1080 64 5 40 [general_kinds bits]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pathtracer_amd as pt
name = sys.argv[1]
w, h, spp, depth, step = (int(x) for x in sys.argv[2:7])
scene = pt.default_scene() if name == "default" else (pt.glass_grid_scene(int(name[5:])) if name.startswith("glass") else pt.synthetic_scene(5000, 4989, 12345))
general = int(sys.argv[7]) if len(sys.argv) > 7 else 0
sc = pt.Scene(*scene, general_kinds=general)
n_rows = len(range(step // 2, h, step))
imgs = {}
for mode, mm in (("parity", pt.PT_MATH_PARITY), ("fast", pt.PT_MATH_FAST)):
    cfg = pt.make_config(w, h, spp, n_bounces=depth, math_mode=mm, row_begin=step // 2, row_end=h, row_step=step, fb_compact=1)
    out = torch.zeros((n_rows, w, 3), device="cuda")
    sc.render(cfg, out=out); torch.cuda.synchronize()
    imgs[mode] = out.double()
    print(mode, "mean", [round(x, 6) for x in out.double().mean(dim=(0, 1)).tolist()], flush=True)
a, b = imgs["parity"], imgs["fast"]
print("rows %d: mean rel diff %s, rmse/mean %.4f, pixels off by > 0.05: %.4f" % (
    n_rows, [round(x, 5) for x in ((b.mean(dim=(0, 1)) - a.mean(dim=(0, 1))) / a.mean(dim=(0, 1))).tolist()],
    float(((a - b) ** 2).mean().sqrt() / a.mean()), float(((a - b).abs().amax(dim=2) > 0.05).double().mean())))
