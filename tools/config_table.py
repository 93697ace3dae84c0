"""The result rows of the five benchmark configurations, both math modes: pt_render kernel time (HIP events, best of a
few), Msamples/s, algorithmic flops per sample and the share of the FP32 vector peak they amount to, and the image
against the CPU oracle on a bounded set of rows of the same frame (same seed and samples per pixel).

    python3 tools/config_table.py [out.json]          (GPU box; the oracle uses the box's CPU share)

The oracle is only the checker here; nothing it computes is timed as the product.  A configuration can be marked to
render only the compared rows in parity order (none is, since the parity-order kernels search a hierarchy too).
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import oracle_binding as ob  # noqa: E402
import pathtracer_amd as pt  # noqa: E402
from test_gpu_configs import _as_oracle  # noqa: E402

PEAK = 157.3e12
FLOPS = json.load(open(os.path.join(ROOT, "profiles", "flops_per_sample.json")))

# name, scene, width, height, spp, depth, rows the oracle renders (step), flops key, rows-only render in parity mode
CONFIGS = [
    ("1", "default", 256, 256, 16, 5, 1, "default_256x256_16spp_depth5_seed0", False),
    ("2", "default", 1920, 1080, 64, 5, 1, "default_1920x1080_64spp_depth5_seed0", False),
    ("3", "default", 3840, 2160, 1024, 5, 400, "default_3840x2160_1024spp_depth5_seed0_rows%270", False),
    ("4", "glass5", 1920, 1080, 64, 31, 8, "glass5_1920x1080_64spp_depth31_seed0_rows%8", False),
    ("5", "synthetic", 1920, 1080, 64, 5, 540, "synthetic_1920x1080_4spp_depth5_seed0_rows%120", False),
]


def get_scene(name):
    if name == "default":
        return pt.default_scene()
    if name.startswith("glass"):
        return pt.glass_grid_scene(int(name[5:]))
    return pt.synthetic_scene(5000, 4989, 12345)


def time_render(sc, cfg, out, reps):
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        sc.render(cfg, out=out)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts)


def main():
    rows_out = []
    for name, scene, w, h, spp, depth, step, fkey, rows_only in CONFIGS:
        prims, n, mats, nm = get_scene(scene)
        sc = pt.Scene(prims, n, mats, nm)
        osc = ob.Scene(*_as_oracle(ob, prims, n, mats, nm))
        rows = np.arange(step // 2 if step > 1 else 0, h, step, dtype=np.int32)
        t0 = time.perf_counter()
        ref = osc.render(ob.make_config(w, h, spp, n_bounces=depth), threads=ob.default_threads(), rows=rows)[rows].astype(np.float64)
        cpu_s = time.perf_counter() - t0
        fps = FLOPS[fkey]["flops_per_sample"]
        rows_out.append({"config": name, "mode": "cpu oracle", "device": "host CPU, %d threads" % ob.default_threads(),
                         "frame": "%dx%d@%d depth %d" % (w, h, spp, depth), "rows": int(len(rows)),
                         "msamples_per_s": len(rows) * w * spp / cpu_s / 1e6, "flops_per_sample": fps, "seconds": cpu_s})
        print(json.dumps(rows_out[-1]), flush=True)
        for mode, mm in (("fast", pt.PT_MATH_FAST), ("parity", pt.PT_MATH_PARITY)):
            partial = rows_only and mode == "parity"
            if partial:        # the compared rows only: row_begin / row_step select them, fb_compact packs them
                cfg = pt.make_config(w, h, spp, n_bounces=depth, math_mode=mm, row_begin=int(rows[0]), row_end=h, row_step=step, fb_compact=1)
                out = torch.zeros((len(rows), w, 3), device="cuda")
                n_samples = len(rows) * w * spp
            else:
                cfg = pt.make_config(w, h, spp, n_bounces=depth, math_mode=mm)
                out = torch.zeros((h, w, 3), device="cuda")
                n_samples = w * h * spp
            sc.render(cfg, out=out)
            torch.cuda.synchronize()
            ms = time_render(sc, cfg, out, 2 if (name in ("3", "5") or mode == "parity") else 5)
            img = out.cpu().numpy().astype(np.float64)
            img = img if partial else img[rows]
            mean = ref.mean()
            rec = {"config": name, "mode": mode, "device": "1x MI355X", "frame": "%dx%d@%d depth %d" % (w, h, spp, depth),
                   "rendered": "compared rows only" if partial else "whole frame",
                   "kernel_ms": ms, "msamples_per_s": n_samples / ms / 1e3, "flops_per_sample": fps,
                   "pct_fp32_peak_algorithmic": 100.0 * fps * n_samples / (ms * 1e-3) / PEAK,
                   "rows_compared": int(len(rows)),
                   "rmse_rel": float(np.sqrt(((img - ref) ** 2).mean(axis=(0, 1))).max() / mean),
                   "max_abs": float(np.abs(img - ref).max()),
                   "bad_pixel_frac": float((np.abs(img - ref).max(axis=2) > 0.05).mean()),
                   "mean_rel": float((np.abs(img.mean(axis=(0, 1)) - ref.mean(axis=(0, 1))) / ref.mean(axis=(0, 1))).max()),
                   "finite": bool(np.isfinite(img).all())}
            rows_out.append(rec)
            print(json.dumps(rec), flush=True)
    if len(sys.argv) > 1:
        os.makedirs(os.path.dirname(os.path.abspath(sys.argv[1])), exist_ok=True)
        json.dump(rows_out, open(sys.argv[1], "w"), indent=1)


main()
