"""Renders one frame configuration a few times so that rocprofv3 can be wrapped around it (no oracle, no timing of
its own beyond one line).  The scenes are the ones the benchmark configurations name:

    python3 tools/profile_frame.py default  1920 1080 64 5            # headline frame
    python3 tools/profile_frame.py glass5   1920 1080 64 31           # glass grid, depth 31 (divergence stress)
    python3 tools/profile_frame.py synthetic 1920 1080 4 5            # 10 000 primitives (hierarchy walk)
    python3 tools/profile_frame.py default  1920 1080 64 5 general    # general record kinds only (a rotated scene's cost)

Optional trailing words: `general` (general_kinds = all), `parity` / `scalar` / `lds` (math mode; default fast),
`nobvh` (no hierarchy: PT_MATH_FAST then runs its typed loops over every primitive), `tune=KNOB:VALUE` (pt_scene_tune,
may repeat), `reps=N` (default 4).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import pathtracer_amd as pt  # noqa: E402


def get_scene(name):
    if name == "default":
        return pt.default_scene()
    if name.startswith("glass"):
        return pt.glass_grid_scene(int(name[5:] or 5))
    if name == "synthetic":
        return pt.synthetic_scene(5000, 4989, 12345)
    raise SystemExit("unknown scene " + name)


def main():
    name = sys.argv[1]
    w, h, spp, depth = (int(x) for x in sys.argv[2:6])
    words = sys.argv[6:]
    reps = 4
    for x in words:
        if x.startswith("reps="):
            reps = int(x[5:])
    general = pt.PT_GENERAL_ALL if "general" in words else 0
    mode = pt.PT_MATH_PARITY if "parity" in words else (pt.PT_MATH_FAST_SCALAR if "scalar" in words else (pt.PT_MATH_FAST_LDS if "lds" in words else pt.PT_MATH_FAST))
    sc = pt.Scene(*get_scene(name), general_kinds=general, bvh_min_prims=(1 << 30) if "nobvh" in words else 0)
    for x in words:
        if x.startswith("tune="):
            k, v = x[5:].split(":")
            sc.tune(int(k), int(v))
    cfg = pt.make_config(w, h, spp, n_bounces=depth, math_mode=mode)
    out = sc.render(cfg)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        sc.render(cfg, out=out)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("%s %s %dx%d@%d depth %d general=%d: min %.3f ms mean %.3f ms, %.1f Msamples/s, image mean %s" % (
        name, {pt.PT_MATH_PARITY: "parity", pt.PT_MATH_FAST: "fast", pt.PT_MATH_FAST_SCALAR: "fast_scalar", pt.PT_MATH_FAST_LDS: "fast_lds"}[mode] + (" nobvh" if "nobvh" in words else ""), w, h, spp, depth, general, min(ts), sum(ts) / len(ts),
        w * h * spp / min(ts) / 1e3, [round(x, 5) for x in out.mean(dim=(0, 1)).tolist()]), flush=True)


main()
