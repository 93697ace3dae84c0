"""Many random scenes through the hierarchy and through the loop over all primitives, fast and parity order, bit for
bit (development aid; the test suite runs a handful of these).
    python3 tools/hierarchy_soak.py [scenes]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import pathtracer_amd as pt
from test_gpu_configs import _mixed_scene_text

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(2024)
bad = 0
for s in range(n_scenes):
    n = int(rng.choice([61, 64, 65, 97, 128, 200, 333, 777, 1500]))
    walls = bool(rng.integers(0, 2))
    text = _mixed_scene_text(n, walls=walls, seed=int(rng.integers(1, 1 << 30)))
    prims, k, mats, nm = pt.parse_scene(text)
    opts = {}
    if rng.integers(0, 3) == 0:
        opts["bvh_split"] = int(rng.choice([1, 2, 3, 7]))
    sc = pt.Scene(prims, k, mats, nm, **opts)
    assert sc.bvh_nodes > 0
    w, h, spp, depth = int(rng.integers(40, 200)), int(rng.integers(20, 100)), int(rng.integers(1, 6)), int(rng.choice([0, 3, 5, 12, 31]))
    seed = int(rng.integers(0, 1000))
    rows = {}
    if s % 2:
        # a wide frame, a few of its rows: tiles narrow enough for the camera batches' per-tile lists of primitives
        w, h = int(rng.integers(600, 2000)), int(rng.integers(400, 1100))
        r0 = int(rng.integers(0, h - 12))
        rows = dict(row_begin=r0, row_end=r0 + int(rng.integers(1, 12)), fb_compact=1, tile_log2=int(rng.choice([0, 3, 4, 5, 6])))
    a = sc.render(pt.make_config(w, h, spp, n_bounces=depth, seed=seed, math_mode=pt.PT_MATH_FAST, **rows))
    b = sc.render(pt.make_config(w, h, spp, n_bounces=depth, seed=seed, math_mode=pt.PT_MATH_FAST_SCALAR, **rows))
    p1 = sc.render(pt.make_config(w, h, spp, n_bounces=depth, seed=seed, math_mode=pt.PT_MATH_PARITY, **rows))
    sc.tune(pt.binding.PT_TUNE_BVH_WALK, 3)
    p2 = sc.render(pt.make_config(w, h, spp, n_bounces=depth, seed=seed, math_mode=pt.PT_MATH_PARITY, **rows))
    sc.tune(pt.binding.PT_TUNE_BVH_WALK, 2)
    a2 = sc.render(pt.make_config(w, h, spp, n_bounces=depth, seed=seed, math_mode=pt.PT_MATH_FAST, **rows))
    ok_f, ok_p = bool(torch.equal(a, b)) and bool(torch.equal(a2, b)), bool(torch.equal(p1, p2))
    if not (ok_f and ok_p):
        bad += 1
        print("MISMATCH scene %d: n %d walls %s opts %s frame %dx%d@%d depth %d seed %d fast %s parity %s" % (s, n, walls, opts, w, h, spp, depth, seed, ok_f, ok_p), flush=True)
    elif s % 10 == 0:
        print("scene %d ok (n %d, %dx%d@%d depth %d, finite %s)" % (s, n, w, h, spp, depth, bool(torch.isfinite(a).all())), flush=True)
    sc.close()
print("%d scenes, %d mismatches" % (n_scenes, bad))
