"""Regenerates tests/golden/*.npz from the CPU oracle (oracle/liboracle.so).

The reference pins only n_intersections counts (tests/shape_test.cc); these
vectors are the repo's own pins of the oracle's t-values, nearest hits, RNG
stream and default-scene image, so that a drift of the oracle (compiler, libm)
or of the kernels shows up as a diff against committed data.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_binding as ob  # noqa: E402


def shape_cases():
    """(name, type, transform12, bounds) — identity, translated, rotated+scaled variants of all four shapes."""
    big = float(np.finfo(np.float32).max)
    prims, n, _, _ = ob.default_scene()
    cases = []
    ident = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0]
    for ty, nm in ((ob.PLANE, "plane"), (ob.CUBE, "cube"), (ob.CYLINDER, "cylinder"), (ob.SPHERE, "sphere")):
        cases.append((nm + "_identity", ty, ident, (-big, -big, big, big)))
        t = list(ident); t[3], t[7], t[11] = 0.3, -0.2, -2.5
        cases.append((nm + "_translated", ty, t, (-1.0, -0.5, 1.0, 0.75)))
    # transforms taken from the default scene: rotated planes, scaled spheres, anisotropic cube
    for k, nm in ((0, "floor"), (1, "left"), (5, "light"), (6, "front"), (7, "glass1"), (10, "box")):
        p = prims[k]
        cases.append(("scene_" + nm, p.type, list(p.transform), tuple(p.bounds)))
    # the same anisotropic transform on a cylinder and a sphere
    p = prims[10]
    cases.append(("cylinder_aniso", ob.CYLINDER, list(p.transform), tuple(p.bounds)))
    cases.append(("sphere_aniso", ob.SPHERE, list(p.transform), tuple(p.bounds)))
    return cases


def rays_for(transform, n, rng):
    """Rays aimed near the primitive (so most of them hit) from origins around the camera."""
    c = np.array([transform[3], transform[7], transform[11]], np.float32)
    o = (rng.uniform(-0.8, 0.8, (n, 3)) + np.array([0, 0, 0.5])).astype(np.float32)
    target = c + rng.uniform(-0.7, 0.7, (n, 3)).astype(np.float32)
    d = target - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    # a few rays that start inside / on the primitive and a few that point away
    o[: n // 16] = c + rng.uniform(-0.05, 0.05, (n // 16, 3)).astype(np.float32)
    d[n // 16: n // 8] *= -1
    return np.ascontiguousarray(o), np.ascontiguousarray(d)


def main():
    ob.build()
    rng = np.random.default_rng(20251111)
    out = {}
    names = []
    for name, ty, tr, b in shape_cases():
        prim = ob.make_prim(ty, transform=tr, bounds=b)
        o, d = rays_for(tr, 1024, rng)
        r = ob.raypath(prim, o, d, filter=True)
        names.append(name)
        out[name + "/type"] = np.int32(ty)
        out[name + "/transform"] = np.array(tr, np.float32)
        out[name + "/bounds"] = np.array(b, np.float32)
        out[name + "/o"], out[name + "/d"] = o, d
        for k in ("n", "hit1", "t1", "pos1", "nrm1", "hit2", "t2", "pos2", "nrm2"):
            out[name + "/" + k] = r[k]
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "raypath_golden.npz"), **out)

    # nearest hits + tracePath radiance on the default scene
    prims, n, mats, nm = ob.default_scene()
    sc = ob.Scene(prims, n, mats, nm)
    cfg = ob.make_config(256, 256, 16)
    pi = rng.integers(0, 256, 4096).astype(np.int32)
    pj = rng.integers(0, 256, 4096).astype(np.int32)
    smp = rng.integers(0, 16, 4096).astype(np.uint32)
    o, d = ob.camera_rays(cfg, pi, pj, smp)
    # second half: rays from inside the box in random directions (bounce-like)
    o[2048:] = rng.uniform([-0.95, -0.95, -4.9], [0.95, 0.95, 0.0], (2048, 3)).astype(np.float32)
    dd = rng.normal(size=(2048, 3)); dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    d[2048:] = dd.astype(np.float32)
    near = sc.nearest(o, d)
    pixel = (pj.astype(np.uint32) * 256 + pi.astype(np.uint32)).astype(np.uint32)
    rgb = sc.trace_paths(o, d, pixel, smp, seed=0, n_bounces=5)
    np.savez_compressed(os.path.join(HERE, "default_scene_paths.npz"), o=o, d=d, pixel=pixel, sample=smp,
                        prim=near["prim"], t=near["t"], pos=near["pos"], nrm=near["nrm"], inside=near["inside"], rgb=rgb)

    # RNG stream
    rb = {}
    for seed, pixel_, sample in ((0, 0, 0), (0, 12345, 7), (1, 2073599, 63), (12345, 8294399, 1023)):
        bits, u = ob.rng(seed, pixel_, sample, 64)
        rb["bits_%d_%d_%d" % (seed, pixel_, sample)] = bits
        rb["u_%d_%d_%d" % (seed, pixel_, sample)] = u
    np.savez_compressed(os.path.join(HERE, "rng_golden.npz"), **rb)

    # default scene, 256x256 @ 16 spp, seed 0 (config 1)
    fb = sc.render(cfg)
    np.savez_compressed(os.path.join(HERE, "default_256x256_16spp.npz"), fb=fb)
    alternative_bsdf()
    print("wrote golden vectors:", sorted(os.listdir(HERE)))


def alternative_bsdf():
    """Radiance of the rays of default_scene_paths.npz under the tracePath of bsdf_alternative.h, plus a 96x64@32 image."""
    g = np.load(os.path.join(HERE, "default_scene_paths.npz"))
    prims, n, mats, nm = ob.default_scene()
    sc = ob.Scene(prims, n, mats, nm)
    rgb = sc.trace_paths(g["o"], g["d"], g["pixel"], g["sample"], seed=0, n_bounces=5, bsdf_model=1)
    fb = sc.render(ob.make_config(96, 64, 32, bsdf_model=1))
    np.savez_compressed(os.path.join(HERE, "alternative_bsdf.npz"), rgb=rgb, fb=fb)


if __name__ == "__main__":
    main()
