"""getNearestIntersection (shapes.h:79-100) as the PT_MATH_FAST kernels run it (pt_nearest_batch_mode), against the
oracle: winning primitive, distance in ULPs, position, facing normal and inside flag.  Rays are camera rays and
secondary rays (origins on surfaces, random directions) of the default scene with the short and the general record
kinds, of the glass grid, of a scene of rotated cylinders / ellipsoids / cubes, and of the 10 000-primitive scene
(hierarchy walk)."""
import ctypes as C

import numpy as np
import pytest

import tolerances as tol
from test_gpu_configs import _as_oracle, _mixed_scene_text

pytestmark = pytest.mark.gpu


def ulp_diff(a, b):
    ai = a.view(np.int32).astype(np.int64); bi = b.view(np.int32).astype(np.int64)
    return np.abs(ai - bi)


def rays_for(osc, n, seed):
    """n/2 camera-like rays from the origin and n/2 rays that leave the first hit of those in a random direction."""
    rng = np.random.default_rng(seed)
    h = n // 2
    o = np.zeros((h, 3), np.float32)
    d = rng.normal(size=(h, 3)); d[:, 2] = -np.abs(d[:, 2]) - 1.5
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    first = osc.nearest(o, d)
    o2 = first["pos"].astype(np.float32)
    d2 = rng.normal(size=(h, 3))
    d2 = (d2 / np.linalg.norm(d2, axis=1, keepdims=True)).astype(np.float32)
    keep = first["prim"] >= 0
    return np.concatenate([o, o2[keep]]), np.concatenate([d, d2[keep]])


def check(name, pt, oracle, scene, n_rays, seed, **scene_kw):
    assert name.startswith("default") or name in ("glass grid", "mixed shapes", "10 000 primitives")
    import torch
    prims, n, mats, nm = scene
    sc = pt.Scene(prims, n, mats, nm, **scene_kw)
    osc = oracle.Scene(*_as_oracle(oracle, prims, n, mats, nm))
    o, d = rays_for(osc, n_rays, seed)
    ref = osc.nearest(o, d)
    got = {k: v.cpu().numpy() for k, v in sc.nearest(torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda(), math_mode=pt.PT_MATH_FAST).items()}
    same = got["prim"] == ref["prim"]
    hit = same & (ref["prim"] >= 0)
    # a ray that leaves a surface almost along it re-hits that surface, or not, depending on the last bit of its origin
    # (the origin is half an ULP off the plane, the direction's normal component 1e-4: t = 1e-3, beyond the 1e-5
    # epsilon of shapes.h:62-73), and two shapes can be equally near: those rays may name another primitive
    tg = np.where(got["prim"] >= 0, got["t"], np.inf).astype(np.float64)
    tr = np.where(ref["prim"] >= 0, ref["t"], np.inf).astype(np.float64)
    with np.errstate(invalid="ignore"):
        explained = (np.minimum(tg, tr) < tol.FAST["rehit_t"]) | (np.abs(tg - tr) <= tol.FAST["t_rel_max"] * np.minimum(tg, tr))
    # what is left are grazing hits (a discriminant within rounding of zero: one side sees the shape, the other misses it)
    assert (~(same | explained)).mean() <= tol.FAST["nearest_grazing"], (name, int((~(same | explained)).sum()))
    assert same.mean() >= 1.0 - tol.FAST["nearest_prim_mismatch"], (name, 1.0 - same.mean())
    # same primitive, other root: a ray that starts on a sphere or cube and sees, or does not see, the surface it leaves
    with np.errstate(invalid="ignore"):          # inf - inf where both miss
        flip = hit & (np.abs(tg - tr) > 1e-2 * np.maximum(np.minimum(tg, tr), 1e-3))
    assert flip.mean() <= tol.FAST["nearest_prim_mismatch"], (name, flip.mean())
    hit &= ~flip
    # distances.  For a camera ray (origin 0) the error is that of the test's arithmetic: in ULPs of t.  For a ray that
    # starts on a surface, t carries the rounding of |o| as well, so the error is taken in ULPs of max(t, |o|).
    cam = hit & (np.abs(o).max(axis=1) == 0.0)
    u_cam = ulp_diff(got["t"][cam], ref["t"][cam])
    scale = np.maximum(np.maximum(tr, np.abs(o).max(axis=1)), 1e-3)
    u_all = np.abs(tg[hit] - tr[hit]) / (scale[hit] * 2.0 ** -23)
    dn = np.abs(got["nrm"][hit] - ref["nrm"][hit]).max(axis=1)
    dp = np.abs(got["pos"][hit] - ref["pos"][hit]).max(axis=1) / np.maximum(np.abs(ref["pos"][hit]).max(axis=1), 1.0)
    ins = int((got["inside"][hit] != ref["inside"][hit]).sum())
    print("%s: %d rays, %d compared, other primitive %.2e, other root %.2e | t of camera rays, ULP: median %d p99 %d max %d | "
          "t of all rays, ULP of max(t,|o|): p99 %.1f max %.1f | pos rel max %.2e | nrm abs p99 %.2e max %.2e | inside flag differs %d"
          % (name, len(o), hit.sum(), 1.0 - same.mean(), flip.mean(), np.median(u_cam), np.percentile(u_cam, 99), u_cam.max(),
             np.percentile(u_all, 99), u_all.max(), dp.max(), np.percentile(dn, 99), dn.max(), ins))
    B = tol.FAST_NEAREST["default" if name.startswith("default") else "any"]
    assert np.median(u_cam) <= B["t_ulp_median"] and np.percentile(u_cam, 99) <= B["t_ulp_p99"] and u_cam.max() <= B["t_ulp_max"], name
    assert u_all.max() <= B["t_ulp_max"], name
    assert dp.max() <= B["pos_rel_max"], name
    assert np.percentile(dn, 99) <= B["nrm_abs_p99"], name
    assert ins <= tol.FAST["nearest_prim_mismatch"] * len(o), name
    # the parity-order search through the same entry point is pt_nearest_batch
    a = sc.nearest(torch.from_numpy(o[:512]).cuda(), torch.from_numpy(d[:512]).cuda())
    L = pt.binding.lib
    b = {k: torch.zeros_like(v) for k, v in a.items()}
    oo, dd_ = torch.from_numpy(o[:512]).cuda(), torch.from_numpy(d[:512]).cuda()
    assert L.pt_nearest_batch_mode(sc.h, C.c_void_p(oo.data_ptr()), C.c_void_p(dd_.data_ptr()), 512, pt.PT_MATH_PARITY,
                                   *(C.c_void_p(b[k].data_ptr()) for k in ("prim", "t", "pos", "nrm", "inside")), None) == 0
    torch.cuda.synchronize()
    assert all(torch.equal(a[k], b[k]) for k in a)


def test_default_scene_short_kinds(oracle):
    import pathtracer_amd as pt
    check("default scene", pt, oracle, pt.default_scene(), 65536, 1)


def test_default_scene_general_kinds(oracle):
    import pathtracer_amd as pt
    check("default scene, general kinds", pt, oracle, pt.default_scene(), 65536, 2, general_kinds=pt.PT_GENERAL_ALL)


def test_glass_grid(oracle):
    import pathtracer_amd as pt
    check("glass grid", pt, oracle, pt.glass_grid_scene(5), 65536, 3)


def test_rotated_cylinders_ellipsoids_cubes(oracle):
    import pathtracer_amd as pt
    check("mixed shapes", pt, oracle, pt.parse_scene(_mixed_scene_text(40)), 32768, 4)


def test_hierarchy_10k_primitives(oracle):
    import pathtracer_amd as pt
    check("10 000 primitives", pt, oracle, pt.synthetic_scene(5000, 4989, 12345), 4096, 5)


def test_unknown_mode_is_refused():
    import torch
    import pathtracer_amd as pt
    sc = pt.Scene(*pt.default_scene())
    z = torch.zeros((4, 3), device="cuda")
    with pytest.raises(pt.PtError):
        sc.nearest(z, z, math_mode=99)
