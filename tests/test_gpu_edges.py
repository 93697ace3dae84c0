"""Edges of the render entry point that the reference leaves undefined or handles differently:
pixel sums (32.32 fixed point where the reference has a float `totalLight`, main.cu:61-67), transforms that cannot be
inverted, and rays that run exactly along a world axis through the hierarchy."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scene(pt, emissive):
    text = ("material odd emissive %s\nmaterial lamp emissive 3 2 1\nmaterial grey albedo 0.5 0.5 0.5\n"
            "sphere odd translate -1.2 0 -6 scale 1.4\nsphere lamp translate 1.2 0 -6 scale 1.4\n"
            "plane grey bounds -15 -15 15 15 translate 0 0 -9\n") % emissive
    return pt.Scene(*pt.parse_scene(text))


@pytest.mark.parametrize("mode", ["fast", "parity", "scalar"])
def test_infinite_and_nan_radiance_stay_in_their_pixels(torch, pt, mode):
    """Depth 0: a pixel's value is the emission of what its camera rays hit.  An infinite emission counts as
    2^30 / n_samples per sample (the reference's pixel would be inf); a material with a NaN emissive component does not emit, as in the
    reference, whose test emissive.norm() > 0 (bsdf.h:90) is false for it.  In both cases every other pixel is bit for bit what it is with an ordinary emission there."""
    mm = {"fast": pt.PT_MATH_FAST, "parity": pt.PT_MATH_PARITY, "scalar": pt.PT_MATH_FAST_SCALAR}[mode]
    w, h, spp = 160, 90, 4
    cfg = pt.make_config(w, h, spp, n_bounces=0, math_mode=mm)
    base = _scene(pt, "5 6 7").render(cfg)
    on_odd = (base[..., 0] > 0) & (base[..., 2] > base[..., 0])          # pixels that see the first sphere (5 6 7: blue largest)
    assert 0.01 < float(on_odd.float().mean()) < 0.2
    inf_img = _scene(pt, "inf inf inf").render(cfg)
    nan_img = _scene(pt, "nan 6 7").render(cfg)
    torch.cuda.synchronize()
    assert torch.isfinite(inf_img).all() and torch.isfinite(nan_img).all()
    assert torch.equal(inf_img[~on_odd], base[~on_odd]) and torch.equal(nan_img[~on_odd], base[~on_odd])
    # fully covered pixels: every sample adds 2^30 / spp (inf) / nothing (NaN)
    full = base[..., 0] == 5.0
    assert bool(full.any())
    assert bool((inf_img[full] == 2.0 ** 30 / spp).all())
    assert bool((nan_img[full] == 0.0).all())
    # a negative emission is summed as it is
    neg = _scene(pt, "-5 6 -0.25").render(cfg)
    assert bool((neg[full][:, 0] == -5.0).all()) and bool((neg[full][:, 2] == -0.25).all())


def test_sums_near_the_fixed_point_range(torch, pt):
    """A sample counts up to 2^30 / n_samples: a light brighter than that is refused; at the cap the frame renders and
    the mean is exact (1024 samples of 2^20 sum to 2^30 in the integer word)."""
    lib = pt.binding.lib
    sc = _scene(pt, "1048576 524288 3")               # 2^20, 2^19
    cfg = pt.make_config(64, 36, 1024, n_bounces=0)   # 1024 x 2^20 = 2^30
    img = sc.render(cfg)
    full = img[..., 2] == 3.0
    assert bool(full.any()) and bool((img[full][:, 0] == 1048576.0).all()) and bool((img[full][:, 1] == 524288.0).all())
    sc2 = _scene(pt, "1048577 1 1")                   # 1024 x (2^20 + 1) > 2^30
    out = torch.zeros((36, 64, 3), dtype=torch.float32, device="cuda")
    rc = lib.pt_render(sc2.h, C.byref(cfg), C.c_void_p(out.data_ptr()), None)
    assert rc != 0 and b"2^30" in lib.pt_last_error()
    assert lib.pt_render(sc2.h, C.byref(pt.make_config(64, 36, 1023, n_bounces=0)), C.c_void_p(out.data_ptr()), None) == 0


def test_upload_refuses_transforms_without_a_finite_inverse(pt):
    lib = pt.binding.lib
    prims, n, mats, nm = pt.default_scene()
    h = C.c_void_p(0)
    for entry, value, what in ((0, float("nan"), b"not finite"), (5, float("inf"), b"not finite")):
        keep = prims[7].transform[entry]
        prims[7].transform[entry] = value
        assert lib.pt_scene_upload(prims, n, mats, nm, 0, C.byref(h)) != 0 and what in lib.pt_last_error()
        prims[7].transform[entry] = keep
    keep = [prims[10].transform[j] for j in range(12)]
    for j in (0, 1, 2):                                # a cube flattened to nothing along x: singular
        prims[10].transform[4 * j] = 0.0
    assert lib.pt_scene_upload(prims, n, mats, nm, 0, C.byref(h)) != 0 and b"singular" in lib.pt_last_error()
    for j in range(12):
        prims[10].transform[j] = keep[j]
    prims[0].bounds[1] = float("nan")
    assert lib.pt_scene_upload(prims, n, mats, nm, 0, C.byref(h)) != 0 and b"bound" in lib.pt_last_error()
    prims[0].bounds[1] = -float("inf")                 # an unbounded side is fine
    assert lib.pt_scene_upload(prims, n, mats, nm, 0, C.byref(h)) == 0
    lib.pt_scene_free(h)
    # 100 shapes, one with a NaN transform: refused before any hierarchy is built
    big = pt.synthetic_scene(60, 40, 7)
    big[0][50].transform[3] = float("nan")
    assert lib.pt_scene_upload(big[0], big[1], big[2], big[3], 0, C.byref(h)) != 0


def test_axis_parallel_rays_through_the_hierarchy(torch, pt):
    """Rays along -z, +x and +y (two direction components exactly 0) through 10 000 primitives: the hierarchy walk
    (PT_MATH_FAST) finds what the loop over all primitives (PT_MATH_FAST_SCALAR) finds.  pt_render with a camera so far
    away that every camera ray is numerically parallel to -z is the image-level form of the same."""
    sc = pt.Scene(*pt.synthetic_scene(5000, 4989, 12345))
    assert sc.bvh_nodes > 0
    n = 4096
    rng = np.random.default_rng(2)
    for axis, sign in ((2, -1.0), (0, 1.0), (1, 1.0)):
        o = rng.uniform([-0.9, -0.9, -4.9], [0.9, 0.9, -1.5], size=(n, 3)).astype(np.float32)
        o[:, axis] = -sign * 8.0 + (0.0 if axis != 2 else -3.0)
        d = np.zeros((n, 3), np.float32); d[:, axis] = sign
        args = [torch.from_numpy(x).cuda() for x in (o, d, np.arange(n, dtype=np.int32), np.zeros(n, np.int32))]
        fast = sc.trace_paths(*args, n_bounces=0, math_mode=pt.PT_MATH_FAST).cpu().numpy()
        par = sc.trace_paths(*args, n_bounces=0, math_mode=pt.PT_MATH_PARITY).cpu().numpy()
        assert (np.abs(fast - par).max(axis=1) > 1e-3).mean() < 0.002, axis
    a = sc.render(pt.make_config(96, 54, 2, math_mode=pt.PT_MATH_FAST))
    b = sc.render(pt.make_config(96, 54, 2, math_mode=pt.PT_MATH_FAST_SCALAR))
    assert torch.equal(a, b)
