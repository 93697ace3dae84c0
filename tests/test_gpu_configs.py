"""GPU parity for BASELINE configs 4 and 5 at sizes the oracle finishes in seconds:
the glass-grid stress scene at depth 31 and the 10 000-primitive synthetic scene."""
import ctypes as C
import os

import numpy as np
import pytest

import tolerances as tol

pytestmark = pytest.mark.gpu


def _as_oracle(ob, prims, n, mats, nm):
    op = (ob.Prim * n).from_buffer_copy(bytes(prims)[: n * C.sizeof(ob.Prim)])
    om = (ob.Material * nm).from_buffer_copy(bytes(mats)[: nm * C.sizeof(ob.Material)])
    return op, n, om, nm


def _stats(img, ref):
    img = img.astype(np.float64); ref = ref.astype(np.float64)
    rmse = float(np.sqrt(((img - ref) ** 2).mean()) / ref.mean())
    bad = float((np.abs(img - ref).max(axis=2) > 0.05).mean())
    return rmse, bad


def test_glass_grid_depth_32_segments(oracle):
    import pathtracer_amd as pt
    prims, n, mats, nm = pt.glass_grid_scene(5)
    sc = pt.Scene(prims, n, mats, nm)
    osc = oracle.Scene(*_as_oracle(oracle, prims, n, mats, nm))
    w, h, spp, depth = 192, 108, 16, 31
    ref = osc.render(oracle.make_config(w, h, spp, n_bounces=depth))
    par = sc.render(pt.make_config(w, h, spp, n_bounces=depth, math_mode=pt.PT_MATH_PARITY)).cpu().numpy()
    fast = sc.render(pt.make_config(w, h, spp, n_bounces=depth, math_mode=pt.PT_MATH_FAST)).cpu().numpy()
    rp, bp = _stats(par, ref)
    rf, bf = _stats(fast, ref)
    print("glass5 depth 31: parity rmse %.2e bad %.5f | fast rmse %.2e bad %.5f" % (rp, bp, rf, bf))
    assert np.isfinite(par).all() and np.isfinite(fast).all()
    # Refraction through spheres is chaotic: a 1e-6 relative difference (fast math) grows along a 32-segment
    # path until the path takes another branch, so the fast build is held to a flipped-pixel budget and to the
    # image mean, not to a per-pixel RMSE at 16 spp (one flipped light hit moves a pixel by 20/16).
    G = tol.GLASS_DEPTH_31
    assert rp <= G["parity_image_rmse"] and bp <= G["parity_bad_pixel_frac"]
    assert rf <= G["fast_image_rmse"] and bf <= G["fast_bad_pixel_frac"]
    np.testing.assert_allclose(fast.mean(axis=(0, 1)), ref.mean(axis=(0, 1)), rtol=G["fast_mean_rel"])


def test_synthetic_10k_primitives(oracle):
    import pathtracer_amd as pt
    prims, n, mats, nm = pt.synthetic_scene(5000, 4989, 12345)
    assert n == 10000
    sc = pt.Scene(prims, n, mats, nm)
    osc = oracle.Scene(*_as_oracle(oracle, prims, n, mats, nm))
    # nearest hits on random camera-like rays: exact primitive index and t within 2 ULP (parity-order kernel)
    import torch
    rng = np.random.default_rng(5)
    nr = 2048
    o = np.zeros((nr, 3), np.float32)
    d = rng.normal(size=(nr, 3)); d[:, 2] = -np.abs(d[:, 2]) - 1.0
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    ref = osc.nearest(o, d)
    got = {k: v.cpu().numpy() for k, v in sc.nearest(torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()).items()}
    assert np.array_equal(got["prim"], ref["prim"])
    assert (ref["prim"] >= 11).mean() > 0.1          # a good share of the rays hit the small shapes
    ai = got["t"].view(np.int32).astype(np.int64); bi = ref["t"].view(np.int32).astype(np.int64)
    assert np.abs(ai - bi).max() <= 2
    # a small render, both math modes
    w, h, spp = 48, 27, 2
    img_ref = osc.render(oracle.make_config(w, h, spp))
    par = sc.render(pt.make_config(w, h, spp, math_mode=pt.PT_MATH_PARITY)).cpu().numpy()
    fast = sc.render(pt.make_config(w, h, spp, math_mode=pt.PT_MATH_FAST)).cpu().numpy()
    rp, bp = _stats(par, img_ref)
    rf, bf = _stats(fast, img_ref)
    print("synthetic 10k: parity rmse %.2e bad %.5f | fast rmse %.2e bad %.5f" % (rp, bp, rf, bf))
    assert rp <= 1e-4 and bp == 0.0
    assert bf <= tol.MANY_SMALL_SHAPES["fast_bad_pixel_frac"]


def test_lds_tiled_loop_equals_scalar_loop_bitwise():
    """The LDS-staged primitive loop and the scalar-load loop run the same arithmetic on the same records."""
    import torch
    import pathtracer_amd as pt
    for scene, (w, h, spp, depth) in ((pt.default_scene(), (320, 200, 8, 5)), (pt.glass_grid_scene(5), (200, 120, 4, 31)),
                                      (pt.synthetic_scene(5000, 4989, 12345), (96, 54, 2, 5)), (pt.synthetic_scene(700, 300, 7), (130, 50, 3, 5))):
        sc = pt.Scene(*scene)
        a = sc.render(pt.make_config(w, h, spp, n_bounces=depth, math_mode=pt.PT_MATH_FAST_SCALAR))
        b = sc.render(pt.make_config(w, h, spp, n_bounces=depth, math_mode=pt.PT_MATH_FAST_LDS))
        assert torch.isfinite(a).all()
        assert torch.equal(a, b)
    # a row tile that leaves some waves of the last workgroup without pixels
    sc = pt.Scene(*pt.default_scene())
    a = sc.render(pt.make_config(70, 3, 4, math_mode=pt.PT_MATH_FAST_SCALAR))
    b = sc.render(pt.make_config(70, 3, 4, math_mode=pt.PT_MATH_FAST_LDS))
    assert torch.equal(a, b)


def _mixed_scene_text(n, walls=True, seed=3):
    """Cylinders, ellipsoids, spheres and cubes with random poses; every fifth shape is an exact duplicate of
    the one before it (equal hit distances: the lower index has to win)."""
    rng = np.random.default_rng(seed)
    lines = ["material base albedo 0.4 0.4 0.4", "material red albedo 1 0 0", "material light emissive 20 18 14",
             "material glass f0 0.02 specularColor 0.8 0.8 0.8 IOR 1.5 transparency 1 bsdfType 1 refractionColor 0.9 0 0",
             "material mirror f0 0.9 specularColor 0.9 0.9 0.9 specularRoughness 0.2"]
    if walls:
        lines += ["plane base bounds -15 -15 15 15 rotate pi/2 1 0 0 translate 0 -1 0",
                  "plane red bounds -15 -15 15 15 translate 0 0 -5",
                  "plane light bounds -0.5 -0.5 0.5 0.5 rotate -pi/2 1 0 0 translate 0 0.97 -3.25"]
    else:
        lines += ["cube light translate 0 3 -3 scale 4 0.1 4"]
    prev = None
    for k in range(n):
        if k % 5 == 4 and prev:
            lines.append(prev)
            continue
        kind = ("cylinder", "sphere", "sphere", "cube")[k % 4]
        mat = ("base", "red", "glass", "mirror")[int(rng.integers(4))]
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        pos = rng.uniform([-0.9, -0.9, -4.8], [0.9, 0.9, -1.6])
        s = rng.uniform(0.05, 0.25, size=3)
        scale = "scale %.4f" % s[0] if (kind == "sphere" and k % 8 < 4) else "scale %.4f %.4f %.4f" % tuple(s)
        prev = "%s %s rotate %.4f %.4f %.4f %.4f translate %.4f %.4f %.4f %s" % (
            kind, mat, rng.uniform(0, 3.1), ax[0], ax[1], ax[2], pos[0], pos[1], pos[2], scale)
        lines.append(prev)
    return "\n".join(lines) + "\n"


def _camera_rays(w, h):
    """Rays from the origin through a w x h grid on the plane z = -1, as device tensors (origins, directions)."""
    import torch
    ys, xs = np.meshgrid(np.linspace(-0.9, 0.9, h), np.linspace(-1.6, 1.6, w), indexing="ij")
    d = np.stack([xs, ys, -np.ones_like(xs)], axis=-1).reshape(-1, 3)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    return torch.zeros((d.shape[0], 3), device="cuda"), torch.from_numpy(d).cuda()


def test_hierarchy_equals_loop_over_all_primitives_bitwise():
    """PT_MATH_FAST searches scenes with many finite primitives through a bounding-volume hierarchy; the nearest
    hit, and so the image, is the one the loop over all primitives (PT_MATH_FAST_SCALAR) finds, bit for bit."""
    import torch
    import pathtracer_amd as pt
    small = pt.Scene(*pt.default_scene())
    assert small.bvh_nodes == 0                        # 4 finite primitives: no hierarchy
    assert pt.Scene(*pt.glass_grid_scene(5)).bvh_nodes == 0   # 26: below the threshold, the loop is faster
    cases = ((pt.synthetic_scene(5000, 4989, 12345), (160, 90, 4, 5)), (pt.synthetic_scene(700, 300, 7), (130, 50, 3, 5)),
             (pt.glass_grid_scene(10), (200, 120, 4, 31)),
             (pt.parse_scene(_mixed_scene_text(120)), (192, 108, 8, 8)),
             (pt.parse_scene(_mixed_scene_text(100, walls=False)), (96, 64, 8, 5)))
    for scene, (w, h, spp, depth) in cases:
        sc = pt.Scene(*scene)
        assert sc.bvh_nodes > 0
        a = sc.render(pt.make_config(w, h, spp, n_bounces=depth, math_mode=pt.PT_MATH_FAST))
        b = sc.render(pt.make_config(w, h, spp, n_bounces=depth, math_mode=pt.PT_MATH_FAST_SCALAR))
        assert torch.isfinite(a).all() and float(a.sum()) > 0.0
        assert torch.equal(a, b)
    # every way of building and walking the tree gives that image too: the split rules, leaves of several primitives
    # (skip-link walk), the skip-link walk of one-primitive leaves, camera rays walking like every other ray (2)
    # instead of being traced in batches against their tile's own list of primitives
    scene, (w, h, spp, depth) = cases[3]
    ref = pt.Scene(*scene).render(pt.make_config(w, h, spp, n_bounces=depth, math_mode=pt.PT_MATH_FAST_SCALAR))
    for options, walk in (({"bvh_split": pt.binding.PT_BVH_SPLIT_MEDIAN}, 0), ({"bvh_split": pt.binding.PT_BVH_SPLIT_AREA}, 0),
                          ({"bvh_split": pt.binding.PT_BVH_SPLIT_AREA + 5}, 0), ({"bvh_leaf_prims": 4}, 0), ({}, 1), ({}, 2)):
        sc = pt.Scene(*scene, **options)
        assert sc.bvh_nodes > 0
        if walk:
            sc.tune(pt.binding.PT_TUNE_BVH_WALK, walk)
        assert torch.equal(sc.render(pt.make_config(w, h, spp, n_bounces=depth, math_mode=pt.PT_MATH_FAST)), ref), (options, walk)
    # the parity-order kernels search the same tree (skip-link form, reference arithmetic at the leaves)
    sc = pt.Scene(*scene)
    par_tree = sc.render(pt.make_config(w, h, spp, n_bounces=depth, math_mode=pt.PT_MATH_PARITY))
    hit_tree = sc.nearest(*_camera_rays(w, h))
    sc.tune(pt.binding.PT_TUNE_BVH_WALK, 3)
    par_loop = sc.render(pt.make_config(w, h, spp, n_bounces=depth, math_mode=pt.PT_MATH_PARITY))
    hit_loop = sc.nearest(*_camera_rays(w, h))
    assert float(par_tree.sum()) > 0.0 and torch.equal(par_tree, par_loop)
    assert int((hit_tree["prim"] >= 0).sum()) > 100
    for name in hit_tree:
        assert torch.equal(hit_tree[name], hit_loop[name]), name
    # 100 000 primitives: the grid of the node boxes is 16 bits per axis over the whole scene
    sc = pt.Scene(*pt.synthetic_scene(50000, 50000, 99))
    a = sc.render(pt.make_config(96, 54, 2, math_mode=pt.PT_MATH_FAST))
    b = sc.render(pt.make_config(96, 54, 2, math_mode=pt.PT_MATH_FAST_SCALAR))
    assert float(a.sum()) > 0.0 and torch.equal(a, b)
    # explicit rays through tracePath: the hierarchy there too, against the parity-order kernel's loop
    prims, n, mats, nm = pt.parse_scene(_mixed_scene_text(120))
    sc = pt.Scene(prims, n, mats, nm)
    rng = np.random.default_rng(11)
    nr = 4096
    o = np.zeros((nr, 3), np.float32)
    d = rng.normal(size=(nr, 3)); d[:, 2] = -np.abs(d[:, 2]) - 1.0
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    px = np.arange(nr, dtype=np.uint32); smp = np.zeros(nr, np.uint32)
    args = [torch.from_numpy(x).cuda() for x in (o, d, px, smp)]
    fast = sc.trace_paths(*args, seed=0, n_bounces=0, math_mode=pt.PT_MATH_FAST).cpu().numpy()
    par = sc.trace_paths(*args, seed=0, n_bounces=0, math_mode=pt.PT_MATH_PARITY).cpu().numpy()
    # one segment: the radiance is the emission of whatever the ray hits first
    assert (np.abs(fast - par).max(axis=1) > 1e-3).mean() < 0.002


def test_camera_batches_of_hierarchy_scenes_equal_the_loop_bitwise():
    """Scenes with a hierarchy trace their camera rays in batches against each tile's own list of primitives (found by
    descending the tree with the tile's view pyramid); tiles whose list does not fit park their camera rays unsearched.
    Wide frames (short lists), tiny frames (every tile sees most of the scene: no list), row subsets and tile sizes all
    give the image of the loop over all primitives, and so does the kernel in which camera rays walk the tree."""
    import torch
    import pathtracer_amd as pt
    for scene, frames in ((pt.synthetic_scene(5000, 4989, 12345), ((1920, 1080, 3, 5, 500, 520, 1), (1920, 1080, 2, 5, 3, 1080, 97), (48, 27, 16, 5, 0, 27, 1))),
                          (pt.glass_grid_scene(10), ((1280, 720, 4, 31, 300, 330, 1), (64, 36, 8, 31, 0, 36, 1))),
                          (pt.parse_scene(_mixed_scene_text(120)), ((1000, 562, 4, 8, 250, 290, 1), (997, 561, 2, 8, 0, 561, 40)))):
        sc = pt.Scene(*scene)
        walk = pt.Scene(*scene)
        walk.tune(pt.binding.PT_TUNE_BVH_WALK, 2)
        assert sc.bvh_nodes > 0
        for w, h, spp, depth, r0, r1, step in frames:
            kw = dict(n_bounces=depth, row_begin=r0, row_end=r1, row_step=step, fb_compact=1)
            ref = sc.render(pt.make_config(w, h, spp, math_mode=pt.PT_MATH_FAST_SCALAR, **kw))
            assert float(ref.sum()) > 0.0
            for tile in (0, 3, 4, 5, 6):
                a = sc.render(pt.make_config(w, h, spp, math_mode=pt.PT_MATH_FAST, tile_log2=tile, **kw))
                assert torch.equal(a, ref), (w, h, spp, tile)
            assert torch.equal(walk.render(pt.make_config(w, h, spp, math_mode=pt.PT_MATH_FAST, **kw)), ref), (w, h, spp)
    # a camera inside the cloud of shapes (boxes around and behind the view point) and one far outside it
    sc = pt.Scene(*pt.synthetic_scene(5000, 4989, 12345))
    for camera in ((0.3, -0.2, -2.0), (0.0, 0.0, 6.5), (-0.95, 0.9, -1.0)):
        kw = dict(row_begin=300, row_end=312, fb_compact=1, camera=camera)
        ref = sc.render(pt.make_config(1920, 1080, 2, math_mode=pt.PT_MATH_FAST_SCALAR, **kw))
        assert float(ref.sum()) > 0.0
        assert torch.equal(sc.render(pt.make_config(1920, 1080, 2, math_mode=pt.PT_MATH_FAST, **kw)), ref), camera
    # path segments are counted the same way by both loops
    sc, walk = pt.Scene(*pt.synthetic_scene(700, 300, 7)), pt.Scene(*pt.synthetic_scene(700, 300, 7))
    walk.tune(pt.binding.PT_TUNE_BVH_WALK, 2)
    cfg = pt.make_config(640, 360, 4, math_mode=pt.PT_MATH_FAST)
    (fa, ca), (fb, cb) = sc.render_counted(cfg), walk.render_counted(cfg)
    assert torch.equal(fa, fb) and int(ca[0]) == int(cb[0]) and int(ca[0]) > 640 * 360 * 4


def test_small_far_quadrics_hit_through_the_hierarchy_as_through_the_loop():
    """A sphere of scale 0.01 seen from several units away is tested in a frame where the ray origin has coordinates of
    a few hundred: its apparent surface moves by a good part of its radius with the rounding of |o_l|^2, and the loop
    over all primitives hits it wherever it lies.  The boxes the hierarchy culls by are grown by that much at upload;
    before, one camera sample of the 132 million of this frame (row 497) went another way than in the loop."""
    import torch
    import pathtracer_amd as pt
    sc = pt.Scene(*pt.synthetic_scene(5000, 4989, 12345))
    walk = pt.Scene(*pt.synthetic_scene(5000, 4989, 12345))
    walk.tune(pt.binding.PT_TUNE_BVH_WALK, 2)
    kw = dict(row_begin=490, row_end=506, fb_compact=1)
    ref = sc.render(pt.make_config(1920, 1080, 64, math_mode=pt.PT_MATH_FAST_SCALAR, **kw))
    assert torch.equal(sc.render(pt.make_config(1920, 1080, 64, math_mode=pt.PT_MATH_FAST, **kw)), ref)
    assert torch.equal(walk.render(pt.make_config(1920, 1080, 64, math_mode=pt.PT_MATH_FAST, **kw)), ref)
    par = sc.render(pt.make_config(1920, 1080, 16, math_mode=pt.PT_MATH_PARITY, **kw))
    sc.tune(pt.binding.PT_TUNE_BVH_WALK, 3)
    assert torch.equal(sc.render(pt.make_config(1920, 1080, 16, math_mode=pt.PT_MATH_PARITY, **kw)), par)


def test_more_materials_and_primitives_than_the_lds_tables_hold(oracle):
    """Above 32 materials (or 64 primitives) the fast kernels gather shading records and materials from global memory
    instead of their LDS copies: same arithmetic, so the LDS-staged loop still matches bit for bit, and the parity-order
    kernel still matches the oracle."""
    import torch
    import pathtracer_amd as pt
    rng = np.random.default_rng(9)
    lines = ["material light emissive 20 18 14"]
    for m in range(40):
        c = rng.uniform(0.1, 0.9, 3)
        lines.append("material m%d albedo %.3f %.3f %.3f" % (m, c[0], c[1], c[2]))
    lines += ["plane m0 bounds -15 -15 15 15 rotate pi/2 1 0 0 translate 0 -1 0",
              "plane m1 bounds -15 -15 15 15 translate 0 0 -5",
              "plane light bounds -0.5 -0.5 0.5 0.5 rotate -pi/2 1 0 0 translate 0 0.97 -3.25"]
    for k in range(70):
        pos = rng.uniform([-0.9, -0.9, -4.8], [0.9, 0.9, -1.6])
        lines.append("%s m%d translate %.4f %.4f %.4f scale %.4f" % (("sphere", "cube")[k % 2], k % 40, pos[0], pos[1], pos[2], rng.uniform(0.05, 0.2)))
    prims, n, mats, nm = pt.parse_scene("\n".join(lines) + "\n")
    assert nm == 41 and n == 73
    sc = pt.Scene(prims, n, mats, nm, bvh_min_prims=1000000)      # the loop over all primitives, not the hierarchy
    assert sc.bvh_nodes == 0
    w, h, spp = 160, 90, 8
    a = sc.render(pt.make_config(w, h, spp, math_mode=pt.PT_MATH_FAST))
    b = sc.render(pt.make_config(w, h, spp, math_mode=pt.PT_MATH_FAST_LDS))
    assert torch.isfinite(a).all() and torch.equal(a, b)
    osc = oracle.Scene(*_as_oracle(oracle, prims, n, mats, nm))
    ref = osc.render(oracle.make_config(w, h, spp))
    par = sc.render(pt.make_config(w, h, spp, math_mode=pt.PT_MATH_PARITY)).cpu().numpy()
    rp, bp = _stats(par, ref)
    rf, bf = _stats(a.cpu().numpy(), ref)
    print("41 materials / 73 primitives: parity rmse %.2e bad %.5f | fast rmse %.2e bad %.5f" % (rp, bp, rf, bf))
    # At 8 samples per pixel one path that takes another branch (a last-bit difference of the device's sinf / cosf) moves
    # its pixel by 20 / 8 and the frame's RMSE by percents: such pixels are counted, and all others must agree closely.
    off = np.abs(par.astype(np.float64) - ref).max(axis=2) > 0.05
    assert off.mean() <= tol.PARITY["bad_pixel_frac"]
    rest = float(np.sqrt(((par.astype(np.float64) - ref)[~off] ** 2).mean()) / ref.mean())
    assert rest <= tol.PARITY_SMALL_FRAME_RMSE, rest
    assert bf <= tol.MANY_SMALL_SHAPES["fast_bad_pixel_frac"]


def test_direction_components_of_exactly_zero(oracle):
    """Rays with one or two direction components of exactly 0 (1/d infinite) that run beside, into and inside every
    shape type: both kernel families find what the oracle finds.  A fast cube test once computed its slab distances
    as inf - inf there and reported a hit for a ray passing beside the cube."""
    import torch
    import pathtracer_amd as pt
    text = """
material glow emissive 3 2 1
material dark albedo 0 0 0
cube glow      translate 0 -0.5 -4.2  scale 2.5 0.3 0.5
sphere glow    translate 2 0 -3       scale 0.8
cylinder glow  translate -2 0 -3      scale 0.6 0.6 1.2
sphere glow    translate 0 2 -3       scale 0.9 0.5 0.7
plane glow     bounds -0.5 -0.5 0.5 0.5  translate 0 -2 -3
"""
    prims, n, mats, nm = pt.parse_scene(text)
    rng = np.random.default_rng(77)
    N = 6000
    centres = np.array([[0, -0.5, -4.2], [2, 0, -3], [-2, 0, -3], [0, 2, -3], [0, -2, -3]], np.float32)
    o = (centres[rng.integers(0, 5, N)] + rng.uniform(-1.6, 1.6, (N, 3))).astype(np.float32)
    d = rng.normal(size=(N, 3)).astype(np.float32)
    zero = rng.integers(0, 3, N)
    d[np.arange(N), zero] = 0.0
    two = rng.random(N) < 0.3
    d[np.arange(N)[two], (zero[two] + 1) % 3] = 0.0
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    assert (d == 0).any(axis=1).all()
    # the ray of the original failure: beside the cube, d.y = 0
    o[0] = (1.0, 0.8554, -3.825); d[0] = (-0.9468, 0.0, -0.3217); d[0] /= np.linalg.norm(d[0])
    px = np.arange(N, dtype=np.int32); sm = np.zeros(N, np.int32)
    osc = oracle.Scene(*_as_oracle(oracle, prims, n, mats, nm))
    ref = osc.trace_paths(o, d, px.astype(np.uint32), sm.astype(np.uint32), seed=0, n_bounces=0)
    assert (ref.sum(axis=1) > 0).mean() > 0.03 and (ref.sum(axis=1) == 0).mean() > 0.1
    assert ref[0].sum() == 0
    for bvh_min in (2, 1000000):                  # through the hierarchy and through the loop
        sc = pt.Scene(prims, n, mats, nm, bvh_min_prims=bvh_min)
        for mode in (pt.PT_MATH_PARITY, pt.PT_MATH_FAST):
            rgb = sc.trace_paths(torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda(), torch.from_numpy(px).cuda(),
                                 torch.from_numpy(sm).cuda(), seed=0, n_bounces=0, math_mode=mode).cpu().numpy()
            differ = (rgb != ref).any(axis=1)
            # emission is added unscaled at the first hit, so a ray's radiance is (3,2,1) or 0: any difference is a
            # different hit decision; grazing rays may flip in the fast build
            assert differ.mean() <= (0.0 if mode == pt.PT_MATH_PARITY else 2e-3), (bvh_min, mode, differ.sum())
            assert not differ[0]


def _random_scene_text(rng, n_shapes, aligned=False):
    """Shapes of all four types under random rotations, non-uniform scales and translations; shape k glows with
    (k+1, 0.5, 0.25), so the radiance of a depth-0 path names the primitive it hit."""
    lines = []
    for k in range(n_shapes):
        lines.append("material m%d emissive %d 0.5 0.25" % (k, k + 1))
    for k in range(n_shapes):
        kind = ("plane", "cube", "cylinder", "sphere")[k % 4]
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        s = rng.uniform(0.3, 1.5, 3)
        if kind == "sphere" and k % 8 == 3:
            s[:] = s[0]                                   # uniformly scaled spheres take the world-space test
        t = rng.uniform(-3, 3, 3) + np.array([0, 0, -6.0])
        if aligned:
            # no rotation, or quarter turns about a world axis (planes then face along x, y or z); mirrored scales
            s *= rng.choice([-1.0, 1.0], 3)
            turn = ("", "rotate pi/2 1 0 0 ", "rotate -pi/2 0 1 0 ", "rotate pi 0 1 0 ")[int(rng.integers(0, 4))] if kind == "plane" else ""
            opt = "%stranslate %.5f %.5f %.5f scale %.5f %.5f %.5f" % (turn, t[0], t[1], t[2], s[0], s[1], s[2])
        else:
            opt = "rotate %.6f %.6f %.6f %.6f translate %.5f %.5f %.5f scale %.5f %.5f %.5f" % (rng.uniform(0, 6.28), ax[0], ax[1], ax[2], t[0], t[1], t[2], s[0], s[1], s[2])
        if kind == "plane":
            opt = "bounds -1.2 -0.7 0.9 1.4 " + opt
        lines.append("%s m%d %s" % (kind, k, opt))
    return "\n".join(lines) + "\n"


@pytest.mark.parametrize("n_shapes,seed,aligned", [(12, 1, False), (40, 2, False), (160, 3, False), (24, 4, True), (120, 5, True)])
def test_fast_hit_decisions_on_random_scenes(oracle, n_shapes, seed, aligned):
    """Which primitive a ray hits first, fast kernels (loop and hierarchy) against the oracle, on random affine shapes
    and random rays: the parity-order kernels are compared bit by bit elsewhere, this pins the restructured tests of
    the fast kernels (folded plane bounds, world-space spheres, slab cube, cylinder case analysis; with `aligned`, the
    axis-aligned plane and cube kinds, mirrored scales included)."""
    import torch
    import pathtracer_amd as pt
    rng = np.random.default_rng(seed)
    prims, n, mats, nm = pt.parse_scene(_random_scene_text(rng, n_shapes, aligned), max_prims=n_shapes, max_mats=n_shapes)
    assert n == n_shapes
    N = 20000
    o = (rng.uniform(-4, 4, (N, 3)) + np.array([0, 0, -6.0])).astype(np.float32)
    target = rng.uniform(-3.5, 3.5, (N, 3)) + np.array([0, 0, -6.0])
    d = target - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    px = np.arange(N, dtype=np.int32); sm = np.zeros(N, np.int32)
    osc = oracle.Scene(*_as_oracle(oracle, prims, n, mats, nm))
    ref = osc.trace_paths(o, d, px.astype(np.uint32), sm.astype(np.uint32), seed=0, n_bounces=0)
    hit_frac = (ref[:, 0] > 0).mean()
    assert hit_frac > 0.3
    assert len(np.unique(ref[:, 0])) >= min(n_shapes, 10)          # many different primitives are first hits
    for bvh_min in (2, 1000000):
        sc = pt.Scene(prims, n, mats, nm, bvh_min_prims=bvh_min)
        for mode in (pt.PT_MATH_PARITY, pt.PT_MATH_FAST):
            rgb = sc.trace_paths(torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda(), torch.from_numpy(px).cuda(),
                                 torch.from_numpy(sm).cuda(), seed=0, n_bounces=0, math_mode=mode).cpu().numpy()
            differ = (rgb != ref).any(axis=1)
            print("random scene %d shapes, bvh_min %s, mode %d: first hit differs for %d of %d rays" % (n_shapes, bvh_min, mode, differ.sum(), N))
            assert differ.mean() <= (0.0 if mode == pt.PT_MATH_PARITY else 1e-3), (bvh_min, mode, int(differ.sum()))
        # the render kernel's own loops (kind-sorted records, fat hierarchy) on camera rays: at depth 0 a sample's
        # radiance is the emission of what it hit, so an image pixel differs only if a sample hit something else
        if "ref_img" not in locals():
            ref_img = osc.render(oracle.make_config(160, 120, 4, n_bounces=0))
            assert (ref_img.sum(axis=2) > 0).mean() > 0.2
        img = sc.render(pt.make_config(160, 120, 4, n_bounces=0, math_mode=pt.PT_MATH_FAST)).cpu().numpy()
        off = (np.abs(img - ref_img).max(axis=2) > 1e-5).mean()
        print("random scene %d shapes, bvh_min %s: depth-0 image differs in %.5f of the pixels" % (n_shapes, bvh_min, off))
        assert off <= 2e-3
