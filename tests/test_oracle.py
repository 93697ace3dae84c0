"""CPU tests of the oracle: the reference's own known answers
(tests/shape_test.cc), hand-derivable distances for the same rays, and the
committed golden vectors.  No GPU."""
import os
import subprocess

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
S3 = np.float32(1.0) / np.sqrt(np.float32(3.0))


def _diag():
    d = np.array([[1, 1, 1]], np.float32)
    return (d / np.linalg.norm(d)).astype(np.float32)


# (shape, translation, ray origin, ray direction, expected n_intersections, shape_test.cc line)
def _kat_rows(ob):
    z = (0, 0, 0)
    diag = _diag()[0]
    return [
        (ob.SPHERE, z, z, diag, 1, 27), (ob.SPHERE, (0, 10, 0), z, diag, 0, 28), (ob.SPHERE, (5, 5, 5), z, diag, 2, 29),
        (ob.CUBE, z, z, diag, 1, 72), (ob.CUBE, (0, 10, 0), z, diag, 0, 73), (ob.CUBE, (5, 5, 5), z, diag, 2, 74),
        (ob.CYLINDER, z, z, (0, 0, -1), 1, 97), (ob.CYLINDER, (0, 10, 0), z, (0, 0, -1), 0, 98),
        (ob.CYLINDER, (0, 0, -5), z, (0, 0, -1), 2, 99),
        (ob.CYLINDER, z, z, (0, 1, 0), 1, 105), (ob.CYLINDER, (0, 10, 0), z, (0, 1, 0), 2, 106),
        (ob.CYLINDER, (0, 0, -5), z, (0, 1, 0), 0, 107),
    ]


def test_reference_known_answers(oracle):
    ob = oracle
    for ty, tr, o, d, expect, line in _kat_rows(ob):
        r = ob.raypath(ob.make_prim(ty, translate=tr), np.array([o], np.float32), np.array([d], np.float32))
        assert r["n"][0] == expect, "shape_test.cc:%d" % line
    o = np.array([[0, 0, -5]], np.float32)
    assert ob.raypath(ob.make_prim(ob.PLANE), o, _diag())["n"][0] == 1            # shape_test.cc:47
    assert ob.raypath(ob.make_prim(ob.PLANE, bounds=(0, 0, 2, 2)), o, _diag())["n"][0] == 0   # shape_test.cc:48


def test_analytic_distances(oracle):
    ob = oracle
    z = np.zeros((1, 3), np.float32)
    r = ob.raypath(ob.make_prim(ob.SPHERE, translate=(5, 5, 5)), z, _diag())
    np.testing.assert_allclose([r["t1"][0], r["t2"][0]], [5 * np.sqrt(3) - 0.5, 5 * np.sqrt(3) + 0.5], rtol=3e-6)
    r = ob.raypath(ob.make_prim(ob.CUBE, translate=(5, 5, 5)), z, _diag())
    np.testing.assert_allclose([r["t1"][0], r["t2"][0]], [4.5 * np.sqrt(3), 5.5 * np.sqrt(3)], rtol=3e-6)
    r = ob.raypath(ob.make_prim(ob.PLANE), np.array([[0, 0, -5]], np.float32), _diag())
    np.testing.assert_allclose(r["t1"][0], 5 * np.sqrt(3), rtol=3e-6)
    r = ob.raypath(ob.make_prim(ob.CYLINDER, translate=(0, 0, -5)), z, np.array([[0, 0, -1]], np.float32))
    assert (r["t1"][0], r["t2"][0]) == (4.5, 5.5)
    r = ob.raypath(ob.make_prim(ob.CYLINDER, translate=(0, 10, 0)), z, np.array([[0, 1, 0]], np.float32))
    assert (r["t1"][0], r["t2"][0]) == (9.5, 10.5)
    # origin inside the sphere: the negative root is filtered and first = second (shapes.h:67-69)
    r = ob.raypath(ob.make_prim(ob.SPHERE), z, _diag())
    assert r["n"][0] == 1 and abs(r["t1"][0] - 0.5) < 1e-6 and r["hit2"][0] == 0


def test_plane_edge_cases(oracle):
    ob = oracle
    # parallel ray (local d.z == 0): the reference leaves n_intersections unset; defined as 0
    r = ob.raypath(ob.make_prim(ob.PLANE), np.array([[0, 0, 1]], np.float32), np.array([[1, 0, 0]], np.float32))
    assert r["n"][0] == 0
    # strict bounds: a hit exactly on xMax is outside (shapes.h:123-124)
    r = ob.raypath(ob.make_prim(ob.PLANE, bounds=(-1, -1, 1, 1)), np.array([[1, 0, 1]], np.float32), np.array([[0, 0, -1]], np.float32))
    assert r["n"][0] == 0
    # hit behind the origin is dropped by the epsilon filter (shapes.h:62-63)
    r = ob.raypath(ob.make_prim(ob.PLANE), np.array([[0, 0, 1]], np.float32), np.array([[0, 0, 1]], np.float32))
    assert r["n"][0] == 0


def test_rng_range_and_golden(oracle):
    ob = oracle
    g = np.load(os.path.join(GOLD, "rng_golden.npz"))
    for key in [k for k in g.files if k.startswith("bits_")]:
        seed, pixel, sample = (int(x) for x in key.split("_")[1:])
        bits, u = ob.rng(seed, pixel, sample, 64)
        assert np.array_equal(bits, g[key])
        assert np.array_equal(u.view(np.uint32), g["u_%d_%d_%d" % (seed, pixel, sample)].view(np.uint32))
    # (0,1]: 0 excluded, 1 included, like curand_uniform (main.cu:62)
    u = np.concatenate([ob.rng(3, p, 0, 256)[1] for p in range(64)])
    assert u.min() > 0.0 and u.max() <= 1.0
    assert abs(u.mean() - 0.5) < 0.01
    # slot map: a branch decision takes the 32 bits of its slot, the coordinates of a point the 16-bit halves of the
    # pair's first slot (jitter 0,1; segment s: 2+6s+{1,2} and 2+6s+{3,4})
    bits, u = ob.rng(5, 77, 3, 62)
    half = lambda h: np.float32(h) * np.float32(2.0 ** -16) + np.float32(2.0 ** -17)
    wide = lambda b: np.float32(b) * np.float32(2.0 ** -32) + np.float32(2.0 ** -33)
    for k in range(62):
        j = k + 1 if k < 2 else (k - 2) % 6
        want = half(bits[k] >> 16) if j in (1, 3) else (half(bits[k - 1] & 0xffff) if j in (2, 4) else wide(bits[k]))
        assert u[k] == want, (k, j)
    # the halves are uniform too, and the two coordinates of a point are not correlated
    bits, u = zip(*[ob.rng(9, p, s, 8) for p in range(128) for s in range(16)])
    u = np.array(u)
    assert abs(u[:, 3].mean() - 0.5) < 0.02 and abs(u[:, 4].mean() - 0.5) < 0.02
    assert abs(np.corrcoef(u[:, 3], u[:, 4])[0, 1]) < 0.06 and abs(np.corrcoef(u[:, 0], u[:, 1])[0, 1]) < 0.06


def test_raypath_golden(oracle):
    ob = oracle
    g = np.load(os.path.join(GOLD, "raypath_golden.npz"))
    for name in g["names"]:
        prim = ob.make_prim(int(g[name + "/type"]), transform=g[name + "/transform"].tolist(), bounds=tuple(g[name + "/bounds"].tolist()))
        r = ob.raypath(prim, g[name + "/o"], g[name + "/d"])
        for k in ("n", "hit1", "hit2"):
            assert np.array_equal(r[k], g[name + "/" + k]), (name, k)
        for k in ("t1", "t2", "pos1", "nrm1", "pos2", "nrm2"):
            assert np.array_equal(r[k].view(np.uint32), g[name + "/" + k].view(np.uint32)), (name, k)


def test_default_scene_golden(oracle):
    ob = oracle
    prims, n, mats, nm = ob.default_scene()
    assert (n, nm) == (11, 8)
    sc = ob.Scene(prims, n, mats, nm)
    g = np.load(os.path.join(GOLD, "default_scene_paths.npz"))
    near = sc.nearest(g["o"], g["d"])
    assert np.array_equal(near["prim"], g["prim"]) and np.array_equal(near["inside"], g["inside"])
    assert np.array_equal(near["t"].view(np.uint32), g["t"].view(np.uint32))
    # the box is closed: every ray hits something
    assert (near["prim"] >= 0).all()
    rgb = sc.trace_paths(g["o"], g["d"], g["pixel"], g["sample"])
    np.testing.assert_allclose(rgb, g["rgb"], rtol=1e-5, atol=1e-6)


def test_alternative_bsdf_golden(oracle):
    """tracePath of bsdf_alternative.h:153-219 in the oracle: committed vectors and what the text implies."""
    ob = oracle
    sc = ob.Scene(*ob.default_scene())
    g = np.load(os.path.join(GOLD, "default_scene_paths.npz"))
    a = np.load(os.path.join(GOLD, "alternative_bsdf.npz"))
    rgb = sc.trace_paths(g["o"], g["d"], g["pixel"], g["sample"], seed=0, n_bounces=5, bsdf_model=1)
    np.testing.assert_allclose(rgb, a["rgb"], rtol=1e-5, atol=1e-6)
    fb = sc.render(ob.make_config(96, 64, 32, bsdf_model=1))
    np.testing.assert_allclose(fb, a["fb"], rtol=1e-5, atol=1e-6)
    # depth 0: only light seen directly, the same under both models
    r0 = sc.trace_paths(g["o"], g["d"], g["pixel"], g["sample"], n_bounces=0, bsdf_model=0)
    r1 = sc.trace_paths(g["o"], g["d"], g["pixel"], g["sample"], n_bounces=0, bsdf_model=1)
    assert np.array_equal(r0, r1) and r0.max() == 20.0
    # Lambertian weight f/pdf is the albedo (N.L <= 1) and the roulette renormalises by its maximum, so a path that
    # reaches the light after diffuse bounces carries emissive x (products of albedo / max albedo): for the grey,
    # red and green walls every channel of the radiance is 0 or one of the light's (20, 18, 14)
    lit = rgb[(rgb.max(axis=1) > 0) & (g["prim"] != 5) & (g["prim"] < 7)]
    assert len(lit) > 10
    assert np.all((np.abs(lit) < 1e-6) | (np.abs(lit - np.array([20.0, 18.0, 14.0], np.float32)) < 1e-3))
    # the class model has no Beer term and other weights: it is not the bsdf.h image
    fb0 = sc.render(ob.make_config(96, 64, 32, bsdf_model=0))
    assert abs(fb0.mean() - fb.mean()) > 0.05


def test_host_compiler_noise_floor(oracle):
    """The oracle built by g++ and by ROCm's clang++ (both -ffp-contract=off): how much of a difference between
    kernel and oracle could be the host compiler's doing?  Intersection distances and hit records must agree bit for
    bit (only +, -, *, /, sqrt, all correctly rounded); radiance goes through libm's sinf/cosf/expf, the same library
    for both builds, so it agrees too."""
    ob = oracle
    try:
        ob.lib(variant="clang")
    except (FileNotFoundError, OSError, subprocess.CalledProcessError):
        pytest.skip("no clang++ build of the oracle (oracle/Makefile builds it when /opt/rocm/lib/llvm/bin/clang++ exists)")
    prims, n, mats, nm = ob.default_scene()
    a, b = ob.Scene(prims, n, mats, nm), ob.Scene(prims, n, mats, nm, variant="clang")
    g = np.load(os.path.join(GOLD, "default_scene_paths.npz"))
    na, nb = a.nearest(g["o"], g["d"]), b.nearest(g["o"], g["d"])
    for k in ("prim", "inside"):
        assert np.array_equal(na[k], nb[k])
    for k in ("t", "pos", "nrm"):
        assert np.array_equal(na[k].view(np.uint32), nb[k].view(np.uint32)), k
    ra = a.trace_paths(g["o"], g["d"], g["pixel"], g["sample"])
    rb = b.trace_paths(g["o"], g["d"], g["pixel"], g["sample"])
    assert np.array_equal(ra.view(np.uint32), rb.view(np.uint32))
    cfg = ob.make_config(96, 64, 8)
    assert np.array_equal(a.render(cfg).view(np.uint32), b.render(cfg).view(np.uint32))


def test_render_golden_and_tiling(oracle):
    ob = oracle
    prims, n, mats, nm = ob.default_scene()
    sc = ob.Scene(prims, n, mats, nm)
    gold = np.load(os.path.join(GOLD, "default_256x256_16spp.npz"))["fb"]
    # rows 100..116 only (keeps the CPU suite short); tiles are independent of the split
    cfg = ob.make_config(256, 256, 16, row_begin=100, row_end=116)
    fb = sc.render(cfg)
    np.testing.assert_allclose(fb[100:116], gold[100:116], rtol=1e-5, atol=1e-6)
    assert not fb[:100].any() and not fb[116:].any()
    # camera quirk: integer nx/ny makes focal_length identical for every named resolution (main.cu:230)
    assert ob.focal_length(1920, 1080) == ob.focal_length(256, 256) == pytest.approx(6.0355339, rel=1e-6)


def test_flop_counter_matches_event_table(oracle):
    ob = oracle
    prims, n, mats, nm = ob.default_scene()
    sc = ob.Scene(prims, n, mats, nm, count=True)
    cfg = ob.make_config(64, 64, 4)
    fb, st = sc.render(cfg, want_stats=True)
    assert st.samples == 64 * 64 * 4
    assert st.tests[0] == 7 * st.segments and st.tests[3] == 3 * st.segments and st.tests[1] == st.segments
    assert sum(st.mode) == st.hits
    per_sample = st.flops / st.samples
    assert 1000 < per_sample < 2000   # a-priori estimate 1.3-1.8 kflop/sample
    # counters do not change the image
    fb2 = ob.Scene(prims, n, mats, nm).render(cfg)
    assert np.array_equal(fb, fb2)


def test_ppm_matches_reference_format(oracle):
    ob = oracle
    fb = np.array([[[0.0, 0.5, 1.0], [2.0, 0.25, 0.999]], [[1e-3, 0.1, 0.2], [0.3, 0.4, 0.5]]], np.float32)
    txt = ob.ppm_text(fb).decode()
    lines = txt.split("\n")
    assert lines[:3] == ["P3", "2 2", "255"]
    # top row of the file is the LAST framebuffer row (main.cu:241); values are int(255.99*c), unclamped
    assert lines[3] == "0 25 51" and lines[4] == "76 102 127"
    assert lines[5] == "0 127 255" and lines[6] == "511 63 255"
