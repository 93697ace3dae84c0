"""GPU parity tests: every call goes through the C ABI of libpt_hip.so and is
checked against the CPU oracle on the same seeded inputs and against the
committed golden vectors.

Stated bounds
  * RNG bits and uniforms: bit-exact.
  * parity-order kernels, intersection distances t: <= 2 ULP (measured 0), hit
    counts exact.
  * parity-order kernels, normals/positions: <= 4 ULP of the largest component.
  * per-path radiance (tracePath): identical up to 1e-4 relative for >= 99.5 % of
    paths; the rest are Monte-Carlo branch flips (a 1-ULP change of sinf/cosf
    flips a Russian-roulette or Fresnel draw).
  * images, same seed: per-channel RMSE <= 1 % of the mean radiance for the
    parity build and <= 2 % for the fast build, at most 0.1 % (parity) / 0.5 %
    (fast) of pixels off by more than 0.05, mean radiance within 0.5 %.
"""
import os

import numpy as np
import pytest

import tolerances as tol

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def ulp_diff(a, b):
    """Distance in units in the last place between two float32 arrays (same sign assumed for non-zero values)."""
    ai = a.astype(np.float32).view(np.int32).astype(np.int64)
    bi = b.astype(np.float32).view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7FFFFFFF), ai)
    bi = np.where(bi < 0, -(bi & 0x7FFFFFFF), bi)
    return np.abs(ai - bi)


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def image_stats(img, ref):
    img = np.asarray(img, np.float64); ref = np.asarray(ref, np.float64)
    mean = ref.mean()
    rmse = np.sqrt(((img - ref) ** 2).mean(axis=(0, 1)))
    bad = (np.abs(img - ref).max(axis=2) > 0.05).mean()
    mean_rel = np.abs(img.mean(axis=(0, 1)) - ref.mean(axis=(0, 1))) / ref.mean(axis=(0, 1))
    return dict(rmse_rel=float(rmse.max() / mean), bad_frac=float(bad), mean_rel=float(mean_rel.max()))


def test_rng_bit_exact(torch, pt, oracle):
    for seed, pixel, sample in ((0, 0, 0), (0, 12345, 7), (1, 2073599, 63), (12345, 8294399, 1023)):
        bits, u = pt.rng_draws(seed, pixel, sample, 200)
        ob, ou = oracle.rng(seed, pixel, sample, 200)
        assert np.array_equal(bits.cpu().numpy().view(np.uint32), ob)
        assert np.array_equal(u.cpu().numpy().view(np.uint32), ou.view(np.uint32))


def test_reference_known_answers_on_gpu(torch, pt, oracle):
    """The 14 rows of tests/shape_test.cc through pt_raypath_batch."""
    s3 = np.array([1, 1, 1], np.float32); s3 = s3 / np.linalg.norm(s3)
    rows = [
        (pt.PT_SPHERE, (0, 0, 0), (0, 0, 0), s3, 1), (pt.PT_SPHERE, (0, 10, 0), (0, 0, 0), s3, 0), (pt.PT_SPHERE, (5, 5, 5), (0, 0, 0), s3, 2),
        (pt.PT_CUBE, (0, 0, 0), (0, 0, 0), s3, 1), (pt.PT_CUBE, (0, 10, 0), (0, 0, 0), s3, 0), (pt.PT_CUBE, (5, 5, 5), (0, 0, 0), s3, 2),
        (pt.PT_CYLINDER, (0, 0, 0), (0, 0, 0), (0, 0, -1), 1), (pt.PT_CYLINDER, (0, 10, 0), (0, 0, 0), (0, 0, -1), 0),
        (pt.PT_CYLINDER, (0, 0, -5), (0, 0, 0), (0, 0, -1), 2), (pt.PT_CYLINDER, (0, 0, 0), (0, 0, 0), (0, 1, 0), 1),
        (pt.PT_CYLINDER, (0, 10, 0), (0, 0, 0), (0, 1, 0), 2), (pt.PT_CYLINDER, (0, 0, -5), (0, 0, 0), (0, 1, 0), 0),
        (pt.PT_PLANE, (0, 0, 0), (0, 0, -5), s3, 1),
    ]
    prims = (pt.pt_prim * (len(rows) + 1))()
    mats = (pt.pt_material * 1)()
    big = float(np.finfo(np.float32).max)
    for k, (ty, tr, _, _, _) in enumerate(rows):
        prims[k].type, prims[k].material = ty, 0
        prims[k].transform[:] = [1, 0, 0, tr[0], 0, 1, 0, tr[1], 0, 0, 1, tr[2]]
        prims[k].bounds[:] = (-big, -big, big, big)
    k = len(rows)
    prims[k].type, prims[k].material = pt.PT_PLANE, 0
    prims[k].transform[:] = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0]
    prims[k].bounds[:] = (0, 0, 2, 2)
    rows.append((pt.PT_PLANE, (0, 0, 0), (0, 0, -5), s3, 0))
    sc = pt.Scene(prims, len(rows), mats, 1)
    for k, (_, _, o, d, expect) in enumerate(rows):
        r = sc.raypath(k, dev(torch, np.array([o], np.float32)), dev(torch, np.array([d], np.float32)))
        assert int(r["n"][0]) == expect, "row %d" % k


def test_raypath_parity_vs_golden_and_oracle(torch, pt, oracle):
    g = np.load(os.path.join(GOLD, "raypath_golden.npz"))
    names = list(g["names"])
    prims = (pt.pt_prim * len(names))()
    mats = (pt.pt_material * 1)()
    for k, name in enumerate(names):
        prims[k].type, prims[k].material = int(g[name + "/type"]), 0
        prims[k].transform[:] = g[name + "/transform"].tolist()
        prims[k].bounds[:] = g[name + "/bounds"].tolist()
    sc = pt.Scene(prims, len(names), mats, 1)
    worst_t, worst_v = 0, 0
    for k, name in enumerate(names):
        o, d = g[name + "/o"], g[name + "/d"]
        r = {key: v.cpu().numpy() for key, v in sc.raypath(k, dev(torch, o), dev(torch, d)).items()}
        assert np.array_equal(r["n"], g[name + "/n"]), name
        assert np.array_equal(r["hit1"], g[name + "/hit1"]) and np.array_equal(r["hit2"], g[name + "/hit2"]), name
        for s in ("1", "2"):
            m = g[name + "/hit" + s] == 1
            if m.any():
                worst_t = max(worst_t, int(ulp_diff(r["t" + s][m], g[name + "/t" + s][m]).max()))
                for key in ("pos", "nrm"):
                    a, b = r[key + s][m], g[name + "/" + key + s][m]
                    scale = np.abs(b).max(axis=1, keepdims=True)
                    err = np.abs(a - b) / (scale * np.float32(2.0 ** -23))
                    worst_v = max(worst_v, float(err.max()))
    print("raypath: worst t ULP %d, worst pos/nrm error %.2f ULP-of-max-component" % (worst_t, worst_v))
    assert worst_t <= 2
    assert worst_v <= 4.0


def test_raypath_local_mode_and_random_rays(torch, pt, oracle):
    """getRayPath (no transform, no epsilon filter) and a larger random batch against the live oracle."""
    rng = np.random.default_rng(7)
    prims, n, mats, nm = pt.default_scene()
    sc = pt.Scene(prims, n, mats, nm)
    oprims, _, _, _ = oracle.default_scene()
    nr = 1 << 16
    o = rng.uniform(-1.0, 1.0, (nr, 3)).astype(np.float32)
    d = rng.normal(size=(nr, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    for k in (0, 5, 7, 10):
        for filt in (True, False):
            r = {key: v.cpu().numpy() for key, v in sc.raypath(k, dev(torch, o), dev(torch, d), filter=filt).items()}
            ref = oracle.raypath(oprims[k], o, d, filter=filt)
            assert np.array_equal(r["n"], ref["n"]), (k, filt)
            for s in ("1", "2"):
                m = ref["hit" + s] == 1
                assert np.array_equal(r["hit" + s], ref["hit" + s])
                if m.any():
                    assert ulp_diff(r["t" + s][m], ref["t" + s][m]).max() <= 2, (k, filt, s)


def test_nearest_parity(torch, pt, oracle):
    g = np.load(os.path.join(GOLD, "default_scene_paths.npz"))
    prims, n, mats, nm = pt.default_scene()
    sc = pt.Scene(prims, n, mats, nm)
    r = {key: v.cpu().numpy() for key, v in sc.nearest(dev(torch, g["o"]), dev(torch, g["d"])).items()}
    assert np.array_equal(r["prim"], g["prim"])
    assert np.array_equal(r["inside"], g["inside"])
    assert ulp_diff(r["t"], g["t"]).max() <= 2
    np.testing.assert_allclose(r["pos"], g["pos"], rtol=0, atol=4e-6)
    np.testing.assert_allclose(r["nrm"], g["nrm"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("mode", ["parity", "fast"])
def test_trace_paths_parity(torch, pt, oracle, mode):
    g = np.load(os.path.join(GOLD, "default_scene_paths.npz"))
    prims, n, mats, nm = pt.default_scene()
    sc = pt.Scene(prims, n, mats, nm)
    mm = pt.PT_MATH_PARITY if mode == "parity" else pt.PT_MATH_FAST
    rgb = sc.trace_paths(dev(torch, g["o"]), dev(torch, g["d"]), dev(torch, g["pixel"].astype(np.int32)).view(torch.int32),
                         dev(torch, g["sample"].astype(np.int32)).view(torch.int32), seed=0, n_bounces=5, math_mode=mm).cpu().numpy()
    ref = g["rgb"]
    tol = 1e-4 if mode == "parity" else 2e-3
    same = np.all(np.abs(rgb - ref) <= tol * np.maximum(np.abs(ref), 1e-3), axis=1)
    frac = same.mean()
    print("tracePath %s: %.4f of paths agree within %.0e relative; mean radiance %s vs %s" % (mode, frac, tol, rgb.mean(0), ref.mean(0)))
    assert frac >= (0.995 if mode == "parity" else 0.98)


@pytest.mark.parametrize("mode", ["parity", "fast"])
def test_render_256_vs_golden(torch, pt, oracle, mode):
    """Config 1: default scene, 256x256, 16 spp, seed 0, against the oracle's committed image."""
    gold = np.load(os.path.join(GOLD, "default_256x256_16spp.npz"))["fb"]
    prims, n, mats, nm = pt.default_scene()
    sc = pt.Scene(prims, n, mats, nm)
    mm = pt.PT_MATH_PARITY if mode == "parity" else pt.PT_MATH_FAST
    cfg = pt.make_config(256, 256, 16, math_mode=mm)
    img = sc.render(cfg).cpu().numpy()
    st = image_stats(img, gold)
    print("render 256x256@16 %s: %s" % (mode, st))
    assert np.isfinite(img).all()
    assert st["rmse_rel"] <= (tol.PARITY["image_rmse"] if mode == "parity" else tol.fast_image_rmse(16))
    assert st["bad_frac"] <= (0.001 if mode == "parity" else 0.005)
    assert st["mean_rel"] <= tol.FAST["mean_rel"]


def test_render_is_deterministic_and_tiling_independent(torch, pt):
    prims, n, mats, nm = pt.default_scene()
    sc = pt.Scene(prims, n, mats, nm)
    for mm in (pt.PT_MATH_PARITY, pt.PT_MATH_FAST):
        cfg = pt.make_config(200, 120, 8, math_mode=mm)      # width not a multiple of 64: tiles wrap rows
        a = sc.render(cfg)
        b = sc.render(cfg)
        assert torch.equal(a, b)
        # rows split as 2, 4 and 8 GPUs would split them (120 = 8*15): bitwise the same image
        for parts in (2, 4, 8):
            out = torch.zeros_like(a)
            rows = 120 // parts
            for r in range(parts):
                c = pt.make_config(200, 120, 8, math_mode=mm, row_begin=r * rows, row_end=(r + 1) * rows)
                sc.render(c, out=out)
            assert torch.equal(a, out), (mm, parts)
        # SoA planes hold the same numbers as the AoS framebuffer
        c = pt.make_config(200, 120, 8, math_mode=mm, fb_layout=pt.PT_FB_SOA)
        soa = sc.render(c)
        assert torch.equal(soa.permute(1, 2, 0).contiguous(), a)
        # interleaved rows (rank r takes rows r, r+N, ...), written in place and written compactly
        for parts in (2, 3, 8):
            out = torch.zeros_like(a)
            for r in range(parts):
                sc.render(pt.make_config(200, 120, 8, math_mode=mm, row_begin=r, row_step=parts), out=out)
            assert torch.equal(a, out), (mm, parts, "interleaved in place")
            for r in range(parts):
                c = pt.make_config(200, 120, 8, math_mode=mm, row_begin=r, row_step=parts, fb_compact=1)
                comp = sc.render(c)
                assert comp.shape[0] == len(range(r, 120, parts))
                assert torch.equal(comp, a[r::parts]), (mm, parts, "compact")
        # pixels per wave (4 .. 64) do not change a single bit: pixel sums are exact integer sums
        for lg in (2, 3, 4, 5, 6):
            assert torch.equal(sc.render(pt.make_config(200, 120, 8, math_mode=mm, tile_log2=lg)), a), (mm, lg)
        for bad in (1, 7, -3):
            with pytest.raises(pt.PtError):
                sc.render(pt.make_config(200, 120, 8, math_mode=mm, tile_log2=bad))


def test_render_edge_cases(torch, pt, oracle):
    prims, n, mats, nm = pt.default_scene()
    sc = pt.Scene(prims, n, mats, nm)
    osc = oracle.Scene(*oracle.default_scene())
    # empty tile: nothing is written
    cfg = pt.make_config(64, 64, 4, row_begin=10, row_end=10)
    out = torch.full((64, 64, 3), 7.0, device="cuda")
    sc.render(cfg, out=out)
    assert (out == 7.0).all()
    # 1x1 image and a 3-pixel-wide image (ragged tiles), 1 spp and depth 0 (one segment)
    for (w, h, spp, nb) in ((1, 1, 1, 0), (3, 5, 2, 1), (65, 2, 3, 5)):
        cfg = pt.make_config(w, h, spp, n_bounces=nb, math_mode=pt.PT_MATH_PARITY)
        img = sc.render(cfg).cpu().numpy()
        ref = osc.render(oracle.make_config(w, h, spp, n_bounces=nb))
        np.testing.assert_allclose(img, ref, rtol=1e-4, atol=1e-5)
    # bad arguments come back as errors in the reference's message shape, not as crashes
    cfg = pt.make_config(64, 64, 4, row_begin=0, row_end=65)
    with pytest.raises(pt.PtError, match="HIP error = "):
        sc.render(cfg)


def test_render_1080p_properties(torch, pt):
    """Full-size config 2 (1920x1080 @ 64 spp): size-independent properties — finite, closed-box energy
    statistics stable between the two math modes, bitwise reproducible, and equal to a two-tile render."""
    prims, n, mats, nm = pt.default_scene()
    sc = pt.Scene(prims, n, mats, nm)
    cfg = pt.make_config(1920, 1080, 64, math_mode=pt.PT_MATH_FAST)
    a = sc.render(cfg)
    assert torch.isfinite(a).all() and (a >= 0).all()
    out = torch.zeros_like(a)
    sc.render(pt.make_config(1920, 1080, 64, math_mode=pt.PT_MATH_FAST, row_begin=0, row_end=540), out=out)
    sc.render(pt.make_config(1920, 1080, 64, math_mode=pt.PT_MATH_FAST, row_begin=540, row_end=1080), out=out)
    assert torch.equal(a, out)
    p = sc.render(pt.make_config(1920, 1080, 64, math_mode=pt.PT_MATH_PARITY))
    ma, mp = a.mean(dim=(0, 1)).cpu().numpy(), p.mean(dim=(0, 1)).cpu().numpy()
    print("1080p mean radiance fast %s parity %s" % (ma, mp))
    np.testing.assert_allclose(ma, mp, rtol=tol.FAST["mean_rel"])
    bad = ((a - p).abs().amax(dim=2) > 0.05).float().mean().item()
    print("1080p fast vs parity: fraction of pixels off by > 0.05: %.5f" % bad)
    assert bad <= 0.005


def test_alternative_bsdf_model(torch, pt, oracle):
    """tracePath of bsdf_alternative.h:153-219 (Lambertian / SmoothDielectric by Material::bsdfType), which the
    reference ships but does not compile: parity-order kernel against the oracle's restatement path by path and as
    an image, fast kernel against the same image statistically."""
    prims, n, mats, nm = pt.default_scene()
    sc = pt.Scene(prims, n, mats, nm)
    osc = oracle.Scene(*oracle.default_scene())
    rng = np.random.default_rng(21)
    nr = 8192
    o = np.zeros((nr, 3), np.float32)
    d = rng.normal(size=(nr, 3)); d[:, 2] = -np.abs(d[:, 2]) - 0.5
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    px = rng.integers(0, 1 << 20, nr).astype(np.uint32); smp = rng.integers(0, 64, nr).astype(np.uint32)
    ref = osc.trace_paths(o, d, px, smp, seed=3, n_bounces=5, bsdf_model=1)
    args = [torch.from_numpy(x).cuda() for x in (o, d, px, smp)]
    got = sc.trace_paths(*args, seed=3, n_bounces=5, math_mode=pt.PT_MATH_PARITY, bsdf_model=pt.PT_BSDF_CLASSES).cpu().numpy()
    assert np.isfinite(got).all()
    close = np.abs(got - ref).max(axis=1) <= 1e-4 * (1.0 + np.abs(ref).max(axis=1))
    assert close.mean() >= 0.999, close.mean()
    assert (ref.max(axis=1) > 0).mean() > 0.005           # some paths reach the light (half the Lambertian samples die per bounce)
    # the model differs from bsdf.h's: same rays, different radiance
    ref0 = osc.trace_paths(o, d, px, smp, seed=3, n_bounces=5, bsdf_model=0)
    assert np.abs(ref0 - ref).max() > 1.0
    # images
    w, h, spp = 160, 100, 16
    img_ref = osc.render(oracle.make_config(w, h, spp, bsdf_model=1))
    par = sc.render(pt.make_config(w, h, spp, math_mode=pt.PT_MATH_PARITY, bsdf_model=pt.PT_BSDF_CLASSES)).cpu().numpy()
    fast = sc.render(pt.make_config(w, h, spp, math_mode=pt.PT_MATH_FAST, bsdf_model=pt.PT_BSDF_CLASSES)).cpu().numpy()
    mean = img_ref.mean()
    rmse_p = np.sqrt(((par - img_ref) ** 2).mean()) / mean
    bad_p = (np.abs(par - img_ref).max(axis=2) > 0.05).mean()
    rmse_f = np.sqrt(((fast - img_ref) ** 2).mean()) / mean
    bad_f = (np.abs(fast - img_ref).max(axis=2) > 0.05).mean()
    print("alternative bsdf: parity rmse %.2e bad %.5f | fast rmse %.2e bad %.5f" % (rmse_p, bad_p, rmse_f, bad_f))
    assert rmse_p <= tol.PARITY["image_rmse"] and bad_p <= tol.PARITY["bad_pixel_frac"]   # the bar of the bsdf.h path
    assert bad_f <= tol.CLASSES_BSDF["fast_bad_pixel_frac"]
    np.testing.assert_allclose(fast.mean(axis=(0, 1)), img_ref.mean(axis=(0, 1)), rtol=tol.CLASSES_BSDF["fast_mean_rel"])
    # LDS-staged loop, hierarchy and explicit rays run the same shading code
    a = sc.render(pt.make_config(w, h, spp, math_mode=pt.PT_MATH_FAST_SCALAR, bsdf_model=pt.PT_BSDF_CLASSES))
    b = sc.render(pt.make_config(w, h, spp, math_mode=pt.PT_MATH_FAST_LDS, bsdf_model=pt.PT_BSDF_CLASSES))
    assert torch.equal(a, b)
    big = pt.Scene(*pt.synthetic_scene(300, 200, 5))
    a = big.render(pt.make_config(96, 54, 4, math_mode=pt.PT_MATH_FAST, bsdf_model=pt.PT_BSDF_CLASSES))
    b = big.render(pt.make_config(96, 54, 4, math_mode=pt.PT_MATH_FAST_SCALAR, bsdf_model=pt.PT_BSDF_CLASSES))
    assert big.bvh_nodes > 0 and torch.equal(a, b)
    with pytest.raises(pt.PtError):
        sc.render(pt.make_config(8, 8, 1, bsdf_model=7))


@pytest.mark.parametrize("bsdf_model", [0, 1])
def test_trace_paths_end_ray(torch, pt, oracle, bsdf_model):
    """pt_trace_paths_end: the ray tracePath leaves in its `ray& r` (bsdf.h:165, bsdf_alternative.h:215) — the ray of
    the segment on which the path ended, or the ray after the last bounce — against the oracle, whose tracePath
    mutates r as the reference's does."""
    osc = oracle.Scene(*oracle.default_scene())
    sc = pt.Scene(*pt.default_scene())
    n = 8192
    cfg = oracle.make_config(256, 256, 16)
    rng = np.random.default_rng(3)
    pi = rng.integers(0, 256, n).astype(np.int32); pj = rng.integers(0, 256, n).astype(np.int32)
    smp = rng.integers(0, 16, n).astype(np.uint32)
    o, d = oracle.camera_rays(cfg, pi, pj, smp)
    px = (pj * 256 + pi).astype(np.uint32)
    for depth in (0, 1, 5):
        ref_rgb, ref_o, ref_d = oracle.trace_paths_end(osc, o, d, px, smp, n_bounces=depth, bsdf_model=bsdf_model)
        args = [torch.from_numpy(x).cuda() for x in (o, d, px.astype(np.int32), smp.astype(np.int32))]
        rgb, eo, ed = sc.trace_paths_end(*args, n_bounces=depth, math_mode=pt.PT_MATH_PARITY, bsdf_model=bsdf_model)
        rgb, eo, ed = rgb.cpu().numpy(), eo.cpu().numpy(), ed.cpu().numpy()
        same_path = np.abs(rgb - ref_rgb).max(axis=1) <= tol.PARITY["path_abs"]
        assert same_path.mean() >= tol.PARITY["path_share"]
        # where the path is the same path, its last ray is the same ray
        # (to 1e-5; a diffuse direction whose normal and sphere point nearly cancel amplifies the one-ULP difference
        # between the device's and the host's sinf/cosf, hence the share and the looser cap)
        err = np.maximum(np.abs(eo - ref_o), np.abs(ed - ref_d)).max(axis=1)[same_path]
        assert (err <= 1e-5).mean() >= 0.999 and err.max() <= 1e-3, (depth, float(err.max()))
        moved = (ref_o != o).any(axis=1)
        assert (depth == 0 and bsdf_model == 0 and moved.mean() > 0.3) or depth > 0 or bsdf_model == 1
        # the radiance is what pt_trace_paths returns
        fn_rgb = sc.trace_paths(*args, n_bounces=depth, math_mode=pt.PT_MATH_PARITY, bsdf_model=bsdf_model).cpu().numpy()
        assert (fn_rgb == rgb).all()
        # fast kernels: same call, end rays finite and unit length
        frgb, feo, fed = sc.trace_paths_end(*args, n_bounces=depth, math_mode=pt.PT_MATH_FAST, bsdf_model=bsdf_model)
        assert torch.isfinite(feo).all() and torch.isfinite(fed).all()
        ln = fed.norm(dim=1)            # unit length, or the zero vector that refract() leaves on total reflection (bsdf_alternative.h model)
        assert bool((((ln - 1).abs() < 1e-4) | (ln == 0)).all())
