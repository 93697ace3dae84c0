"""Randomised scenes: whatever the mixture of record kinds, the camera position and the depth, PT_MATH_FAST (camera
batches with per-tile culling, glass queue, loops compiled per kind set, hierarchy above its threshold) must give the
image of PT_MATH_FAST_SCALAR, the loop over the run table in scene order, bit for bit — a culled record that a camera
ray could hit, or a kind loop that reads the wrong offset, shows up here as a differing pixel."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MATERIALS = [
    "material base albedo 0.4 0.4 0.4", "material red albedo 0.9 0.1 0.1", "material light emissive 20 18 14",
    "material glow albedo 0.6 0.6 0.5 emissive 0.4 0.3 0.2",
    "material glass f0 0.02 specularColor 0.8 0.8 0.8 IOR 1.5 transparency 1 refractionColor 0.9 0.1 0",
    "material frosted f0 0.04 specularColor 0.9 0.9 0.9 IOR 1.3 transparency 0.7 refractionRoughness 0.3 specularRoughness 0.2 albedo 0.2 0.3 0.4",
    "material mirror f0 0.9 specularColor 0.9 0.9 0.9 specularRoughness 0.05",
]
NAMES = ["base", "red", "light", "glow", "glass", "frosted", "mirror"]


def random_scene(rng, n_shapes, p_rotated, walls):
    lines = list(MATERIALS)
    if walls:
        lines += ["plane base bounds -15 -15 15 15 rotate pi/2 1 0 0 translate 0 -1 0",
                  "plane red bounds -15 -15 15 15 rotate pi/2 0 1 0 translate -1.2 0 -5",
                  "plane base bounds -15 -15 15 15 translate 0 0 -6",
                  "plane light bounds -0.6 -0.6 0.6 0.6 rotate -pi/2 1 0 0 translate 0 1.2 -3"]
    if rng.random() < 0.5:
        lines.append("plane mirror bounds -1 -0.5 1 0.5 rotate %.4f 0 1 0 translate %.3f 0 -4.5" % (rng.uniform(-0.8, 0.8), rng.uniform(-0.5, 0.5)))
    for _ in range(n_shapes):
        kind = ("sphere", "cube", "cylinder", "sphere", "cube")[rng.integers(5)]
        mat = NAMES[rng.integers(len(NAMES))]
        pos = rng.uniform([-1.0, -0.9, -5.5], [1.0, 1.0, -1.2])
        parts = [kind, mat]
        rotated = rng.random() < p_rotated
        if rotated:
            ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
            parts.append("rotate %.4f %.4f %.4f %.4f" % (rng.uniform(0.1, 3.0), ax[0], ax[1], ax[2]))
        parts.append("translate %.4f %.4f %.4f" % tuple(pos))
        s = rng.uniform(0.08, 0.5, size=3)
        if kind == "sphere" and rng.random() < 0.6:
            parts.append("scale %.4f" % s[0])                       # uniformly scaled: the short sphere record (if not rotated)
        else:
            parts.append("scale %.4f %.4f %.4f" % tuple(s))
        lines.append(" ".join(parts))
    return "\n".join(lines) + "\n"


def test_fast_equals_the_run_table_loop_on_random_scenes(torch, pt):
    rng = np.random.default_rng(2024)
    seen_kinds = set()
    for trial in range(24):
        n_shapes = int(rng.integers(1, 60)) if trial % 5 else int(rng.integers(100, 140))     # every fifth scene has a hierarchy
        text = random_scene(rng, n_shapes, p_rotated=(0.0, 0.3, 1.0)[trial % 3], walls=trial % 4 != 3)
        sc = pt.Scene(*pt.parse_scene(text))
        seen_kinds.add((sc.bvh_nodes > 0, trial % 3))
        cam = tuple(float(x) for x in rng.uniform([-0.8, -0.7, -0.5], [0.8, 0.8, 3.0]))
        depth = int(rng.integers(0, 12))
        w, h, spp = int(rng.integers(40, 260)), int(rng.integers(20, 150)), int(rng.integers(1, 12))
        kw = dict(n_bounces=depth, camera=cam, seed=int(rng.integers(1 << 30)))
        fast = sc.render(pt.make_config(w, h, spp, math_mode=pt.PT_MATH_FAST, **kw))
        ref = sc.render(pt.make_config(w, h, spp, math_mode=pt.PT_MATH_FAST_SCALAR, **kw))
        assert torch.isfinite(ref).all(), trial
        assert torch.equal(fast, ref), (trial, n_shapes, cam, depth, (w, h, spp), int((fast != ref).any(dim=2).sum()))
    assert len(seen_kinds) >= 5          # short, mixed and general kind sets, with and without a hierarchy


def test_unusual_viewports(torch, pt):
    """The culling of camera rays works from the sample plane: a plane behind the camera (negative focal length: the
    camera looks along +z), a mirrored viewport (negative height), a very wide and a very narrow view, a focal length
    of zero (every ray direction degenerate) must all give the run-table loop's image."""
    rng = np.random.default_rng(7)
    sc = pt.Scene(*pt.parse_scene(random_scene(rng, 30, 0.3, True)))
    front = pt.Scene(*pt.parse_scene("\n".join(MATERIALS + [
        "plane light bounds -2 -2 2 2 rotate pi 0 1 0 translate 0 0 6", "sphere glass translate 0.3 0.2 3 scale 0.9",
        "cube red translate -0.8 -0.4 2.5 scale 0.5 0.7 0.4", "sphere base translate 0 0 -3 scale 0.8"]) + "\n"))
    for scene in (sc, front):
        for focal, vph in ((-6.0, 5.0), (6.0, -5.0), (0.3, 5.0), (60.0, 5.0), (0.0, 5.0), (6.0, 1e-3), (-0.5, -40.0)):
            imgs = []
            for mode in (pt.PT_MATH_FAST, pt.PT_MATH_FAST_SCALAR):
                cfg = pt.make_config(150, 90, 4, n_bounces=4, math_mode=mode, camera=(0.1, -0.2, 0.4))
                cfg.focal_length, cfg.view_port_height = focal, vph
                imgs.append(scene.render(cfg))
            # (NaN == NaN is false: compare bit patterns, a degenerate view may legitimately produce NaNs in both)
            assert torch.equal(imgs[0].view(torch.int32), imgs[1].view(torch.int32)), (focal, vph)
