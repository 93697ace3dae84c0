"""PT_MATH_FAST without a hierarchy runs the loop with camera batches (per-tile culling of the records camera rays
can hit) and the queue of set-aside glass hits.  Its arithmetic per path is the plain loop's and pixel sums are exact,
so its image must equal, bit for bit, the plain loop's: PT_MATH_FAST_SCALAR (runs in scene order) and PT_MATH_FAST
with the queue switched off (PT_TUNE_GLASS_QUEUE = -1, kind-sorted records).  Everything here is exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scene_text_box_with(extra):
    return "\n".join([
        "material base albedo 0.4 0.4 0.4", "material red albedo 1 0 0", "material green albedo 0 1 0",
        "material light emissive 20 18 14",
        "material glow albedo 0.7 0.6 0.5 emissive 0.3 0.2 0.1",          # lit AND reflective: the path goes on with light
        "material glass f0 0.02 specularColor 0.8 0.8 0.8 IOR 1.5 transparency 1 refractionColor 0.9 0 0",
        "material rough_glass f0 0.04 specularColor 0.9 0.9 0.9 IOR 1.3 transparency 0.7 refractionRoughness 0.3 specularRoughness 0.2 albedo 0.2 0.3 0.4",
        "material mirror f0 0.9 specularColor 0.9 0.9 0.9 specularRoughness 0.1",
        "material glow_glass f0 0.02 specularColor 0.8 0.8 0.8 IOR 1.5 transparency 1 emissive 0.05 0.1 0.02",
        "plane base bounds -15 -15 15 15 rotate pi/2 1 0 0 translate 0 -1 0",
        "plane red bounds -15 -15 15 15 rotate pi/2 0 1 0 translate -1 0 -5",
        "plane green bounds -15 -15 15 15 rotate -pi/2 0 1 0 translate 1 0 -5",
        "plane base bounds -15 -15 15 15 translate 0 0 -5",
        "plane base bounds -15 -15 15 15 rotate -pi/2 1 0 0 translate 0 1 0",
        "plane light bounds -0.5 -0.5 0.5 0.5 rotate -pi/2 1 0 0 translate 0 0.97 -3.25",
    ] + extra) + "\n"


def _scenes(pt):
    rng = np.random.default_rng(5)
    many = []
    for k in range(40):                     # rotated shapes of all kinds and materials, some overlapping
        kind = ("cylinder", "sphere", "cube", "sphere")[k % 4]
        mat = ("base", "glass", "rough_glass", "mirror", "glow", "glow_glass", "red")[k % 7]
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        pos = rng.uniform([-0.9, -0.9, -4.8], [0.9, 0.9, -1.6])
        s = rng.uniform(0.1, 0.45, size=3)
        scale = "scale %.4f" % s[0] if (kind == "sphere" and k % 8 < 4) else "scale %.4f %.4f %.4f" % tuple(s)
        many.append("%s %s rotate %.4f %.4f %.4f %.4f translate %.4f %.4f %.4f %s" % (
            kind, mat, rng.uniform(0, 3.1), ax[0], ax[1], ax[2], pos[0], pos[1], pos[2], scale))
    return {
        "default": (pt.default_scene(), 5),
        "glass grid, 32 segments": (pt.glass_grid_scene(5), 31),
        "all kinds and materials": (pt.parse_scene(_scene_text_box_with(many)), 9),
        # the camera inside a glass sphere and inside a glowing cube; an unbounded plane; a rotated wall
        "camera inside shapes": (pt.parse_scene(_scene_text_box_with([
            "sphere glass translate 0 0 0 scale 1.5", "cube glow translate 0 0 -0.2 scale 3.5 1.8 3",
            "plane mirror rotate 0.6 0 1 0 translate 0.2 0 -4", "plane base rotate 0.3 1 0 0 translate 0 -0.8 -3",
            "cylinder rough_glass rotate 1.1 1 0 0 translate -0.4 -0.3 -2.5 scale 0.5 0.5 0.9"])), 7),
        "65 primitives (no culling mask)": (pt.parse_scene(_scene_text_box_with(many + many[:19])), 5),
    }


CONFIGS = [
    # width, height, spp, extra config
    (97, 61, 5, {}),                                   # widths and sample counts that fill no tile or batch evenly
    (64, 3, 1, {}),
    (200, 120, 16, dict(fb_layout=1)),                 # SoA planes
    (160, 90, 7, dict(row_begin=3, row_step=4, fb_compact=1)),
    (160, 90, 7, dict(row_begin=10, row_end=37)),
    (33, 200, 3, dict(tile_log2=6)),                   # tiles that span several image rows
    (320, 8, 2, dict(tile_log2=2)),
    (120, 70, 4, dict(seed=9)),
]


def _render_all_ways(torch, pt, scene, depth, w, h, spp, extra):
    from pathtracer_amd import binding as b
    kw = dict(extra)
    kw.setdefault("n_bounces", depth)
    def cfg(mode):
        return pt.make_config(w, h, spp, math_mode=mode, **kw)
    rows = pt.rendered_rows(cfg(pt.PT_MATH_FAST)) if kw.get("fb_compact") else h
    shape = (rows, w, 3) if kw.get("fb_layout", 0) == 0 else (3, rows, w)
    def fresh():
        return torch.full(shape, -5.0, dtype=torch.float32, device="cuda")
    ref = scene.render(cfg(pt.PT_MATH_FAST_SCALAR), out=fresh())
    out = {}
    for name, cap, gate in (("automatic", 0, 0), ("8 entries", 8, 1), ("64 entries", 64, 64), ("16 entries, gate 16", 16, 16)):
        scene.tune(b.PT_TUNE_GLASS_QUEUE, cap).tune(b.PT_TUNE_GLASS_GATE, gate)
        out[name] = scene.render(cfg(pt.PT_MATH_FAST), out=fresh())
    scene.tune(b.PT_TUNE_GLASS_QUEUE, -1).tune(b.PT_TUNE_GLASS_GATE, 0)
    out["plain loop, sorted records"] = scene.render(cfg(pt.PT_MATH_FAST), out=fresh())
    scene.tune(b.PT_TUNE_GLASS_QUEUE, 0)
    torch.cuda.synchronize()
    return ref, out


def test_batched_loop_equals_plain_loop_bitwise(torch, pt):
    for name, (scene_data, depth) in _scenes(pt).items():
        scene = pt.Scene(*scene_data)
        assert scene.bvh_nodes == 0, name
        for (w, h, spp, extra) in CONFIGS:
            ref, out = _render_all_ways(torch, pt, scene, depth, w, h, spp, extra)
            assert torch.isfinite(ref[ref != -5.0]).all()
            for how, img in out.items():
                assert torch.equal(img, ref), (name, (w, h, spp, extra), how)
        # depths 0 and 1: paths that end at the camera ray's hit, or one segment later
        for d in (0, 1):
            ref, out = _render_all_ways(torch, pt, scene, d, 150, 80, 6, dict(n_bounces=d))
            for how, img in out.items():
                assert torch.equal(img, ref), (name, "depth", d, how)
        assert float(ref.sum()) > 0.0, name


def test_parity_order_batched_loop_equals_its_plain_loop_bitwise(torch, pt):
    """PT_MATH_PARITY on a scene without a hierarchy runs the same loop (typed loops over the table sorted by shape
    type, camera batches over the tile's candidates, diffuse hits shaded in the loop body, the others set aside) with
    the reference's operation order: every float operation of a path is the plain parity-order loop's, so the two
    images are equal bit for bit.  The plain loop is what PT_TUNE_GLASS_QUEUE = -1 runs."""
    from pathtracer_amd import binding as b
    for name, (scene_data, depth) in _scenes(pt).items():
        scene = pt.Scene(*scene_data)
        for (w, h, spp, extra) in CONFIGS + [(150, 80, 6, dict(n_bounces=0)), (150, 80, 6, dict(n_bounces=1))]:
            kw = dict(extra)
            kw.setdefault("n_bounces", depth)
            cfg = pt.make_config(w, h, spp, math_mode=pt.PT_MATH_PARITY, **kw)
            rows = pt.rendered_rows(cfg) if kw.get("fb_compact") else h
            shape = (rows, w, 3) if kw.get("fb_layout", 0) == 0 else (3, rows, w)
            fresh = lambda: torch.full(shape, -5.0, dtype=torch.float32, device="cuda")
            scene.tune(b.PT_TUNE_GLASS_QUEUE, -1)
            ref = scene.render(cfg, out=fresh())
            for cap, gate in ((0, 0), (8, 1), (64, 64)):
                scene.tune(b.PT_TUNE_GLASS_QUEUE, cap).tune(b.PT_TUNE_GLASS_GATE, gate)
                img = scene.render(cfg, out=fresh())
                assert torch.equal(img, ref), (name, (w, h, spp, extra), cap, gate)
            scene.tune(b.PT_TUNE_GLASS_QUEUE, 0).tune(b.PT_TUNE_GLASS_GATE, 0)
        assert float(ref[ref != -5.0].sum()) > 0.0, name


def test_parity_order_loops_agree_on_the_headline_frame(torch, pt):
    """The 1920x1080 @ 64 spp frame of the default scene and a 960x540 @ 16 frame of the glass grid at depth 31,
    PT_MATH_PARITY: the loop with camera batches and the plain loop, bit for bit."""
    from pathtracer_amd import binding as b
    for data, w, h, spp, depth in ((pt.default_scene(), 1920, 1080, 64, 5), (pt.glass_grid_scene(5), 960, 540, 16, 31)):
        scene = pt.Scene(*data)
        cfg = pt.make_config(w, h, spp, n_bounces=depth, math_mode=pt.PT_MATH_PARITY)
        scene.tune(b.PT_TUNE_GLASS_QUEUE, -1)
        plain = scene.render(cfg)
        scene.tune(b.PT_TUNE_GLASS_QUEUE, 0)
        batched = scene.render(cfg)
        assert torch.equal(plain, batched), (w, h, spp, depth)


def test_camera_elsewhere_and_looking_at_nothing(torch, pt):
    """Another camera position (the culling test works relative to it), and a view in which most tiles see no
    primitive at all: batches that park nothing must not stall the wave."""
    from pathtracer_amd import binding as b
    text = "material light emissive 3 2 1\nmaterial base albedo 0.5 0.5 0.5\n" \
           "sphere light translate 4 3 -9 scale 0.8\ncube base translate -3 -2 -7 scale 1 1 1\nsphere base translate 0.3 0.2 -6 scale 0.7\n"
    sc = pt.Scene(*pt.parse_scene(text))
    for cam in ((0.0, 0.0, 0.0), (1.5, -0.7, 2.0), (4.0, 3.0, -7.0)):
        cfg = lambda mode: pt.make_config(300, 170, 8, n_bounces=4, camera=cam, math_mode=mode)
        a = sc.render(cfg(pt.PT_MATH_FAST))
        r = sc.render(cfg(pt.PT_MATH_FAST_SCALAR))
        assert torch.equal(a, r), cam
        assert float((r.sum(dim=2) == 0).float().mean()) > 0.5          # mostly empty
    box = pt.Scene(*pt.default_scene())
    for cam in ((0.5, -0.3, 1.0), (-0.9, 0.9, -4.5), (0.0, 0.0, 6.5)):
        cfg = lambda mode: pt.make_config(240, 135, 8, camera=cam, math_mode=mode)
        assert torch.equal(box.render(cfg(pt.PT_MATH_FAST)), box.render(cfg(pt.PT_MATH_FAST_SCALAR))), cam


def test_tune_rejects_bad_values(pt):
    from pathtracer_amd import binding as b
    sc = pt.Scene(*pt.default_scene())
    for knob, bad in ((b.PT_TUNE_GLASS_QUEUE, 7), (b.PT_TUNE_GLASS_QUEUE, 65), (b.PT_TUNE_GLASS_GATE, 65), (b.PT_TUNE_TILE_LOG2, 7),
                      (b.PT_TUNE_SHADE_GATE, 65), (b.PT_TUNE_BVH_WALK, 4), (99, 1), (-1, 1)):
        assert b.lib.pt_scene_tune(sc.h, knob, bad) != 0
        assert b"pt_scene_tune" in b.lib.pt_last_error()


def test_every_compiled_kind_set_equals_the_run_table_loop(torch, pt):
    """The loops are compiled per set of record kinds (short kinds only, general kinds only, all kinds with offsets and
    counts read from the head of the record buffer); each must give the image of the loop over the run table in scene
    order, bit for bit, with the batched loop and with the plain one."""
    from pathtracer_amd import binding as b
    for general in (0, b.PT_GENERAL_ALL, b.PT_GENERAL_PLANES, b.PT_GENERAL_CUBES, b.PT_GENERAL_SPHERES, b.PT_GENERAL_PLANES | b.PT_GENERAL_SPHERES):
        for scene_data, depth in ((pt.default_scene(), 5), (pt.glass_grid_scene(3), 12)):
            sc = pt.Scene(*scene_data, general_kinds=general)
            for model in (pt.PT_BSDF_REFERENCE, pt.PT_BSDF_CLASSES):        # the class-based model runs the plain loop
                cfg = lambda mode: pt.make_config(210, 118, 6, n_bounces=depth, math_mode=mode, bsdf_model=model)
                ref = sc.render(cfg(pt.PT_MATH_FAST_SCALAR))
                assert torch.equal(sc.render(cfg(pt.PT_MATH_FAST)), ref), (general, depth, model)
            sc.tune(b.PT_TUNE_GLASS_QUEUE, -1)
            cfg = pt.make_config(210, 118, 6, n_bounces=depth, math_mode=pt.PT_MATH_FAST)
            assert torch.equal(sc.render(cfg), sc.render(pt.make_config(210, 118, 6, n_bounces=depth, math_mode=pt.PT_MATH_FAST_SCALAR))), (general, "plain")
