"""The sample loop of main.cu:61-67 cut into several launches (pt_render_samples + pt_resolve): a pixel's sum is an
exact integer sum and sample s draws from the same stream whichever launch traces it, so the frame is pt_render's bit
for bit however the samples are cut up, in every kernel family and layout."""
import pytest

pytestmark = pytest.mark.gpu


def _chunked(torch, scene, cfg, cuts, streams=False):
    sums = None
    side = [torch.cuda.Stream() for _ in range(2)] if streams else None
    for k, (b, n) in enumerate(cuts):
        if side:
            if sums is None:
                sums = scene.render_samples(cfg, b, n)
                torch.cuda.synchronize()
                continue
            with torch.cuda.stream(side[k % 2]):
                sums = scene.render_samples(cfg, b, n, sums)
        else:
            sums = scene.render_samples(cfg, b, n, sums)
    torch.cuda.synchronize()
    return scene.resolve(cfg, sums)


@pytest.mark.parametrize("mode", ["fast", "parity", "fast_scalar", "fast_lds"])
def test_sample_ranges_equal_one_launch(torch, pt, mode):
    mm = {"fast": pt.PT_MATH_FAST, "parity": pt.PT_MATH_PARITY, "fast_scalar": pt.PT_MATH_FAST_SCALAR, "fast_lds": pt.PT_MATH_FAST_LDS}[mode]
    for data, depth in ((pt.default_scene(), 5), (pt.glass_grid_scene(3), 12), (pt.synthetic_scene(300, 300, 7), 4)):
        scene = pt.Scene(*data)
        for extra in (dict(), dict(fb_layout=1), dict(row_begin=3, row_step=4, fb_compact=1), dict(row_begin=10, row_end=37), dict(bsdf_model=pt.PT_BSDF_CLASSES)):
            cfg = pt.make_config(160, 90, 24, n_bounces=depth, math_mode=mm, **extra)
            whole = scene.render(cfg)
            for cuts in ([(0, 24)], [(0, 8), (8, 8), (16, 8)], [(5, 19), (0, 5)], [(0, 1), (1, 22), (23, 1)]):
                img = _chunked(torch, scene, cfg, cuts)
                assert torch.equal(img, whole), (mode, extra, cuts)
        # pieces in flight on two streams at once add into the same buffer
        cfg = pt.make_config(200, 120, 32, n_bounces=depth, math_mode=mm)
        assert torch.equal(_chunked(torch, scene, cfg, [(0, 8), (8, 8), (16, 8), (24, 8)], streams=True), scene.render(cfg))


def test_config_3_in_eight_pieces(torch, pt):
    """BASELINE config 3 (3840x2160 @ 1024 spp, default scene) as 8 launches of 128 samples against the single launch."""
    scene = pt.Scene(*pt.default_scene())
    cfg = pt.make_config(3840, 2160, 1024)
    whole = scene.render(cfg)
    img = _chunked(torch, scene, cfg, [(k * 128, 128) for k in range(8)])
    assert torch.equal(img, whole)


def test_bad_ranges_are_refused(torch, pt):
    scene = pt.Scene(*pt.default_scene())
    cfg = pt.make_config(64, 36, 16)
    for b, n in ((-1, 4), (0, 0), (10, 7), (16, 1)):
        with pytest.raises(pt.PtError):
            scene.render_samples(cfg, b, n)
    with pytest.raises(pt.PtError):
        scene.render(pt.make_config(64, 36, 4, n_bounces=1 << 24))


def test_segment_counter_is_the_same_in_every_kernel_family(torch, pt):
    """pt_render_counted: the number of path segments (nearest-hit searches) is a property of the paths, which the
    loops share at a seed: plain loop, loop with camera batches and LDS-staged loop count the same, the image is
    pt_render's, and the count lies between one and n_bounces + 1 per sample.  With a hierarchy the search is spread
    over loop iterations and still counted once per segment."""
    from pathtracer_amd import binding as b
    for data, depth in ((pt.default_scene(), 5), (pt.glass_grid_scene(3), 9), (pt.synthetic_scene(300, 300, 7), 4)):
        scene = pt.Scene(*data)
        counts = {}
        for name, mm in (("fast", pt.PT_MATH_FAST), ("scalar", pt.PT_MATH_FAST_SCALAR), ("lds", pt.PT_MATH_FAST_LDS)):
            cfg = pt.make_config(160, 90, 12, n_bounces=depth, math_mode=mm)
            img, c = scene.render_counted(cfg)
            assert torch.equal(img, scene.render(cfg)), name
            counts[name] = int(c[0])
        samples = 160 * 90 * 12
        assert counts["fast"] == counts["scalar"] == counts["lds"], counts
        assert samples <= counts["fast"] <= samples * (depth + 1)
        cfg = pt.make_config(160, 90, 12, n_bounces=depth, math_mode=pt.PT_MATH_PARITY)
        scene.tune(b.PT_TUNE_GLASS_QUEUE, -1)
        plain = int(scene.render_counted(cfg)[1][0])
        scene.tune(b.PT_TUNE_GLASS_QUEUE, 0)
        assert int(scene.render_counted(cfg)[1][0]) == plain and samples <= plain <= samples * (depth + 1)
