"""Bias of the fast kernels, separated from branch-flip noise.

At one seed the fast and the parity-order kernel trace the same random streams; where a rounding difference flips a
branch, the two paths part and the pixel differs by an unrelated Monte-Carlo sample.  That noise has zero mean and is
independent from seed to seed, so the difference of the SEED-AVERAGED images must shrink like 1/sqrt(S); a systematic
error of the fast arithmetic would not shrink.  The per-pixel RMSE bounds elsewhere cannot tell the two apart."""
import numpy as np
import pytest

import tolerances as tol

pytestmark = pytest.mark.gpu

W, H, SPP, SEEDS = 128, 72, 16, 32
SEEDS_SMALL_SHAPES = 512        # the 10 000-shape scene: until the noise of the mean is 0.015 % of the mean


def _seed_averaged_difference(pt, scene, depth, seeds=SEEDS):
    diffs = []
    par_mean = 0.0
    for seed in range(seeds):
        f = scene.render(pt.make_config(W, H, SPP, n_bounces=depth, seed=seed, math_mode=pt.PT_MATH_FAST)).double()
        p = scene.render(pt.make_config(W, H, SPP, n_bounces=depth, seed=seed, math_mode=pt.PT_MATH_PARITY)).double()
        diffs.append((f - p).cpu().numpy())
        par_mean = par_mean + p.mean(dim=(0, 1)).cpu().numpy() / seeds
    return np.stack(diffs), par_mean


def _rms_by_group_size(d, par_mean):
    rms = {}
    for s in (1, 2, 4, 8, 16, 32):
        # mean over disjoint groups of s seeds, then the RMS over pixels, channels and groups
        groups = d[: (SEEDS // s) * s].reshape(SEEDS // s, s, H, W, 3).mean(axis=1)
        rms[s] = float(np.sqrt((groups ** 2).mean()))
    return rms, {s: v / par_mean.mean() for s, v in rms.items()}


@pytest.mark.parametrize("name,depth,general", [("default", 5, 0), ("glass grid", 31, 0), ("glass grid, general sphere records", 31, 4),
                                                ("10 000 small shapes", 5, 0)])
def test_fast_minus_parity_shrinks_with_the_number_of_seeds(pt, name, depth, general):
    from pathtracer_amd import binding as b
    assert general in (0, b.PT_GENERAL_SPHERES)
    small = name == "10 000 small shapes"
    scene = pt.Scene(*(pt.default_scene() if name == "default" else (pt.synthetic_scene(5000, 4989, 12345) if small else pt.glass_grid_scene(5))),
                     general_kinds=general)
    d, par_mean = _seed_averaged_difference(pt, scene, depth, SEEDS_SMALL_SHAPES if small else SEEDS)
    rms, rel = _rms_by_group_size(d[:SEEDS], par_mean)
    print("%s: RMS of the seed-averaged difference / mean radiance: %s" % (name, {s: "%.4f" % v for s, v in rel.items()}))
    assert rms[1] > 0.0                                                  # the two kernels do differ somewhere
    # one seed: branch-flip noise (small frame, depth 31: x4, measured 0.109 = x3.4; shapes smaller than a pixel, whose
    # every sample is a near miss: x12, measured 0.251 = x7.7)
    assert rel[1] <= (12.0 if small else 4.0) * tol.fast_image_rmse(SPP)
    # noise law: each doubling of the seeds takes sqrt(2) off; a bias floor would stop the fall
    assert rms[32] <= 1.5 * rms[1] / np.sqrt(32.0), rel
    per_seed = d.mean(axis=(1, 2))                                       # the paired difference of the frame means, per seed and channel
    bias = np.abs(per_seed.mean(axis=0)) / par_mean
    err = per_seed.std(axis=0, ddof=1) / np.sqrt(len(per_seed)) / par_mean
    print("%s: per-channel mean of fast - parity over %d seeds, relative: %s +- %s" % (name, len(per_seed), bias, err))
    assert (bias <= tol.FAST["bias_mean_rel_small_shapes" if small else "bias_mean_rel"]).all(), (bias, err)
    if small: assert (err <= 0.0003).all(), err                          # the measurement resolves the bound


# History of the glass-grid case: uniformly scaled spheres used to be tested in world space (|o - centre|^2 - r^2).
# The reference tests them in a local frame in which a small sphere far from the origin has coordinates of magnitude
# |centre| / scale, so the next test of the sphere a ray has just left carries a larger rounding error, and a ray
# reflected at a grazing angle re-hits its own sphere beyond the 1e-5 filter of shapes.h:62-73 (and is lost) more often.
# The world-space test lost fewer: +0.3 % in the mean, up to +40 % in pixels on the rims of the spheres nearest the light,
# an RMS floor of 3.4 % that no number of seeds removed.  The test above found it; the record kind now evaluates
# o_l = k*o + q as the reference does, and the floor is gone (mean 0.03 %, within the noise of 32 seeds).
#
# The 10 000-shape case: spheres of diameter 0.01-0.03 up to 5 units away have their discriminant b*b - 4*a*c cancel to
# a few bits in float.  The reference rounds the two products before it subtracts them, so for a band of rays that just
# miss they come out equal, the discriminant is exactly 0 and shapes.h:337 counts a tangent hit: its small far spheres are
# in effect a little larger than their radius.  A fused multiply-add has no such band, and with it the fast kernels
# rendered this scene 0.12 % (red, green) / 0.07 % (blue) brighter than the parity-order kernel, in every class of pixels
# (tools/bias_study.py, 1024 seeds, standard error 0.009 %).  Spheres that far out (|centre| / scale > 32) get the general
# sphere record, which now rounds every product and sum as the reference does (test_sphere, take_quadratic<true>): the
# difference is 0.002 / -0.005 / -0.005 % +- 0.009 over 768 seeds.  Which part carried it was found by giving the
# parity-order kernel the fast kernels' habits one at a time: hit position as o + d*t no change, fused transforms no
# change, fused discriminant half of it.
