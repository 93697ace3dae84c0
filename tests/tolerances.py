"""The parity bounds of this repository in one place: every GPU test and __graft_entry__.smoke() read them
from here.  All image quantities are relative to the mean radiance of the reference image;
"bad pixel" = a pixel whose largest channel difference exceeds BAD_PIXEL_ABS.

The reference pins hit counts only (tests/shape_test.cc), so everything below is kernel-versus-oracle (the CPU
restatement in oracle/): "parity unpinned" beyond those counts."""

BAD_PIXEL_ABS = 0.05

# PT_MATH_PARITY: the reference's operation order, IEEE divide / sqrt, no FMA contraction.
PARITY = {
    "t_ulp": 2,                 # hit distances against the oracle (measured 0)
    "path_abs": 1e-4,           # per-path radiance of explicit rays, share of paths within it >= path_share
    "path_share": 0.98,
    "image_rmse": 0.01,         # RMSE of the image (measured 5e-9 at 256x256@16, 1.0e-3 on 1080p@64 rows: a path whose
                                # device sinf/cosf/expf differs from the host's in the last bit can take another branch)
    "bad_pixel_frac": 1e-3,
    "mean_rel": 0.001,          # per-channel mean radiance
}

# PT_MATH_FAST: same algorithm and random streams, v_rcp/v_rsq/v_sin/v_cos, FMA contraction.  A rounding difference
# flips a branch for ~2 % of the paths, which then return an unrelated sample: per-pixel differences are Monte-Carlo
# noise of the flipped paths and shrink like 1/sqrt(spp); what a fast build must not have is BIAS, which
# tests/test_gpu_bias.py measures by averaging seeds.
FAST = {
    "t_rel_max": 1e-4,            # pt_nearest_batch_mode(PT_MATH_FAST): two hits closer than this (relative) count as a tie
    "nearest_prim_mismatch": 5e-3,  # rays naming another primitive (measured 2e-3): re-hits of the surface a grazing ray
    "rehit_t": 1e-2,                # leaves, at a distance below rehit_t, ties within t_rel_max, and
    "nearest_grazing": 5e-4,        # grazing hits of small shapes (measured: 1 ray of 4096 on the 10 000-primitive scene)
    "image_rmse_at_1spp": 0.13,   # fast_image_rmse(spp) = this / sqrt(spp); measured x sqrt(spp): 0.086 (256x256@16, 2.1 %),
                                  # 0.097 (1080p@64, 1.21 %), 0.086 (every 400th row of 4K@1024, 0.27 %)
    "bad_pixel_frac": 3e-3,       # measured 1.0e-3 at 1080p@64
    "mean_rel": 0.005,            # per-channel mean radiance of one frame (converged-statistics bound)
    "bias_mean_rel": 0.001,       # per-channel mean of the seed-averaged difference to the parity-order kernel
                                  # (measured 1.5e-4 default scene, 3e-4 glass grid at depth 31)
    "bias_mean_rel_small_shapes": 0.0006,  # the same on the 10 000-shape scene over 512 seeds (measured 0.00002 / 0.00005 /
                                  # 0.00005 with a standard error of 0.00009 over 768 seeds; it was 0.0012 before the general
                                  # sphere record rounded as the reference does)
    "bias_mean_rel_glass": 0.002, # one 1080p@64 frame of the glass grid at depth 31 against the parity-order kernel
}


# getNearestIntersection of the fast kernels (pt_nearest_batch_mode) against the oracle, rays that name the same
# primitive and root.  Distances of camera rays in ULPs of t, of rays that start on a surface in ULPs of max(t, |o|).
# The median is the arithmetic's own error; the tail is conditioning: a sphere of radius r at distance D has its
# discriminant b^2 - c cancel to r^2/D^2 of its terms, in the reference's local frame as much as here, so two correct
# float evaluations differ by hundreds of ULPs on small far shapes and by more at grazing incidence.  Measured
# (median / p99 / max): default scene 1 / 24 / 421, general kinds 0 / 5 / 224, glass grid 1 / 73 / 735, rotated
# cylinders-ellipsoids-cubes 1 / 27 / 1377, 10 000 small shapes 1 / 238 / 2825.  Positions relative to the largest
# coordinate: 3.3e-5 default, 2.2e-4 small shapes; normals (99th percentile, per component): 7.7e-6 and 1.8e-3 (a
# position error divided by the radius).
FAST_NEAREST = {
    "default": {"t_ulp_median": 1, "t_ulp_p99": 64, "t_ulp_max": 512, "pos_rel_max": 1e-4, "nrm_abs_p99": 1e-4},
    "any": {"t_ulp_median": 2, "t_ulp_p99": 1024, "t_ulp_max": 8192, "pos_rel_max": 1e-3, "nrm_abs_p99": 1e-2},
}

# Scenes whose paths are long or whose hits are decided by near-ties get their own, looser, image bounds (same
# quantities).
GLASS_DEPTH_31 = {       # 5x5 glass spheres, 32 segments, 160x90@16: almost every path crosses several refractions
    "parity_image_rmse": 0.02, "parity_bad_pixel_frac": 0.002,
    "fast_image_rmse": 0.17, "fast_bad_pixel_frac": 0.02, "fast_mean_rel": 0.02,     # measured 0.111 / 0.0076
}
MANY_SMALL_SHAPES = {"fast_bad_pixel_frac": 0.02}      # 10 000 primitives / 73 primitives with 41 materials, small frames
CLASSES_BSDF = {"fast_bad_pixel_frac": 0.02, "fast_mean_rel": 0.03}   # bsdf_alternative.h model; parity unpinned (no reference output)
PARITY_SMALL_FRAME_RMSE = 1e-5                         # parity-order kernel on frames of a few thousand pixels (measured 5e-9)


# pt_bsdf_batch: largest error of each helper function against the oracle over 65 536 random arguments, in units of
# the float spacing at the result's largest component.  Parity-order code is bit-exact except where it calls the
# device's sinf/cosf.  Fast code, measured: sampleHemisphere 10.5 (v_sin/v_cos in revolutions), FresnelSchlick 4,
# reflect 5, lerp 4, normalized 3 (v_rsq); getFresnelRatio 228 and refract 202 are reached only within rounding of the
# critical angle, where sqrt(1 - s2) and sqrt(k) amplify any error (mean error 0.6 and 0.2).
BSDF_ULP = {
    "parity": {"sampleHemisphere": 2, "FresnelSchlick": 0, "getFresnelRatio": 0, "reflect": 0, "refract": 0, "lerp": 0, "normalized": 0},
    "fast": {"sampleHemisphere": 16, "FresnelSchlick": 8, "getFresnelRatio": 512, "reflect": 8, "refract": 512, "lerp": 8, "normalized": 4},
}
BSDF_FAST_MEAN_ULP = 1.0     # mean error of every fast helper function


def fast_image_rmse(spp):
    """Bound on the fast kernel's image RMSE against the oracle at `spp` samples per pixel, same seed."""
    return FAST["image_rmse_at_1spp"] / float(spp) ** 0.5
