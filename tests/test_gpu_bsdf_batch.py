"""sampleHemisphere, FresnelSchlick, getFresnelRatio, reflect, refract, lerp and .normalized() on their own
(pt_bsdf_batch), in both kernel families, against the oracle's restatement of bsdf.h:29-69 and vec_math.h:10-30.

Error unit for a vector result: the spacing of floats at its largest reference component (a component that
cancels to near zero has no meaningful ULP of its own)."""
import numpy as np
import pytest

import tolerances as tol

pytestmark = pytest.mark.gpu

N = 1 << 16


def _unit(rng, n):
    v = rng.normal(size=(n, 3))
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def _inputs(fn, rng):
    u = lambda *shape: rng.random(shape, dtype=np.float32)
    if fn == 0:                                                   # u1, u2 in (0,1], both ends of the range included
        x = np.maximum(u(N, 2), np.float32(2.0 ** -33))
        x[:8] = [[1, 1], [1, 0.5], [0.5, 1], [2.0 ** -33, 2.0 ** -33], [0.5, 0.25], [0.5, 0.75], [1, 0.25], [0.25, 2.0 ** -33]]
        return x
    if fn == 1:                                                   # n1, n2, cos
        x = np.stack([1 + u(N), 1 + u(N), u(N)], axis=1)
        x[:4] = [[1, 1.5, 1], [1.5, 1, 0], [1, 1, 0.3], [1.5, 1, 1]]
        return x
    if fn == 2:                                                   # N, I, f0, f90, n1, n2: both media orders, total reflection included
        nrm, d = _unit(rng, N), _unit(rng, N)
        dense = rng.random(N) < 0.5
        n1 = np.where(dense, 1.5, 1.0).astype(np.float32); n2 = np.where(dense, 1.0, 1.5).astype(np.float32)
        return np.concatenate([nrm, d, 0.02 + 0.2 * u(N, 1), np.ones((N, 1), np.float32), n1[:, None], n2[:, None]], axis=1)
    if fn == 3:
        return np.concatenate([_unit(rng, N), _unit(rng, N)], axis=1)
    if fn == 4:                                                   # I, N (facing I), eta = 1/1.5 (entering) or 1.5 (leaving: total reflection beyond 41.8 degrees)
        d, nrm = _unit(rng, N), _unit(rng, N)
        nrm *= -np.sign((d * nrm).sum(axis=1, keepdims=True)).astype(np.float32)
        eta = np.where(rng.random(N) < 0.5, 1 / 1.5, 1.5).astype(np.float32)
        return np.concatenate([d, nrm, eta[:, None]], axis=1)
    if fn == 5:
        return np.concatenate([_unit(rng, N), _unit(rng, N), u(N, 1) ** 2], axis=1)
    x = (rng.normal(size=(N, 3)) * 10.0 ** rng.uniform(-3, 3, size=(N, 1))).astype(np.float32)
    x[:3] = [[0, 0, 0], [0, 0, 2], [-3, 4, 0]]
    return x


def _errors_in_ulp(got, ref):
    scale = np.maximum(np.abs(ref).max(axis=1, keepdims=True), np.float32(1e-30)).astype(np.float64)
    spacing = 2.0 ** (np.floor(np.log2(scale)) - 23)
    return np.abs(got.astype(np.float64) - ref.astype(np.float64)) / spacing


NAMES = ["sampleHemisphere", "FresnelSchlick", "getFresnelRatio", "reflect", "refract", "lerp", "normalized"]


@pytest.mark.parametrize("fn", range(7))
def test_helper_functions_against_the_oracle(torch, pt, oracle, fn):
    from pathtracer_amd import binding as b
    rng = np.random.default_rng(100 + fn)
    x = _inputs(fn, rng)
    ref = oracle.bsdf_batch(fn, x)
    dx = torch.from_numpy(x).cuda()
    for mode, name in ((b.PT_MATH_PARITY, "parity"), (b.PT_MATH_FAST, "fast")):
        got = b.bsdf_batch(fn, dx, math_mode=mode).cpu().numpy()
        assert got.shape == ref.shape and np.isfinite(got).all()
        err = _errors_in_ulp(got, ref)
        bound = tol.BSDF_ULP[name][NAMES[fn]]
        print("%-16s %-6s max %.2f ulp, mean %.3f (bound %g)" % (NAMES[fn], name, err.max(), err.mean(), bound))
        assert err.max() <= bound, (NAMES[fn], name, float(err.max()))
        assert name == "parity" or err.mean() <= tol.BSDF_FAST_MEAN_ULP
    # structural cases, exact in both modes
    if fn == 4:
        tir = (ref == 0).all(axis=1)
        assert 0.05 < tir.mean() < 0.5                                  # total reflection: the zero vector (vec_math.h:26-27)
        for mode in (b.PT_MATH_PARITY, b.PT_MATH_FAST):
            got = b.bsdf_batch(fn, dx, math_mode=mode).cpu().numpy()
            # the k < 0 decision may differ for rays within rounding of the critical angle
            assert ((got == 0).all(axis=1) != tir).mean() < 1e-4
    if fn == 2:
        total = ref[:, 0] == 1.0
        assert 0.05 < total.mean() < 0.5                                # n1 > n2 and s2 > 1: f90 (bsdf.h:59-61)
    if fn == 6:
        for mode in (b.PT_MATH_PARITY, b.PT_MATH_FAST):
            got = b.bsdf_batch(fn, dx[:1], math_mode=mode).cpu().numpy()
            assert (got == 0).all()                                     # normalized() of the zero vector is the zero vector


def test_bad_arguments(pt):
    import ctypes as C
    lib = pt.binding.lib
    assert lib.pt_bsdf_batch(0, 7, 0, None, 0, None, None) != 0 and b"unknown function" in lib.pt_last_error()
    assert lib.pt_bsdf_batch(0, 0, 2, None, 0, None, None) != 0
    assert lib.pt_bsdf_batch(0, 0, 0, None, 4, None, None) != 0
    assert lib.pt_bsdf_batch(0, 0, 0, None, 0, None, None) == 0
    a, b_ = C.c_int(0), C.c_int(0)
    assert lib.pt_bsdf_arity(2, C.byref(a), C.byref(b_)) == 0 and (a.value, b_.value) == (10, 1)
    assert lib.pt_bsdf_arity(-1, C.byref(a), C.byref(b_)) != 0
