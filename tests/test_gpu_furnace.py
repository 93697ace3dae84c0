"""The tracePath estimator on the GPU against a closed form that owes nothing to the oracle: inside a closed room whose
every wall emits e and reflects diffusely with albedo a, the expected radiance is e (1 + a + ... + a^n_bounces) per
channel (emission added at every hit, bsdf.h:90-92; throughput times albedo; roulette survival probability and
compensation cancel, bsdf.h:155-164).  Both kernel families.  (The class-based model of bsdf_alternative.h is not held
to this: its Lambertian draws directions of which half lie below the surface and carry no weight, as that file has it,
so its furnace converges to another value.)"""
import numpy as np
import pytest

from test_oracle_analytic import FURNACE, furnace_expectation

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["parity", "fast"])
def test_furnace(torch, pt, mode, model="reference"):
    sc = pt.Scene(*pt.parse_scene(FURNACE))
    mm = pt.PT_MATH_PARITY if mode == "parity" else pt.PT_MATH_FAST
    bm = pt.PT_BSDF_REFERENCE if model == "reference" else pt.PT_BSDF_CLASSES
    for depth in (0, 1, 5, 12):
        img = sc.render(pt.make_config(256, 256, 64, n_bounces=depth, math_mode=mm, bsdf_model=bm)).double()
        mean = img.mean(dim=(0, 1)).cpu().numpy()
        want = furnace_expectation(depth)
        print("furnace %s %s depth %d: mean %s, closed form %s" % (mode, model, depth, mean, want))
        assert np.abs(mean / want - 1.0).max() < 0.005, (depth, mean, want)        # 4.2 M samples
