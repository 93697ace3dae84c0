"""The tracePath estimator on the GPU against a closed form that owes nothing to the oracle: inside a closed room whose
every wall emits e and reflects diffusely with albedo a, the expected radiance is e (1 + a + ... + a^n_bounces) per
channel (emission added at every hit, bsdf.h:90-92; throughput times albedo; roulette survival probability and
compensation cancel, bsdf.h:155-164).  Under the class-based model of bsdf_alternative.h half of the Lambertian's
directions lie below the surface and carry no weight (:94-112), so the throughput after a bounce is a / 2 in
expectation and the closed form is e (1 + a/2 + ... + (a/2)^n_bounces) — the one pin that model can have, since the
reference holds no output for that file.  Both kernel families, both models."""
import numpy as np
import pytest

from test_oracle_analytic import FURNACE, furnace_expectation, furnace_expectation_classes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model", ["reference", "classes"])
@pytest.mark.parametrize("mode", ["parity", "fast"])
def test_furnace(torch, pt, mode, model):
    sc = pt.Scene(*pt.parse_scene(FURNACE))
    mm = pt.PT_MATH_PARITY if mode == "parity" else pt.PT_MATH_FAST
    bm = pt.PT_BSDF_REFERENCE if model == "reference" else pt.PT_BSDF_CLASSES
    for depth in (0, 1, 5, 12):
        img = sc.render(pt.make_config(256, 256, 64, n_bounces=depth, math_mode=mm, bsdf_model=bm)).double()
        mean = img.mean(dim=(0, 1)).cpu().numpy()
        want = furnace_expectation(depth) if model == "reference" else furnace_expectation_classes(depth)
        print("furnace %s %s depth %d: mean %s, closed form %s" % (mode, model, depth, mean, want))
        assert np.abs(mean / want - 1.0).max() < 0.005, (depth, mean, want)        # 4.2 M samples
