"""CPU tests of the C-ABI boundary: the library loads, exports every symbol the
header declares, and its host-only entry points (scene builders, focal length,
PPM writer) agree with the oracle.  No compute calls, no GPU."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "pt_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from pathtracer_amd import binding
    syms = _declared_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(binding.lib, s), "libpt_hip.so does not export %s" % s
    assert sorted(binding.EXPORTS) == syms
    assert b"gfx950" in binding.lib.pt_version()


def test_struct_layouts_match_the_reference_sizes():
    from pathtracer_amd import binding
    assert C.sizeof(binding.pt_material) == 72          # materials.h: 17 floats + int
    assert C.sizeof(binding.pt_prim) == 72
    assert C.sizeof(binding.pt_config) == 72
    assert C.sizeof(binding.pt_upload_options) == 16    # four int32 (pt_host.cpp asserts the same of the C struct)


def test_default_scene_builder_matches_oracle_bitwise(oracle):
    from pathtracer_amd import binding
    p1, n1, m1, nm1 = binding.default_scene()
    p2, n2, m2, nm2 = oracle.default_scene()
    assert (n1, nm1) == (n2, nm2) == (11, 8)
    assert bytes(p1) == bytes(p2)
    assert bytes(m1) == bytes(m2)


def test_focal_length_and_ppm_match_oracle(oracle):
    from pathtracer_amd import binding
    for nx, ny in ((800, 800), (256, 256), (1920, 1080), (3840, 2160), (1080, 1920)):
        assert binding.focal_length(nx, ny) == oracle.focal_length(nx, ny)
    rng = np.random.default_rng(1)
    fb = (rng.random((5, 7, 3)) * 3).astype(np.float32)
    assert binding.ppm_p3(fb) == oracle.ppm_text(fb)


def test_scene_builders_are_well_formed():
    from pathtracer_amd import binding
    prims, n, mats, nm = binding.glass_grid_scene(5)
    assert n == 8 + 25 and nm == 8
    assert sum(1 for k in range(n) if prims[k].type == binding.PT_SPHERE) == 25
    prims, n, mats, nm = binding.synthetic_scene(5000, 4989, 12345)
    assert n == 10000
    cs = np.array([[prims[k].transform[3], prims[k].transform[7], prims[k].transform[11]] for k in range(11, n)])
    assert cs[:, 0].min() >= -0.9 and cs[:, 0].max() <= 0.9 and cs[:, 2].min() >= -4.9 and cs[:, 2].max() <= -1.5
    s = np.array([prims[k].transform[0] for k in range(11, n)])
    assert s.min() >= 0.01 and s.max() <= 0.03
    # deterministic in the seed
    prims2, _, _, _ = binding.synthetic_scene(5000, 4989, 12345)
    assert bytes(prims) == bytes(prims2)


def test_bad_arguments_are_reported_not_fatal():
    from pathtracer_amd import binding
    h = C.c_void_p(0)
    prims, n, mats, nm = binding.default_scene()
    prims[0].material = 99
    rc = binding.lib.pt_scene_upload(prims, n, mats, nm, 0, C.byref(h))
    assert rc != 0
    msg = binding.lib.pt_last_error().decode()
    assert msg.startswith("HIP error = ") and " at " in msg and "material index" in msg   # check_cuda shape, main.cu:14-15


def test_scene_text_reproduces_the_default_scene_bitwise():
    from pathtracer_amd import binding
    text = open(os.path.join(ROOT, "scenes", "default.scene")).read()
    p1, n1, m1, nm1 = binding.parse_scene(text)
    p2, n2, m2, nm2 = binding.default_scene()
    assert (n1, nm1) == (n2, nm2) == (11, 8)
    assert bytes(p1)[: n1 * C.sizeof(binding.pt_prim)] == bytes(p2)
    assert bytes(m1)[: nm1 * C.sizeof(binding.pt_material)] == bytes(m2)


def test_scene_text_errors_and_options():
    import pytest
    from pathtracer_amd import binding
    with pytest.raises(binding.PtError, match="scene line 2: shape needs a material"):
        binding.parse_scene("material a albedo 1 1 1\nsphere b translate 0 0 -3\n")
    with pytest.raises(binding.PtError, match="scene line 1: unknown statement"):
        binding.parse_scene("torus a\n")
    with pytest.raises(binding.PtError, match="expected a number"):
        binding.parse_scene("material a albedo 1 x 1\n")
    prims, n, mats, nm = binding.parse_scene("# c\nmaterial m IOR 1.3\ncylinder m rotate pi/4 0 0 1 scale 2 translate 1 2 3\nplane m\n")
    assert (n, nm) == (2, 1) and abs(mats[0].IOR - 1.3) < 1e-7
    t = np.array(prims[0].transform[:]).reshape(3, 4)
    np.testing.assert_allclose(t[:, 3], [1, 2, 3])
    np.testing.assert_allclose(t[:2, :2], 2 * np.array([[np.cos(np.pi / 4), -np.sin(np.pi / 4)], [np.sin(np.pi / 4), np.cos(np.pi / 4)]]), rtol=1e-6)
    assert prims[1].bounds[0] == -np.finfo(np.float32).max and prims[1].bounds[2] == np.finfo(np.float32).max


def test_binary_image_writers():
    from pathtracer_amd import binding
    rng = np.random.default_rng(3)
    fb = (rng.random((4, 6, 3)) * 1.5 - 0.1).astype(np.float32)
    p6 = binding.ppm_p6(fb)
    head = b"P6\n6 4\n255\n"
    assert p6.startswith(head)
    want = np.clip((255.99 * fb[::-1].astype(np.float64)).astype(np.int64), 0, 255).astype(np.uint8)   # top row first
    assert p6[len(head):] == want.tobytes()
    pf = binding.pfm(fb)
    head = b"PF\n6 4\n-1.0\n"
    assert pf.startswith(head) and pf[len(head):] == fb.tobytes()
