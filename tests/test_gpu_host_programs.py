"""GPU tests of the C++ host side: the reference's known answers and the rest of the host API, compiled
against include/pt/api.h, and the pathtracer program that reproduces the
reference's main() flow (scene construction -> render -> PPM on stdout)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "pathtracer_amd", "host")


def _need(name):
    path = os.path.join(HOST, name)
    assert os.path.exists(path), "%s is not built; run __graft_entry__.build()" % path
    return path


def test_reference_shape_tests_through_the_cpp_api():
    r = subprocess.run([_need("shape_api_test")], capture_output=True, text=True, timeout=120)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "6 test(s), 0 failure(s)" in r.stdout


def test_host_program_matches_library_render_and_reference_ppm_format():
    import pathtracer_amd as pt
    r = subprocess.run([_need("pathtracer"), "--width", "96", "--height", "64", "--spp", "8", "--math", "parity", "--stats"],
                       capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr.decode()
    assert b"Scanlines remaining: 64" in r.stderr           # progress line of main.cu:242
    assert b"msamples_per_s" in r.stderr
    import json
    stats = json.loads([l for l in r.stderr.decode().splitlines() if l.startswith("{")][-1])
    # a segment per camera ray, at most n_bounces + 1 per sample; the default scene is closed, so every camera ray hits
    assert stats["samples"] == 96 * 64 * 8 and 1.5 < stats["segments_per_sample"] <= 6.0
    assert 1000.0 < stats["flops_per_sample_event_table"] < 3500.0 and 0.0 < stats["pct_of_fp32_vector_peak"] < 100.0
    # the same render through the Python binding, written by the library's P3 writer
    prims, n, mats, nm = pt.default_scene()
    img = pt.Scene(prims, n, mats, nm).render(pt.make_config(96, 64, 8, math_mode=pt.PT_MATH_PARITY)).cpu().numpy()
    assert r.stdout == pt.ppm_p3(img)
    head = r.stdout.split(b"\n", 3)[:3]
    assert head == [b"P3", b"96 64", b"255"]


def test_reference_main_example_prints_the_same_file():
    """examples/reference_main.cpp keeps the reference's constructScene body and main() flow (main.cu:71-257) on the
    source-level API; the product's program takes its default scene from pt_default_scene.  Same scene, same file."""
    exe = os.path.join(ROOT, "examples", "reference_main")
    assert os.path.exists(exe), "%s is not built; run __graft_entry__.build()" % exe
    a = subprocess.run([exe, "96", "64", "8"], capture_output=True, timeout=120)
    b = subprocess.run([_need("pathtracer"), "--width", "96", "--height", "64", "--spp", "8"], capture_output=True, timeout=120)
    assert a.returncode == 0 and b.returncode == 0, a.stderr[-300:] + b.stderr[-300:]
    assert a.stdout == b.stdout and a.stdout.startswith(b"P3\n96 64\n255\n")


def test_host_program_rejects_bad_arguments():
    r = subprocess.run([_need("pathtracer"), "--gpus", "99"], capture_output=True, timeout=60)
    assert r.returncode == 2


def test_host_program_scene_file_and_binary_formats():
    import pathtracer_amd as pt
    scene_file = os.path.join(ROOT, "scenes", "default.scene")
    args = ["--width", "64", "--height", "48", "--spp", "4", "--math", "parity"]
    a = subprocess.run([_need("pathtracer")] + args, capture_output=True, timeout=120)
    b = subprocess.run([_need("pathtracer")] + args + ["--scene-file", scene_file], capture_output=True, timeout=120)
    assert a.returncode == 0 and b.returncode == 0 and a.stdout == b.stdout      # the text scene IS the default scene
    img = pt.Scene(*pt.default_scene()).render(pt.make_config(64, 48, 4, math_mode=pt.PT_MATH_PARITY)).cpu().numpy()
    p6 = subprocess.run([_need("pathtracer")] + args + ["--format", "p6"], capture_output=True, timeout=120)
    assert p6.returncode == 0 and p6.stdout == pt.ppm_p6(img)
    pf = subprocess.run([_need("pathtracer")] + args + ["--format", "pfm"], capture_output=True, timeout=120)
    assert pf.returncode == 0 and pf.stdout == pt.pfm(img)


def test_host_program_interleaved_tiles_and_alternative_bsdf():
    """The multi-GPU split of the C++ program (interleaved row sets, compact tiles, strided copy into the host image)
    on one device: any number of row sets gives the same file; --bsdf classes is pt_config.bsdf_model = 1."""
    import pathtracer_amd as pt
    args = ["--width", "80", "--height", "50", "--spp", "4", "--format", "pfm"]
    one = subprocess.run([_need("pathtracer")] + args, capture_output=True, timeout=120)
    assert one.returncode == 0
    for tiles in ("3", "7", "64"):
        r = subprocess.run([_need("pathtracer")] + args + ["--tiles", tiles], capture_output=True, timeout=120)
        assert r.returncode == 0 and r.stdout == one.stdout, tiles
    alt = subprocess.run([_need("pathtracer")] + args + ["--bsdf", "classes", "--tiles", "2"], capture_output=True, timeout=120)
    img = pt.Scene(*pt.default_scene()).render(pt.make_config(80, 50, 4, bsdf_model=pt.PT_BSDF_CLASSES)).cpu().numpy()
    assert alt.returncode == 0 and alt.stdout == pt.pfm(img) and alt.stdout != one.stdout


def test_host_program_sample_loop_in_pieces():
    """--spp-chunk runs the sample loop (main.cu:61-67) as several launches whose exact sums stay in device memory; the
    image is the single launch's, byte for byte, also with row sets on several tiles."""
    args = ["--width", "120", "--height", "68", "--spp", "24", "--format", "pfm"]
    whole = subprocess.run([_need("pathtracer")] + args, capture_output=True, timeout=120)
    assert whole.returncode == 0, whole.stderr[-400:]
    for extra in (["--spp-chunk", "7"], ["--spp-chunk", "8", "--tiles", "3"], ["--spp-chunk", "1", "--math", "fast"], ["--spp-chunk", "100"]):
        r = subprocess.run([_need("pathtracer")] + args + extra, capture_output=True, timeout=120)
        assert r.returncode == 0, r.stderr[-400:]
        assert r.stdout == whole.stdout, extra


def test_host_program_stops_and_goes_on_from_its_state_file(tmp_path):
    """--checkpoint writes the pixels' exact sums and the samples done after every launch; --stop-after ends the run
    early (exit status 3, no image); the same command then goes on from the file and prints the file one uninterrupted
    launch prints, byte for byte — also with several row sets, each with its own state file.  A state file of another
    frame is refused (exit status 2)."""
    args = ["--width", "120", "--height", "68", "--spp", "24", "--format", "pfm"]
    whole = subprocess.run([_need("pathtracer")] + args, capture_output=True, timeout=120)
    assert whole.returncode == 0, whole.stderr[-400:]
    for name, extra in (("one", []), ("tiles", ["--tiles", "3"])):
        ck = str(tmp_path / (name + ".sums"))
        cmd = [_need("pathtracer")] + args + extra + ["--spp-chunk", "5", "--checkpoint", ck]
        first = subprocess.run(cmd + ["--stop-after", "9"], capture_output=True, timeout=120)
        assert first.returncode == 3 and first.stdout == b"" and b"stopped after 10 of 24 samples" in first.stderr, first.stderr[-400:]
        assert os.path.exists(ck + (".0" if extra else ""))
        second = subprocess.run(cmd + ["--stop-after", "15"], capture_output=True, timeout=120)
        assert second.returncode == 3 and b"stopped after 15 of 24 samples" in second.stderr, second.stderr[-400:]
        last = subprocess.run(cmd, capture_output=True, timeout=120)
        assert last.returncode == 0, last.stderr[-400:]
        assert last.stdout == whole.stdout, name
        again = subprocess.run(cmd, capture_output=True, timeout=120)             # a finished state: nothing left to trace
        assert again.returncode == 0 and again.stdout == whole.stdout
        other = subprocess.run(cmd + ["--seed", "1"], capture_output=True, timeout=120)
        assert other.returncode == 2 and b"is not the state of this frame" in other.stderr, other.stderr[-400:]


def test_host_program_rccl_gather_on_one_device():
    """--gather rccl: the tiles travel through pt_comm_init_all + pt_gather_framebuffer instead of per-tile host
    copies (one GPU here, so a communicator of one rank); same file."""
    args = ["--width", "80", "--height", "50", "--spp", "4", "--format", "pfm"]
    one = subprocess.run([_need("pathtracer")] + args, capture_output=True, timeout=120)
    r = subprocess.run([_need("pathtracer")] + args + ["--gather", "rccl", "--stats"], capture_output=True, timeout=180)
    assert b'"bitwise_equal_to_single_gpu": true' in r.stderr, r.stderr[-400:]
    assert one.returncode == 0 and r.returncode == 0, r.stderr.decode()
    assert r.stdout == one.stdout
