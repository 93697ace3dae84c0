"""Host code under AddressSanitizer + UndefinedBehaviorSanitizer: the CPU oracle and the product's scene-text parser
and image writers, driven by tests/cpp/sanitize_host.cc (ordinary, degenerate and malformed inputs).  Host code only:
GPU sanitizers do not exist for this target."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_and_parser_under_asan_ubsan(tmp_path):
    cxx = shutil.which("g++")
    if cxx is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "sanitize_host")
    cmd = [cxx, "-std=c++17", "-O1", "-g", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-static-libasan", "-static-libubsan", "-pthread",
           "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
           os.path.join(ROOT, "tests", "cpp", "sanitize_host.cc"), os.path.join(ROOT, "oracle", "oracle_capi.cpp"),
           os.path.join(ROOT, "pathtracer_amd", "csrc", "pt_io.cpp"), "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if b.returncode != 0 and ("asan" in b.stderr or "ubsan" in b.stderr) and "error:" not in b.stderr:
        pytest.skip("sanitizer runtimes are not installed: " + b.stderr[-300:])
    assert b.returncode == 0, b.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, os.path.join(ROOT, "scenes", "default.scene")], capture_output=True, text=True, timeout=600, env=env)
    if "ASan runtime does not come first" in r.stderr:
        pytest.skip("another library is preloaded ahead of the sanitizer runtime on this machine")
    assert r.returncode == 0 and "sanitize_host: ok" in r.stdout, (r.stdout + r.stderr)[-3000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
