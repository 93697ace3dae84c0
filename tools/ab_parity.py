"""A/B timing of two builds of the library in PT_MATH_PARITY on the 1080p frame, with a bitwise comparison of their images
(development aid).    python tools/ab_parity.py pathtracer_amd/libpt_hip.so pathtracer_amd/libpt_hip_exp.so"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pathtracer_amd import binding as b
libs = []
for path in sys.argv[1:]:
    L = C.CDLL(os.path.join(ROOT, path))
    for name in ("pt_scene_upload", "pt_render", "pt_scene_free"):
        getattr(L, name).argtypes = getattr(b.lib, name).argtypes
    prims, n, mats, nm = b.default_scene()
    h = C.c_void_p()
    assert L.pt_scene_upload(prims, n, mats, nm, 0, C.byref(h)) == 0
    libs.append((path, L, h))
cfg = b.make_config(1920, 1080, 64, math_mode=b.PT_MATH_PARITY)
fbs = {p: torch.zeros((1080, 1920, 3), device="cuda") for p, _, _ in libs}
times = {p: [] for p, _, _ in libs}
for rnd in range(7):
    for path, L, h in libs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); assert L.pt_render(h, C.byref(cfg), C.c_void_p(fbs[path].data_ptr()), None) == 0; e1.record()
        torch.cuda.synchronize()
        if rnd >= 2: times[path].append(e0.elapsed_time(e1))
ref = None
for path in times:
    t = sorted(times[path])
    if ref is None: ref = fbs[path]
    print("%-40s min %.3f median %.3f ms  bitwise equal to first: %s" % (path, t[0], t[len(t) // 2], bool(torch.equal(ref, fbs[path]))))
