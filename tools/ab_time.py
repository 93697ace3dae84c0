"""A/B timing of two builds of the library on the 1080p frame, interleaved rounds in one process (development aid).
    python tools/ab_time.py pathtracer_amd/libpt_hip.so pathtracer_amd/libpt_hip_exp.so
AB_SCENE=synthetic|glassN (default: the default scene), AB_SPP, AB_DEPTH, AB_MODE (math_mode) choose another frame; a library given as
path@knob=value[,knob=value] gets pt_scene_tune(knob, value) on its scene (knob numbers of include/pt_hip.h)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pathtracer_amd import binding as b
libs = []
for arg in sys.argv[1:]:
    path, _, tunes = arg.partition("@")
    L = C.CDLL(os.path.join(ROOT, path))
    for name in ("pt_scene_upload", "pt_render", "pt_scene_free", "pt_scene_tune"):
        getattr(L, name).argtypes = getattr(b.lib, name).argtypes
    scene_name = os.environ.get("AB_SCENE", "default")
    prims, n, mats, nm = (b.synthetic_scene(5000, 4989, 12345) if scene_name == "synthetic" else
                          (b.glass_grid_scene(int(scene_name[5:])) if scene_name.startswith("glass") else b.default_scene()))
    h = C.c_void_p()
    assert L.pt_scene_upload(prims, n, mats, nm, 0, C.byref(h)) == 0
    for kv in filter(None, tunes.split(",")):
        k, v = kv.split("=")
        assert L.pt_scene_tune(h, int(k), int(v)) == 0, kv
    libs.append((arg, L, h))
cfg = b.make_config(1920, 1080, int(os.environ.get("AB_SPP", "64")), n_bounces=int(os.environ.get("AB_DEPTH", "5")),
                    math_mode=int(os.environ.get("AB_MODE", str(b.PT_MATH_FAST))))
fb = torch.zeros((1080, 1920, 3), device="cuda")
times = {p: [] for p, _, _ in libs}
sums = {}
for rnd in range(12):
    for path, L, h in libs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); assert L.pt_render(h, C.byref(cfg), C.c_void_p(fb.data_ptr()), None) == 0; e1.record()
        torch.cuda.synchronize()
        if rnd >= 2: times[path].append(e0.elapsed_time(e1))
        if rnd == 0: sums[path] = float(fb.double().sum())
for path in times:
    t = sorted(times[path])
    print("%-40s min %.3f median %.3f ms  (checksum %.6f)" % (path, t[0], t[len(t) // 2], sums[path]))
