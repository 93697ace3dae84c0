"""Per-phase shader-cycle shares of the fast render loop (development aid).
Needs the instrumented build:  make -C pathtracer_amd/csrc prof
    python tools/phase_profile.py [scene width height spp depth]"""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pathtracer_amd import binding as b

prof = C.CDLL(os.path.join(ROOT, "pathtracer_amd", "libpt_hip_prof.so"))
for name in ("pt_scene_upload", "pt_render", "pt_scene_free"):
    getattr(prof, name).argtypes = getattr(b.lib, name).argtypes
prof.pt_dev_phase_cycles.argtypes = [C.c_void_p, C.c_int]
scene = sys.argv[1] if len(sys.argv) > 1 else "default"
w, h, spp, depth = (int(x) for x in sys.argv[2:6]) if len(sys.argv) > 5 else (1920, 1080, 64, 5)
prims, n, mats, nm = b.default_scene() if scene == "default" else (b.glass_grid_scene(int(scene[5:] or 5)) if scene.startswith("glass") else b.synthetic_scene(5000, 4989))
hs = C.c_void_p()
assert prof.pt_scene_upload(prims, n, mats, nm, 0, C.byref(hs)) == 0
prof.pt_scene_tune.argtypes = b.lib.pt_scene_tune.argtypes
if os.environ.get("PHASE_PLAIN") == "1":           # the plain loop instead of the one that sets glass hits aside
    assert prof.pt_scene_tune(hs, b.PT_TUNE_GLASS_QUEUE, -1) == 0
cfg = b.make_config(w, h, spp, n_bounces=depth)
fb = torch.zeros((h, w, 3), device="cuda")
buf = (C.c_ulonglong * 24)()
assert prof.pt_render(hs, C.byref(cfg), C.c_void_p(fb.data_ptr()), None) == 0
prof.pt_dev_phase_cycles(buf, 1)
assert prof.pt_render(hs, C.byref(cfg), C.c_void_p(fb.data_ptr()), None) == 0
prof.pt_dev_phase_cycles(buf, 1)
names = ["refill", "nearest", "surface+material", "emissive/Beer/Fresnel/mode", "direction", "throughput+roulette", "accumulate", "primitives outside the hierarchy"]
tot = float(sum(buf[:8]))
for k in range(8):
    print("%-28s %6.2f %% of cycles, %5.1f %% of lanes active at the stamp, %10d stamps, %7.0f cycles per stamp" % (
        names[k], 100.0 * buf[k] / tot, 100.0 * buf[8 + k] / (64.0 * buf[k]) if buf[k] else 0.0, buf[16 + k], buf[k] / max(1, buf[16 + k])))
