"""Where the lanes are during the hierarchy walk (development aid).  Needs a library built with -DPT_WALK_CENSUS:
    python tools/walk_census.py pathtracer_amd/libpt_hip_census.so [spp] [knob=value ...]     (pt_scene_tune knobs)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pathtracer_amd import binding as b
L = C.CDLL(os.path.join(ROOT, sys.argv[1]))
for name in ("pt_scene_upload", "pt_render", "pt_scene_free", "pt_scene_tune"):
    getattr(L, name).argtypes = getattr(b.lib, name).argtypes
L.pt_dev_walk_census.argtypes = [C.c_void_p, C.c_int]
prims, n, mats, nm = b.synthetic_scene(5000, 4989, 12345)
h = C.c_void_p()
assert L.pt_scene_upload(prims, n, mats, nm, 0, C.byref(h)) == 0
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 4
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    assert L.pt_scene_tune(h, int(k), int(v)) == 0
cfg = b.make_config(1920, 1080, spp)
fb = torch.zeros((1080, 1920, 3), device="cuda")
buf = (C.c_ulonglong * 16)()
L.pt_dev_walk_census(buf, 1)
assert L.pt_render(h, C.byref(cfg), C.c_void_p(fb.data_ptr()), None) == 0
L.pt_dev_walk_census(buf, 1)
trips = float(buf[0]); segs = 1920 * 1080 * spp
print("node-loop trips per wave-trip: lanes visiting %.1f, holding a leaf %.1f, finished %.1f, about to pop %.1f, in the wave %.1f" %
      tuple(buf[i] / trips for i in (1, 2, 3, 4, 5)))
print("trips %.0f (%.1f lane-visits per sample), leaf phases %.0f with %.1f lanes each, %.2f leaf tests per sample" %
      (trips, buf[1] / segs, buf[6], buf[7] / max(1.0, float(buf[6])), buf[7] / segs))
if buf[8]:
    print("loop with camera batches: %d tiles, %.1f %% with a candidate list (%.1f candidates each), %.0f camera batches parking %.1f entries each, "
          "%.0f main-loop iterations with %.1f lanes holding a path" %
          (buf[8], 100.0 * buf[9] / buf[8], buf[10] / max(1.0, float(buf[9])), buf[11], buf[12] / max(1.0, float(buf[11])), buf[13], buf[14] / max(1.0, float(buf[13]))))
