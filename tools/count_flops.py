"""Counts the algorithmic flops per sample of a config with the instrumented CPU
oracle (oracle/liboracle_count.so) and writes profiles/flops_per_sample.json.

The flop weight of each event (camera ray, primitive test, finalize, shade,
Fresnel, direction by mode) is fixed in oracle/pt_oracle.hpp Stats::flops(); this
script only supplies the event counts for a scene / resolution / spp / seed.

    python tools/count_flops.py default 1920 1080 64
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding as ob  # noqa: E402


def scene_by_name(name):
    if name == "default":
        return ob.default_scene()
    import ctypes as C
    from pathtracer_amd import binding as b
    if name.startswith("glass"):
        prims, n, mats, nm = b.glass_grid_scene(int(name[5:] or 5))
    elif name.startswith("synthetic"):
        prims, n, mats, nm = b.synthetic_scene(5000, 4989, 12345)
    else:
        raise SystemExit("unknown scene " + name)
    # same memory layout: reinterpret the product's arrays as the oracle's
    op = (ob.Prim * n).from_buffer_copy(bytes(prims)[: n * C.sizeof(ob.Prim)])
    om = (ob.Material * nm).from_buffer_copy(bytes(mats)[: nm * C.sizeof(ob.Material)])
    return op, n, om, nm


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "default"
    w, h, spp = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (1920, 1080, 64)
    nb = int(sys.argv[5]) if len(sys.argv) > 5 else 5
    row_step = int(sys.argv[6]) if len(sys.argv) > 6 else 1
    prims, n, mats, nm = scene_by_name(name)
    sc = ob.Scene(prims, n, mats, nm, count=True)
    cfg = ob.make_config(w, h, spp, n_bounces=nb)
    rows = list(range(0, h, row_step)) if row_step > 1 else None
    t = time.time()
    _, st = sc.render(cfg, want_stats=True, rows=rows)
    dt = time.time() - t
    key = "%s_%dx%d_%dspp_depth%d_seed0" % (name, w, h, spp, nb) + ("" if row_step == 1 else "_rows%%%d" % row_step)
    rec = dict(st.as_dict(), flops_per_sample=st.flops / st.samples, segments_per_sample=st.segments / st.samples,
               n_prims=n, oracle_seconds=dt, threads=ob.default_threads())
    path = os.path.join(ROOT, "profiles", "flops_per_sample.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[key] = rec
    json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    print(key, "flops/sample %.2f segments/sample %.4f (%.1f s)" % (rec["flops_per_sample"], rec["segments_per_sample"], dt))


if __name__ == "__main__":
    main()
