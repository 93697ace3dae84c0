"""ctypes binding of the CPU oracle (oracle/liboracle*.so).

Test infrastructure only: imported by tests/, by __graft_entry__.smoke() and by
the cpu_baseline leg of bench.py.  The product package never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ORACLE_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")

PLANE, CUBE, CYLINDER, SPHERE = 0, 1, 2, 3


class Material(C.Structure):
    _fields_ = [
        ("albedo", C.c_float * 3), ("emissive", C.c_float * 3), ("f0", C.c_float),
        ("specularRoughness", C.c_float), ("specularColor", C.c_float * 3), ("IOR", C.c_float),
        ("transparency", C.c_float), ("refractionRoughness", C.c_float),
        ("refractionColor", C.c_float * 3), ("bsdfType", C.c_int),
    ]


class Prim(C.Structure):
    _fields_ = [("type", C.c_int), ("material", C.c_int), ("transform", C.c_float * 12), ("bounds", C.c_float * 4)]


class Config(C.Structure):
    _fields_ = [
        ("camera_pos", C.c_float * 3), ("max_x", C.c_int), ("max_y", C.c_int),
        ("focal_length", C.c_float), ("view_port_height", C.c_float),
        ("n_samples", C.c_int), ("n_bounces", C.c_int), ("seed", C.c_uint),
        ("row_begin", C.c_int), ("row_end", C.c_int), ("bsdf_model", C.c_int),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("samples", C.c_uint64), ("segments", C.c_uint64), ("hits", C.c_uint64), ("tests", C.c_uint64 * 4),
        ("sphere_roots", C.c_uint64), ("cyl_roots", C.c_uint64), ("emissive", C.c_uint64), ("beer", C.c_uint64),
        ("fresnel", C.c_uint64), ("fresnel_dense", C.c_uint64), ("mode", C.c_uint64 * 4), ("flops", C.c_double),
    ]

    def as_dict(self):
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            d[name] = list(v) if hasattr(v, "__len__") else v
        return d


def build(force=False):
    """Compile the oracle with its own Makefile (g++ only)."""
    need = force or not all(os.path.exists(os.path.join(ORACLE_DIR, n)) for n in ("liboracle.so", "liboracle_count.so"))
    if need:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])


_libs = {}


def lib(count=False, variant=None):
    """The oracle library: plain, with event counters (count), or another compiler's build (variant="clang")."""
    name = "liboracle_%s.so" % variant if variant else ("liboracle_count.so" if count else "liboracle.so")
    if name not in _libs:
        path = os.path.join(ORACLE_DIR, name)
        if not os.path.exists(path):
            build(force=variant is not None)
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        L = C.CDLL(path)
        L.oracle_scene_create.restype = C.c_void_p
        L.oracle_scene_create.argtypes = [C.POINTER(Prim), C.c_int, C.POINTER(Material), C.c_int]
        L.oracle_scene_free.argtypes = [C.c_void_p]
        L.oracle_default_scene.argtypes = [C.POINTER(Prim), C.POINTER(Material), C.POINTER(C.c_int)]
        L.oracle_focal_length.restype = C.c_float
        L.oracle_focal_length.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float]
        L.oracle_render.argtypes = [C.c_void_p, C.POINTER(Config), C.c_void_p, C.c_int, C.POINTER(Stats)]
        L.oracle_render_rows.argtypes = [C.c_void_p, C.POINTER(Config), C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(Stats)]
        L.oracle_ppm.restype = C.c_long
        L.oracle_ppm.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_long]
        _libs[name] = L
    return _libs[name]


def _fp(a):
    return a.ctypes.data_as(C.c_void_p)


def default_threads():
    """Host threads for the oracle: the process's CPU share (OMP_NUM_THREADS on the GPU box), not the machine's CPU count."""
    env = os.environ.get("OMP_NUM_THREADS")
    if env and env.isdigit() and int(env) > 0:
        return int(env)
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        return os.cpu_count() or 1


def default_scene():
    prims = (Prim * 11)()
    mats = (Material * 8)()
    nm = C.c_int(0)
    n = lib().oracle_default_scene(prims, mats, C.byref(nm))
    return prims, n, mats, nm.value


def focal_length(nx, ny, v_height=5.0, fov=45.0):
    return float(lib().oracle_focal_length(nx, ny, v_height, fov))


def make_config(width, height, spp, n_bounces=5, seed=0, row_begin=0, row_end=None, camera=(0.0, 0.0, 0.0), bsdf_model=0):
    cfg = Config()
    cfg.camera_pos[:] = camera
    cfg.max_x, cfg.max_y = width, height
    cfg.focal_length = focal_length(width, height)
    cfg.view_port_height = 5.0
    cfg.n_samples, cfg.n_bounces, cfg.seed = spp, n_bounces, seed
    cfg.row_begin = row_begin
    cfg.row_end = height if row_end is None else row_end
    cfg.bsdf_model = bsdf_model
    return cfg


class Scene:
    """Owns an oracle scene built from flat (Prim[], Material[]) arrays."""

    def __init__(self, prims, n, mats, n_mats, count=False, variant=None):
        self.L = lib(count, variant)
        self.prims, self.n, self.mats, self.n_mats = prims, n, mats, n_mats
        self.h = self.L.oracle_scene_create(prims, n, mats, n_mats)
        if not self.h:
            raise RuntimeError("oracle_scene_create failed")

    def close(self):
        if self.h:
            self.L.oracle_scene_free(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def render(self, cfg, threads=None, want_stats=False, rows=None):
        """Returns fb[H,W,3] float32 (row 0 = image bottom); rows outside the tile (or not in `rows`) stay 0."""
        fb = np.zeros((cfg.max_y, cfg.max_x, 3), dtype=np.float32)
        st = Stats()
        threads = threads or default_threads()
        if rows is not None:
            rows = np.ascontiguousarray(rows, np.int32)
            rc = self.L.oracle_render_rows(self.h, C.byref(cfg), _fp(rows), int(rows.shape[0]), _fp(fb), int(threads), C.byref(st))
        else:
            rc = self.L.oracle_render(self.h, C.byref(cfg), _fp(fb), int(threads), C.byref(st))
        if rc != 0:
            raise RuntimeError("oracle_render rc=%d" % rc)
        return (fb, st) if want_stats else fb

    def nearest(self, o, d):
        o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32)
        n = o.shape[0]
        prim = np.zeros(n, np.int32); t = np.zeros(n, np.float32); inside = np.zeros(n, np.int32)
        pos = np.zeros((n, 3), np.float32); nrm = np.zeros((n, 3), np.float32)
        self.L.oracle_nearest_batch(C.c_void_p(self.h), _fp(o), _fp(d), C.c_int(n), _fp(prim), _fp(t), _fp(pos), _fp(nrm), _fp(inside))
        return dict(prim=prim, t=t, pos=pos, nrm=nrm, inside=inside)

    def trace_paths(self, o, d, pixel, sample, seed=0, n_bounces=5, bsdf_model=0):
        o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32)
        pixel = np.ascontiguousarray(pixel, np.uint32); sample = np.ascontiguousarray(sample, np.uint32)
        n = o.shape[0]
        rgb = np.zeros((n, 3), np.float32)
        fn = self.L.oracle_trace_paths_alt if bsdf_model == 1 else self.L.oracle_trace_paths
        fn(C.c_void_p(self.h), _fp(o), _fp(d), _fp(pixel), _fp(sample), C.c_int(n), C.c_uint(seed), C.c_int(n_bounces), _fp(rgb))
        return rgb


def trace_paths_end(scene, o, d, pixel, sample, seed=0, n_bounces=5, bsdf_model=0):
    """tracePath with the ray it leaves in `r`: (rgb, end_o, end_d)."""
    o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32)
    pixel = np.ascontiguousarray(pixel, np.uint32); sample = np.ascontiguousarray(sample, np.uint32)
    n = o.shape[0]
    rgb, eo, ed = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    scene.L.oracle_trace_paths_end(C.c_void_p(scene.h), _fp(o), _fp(d), _fp(pixel), _fp(sample), C.c_int(n), C.c_uint(seed),
                                   C.c_int(n_bounces), C.c_int(bsdf_model), _fp(rgb), _fp(eo), _fp(ed))
    return rgb, eo, ed


def raypath(prim, o, d, filter=True):
    """Shape::getIntersections (filter) / getRayPath (no filter, local ray) for one primitive."""
    o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32)
    n = o.shape[0]
    out = dict(n=np.zeros(n, np.int32))
    for s in ("1", "2"):
        out["hit" + s] = np.zeros(n, np.int32); out["t" + s] = np.zeros(n, np.float32)
        out["pos" + s] = np.zeros((n, 3), np.float32); out["nrm" + s] = np.zeros((n, 3), np.float32)
    lib().oracle_raypath_batch(C.byref(prim), _fp(o), _fp(d), C.c_int(n), C.c_int(1 if filter else 0), _fp(out["n"]),
                               _fp(out["hit1"]), _fp(out["t1"]), _fp(out["pos1"]), _fp(out["nrm1"]),
                               _fp(out["hit2"]), _fp(out["t2"]), _fp(out["pos2"]), _fp(out["nrm2"]))
    return out


BSDF_ARITY = {0: (2, 3), 1: (3, 1), 2: (10, 1), 3: (6, 3), 4: (7, 3), 5: (7, 3), 6: (3, 3)}


def bsdf_batch(fn, x):
    """sampleHemisphere / FresnelSchlick / getFresnelRatio / reflect / refract / lerp / normalized on rows of x."""
    n_in, n_out = BSDF_ARITY[fn]
    x = np.ascontiguousarray(x, np.float32).reshape(-1, n_in)
    out = np.zeros((x.shape[0], n_out), np.float32)
    rc = lib().oracle_bsdf_batch(C.c_int(fn), _fp(x), C.c_int(x.shape[0]), _fp(out))
    assert rc == 0
    return out


def rng(seed, pixel, sample, n_slots):
    bits = np.zeros(n_slots, np.uint32); u = np.zeros(n_slots, np.float32)
    lib().oracle_rng(C.c_uint(seed), C.c_uint(pixel), C.c_uint(sample), C.c_int(n_slots), _fp(bits), _fp(u))
    return bits, u


def camera_rays(cfg, pi, pj, sample):
    pi = np.ascontiguousarray(pi, np.int32); pj = np.ascontiguousarray(pj, np.int32)
    sample = np.ascontiguousarray(sample, np.uint32)
    n = pi.shape[0]
    o = np.zeros((n, 3), np.float32); d = np.zeros((n, 3), np.float32)
    lib().oracle_camera_rays(C.byref(cfg), _fp(pi), _fp(pj), _fp(sample), C.c_int(n), _fp(o), _fp(d))
    return o, d


def ppm_text(fb):
    h, w, _ = fb.shape
    cap = 16 + h * w * 40
    buf = C.create_string_buffer(cap)
    fbc = np.ascontiguousarray(fb, np.float32)
    n = lib().oracle_ppm(_fp(fbc), w, h, buf, cap)
    if n < 0:
        raise RuntimeError("oracle_ppm buffer too small")
    return buf.raw[:n]


def make_prim(ptype, transform=None, translate=None, bounds=None, material=0):
    """Convenience: a Prim with identity linear part unless `transform` (12 floats) is given."""
    p = Prim()
    p.type, p.material = ptype, material
    t = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0] if transform is None else list(transform)
    if translate is not None:
        t[3], t[7], t[11] = translate
    p.transform[:] = t
    big = float(np.finfo(np.float32).max)
    p.bounds[:] = bounds if bounds is not None else (-big, -big, big, big)
    return p
