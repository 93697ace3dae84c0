"""ctypes binding of libpt_hip.so (include/pt_hip.h).

The library is the product; this module only marshals arguments.  Device memory
is held in torch tensors (torch is used for allocation, streams and
torch.distributed, nothing else) and passed to the C ABI as raw device pointers.
There is no CPU fallback: if the shared library is missing, importing fails.
"""
import ctypes as C
import os

import numpy as np
import torch  # noqa: F401  (first: torch must bring up ITS HIP runtime before libpt_hip.so loads one)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpt_hip.so")

PT_PLANE, PT_CUBE, PT_CYLINDER, PT_SPHERE = 0, 1, 2, 3
PT_MATH_PARITY, PT_MATH_FAST, PT_MATH_FAST_SCALAR, PT_MATH_FAST_LDS = 0, 1, 2, 3
PT_FB_AOS, PT_FB_SOA = 0, 1
PT_BSDF_REFERENCE, PT_BSDF_CLASSES = 0, 1


class pt_material(C.Structure):
    _fields_ = [
        ("albedo", C.c_float * 3), ("emissive", C.c_float * 3), ("f0", C.c_float),
        ("specularRoughness", C.c_float), ("specularColor", C.c_float * 3), ("IOR", C.c_float),
        ("transparency", C.c_float), ("refractionRoughness", C.c_float),
        ("refractionColor", C.c_float * 3), ("bsdfType", C.c_int32),
    ]


class pt_prim(C.Structure):
    _fields_ = [("type", C.c_int32), ("material", C.c_int32), ("transform", C.c_float * 12), ("bounds", C.c_float * 4)]


class pt_config(C.Structure):
    _fields_ = [
        ("camera_pos", C.c_float * 3), ("max_x", C.c_int32), ("max_y", C.c_int32),
        ("focal_length", C.c_float), ("view_port_height", C.c_float),
        ("n_samples", C.c_int32), ("n_bounces", C.c_int32), ("seed", C.c_uint32),
        ("row_begin", C.c_int32), ("row_end", C.c_int32), ("fb_layout", C.c_int32), ("math_mode", C.c_int32),
        ("row_step", C.c_int32), ("fb_compact", C.c_int32), ("bsdf_model", C.c_int32), ("tile_log2", C.c_int32),
    ]


class pt_upload_options(C.Structure):
    _fields_ = [("bvh_min_prims", C.c_int32), ("bvh_leaf_prims", C.c_int32), ("general_kinds", C.c_int32), ("bvh_split", C.c_int32)]


PT_GENERAL_PLANES, PT_GENERAL_CUBES, PT_GENERAL_SPHERES, PT_GENERAL_ALL = 1, 2, 4, 7
PT_BVH_SPLIT_MEDIAN, PT_BVH_SPLIT_AREA = 1, 2
PT_TUNE_TILE_LOG2, PT_TUNE_WALK_BUDGET, PT_TUNE_SHADE_GATE, PT_TUNE_BVH_WALK, PT_TUNE_GLASS_QUEUE, PT_TUNE_GLASS_GATE, PT_TUNE_LEAF_GATE = range(7)


class PtError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "pathtracer_amd: %s is missing; build it with `make -C pathtracer_amd/csrc` "
            "(or __graft_entry__.build()). There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i32, u32, f32 = C.c_void_p, C.c_int, C.c_uint32, C.c_float
    L.pt_last_error.restype = C.c_char_p
    L.pt_version.restype = C.c_char_p
    L.pt_focal_length.restype = f32
    L.pt_focal_length.argtypes = [i32, i32, f32, f32]
    L.pt_default_scene.argtypes = [C.POINTER(pt_prim), C.POINTER(pt_material), C.POINTER(i32)]
    L.pt_glass_grid_scene.argtypes = [i32, C.POINTER(pt_prim), C.POINTER(pt_material), C.POINTER(i32)]
    L.pt_synthetic_scene.argtypes = [i32, i32, u32, C.POINTER(pt_prim), C.POINTER(pt_material), C.POINTER(i32)]
    L.pt_scene_upload.argtypes = [C.POINTER(pt_prim), i32, C.POINTER(pt_material), i32, i32, C.POINTER(vp)]
    L.pt_scene_upload_opt.argtypes = [C.POINTER(pt_prim), i32, C.POINTER(pt_material), i32, i32, C.POINTER(pt_upload_options), C.POINTER(vp)]
    L.pt_scene_tune.argtypes = [vp, i32, i32]
    L.pt_scene_free.argtypes = [vp]
    L.pt_scene_device.argtypes = [vp]
    L.pt_scene_bvh_nodes.argtypes = [vp]
    L.pt_render.argtypes = [vp, C.POINTER(pt_config), vp, vp]
    L.pt_render_counted.argtypes = [vp, C.POINTER(pt_config), vp, vp, vp]
    L.pt_render_samples.argtypes = [vp, C.POINTER(pt_config), i32, i32, vp, vp]
    L.pt_sums_count.restype = C.c_longlong
    L.pt_sums_count.argtypes = [C.POINTER(pt_config)]
    L.pt_resolve.argtypes = [vp, C.POINTER(pt_config), vp, vp, vp]
    L.pt_raypath_batch.argtypes = [vp, i32, vp, vp, i32, i32] + [vp] * 9 + [vp]
    L.pt_nearest_batch.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp, vp, vp]
    L.pt_nearest_batch_mode.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp]
    L.pt_trace_paths.argtypes = [vp, vp, vp, vp, vp, i32, u32, i32, i32, vp, vp]
    L.pt_trace_paths_alt.argtypes = [vp, vp, vp, vp, vp, i32, u32, i32, i32, vp, vp]
    L.pt_trace_paths_end.argtypes = [vp, vp, vp, vp, vp, i32, u32, i32, i32, i32, vp, vp, vp, vp]
    L.pt_rng_draws.argtypes = [i32, u32, u32, u32, i32, vp, vp, vp]
    L.pt_bsdf_batch.argtypes = [i32, i32, i32, vp, i32, vp, vp]
    L.pt_bsdf_arity.argtypes = [i32, C.POINTER(i32), C.POINTER(i32)]
    L.pt_ppm_p3.restype = C.c_long
    L.pt_ppm_p3.argtypes = [vp, i32, i32, C.c_char_p, C.c_long]
    L.pt_ppm_p6.restype = C.c_long
    L.pt_ppm_p6.argtypes = [vp, i32, i32, C.c_char_p, C.c_long]
    L.pt_pfm.restype = C.c_long
    L.pt_pfm.argtypes = [vp, i32, i32, C.c_char_p, C.c_long]
    L.pt_scene_parse.argtypes = [C.c_char_p, C.POINTER(pt_prim), i32, C.POINTER(pt_material), i32, C.POINTER(i32)]
    L.pt_tile_rows.argtypes = [i32, i32]
    L.pt_comm_unique_id.argtypes = [C.c_char_p]
    L.pt_comm_init_rank.argtypes = [C.c_char_p, i32, i32, i32, C.POINTER(vp)]
    L.pt_comm_init_all.argtypes = [i32, C.POINTER(i32), C.POINTER(vp)]
    L.pt_comm_free.argtypes = [vp]
    L.pt_comm_rank.argtypes = [vp]
    L.pt_comm_world.argtypes = [vp]
    L.pt_gather_framebuffer.argtypes = [vp, vp, vp, vp, i32, i32, vp]
    L.pt_assemble_rows.argtypes = [vp, vp, i32, i32, i32, vp]
    L.pt_gather_framebuffer_noalloc.argtypes = [vp, vp, vp, i32, i32, vp]
    return L


lib = _load()

EXPORTS = [
    "pt_scene_upload", "pt_scene_upload_opt", "pt_scene_tune", "pt_scene_free", "pt_scene_device", "pt_scene_bvh_nodes", "pt_default_scene", "pt_glass_grid_scene",
    "pt_synthetic_scene", "pt_focal_length", "pt_render", "pt_render_counted", "pt_render_samples", "pt_sums_count", "pt_resolve", "pt_raypath_batch", "pt_nearest_batch", "pt_nearest_batch_mode",
    "pt_trace_paths", "pt_trace_paths_alt", "pt_trace_paths_end", "pt_rng_draws", "pt_bsdf_batch", "pt_bsdf_arity", "pt_ppm_p3", "pt_last_error", "pt_version",
    "pt_scene_parse", "pt_ppm_p6", "pt_pfm",
    "pt_tile_rows", "pt_comm_unique_id", "pt_comm_init_rank", "pt_comm_init_all", "pt_comm_free", "pt_comm_rank",
    "pt_comm_world", "pt_gather_framebuffer", "pt_assemble_rows", "pt_gather_framebuffer_noalloc",
]


def _check(rc):
    if rc != 0:
        raise PtError(lib.pt_last_error().decode(errors="replace").strip() or ("pt status %d" % rc))


def focal_length(nx, ny, v_height=5.0, fov=45.0):
    """Host-side focal length of main.cu:229-232 (integer nx/ny)."""
    return float(lib.pt_focal_length(nx, ny, v_height, fov))


def default_scene():
    """The reference's default scene (main.cu:73-195) as (prims, n, mats, n_mats)."""
    prims = (pt_prim * 11)()
    mats = (pt_material * 8)()
    nm = C.c_int(0)
    n = lib.pt_default_scene(prims, mats, C.byref(nm))
    return prims, n, mats, nm.value


def glass_grid_scene(n):
    prims = (pt_prim * (8 + n * n))()
    mats = (pt_material * 8)()
    nm = C.c_int(0)
    k = lib.pt_glass_grid_scene(n, prims, mats, C.byref(nm))
    return prims, k, mats, nm.value


def synthetic_scene(n_spheres, n_cubes, seed=12345):
    prims = (pt_prim * (11 + n_spheres + n_cubes))()
    mats = (pt_material * 8)()
    nm = C.c_int(0)
    k = lib.pt_synthetic_scene(n_spheres, n_cubes, seed, prims, mats, C.byref(nm))
    return prims, k, mats, nm.value


def make_config(width, height, spp, n_bounces=5, seed=0, row_begin=0, row_end=None, camera=(0.0, 0.0, 0.0),
                fb_layout=PT_FB_AOS, math_mode=PT_MATH_FAST, row_step=1, fb_compact=0, bsdf_model=PT_BSDF_REFERENCE, tile_log2=0):
    """config_t of main.cu:234 with the reference's camera (origin, v_height 5, fov 45)."""
    cfg = pt_config()
    cfg.camera_pos[:] = camera
    cfg.max_x, cfg.max_y = width, height
    cfg.focal_length = focal_length(width, height)
    cfg.view_port_height = 5.0
    cfg.n_samples, cfg.n_bounces, cfg.seed = spp, n_bounces, seed
    cfg.row_begin = row_begin
    cfg.row_end = height if row_end is None else row_end
    cfg.fb_layout, cfg.math_mode = fb_layout, math_mode
    cfg.row_step, cfg.fb_compact = row_step, fb_compact
    cfg.bsdf_model = bsdf_model
    cfg.tile_log2 = tile_log2
    return cfg


def rendered_rows(cfg):
    """Number of rows a pt_render call with this config writes."""
    step = cfg.row_step or 1
    return max(0, (cfg.row_end - cfg.row_begin + step - 1) // step)


def ppm_p3(fb):
    """PPM P3 bytes of a host [H,W,3] float32 framebuffer, as main.cu:240-253 prints it."""
    fb = np.ascontiguousarray(fb, np.float32)
    h, w, _ = fb.shape
    cap = 32 + h * w * 40
    buf = C.create_string_buffer(cap)
    n = lib.pt_ppm_p3(fb.ctypes.data_as(C.c_void_p), w, h, buf, cap)
    if n < 0:
        raise PtError("pt_ppm_p3: buffer too small")
    return buf.raw[:n]


def parse_scene(text, max_prims=None, max_mats=None):
    """Scene description text (see scenes/default.scene) -> (prims, n, mats, n_mats)."""
    if isinstance(text, str):
        text = text.encode()
    max_prims = max_prims or max(1, text.count(b"\n") + 1)
    max_mats = max_mats or max_prims
    prims = (pt_prim * max_prims)()
    mats = (pt_material * max_mats)()
    nm = C.c_int(0)
    n = lib.pt_scene_parse(text, prims, max_prims, mats, max_mats, C.byref(nm))
    if n < 0:
        raise PtError(lib.pt_last_error().decode(errors="replace").strip())
    return prims, n, mats, nm.value


def _write_image(fn, fb, bytes_per_pixel):
    fb = np.ascontiguousarray(fb, np.float32)
    h, w, _ = fb.shape
    cap = 64 + h * w * bytes_per_pixel
    buf = C.create_string_buffer(cap)
    n = fn(fb.ctypes.data_as(C.c_void_p), w, h, buf, cap)
    if n < 0:
        raise PtError("image buffer too small")
    return buf.raw[:n]


def ppm_p6(fb):
    """Binary PPM bytes of a host [H,W,3] framebuffer (clamped 8-bit)."""
    return _write_image(lib.pt_ppm_p6, fb, 3)


def pfm(fb):
    """PFM bytes (float32, bottom row first) of a host [H,W,3] framebuffer."""
    return _write_image(lib.pt_pfm, fb, 12)


def _torch():
    return torch


def _stream():
    return C.c_void_p(_torch().cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class Scene:
    """Device copy of a scene (pt_scene_upload / pt_scene_free)."""

    def __init__(self, prims, n, mats, n_mats, device=0, bvh_min_prims=0, bvh_leaf_prims=0, general_kinds=False, bvh_split=0):
        """The keyword options are pt_upload_options (0 / False = the library's defaults)."""
        self.n, self.n_mats, self.device = n, n_mats, device
        h = C.c_void_p(0)
        if bvh_min_prims or bvh_leaf_prims or general_kinds or bvh_split:
            opt = pt_upload_options(bvh_min_prims, bvh_leaf_prims, PT_GENERAL_ALL if general_kinds is True else int(general_kinds), bvh_split)
            _check(lib.pt_scene_upload_opt(prims, n, mats, n_mats, device, C.byref(opt), C.byref(h)))
        else:
            _check(lib.pt_scene_upload(prims, n, mats, n_mats, device, C.byref(h)))
        self.h = h

    def tune(self, knob, value):
        """pt_scene_tune: a launch parameter of this scene's renders (PT_TUNE_*; 0 = the library's choice)."""
        _check(lib.pt_scene_tune(self.h, knob, value))
        return self

    def close(self):
        if getattr(self, "h", None) and lib is not None:      # at interpreter shutdown the module's globals may be gone
            lib.pt_scene_free(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _dev(self):
        return _torch().device("cuda", self.device)

    @property
    def bvh_nodes(self):
        """Nodes of the bounding-volume hierarchy PT_MATH_FAST searches (0 = none, every primitive is tested)."""
        return lib.pt_scene_bvh_nodes(self.h)

    def render(self, cfg, out=None):
        """pt_render into a [H,W,3] (AoS) or [3,H,W] (SoA) float32 device tensor; rows outside the tile are untouched."""
        torch = _torch()
        rows = rendered_rows(cfg) if cfg.fb_compact else cfg.max_y
        shape = (rows, cfg.max_x, 3) if cfg.fb_layout == PT_FB_AOS else (3, rows, cfg.max_x)
        if out is None:
            out = torch.zeros(shape, dtype=torch.float32, device=self._dev())
        assert out.is_contiguous() and out.dtype == torch.float32 and out.numel() >= rows * cfg.max_x * 3
        with torch.cuda.device(self.device):
            _check(lib.pt_render(self.h, C.byref(cfg), _ptr(out), _stream()))
        return out

    def render_counted(self, cfg, out=None):
        """pt_render_counted: (image, counters) — counters[0] = path segments traced (int64 device tensor of 4)."""
        torch = _torch()
        rows = rendered_rows(cfg) if cfg.fb_compact else cfg.max_y
        shape = (rows, cfg.max_x, 3) if cfg.fb_layout == PT_FB_AOS else (3, rows, cfg.max_x)
        if out is None:
            out = torch.zeros(shape, dtype=torch.float32, device=self._dev())
        counters = torch.zeros(4, dtype=torch.int64, device=self._dev())
        with torch.cuda.device(self.device):
            _check(lib.pt_render_counted(self.h, C.byref(cfg), _ptr(out), _ptr(counters), _stream()))
        return out, counters

    def render_samples(self, cfg, sample_begin, sample_count, sums=None):
        """pt_render_samples: adds the 32.32 sums of samples [sample_begin, sample_begin + sample_count) to `sums`
        (an int64 device tensor of pt_sums_count(cfg) elements, made and zeroed here when None)."""
        torch = _torch()
        n = int(lib.pt_sums_count(C.byref(cfg)))
        if n < 0:
            raise PtError("pt_sums_count: bad image size or rows")
        if sums is None:
            sums = torch.zeros(n, dtype=torch.int64, device=self._dev())
        assert sums.is_contiguous() and sums.dtype == torch.int64 and sums.numel() >= n
        with torch.cuda.device(self.device):
            _check(lib.pt_render_samples(self.h, C.byref(cfg), sample_begin, sample_count, _ptr(sums), _stream()))
        return sums

    def resolve(self, cfg, sums, out=None):
        """pt_resolve: the float framebuffer of the rendered rows from the sums of all cfg.n_samples samples."""
        torch = _torch()
        rows = rendered_rows(cfg) if cfg.fb_compact else cfg.max_y
        shape = (rows, cfg.max_x, 3) if cfg.fb_layout == PT_FB_AOS else (3, rows, cfg.max_x)
        if out is None:
            out = torch.zeros(shape, dtype=torch.float32, device=self._dev())
        assert out.is_contiguous() and out.dtype == torch.float32 and out.numel() >= rows * cfg.max_x * 3
        with torch.cuda.device(self.device):
            _check(lib.pt_resolve(self.h, C.byref(cfg), _ptr(sums), _ptr(out), _stream()))
        return out

    def raypath(self, prim, o, d, filter=True):
        """Shape::getIntersections / getRayPath of one primitive for [n,3] device rays."""
        torch = _torch()
        n = o.shape[0]
        dev = self._dev()
        res = dict(n=torch.zeros(n, dtype=torch.int32, device=dev))
        for s in ("1", "2"):
            res["hit" + s] = torch.zeros(n, dtype=torch.int32, device=dev)
            res["t" + s] = torch.zeros(n, dtype=torch.float32, device=dev)
            res["pos" + s] = torch.zeros((n, 3), dtype=torch.float32, device=dev)
            res["nrm" + s] = torch.zeros((n, 3), dtype=torch.float32, device=dev)
        with torch.cuda.device(self.device):
            _check(lib.pt_raypath_batch(self.h, prim, _ptr(o), _ptr(d), n, 1 if filter else 0, _ptr(res["n"]),
                                        _ptr(res["hit1"]), _ptr(res["t1"]), _ptr(res["pos1"]), _ptr(res["nrm1"]),
                                        _ptr(res["hit2"]), _ptr(res["t2"]), _ptr(res["pos2"]), _ptr(res["nrm2"]), _stream()))
        return res

    def nearest(self, o, d, math_mode=PT_MATH_PARITY):
        """getNearestIntersection for rays o, d ([n][3] device tensors); math_mode PT_MATH_FAST runs the search of the
        fast render kernels (pt_nearest_batch_mode)."""
        torch = _torch()
        n = o.shape[0]
        dev = self._dev()
        res = dict(prim=torch.zeros(n, dtype=torch.int32, device=dev), t=torch.zeros(n, dtype=torch.float32, device=dev),
                   pos=torch.zeros((n, 3), dtype=torch.float32, device=dev),
                   nrm=torch.zeros((n, 3), dtype=torch.float32, device=dev),
                   inside=torch.zeros(n, dtype=torch.int32, device=dev))
        with torch.cuda.device(self.device):
            if math_mode == PT_MATH_PARITY:
                _check(lib.pt_nearest_batch(self.h, _ptr(o), _ptr(d), n, _ptr(res["prim"]), _ptr(res["t"]),
                                            _ptr(res["pos"]), _ptr(res["nrm"]), _ptr(res["inside"]), _stream()))
            else:
                _check(lib.pt_nearest_batch_mode(self.h, _ptr(o), _ptr(d), n, math_mode, _ptr(res["prim"]), _ptr(res["t"]),
                                                 _ptr(res["pos"]), _ptr(res["nrm"]), _ptr(res["inside"]), _stream()))
        return res

    def trace_paths_end(self, o, d, pixel, sample, seed=0, n_bounces=5, math_mode=PT_MATH_PARITY, bsdf_model=PT_BSDF_REFERENCE):
        """pt_trace_paths_end: (radiance, origin, direction of the ray tracePath leaves in `r`)."""
        torch = _torch()
        n = o.shape[0]
        rgb, eo, ed = (torch.zeros((n, 3), dtype=torch.float32, device=self._dev()) for _ in range(3))
        with torch.cuda.device(self.device):
            _check(lib.pt_trace_paths_end(self.h, _ptr(o), _ptr(d), _ptr(pixel), _ptr(sample), n, seed, n_bounces, math_mode, bsdf_model,
                                          _ptr(rgb), _ptr(eo), _ptr(ed), _stream()))
        return rgb, eo, ed

    def trace_paths(self, o, d, pixel, sample, seed=0, n_bounces=5, math_mode=PT_MATH_PARITY, bsdf_model=PT_BSDF_REFERENCE):
        torch = _torch()
        n = o.shape[0]
        rgb = torch.zeros((n, 3), dtype=torch.float32, device=self._dev())
        with torch.cuda.device(self.device):
            fn = lib.pt_trace_paths_alt if bsdf_model == PT_BSDF_CLASSES else lib.pt_trace_paths
            _check(fn(self.h, _ptr(o), _ptr(d), _ptr(pixel), _ptr(sample), n, seed, n_bounces, math_mode, _ptr(rgb), _stream()))
        return rgb


def rng_draws(seed, pixel, sample, n_slots, device=0):
    torch = _torch()
    dev = torch.device("cuda", device)
    bits = torch.zeros(n_slots, dtype=torch.int32, device=dev)
    u = torch.zeros(n_slots, dtype=torch.float32, device=dev)
    with torch.cuda.device(device):
        _check(lib.pt_rng_draws(device, seed, pixel, sample, n_slots, _ptr(bits), _ptr(u), _stream()))
    return bits, u


PT_FN_SAMPLE_HEMISPHERE, PT_FN_FRESNEL_SCHLICK, PT_FN_FRESNEL_RATIO, PT_FN_REFLECT, PT_FN_REFRACT, PT_FN_LERP, PT_FN_NORMALIZED = range(7)


def bsdf_batch(fn, x, math_mode=PT_MATH_PARITY, device=0):
    """pt_bsdf_batch: the helper functions of bsdf.h / vec_math.h on the rows of the device tensor x."""
    torch = _torch()
    n_in, n_out = C.c_int(0), C.c_int(0)
    _check(lib.pt_bsdf_arity(fn, C.byref(n_in), C.byref(n_out)))
    x = x.contiguous().view(-1, n_in.value)
    out = torch.zeros((x.shape[0], n_out.value), dtype=torch.float32, device=x.device)
    with torch.cuda.device(device):
        _check(lib.pt_bsdf_batch(device, fn, math_mode, _ptr(x), x.shape[0], _ptr(out), _stream()))
    return out


PT_COMM_ID_BYTES = 128


def comm_unique_id():
    """The id rank 0 makes and hands to every rank (ncclGetUniqueId)."""
    buf = C.create_string_buffer(PT_COMM_ID_BYTES)
    _check(lib.pt_comm_unique_id(buf))
    return buf.raw


def assemble_rows(gathered, full, height, width, world):
    """gathered[r][k] -> image row k*world + r, on the current stream."""
    assert gathered.is_contiguous() and full.is_contiguous()
    assert gathered.numel() >= world * lib.pt_tile_rows(height, world) * width * 3 and full.numel() >= height * width * 3
    with _torch().cuda.device(full.device):
        _check(lib.pt_assemble_rows(_ptr(gathered), _ptr(full), height, width, world, _stream()))
    return full


class Comm:
    """One rank's RCCL communicator (pt_comm_init_rank / pt_comm_free) and the gather of the ranks' row tiles."""

    def __init__(self, unique_id, rank, world, device=0):
        assert len(unique_id) == PT_COMM_ID_BYTES
        self.rank, self.world, self.device = rank, world, device
        h = C.c_void_p(0)
        _check(lib.pt_comm_init_rank(unique_id, rank, world, device, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            lib.pt_comm_free(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def gather(self, tile, gathered, full, height, width):
        """pt_gather_framebuffer on the current stream: every rank gets the assembled [height, width, 3] frame."""
        rows = lib.pt_tile_rows(height, self.world)
        assert tile.is_contiguous() and gathered.is_contiguous() and full.is_contiguous()
        assert tile.numel() >= rows * width * 3 and gathered.numel() >= self.world * rows * width * 3
        assert full.numel() >= height * width * 3
        with _torch().cuda.device(self.device):
            _check(lib.pt_gather_framebuffer(self.h, _ptr(tile), _ptr(gathered), _ptr(full), height, width, _stream()))
        return full

    def gather_noalloc(self, tile, full, height, width):
        """pt_gather_framebuffer_noalloc: the same gather staged in the library's static device array (at most
        32 MiB of gathered tiles); allocates nothing.  Calls that overlap in time on one GPU race."""
        rows = lib.pt_tile_rows(height, self.world)
        assert tile.is_contiguous() and full.is_contiguous()
        assert tile.numel() >= rows * width * 3 and full.numel() >= height * width * 3
        with _torch().cuda.device(self.device):
            _check(lib.pt_gather_framebuffer_noalloc(self.h, _ptr(tile), _ptr(full), height, width, _stream()))
        return full
