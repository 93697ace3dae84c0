// api.h — the reference's source-level API for its hot path, hosted on the GPU.
//
// Same names, members and argument meaning as the reference headers so that host
// code written against them keeps compiling:
//   ray                      (ray.h:4-14)
//   Material                 (materials.h:7-39)
//   IntersectionPoint, RayPath, Shape, Plane, Cube, Cylinder, Sphere,
//   getNearestIntersection   (shapes.h:11-361)
//   tracePath                (bsdf.h:73-168) — curandState* becomes pt_rng*
//   config_t, render         (main.cu:22-69)
// None of these compute on the CPU.  A Shape is a description (transform,
// bounds, material); getRayPath / getIntersections / getNearestIntersection /
// tracePath / render marshal their arguments into the C ABI of pt_hip.h and run
// the gfx950 kernels of libpt_hip.so.  The one-ray forms exist for drop-in
// compatibility and cost one launch and two copies each (tens of microseconds):
// the device copy of the scene they name is kept from call to call (looked up
// by the shapes' contents, so a changed shape or material uploads anew) and
// all staging goes through one buffer per host thread that is allocated once.
// The *_batch forms take arrays.
// `ray&` parameters are `const ray&` (the reference binds temporaries to
// non-const references, shapes.h:61, which g++ and clang++ reject).
//
// Errors follow the reference's check_cuda (main.cu:11-19): the message
// "HIP error = <n> at <file>:<line> '<expr>'" goes to stderr, the device is
// reset and the process exits with status 99.
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../pt_hip.h"
#include "linalg.h"

#define checkHipErrors(val) pt_check_status((int)(val), #val, __FILE__, __LINE__)
inline void pt_check_status(int result, char const* const func, const char* const file, int const line) {
    if (result) {
        const char* detail = pt_last_error();
        if (detail && detail[0]) std::fprintf(stderr, "%s\n", detail);
        else std::fprintf(stderr, "HIP error = %u at %s:%d '%s' \n", (unsigned)result, file, line, func);
        (void)hipDeviceReset();
        std::exit(99);
    }
}

class ray {
    vec3 o, d;
  public:
    ray() {}
    ray(vec3 origin, vec3 direction) : o(origin), d(direction) {}
    vec3 origin() const { return o; }
    vec3 direction() const { return d; }
    vec3 at(float t) const { return o + d * t; }
};

class Shape;

// Ten public fields with the reference's defaults.
class Material {
  public:
    Material() {}
    vec3 albedo = vec3(0.0f, 0.0f, 0.0f);           // diffuse colour
    vec3 emissive = vec3(0.0f, 0.0f, 0.0f);         // emitted radiance
    float f0 = 0.0f;                                // reflectivity at normal incidence
    float specularRoughness = 0.0f;
    vec3 specularColor = vec3(0.0f, 0.0f, 0.0f);
    float IOR = 1.0f;
    float transparency = 0.0f;                      // refraction probability scale
    float refractionRoughness = 0.0f;
    vec3 refractionColor = vec3(0.0f, 0.0f, 0.0f);  // Beer absorption per unit length
    int bsdfType = 0;
};

struct IntersectionPoint {
    bool intersects = false;
    float distance = 0.0f;
    Material material;
    vec3 position;
    vec3 normal;
    bool inside = false;
};

struct RayPath {
    int n_intersections = 0;
    IntersectionPoint first;
    IntersectionPoint second;
};

namespace pt {

inline pt_material to_abi(const Material& m) {
    pt_material r;
    std::memset(&r, 0, sizeof(r));
    for (int k = 0; k < 3; k++) {
        r.albedo[k] = m.albedo[k]; r.emissive[k] = m.emissive[k];
        r.specularColor[k] = m.specularColor[k]; r.refractionColor[k] = m.refractionColor[k];
    }
    r.f0 = m.f0; r.specularRoughness = m.specularRoughness; r.IOR = m.IOR; r.transparency = m.transparency;
    r.refractionRoughness = m.refractionRoughness; r.bsdfType = m.bsdfType;
    return r;
}

// Staging memory of the calls below: one device buffer and one pinned host buffer per host thread AND device, grown
// to the largest request so far and kept (pt::release_cached() frees them, and so does the end of the thread).  A call
// packs its inputs at the front and reads its outputs behind them: one copy in, the launch, one copy out.  The device
// buffer lives on the device of the scene the call is for, whatever device is current.
struct Staging {
    int device = 0;
    char* dev = nullptr;
    char* host = nullptr;
    size_t cap = 0;
    Staging() = default;
    Staging(const Staging&) = delete;
    Staging& operator=(const Staging&) = delete;
    ~Staging() { release(); }
    void reserve(size_t bytes) {
        if (bytes <= cap) return;
        release();
        size_t want = 1 << 16;
        while (want < bytes) want <<= 1;
        int before = 0;
        checkHipErrors(hipGetDevice(&before));
        checkHipErrors(hipSetDevice(device));
        checkHipErrors(hipMalloc((void**)&dev, want));
        checkHipErrors(hipHostMalloc((void**)&host, want, hipHostMallocDefault));
        checkHipErrors(hipSetDevice(before));
        cap = want;
    }
    void release() {
        if (dev) (void)hipFree(dev);
        if (host) (void)hipHostFree(host);
        dev = nullptr; host = nullptr; cap = 0;
    }
};
struct StagingSet {
    std::vector<Staging*> per_device;
    ~StagingSet() { clear(); }
    Staging& of(int device) {
        if (device < 0) device = 0;
        if ((size_t)device >= per_device.size()) per_device.resize((size_t)device + 1, nullptr);
        if (!per_device[(size_t)device]) { per_device[(size_t)device] = new Staging(); per_device[(size_t)device]->device = device; }
        return *per_device[(size_t)device];
    }
    void clear() { for (Staging* s : per_device) delete s; per_device.clear(); }
};
inline StagingSet& staging_set() { static thread_local StagingSet s; return s; }
// This thread's staging buffers on the device of `sc`.
inline Staging& staging(const pt_scene* sc) { return staging_set().of(pt_scene_device(sc)); }

// Carves arrays out of the staging buffers, device and host side by side.
struct Carver {
    Staging& st;
    size_t at = 0;
    explicit Carver(Staging& s) : st(s) {}
    template <typename T> size_t take(size_t count) { const size_t off = at; at += (count * sizeof(T) + 15) / 16 * 16; return off; }
    template <typename T> T* dev(size_t off) const { return reinterpret_cast<T*>(st.dev + off); }
    template <typename T> T* host(size_t off) const { return reinterpret_cast<T*>(st.host + off); }
};

// A scene on one device: what constructScene + the Shape** array are in the
// reference (main.cu:213-220).
class DeviceScene {
    pt_scene* h = nullptr;
    std::vector<pt_material> mats;
  public:
    DeviceScene(Shape* const* scene, int n_objects, int device = 0);
    ~DeviceScene() { if (h) pt_scene_free(h); }
    DeviceScene(const DeviceScene&) = delete;
    DeviceScene& operator=(const DeviceScene&) = delete;
    pt_scene* handle() const { return h; }
    const pt_material& material(int i) const { return mats[i]; }
};

// The tables a Shape* array flattens to: what pt_scene_upload takes, and the key under which uploads are kept.
struct SceneTables {
    std::vector<pt_prim> prims;
    std::vector<pt_material> mats;
    bool operator==(const SceneTables& o) const {
        return prims.size() == o.prims.size() && mats.size() == o.mats.size() &&
               std::memcmp(prims.data(), o.prims.data(), prims.size() * sizeof(pt_prim)) == 0 &&
               std::memcmp(mats.data(), o.mats.data(), mats.size() * sizeof(pt_material)) == 0;
    }
};
SceneTables flatten(Shape* const* scene, int n_objects);

// Uploaded scenes of the one-ray calls, per host thread: found again by content (so mutating a shape or its material
// between calls simply misses), the least recently used of more than eight is freed.
struct SceneCache {
    struct Entry { SceneTables key; int device; pt_scene* h; unsigned long long used; };
    std::vector<Entry> entries;
    unsigned long long clock = 0;
    pt_scene* get(Shape* const* scene, int n_objects, int device = 0) {
        SceneTables key = flatten(scene, n_objects);
        for (Entry& e : entries)
            if (e.device == device && e.key == key) { e.used = ++clock; return e.h; }
        if (entries.size() >= 8) {
            size_t lru = 0;
            for (size_t k = 1; k < entries.size(); k++) if (entries[k].used < entries[lru].used) lru = k;
            pt_scene_free(entries[lru].h);
            entries.erase(entries.begin() + (long)lru);
        }
        pt_scene* h = nullptr;
        checkHipErrors(pt_scene_upload(key.prims.data(), n_objects, key.mats.data(), (int)key.mats.size(), device, &h));
        entries.push_back(Entry{std::move(key), device, h, ++clock});
        return h;
    }
    void clear() { for (Entry& e : entries) pt_scene_free(e.h); entries.clear(); }
    ~SceneCache() { clear(); }
};
inline SceneCache& scene_cache() { static thread_local SceneCache c; return c; }

// Frees this thread's staging buffers and cached scenes (they are otherwise kept until the process ends).
inline void release_cached() { staging_set().clear(); scene_cache().clear(); }

}  // namespace pt

// The counter RNG's handle: which (seed, pixel, sample) stream a path draws from.
// Stands in for curandState in tracePath's signature (bsdf.h:73, main.cu:58-60).
struct pt_rng {
    unsigned seed = 0;
    unsigned pixel = 0;
    unsigned sample = 0;
};

class Shape {
  protected:
    pt_prim desc;
    explicit Shape(int type, Affine3f transform) {
        std::memset(&desc, 0, sizeof(desc));
        desc.type = type;
        for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) desc.transform[i * 4 + j] = transform.m[i][j];
        const float big = std::numeric_limits<float>::max();
        desc.bounds[0] = -big; desc.bounds[1] = -big; desc.bounds[2] = big; desc.bounds[3] = big;
    }
    RayPath query(const ray& r, int filter) const;
  public:
    Material material;
    virtual ~Shape() {}
    const pt_prim& description() const { return desc; }
    // shapes.h:58 — the ray is taken as already in the shape's local frame.
    virtual RayPath getRayPath(const ray& r) const { return query(r, 0); }
    // shapes.h:60-75 — world ray, epsilon filter applied.
    RayPath getIntersections(const ray& r) const { return query(r, 1); }
    // Array forms: n rays in, n RayPaths out, one kernel launch.
    void getIntersections_batch(const ray* rays, int n, RayPath* out, int filter = 1) const;
};

class Plane : public Shape {
  public:
    Plane(Material material, Affine3f transform, float xMin = std::numeric_limits<float>::lowest(),
          float yMin = std::numeric_limits<float>::lowest(), float xMax = std::numeric_limits<float>::max(),
          float yMax = std::numeric_limits<float>::max())
        : Shape(PT_PLANE, transform) {
        this->material = material;
        desc.bounds[0] = xMin; desc.bounds[1] = yMin; desc.bounds[2] = xMax; desc.bounds[3] = yMax;
    }
};
class Cube : public Shape {
  public:
    Cube(Material material, Affine3f transform) : Shape(PT_CUBE, transform) { this->material = material; }
};
class Cylinder : public Shape {
  public:
    Cylinder(Material material, Affine3f transform) : Shape(PT_CYLINDER, transform) { this->material = material; }
};
class Sphere : public Shape {
  public:
    Sphere(Material material, Affine3f transform) : Shape(PT_SPHERE, transform) { this->material = material; }
};

inline pt::SceneTables pt::flatten(Shape* const* scene, int n_objects) {
    SceneTables t;
    t.prims.resize((size_t)n_objects);
    for (int k = 0; k < n_objects; k++) {
        t.prims[k] = scene[k]->description();
        // shapes own their Material by value (shapes.h:41); identical ones share a table slot
        const pt_material m = pt::to_abi(scene[k]->material);
        size_t slot = 0;
        while (slot < t.mats.size() && std::memcmp(&t.mats[slot], &m, sizeof(m)) != 0) slot++;
        if (slot == t.mats.size()) t.mats.push_back(m);
        t.prims[k].material = (int32_t)slot;
    }
    return t;
}

inline pt::DeviceScene::DeviceScene(Shape* const* scene, int n_objects, int device) {
    SceneTables t = flatten(scene, n_objects);
    mats = t.mats;
    checkHipErrors(pt_scene_upload(t.prims.data(), n_objects, mats.data(), (int)mats.size(), device, &h));
}

namespace pt {
inline void pack_rays(const ray* rays, int n, float* o, float* d) {
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) { o[3 * i + k] = rays[i].origin()[k]; d[3 * i + k] = rays[i].direction()[k]; }
}
inline vec3 vec3_at(const float* p, int i) { return vec3(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }
}  // namespace pt

inline void Shape::getIntersections_batch(const ray* rays, int n, RayPath* out, int filter) const {
    if (n <= 0) return;
    Shape* self = const_cast<Shape*>(this);
    pt_scene* sc = pt::scene_cache().get(&self, 1);
    const size_t N = (size_t)n;
    pt::Carver c(pt::staging(sc));
    const size_t o = c.take<float>(3 * N), d = c.take<float>(3 * N), in_end = c.at;
    const size_t cnt = c.take<int32_t>(N), h1 = c.take<int32_t>(N), h2 = c.take<int32_t>(N), t1 = c.take<float>(N), t2 = c.take<float>(N);
    const size_t p1 = c.take<float>(3 * N), p2 = c.take<float>(3 * N), n1 = c.take<float>(3 * N), n2 = c.take<float>(3 * N);
    c.st.reserve(c.at);
    pt::pack_rays(rays, n, c.host<float>(o), c.host<float>(d));
    checkHipErrors(hipMemcpy(c.st.dev, c.st.host, in_end, hipMemcpyHostToDevice));
    checkHipErrors(pt_raypath_batch(sc, 0, c.dev<float>(o), c.dev<float>(d), n, filter, c.dev<int32_t>(cnt), c.dev<int32_t>(h1), c.dev<float>(t1),
                                    c.dev<float>(p1), c.dev<float>(n1), c.dev<int32_t>(h2), c.dev<float>(t2), c.dev<float>(p2), c.dev<float>(n2), nullptr));
    checkHipErrors(hipMemcpy(c.st.host + in_end, c.st.dev + in_end, c.at - in_end, hipMemcpyDeviceToHost));   // waits for the kernel
    for (int i = 0; i < n; i++) {
        RayPath& rp = out[i];
        rp = RayPath();
        rp.n_intersections = c.host<int32_t>(cnt)[i];
        rp.first.intersects = c.host<int32_t>(h1)[i] != 0; rp.first.distance = c.host<float>(t1)[i]; rp.first.material = material;
        rp.first.position = pt::vec3_at(c.host<float>(p1), i); rp.first.normal = pt::vec3_at(c.host<float>(n1), i);
        rp.second.intersects = c.host<int32_t>(h2)[i] != 0; rp.second.distance = c.host<float>(t2)[i]; rp.second.material = material;
        rp.second.position = pt::vec3_at(c.host<float>(p2), i); rp.second.normal = pt::vec3_at(c.host<float>(n2), i);
    }
}

inline RayPath Shape::query(const ray& r, int filter) const {
    RayPath rp;
    getIntersections_batch(&r, 1, &rp, filter);
    return rp;
}

// shapes.h:79-100 for n rays against an uploaded scene.
inline void pt_nearest_rays(const ray* rays, int n, pt_scene* sc, Shape* const* scene, IntersectionPoint* out) {
    if (n <= 0) return;
    const size_t N = (size_t)n;
    pt::Carver c(pt::staging(sc));
    const size_t o = c.take<float>(3 * N), d = c.take<float>(3 * N), in_end = c.at;
    const size_t prim = c.take<int32_t>(N), in = c.take<int32_t>(N), t = c.take<float>(N), p = c.take<float>(3 * N), nr = c.take<float>(3 * N);
    c.st.reserve(c.at);
    pt::pack_rays(rays, n, c.host<float>(o), c.host<float>(d));
    checkHipErrors(hipMemcpy(c.st.dev, c.st.host, in_end, hipMemcpyHostToDevice));
    checkHipErrors(pt_nearest_batch(sc, c.dev<float>(o), c.dev<float>(d), n, c.dev<int32_t>(prim), c.dev<float>(t), c.dev<float>(p), c.dev<float>(nr),
                                    c.dev<int32_t>(in), nullptr));
    checkHipErrors(hipMemcpy(c.st.host + in_end, c.st.dev + in_end, c.at - in_end, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) {
        IntersectionPoint& ip = out[i];
        ip = IntersectionPoint();
        const int32_t k = c.host<int32_t>(prim)[i];
        ip.intersects = k >= 0;
        if (!ip.intersects) continue;
        ip.distance = c.host<float>(t)[i];
        ip.material = scene[k]->material;
        ip.position = pt::vec3_at(c.host<float>(p), i);
        ip.normal = pt::vec3_at(c.host<float>(nr), i);
        ip.inside = c.host<int32_t>(in)[i] != 0;
    }
}
inline void getNearestIntersection_batch(const ray* rays, int n, const pt::DeviceScene& sc, Shape* const* scene, IntersectionPoint* out) {
    pt_nearest_rays(rays, n, sc.handle(), scene, out);
}

// shapes.h:79 — one ray, scene given as the reference's Shape** array.
inline IntersectionPoint getNearestIntersection(const ray& r, Shape** scene, int n_objects) {
    IntersectionPoint ip;
    pt_nearest_rays(&r, 1, pt::scene_cache().get(scene, n_objects), scene, &ip);
    return ip;
}

// bsdf.h:73-168 for n rays; ray i draws from stream s[i] starting after the two
// camera-jitter slots.  n_bounces defaults to the reference's literal (bsdf.h:75).
// end (optional): the ray each call of the reference's tracePath leaves in its `ray& r` argument (bsdf.h:165).
inline void pt_trace_rays(pt_scene* sc, const ray* rays, const pt_rng* s, int n, vec3* out, ray* end, int n_bounces, int math_mode, int bsdf_model) {
    if (n <= 0) return;
    const size_t N = (size_t)n;
    pt::Carver c(pt::staging(sc));
    const size_t o = c.take<float>(3 * N), d = c.take<float>(3 * N), px = c.take<uint32_t>(N), sm = c.take<uint32_t>(N), in_end = c.at;
    const size_t rgb = c.take<float>(3 * N), eo = c.take<float>(3 * N), ed = c.take<float>(3 * N);
    c.st.reserve(c.at);
    pt::pack_rays(rays, n, c.host<float>(o), c.host<float>(d));
    for (int i = 0; i < n; i++) { c.host<uint32_t>(px)[i] = s[i].pixel; c.host<uint32_t>(sm)[i] = s[i].sample; }
    checkHipErrors(hipMemcpy(c.st.dev, c.st.host, in_end, hipMemcpyHostToDevice));
    checkHipErrors(pt_trace_paths_end(sc, c.dev<float>(o), c.dev<float>(d), c.dev<uint32_t>(px), c.dev<uint32_t>(sm), n, s[0].seed, n_bounces, math_mode,
                                      bsdf_model, c.dev<float>(rgb), c.dev<float>(eo), c.dev<float>(ed), nullptr));
    checkHipErrors(hipMemcpy(c.st.host + in_end, c.st.dev + in_end, c.at - in_end, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) {
        out[i] = pt::vec3_at(c.host<float>(rgb), i);
        if (end) end[i] = ray(pt::vec3_at(c.host<float>(eo), i), pt::vec3_at(c.host<float>(ed), i));
    }
}
inline void tracePath_batch(const pt::DeviceScene& sc, const ray* rays, const pt_rng* s, int n, vec3* out,
                            int n_bounces = 5, int math_mode = PT_MATH_PARITY, int bsdf_model = PT_BSDF_REFERENCE, ray* end = nullptr) {
    pt_trace_rays(sc.handle(), rays, s, n, out, end, n_bounces, math_mode, bsdf_model);
}

// bsdf.h:73 — the reference signature: returns the radiance and, like the reference, leaves in `r` the last ray the
// path assigned (bsdf.h:165).
// A host that includes bsdf_alternative.h instead of bsdf.h gets that file's tracePath (:153-219) under the same
// name; here that choice is the macro PT_USE_BSDF_ALTERNATIVE, defined before this header is included.
inline vec3 tracePath(Shape** scene, int n_objects, ray& r, pt_rng* s, int n_bounces = 5) {
    vec3 out;
    ray end;
#ifdef PT_USE_BSDF_ALTERNATIVE
    pt_trace_rays(pt::scene_cache().get(scene, n_objects), &r, s, 1, &out, &end, n_bounces, PT_MATH_PARITY, PT_BSDF_CLASSES);
#else
    pt_trace_rays(pt::scene_cache().get(scene, n_objects), &r, s, 1, &out, &end, n_bounces, PT_MATH_PARITY, PT_BSDF_REFERENCE);
#endif
    r = end;
    return out;
}

// main.cu:22-28
typedef struct config_s {
    vec3 camera_pos;
    int max_x;
    int max_y;
    float focal_length;
    float view_port_height;
} config_t;

// The values the reference hard-codes around its render kernel: n_samples
// (main.cu:54), n_bounces (bsdf.h:75); plus seed, row tile and math mode.
struct render_options {
    int n_samples = 2000;
    int n_bounces = 5;
    unsigned seed = 0;
    int row_begin = 0;
    int row_end = -1;            // -1 = max_y
    int row_step = 1;            // rows row_begin, row_begin + row_step, ... are rendered
    int fb_compact = 0;          // 1: the k-th rendered row is written as row k of fb
    int math_mode = PT_MATH_FAST;
    int bsdf_model = PT_BSDF_REFERENCE;   // PT_BSDF_CLASSES: tracePath of bsdf_alternative.h
    int tile_log2 = 0;           // log2 of the pixels per wave64 (2..6), 0 = chosen from the launch size
    int device = 0;
    unsigned long long* d_counters = nullptr;   // device memory for PT_COUNTER_COUNT counters the render adds to (pt_render_counted), or null
    int spp_chunk = 0;           // > 0: the sample loop (main.cu:61-67) runs as launches of at most this many samples each,
                                 // their exact sums kept in device memory between them; same image bit for bit
};

inline pt_config pt_make_config(const config_t& config, const render_options& opt) {
    pt_config c;
    std::memset(&c, 0, sizeof(c));
    for (int k = 0; k < 3; k++) c.camera_pos[k] = config.camera_pos[k];
    c.max_x = config.max_x; c.max_y = config.max_y;
    c.focal_length = config.focal_length; c.view_port_height = config.view_port_height;
    c.n_samples = opt.n_samples; c.n_bounces = opt.n_bounces; c.seed = opt.seed;
    c.row_begin = opt.row_begin; c.row_end = opt.row_end < 0 ? config.max_y : opt.row_end;
    c.fb_layout = PT_FB_AOS; c.math_mode = opt.math_mode;
    c.row_step = opt.row_step; c.fb_compact = opt.fb_compact; c.bsdf_model = opt.bsdf_model; c.tile_log2 = opt.tile_log2;
    return c;
}

// render (main.cu:30-69, launched at main.cu:236): fb is DEVICE (or managed)
// memory of max_x*max_y vec3, as in the reference; stream-ordered, no sync.
inline void render(vec3* fb, config_t config, const pt::DeviceScene& scene, const render_options& opt = render_options(), hipStream_t stream = nullptr) {
    static_assert(sizeof(vec3) == 12, "vec3 must be three packed floats");
    const pt_config c = pt_make_config(config, opt);
    if (opt.spp_chunk <= 0 || opt.spp_chunk >= c.n_samples) {
        if (opt.d_counters) checkHipErrors(pt_render_counted(scene.handle(), &c, reinterpret_cast<float*>(fb), opt.d_counters, stream));
        else checkHipErrors(pt_render(scene.handle(), &c, reinterpret_cast<float*>(fb), stream));
        return;
    }
    const long long n = pt_sums_count(&c);
    if (n <= 0) return;
    unsigned long long* sums = nullptr;
    checkHipErrors(hipSetDevice(pt_scene_device(scene.handle())));
    checkHipErrors(hipMalloc((void**)&sums, (size_t)n * sizeof(unsigned long long)));
    checkHipErrors(hipMemsetAsync(sums, 0, (size_t)n * sizeof(unsigned long long), stream));
    for (int s0 = 0; s0 < c.n_samples; s0 += opt.spp_chunk)
        checkHipErrors(pt_render_samples(scene.handle(), &c, s0, std::min(opt.spp_chunk, c.n_samples - s0), sums, stream));
    checkHipErrors(pt_resolve(scene.handle(), &c, sums, reinterpret_cast<float*>(fb), stream));
    checkHipErrors(hipStreamSynchronize(stream));
    checkHipErrors(hipFree(sums));
}

// The sample loop with its state in a file (no counterpart in the reference, whose loop of main.cu:61-67 runs inside
// one launch and cannot be interrupted): samples are traced in pieces of opt.spp_chunk (all at once if 0), and after
// every piece the pixels' exact sums and the number of samples done are written to ck.path (to a temporary file that is
// then renamed, so a kill leaves the previous state intact).  A call that finds such a file for the same frame —
// pt_config, scene tag and buffer size all equal — goes on from it.  With ck.stop_after > 0 the call returns once at
// least that many samples are done.  Returns the samples done: opt.n_samples when fb holds the finished image
// (bit for bit the image of one launch, however the samples were cut); fewer when it stopped early; -1 with a message
// on stderr when the file belongs to another frame or cannot be written.
struct render_checkpoint {
    std::string path;
    unsigned long long scene_tag = 0;    // pt::scene_tag of the shapes: a state file of another scene is refused
    int stop_after = 0;
};
namespace pt {
inline unsigned long long scene_tag(Shape* const* scene, int n_objects) {
    const SceneTables t = flatten(scene, n_objects);
    unsigned long long h = 1469598103934665603ull;                       // FNV-1a over both tables
    auto eat = [&](const void* p, size_t n) { const unsigned char* b = static_cast<const unsigned char*>(p); for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } };
    eat(t.prims.data(), t.prims.size() * sizeof(pt_prim));
    eat(t.mats.data(), t.mats.size() * sizeof(pt_material));
    return h;
}
struct CheckpointHeader {
    char magic[8];
    pt_config config;
    unsigned long long scene_tag;
    long long count;                     // 64-bit sums that follow
    int done, pad;
};
}  // namespace pt
inline int render_resumable(vec3* fb, config_t config, const pt::DeviceScene& scene, const render_options& opt, const render_checkpoint& ck,
                            hipStream_t stream = nullptr) {
    const pt_config c = pt_make_config(config, opt);
    const long long n = pt_sums_count(&c);
    if (n <= 0 || ck.path.empty()) { std::fprintf(stderr, "render_resumable: no pixels to render, or no file named\n"); return -1; }
    pt::CheckpointHeader want;
    std::memset(&want, 0, sizeof(want));
    std::memcpy(want.magic, "PTSUMS1", 8);
    want.config = c; want.scene_tag = ck.scene_tag; want.count = n;
    std::vector<unsigned long long> host((size_t)n, 0ull);
    int done = 0;
    if (std::FILE* f = std::fopen(ck.path.c_str(), "rb")) {
        pt::CheckpointHeader got;
        const bool ok = std::fread(&got, sizeof(got), 1, f) == 1 && std::memcmp(got.magic, want.magic, 8) == 0 &&
                        std::memcmp(&got.config, &want.config, sizeof(pt_config)) == 0 && got.scene_tag == want.scene_tag && got.count == n &&
                        got.done >= 0 && got.done <= c.n_samples && std::fread(host.data(), sizeof(unsigned long long), (size_t)n, f) == (size_t)n;
        std::fclose(f);
        if (!ok) { std::fprintf(stderr, "render_resumable: %s is not the state of this frame (other size, samples, seed, rows, mode or scene), or is damaged\n", ck.path.c_str()); return -1; }
        done = got.done;
    }
    unsigned long long* sums = nullptr;
    checkHipErrors(hipSetDevice(pt_scene_device(scene.handle())));
    checkHipErrors(hipMalloc((void**)&sums, (size_t)n * sizeof(unsigned long long)));
    checkHipErrors(hipMemcpyAsync(sums, host.data(), (size_t)n * sizeof(unsigned long long), hipMemcpyHostToDevice, stream));
    const int chunk = opt.spp_chunk > 0 ? opt.spp_chunk : c.n_samples;
    const std::string tmp = ck.path + ".tmp";
    while (done < c.n_samples) {
        const int count = std::min(chunk, c.n_samples - done);
        checkHipErrors(pt_render_samples(scene.handle(), &c, done, count, sums, stream));
        checkHipErrors(hipMemcpyAsync(host.data(), sums, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
        checkHipErrors(hipStreamSynchronize(stream));
        done += count;
        want.done = done;
        std::FILE* f = std::fopen(tmp.c_str(), "wb");
        const bool ok = f && std::fwrite(&want, sizeof(want), 1, f) == 1 && std::fwrite(host.data(), sizeof(unsigned long long), (size_t)n, f) == (size_t)n;
        if (f && std::fclose(f) != 0) { f = nullptr; }
        if (!ok || !f || std::rename(tmp.c_str(), ck.path.c_str()) != 0) {
            std::fprintf(stderr, "render_resumable: cannot write %s\n", ck.path.c_str());
            checkHipErrors(hipFree(sums));
            return -1;
        }
        if (ck.stop_after > 0 && done >= ck.stop_after && done < c.n_samples) { checkHipErrors(hipFree(sums)); return done; }
    }
    checkHipErrors(pt_resolve(scene.handle(), &c, sums, reinterpret_cast<float*>(fb), stream));
    checkHipErrors(hipStreamSynchronize(stream));
    checkHipErrors(hipFree(sums));
    return done;
}

// The reference's launch shape: render(fb, config, scene, n_objects).
inline void render(vec3* fb, config_t config, Shape** scene, int n_objects, const render_options& opt = render_options()) {
    pt::DeviceScene sc(scene, n_objects, opt.device);
    render(fb, config, sc, opt);
    checkHipErrors(hipDeviceSynchronize());
}
