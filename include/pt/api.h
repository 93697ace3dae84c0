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
// compatibility and cost a kernel launch each; the *_batch forms take arrays.
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
#include <cstring>
#include <limits>
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

// Device array that frees itself.
template <typename T> struct DeviceBuffer {
    T* p = nullptr;
    size_t n = 0;
    explicit DeviceBuffer(size_t count) : n(count) { if (n) checkHipErrors(hipMalloc((void**)&p, n * sizeof(T))); }
    DeviceBuffer(const T* host, size_t count) : DeviceBuffer(count) { if (n) checkHipErrors(hipMemcpy(p, host, n * sizeof(T), hipMemcpyHostToDevice)); }
    ~DeviceBuffer() { if (p) (void)hipFree(p); }
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    void to_host(T* host) const { if (n) checkHipErrors(hipMemcpy(host, p, n * sizeof(T), hipMemcpyDeviceToHost)); }
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

inline pt::DeviceScene::DeviceScene(Shape* const* scene, int n_objects, int device) {
    std::vector<pt_prim> prims((size_t)n_objects);
    for (int k = 0; k < n_objects; k++) {
        prims[k] = scene[k]->description();
        // shapes own their Material by value (shapes.h:41); identical ones share a table slot
        const pt_material m = pt::to_abi(scene[k]->material);
        size_t slot = 0;
        while (slot < mats.size() && std::memcmp(&mats[slot], &m, sizeof(m)) != 0) slot++;
        if (slot == mats.size()) mats.push_back(m);
        prims[k].material = (int32_t)slot;
    }
    checkHipErrors(pt_scene_upload(prims.data(), n_objects, mats.data(), (int)mats.size(), device, &h));
}

inline void Shape::getIntersections_batch(const ray* rays, int n, RayPath* out, int filter) const {
    if (n <= 0) return;
    Shape* self = const_cast<Shape*>(this);
    pt::DeviceScene sc(&self, 1);
    std::vector<float> o((size_t)n * 3), d((size_t)n * 3);
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) { o[3 * i + k] = rays[i].origin()[k]; d[3 * i + k] = rays[i].direction()[k]; }
    pt::DeviceBuffer<float> d_o(o.data(), o.size()), d_d(d.data(), d.size());
    pt::DeviceBuffer<int32_t> d_n(n), d_h1(n), d_h2(n);
    pt::DeviceBuffer<float> d_t1(n), d_t2(n), d_p1((size_t)n * 3), d_p2((size_t)n * 3), d_n1((size_t)n * 3), d_n2((size_t)n * 3);
    checkHipErrors(pt_raypath_batch(sc.handle(), 0, d_o.p, d_d.p, n, filter, d_n.p, d_h1.p, d_t1.p, d_p1.p, d_n1.p,
                                    d_h2.p, d_t2.p, d_p2.p, d_n2.p, nullptr));
    checkHipErrors(hipDeviceSynchronize());
    std::vector<int32_t> hn(n), h1(n), h2(n);
    std::vector<float> t1(n), t2(n), p1((size_t)n * 3), p2((size_t)n * 3), n1((size_t)n * 3), n2((size_t)n * 3);
    d_n.to_host(hn.data()); d_h1.to_host(h1.data()); d_h2.to_host(h2.data());
    d_t1.to_host(t1.data()); d_t2.to_host(t2.data());
    d_p1.to_host(p1.data()); d_p2.to_host(p2.data()); d_n1.to_host(n1.data()); d_n2.to_host(n2.data());
    for (int i = 0; i < n; i++) {
        RayPath& rp = out[i];
        rp = RayPath();
        rp.n_intersections = hn[i];
        rp.first.intersects = h1[i] != 0; rp.first.distance = t1[i]; rp.first.material = material;
        rp.first.position = vec3(p1[3 * i], p1[3 * i + 1], p1[3 * i + 2]); rp.first.normal = vec3(n1[3 * i], n1[3 * i + 1], n1[3 * i + 2]);
        rp.second.intersects = h2[i] != 0; rp.second.distance = t2[i]; rp.second.material = material;
        rp.second.position = vec3(p2[3 * i], p2[3 * i + 1], p2[3 * i + 2]); rp.second.normal = vec3(n2[3 * i], n2[3 * i + 1], n2[3 * i + 2]);
    }
}

inline RayPath Shape::query(const ray& r, int filter) const {
    RayPath rp;
    getIntersections_batch(&r, 1, &rp, filter);
    return rp;
}

// shapes.h:79-100 for n rays against an uploaded scene.
inline void getNearestIntersection_batch(const ray* rays, int n, const pt::DeviceScene& sc, Shape* const* scene, IntersectionPoint* out) {
    if (n <= 0) return;
    std::vector<float> o((size_t)n * 3), d((size_t)n * 3);
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) { o[3 * i + k] = rays[i].origin()[k]; d[3 * i + k] = rays[i].direction()[k]; }
    pt::DeviceBuffer<float> d_o(o.data(), o.size()), d_d(d.data(), d.size()), d_t(n), d_p((size_t)n * 3), d_nr((size_t)n * 3);
    pt::DeviceBuffer<int32_t> d_prim(n), d_in(n);
    checkHipErrors(pt_nearest_batch(sc.handle(), d_o.p, d_d.p, n, d_prim.p, d_t.p, d_p.p, d_nr.p, d_in.p, nullptr));
    checkHipErrors(hipDeviceSynchronize());
    std::vector<int32_t> prim(n), in(n);
    std::vector<float> t(n), p((size_t)n * 3), nr((size_t)n * 3);
    d_prim.to_host(prim.data()); d_in.to_host(in.data()); d_t.to_host(t.data()); d_p.to_host(p.data()); d_nr.to_host(nr.data());
    for (int i = 0; i < n; i++) {
        IntersectionPoint& ip = out[i];
        ip = IntersectionPoint();
        ip.intersects = prim[i] >= 0;
        if (!ip.intersects) continue;
        ip.distance = t[i];
        ip.material = scene[prim[i]]->material;
        ip.position = vec3(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
        ip.normal = vec3(nr[3 * i], nr[3 * i + 1], nr[3 * i + 2]);
        ip.inside = in[i] != 0;
    }
}

// shapes.h:79 — one ray, scene given as the reference's Shape** array.
inline IntersectionPoint getNearestIntersection(const ray& r, Shape** scene, int n_objects) {
    pt::DeviceScene sc(scene, n_objects);
    IntersectionPoint ip;
    getNearestIntersection_batch(&r, 1, sc, scene, &ip);
    return ip;
}

// bsdf.h:73-168 for n rays; ray i draws from stream s[i] starting after the two
// camera-jitter slots.  n_bounces defaults to the reference's literal (bsdf.h:75).
inline void tracePath_batch(const pt::DeviceScene& sc, const ray* rays, const pt_rng* s, int n, vec3* out,
                            int n_bounces = 5, int math_mode = PT_MATH_PARITY, int bsdf_model = PT_BSDF_REFERENCE) {
    if (n <= 0) return;
    std::vector<float> o((size_t)n * 3), d((size_t)n * 3);
    std::vector<uint32_t> px(n), sm(n);
    for (int i = 0; i < n; i++) {
        for (int k = 0; k < 3; k++) { o[3 * i + k] = rays[i].origin()[k]; d[3 * i + k] = rays[i].direction()[k]; }
        px[i] = s[i].pixel; sm[i] = s[i].sample;
    }
    pt::DeviceBuffer<float> d_o(o.data(), o.size()), d_d(d.data(), d.size()), d_rgb((size_t)n * 3);
    pt::DeviceBuffer<uint32_t> d_px(px.data(), px.size()), d_sm(sm.data(), sm.size());
    if (bsdf_model == PT_BSDF_CLASSES)
        checkHipErrors(pt_trace_paths_alt(sc.handle(), d_o.p, d_d.p, d_px.p, d_sm.p, n, s[0].seed, n_bounces, math_mode, d_rgb.p, nullptr));
    else
        checkHipErrors(pt_trace_paths(sc.handle(), d_o.p, d_d.p, d_px.p, d_sm.p, n, s[0].seed, n_bounces, math_mode, d_rgb.p, nullptr));
    checkHipErrors(hipDeviceSynchronize());
    std::vector<float> rgb((size_t)n * 3);
    d_rgb.to_host(rgb.data());
    for (int i = 0; i < n; i++) out[i] = vec3(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]);
}

// bsdf.h:73 — the reference signature.  The reference mutates `r` as the path
// advances and returns the radiance; only the radiance is observable to callers
// (main.cu:65), so `r` is left as passed.
// A host that includes bsdf_alternative.h instead of bsdf.h gets that file's tracePath (:153-219) under the same
// name; here that choice is the macro PT_USE_BSDF_ALTERNATIVE, defined before this header is included.
inline vec3 tracePath(Shape** scene, int n_objects, ray& r, pt_rng* s, int n_bounces = 5) {
    pt::DeviceScene sc(scene, n_objects);
    vec3 out;
#ifdef PT_USE_BSDF_ALTERNATIVE
    tracePath_batch(sc, &r, s, 1, &out, n_bounces, PT_MATH_PARITY, PT_BSDF_CLASSES);
#else
    tracePath_batch(sc, &r, s, 1, &out, n_bounces);
#endif
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
    checkHipErrors(pt_render(scene.handle(), &c, reinterpret_cast<float*>(fb), stream));
}

// The reference's launch shape: render(fb, config, scene, n_objects).
inline void render(vec3* fb, config_t config, Shape** scene, int n_objects, const render_options& opt = render_options()) {
    pt::DeviceScene sc(scene, n_objects, opt.device);
    render(fb, config, sc, opt);
    checkHipErrors(hipDeviceSynchronize());
}
