// CPU ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see pt_oracle.hpp).
// C entry points over the restatement so that tests and the bench's
// cpu_baseline leg can drive it through ctypes.  "parity unpinned" beyond the
// reference's own n_intersections known answers.
#include "pt_oracle.hpp"

#include <atomic>
#include <cstdio>
#include <thread>

namespace ptor {
#ifdef PT_ORACLE_COUNT
thread_local Stats* g_stats = nullptr;
#endif
}

using namespace ptor;

extern "C" {

// Flat scene description handed in by the caller (same field order as Material,
// materials.h:12-38, and the Shape constructor arguments, shapes.h:111,147,229,325).
struct oracle_material {
    float albedo[3];
    float emissive[3];
    float f0;
    float specularRoughness;
    float specularColor[3];
    float IOR;
    float transparency;
    float refractionRoughness;
    float refractionColor[3];
    int bsdfType;
};
struct oracle_prim {
    int type;            // 0 plane, 1 cube, 2 cylinder, 3 sphere
    int material;        // index into the material array
    float transform[12]; // rows of [L | t]
    float bounds[4];     // plane only: xMin, yMin, xMax, yMax
};
struct oracle_config {
    float camera_pos[3];
    int max_x, max_y;
    float focal_length, view_port_height;   // config_t, main.cu:22-28
    int n_samples;                          // main.cu:54
    int n_bounces;                          // bsdf.h:75
    unsigned seed;
    int row_begin, row_end;                 // rows [row_begin,row_end) are rendered
    int bsdf_model;                         // 0 = tracePath of bsdf.h, 1 = tracePath of bsdf_alternative.h
};
struct oracle_stats {
    uint64_t samples, segments, hits, tests[4], sphere_roots, cyl_roots, emissive, beer, fresnel, fresnel_dense, mode[4];
    double flops;
};

}  // extern "C"

namespace {

struct Scene {
    std::vector<std::unique_ptr<Shape>> owned;
    std::vector<Shape*> ptrs;
};

Material to_material(const oracle_material& m, int id) {
    Material r;
    r.albedo = vec3(m.albedo[0], m.albedo[1], m.albedo[2]);
    r.emissive = vec3(m.emissive[0], m.emissive[1], m.emissive[2]);
    r.f0 = m.f0;
    r.specularRoughness = m.specularRoughness;
    r.specularColor = vec3(m.specularColor[0], m.specularColor[1], m.specularColor[2]);
    r.IOR = m.IOR;
    r.transparency = m.transparency;
    r.refractionRoughness = m.refractionRoughness;
    r.refractionColor = vec3(m.refractionColor[0], m.refractionColor[1], m.refractionColor[2]);
    r.bsdfType = m.bsdfType;
    r.id = id;
    return r;
}

void from_material(const Material& m, oracle_material* o) {
    for (int k = 0; k < 3; k++) {
        o->albedo[k] = m.albedo[k]; o->emissive[k] = m.emissive[k];
        o->specularColor[k] = m.specularColor[k]; o->refractionColor[k] = m.refractionColor[k];
    }
    o->f0 = m.f0; o->specularRoughness = m.specularRoughness; o->IOR = m.IOR;
    o->transparency = m.transparency; o->refractionRoughness = m.refractionRoughness; o->bsdfType = m.bsdfType;
}

Affine3f to_affine(const float* t) {
    Affine3f a;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) a.m[i][j] = t[i * 4 + j];
    return a;
}

Shape* make_shape(const oracle_prim& p, const oracle_material* mats, int n_mats) {
    Material m;
    if (p.material >= 0 && p.material < n_mats) m = to_material(mats[p.material], p.material);
    Affine3f t = to_affine(p.transform);
    switch (p.type) {
        case PT_PLANE: return new Plane(m, t, p.bounds[0], p.bounds[1], p.bounds[2], p.bounds[3]);
        case PT_CUBE: return new Cube(m, t);
        case PT_CYLINDER: return new Cylinder(m, t);
        case PT_SPHERE: return new Sphere(m, t);
    }
    return nullptr;
}

void fill_stats(const Stats& s, oracle_stats* o) {
    o->samples = s.samples; o->segments = s.segments; o->hits = s.hits;
    for (int i = 0; i < 4; i++) { o->tests[i] = s.tests[i]; o->mode[i] = s.mode[i]; }
    o->sphere_roots = s.sphere_roots; o->cyl_roots = s.cyl_roots; o->emissive = s.emissive; o->beer = s.beer;
    o->fresnel = s.fresnel; o->fresnel_dense = s.fresnel_dense; o->flops = s.flops();
}

}  // namespace

extern "C" {

int oracle_counts_enabled() {
#ifdef PT_ORACLE_COUNT
    return 1;
#else
    return 0;
#endif
}

void* oracle_scene_create(const oracle_prim* prims, int n, const oracle_material* mats, int n_mats) {
    Scene* s = new Scene();
    for (int k = 0; k < n; k++) {
        Shape* sh = make_shape(prims[k], mats, n_mats);
        if (!sh) { delete s; return nullptr; }
        s->owned.emplace_back(sh);
        s->ptrs.push_back(sh);
    }
    return s;
}

void oracle_scene_free(void* scene) { delete static_cast<Scene*>(scene); }

// The reference's default scene, restated as data (main.cu:73-195).  Fills up
// to 11 primitives / 8 materials; returns the primitive count.
int oracle_default_scene(oracle_prim* prims, oracle_material* mats, int* n_mats) {
    const float HALF_PI = (float)(PI / 2), NEG_HALF_PI = (float)(-PI / 2), FPI = (float)PI;
    Affine3f floor = Affine3f::Identity();                         // main.cu:77-79
    floor.setRotation(HALF_PI, vec3(1, 0, 0)); floor.setTranslation(0.0f, -1.0f, 0.0f);
    Affine3f back = Affine3f::Identity();                          // :82-83
    back.setTranslation(0.0f, 0.0f, -5.0f);
    Affine3f left = Affine3f::Identity();                          // :86-88
    left.setRotation(HALF_PI, vec3(0, 1, 0)); left.setTranslation(-1.0f, 0.0f, -5.0f);
    Affine3f right = Affine3f::Identity();                         // :91-93
    right.setRotation(NEG_HALF_PI, vec3(0, 1, 0)); right.setTranslation(1.0f, 0.0f, -5.0f);
    Affine3f ceiling = Affine3f::Identity();                       // :96-98
    ceiling.setRotation(NEG_HALF_PI, vec3(1, 0, 0)); ceiling.setTranslation(0.0f, 1.0f, 0.0f);
    Affine3f light = Affine3f::Identity();                         // :101-103
    light.setRotation(NEG_HALF_PI, vec3(1, 0, 0)); light.setTranslation(0.0f, 0.97f, -3.25f);
    Affine3f tplane6 = Affine3f::Identity();                       // :106-108
    tplane6.setRotation(FPI, vec3(0, 1, 0)); tplane6.setTranslation(0.0f, 0.0f, 7.0f);
    Affine3f t5 = Affine3f::Identity(); t5.setTranslation(0.0f, -0.5f, -3.0f); t5.scale(0.4f);    // :116-119
    Affine3f t6 = Affine3f::Identity(); t6.setTranslation(0.6f, -0.5f, -3.0f); t6.scale(0.4f);    // :121-124
    Affine3f t7 = Affine3f::Identity(); t7.setTranslation(-0.6f, -0.5f, -3.0f); t7.scale(0.4f);   // :126-129
    Affine3f t8 = Affine3f::Identity(); t8.setTranslation(0.0f, -0.5f, -4.2f); t8.scale(vec3(2.5f, 0.3f, 0.5f));  // :131-134

    Material light_material; light_material.emissive = vec3(1.0f, 0.9f, 0.7f) * 20.0f;   // :138-139
    Material base; base.albedo = vec3(0.4f, 0.4f, 0.4f);                                 // :141-142
    Material green; green.albedo = vec3(0.0f, 1.0f, 0.0f);                               // :144-145
    Material red; red.albedo = vec3(1.0f, 0.0f, 0.0f);                                   // :147-148
    Material glass;                                                                      // :150-157
    glass.f0 = 0.02f; glass.specularRoughness = 0; glass.specularColor = vec3(1.0f, 1.0f, 1.0f) * 0.8f;
    glass.IOR = 1.5f; glass.transparency = 1.0f; glass.refractionRoughness = 0.0f; glass.bsdfType = 1;
    Material diffuse_pure; diffuse_pure.albedo = vec3(0.6f, 0.6f, 0.6f);                 // :172-173
    Material glass1 = glass, glass2 = glass, glass3 = glass;                             // :175-183
    glass2.refractionColor = vec3(0.9f, 0.0f, 0.0f);
    glass3.refractionColor = vec3(2.0f, 0.0f, 0.0f);

    // material ids: 0 base, 1 red, 2 green, 3 light, 4 glass1, 5 glass2, 6 glass3, 7 diffuse_pure
    const Material* ms[8] = {&base, &red, &green, &light_material, &glass1, &glass2, &glass3, &diffuse_pure};
    for (int i = 0; i < 8; i++) from_material(*ms[i], &mats[i]);
    *n_mats = 8;

    struct Row { int type; int mat; const Affine3f* t; float b; };
    const Row rows[11] = {                                          // main.cu:185-195
        {PT_PLANE, 0, &floor, 15.0f}, {PT_PLANE, 1, &left, 15.0f}, {PT_PLANE, 2, &right, 15.0f},
        {PT_PLANE, 0, &back, 15.0f}, {PT_PLANE, 0, &ceiling, 15.0f}, {PT_PLANE, 3, &light, 0.5f},
        {PT_PLANE, 0, &tplane6, 15.0f}, {PT_SPHERE, 4, &t7, 0.0f}, {PT_SPHERE, 5, &t5, 0.0f},
        {PT_SPHERE, 6, &t6, 0.0f}, {PT_CUBE, 7, &t8, 0.0f}};
    for (int k = 0; k < 11; k++) {
        prims[k].type = rows[k].type;
        prims[k].material = rows[k].mat;
        for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) prims[k].transform[i * 4 + j] = rows[k].t->m[i][j];
        prims[k].bounds[0] = -rows[k].b; prims[k].bounds[1] = -rows[k].b;
        prims[k].bounds[2] = rows[k].b; prims[k].bounds[3] = rows[k].b;
    }
    return 11;
}

float oracle_focal_length(int nx, int ny, float v_height, float fov_deg) { return host_focal_length(nx, ny, v_height, fov_deg); }

// Shape::getIntersections (filter != 0, world ray) or Shape::getRayPath
// (filter == 0, the ray is taken as already local) for one primitive and a
// batch of rays.  Outputs are SoA: n_isect[n], and for first/second
// hit[n], t[n], pos[3n], nrm[3n].
void oracle_raypath_batch(const oracle_prim* prim, const float* o, const float* d, int n_rays, int filter,
                          int* n_isect, int* hit1, float* t1, float* pos1, float* nrm1,
                          int* hit2, float* t2, float* pos2, float* nrm2) {
    oracle_material dummy;
    std::memset(&dummy, 0, sizeof(dummy));
    oracle_prim p = *prim;
    p.material = -1;
    std::unique_ptr<Shape> sh(make_shape(p, &dummy, 0));
    for (int i = 0; i < n_rays; i++) {
        ray r(vec3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), vec3(d[3 * i], d[3 * i + 1], d[3 * i + 2]));
        RayPath rp = filter ? sh->getIntersections(r) : sh->getRayPath(r);
        n_isect[i] = rp.n_intersections;
        const IntersectionPoint* pts[2] = {&rp.first, &rp.second};
        int* hits[2] = {hit1, hit2}; float* ts[2] = {t1, t2}; float* ps[2] = {pos1, pos2}; float* ns[2] = {nrm1, nrm2};
        for (int s = 0; s < 2; s++) {
            bool valid = s < rp.n_intersections && pts[s]->intersects;
            hits[s][i] = valid ? 1 : 0;
            ts[s][i] = valid ? pts[s]->distance : 0.0f;
            for (int k = 0; k < 3; k++) {
                ps[s][3 * i + k] = valid ? pts[s]->position[k] : 0.0f;
                ns[s][3 * i + k] = valid ? pts[s]->normal[k] : 0.0f;
            }
        }
    }
}

// getNearestIntersection over a whole scene for a batch of rays.
void oracle_nearest_batch(void* scene, const float* o, const float* d, int n_rays,
                          int* prim, float* t, float* pos, float* nrm, int* inside) {
    Scene* sc = static_cast<Scene*>(scene);
    for (int i = 0; i < n_rays; i++) {
        ray r(vec3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), vec3(d[3 * i], d[3 * i + 1], d[3 * i + 2]));
        IntersectionPoint p = getNearestIntersection(r, sc->ptrs.data(), (int)sc->ptrs.size());
        prim[i] = p.intersects ? p.prim : -1;
        t[i] = p.intersects ? p.distance : 0.0f;
        inside[i] = p.intersects && p.inside ? 1 : 0;
        for (int k = 0; k < 3; k++) {
            pos[3 * i + k] = p.intersects ? p.position[k] : 0.0f;
            nrm[3 * i + k] = p.intersects ? p.normal[k] : 0.0f;
        }
    }
}

// tracePath for a batch of explicit rays; ray i uses the stream
// (seed, pixel[i], sample[i]) starting at draw slot 2 (slots 0,1 belong to the
// camera jitter).
void oracle_trace_paths(void* scene, const float* o, const float* d, const unsigned* pixel, const unsigned* sample,
                        int n_rays, unsigned seed, int n_bounces, float* rgb) {
    Scene* sc = static_cast<Scene*>(scene);
    for (int i = 0; i < n_rays; i++) {
        ray r(vec3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), vec3(d[3 * i], d[3 * i + 1], d[3 * i + 2]));
        pt_rng s = rng_init(seed, pixel[i], sample[i]);
        s.slot = 2;
        vec3 c = tracePath(sc->ptrs.data(), (int)sc->ptrs.size(), r, &s, n_bounces);
        for (int k = 0; k < 3; k++) rgb[3 * i + k] = c[k];
    }
}

// The same for the tracePath of bsdf_alternative.h:153-219.
void oracle_trace_paths_alt(void* scene, const float* o, const float* d, const unsigned* pixel, const unsigned* sample,
                            int n_rays, unsigned seed, int n_bounces, float* rgb) {
    Scene* sc = static_cast<Scene*>(scene);
    for (int i = 0; i < n_rays; i++) {
        ray r(vec3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), vec3(d[3 * i], d[3 * i + 1], d[3 * i + 2]));
        pt_rng s = rng_init(seed, pixel[i], sample[i]);
        s.slot = 2;
        vec3 c = tracePathAlternative(sc->ptrs.data(), (int)sc->ptrs.size(), r, &s, n_bounces);
        for (int k = 0; k < 3; k++) rgb[3 * i + k] = c[k];
    }
}

// tracePath with its side effect on `r` (bsdf.h:165, bsdf_alternative.h:215): radiance plus the ray left in r.
void oracle_trace_paths_end(void* scene, const float* o, const float* d, const unsigned* pixel, const unsigned* sample,
                            int n_rays, unsigned seed, int n_bounces, int bsdf_model, float* rgb, float* end_o, float* end_d) {
    Scene* sc = static_cast<Scene*>(scene);
    for (int i = 0; i < n_rays; i++) {
        ray r(vec3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), vec3(d[3 * i], d[3 * i + 1], d[3 * i + 2]));
        pt_rng s = rng_init(seed, pixel[i], sample[i]);
        s.slot = 2;
        vec3 c = bsdf_model == 1 ? tracePathAlternative(sc->ptrs.data(), (int)sc->ptrs.size(), r, &s, n_bounces)
                                 : tracePath(sc->ptrs.data(), (int)sc->ptrs.size(), r, &s, n_bounces);
        for (int k = 0; k < 3; k++) { rgb[3 * i + k] = c[k]; end_o[3 * i + k] = r.origin()[k]; end_d[3 * i + k] = r.direction()[k]; }
    }
}

// Raw RNG access: bits and uniforms of draw slots [0, n_slots) of one stream.
// The helper functions of bsdf.h / vec_math.h on arrays; fn and the array shapes as pt_bsdf_batch of the product's ABI:
// 0 sampleHemisphere (2 -> 3), 1 FresnelSchlick (3 -> 1), 2 getFresnelRatio (10 -> 1), 3 reflect (6 -> 3),
// 4 refract (7 -> 3), 5 lerp<vec3> (7 -> 3), 6 vec3::normalized (3 -> 3).
int oracle_bsdf_batch(int fn, const float* in, int n, float* out) {
    for (int i = 0; i < n; i++) {
        vec3 r(0, 0, 0);
        switch (fn) {
            case 0: r = sampleHemisphere(in[2 * i], in[2 * i + 1]); break;
            case 1: out[i] = FresnelSchlick(in[3 * i], in[3 * i + 1], in[3 * i + 2]); continue;
            case 2: { const float* a = in + 10 * i; out[i] = getFresnelRatio(vec3(a[0], a[1], a[2]), vec3(a[3], a[4], a[5]), a[6], a[7], a[8], a[9]); continue; }
            case 3: { const float* a = in + 6 * i; r = reflect(vec3(a[0], a[1], a[2]), vec3(a[3], a[4], a[5])); break; }
            case 4: { const float* a = in + 7 * i; r = refract(vec3(a[0], a[1], a[2]), vec3(a[3], a[4], a[5]), a[6]); break; }
            case 5: { const float* a = in + 7 * i; r = lerp(vec3(a[0], a[1], a[2]), vec3(a[3], a[4], a[5]), a[6]); break; }
            case 6: r = vec3(in[3 * i], in[3 * i + 1], in[3 * i + 2]).normalized(); break;
            default: return -1;
        }
        out[3 * i] = r.x(); out[3 * i + 1] = r.y(); out[3 * i + 2] = r.z();
    }
    return 0;
}

void oracle_rng(unsigned seed, unsigned pixel, unsigned sample, int n_slots, unsigned* bits, float* uniforms) {
    pt_rng s = rng_init(seed, pixel, sample);
    for (int k = 0; k < n_slots; k++) {
        bits[k] = rng_bits(s.key, (uint32_t)k);
        uniforms[k] = rng_slot_uniform(s.key, (uint32_t)k);
    }
}

// Camera rays exactly as render builds them (main.cu:40-64): for pixel (i,j)
// and sample s, writes origin and direction.
void oracle_camera_rays(const oracle_config* cfg, const int* pi, const int* pj, const unsigned* sample, int n,
                        float* o, float* d) {
    int max_x = cfg->max_x, max_y = cfg->max_y;
    vec3 camera_pos(cfg->camera_pos[0], cfg->camera_pos[1], cfg->camera_pos[2]);
    float viewport_height = cfg->view_port_height;
    float viewport_width = viewport_height * (static_cast<float>(max_x) / max_y);
    vec3 viewport_u = vec3(viewport_width, 0, 0), viewport_v = vec3(0, -viewport_height, 0);
    vec3 pixel_delta_u = viewport_u / (float)max_x, pixel_delta_v = viewport_v / (float)max_y;
    vec3 viewport_upper_left = camera_pos - vec3(0, 0, cfg->focal_length) - viewport_u / 2.0f - viewport_v / 2.0f;
    vec3 pixel00_loc = viewport_upper_left + (pixel_delta_u + pixel_delta_v) * 0.5f;
    for (int k = 0; k < n; k++) {
        int i = pi[k], j = pj[k];
        pt_rng s = rng_init(cfg->seed, (uint32_t)(j * max_x + i), sample[k]);
        float jx = rng_uniform(&s), jy = rng_uniform(&s);
        vec3 p_center = pixel00_loc + (pixel_delta_u * (i - 0.5f + jx)) + (pixel_delta_v * (max_y - j - 0.5f + jy));
        vec3 direction = (p_center - camera_pos).normalized();
        for (int c = 0; c < 3; c++) { o[3 * k + c] = camera_pos[c]; d[3 * k + c] = direction[c]; }
    }
}

// render (main.cu:30-69) over rows [row_begin,row_end), multithreaded over rows.
// fb is the full-image AoS buffer fb[(j*max_x+i)*3+c]; only the tile's rows are
// written.  stats may be null; it is filled only by the counting build.
// `rows` (may be null) lists the rows to render; when null, rows
// [row_begin,row_end) are rendered.
int oracle_render_rows(void* scene, const oracle_config* cfg, const int* rows, int n_rows, float* fb, int n_threads,
                       oracle_stats* stats);

int oracle_render(void* scene, const oracle_config* cfg, float* fb, int n_threads, oracle_stats* stats) {
    return oracle_render_rows(scene, cfg, nullptr, 0, fb, n_threads, stats);
}

int oracle_render_rows(void* scene, const oracle_config* cfg, const int* rows, int n_rows, float* fb, int n_threads,
                       oracle_stats* stats) {
    Scene* sc = static_cast<Scene*>(scene);
    config_t config;
    config.camera_pos = vec3(cfg->camera_pos[0], cfg->camera_pos[1], cfg->camera_pos[2]);
    config.max_x = cfg->max_x; config.max_y = cfg->max_y;
    config.focal_length = cfg->focal_length; config.view_port_height = cfg->view_port_height;
    if (n_threads < 1) n_threads = 1;
    int row_begin = cfg->row_begin, row_end = cfg->row_end;
    if (row_begin < 0 || row_end > cfg->max_y || row_begin > row_end) return -1;
    if (rows) {
        for (int k = 0; k < n_rows; k++) if (rows[k] < 0 || rows[k] >= cfg->max_y) return -1;
        row_begin = 0; row_end = n_rows;
    }
    std::atomic<int> next_row(row_begin);
    std::vector<Stats> per_thread(n_threads);
    auto worker = [&](int tid) {
#ifdef PT_ORACLE_COUNT
        g_stats = &per_thread[tid];
#else
        (void)tid;
#endif
        for (;;) {
            int j = next_row.fetch_add(1);
            if (j >= row_end) break;
            if (rows) j = rows[j];
            for (int i = 0; i < config.max_x; i++) {
                vec3 c = renderPixel(config, sc->ptrs.data(), (int)sc->ptrs.size(), i, j,
                                     cfg->n_samples, cfg->n_bounces, cfg->seed, cfg->bsdf_model);
                float* px = fb + ((size_t)j * config.max_x + i) * 3;
                px[0] = c[0]; px[1] = c[1]; px[2] = c[2];
            }
        }
#ifdef PT_ORACLE_COUNT
        g_stats = nullptr;
#endif
    };
    std::vector<std::thread> threads;
    for (int t = 1; t < n_threads; t++) threads.emplace_back(worker, t);
    worker(0);
    for (auto& t : threads) t.join();
    if (stats) {
        Stats total;
        for (auto& s : per_thread) total.add(s);
        fill_stats(total, stats);
    }
    return 0;
}

// PPM P3 text exactly as main.cu:240-253 writes it: rows top to bottom
// (j = ny-1 .. 0), int(255.99*c) in double, no clamp.  Returns bytes written
// (excluding the terminating NUL) or -1 if cap is too small.
long oracle_ppm(const float* fb, int nx, int ny, char* out, long cap) {
    long n = 0;
    int w = snprintf(out, (size_t)cap, "P3\n%d %d\n255\n", nx, ny);
    if (w < 0 || w >= cap) return -1;
    n += w;
    for (int j = ny - 1; j >= 0; j--) {
        for (int i = 0; i < nx; i++) {
            size_t pixel_index = (size_t)j * nx + i;
            float r = fb[pixel_index * 3], g = fb[pixel_index * 3 + 1], b = fb[pixel_index * 3 + 2];
            int ir = int(255.99 * r), ig = int(255.99 * g), ib = int(255.99 * b);
            w = snprintf(out + n, (size_t)(cap - n), "%d %d %d\n", ir, ig, ib);
            if (w < 0 || w >= cap - n) return -1;
            n += w;
        }
    }
    return n;
}

}  // extern "C"
