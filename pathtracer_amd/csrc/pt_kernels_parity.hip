// Kernels in the reference's operation order.  Build this file with
// -ffp-contract=off (see Makefile): every float operation rounds on its own.
#include "pt_parity.h"
#include "pt_wavepool.h"
#include "pt_batched.h"

using namespace ptp;

namespace {

struct ParityCtx {};

struct ParityTracer {
    using Path = ptp::Path;
    static constexpr bool STEP_IDLE_LANES = true;
    static __device__ __forceinline__ void idle(Path& p) {
        p.o = v3(0.0f, 0.0f, 0.0f); p.d = v3(0.0f, 0.0f, 0.0f);
        p.coef = v3(0.0f, 0.0f, 0.0f); p.contrib = v3(0.0f, 0.0f, 0.0f); p.key = 0; p.seg = 0;
    }
    // A pool item past the launch's last pixel (ragged tile) makes a ray that
    // is never begun, so its pixel is clamped into the launch.
    static __device__ __forceinline__ void make_ray(const RenderArgs& a, ParityCtx&, uint32_t, uint32_t lin, uint32_t sample, float4& ray) {
        const ptw::PixelRef ref = ptw::pixel_of(a, lin < a.n_lin ? lin : 0u);
        const uint32_t key = rng_key(a.seed_hash, ref.index, sample);
        V3 o, d;
        camera_ray(a, ref.i, ref.j, key, &o, &d);
        ray = make_float4(d.x, d.y, d.z, __uint_as_float(key));
    }
    static __device__ __forceinline__ void begin(Path& p, const RenderArgs& a, ParityCtx&, const float4& ray) {
        p.key = __float_as_uint(ray.w);
        p.o = v3(a.cam[0], a.cam[1], a.cam[2]);
        p.d = v3(ray.x, ray.y, ray.z);
        p.coef = v3(1.0f, 1.0f, 1.0f);
        p.contrib = v3(0.0f, 0.0f, 0.0f);
        p.seg = 0;
    }
    static __device__ __forceinline__ void prologue(const RenderArgs&, ParityCtx&, uint32_t, const ptw::PixelRef&) {}
    static __device__ __forceinline__ bool step(Path& p, const RenderArgs& a, ParityCtx&) {
        if (a.bsdf_model == PT_BSDF_CLASSES) return path_step_alt(p, a.pscene, a.mats, a.n_bounces);
        return path_step(p, a.pscene, a.mats, a.n_bounces);
    }
    static __device__ __forceinline__ void radiance(const Path& p, float rgb[3]) {
        rgb[0] = p.contrib.x; rgb[1] = p.contrib.y; rgb[2] = p.contrib.z;
    }
    static __device__ __forceinline__ void phase(ParityCtx&, int) {}
    static __device__ __forceinline__ bool begins_segment(const Path&) { return true; }
};

__global__ __launch_bounds__(ptw::BLOCK) void render_parity_kernel(RenderArgs a, uint32_t n_tiles) {
    __shared__ ptw::WaveLdsStore lds[ptw::WAVES_PER_BLOCK];
    __shared__ float4 stash[ptw::WAVES_PER_BLOCK][64];
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x * ptw::WAVES_PER_BLOCK + wave;
    if (tile >= n_tiles) return;
    ParityCtx ctx;
    ptw::render_tile<ParityTracer>(a, tile, lds[wave].refs(), stash[wave], ctx);
}

// The reference's operation order for the loop of pt_batched.h (scenes without a hierarchy, bsdf.h shading): typed
// loops over the table sorted by shape type, camera batches that test only their tile's candidates, and the common
// diffuse case shaded in the loop body while hits that need Beer term, Fresnel ratio and the other directions are set
// aside and shaded together.  Every float operation of a path is the plain loop's (path_step), so the image is
// render_parity_kernel's bit for bit.  A set-aside hit keeps its ray's origin: the winner's record is rebuilt from the
// ray in local space, as the reference does (shapes.h:42-48,125-129).
struct ParityBatched {
    using V3 = ptp::V3;
    using Path = ptp::Path;
    struct Best { float t; int id; };                  // id = (primitive << 3) | surface code, negative = miss
    struct Surf { V3 pos, nrm; bool inside; int material; };
    struct Ctx { const DMat* mats; const DPrim* prims; float* pxt; uint32_t pxs; };
    using Scene = PScene;
    static __device__ __forceinline__ V3 v3(float x, float y, float z) { return ptp::v3(x, y, z); }
    static __device__ __forceinline__ Scene scene(const RenderArgs& a) { return a.pscene; }
    // Lane px prepares pixel px of the tile: i - 0.5 and max_y - j - 0.5 of main.cu:62 (both exact in float) and the
    // pixel's RNG key.
    static __device__ __forceinline__ void prologue(const RenderArgs& a, Ctx& c, uint32_t px, const ptw::PixelRef& ref) {
        c.pxt[px] = ref.i - 0.5f;
        c.pxt[c.pxs + px] = a.max_y - ref.j - 0.5f;
        c.pxt[2u * c.pxs + px] = __uint_as_float(rng_hash(ref.index ^ a.seed_hash));
    }
    static __device__ __forceinline__ void make_ray(const RenderArgs& a, Ctx& c, uint32_t px, uint32_t sample, float4& ray) {
        const uint32_t key = rng_hash(__float_as_uint(c.pxt[2u * c.pxs + px]) + 0x9E3779B9u * (sample + 1u));
        V3 o, d;
        camera_ray_at(a, c.pxt[px], c.pxt[c.pxs + px], key, &o, &d);
        ray = make_float4(d.x, d.y, d.z, __uint_as_float(key));
    }
    static __device__ __forceinline__ const float* cull_boxes(const RenderArgs& a, int& n) { n = a.pscene.n_prims; return a.pscene.boxes; }
    static __device__ __forceinline__ Best pack(const Nearest& n) { Best b; b.t = n.t; b.id = n.prim < 0 ? -8 : ((n.prim << 3) | n.surf); return b; }
    static __device__ __forceinline__ Best scan(V3 o, V3 d, const Scene& S) { return pack(nearest_scan_typed<false>(o, d, S, ~0ull)); }
    static __device__ __forceinline__ Best scan_camera(V3 o, V3 d, const Scene& S, const RenderArgs&, unsigned long long cand) {
        if (S.boxes != nullptr && S.n_prims <= 64) return pack(nearest_scan_typed<true>(o, d, S, cand));
        return pack(nearest_scan_typed<false>(o, d, S, ~0ull));
    }
    static __device__ __forceinline__ Surf surface(const Best& b, V3 o, V3 d, Ctx& c) {
        Nearest n; n.prim = b.id >> 3; n.t = b.t; n.surf = b.id & 7;
        const Hit h = finish_hit(n, o, d, c.prims);
        Surf s; s.pos = h.pos; s.nrm = h.nrm; s.inside = h.inside; s.material = c.prims[n.prim].material;
        return s;
    }
    static __device__ __forceinline__ Surf surface_parked(const Best& b, V3 o, V3 d, Ctx& c) { return surface(b, o, d, c); }
    static __device__ __forceinline__ V3 park_point(V3 o, V3) { return o; }
    static __device__ __forceinline__ V3 camera_hit_point(V3 cam, V3, float) { return cam; }
    static __device__ __forceinline__ Hit as_hit(const Surf& s, float t) { Hit h; h.prim = 0; h.t = t; h.pos = s.pos; h.nrm = s.nrm; h.inside = s.inside; return h; }
    static __device__ __forceinline__ void shade_emissive(Path& p, bool hit, const DMat& M) { step_emissive(p, hit, M); }
    static __device__ __forceinline__ bool shade_diffuse(Path& p, const Surf& s, const DMat& M, bool hit, bool keep, int n_bounces) {
        return step_diffuse(p, as_hit(s, 0.0f), hit, keep, M, n_bounces);
    }
    static __device__ __forceinline__ bool shade_scatter(Path& p, const Surf& s, const DMat& M, float t, bool hit, int n_bounces, Ctx&) {
        return step_scatter(p, as_hit(s, t), hit, M, n_bounces);
    }
    // what step_diffuse leaves of a throughput of 1: (1 * albedo) / max(albedo)
    static __device__ __forceinline__ V3 first_diffuse_coef(const DMat& M) {
        const V3 coef = ptp::v3(1.0f, 1.0f, 1.0f) * ptp::v3(M.albedo[0], M.albedo[1], M.albedo[2]);
        return coef / fmaxf(coef.x, fmaxf(coef.y, coef.z));
    }
    // a lane without a path steps a harmless ray that is not degenerate (a zero direction would send the wave's
    // search down the scene-order loop)
    static __device__ __forceinline__ void idle(Path& p) {
        p.o = ptp::v3(0.0f, 0.0f, 0.0f); p.d = ptp::v3(0.0f, 0.0f, 1.0f);
        p.coef = ptp::v3(0.0f, 0.0f, 0.0f); p.contrib = ptp::v3(0.0f, 0.0f, 0.0f); p.key = 0; p.seg = 0;
    }
    static __device__ __forceinline__ void phase(Ctx&, int) {}
};

template <bool LDS_MATS>
__global__ __launch_bounds__(ptw::BLOCK) __attribute__((amdgpu_waves_per_eu(6, 6))) void render_parity_batched_kernel(RenderArgs a, uint32_t n_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    const uint32_t P = 1u << a.tile_log2;
    const size_t per_wave = ptb::batched_wave_bytes(P, (uint32_t)a.q_cap);
    ParityBatched::Ctx ctx;
    ctx.prims = a.pscene.prims;
    if (LDS_MATS) {
        DMat* lmats = reinterpret_cast<DMat*>(dyn_lds + per_wave * ptw::WAVES_PER_BLOCK);
        const uint32_t* gm = reinterpret_cast<const uint32_t*>(a.mats);
        uint32_t* lm = reinterpret_cast<uint32_t*>(lmats);
        for (uint32_t w = threadIdx.x; w < (uint32_t)a.n_mats * (sizeof(DMat) / 4); w += ptw::BLOCK) lm[w] = gm[w];
        __syncthreads();
        ctx.mats = lmats;
    } else {
        ctx.mats = a.mats;
    }
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x * ptw::WAVES_PER_BLOCK + wave;
    if (tile >= n_tiles) return;
    ptb::BatchLds L;
    ptb::carve_wave_lds(ptb::lds_off(dyn_lds + per_wave * wave), P, (uint32_t)a.q_cap, L);
    ctx.pxt = L.pxt; ctx.pxs = P;
    ptb::render_tile_batched<ParityBatched>(a, tile, L, ctx);
}

__device__ __forceinline__ V3 load3(const float* p, int i) { return v3(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }
__device__ __forceinline__ void store3(float* p, int i, V3 v) { if (p) { p[3 * i] = v.x; p[3 * i + 1] = v.y; p[3 * i + 2] = v.z; } }

__global__ __launch_bounds__(256) void raypath_kernel(const DPrim* __restrict__ prims, int prim, const float* __restrict__ o,
                                                      const float* __restrict__ d, int n, int filter, int32_t* dn,
                                                      int32_t* h1, float* t1, float* p1, float* n1,
                                                      int32_t* h2, float* t2, float* p2, float* n2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DPrim& P = prims[prim];
    V3 ro = load3(o, i), rd = load3(d, i);
    V3 ol = ro, dl = rd;
    if (filter) { ol = xpoint(P.inv, ro); dl = xvec(P.inv, rd); }
    LHit h = shape_local(P.type, ol, dl, P.bounds);
    if (filter) eps_filter(h);
    if (dn) dn[i] = h.n;
    const bool v1 = h.n >= 1 && h.h1, v2 = h.n >= 2 && h.h2;
    V3 pw = v3(0, 0, 0), nw = v3(0, 0, 0);
    if (v1) world_point(P, ol, dl, h.t1, h.s1, &pw, &nw);
    if (h1) h1[i] = v1 ? 1 : 0;
    if (t1) t1[i] = v1 ? h.t1 : 0.0f;
    store3(p1, i, pw); store3(n1, i, nw);
    pw = v3(0, 0, 0); nw = v3(0, 0, 0);
    if (v2) world_point(P, ol, dl, h.t2, h.s2, &pw, &nw);
    if (h2) h2[i] = v2 ? 1 : 0;
    if (t2) t2[i] = v2 ? h.t2 : 0.0f;
    store3(p2, i, pw); store3(n2, i, nw);
}

__global__ __launch_bounds__(256) void nearest_kernel(PScene S, const float* __restrict__ o,
                                                      const float* __restrict__ d, int n, int32_t* prim, float* t,
                                                      float* pos, float* nrm, int32_t* inside) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Hit h = nearest_hit(load3(o, i), load3(d, i), S);
    const bool hit = h.prim >= 0;
    if (prim) prim[i] = h.prim;
    if (t) t[i] = hit ? h.t : 0.0f;
    store3(pos, i, hit ? h.pos : v3(0, 0, 0));
    store3(nrm, i, hit ? h.nrm : v3(0, 0, 0));
    if (inside) inside[i] = hit && h.inside ? 1 : 0;
}

__global__ __launch_bounds__(256) void trace_parity_kernel(PScene S, const DMat* __restrict__ mats,
                                                           const float* __restrict__ o, const float* __restrict__ d,
                                                           const uint32_t* __restrict__ pixel, const uint32_t* __restrict__ sample,
                                                           int n, uint32_t seed_hash, int n_bounces, int bsdf_model, float* rgb,
                                                           float* end_o, float* end_d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Path p;
    p.key = rng_key(seed_hash, pixel[i], sample[i]);
    p.o = load3(o, i); p.d = load3(d, i);
    p.coef = v3(1.0f, 1.0f, 1.0f); p.contrib = v3(0.0f, 0.0f, 0.0f); p.seg = 0;
    // The steps are asked for "no depth limit", so that false means the path ended (miss, roulette); the depth is
    // counted here.  The reference leaves in `r` the last ray it assigned (bsdf.h:165): a segment that ends the path
    // assigns none, the last segment of a path that runs out of bounces does.
    for (;;) {
        const V3 o0 = p.o, d0 = p.d;
        const bool alive = bsdf_model == PT_BSDF_CLASSES ? path_step_alt(p, S, mats, 0x7fffffff)
                                                         : path_step(p, S, mats, 0x7fffffff);
        if (!alive) { p.o = o0; p.d = d0; break; }
        if (p.seg > n_bounces) break;
    }
    store3(rgb, i, p.contrib);
    if (end_o) store3(end_o, i, p.o);
    if (end_d) store3(end_d, i, p.d);
}

__global__ void rng_kernel(uint32_t seed_hash, uint32_t pixel, uint32_t sample, int n_slots, uint32_t* bits, float* u) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_slots) return;
    const uint32_t key = rng_key(seed_hash, pixel, sample);
    const uint32_t b = rng_bits(key, (uint32_t)k);
    if (bits) bits[k] = b;
    if (u) u[k] = rng_slot_uniform(key, (uint32_t)k);
}

}  // namespace

int ptk_render_parity(const RenderArgs& a, hipStream_t s) {
    if (a.n_lin == 0) return 0;
    const uint32_t n_tiles = (a.n_lin + (1u << a.tile_log2) - 1u) >> a.tile_log2;
    const uint32_t blocks = (n_tiles + ptw::WAVES_PER_BLOCK - 1) / ptw::WAVES_PER_BLOCK;
    if (a.q_cap > 0) {
        // the loop with camera batches (pt_render sets q_cap for scenes it can serve)
        const bool lds_mats = a.n_mats <= 32;
        const size_t lds = ptb::batched_wave_bytes(1u << a.tile_log2, (uint32_t)a.q_cap) * ptw::WAVES_PER_BLOCK + (lds_mats ? (size_t)a.n_mats * sizeof(DMat) : 0u);
        if (lds_mats) hipLaunchKernelGGL(render_parity_batched_kernel<true>, dim3(blocks), dim3(ptw::BLOCK), lds, s, a, n_tiles);
        else hipLaunchKernelGGL(render_parity_batched_kernel<false>, dim3(blocks), dim3(ptw::BLOCK), lds, s, a, n_tiles);
    } else {
        hipLaunchKernelGGL(render_parity_kernel, dim3(blocks), dim3(ptw::BLOCK), 0, s, a, n_tiles);
    }
    PT_CHECK(hipGetLastError());
    return 0;
}

int ptk_raypath_batch(const pt_scene* sc, int prim, const float* o, const float* d, int n, int filter,
                      int32_t* dn, int32_t* h1, float* t1, float* p1, float* n1,
                      int32_t* h2, float* t2, float* p2, float* n2, hipStream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(raypath_kernel, dim3((n + 255) / 256), dim3(256), 0, s, sc->d_prims, prim, o, d, n, filter,
                       dn, h1, t1, p1, n1, h2, t2, p2, n2);
    PT_CHECK(hipGetLastError());
    return 0;
}

int ptk_nearest_batch(const pt_scene* sc, const float* o, const float* d, int n, int32_t* prim, float* t,
                      float* pos, float* nrm, int32_t* inside, hipStream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(nearest_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pt_parity_scene(sc), o, d, n,
                       prim, t, pos, nrm, inside);
    PT_CHECK(hipGetLastError());
    return 0;
}

int ptk_trace_paths_parity(const pt_scene* sc, const float* o, const float* d, const uint32_t* pixel,
                           const uint32_t* sample, int n, uint32_t seed_hash, int n_bounces, int bsdf_model, float* rgb,
                           float* end_o, float* end_d, hipStream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(trace_parity_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pt_parity_scene(sc), sc->d_mats,
                       o, d, pixel, sample, n, seed_hash, n_bounces, bsdf_model, rgb, end_o, end_d);
    PT_CHECK(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void bsdf_parity_kernel(int fn, const float* __restrict__ in, int n, float* __restrict__ out) {
    using namespace ptp;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    switch (fn) {
        case PT_FN_SAMPLE_HEMISPHERE: { const V3 r = sample_hemisphere(in[2 * i], in[2 * i + 1]); out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z; break; }
        case PT_FN_FRESNEL_SCHLICK: out[i] = fresnel_schlick(in[3 * i], in[3 * i + 1], in[3 * i + 2]); break;
        case PT_FN_FRESNEL_RATIO: {
            const float* a = in + 10 * i;
            out[i] = fresnel_ratio(v3(a[0], a[1], a[2]), v3(a[3], a[4], a[5]), a[6], a[7], a[8], a[9]);
            break;
        }
        case PT_FN_REFLECT: { const float* a = in + 6 * i; const V3 r = reflect(v3(a[0], a[1], a[2]), v3(a[3], a[4], a[5])); out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z; break; }
        case PT_FN_REFRACT: { const float* a = in + 7 * i; const V3 r = refract(v3(a[0], a[1], a[2]), v3(a[3], a[4], a[5]), a[6]); out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z; break; }
        case PT_FN_LERP: { const float* a = in + 7 * i; const V3 r = lerp3(v3(a[0], a[1], a[2]), v3(a[3], a[4], a[5]), a[6]); out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z; break; }
        default: { const V3 r = normalized(v3(in[3 * i], in[3 * i + 1], in[3 * i + 2])); out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z; break; }
    }
}

int ptk_bsdf_batch_parity(int fn, const float* in, int n, float* out, hipStream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(bsdf_parity_kernel, dim3((n + 255) / 256), dim3(256), 0, s, fn, in, n, out);
    PT_CHECK(hipGetLastError());
    return 0;
}

int ptk_rng_draws(uint32_t seed_hash, uint32_t pixel, uint32_t sample, int n_slots, uint32_t* bits, float* u, hipStream_t s) {
    if (n_slots <= 0) return 0;
    hipLaunchKernelGGL(rng_kernel, dim3((n_slots + 63) / 64), dim3(64), 0, s, seed_hash, pixel, sample, n_slots, bits, u);
    PT_CHECK(hipGetLastError());
    return 0;
}
