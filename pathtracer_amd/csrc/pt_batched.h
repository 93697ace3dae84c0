// The render loop with camera rays in batches, per-tile culling and a queue of set-aside hits, shared by the fast
// kernels and the kernels in the reference's operation order.  It replaces the thread-per-pixel sample loop of
// main.cu:61-66 around tracePath (bsdf.h:73-168); the arithmetic of a path is the includer's (template parameter X).
//
// The plain wave-pool loop (pt_wavepool.h: render_tile) treats every segment alike: each iteration tests all
// primitives for all 64 lanes and runs the whole of the shading.  Two things are different here.
//
// Camera rays in batches.  The first segment of 64 samples is traced by all 64 lanes at once (camera_batch): rays
// are made, tested and shaded together, and what continues is parked in LDS as a secondary ray of 32 bytes (origin,
// RNG key | direction, pixel and material) for a lane of the main loop to pick up; paths that end at their first hit
// add their light to the pixel at once.  The rays of a batch leave one point towards a few adjacent pixels, so most
// of the scene cannot be hit by any of them: once per tile, lane l tests the box of record l against the tile's view
// pyramid (tile_candidates) and the ballot of the survivors is the list of records the batches test.  The main loop
// is then left with secondary segments only.
//
// Hits that need the full shading set aside.  In the plain loop every iteration runs Beer term, Fresnel ratio, mode
// draw, reflected and refracted directions for the few lanes whose hit needs them (7 % of the hits of the default
// scene), at the price of a full wave each time.  Here the loop body shades the common case only (X::shade_diffuse).
// A lane whose hit needs more writes its path and hit to a small queue in LDS and is free for new work at once; when
// enough have gathered (q_gate), free lanes take them back and X::shade_scatter runs for all of them together.  If
// many lanes need it in the same iteration (q_inline or more: a wave looking at glass) they keep their state in
// registers instead and are shaded at the same place one iteration later; camera rays that hit glass are parked with
// their hit and join them.  The queue never holds more than (q_gate - 1) + (q_inline - 1) <= q_cap entries: after the
// take-back step it is below q_gate (it takes whenever it holds q_gate or more, and the lanes that put entries in are
// free by then), and fewer than q_inline are put in per iteration.
//
// Arithmetic per path is the plain loop's, and pixel sums do not depend on the order of the additions, so the image
// is the plain loop's bit for bit (tested for both families).
//
// X supplies (all static, force-inlined):
//   types   V3 (x, y, z), Path (o, d, coef, contrib, key, seg), Best (t, id; id < 0 = miss; id >> 3 = primitive),
//           Surf (pos, nrm, inside, material), Ctx (mats, pxt, pxs, ...), Scene
//   Scene   scene(const RenderArgs&)                                   what the searches read
//   void    prologue(a, ctx, px, PixelRef), make_ray(a, ctx, px, sample, float4& ray)   ray = (direction, RNG key)
//   Best    scan(o, d, scene), scan_camera(o, d, scene, a, cand)       nearest hit: all records / a tile's candidates
//   const float* cull_boxes(a, int& n)                                 boxes of the records in scan_camera's bit order
//   Surf    surface(best, o, d, ctx)                                   the winner's record, from the ray
//   Surf    surface_parked(best, p, d, ctx)                            the same from the point park_point kept
//   V3      park_point(o, pos), camera_hit_point(cam, d, t)            what a set-aside hit keeps of its ray
//   void    shade_emissive(path, hit, M)                               bsdf.h:90-92
//   bool    shade_diffuse(path, surf, M, hit, keep, n_bounces)         bsdf.h:95-165 for f0 = transparency = 0, outside
//   bool    shade_scatter(path, surf, M, t, hit, n_bounces, ctx)       bsdf.h:95-165
//   V3      first_diffuse_coef(M)                                      throughput shade_diffuse leaves after a first hit
//   void    idle(path), phase(ctx, i)
#pragma once

#include "pt_internal.h"
#include "pt_wavepool.h"

namespace ptb {

#define PTB __device__ __forceinline__

template <class T> PTB T* lds_ptr(uint32_t off) { return (T*)(T __attribute__((address_space(3)))*)(uintptr_t)off; }
PTB uint32_t lds_off(const void* p) { return (uint32_t)(uintptr_t)(const void __attribute__((address_space(3)))*)p; }
PTB uint32_t uniform(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }

// The launch arguments as the kernel finds them in memory (they are its first parameter), behind a pointer the
// compiler cannot see through.  What the loop reads once per tile or once per batch (camera, framebuffer, culling
// boxes) is read through it where it is used, with scalar loads, instead of living in scalar registers across the
// whole loop: the loop's own scalars (record offsets and counts, queue counters, masks) then fit the register file
// without spills to vector lanes.
typedef const RenderArgs __attribute__((address_space(4)))* KArgsP;
PTB RenderArgs lazy_args(const RenderArgs& a) {
#if defined(__HIP_DEVICE_COMPILE__)   // (the host pass of the compiler only parses this function)
    KArgsP p = (KArgsP)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *p;
#else
    return a;
#endif
}

struct BatchLds {            // one wave's share of the launch's dynamic LDS
    unsigned long long* acc; // [3*P] pixel sums, 32.32 fixed point
    uint32_t* out_px;        // [P] framebuffer slot of each tile pixel
    float4* stash;           // [2][64] parked work: secondary ray (origin, key | direction, pixel | material << 8), or a
                             // camera ray with a hit that needs the full shading (t, id, -, key | direction, pixel | 1 << 31)
    float4* gq;              // [4][q_cap] set-aside hits: (point, t) | (d, id) | (coef, key) | (contrib, seg << 8 | pixel)
    float* pxt;              // [3][P] per tile pixel: i - 0.5, H - j - 0.5, K_px bits
};
constexpr uint32_t STASH_FLAG = 0x80000000u;

// Dynamic LDS of the launch, in this order: per wave [acc 3*P u64 | gq 4*q_cap float4 | stash 128 float4 |
// pxt 3*P float | out_px P u32], then whatever tables the kernel keeps in LDS.
__host__ __device__ inline size_t batched_wave_bytes(uint32_t P, uint32_t q_cap) { return (size_t)P * 24u + (size_t)q_cap * 64u + 2048u + (size_t)P * 12u + (size_t)P * 4u; }
// A wave's arrays from the LDS address of its block.
PTB void carve_wave_lds(uint32_t base, uint32_t P, uint32_t q_cap, BatchLds& L) {
    L.acc = lds_ptr<unsigned long long>(base);
    L.gq = lds_ptr<float4>(base + P * 24u);
    L.stash = L.gq + 4u * q_cap;
    L.pxt = reinterpret_cast<float*>(L.stash + 128);
    L.out_px = reinterpret_cast<uint32_t*>(L.pxt + 3u * P);
}

PTB void add_light(const BatchLds& L, uint32_t px, float r, float g, float b, float cap) {
    // most paths end without having met a light: a zero adds nothing to a fixed-point sum (nor does a negative zero)
    if (r == 0.0f && g == 0.0f && b == 0.0f) return;
    // one test for "some channel is a NaN or beyond the cap" (the compare is false for a NaN); the clamp itself
    // (ptw::to_fixed) then only runs for such samples
    if (!(__builtin_fmaxf(__builtin_fabsf(r), __builtin_fmaxf(__builtin_fabsf(g), __builtin_fabsf(b))) <= cap) || r != r || g != g || b != b) {
        __hip_atomic_fetch_add(&L.acc[px * 3u], ptw::to_fixed(r, cap), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_fetch_add(&L.acc[px * 3u + 1u], ptw::to_fixed(g, cap), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_fetch_add(&L.acc[px * 3u + 2u], ptw::to_fixed(b, cap), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        return;
    }
    __hip_atomic_fetch_add(&L.acc[px * 3u], ptw::to_fixed_in_range(r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    __hip_atomic_fetch_add(&L.acc[px * 3u + 1u], ptw::to_fixed_in_range(g), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    __hip_atomic_fetch_add(&L.acc[px * 3u + 2u], ptw::to_fixed_in_range(b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// The rectangle that the camera rays of a tile cross on the sample plane, relative to the camera: rays are
// cam + t * (X, Y, Z0) with Z0 = p00.z - cam.z and (X, Y) inside [X0, X1] x [Y0, Y1], the span of the tile's pixels
// (main.cu:40-63; jitter included, one pixel of margin).
struct TileRect { float X0, X1, Y0, Y1, Z0; };
PTB TileRect tile_rect(const RenderArgs& a, uint32_t lin_base, uint32_t npx) {
    const uint32_t W = (uint32_t)a.max_x;
    const uint32_t k0 = lin_base / W, k1 = (lin_base + npx - 1u) / W;
    const uint32_t i0 = k0 == k1 ? lin_base - k0 * W : 0u, i1 = k0 == k1 ? lin_base + npx - 1u - k0 * W : W - 1u;
    const int j0 = a.row_begin + (int)k0 * a.row_step, j1 = a.row_begin + (int)k1 * a.row_step;
    // sample-plane coordinates (main.cu:62): sx = i - 0.5 + U, sy = max_y - j - 0.5 + U, U in (0,1]
    const float sx0 = (float)i0 - 1.5f, sx1 = (float)i1 + 1.5f;
    const float sy0 = (float)(a.max_y - max(j0, j1)) - 1.5f, sy1 = (float)(a.max_y - min(j0, j1)) + 1.5f;
    const float xa = a.du * sx0 + (a.p00[0] - a.cam[0]), xb = a.du * sx1 + (a.p00[0] - a.cam[0]);
    const float ya = a.dv * sy0 + (a.p00[1] - a.cam[1]), yb = a.dv * sy1 + (a.p00[1] - a.cam[1]);
    TileRect r;
    r.X0 = __builtin_fminf(xa, xb); r.X1 = __builtin_fmaxf(xa, xb); r.Y0 = __builtin_fminf(ya, yb); r.Y1 = __builtin_fmaxf(ya, yb);
    r.Z0 = a.p00[2] - a.cam[2];
    return r;
}
// Can no camera ray of the tile pass through the box from (lx, ly, lz) to (hx, hy, hz) (relative to the camera)?  The
// box is projected onto the sample plane, clipped to the half space in front of the camera, by interval arithmetic.
// Anything undecidable (NaN from inf/inf, a box around the camera) fails the compares and is "not missed".
PTB bool rect_misses_box(const TileRect& r, float lx, float ly, float lz, float hx, float hy, float hz) {
    // ray parameter t = z / Z0 over the box; only t > 0 is in front of the camera
    const float ta = lz / r.Z0, tb = hz / r.Z0;
    const float t_hi = __builtin_fmaxf(ta, tb);
    const float t_lo = __builtin_fmaxf(__builtin_fminf(ta, tb), 1e-30f);
    const float slack = 1e-4f;
    // projection x / t over x in [lx, hx], t in [t_lo, t_hi] (t > 0): lowest from lx, highest from hx
    const float px_lo = __builtin_fminf(lx / t_lo, lx / t_hi), px_hi = __builtin_fmaxf(hx / t_lo, hx / t_hi);
    const float py_lo = __builtin_fminf(ly / t_lo, ly / t_hi), py_hi = __builtin_fmaxf(hy / t_lo, hy / t_hi);
    const float mx = slack * (__builtin_fabsf(r.X0) + __builtin_fabsf(r.X1)), my = slack * (__builtin_fabsf(r.Y0) + __builtin_fabsf(r.Y1));
    return (t_hi <= 0.0f) || (px_hi < r.X0 - mx) || (px_lo > r.X1 + mx) || (py_hi < r.Y0 - my) || (py_lo > r.Y1 + my);
}

// Which records can a camera ray of this tile hit?  Bit s = record s of `boxes` (6 floats each: low corner, high
// corner, padded by the uploader); lane s tests box s.  Without boxes, or with more than 64 records, every bit is set.
PTB unsigned long long tile_candidates(const RenderArgs& a, const float* boxes, int n_boxes, uint32_t lin_base, uint32_t npx, unsigned lane) {
    if (boxes == nullptr || n_boxes > 64) return ~0ull;
    const TileRect r = tile_rect(a, lin_base, npx);
    bool keep = false;
    if ((int)lane < n_boxes) {
        const float* B = boxes + 6 * lane;
        keep = !rect_misses_box(r, B[0] - a.cam[0], B[1] - a.cam[1], B[2] - a.cam[2], B[3] - a.cam[0], B[4] - a.cam[1], B[5] - a.cam[2]);
    }
    return __ballot(keep);
}

// The first segment of pool items made .. made + 63, one per lane: camera ray (main.cu:62-64), nearest hit among the
// tile's candidate records, surface, emissive term and — for a hit that shade_diffuse covers — new direction and
// roulette.  What goes on is parked in L.stash, packed from entry 0; returns the number of entries.
template <class X>
PTB uint32_t camera_batch(const RenderArgs& a, typename X::Ctx& ctx, const BatchLds& L, const typename X::Scene& sc, unsigned long long cand,
                          uint32_t made, uint32_t total, uint32_t npx, unsigned lane, uint32_t& segments) {
    using V3 = typename X::V3;
    using Best = typename X::Best;
    using Surf = typename X::Surf;
    const uint32_t lg = (uint32_t)a.tile_log2, P = 1u << lg;
    const uint32_t w = made + lane;
    const uint32_t px = w & (P - 1u);
    const bool valid = w < total && px < npx;
    segments += (uint32_t)__popcll(__ballot(valid));
    float4 ray;
    X::make_ray(a, ctx, px, (w >> lg) + (uint32_t)a.sample_begin, ray);
    const V3 o = X::v3(a.cam[0], a.cam[1], a.cam[2]), d = X::v3(ray.x, ray.y, ray.z);
    const uint32_t key = __float_as_uint(ray.w);
    const Best b0 = X::scan_camera(o, d, sc, a, cand);
    const bool hit = valid && b0.id >= 0;
    Best b = b0;
    b.id = hit ? b0.id : 0;                            // a miss reads primitive 0's records and is dropped below
    const Surf s = X::surface(b, o, d, ctx);
    const DMat& M = ctx.mats[s.material];
    // the path so far: coefficient 1, no light; shade_emissive and shade_diffuse as the main loop runs them
    typename X::Path p;
    p.o = o; p.d = d; p.coef = X::v3(1.0f, 1.0f, 1.0f); p.contrib = X::v3(0.0f, 0.0f, 0.0f); p.key = key; p.seg = 0;
    X::shade_emissive(p, hit, M);
    const bool lit = (p.contrib.x != 0.0f) || (p.contrib.y != 0.0f) || (p.contrib.z != 0.0f);
    const float m_f0 = M.f0, m_tr = M.transparency;
    const bool special = hit & ((m_f0 != 0.0f) | (m_tr != 0.0f));
    const V3 light = p.contrib;
    const bool go_on = X::shade_diffuse(p, s, M, hit, false, a.n_bounces);
    // a secondary ray carries no light of its own: a hit that is lit and goes on keeps its camera ray and is shaded
    // again from its hit, with the hits that need shade_scatter anyway
    const bool as_hit = special || (hit && go_on && lit);
    const bool as_ray = hit && go_on && !as_hit;
    if (hit && !as_hit && !go_on && lit) add_light(L, px, light.x, light.y, light.z, a.sample_cap);
    const unsigned long long park = __ballot(as_hit || as_ray);
    const uint32_t e = ptw::prefix_count(park);
    if (as_ray) {
        L.stash[e] = make_float4(p.o.x, p.o.y, p.o.z, __uint_as_float(key));
        L.stash[64u + e] = make_float4(p.d.x, p.d.y, p.d.z, __uint_as_float(px | ((uint32_t)s.material << 8)));
    }
    if (as_hit) {
        L.stash[e] = make_float4(b.t, __int_as_float(b.id), 0.0f, __uint_as_float(key));
        L.stash[64u + e] = make_float4(d.x, d.y, d.z, __uint_as_float(px | STASH_FLAG));
    }
    return (uint32_t)__popcll(park);
}

template <class X>
PTB void render_tile_batched(const RenderArgs& a, uint32_t tile, const BatchLds& L, typename X::Ctx& ctx) {
    using V3 = typename X::V3;
    using Best = typename X::Best;
    using Surf = typename X::Surf;
    enum { FREE = 0, ACTIVE = 1, GLASS = 2 };
    const unsigned lane = ptw::lane_id();
    const uint32_t lg = (uint32_t)a.tile_log2, P = 1u << lg;
    const uint32_t lin_base = tile << lg;
    const uint32_t npx = min(P, a.n_lin - lin_base);
    const uint32_t total = (uint32_t)a.n_samples << lg;
    const uint32_t QC = (uint32_t)a.q_cap, QG = (uint32_t)a.q_gate, QI = (uint32_t)a.q_inline;

    for (uint32_t k = lane; k < 3u * P; k += 64u) L.acc[k] = 0ull;
    unsigned long long cand = ~0ull;
    {
        const RenderArgs la = lazy_args(a);
        if (lane < npx) {
            const ptw::PixelRef ref = ptw::pixel_of(la, lin_base + lane);
            L.out_px[lane] = ref.out;
            X::prologue(la, ctx, lane, ref);
        }
#ifndef PT_NO_CULL                   // (measurement aid: camera rays test every record)
        int n_boxes = 0;
        const float* boxes = X::cull_boxes(la, n_boxes);
        cand = tile_candidates(la, boxes, n_boxes, lin_base, npx, lane);
#endif
    }

    typename X::Path path;
    X::idle(path);
    uint32_t my_px = 0;
    int st = FREE;
    float ht = 0.0f;                 // a GLASS lane's hit: distance and code; what park_point kept of its ray is path.o
    int hid = 0;
    bool camera_hit = false;         // ... and whether it is a camera ray's, whose emissive term is still to be added
    uint32_t made = 0;               // wave-uniform: pool items the batches have traced
    uint32_t parked = 0;             // wave-uniform: entries in the stash
    uint32_t qn = 0;                 // wave-uniform: entries in the queue
    uint32_t segments = 0;           // wave-uniform: path segments traced (camera rays of the batches + ACTIVE lanes of step 4)
    const typename X::Scene sc = X::scene(a);
    for (;;) {
        // wave-uniform by construction; said once per iteration so that the compiler keeps them in scalar registers
        made = uniform(made); parked = uniform(parked); qn = uniform(qn);
        // 1. free lanes take set-aside hits back
        unsigned long long want = __ballot(st == FREE);
        if (want != 0ull && qn != 0u && (qn >= QG || __ballot(st == GLASS) != 0ull || (made >= total && parked == 0u))) {
            const uint32_t rank = ptw::prefix_count(want);
            const uint32_t take = min((uint32_t)__popcll(want), qn);
            if (st == FREE && rank < take) {
                const uint32_t e = qn - 1u - rank;
                const float4 r0 = L.gq[e], r1 = L.gq[QC + e], r2 = L.gq[2u * QC + e], r3 = L.gq[3u * QC + e];
                path.o = X::v3(r0.x, r0.y, r0.z); ht = r0.w;
                path.d = X::v3(r1.x, r1.y, r1.z); hid = __float_as_int(r1.w);
                path.coef = X::v3(r2.x, r2.y, r2.z); path.key = __float_as_uint(r2.w);
                path.contrib = X::v3(r3.x, r3.y, r3.z);
                const uint32_t sp = __float_as_uint(r3.w);
                path.seg = (int)(sp >> 8); my_px = sp & 255u;
                camera_hit = false;
                st = GLASS;
            }
            qn -= take;
            want = __ballot(st == FREE);
        }
        // 2. parked work for the lanes that are still free: what is parked, then a camera batch if that was not
        //    enough, then what the batch parked (lanes it cannot serve wait one iteration)
        uint32_t need = (uint32_t)__popcll(want);
        if (need != 0u) {
            const uint32_t rank = ptw::prefix_count(want);
            uint32_t served = 0;
            for (int pass = 0; pass < 2 && need != 0u; pass++) {
                if (parked == 0u) {
                    if (made >= total) break;
                    { const RenderArgs la = lazy_args(a); parked = camera_batch<X>(la, ctx, L, sc, cand, made, total, npx, lane, segments); }
#ifdef PT_BATCH_ONLY                 // measurement aid: camera batches alone (what they park is dropped)
                    parked = 0;
#endif
                    made = min(made + 64u, total);
                    X::phase(ctx, 7);
                }
                const uint32_t take = min(need, parked);
                if (st == FREE && rank >= served && rank < served + take) {
                    const uint32_t e = parked - 1u - (rank - served);
                    const float4 A = L.stash[e], B = L.stash[64u + e];
                    const uint32_t meta = __float_as_uint(B.w);
                    my_px = meta & 255u;
                    path.d = X::v3(B.x, B.y, B.z);
                    path.key = __float_as_uint(A.w);
                    path.contrib = X::v3(0.0f, 0.0f, 0.0f);
                    if (meta & STASH_FLAG) {
                        ht = A.x; hid = __float_as_int(A.y);
                        { const RenderArgs la = lazy_args(a); path.o = X::camera_hit_point(X::v3(la.cam[0], la.cam[1], la.cam[2]), path.d, ht); }
                        path.coef = X::v3(1.0f, 1.0f, 1.0f);
                        path.seg = 0;
                        camera_hit = true;
                        st = GLASS;
                    } else {
                        // throughput after the first hit, as shade_diffuse left it
                        path.o = X::v3(A.x, A.y, A.z);
                        path.coef = X::first_diffuse_coef(ctx.mats[(meta & ~STASH_FLAG) >> 8]);
                        path.seg = 1;
                        st = ACTIVE;
                    }
                }
                parked -= take; served += take; need -= take;
            }
        }
        X::phase(ctx, 0);
        // 3. the full shading for the lanes that hold a hit which needs it
        if (__ballot(st == GLASS) != 0ull) {
            if (st == GLASS) {
                Best b; b.t = ht; b.id = hid;
                const Surf s = X::surface_parked(b, path.o, path.d, ctx);
                const DMat& M = ctx.mats[s.material];
                X::shade_emissive(path, camera_hit, M);
                const bool go_on = X::shade_scatter(path, s, M, ht, true, a.n_bounces, ctx);
                st = ACTIVE;
                if (!go_on) {
                    add_light(L, my_px, path.contrib.x, path.contrib.y, path.contrib.z, a.sample_cap);
                    st = FREE;
                }
            }
        }
        X::phase(ctx, 3);
        if (__ballot(st != FREE) == 0ull) {
            if (made >= total && parked == 0u && qn == 0u) break;
            continue;
        }
        segments += (uint32_t)__popcll(__ballot(st == ACTIVE));
        // 4. one segment for every lane (a free lane steps whatever state it holds; the result is dropped)
        const Best b0 = X::scan(path.o, path.d, sc);
        X::phase(ctx, 1);
        const bool hit = b0.id >= 0;
        Best b = b0;
        b.id = hit ? b0.id : 0;                        // a miss reads primitive 0's records and is dropped below
        const Surf s = X::surface(b, path.o, path.d, ctx);
        const DMat& M = ctx.mats[s.material];
        X::phase(ctx, 2);
        X::shade_emissive(path, hit, M);
        // (plain & and |: both material fields are fetched at once instead of one behind a branch on the other)
        const float m_f0 = M.f0, m_tr = M.transparency;
        const bool special = hit & (st == ACTIVE) & ((m_f0 != 0.0f) | (m_tr != 0.0f) | ((path.seg > 0) & s.inside));
        // 5. hits that need the full shading: into the queue (the lane is free at once), or kept in registers
        const unsigned long long special_m = __ballot(special);
        bool keep = false;
        if (special_m != 0ull) {
            const uint32_t g = (uint32_t)__popcll(special_m);
            if (g >= QI || qn + g > QC) {
                keep = special;
            } else {
                if (special) {
                    const uint32_t e = qn + ptw::prefix_count(special_m);
                    const V3 pp = X::park_point(path.o, s.pos);
                    L.gq[e] = make_float4(pp.x, pp.y, pp.z, b.t);
                    L.gq[QC + e] = make_float4(path.d.x, path.d.y, path.d.z, __int_as_float(b.id));
                    L.gq[2u * QC + e] = make_float4(path.coef.x, path.coef.y, path.coef.z, __uint_as_float(path.key));
                    L.gq[3u * QC + e] = make_float4(path.contrib.x, path.contrib.y, path.contrib.z, __uint_as_float(((uint32_t)path.seg << 8) | my_px));
                    st = FREE;
                }
                qn += g;
            }
        }
        if (keep) { st = GLASS; ht = b.t; hid = b.id; camera_hit = false; }
        X::phase(ctx, 4);
        // 6. the common case for everyone else (a `keep` lane comes out with d, coef and seg unchanged and o = park_point)
        const bool go_on = X::shade_diffuse(path, s, M, hit, keep, a.n_bounces);
        X::phase(ctx, 5);
        if (st == ACTIVE && !go_on) {
            add_light(L, my_px, path.contrib.x, path.contrib.y, path.contrib.z, a.sample_cap);
            st = FREE;
        }
        X::phase(ctx, 6);
    }
    // p_color = totalLight / n_samples (main.cu:67); consecutive floats of consecutive pixels leave together
    const RenderArgs le = lazy_args(a);
    ptw::add_counters(le, segments, lane);
    if (le.sums != nullptr) {
        ptw::WaveLds l; l.acc = L.acc; l.out_px = L.out_px;
        ptw::store_tile(le, l, npx, lane);
        return;
    }
    const float ns = (float)le.n_samples;
    if (le.fb_layout == PT_FB_AOS) {
        for (uint32_t f = lane; f < npx * 3u; f += 64u) {
            const uint32_t px = (f * 43691u) >> 17;          // f / 3 for f < 2^16
            le.fb[(size_t)L.out_px[px] * 3u + (f - px * 3u)] = ptw::from_fixed(L.acc[f]) / ns;
        }
    } else {
        const size_t plane = le.fb_compact ? (size_t)le.n_lin : (size_t)le.max_x * (size_t)le.max_y;
        if (lane < npx)
            for (int c = 0; c < 3; c++) le.fb[c * plane + L.out_px[lane]] = ptw::from_fixed(L.acc[lane * 3 + c]) / ns;
    }
}

}  // namespace ptb
