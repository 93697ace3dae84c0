// Wave-level sample pool shared by the render kernels.
//
// The pixels of a launch are numbered row by row over the rows it renders
// (row_begin, row_begin + row_step, ...).  One wave64 owns a tile of P = 4 .. 64
// consecutive pixels of that numbering and the tile's P x spp samples form
// its pool, ordered item w = sample*P + pixel_in_tile.  Every loop iteration
// advances each lane's path by ONE segment.  Lanes whose path ended ask for new
// work: a ballot of the asking lanes plus a prefix count (mbcnt) hands them the
// next pool items in lane order, so all 64 lanes stay busy until the pool drains.
//
// A finished sample is added to its pixel's accumulator in LDS (the lane that
// traced a sample is usually not the lane that owns the pixel).  Accumulators
// are 32.32 fixed-point integers, so a pixel's sum does not depend on the order
// of the additions: the image is bitwise the same for every tile size, every
// split of the rows over launches or GPUs, and every run.  When the pool is
// empty the wave converts, divides by spp and writes the tile with coalesced
// stores.
#pragma once

#include "pt_internal.h"

namespace ptw {

constexpr int WAVES_PER_BLOCK = 4;
constexpr int BLOCK = 64 * WAVES_PER_BLOCK;
constexpr int ACC_WORDS = 64 * 3;   // 64-bit accumulators per wave

// LDS of one wave: accumulators and where each tile pixel goes in the framebuffer.
// Per-wave arrays in LDS: the tile's pixel sums (3 per pixel) and each pixel's framebuffer slot.
struct WaveLds {
    unsigned long long* acc;
    uint32_t* out_px;
};
// The same as static storage for a tile of up to 64 pixels.
struct WaveLdsStore {
    unsigned long long acc[ACC_WORDS];
    uint32_t out_px[64];
    __device__ __forceinline__ WaveLds refs() { WaveLds l; l.acc = acc; l.out_px = out_px; return l; }
};

__device__ __forceinline__ unsigned lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// Number of set bits of `mask` below this lane.
__device__ __forceinline__ unsigned prefix_count(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

struct PixelRef {
    int i, j;           // column, image row (row 0 = bottom, main.cu:38)
    uint32_t index;     // j*max_x + i: the RNG stream's pixel key (main.cu:38,60)
    uint32_t out;       // pixel slot in the framebuffer this launch writes
};
__device__ __forceinline__ PixelRef pixel_of(const RenderArgs& a, uint32_t lin) {
    PixelRef r;
    const uint32_t k = lin / (uint32_t)a.max_x;
    r.i = (int)(lin - k * (uint32_t)a.max_x);
    r.j = a.row_begin + (int)k * a.row_step;
    r.index = (uint32_t)r.j * (uint32_t)a.max_x + (uint32_t)r.i;
    r.out = a.fb_compact ? lin : r.index;
    return r;
}

// x as signed 32.32 fixed point (floor(x) in the high word, the fraction in the low word).  A NaN adds nothing;
// anything beyond +-cap (an infinite radiance included) counts as +-cap, where cap = 2^30 / n_samples: whatever the
// samples of a pixel are, their sum stays inside the 32 integer bits.
__device__ __forceinline__ unsigned long long to_fixed_in_range(float x) {      // |x| <= 2^30, not a NaN
    const float fl = __builtin_floorf(x);
    // (x - fl rounds to 1 for a tiny negative x; 2^32 is outside unsigned's range, so the fraction stops one float below)
    const unsigned lo = (unsigned)__builtin_fminf((x - fl) * 4294967296.0f, 4294967040.0f);
    const int hi = (int)fl;
    return ((unsigned long long)(unsigned)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long to_fixed(float x0, float cap) {
    return to_fixed_in_range(x0 == x0 ? __builtin_fminf(__builtin_fmaxf(x0, -cap), cap) : 0.0f);
}
__device__ __forceinline__ float from_fixed(unsigned long long v) {
    return (float)(int)(v >> 32) + (float)(unsigned)v * 2.3283064365386963e-10f;
}

__device__ __forceinline__ void add_sample(const WaveLds& l, uint32_t px, const float rgb[3], float cap) {
    // most paths end without having met a light: a zero adds nothing to a fixed-point sum (nor does a negative zero)
    if (rgb[0] == 0.0f && rgb[1] == 0.0f && rgb[2] == 0.0f) return;
    for (int c = 0; c < 3; c++)
        __hip_atomic_fetch_add(&l.acc[px * 3 + c], to_fixed(rgb[c], cap), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// What a wave traced, added to the launch's counters (pt_render_counted) by its first lane.
__device__ __forceinline__ void add_counters(const RenderArgs& a, uint32_t segments, unsigned lane) {
    if (a.counters != nullptr && lane == 0u) atomicAdd(&a.counters[0], (unsigned long long)segments);
}

// p_color = totalLight / n_samples (main.cu:67); consecutive floats of consecutive pixels leave together.
// A launch that traces a range of the frame's samples adds its sums to the caller's instead (same slots, 64-bit
// integers; atomically, so that launches on different streams may share the buffer).
__device__ __forceinline__ void store_tile(const RenderArgs& a, const WaveLds& l, uint32_t npx, unsigned lane) {
    if (a.sums != nullptr) {
        if (a.fb_layout == PT_FB_AOS) {
            for (uint32_t f = lane; f < npx * 3u; f += 64u) {
                const uint32_t px = (f * 43691u) >> 17;
                atomicAdd(&a.sums[(size_t)l.out_px[px] * 3u + (f - px * 3u)], l.acc[f]);
            }
        } else {
            const size_t plane = a.fb_compact ? (size_t)a.n_lin : (size_t)a.max_x * (size_t)a.max_y;
            if (lane < npx)
                for (int c = 0; c < 3; c++) atomicAdd(&a.sums[c * plane + l.out_px[lane]], l.acc[lane * 3 + c]);
        }
        return;
    }
    const float ns = (float)a.n_samples;
    if (a.fb_layout == PT_FB_AOS) {
        for (uint32_t f = lane; f < npx * 3u; f += 64u) {
            const uint32_t px = (f * 43691u) >> 17;          // f / 3 for f < 2^16
            a.fb[(size_t)l.out_px[px] * 3u + (f - px * 3u)] = from_fixed(l.acc[f]) / ns;
        }
    } else {
        const size_t plane = a.fb_compact ? (size_t)a.n_lin : (size_t)a.max_x * (size_t)a.max_y;
        if (lane < npx)
            for (int c = 0; c < 3; c++) a.fb[c * plane + l.out_px[lane]] = from_fixed(l.acc[lane * 3 + c]) / ns;
    }
}

// T supplies:  struct Path;
//              static void prologue(const RenderArgs&, Ctx&, uint32_t px, const PixelRef&);   // per tile pixel, by lane px
//              static void make_ray(const RenderArgs&, Ctx&, uint32_t px, uint32_t lin, uint32_t sample, float4& ray);
//              static void begin(Path&, const RenderArgs&, Ctx&, const float4& ray);
//              static bool step(Path&, const RenderArgs&, Ctx&);     // true = path continues
//              static void radiance(const Path&, float rgb[3]);
//              static void phase(Ctx&, int);                          // development profiling hook, normally empty
//              static constexpr bool STEP_IDLE_LANES;                 // step() may run for lanes without a path
//              static bool begins_segment(const Path&);               // the next step() starts a nearest-hit search (a tracer
//                                                                     // may spread one search over several steps)
//              static void idle(Path&);                               // a state step() can run on harmlessly
//
// Handing out work lane by lane would run the camera-ray code for the lanes whose path just ended, about 27 of 64 per
// iteration, at the price of a full wave.  Instead all 64 lanes make the rays of the next 64 pool items at once and park
// them in `stash` (LDS, one float4 per item: direction and RNG key); a lane that needs work takes the next parked ray.
// Items are handed out in pool order, lane by lane.
template <class T, class Ctx>
__device__ __forceinline__ void render_tile(const RenderArgs& a, uint32_t tile, const WaveLds& l, float4* stash, Ctx& ctx) {
    const unsigned lane = lane_id();
    const uint32_t lg = (uint32_t)a.tile_log2, P = 1u << lg;
    const uint32_t lin_base = tile << lg;
    const uint32_t npx = min(P, a.n_lin - lin_base);
    const uint32_t total = (uint32_t)a.n_samples << lg;

    for (uint32_t f = lane; f < 3u * P; f += 64u) l.acc[f] = 0ull;
    if (lane < npx) {
        const PixelRef ref = pixel_of(a, lin_base + lane);
        l.out_px[lane] = ref.out;
        T::prologue(a, ctx, lane, ref);
    }

    typename T::Path path;
    T::idle(path);
    uint32_t my_px = 0;
    bool active = false;
    uint32_t made = 0, taken = 0;   // wave-uniform: pool items whose ray has been made / handed out; parked = made - taken
    uint32_t segments = 0;          // wave-uniform: path segments traced
    for (;;) {
        const unsigned long long want = __ballot(!active);
        uint32_t need = (uint32_t)__popcll(want);
        if (need != 0u) {
            const uint32_t rank = prefix_count(want);
            uint32_t served = 0;
            for (int pass = 0; pass < 2 && need != 0u; pass++) {   // what is parked, then a fresh batch for the rest
                if (made == taken) {
                    if (made >= total) break;
                    const uint32_t w = made + lane;
                    float4 ray;
                    T::make_ray(a, ctx, w & (P - 1u), lin_base + (w & (P - 1u)), (w >> lg) + (uint32_t)a.sample_begin, ray);
                    stash[w & 63u] = ray;
                    made = min(made + 64u, total);
                }
                const uint32_t take = min(need, made - taken);
                if (!active && rank >= served && rank < served + take) {
                    const uint32_t item = taken + (rank - served);
                    const uint32_t px = item & (P - 1u);
                    if (px < npx) {
                        my_px = px;
                        T::begin(path, a, ctx, stash[item & 63u]);
                        active = true;
                    }
                }
                taken += take; served += take; need -= take;
            }
        }
        T::phase(ctx, 0);
        const unsigned long long active_m = __ballot(active);
        if (active_m == 0ull) {
            if (taken >= total) break;
            continue;
        }
        segments += (uint32_t)__popcll(__ballot(active && T::begins_segment(path)));
        if (T::STEP_IDLE_LANES) {
            // Lanes without a path step whatever state they hold (a finished path or the idle ray) and the result is
            // dropped: the wave spends the same time, and the path state has one definition per iteration instead of a
            // merge of "stepped" and "left alone" that costs a round of register copies.
            const bool go_on = T::step(path, a, ctx);
            T::phase(ctx, 5);
            if (active && !go_on) {
                float rgb[3];
                T::radiance(path, rgb);
                add_sample(l, my_px, rgb, a.sample_cap);
                active = false;
            }
        } else if (active) {
            const bool go_on = T::step(path, a, ctx);
            T::phase(ctx, 5);
            if (!go_on) {
                float rgb[3];
                T::radiance(path, rgb);
                add_sample(l, my_px, rgb, a.sample_cap);
                active = false;
            }
        }
        T::phase(ctx, 6);
    }
    add_counters(a, segments, lane);
    store_tile(a, l, npx, lane);
}

// The same pool, for tracers whose nearest-hit search is a WORKGROUP-wide operation (records staged through LDS
// with barriers inside): every wave of the block calls this, also one whose tile lies beyond the image, and all
// of them keep taking part in the search until the last wave of the block has drained its pool.
// T additionally supplies:  static void start(Path&, const RenderArgs&, Ctx&, uint32_t px, uint32_t lin, uint32_t sample);   // make_ray + begin
//                           static Best nearest_block(const Path&, const RenderArgs&, Ctx&);   // contains barriers
//                           static bool shade(Path&, const Best&, const RenderArgs&, Ctx&);    // true = path continues
template <class T, class Ctx>
__device__ __forceinline__ void render_tile_blocksync(const RenderArgs& a, uint32_t tile, bool tile_valid, const WaveLds& l, Ctx& ctx) {
    const unsigned lane = lane_id();
    const uint32_t lg = (uint32_t)a.tile_log2, P = 1u << lg;
    const uint32_t lin_base = tile_valid ? tile << lg : 0u;
    const uint32_t npx = tile_valid ? min(P, a.n_lin - lin_base) : 0u;
    const uint32_t total = tile_valid ? (uint32_t)a.n_samples << lg : 0u;

    for (uint32_t f = lane; f < 3u * P; f += 64u) l.acc[f] = 0ull;
    if (lane < npx) {
        const PixelRef ref = pixel_of(a, lin_base + lane);
        l.out_px[lane] = ref.out;
        T::prologue(a, ctx, lane, ref);
    }

    typename T::Path path;
    T::idle(path);
    uint32_t my_px = 0;
    bool active = false;
    uint32_t next = 0;   // wave-uniform
    uint32_t segments = 0;
    for (;;) {
        const unsigned long long want = __ballot(!active);
        if (want != 0ull && next < total) {
            const uint32_t w = next + prefix_count(want);
            next += (uint32_t)__popcll(want);
            if (!active && w < total) {
                const uint32_t px = w & (P - 1u), smp = (w >> lg) + (uint32_t)a.sample_begin;
                if (px < npx) {
                    my_px = px;
                    T::start(path, a, ctx, px, lin_base + px, smp);
                    active = true;
                }
            }
        }
        T::phase(ctx, 0);
        const unsigned long long active_m = __ballot(active);
        const int wave_busy = (active_m != 0ull) || (next < total);
        if (!__syncthreads_or(wave_busy)) break;            // block-uniform
        segments += (uint32_t)__popcll(active_m);
        const auto best = T::nearest_block(path, a, ctx);    // all lanes of all waves
        if (active) {
            const bool go_on = T::shade(path, best, a, ctx);
            T::phase(ctx, 5);
            if (!go_on) {
                float rgb[3];
                T::radiance(path, rgb);
                add_sample(l, my_px, rgb, a.sample_cap);
                active = false;
                T::idle(path);
            }
        }
        T::phase(ctx, 6);
    }
    if (tile_valid) { add_counters(a, segments, lane); store_tile(a, l, npx, lane); }
}

}  // namespace ptw
