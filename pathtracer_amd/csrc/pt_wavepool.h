// Wave-level sample pool shared by the render kernels.
//
// One wave64 owns a tile of 64 consecutive pixels (linear pixel index, so any
// image width and any row range works) and the tile's 64 x spp samples form
// its pool, ordered item w = sample*64 + pixel_in_tile.  Every loop iteration
// advances each lane's path by ONE segment.  Lanes whose path ended ask for new
// work: a ballot of the asking lanes plus a prefix count (mbcnt) hands them the
// next pool items in lane order, so all 64 lanes stay busy until the pool
// drains.  The assignment depends only on the tile's own paths, never on
// timing, so an image is reproducible run to run and independent of how rows
// are split over GPUs.
//
// A finished sample is added to its pixel's accumulator in LDS (the lane that
// traced a sample is usually not the lane that owns the pixel); when the pool is
// empty the wave divides by spp and writes the tile with coalesced stores.
#pragma once

#include "pt_internal.h"

namespace ptw {

constexpr int WAVES_PER_BLOCK = 4;
constexpr int BLOCK = 64 * WAVES_PER_BLOCK;

__device__ __forceinline__ unsigned lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// Number of set bits of `mask` below this lane.
__device__ __forceinline__ unsigned prefix_count(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// T supplies:  struct Path;
//              static void prologue(const RenderArgs&, Ctx&, uint32_t base, uint32_t lane);   // once per tile
//              static void start(Path&, const RenderArgs&, Ctx&, uint32_t px, uint32_t pixel_index, uint32_t sample);
//              static bool step(Path&, const RenderArgs&, Ctx&);     // true = path continues
//              static void radiance(const Path&, float rgb[3]);
//              static void phase(Ctx&, int);                          // development profiling hook, normally empty
template <class T, class Ctx>
__device__ __forceinline__ void render_tile(const RenderArgs& a, uint32_t tile, float* acc /* LDS, 192 floats of this wave */, Ctx& ctx) {
    const unsigned lane = lane_id();
    const uint32_t base = a.pix_begin + tile * 64u;
    const uint32_t npx = min(64u, a.pix_end - base);
    const uint32_t total = (uint32_t)a.n_samples * 64u;

    acc[lane] = 0.0f; acc[lane + 64] = 0.0f; acc[lane + 128] = 0.0f;
    T::prologue(a, ctx, base, lane);
    // One wave touches this LDS slice, in program order: no barrier needed.

    typename T::Path path;
    uint32_t my_px = 0;
    bool active = false;
    uint32_t next = 0;   // wave-uniform
    for (;;) {
        const unsigned long long want = __ballot(!active);
        if (want != 0ull && next < total) {
            const uint32_t w = next + prefix_count(want);
            next += (uint32_t)__popcll(want);
            if (!active && w < total) {
                const uint32_t px = w & 63u, smp = w >> 6;
                if (px < npx) {
                    my_px = px;
                    T::start(path, a, ctx, px, base + px, smp);
                    active = true;
                }
            }
        }
        T::phase(ctx, 0);
        if (__ballot(active) == 0ull) {
            if (next >= total) break;
            continue;
        }
        if (active) {
            const bool go_on = T::step(path, a, ctx);
            T::phase(ctx, 5);
            if (!go_on) {
                float rgb[3];
                T::radiance(path, rgb);
                __hip_atomic_fetch_add(&acc[my_px * 3 + 0], rgb[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                __hip_atomic_fetch_add(&acc[my_px * 3 + 1], rgb[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                __hip_atomic_fetch_add(&acc[my_px * 3 + 2], rgb[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                active = false;
            }
        }
        T::phase(ctx, 6);
    }

    // p_color = totalLight / n_samples (main.cu:67), then the tile leaves as
    // three fully coalesced 256-byte stores per layout.
    const float ns = (float)a.n_samples;
    const uint32_t nfl = npx * 3u;
    if (a.fb_layout == PT_FB_AOS) {
        float* out = a.fb + (size_t)base * 3u;
        for (uint32_t f = lane; f < nfl; f += 64u) out[f] = acc[f] / ns;
    } else {
        const size_t plane = (size_t)a.max_x * (size_t)a.max_y;
        if (lane < npx) {
            for (int c = 0; c < 3; c++) a.fb[c * plane + base + lane] = acc[lane * 3 + c] / ns;
        }
    }
}

// The same pool, for tracers whose nearest-hit search is a WORKGROUP-wide operation (records staged through LDS
// with barriers inside): every wave of the block calls this, also one whose tile lies beyond the image, and all
// of them keep taking part in the search until the last wave of the block has drained its pool.
// T additionally supplies:  static void idle(Path&);                                  // a ray nothing can hit
//                           static Best nearest_block(const Path&, const RenderArgs&, Ctx&);   // contains barriers
//                           static bool shade(Path&, const Best&, const RenderArgs&, Ctx&);    // true = path continues
template <class T, class Ctx>
__device__ __forceinline__ void render_tile_blocksync(const RenderArgs& a, uint32_t tile, bool tile_valid, float* acc, Ctx& ctx) {
    const unsigned lane = lane_id();
    const uint32_t base = a.pix_begin + (tile_valid ? tile : 0u) * 64u;
    const uint32_t npx = tile_valid ? min(64u, a.pix_end - base) : 0u;
    const uint32_t total = tile_valid ? (uint32_t)a.n_samples * 64u : 0u;

    acc[lane] = 0.0f; acc[lane + 64] = 0.0f; acc[lane + 128] = 0.0f;
    if (tile_valid) T::prologue(a, ctx, base, lane);

    typename T::Path path;
    T::idle(path);
    uint32_t my_px = 0;
    bool active = false;
    uint32_t next = 0;   // wave-uniform
    for (;;) {
        const unsigned long long want = __ballot(!active);
        if (want != 0ull && next < total) {
            const uint32_t w = next + prefix_count(want);
            next += (uint32_t)__popcll(want);
            if (!active && w < total) {
                const uint32_t px = w & 63u, smp = w >> 6;
                if (px < npx) {
                    my_px = px;
                    T::start(path, a, ctx, px, base + px, smp);
                    active = true;
                }
            }
        }
        T::phase(ctx, 0);
        const int wave_busy = (__ballot(active) != 0ull) || (next < total);
        if (!__syncthreads_or(wave_busy)) break;            // block-uniform
        const auto best = T::nearest_block(path, a, ctx);    // all lanes of all waves
        if (active) {
            const bool go_on = T::shade(path, best, a, ctx);
            T::phase(ctx, 5);
            if (!go_on) {
                float rgb[3];
                T::radiance(path, rgb);
                __hip_atomic_fetch_add(&acc[my_px * 3 + 0], rgb[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                __hip_atomic_fetch_add(&acc[my_px * 3 + 1], rgb[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                __hip_atomic_fetch_add(&acc[my_px * 3 + 2], rgb[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                active = false;
                T::idle(path);
            }
        }
        T::phase(ctx, 6);
    }
    if (!tile_valid) return;
    const float ns = (float)a.n_samples;
    const uint32_t nfl = npx * 3u;
    if (a.fb_layout == PT_FB_AOS) {
        float* out = a.fb + (size_t)base * 3u;
        for (uint32_t f = lane; f < nfl; f += 64u) out[f] = acc[f] / ns;
    } else {
        const size_t plane = (size_t)a.max_x * (size_t)a.max_y;
        if (lane < npx) {
            for (int c = 0; c < 3; c++) a.fb[c * plane + base + lane] = acc[lane * 3 + c] / ns;
        }
    }
}

}  // namespace ptw
