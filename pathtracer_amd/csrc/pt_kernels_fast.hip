// Fast-math render and tracePath kernels for gfx950.
//
// Same algorithm, same counter-RNG streams and same branch structure as the
// parity-order kernels, but written for VALU throughput:
//  * primitive records are read through scalar loads (every index is
//    wave-uniform), so they live in SGPRs and cost no VGPRs and no LDS bandwidth;
//    PT_MATH_FAST reads them sorted by kind, one typed loop per kind, which the
//    order-free (distance, index) rule of the tests allows; the other variants
//    walk runs of same-kind primitives in scene order;
//  * planes are tested as t = -(n.o + w) / (n.d) with the bounds check on two
//    more rows of T^-1, planes and cubes along the world axes by coordinates;
//    uniformly scaled spheres are tested in world space; cubes keep the slab
//    test and remember the face axis so the normal needs no second transform;
//  * scenes with many finite primitives are searched through a bounding-volume
//    hierarchy (per-lane walk, stack in LDS);
//  * the hit position is o + d*t in world space; normals come from a per-
//    primitive shading record gathered from LDS for the winner only;
//  * v_rcp/v_rsq/v_sqrt/v_exp and v_sin/v_cos (which take revolutions, so the
//    2*pi multiply of bsdf.h:31 disappears), FMA contraction everywhere.
// Results differ from the parity-order kernels by rounding only.
#include "pt_internal.h"
#include "pt_wavepool.h"
#include "pt_batched.h"

#include <cstdlib>
#include <type_traits>

#define PTD __device__ __forceinline__

// Development aid (make prof): per-phase shader-cycle shares of the render loop, stamped with s_memtime.
// Phases: 0 refill, 1 nearest, 2 surface + material fetch, 3 emissive/Beer/Fresnel/mode, 4 direction, 5 throughput +
// roulette, 6 accumulate, 7 (hierarchy kernels) the loop over the primitives outside the hierarchy.  Not compiled into
// libpt_hip.so.
#ifdef PT_PHASE_PROF
__device__ unsigned long long g_phase_cycles[24];   // [0..7] cycles, [8..15] cycles x active lanes at the stamp, [16..23] stamps
struct PhaseClock {
    unsigned long long last, acc[8], lanes[8], cnt[8];
    __device__ void start() { for (int i = 0; i < 8; i++) { acc[i] = 0; lanes[i] = 0; cnt[i] = 0; } last = __builtin_amdgcn_s_memtime(); }
    __device__ void mark(int i) {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        acc[i] += now - last; lanes[i] += (now - last) * (unsigned long long)__popcll(__ballot(1)); last = now; cnt[i]++;
    }
    __device__ void flush(unsigned lane) {
        if (lane == 0) for (int i = 0; i < 8; i++) { atomicAdd(&g_phase_cycles[i], acc[i]); atomicAdd(&g_phase_cycles[8 + i], lanes[i]); atomicAdd(&g_phase_cycles[16 + i], cnt[i]); }
    }
};
#define PHASE(clk, i) (clk).mark(i)
#elif defined(PT_PHASE_MARK)
// Development aid: the phase boundaries as comments in the compiler's assembly output (tools/isa_phases.py counts the
// instructions between them).  Not compiled into libpt_hip.so.
struct PhaseClock { __device__ void start() {} __device__ void mark(int) {} __device__ void flush(unsigned) {} };
#define PHASE(clk, i) asm volatile("; PHASE %0" :: "i"(i))
#else
struct PhaseClock { __device__ void start() {} __device__ void mark(int) {} __device__ void flush(unsigned) {} };
#define PHASE(clk, i) do { } while (0)
#endif

#ifdef PT_WALK_CENSUS
// Development aid: where the lanes of a wave are during the hierarchy walk.  [0] trips of the node loop, then lanes
// summed over those trips: [1] visiting a node, [2] holding a leaf, [3] finished, [4] about to pop, [5] in the wave;
// [6] leaf phases, [7] lanes testing a leaf in them; per tile of the loop with camera batches: [8] tiles, [9] tiles whose
// candidate list fitted, [10] candidates in those, [11] camera batches, [12] entries they parked, [13] iterations of the
// main loop, [14] lanes with a path in them.  Not compiled into libpt_hip.so.
__device__ unsigned long long g_walk_census[16];
#define CENSUS(i, v) do { const unsigned long long v_ = (unsigned long long)(v); if (__builtin_amdgcn_mbcnt_hi(__builtin_amdgcn_read_exec_hi(), __builtin_amdgcn_mbcnt_lo(__builtin_amdgcn_read_exec_lo(), 0u)) == 0u) atomicAdd(&g_walk_census[i], v_); } while (0)
#else
#define CENSUS(i, v) do { } while (0)
#endif

namespace ptf {

struct V3 { float x, y, z; };
PTD V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
PTD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
PTD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
PTD V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
PTD V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
PTD V3 neg(V3 a) { return v3(-a.x, -a.y, -a.z); }
PTD float dot(V3 a, V3 b) { return __builtin_fmaf(a.x, b.x, __builtin_fmaf(a.y, b.y, a.z * b.z)); }
PTD V3 madd(V3 d, float t, V3 o) { return v3(__builtin_fmaf(d.x, t, o.x), __builtin_fmaf(d.y, t, o.y), __builtin_fmaf(d.z, t, o.z)); }
PTD float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
PTD float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
PTD float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
PTD float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
PTD float max3(float a, float b, float c) { return __builtin_fmaxf(a, __builtin_fmaxf(b, c)); }
PTD float min3(float a, float b, float c) { return __builtin_fminf(a, __builtin_fminf(b, c)); }
// Unit vector.  A zero input gives NaNs where Eigen's normalized() returns the zero vector; for a ray direction
// both mean the same thing one segment later: no primitive test passes, the path ends as a miss.
PTD V3 unit(V3 a) { return a * rsq(dot(a, a)); }
// Unit vector that leaves the zero vector alone (Eigen's normalized()).
PTD V3 unit_or_zero(V3 a) { float n = dot(a, a); return n > 0.0f ? a * rsq(n) : a; }
// Scene data that every lane reads at the same, wave-uniform address is read through pointers to the constant
// address space: such loads are scalar loads (one fetch per wave, operands in SGPRs) whatever else the kernel
// writes to global memory.  The LDS-staged loop passes plain pointers into its LDS image instead.
typedef const float __attribute__((address_space(4)))* KFloatP;
typedef const int32_t __attribute__((address_space(4)))* KIntP;
PTD KFloatP as_constant(const float* p) { return (KFloatP)p; }
PTD KIntP as_constant(const FRun* p) { return (KIntP)p; }

// row . (p, 1)
template <class RP> PTD float row_point(RP r, V3 p) { return __builtin_fmaf(r[0], p.x, __builtin_fmaf(r[1], p.y, __builtin_fmaf(r[2], p.z, r[3]))); }
template <class RP> PTD float row_vec(RP r, V3 p) { return __builtin_fmaf(r[0], p.x, __builtin_fmaf(r[1], p.y, r[2] * p.z)); }

// ---- counter RNG: identical integer streams to the parity-order kernels ------
PTD uint32_t rng_hash(uint32_t x) {
    x ^= x >> 16; x *= 0x21f0aaadu; x ^= x >> 15; x *= 0x735a2d97u; x ^= x >> 15;
    return x;
}
PTD float rng_u(uint32_t key, uint32_t slot) {
    const uint32_t bits = rng_hash(key ^ (0x85EBCA6Bu * (slot + 1u)));
    return __builtin_fmaf((float)bits, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
}

// The two coordinates of a point (pixel jitter; u1, u2 of a direction) come from the 16-bit halves of ONE hash, the
// one of the pair's first slot: cell centres of 65 536 levels each.  One hash and two conversions instead of two
// hashes (a hash is 3 slow-class multiplies among 9 integer instructions).
PTD void rng_pair(uint32_t key, uint32_t first, float& a, float& b) {
    const uint32_t bits = rng_hash(key ^ (0x85EBCA6Bu * (first + 1u)));
    a = __builtin_fmaf((float)(bits >> 16), 1.52587890625e-05f, 7.62939453125e-06f);
    b = __builtin_fmaf((float)(bits & 0xffffu), 1.52587890625e-05f, 7.62939453125e-06f);
}

constexpr float EPS = 1e-5f;   // SMALL_NUMBER (vec_math.h:7) as the float predicate t <= 1e-5

struct Best {
    float t;      // +inf = miss
    int id;       // (primitive index << 3) | code; code: cube = face axis 0..2 (+4 = exit face),
                  // cylinder = surface 0 body, 1 disc +z, 2 disc -z
};

// Every test below ends the same way: the candidate replaces the best hit when t > EPS (plus its own validity mask)
// and (t, id) is below (best.t, best.id) in dictionary order.  Tested in scene order that is the strict '<' of
// getNearestIntersection (shapes.h:93), which keeps the lowest primitive index on ties; written this way the result
// does not depend on the order of the tests.  Positive floats order like their bit patterns; a negative or NaN t has
// already failed t > EPS.
PTD bool nearer(float t, int id, const Best& b) {
    const unsigned long long kc = ((unsigned long long)__float_as_uint(t) << 32) | (unsigned)id;
    const unsigned long long kb = ((unsigned long long)__float_as_uint(b.t) << 32) | (unsigned)b.id;
    return kc < kb;
}

// Plane with the bounds folded into rows x', y' (record: z row | x' row | y' row).
template <class RP> PTD void test_plane(RP R, V3 o, V3 d, int id, Best& b) {
    const float oz = row_point(R, o);
    const float dz = row_vec(R, d);
    const float t = -oz * rcp(dz);                 // dz == 0 gives inf/nan: every compare below fails
    const V3 p = madd(d, t, o);
    const float xl = row_point(R + 4, p), yl = row_point(R + 8, p);
    const bool ok = (t > EPS) & nearer(t, id, b) & (__builtin_fabsf(xl) < 1.0f) & (__builtin_fabsf(yl) < 1.0f);
    b.t = ok ? t : b.t;
    b.id = ok ? id : b.id;
}

// Plane with explicit bounds (record: rows z, x, y | xMin, yMin, xMax, yMax).
template <class RP> PTD void test_plane_gen(RP R, V3 o, V3 d, int id, Best& b) {
    const float oz = row_point(R, o);
    const float dz = row_vec(R, d);
    const float t = -oz * rcp(dz);
    const V3 p = madd(d, t, o);
    const float xl = row_point(R + 4, p), yl = row_point(R + 8, p);
    const bool ok = (t > EPS) & nearer(t, id, b) & (xl > R[12]) & (yl > R[13]) & (xl < R[14]) & (yl < R[15]);
    b.t = ok ? t : b.t;
    b.id = ok ? id : b.id;
}

// Plane with its normal along world axis N and its edges along the other two (record: plane coordinate | centre and
// half-width along the lower of the two axes | the same along the higher).  inv = 1/d.
template <int N, class RP> PTD void test_plane_axis(RP R, V3 o, V3 d, V3 inv, int id, Best& b) {
    const float on = N == 0 ? o.x : (N == 1 ? o.y : o.z), in = N == 0 ? inv.x : (N == 1 ? inv.y : inv.z);
    const float ou = N == 0 ? o.y : o.x, du = N == 0 ? d.y : d.x;
    const float ov = N == 2 ? o.y : o.z, dv = N == 2 ? d.y : d.z;
    const float t = (R[0] - on) * in;              // d_n == 0 gives inf/nan: a compare below fails
    const float pu = __builtin_fmaf(du, t, ou) - R[1], pv = __builtin_fmaf(dv, t, ov) - R[3];
    const bool ok = (t > EPS) & nearer(t, id, b) & (__builtin_fabsf(pu) < R[2]) & (__builtin_fabsf(pv) < R[4]);
    b.t = ok ? t : b.t;
    b.id = ok ? id : b.id;
}

// A product rounded on its own: the empty asm hides it from the compiler, which under -ffp-contract=fast would fuse
// it into the sum that follows (also inside `#pragma clang fp contract(off)` and through __fmul_rn / __fadd_rn).
PTD float mul_rn(float a, float b) { float p = a * b; asm("" : "+v"(p)); return p; }
// Eigen's x + (y + z) with every product and sum rounded, as the reference evaluates a.dot(b)
PTD float dot_rn(V3 a, V3 b) { return mul_rn(a.x, b.x) + (mul_rn(a.y, b.y) + mul_rn(a.z, b.z)); }
template <class RP> PTD float row_vec_rn(RP r, V3 p) { return mul_rn(r[0], p.x) + (mul_rn(r[1], p.y) + mul_rn(r[2], p.z)); }

// Quadratic a t^2 + 2 h t + c = 0 with ia = 1/a: the root Shape::getIntersections
// + getNearestIntersection keep is the smaller one if it is beyond EPS, else the
// larger.  A negative discriminant makes s NaN and every compare false.
// ROUNDED: h*h and a*c are rounded before they are subtracted, as the reference's b*b - 4*a*c is (shapes.h:331; h = b/2
// scales both products by an exact 4).  For a small shape far away the two products agree to all but a few bits, and
// rounded separately they come out EQUAL for a whole band of rays that just miss: the discriminant is then exactly 0,
// which counts as a (tangent) hit.  The reference's small far spheres are in effect a little larger than their
// radius, and a fused h*h - a*c, which has no such band, rendered the 10 000-shape scene 0.12 % brighter than the
// reference's arithmetic does (measured over 1024 seeds, tools/bias_study.py).  The general kinds, which such shapes
// get, therefore round as the reference does; the short sphere record (spheres within 32 diameters of the origin,
// where the band is a 1e-4 of the cross-section) keeps the fused form.
template <bool ROUNDED = false>
PTD void take_quadratic(float a, float ia, float h, float c, int id, Best& b) {
    const float disc = ROUNDED ? mul_rn(h, h) - mul_rn(a, c) : __builtin_fmaf(h, h, -a * c);
#ifndef PT_NO_QUADRATIC_SKIP          // (measurement aid)
    // most waves miss a small shape with every lane: the roots, the choice between them and the compare are then skipped
    if (__ballot(disc >= 0.0f) == 0ull) return;
#endif
    const float s = fsqrt(disc);
    const float t1 = (-h - s) * ia, t2 = (s - h) * ia;
    const float t = t1 > EPS ? t1 : t2;
    const bool ok = (t > EPS) & nearer(t, id, b);
    b.t = ok ? t : b.t;
    b.id = ok ? id : b.id;
}

// Sphere under L = s * I (uniform scale, no rotation).  The reference's local frame (shapes.h:42-48,327-343) is then
// o_l = k*o + q and d_l = k*d with k = 1/s, q = -centre/s (record: k | q | k^2 | 1/k^2), so no matrix is needed — but
// the test is still made in that frame, product and sum rounded as the reference rounds them: for a small sphere far
// from the origin o_l is the difference of two numbers of magnitude |centre|/s, and it is that cancellation which
// decides whether a ray leaving the sphere at a grazing angle finds it again beyond the 1e-5 filter.  (Tested in
// world space instead — |o - centre|^2 - r^2 — such rays are lost less often than in the reference: a measurable
// bias on the rims of small bright spheres.)  dd = d.d and idd = 1/dd are the caller's.
template <class RP> PTD void test_sphere_ws(RP R, V3 o, V3 d, float dd, float idd, int id, Best& b) {
    const float k = R[0];
    // (the empty asm keeps product and sum apart: under -ffp-contract=fast the compiler fuses them even inside
    // `#pragma clang fp contract(off)` and through __fmul_rn / __fadd_rn)
    float mx = o.x * k, my = o.y * k, mz = o.z * k;
    asm volatile("" : "+v"(mx), "+v"(my), "+v"(mz));
    const V3 ol = v3(mx + R[1], my + R[2], mz + R[3]);
    take_quadratic(R[4] * dd, R[5] * idd, k * dot(d, ol), dot(ol, ol) - 0.25f, id, b);
}

// Sphere under any transform (record: rows of T^-1), in the reference's local frame and with its rounding
// (shapes.h:42-48,328-331: every product and sum rounded, sums as x + (y + z)): see take_quadratic.
template <class RP> PTD void test_sphere(RP R, V3 o, V3 d, int id, Best& b) {
    const V3 ol = v3(row_vec_rn(R, o) + R[3], row_vec_rn(R + 4, o) + R[7], row_vec_rn(R + 8, o) + R[11]);
    const V3 dl = v3(row_vec_rn(R, d), row_vec_rn(R + 4, d), row_vec_rn(R + 8, d));
    const float a = dot_rn(dl, dl);
    take_quadratic<true>(a, rcp(a), dot_rn(dl, ol), dot_rn(ol, ol) - 0.25f, id, b);
}

template <class RP> PTD void test_cube(RP R, V3 o, V3 d, int id, Best& b) {
    const V3 ol = v3(row_point(R, o), row_point(R + 4, o), row_point(R + 8, o));
    const V3 dl = v3(row_vec(R, d), row_vec(R + 4, d), row_vec(R + 8, d));
    // 1/d kept finite: with an infinite reciprocal (a direction component of exactly 0) the entry / exit distances
    // below would be inf - inf = NaN, which v_min / v_max drop, and a ray running beside the cube would hit it.  With
    // +-1e30 instead they come out as the reference's (-0.5 - o)/0 and (0.5 - o)/0 do (shapes.h:172-177): both far
    // negative or both far positive outside the slab, one of each inside it.
    const float ix = __builtin_amdgcn_fmed3f(rcp(dl.x), -1e30f, 1e30f), iy = __builtin_amdgcn_fmed3f(rcp(dl.y), -1e30f, 1e30f),
                iz = __builtin_amdgcn_fmed3f(rcp(dl.z), -1e30f, 1e30f);
    const float hx = 0.5f * __builtin_fabsf(ix), hy = 0.5f * __builtin_fabsf(iy), hz = 0.5f * __builtin_fabsf(iz);
    // entry / exit distance per slab
    const float nx = __builtin_fmaf(-ol.x, ix, -hx), ny = __builtin_fmaf(-ol.y, iy, -hy), nz = __builtin_fmaf(-ol.z, iz, -hz);
    const float fx = __builtin_fmaf(-ol.x, ix, hx), fy = __builtin_fmaf(-ol.y, iy, hy), fz = __builtin_fmaf(-ol.z, iz, hz);
    const float tmin = max3(nx, ny, nz), tmax = min3(fx, fy, fz);
    const bool use_exit = !(tmin > EPS);
    const float t = use_exit ? tmax : tmin;
    // tmax >= 0 (shapes.h:191) is implied by t > EPS and tmin <= tmax.
    const bool ok = (tmin <= tmax) & (t > EPS) & nearer(t, id, b);
    if (ok) {
        const float ax = use_exit ? fx : nx, ay = use_exit ? fy : ny;
        const int axis = (ax == t) ? 0 : ((ay == t) ? 1 : 2);
        b.t = t;
        b.id = id | axis | (use_exit ? 4 : 0);
    }
}

// Cube under a diagonal transform, as its world-space box (record: low corner | high corner); inv = 1/d, infinite
// where a direction component is 0: both planes of that slab are then at -inf, at +inf, or one each, as in
// shapes.h:172-177.  Codes as test_cube.
template <class RP> PTD void test_cube_box(RP R, V3 o, V3 inv, int id, Best& b) {
    const float x0 = (R[0] - o.x) * inv.x, x1 = (R[3] - o.x) * inv.x;
    const float y0 = (R[1] - o.y) * inv.y, y1 = (R[4] - o.y) * inv.y;
    const float z0 = (R[2] - o.z) * inv.z, z1 = (R[5] - o.z) * inv.z;
    const float nx = __builtin_fminf(x0, x1), ny = __builtin_fminf(y0, y1), nz = __builtin_fminf(z0, z1);
    const float fx = __builtin_fmaxf(x0, x1), fy = __builtin_fmaxf(y0, y1), fz = __builtin_fmaxf(z0, z1);
    const float tmin = max3(nx, ny, nz), tmax = min3(fx, fy, fz);
    const bool use_exit = !(tmin > EPS);
    const float t = use_exit ? tmax : tmin;
    const bool ok = (tmin <= tmax) & (t > EPS) & nearer(t, id, b);
    if (ok) {
        const float ax = use_exit ? fx : nx, ay = use_exit ? fy : ny;
        const int axis = (ax == t) ? 0 : ((ay == t) ? 1 : 2);
        b.t = t;
        b.id = id | axis | (use_exit ? 4 : 0);
    }
}

// Cylinder: the case analysis of shapes.h:231-318 in local space.
template <class RP> PTD void test_cylinder(RP R, V3 o, V3 d, int id, Best& b) {
    const V3 ol = v3(row_point(R, o), row_point(R + 4, o), row_point(R + 8, o));
    const V3 dl = v3(row_vec(R, d), row_vec(R + 4, d), row_vec(R + 8, d));
    // (products and sums rounded as shapes.h:258-261 rounds them: see take_quadratic)
    const float a = mul_rn(dl.x, dl.x) + mul_rn(dl.y, dl.y);
    const float h = mul_rn(dl.x, ol.x) + mul_rn(dl.y, ol.y);
    const float c = (mul_rn(ol.x, ol.x) + mul_rn(ol.y, ol.y)) - 0.25f;
    const float disc = mul_rn(h, h) - mul_rn(a, c);
    const float iz = rcp(dl.z);
    const float td1 = (0.5f - ol.z) * iz, td2 = (-0.5f - ol.z) * iz;
    bool d1 = false, d2 = false;
    if (dl.z != 0.0f) {
        const float x1 = __builtin_fmaf(dl.x, td1, ol.x), y1 = __builtin_fmaf(dl.y, td1, ol.y);
        const float x2 = __builtin_fmaf(dl.x, td2, ol.x), y2 = __builtin_fmaf(dl.y, td2, ol.y);
        d1 = !(__builtin_fmaf(x1, x1, y1 * y1) > 0.25f);
        d2 = !(__builtin_fmaf(x2, x2, y2 * y2) > 0.25f);
    }
    bool c1 = false, c2 = false;
    float tc1 = 0.0f, tc2 = 0.0f;
    if (disc > 0.0f) {
        const float s = fsqrt(disc), ia = rcp(a);
        tc1 = (-h - s) * ia; tc2 = (s - h) * ia;
        const float z1 = __builtin_fmaf(dl.z, tc1, ol.z), z2 = __builtin_fmaf(dl.z, tc2, ol.z);
        c1 = (-0.5f < z1) & (z1 < 0.5f);
        c2 = (-0.5f < z2) & (z2 < 0.5f);
    }
    int n = 0; bool h1 = false, h2 = false; float t1 = 0.0f, t2 = 0.0f; int s1 = 0, s2 = 0;
    if (d1 && d2) { n = 2; h1 = true; t1 = td1; s1 = 1; h2 = true; t2 = td2; s2 = 2; }
    else if (d1 && disc > 0.0f) { n = 2; h1 = true; t1 = td1; s1 = 1; h2 = c1 ? c1 : c2; t2 = c1 ? tc1 : tc2; s2 = 0; }
    else if (d2 && disc > 0.0f) { n = 2; h1 = true; t1 = td2; s1 = 2; h2 = c1 ? c1 : c2; t2 = c1 ? tc1 : tc2; s2 = 0; }
    else if (disc > 0.0f && c1 && c2) { n = 2; h1 = true; t1 = tc1; s1 = 0; h2 = true; t2 = tc2; s2 = 0; }
    if (n == 2) {
        const bool e1 = t1 <= EPS, e2 = t2 <= EPS;
        bool valid; float t; int s;
        if (e1 && e2) { valid = false; t = 0.0f; s = 0; }
        else if (e1) { valid = h2; t = t2; s = s2; }
        else if (e2) { valid = h1; t = t1; s = s1; }
        else { const bool sec = t2 < t1; valid = sec ? h2 : h1; t = sec ? t2 : t1; s = sec ? s2 : s1; }
        if (valid && nearer(t, id, b)) { b.t = t; b.id = id | s; }
    }
}

// The loop of getNearestIntersection (shapes.h:82-98) over runs [r0, r1) of same-kind primitives, in scene order.
// Record k of a run lives at base + run.offset + k*len; every index is wave-uniform.  `base` is the float array in
// global memory (records then arrive by scalar loads) or an LDS image of one chunk of it.
template <class RP>
PTD void scan_runs(V3 o, V3 d, float dd, float idd, RP base, KIntP runs, int r0, int r1, Best& b) {
    const V3 inv = v3(rcp(d.x), rcp(d.y), rcp(d.z));
    for (int r = r0; r < r1; r++) {
        FRun run;
        run.type = runs[4 * r]; run.first = runs[4 * r + 1]; run.count = runs[4 * r + 2]; run.offset = runs[4 * r + 3];
        RP R = base + run.offset;
        int id = run.first << 3;
        switch (run.type) {
            case FP_PLANE: {
                // Two planes per trip: one wait covers both records' scalar loads.  Measured 4.02 -> 3.97 ms; three
                // and more per trip are slower (4.14, 4.20, 4.91 ms for 3, 4, 7: the records no longer fit the SGPRs).
                int i = 0;
                for (; i + 2 <= run.count; i += 2, R += 24, id += 16) { test_plane(R, o, d, id, b); test_plane(R + 12, o, d, id + 8, b); }
                for (; i < run.count; i++, R += 12, id += 8) test_plane(R, o, d, id, b);
                break;
            }
            case FP_SPHERE_WS:
                for (int i = 0; i < run.count; i++, R += 6, id += 8) test_sphere_ws(R, o, d, dd, idd, id, b);
                break;
            case FP_CUBE:
                for (int i = 0; i < run.count; i++, R += 12, id += 8) test_cube(R, o, d, id, b);
                break;
            case FP_SPHERE:
                for (int i = 0; i < run.count; i++, R += 12, id += 8) test_sphere(R, o, d, id, b);
                break;
            case FP_PLANE_GEN:
                for (int i = 0; i < run.count; i++, R += 16, id += 8) test_plane_gen(R, o, d, id, b);
                break;
            case FP_CUBE_AA:
                for (int i = 0; i < run.count; i++, R += 6, id += 8) test_cube_box(R, o, inv, id, b);
                break;
            case FP_PLANE_X:
                for (int i = 0; i < run.count; i++, R += 5, id += 8) test_plane_axis<0>(R, o, d, inv, id, b);
                break;
            case FP_PLANE_Y:
                for (int i = 0; i < run.count; i++, R += 5, id += 8) test_plane_axis<1>(R, o, d, inv, id, b);
                break;
            case FP_PLANE_Z:
                for (int i = 0; i < run.count; i++, R += 5, id += 8) test_plane_axis<2>(R, o, d, inv, id, b);
                break;
            default:
                for (int i = 0; i < run.count; i++, R += 12, id += 8) test_cylinder(R, o, d, id, b);
                break;
        }
    }
}

// Axis-aligned planes of one kind, two per trip (records of 8 floats: five of data, the scene index << 3, padding).
template <int N> PTD void axis_planes(KFloatP R, int n, V3 o, V3 d, V3 inv, Best& b) {
    int i = 0;
    for (; i + 2 <= n; i += 2, R += 16) {
        test_plane_axis<N>(R, o, d, inv, __float_as_int(R[5]), b);
        test_plane_axis<N>(R + 8, o, d, inv, __float_as_int(R[13]), b);
    }
    if (i < n) test_plane_axis<N>(R, o, d, inv, __float_as_int(R[5]), b);
}

// The loop of getNearestIntersection over the kind-sorted records: one typed loop per kind, offsets and counts from
// the kernel arguments (no run table to fetch, no switch).  Every index is wave-uniform, so records arrive by scalar
// loads; a record's scene index << 3 sits right behind it.  Planes go two per trip: one wait covers both records.
//
// KM: the kinds the caller knows the table can hold (bit FP_*); a loop whose kind is not in it is not compiled, and
// its offset and count are never read.  KM_SHORT is the table of a scene whose planes and cubes lie along the world
// axes and whose spheres are uniformly scaled.
constexpr int KM_ALL = (1 << FP_KINDS) - 1;
constexpr int KM_PLANES = (1 << FP_PLANE) | (1 << FP_PLANE_GEN) | (1 << FP_PLANE_X) | (1 << FP_PLANE_Y) | (1 << FP_PLANE_Z);   // what stays outside a hierarchy
constexpr int KM_GENERAL = (1 << FP_PLANE) | (1 << FP_CUBE) | (1 << FP_CYLINDER) | (1 << FP_SPHERE) | (1 << FP_PLANE_GEN);   // no short kind
constexpr int KM_SHORT = (1 << FP_SPHERE_WS) | (1 << FP_PLANE_X) | (1 << FP_PLANE_Y) | (1 << FP_PLANE_Z) | (1 << FP_CUBE_AA);
// The offsets and counts of the kinds.  From the kernel arguments they are loop invariants that the compiler keeps in
// scalar registers for the whole kernel (twenty of them, plus a hoisted "count > 0" mask each): with every kind
// compiled in, that costs 80 spilled scalars per loop iteration.  LAZY reads them from the copy at the head of the
// record buffer instead, behind a pointer the compiler cannot see through, so they live only while a scan runs.
#ifndef PT_LAZY_HEADER
#define PT_LAZY_HEADER(KM) ((KM) == KM_ALL)
#endif
template <int KM>
PTD void sorted_header(const FSorted& S, KFloatP base, int (&off)[FP_KINDS], int (&cnt)[FP_KINDS]) {
    if (PT_LAZY_HEADER(KM)) {
        KFloatP h = base;
        asm volatile("" : "+s"(h));
        for (int k = 0; k < FP_KINDS; k++) { off[k] = __float_as_int(h[k]); cnt[k] = __float_as_int(h[FP_KINDS + k]); }
    } else {
        for (int k = 0; k < FP_KINDS; k++) { off[k] = S.off[k]; cnt[k] = S.count[k]; }
    }
}

template <int KM = KM_ALL>
PTD void scan_sorted(V3 o, V3 d, float dd, float idd, const FSorted& S0, Best& b) {
    const KFloatP base = as_constant(S0.data);
    struct { int off[FP_KINDS], count[FP_KINDS]; } S;
    sorted_header<KM>(S0, base, S.off, S.count);
    constexpr int AXIS = (1 << FP_PLANE_X) | (1 << FP_PLANE_Y) | (1 << FP_PLANE_Z) | (1 << FP_CUBE_AA);
    if ((KM & AXIS) != 0 && (KM == KM_SHORT || S.count[FP_PLANE_X] + S.count[FP_PLANE_Y] + S.count[FP_PLANE_Z] + S.count[FP_CUBE_AA] > 0)) {
        const V3 inv = v3(rcp(d.x), rcp(d.y), rcp(d.z));
        if (KM & (1 << FP_PLANE_X)) axis_planes<0>(base + S.off[FP_PLANE_X], S.count[FP_PLANE_X], o, d, inv, b);
        if (KM & (1 << FP_PLANE_Y)) axis_planes<1>(base + S.off[FP_PLANE_Y], S.count[FP_PLANE_Y], o, d, inv, b);
        if (KM & (1 << FP_PLANE_Z)) axis_planes<2>(base + S.off[FP_PLANE_Z], S.count[FP_PLANE_Z], o, d, inv, b);
        if (KM & (1 << FP_CUBE_AA)) {
            KFloatP R = base + S.off[FP_CUBE_AA];
            for (int i = S.count[FP_CUBE_AA]; i > 0; i--, R += 8) test_cube_box(R, o, inv, __float_as_int(R[6]), b);
        }
    }
    if (KM & (1 << FP_PLANE)) {
        KFloatP R = base + S.off[FP_PLANE];
        int i = 0;
        const int n = S.count[FP_PLANE];
        for (; i + 2 <= n; i += 2, R += 32) {
            test_plane(R, o, d, __float_as_int(R[12]), b);
            test_plane(R + 16, o, d, __float_as_int(R[28]), b);
        }
        if (i < n) test_plane(R, o, d, __float_as_int(R[12]), b);
    }
    if (KM & (1 << FP_PLANE_GEN)) {
        KFloatP R = base + S.off[FP_PLANE_GEN];
        for (int i = S.count[FP_PLANE_GEN]; i > 0; i--, R += 20) test_plane_gen(R, o, d, __float_as_int(R[16]), b);
    }
    if (KM & (1 << FP_SPHERE_WS)) {
        KFloatP R = base + S.off[FP_SPHERE_WS];
        for (int i = S.count[FP_SPHERE_WS]; i > 0; i--, R += 8) test_sphere_ws(R, o, d, dd, idd, __float_as_int(R[6]), b);
    }
    if (KM & (1 << FP_SPHERE)) {
        KFloatP R = base + S.off[FP_SPHERE];
        for (int i = S.count[FP_SPHERE]; i > 0; i--, R += 16) test_sphere(R, o, d, __float_as_int(R[12]), b);
    }
    if (KM & (1 << FP_CUBE)) {
        KFloatP R = base + S.off[FP_CUBE];
        for (int i = S.count[FP_CUBE]; i > 0; i--, R += 16) test_cube(R, o, d, __float_as_int(R[12]), b);
    }
    if (KM & (1 << FP_CYLINDER)) {
        KFloatP R = base + S.off[FP_CYLINDER];
        for (int i = S.count[FP_CYLINDER]; i > 0; i--, R += 16) test_cylinder(R, o, d, __float_as_int(R[12]), b);
    }
}

PTD Best best_none() { Best b; b.t = __builtin_inff(); b.id = -8; return b; }

// The finite primitives through their bounding-volume hierarchy, per lane.  `b` comes in holding the best hit
// among the primitives outside the hierarchy.  The result is the one the loop over all primitives gives: every
// primitive whose own test would accept lies in boxes the ray enters no later than the best hit so far, and of two
// hits at the same distance the lower scene index wins, as the strict '<' of shapes.h:93 has it in scene order.
// The walk is resumable: it starts at node `n` (0 = the root), visits at most `budget` nodes and leaves the node
// to go on from in `n` (n >= n_nodes: finished).
PTD void walk_hierarchy(V3 o, V3 d, float dd, float idd, const BNode* __restrict__ nodes, int n_nodes,
                        const BPrim* __restrict__ prims, Best& b, int& n, int budget) {
    const V3 inv = v3(rcp(d.x), rcp(d.y), rcp(d.z));
    for (;;) {
        // Walk until this lane stands on a leaf its ray enters (or has no node left).  The lanes of a wave leave this
        // loop together, so the primitive tests below run with most lanes on, not for one lane at a time.
        int leaf = -1;
        while (n < n_nodes && budget > 0) {
            budget--;
            const float4 A = reinterpret_cast<const float4*>(nodes)[2 * n];
            const float4 B = reinterpret_cast<const float4*>(nodes)[2 * n + 1];
            // slab test; v_min/v_max drop a NaN operand (0 * inf when the origin lies on a slab plane of an axis
            // the ray does not move along), which leaves that axis unconstrained on one side: conservative
            const float ax = (A.x - o.x) * inv.x, bx = (B.x - o.x) * inv.x;
            const float ay = (A.y - o.y) * inv.y, by = (B.y - o.y) * inv.y;
            const float az = (A.z - o.z) * inv.z, bz = (B.z - o.z) * inv.z;
            const float tn = max3(__builtin_fminf(ax, bx), __builtin_fminf(ay, by), __builtin_fmaxf(__builtin_fminf(az, bz), 0.0f));
            const float tf = min3(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by), __builtin_fminf(__builtin_fmaxf(az, bz), b.t));
            const bool hit = tn <= tf;
            const int skip = __float_as_int(A.w), lf = __float_as_int(B.w);
            n = (hit && lf < 0) ? n + 1 : skip;
            if (hit && lf >= 0) { leaf = lf; break; }
        }
        if (leaf < 0) break;
        const BPrim* q = prims + (leaf >> 4);
        for (int k = leaf & 15; k > 0; k--, q++) {
            const float* R = q->rec;
            const int id = q->id;
            switch (q->kind) {
                case FP_SPHERE_WS: test_sphere_ws(R, o, d, dd, idd, id, b); break;
                case FP_CUBE: test_cube(R, o, d, id, b); break;
                case FP_CUBE_AA: test_cube_box(R, o, inv, id, b); break;
                case FP_SPHERE: test_sphere(R, o, d, id, b); break;
                default: test_cylinder(R, o, d, id, b); break;
            }
        }
    }
}

// The same search through the fat form of the tree (both children's boxes in the parent), nearer child first, the
// other one parked on a per-lane stack in LDS (`stack`: this lane's column, entries 64 words apart).  A parked
// entry keeps the top 12 bits of its box's entry distance above the 20-bit reference, so a subtree that a hit found
// in the meantime has put out of reach is dropped when it comes off the stack without touching memory.  `cur` is
// the reference in hand (WALK_NONE: take the next one off the stack), `sp` the stack height; the search is over
// when cur == WALK_NONE and sp == 0.  Same result as walk_hierarchy: the order of the visits does not matter to
// the (distance, scene index) rule the leaves apply.
// The control flow is the wave's, not the lane's: every trip of the node loop is, for each lane, an optional pop
// followed by an optional visit, and the wave leaves the loop when no lane can go on, when `leaf_gate` lanes hold a
// leaf, or after `budget` trips; then the lanes that hold a leaf test it together, its whole record fetched at once.
PTD void walk_fat(V3 o, V3 d, float dd, float idd, const BFat* __restrict__ fat, V3 origin, V3 step,
                  const BPrim* __restrict__ prims, Best& b, int& cur, int& sp, uint32_t* stack, int budget, int leaf_gate) {
    // 1/d kept finite (+-1e30): with an infinite reciprocal, plane*inv - o*inv is inf - inf = NaN for every box on the
    // origin's side of an axis the ray does not move along, v_min/v_max drop the NaN, and that axis never culls — the
    // walk stays correct but an axis-parallel ray would descend most of the tree
    const V3 inv = v3(__builtin_amdgcn_fmed3f(rcp(d.x), -1e30f, 1e30f), __builtin_amdgcn_fmed3f(rcp(d.y), -1e30f, 1e30f),
                      __builtin_amdgcn_fmed3f(rcp(d.z), -1e30f, 1e30f));
    // a box between the grid coordinates (sum - diff) and (sum + diff) is met from sum * ga + gb - diff * |ga| to
    // sum * ga + gb + diff * |ga|: (origin + step * q - o) * inv = q * ga + gb
    const V3 ga = step * inv, gb = (origin - o) * inv;
    const V3 gh = v3(__builtin_fabsf(ga.x), __builtin_fabsf(ga.y), __builtin_fabsf(ga.z));
    for (;;) {
        for (; budget > 0; budget--) {
            budget = __builtin_amdgcn_readfirstlane(budget);
            if (cur == WALK_NONE && sp != 0) {
                const uint32_t e = stack[(--sp) * 64];
                const int ref = (int)(e << 12) >> 12;       // the low 20 bits, sign-extended: a leaf is ~index
                cur = __uint_as_float(e & 0xFFF00000u) > b.t ? WALK_NONE : ref;
            }
            const bool inner = (uint32_t)cur < (uint32_t)WALK_NONE;
            const bool pending = cur == WALK_NONE && sp != 0;
            if (__ballot(inner || pending) == 0ull || __popcll(__ballot(cur < 0)) >= leaf_gate) break;
            CENSUS(0, 1); CENSUS(1, __popcll(__ballot(inner))); CENSUS(2, __popcll(__ballot(cur < 0)));
            CENSUS(3, __popcll(__ballot(cur == WALK_NONE && sp == 0))); CENSUS(4, __popcll(__ballot(pending))); CENSUS(5, __popcll(__ballot(true)));
            if (inner) {
                const uint4* N = reinterpret_cast<const uint4*>(fat + cur);
                const uint4 W0 = N[0], W1 = N[1];
                // slab tests of both boxes (inv is finite, see above; an origin exactly on a plane of an axis the ray
                // does not move along gives 0 * 1e30 = 0: inside that slab)
#define PT_GRID_LO(w) ((float)((w) & 0xFFFFu))
#define PT_GRID_HI(w) ((float)((w) >> 16))
                const float amx = __builtin_fmaf(PT_GRID_LO(W0.x), ga.x, gb.x), amy = __builtin_fmaf(PT_GRID_HI(W0.x), ga.y, gb.y);
                const float amz = __builtin_fmaf(PT_GRID_LO(W0.y), ga.z, gb.z), ahx = PT_GRID_HI(W0.y) * gh.x;
                const float ahy = PT_GRID_LO(W0.z) * gh.y, ahz = PT_GRID_HI(W0.z) * gh.z;
                const float bmx = __builtin_fmaf(PT_GRID_LO(W0.w), ga.x, gb.x), bmy = __builtin_fmaf(PT_GRID_HI(W0.w), ga.y, gb.y);
                const float bmz = __builtin_fmaf(PT_GRID_LO(W1.x), ga.z, gb.z), bhx = PT_GRID_HI(W1.x) * gh.x;
                const float bhy = PT_GRID_LO(W1.y) * gh.y, bhz = PT_GRID_HI(W1.y) * gh.z;
#undef PT_GRID_LO
#undef PT_GRID_HI
                const float tna = max3(amx - ahx, amy - ahy, __builtin_fmaxf(amz - ahz, 0.0f));
                const float tfa = min3(amx + ahx, amy + ahy, __builtin_fminf(amz + ahz, b.t));
                const float tnb = max3(bmx - bhx, bmy - bhy, __builtin_fmaxf(bmz - bhz, 0.0f));
                const float tfb = min3(bmx + bhx, bmy + bhy, __builtin_fminf(bmz + bhz, b.t));
                const bool ha = tna <= tfa, hb = tnb <= tfb;
                const int ra = (int)W1.z, rb = (int)W1.w;
                const bool a_first = tna <= tnb;
                const int far = a_first ? rb : ra, near = a_first ? ra : rb;
                const float tfar = a_first ? tnb : tna;
                if (ha & hb) {
                    stack[sp * 64] = (__float_as_uint(tfar) & 0xFFF00000u) | ((uint32_t)far & 0xFFFFFu);
                    sp++;
                }
                cur = (ha & hb) ? near : (ha ? ra : (hb ? rb : WALK_NONE));
            }
        }
        const bool leaf = cur < 0;
        if (__ballot(leaf) == 0ull) break;            // out of trips, or every lane's search is over
        CENSUS(6, 1); CENSUS(7, __popcll(__ballot(leaf)));
        if (leaf) {
            const float4* Q = reinterpret_cast<const float4*>(prims + ~cur);
            const float4 q0 = Q[0], q1 = Q[1], q2 = Q[2], q3 = Q[3];
            const float R[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
            const int id = __float_as_int(q3.y);
            cur = WALK_NONE;
            switch (__float_as_int(q3.x)) {
                case FP_SPHERE_WS: test_sphere_ws(R, o, d, dd, idd, id, b); break;
                case FP_CUBE: test_cube(R, o, d, id, b); break;
                case FP_CUBE_AA: test_cube_box(R, o, inv, id, b); break;
                case FP_SPHERE: test_sphere(R, o, d, id, b); break;
                default: test_cylinder(R, o, d, id, b); break;
            }
        }
    }
}

struct Surf {
    V3 pos, nrm;     // world position; unit normal facing against the ray
    bool inside;
    int material;
};

// The winner's record: position, facing normal and the inside flag of
// shapes.h:93-97, from the per-primitive shading record.
// `pos` is the hit position o + d*t.
PTD Surf surface_at(const Best& b, V3 pos, V3 d, const FShade* shade) {
    Surf s;
    const FShade& S = shade[b.id >> 3];
    const int code = b.id & 7;
    s.material = S.material;
    s.pos = pos;
    const int type = S.type;
    V3 n;
    bool inside_known = false, inside = false;
    if (type == PT_CYLINDER) {
        if (code == 0) {
            const float x = row_point(S.a + 9, s.pos), y = row_point(S.a + 13, s.pos);
            n = unit_or_zero(v3(__builtin_fmaf(S.a[0], x, S.a[3] * y), __builtin_fmaf(S.a[1], x, S.a[4] * y), __builtin_fmaf(S.a[2], x, S.a[5] * y)));
        } else {
            const float sg = code == 1 ? 1.0f : -1.0f;
            n = v3(S.a[6] * sg, S.a[7] * sg, S.a[8] * sg);
        }
    } else {
        const int off = type == PT_CUBE ? (code & 3) * 3 : 0;
        n = v3(S.a[off], S.a[off + 1], S.a[off + 2]);
        if (type == PT_SPHERE) n = unit(s.pos - n);
        if (type == PT_PLANE) { inside_known = true; inside = false; }
        if (type == PT_CUBE) { inside_known = true; inside = (code & 4) != 0; }
    }
    const bool along = dot(n, d) > 0.0f;
    s.inside = inside_known ? inside : along;
    s.nrm = along ? neg(n) : n;
    return s;
}
PTD Surf surface(const Best& b, V3 o, V3 d, const FShade* shade) { return surface_at(b, madd(d, b.t, o), d, shade); }

PTD V3 sphere_point(float u1, float u2) {            // bsdf.h:29-36; v_sin/v_cos take revolutions
    const float z = __builtin_fmaf(u1, 2.0f, -1.0f);
    const float r = fsqrt(__builtin_fmaxf(__builtin_fmaf(-z, z, 1.0f), 0.0f));
    return v3(r * __builtin_amdgcn_cosf(u2), r * __builtin_amdgcn_sinf(u2), z);
}

PTD float fresnel_schlick(float n1, float n2, float c) {                 // bsdf.h:48-51
    const float q = (n1 - n2) * rcp(n1 + n2);
    const float r0 = q * q;
    const float m = 1.0f - c;
    const float m2 = m * m;
    return __builtin_fmaf(1.0f - r0, m2 * m2 * m, r0);
}

// bsdf.h:53-69.  tracePath passes f90 = 1 (bsdf.h:113), for which the last line is lerp(f0, 1, fr) = (1 - fr)*f0 + fr.
PTD float fresnel_ratio(V3 dir, V3 nrm, float f0, float f90, float n1, float n2) {
    float cX = -dot(dir, nrm);
    if (n1 > n2) {
        const float n = n1 * rcp(n2);
        const float s2 = n * n * __builtin_fmaf(-cX, cX, 1.0f);
        if (s2 > 1.0f) return f90;
        cX = fsqrt(1.0f - s2);
    }
    const float fr = fresnel_schlick(n1, n2, cX);
    return __builtin_fmaf(1.0f - fr, f0, fr * f90);
}

PTD V3 reflect(V3 v, V3 n) { return madd(n, -2.0f * dot(v, n), v); }      // vec_math.h:19-21
// vec_math.h:23-30: the zero vector on total reflection (k < 0)
PTD V3 refract(V3 I, V3 N, float eta) {
    const float ndi = dot(N, I);
    const float k = __builtin_fmaf(-eta * eta, __builtin_fmaf(-ndi, ndi, 1.0f), 1.0f);
    V3 r = v3(0.0f, 0.0f, 0.0f);
    if (!(k < 0.0f)) r = madd(N, -__builtin_fmaf(eta, ndi, fsqrt(k)), I * eta);
    return r;
}
PTD V3 lerp3(V3 v0, V3 v1, float t) { return madd(v1, t, v0 * (1.0f - t)); }   // vec_math.h:10-12

struct Path {
    V3 o, d, coef, contrib;
    uint32_t key;
    int seg;
};

// The tracePath loop body after the nearest hit is known (bsdf.h:84-165), in two parts: the emissive term, and
// everything from the Beer attenuation to the roulette.  Written with one exit and unconditional updates of the path: a
// lane whose path ends here (miss, roulette, depth) is refilled before its state is read again, so its o, d, coef may
// hold anything, and the compiler need not keep the old and the new state side by side (every early return used to cost
// a round of register copies per iteration).
PTD void shade_emissive(Path& p, bool hit, const DMat& M) {      // bsdf.h:90-92; M.emits: emissive.norm() > 0, decided at upload
    if (hit && M.emits != 0) {
        const V3 em = v3(M.emissive[0], M.emissive[1], M.emissive[2]);
        p.contrib = v3(__builtin_fmaf(em.x, p.coef.x, p.contrib.x), __builtin_fmaf(em.y, p.coef.y, p.contrib.y), __builtin_fmaf(em.z, p.coef.z, p.contrib.z));
    }
}

// bsdf.h:95-165 for a hit at distance t with surface record s; true = continue.
PTD bool shade_scatter(Path& p, const Surf& s, const DMat& M, float t, bool hit, int n_bounces, PhaseClock& clk) {
    const float f0 = M.f0, transparency = M.transparency;
    float p_specular = f0, p_refract = transparency;
    const bool glassy = (f0 > 0.0f) || (transparency > 0.0f);
    float ior = 1.0f;
    if (p.seg > 0 && s.inside) {                       // Beer attenuation, bsdf.h:95-98
        const V3 rc = v3(M.refractionColor[0], M.refractionColor[1], M.refractionColor[2]);
        p.coef = p.coef * v3(fexp(-rc.x * t), fexp(-rc.y * t), fexp(-rc.z * t));
    }
    if (glassy) {
        ior = M.IOR;
        if (f0 > 0.0f) {
            p_specular = fresnel_ratio(p.d, s.nrm, f0, 1.0f, s.inside ? ior : 1.0f, s.inside ? 1.0f : ior);
            p_refract = transparency * (1.0f - p_specular) * rcp(1.0f - f0);
        }
    }
    const uint32_t slot = 2u + 6u * (uint32_t)p.seg;
    int mode = 3;
    float p_ray = 1.0f - (p_specular + p_refract);
    if (glassy) {
        const float u = rng_u(p.key, slot);
        if (p_specular > 0.0f && u < p_specular) { mode = 1; p_ray = p_specular; }
        else if (p_refract > 0.0f && u < p_specular + p_refract) { mode = 2; p_ray = p_refract; }
    }
    p_ray = __builtin_fmaxf(p_ray, 0.001f);
    PHASE(clk, 3);
    V3 ndir;
    V3 tint;
    if (mode == 2) {
        const float eta = s.inside ? ior : rcp(ior);
        V3 rd = unit_or_zero(refract(p.d, s.nrm, eta));          // .normalized() of the zero vector is the zero vector
        const float rr = M.refractionRoughness * M.refractionRoughness;
        if (rr != 0.0f) {
            float r1, r2;
            rng_pair(p.key, slot + 3u, r1, r2);
            const V3 hd = unit(s.nrm + sphere_point(r1, r2));
            rd = unit(lerp3(rd, hd, rr));
        }
        ndir = rd;
        tint = v3(1.0f, 1.0f, 1.0f);
    } else {
        // the diffuse direction is drawn unconditionally in the reference (bsdf.h:139); a mirror-smooth specular
        // bounce never reads it
        const float sr = M.specularRoughness * M.specularRoughness;
        V3 dd = v3(0.0f, 0.0f, 0.0f);
        if (mode != 1 || sr != 0.0f) {
            float u1, u2;
            rng_pair(p.key, slot + 1u, u1, u2);
            dd = unit(s.nrm + sphere_point(u1, u2));
        }
        if (mode == 1) {
            V3 sd = unit(reflect(p.d, s.nrm));
            if (sr != 0.0f) sd = unit(lerp3(sd, dd, sr));
            ndir = sd;
            tint = v3(M.specularColor[0], M.specularColor[1], M.specularColor[2]);
        } else {
            ndir = dd;
            tint = v3(M.albedo[0], M.albedo[1], M.albedo[2]);
        }
    }
    PHASE(clk, 4);
    p.coef = p.coef * tint * rcp(p_ray);
    const float q = max3(p.coef.x, p.coef.y, p.coef.z);
    // Russian roulette (bsdf.h:160-164): u in (0,1], so q >= 1 always survives
    // and q <= 0 never does; only 0 < q < 1 (or NaN) needs the draw.
    bool dead = !hit || q <= 0.0f;
    if (!(q >= 1.0f)) dead = dead || rng_u(p.key, slot + 5u) > q;
    p.coef = p.coef * rcp(q);
    p.o = s.pos;
    p.d = ndir;
    p.seg++;
    return !dead && p.seg <= n_bounces;
}

// The whole loop body: surface record and material of the winner, then the two parts above.
PTD bool path_shade(Path& p, const Best& b0, const FShade* shade, const DMat* mats, int n_bounces, PhaseClock& clk) {
    const bool hit = b0.id >= 0;
    Best b = b0;
    b.id = hit ? b0.id : 0;                            // a miss reads primitive 0's records and is dropped at the end
    const Surf s = surface(b, p.o, p.d, shade);
    const DMat& M = mats[s.material];
    PHASE(clk, 2);
    shade_emissive(p, hit, M);
    return shade_scatter(p, s, M, b.t, hit, n_bounces, clk);
}

// What is left of the loop body for a hit on a material with f0 = 0 and transparency = 0 that is not reached from
// inside (no Beer term): mode 3 with p_ray = 1 (bsdf.h:130-133,139,152-165).  `keep` lanes (hits that need the full
// shade_scatter) come out with d, coef and seg unchanged and o moved to the hit position.
PTD bool shade_diffuse(Path& p, const Surf& s, const DMat& M, bool hit, bool keep, int n_bounces) {
    const uint32_t slot = 2u + 6u * (uint32_t)p.seg;
    float u1, u2;
    rng_pair(p.key, slot + 1u, u1, u2);
    const V3 dd = unit(s.nrm + sphere_point(u1, u2));
    V3 coef = p.coef * v3(M.albedo[0], M.albedo[1], M.albedo[2]);
    const float q = max3(coef.x, coef.y, coef.z);
    // (the roulette draw is made for every lane: a diffuse albedo below 1 needs it nearly always, and a branch around
    // eleven instructions costs more than it saves)
    const bool dead = !hit | (q <= 0.0f) | (!(q >= 1.0f) & (rng_u(p.key, slot + 5u) > q));
    coef = coef * rcp(q);
    p.coef = v3(keep ? p.coef.x : coef.x, keep ? p.coef.y : coef.y, keep ? p.coef.z : coef.z);
    p.d = v3(keep ? p.d.x : dd.x, keep ? p.d.y : dd.y, keep ? p.d.z : dd.z);
    p.o = s.pos;
    p.seg += keep ? 0 : 1;
    return !dead && p.seg <= n_bounces;
}

// The loop body of bsdf_alternative.h:158-217 after the nearest hit is known: Lambertian or SmoothDielectric by
// Material::bsdfType; true = continue.  A Lambertian sample survives only above the surface (f = 0 otherwise, and the
// roulette then always ends the path) with weight albedo * min(N.L, 1) / N.L.
PTD bool path_shade_alt(Path& p, const Best& b, const FShade* shade, const DMat* mats, int n_bounces, PhaseClock& clk) {
    if (b.id < 0) return false;
    const Surf s = surface(b, p.o, p.d, shade);
    const DMat& M = mats[s.material];
    const V3 em = v3(M.emissive[0], M.emissive[1], M.emissive[2]);
    PHASE(clk, 2);
    if (M.emits != 0) p.contrib = v3(__builtin_fmaf(em.x, p.coef.x, p.contrib.x), __builtin_fmaf(em.y, p.coef.y, p.contrib.y), __builtin_fmaf(em.z, p.coef.z, p.contrib.z));
    const uint32_t slot = 2u + 6u * (uint32_t)p.seg;
    V3 ndir, w;
    if (M.bsdfType == 1) {
        const float ior = M.IOR;
        float pr = fresnel_ratio(p.d, s.nrm, M.f0, 1.0f, s.inside ? ior : 1.0f, s.inside ? ior : 1.0f);
        float pt = 1.0f - pr;
        pr = M.f0 > 0.0f ? pr : 0.0f;
        pt = M.transparency > 0.0f ? pt : 0.0f;
        bool transmission = false;
        ndir = v3(0.0f, 0.0f, 0.0f);
        if (pr != 0.0f || pt != 0.0f) {
            const float ndi = dot(s.nrm, p.d);
            if (rng_u(p.key, slot) < pr * rcp(pr + pt)) {
                ndir = unit_or_zero(madd(s.nrm, -2.0f * ndi, p.d));
            } else {
                transmission = true;
                const float eta = s.inside ? ior : rcp(ior);
                const float k = __builtin_fmaf(-eta * eta, __builtin_fmaf(-ndi, ndi, 1.0f), 1.0f);
                if (!(k < 0.0f)) ndir = unit_or_zero(madd(s.nrm, -__builtin_fmaf(eta, ndi, fsqrt(k)), p.d * eta));
            }
        }
        const float pdf = transmission ? pt : 1.0f - pt;
        if (!(pdf > 0.0f)) return false;
        w = v3(M.specularColor[0], M.specularColor[1], M.specularColor[2]) * rcp(pdf);
    } else {
        const V3 N = s.nrm;
        const float sign = N.z < 0.0f ? -1.0f : 1.0f;
        const float a = -rcp(sign + N.z);
        const float bb = N.x * N.y * a;
        const V3 ox = v3(__builtin_fmaf(sign * N.x * N.x, a, 1.0f), sign * bb, -sign * N.x);
        const V3 oy = v3(bb, __builtin_fmaf(N.y * N.y, a, sign), -N.y);
        float u1, u2;
        rng_pair(p.key, slot + 1u, u1, u2);
        const V3 r = sphere_point(u1, u2);
        ndir = unit_or_zero(madd(ox, r.x, madd(oy, r.y, N * r.z)));
        const float ndl = dot(N, ndir), ndv = -dot(N, p.d);
        if (!(ndl > 0.0f) || !(ndv > 0.0f)) return false;
        w = v3(M.albedo[0], M.albedo[1], M.albedo[2]) * (__builtin_fminf(ndl, 1.0f) * rcp(ndl));
    }
    PHASE(clk, 4);
    p.coef = p.coef * w;
    const float q = max3(p.coef.x, p.coef.y, p.coef.z);
    if (q <= 0.0f) return false;
    if (!(q >= 1.0f)) {
        if (q < rng_u(p.key, slot + 5u)) return false;
    }
    p.coef = p.coef * rcp(q);
    p.o = s.pos;
    p.d = ndir;
    p.seg++;
    return p.seg <= n_bounces;
}

// What the nearest-hit search reads: the record array and runs of the primitives the loop tests, and the
// hierarchy over the others (n_nodes = 0: none).
struct SceneRefs {
    const float* fdata;
    const FRun* runs;
    int n_runs;
    const BNode* nodes;
    const BPrim* bprims;
    int n_nodes;
    const BFat* fat;
    V3 fat_origin, fat_step;
    FSorted sorted;       // data == null: loop over the runs instead
};

// One iteration of the tracePath loop (bsdf.h:78-166); true = continue.
template <bool BVH, bool SORTED, int KM = KM_ALL>
PTD bool path_step(Path& p, const SceneRefs& sc, const FShade* shade, const DMat* mats, int n_bounces, int bsdf_model, PhaseClock& clk) {
    Best b = best_none();
    const float dd = dot(p.d, p.d), idd = rcp(dd);
    if (SORTED) scan_sorted<KM>(p.o, p.d, dd, idd, sc.sorted, b);
    else scan_runs(p.o, p.d, dd, idd, as_constant(sc.fdata), as_constant(sc.runs), 0, sc.n_runs, b);
    if (BVH) { int n = 0; walk_hierarchy(p.o, p.d, dd, idd, sc.nodes, sc.n_nodes, sc.bprims, b, n, 0x7fffffff); }
    PHASE(clk, 1);
    if (bsdf_model == PT_BSDF_CLASSES) return path_shade_alt(p, b, shade, mats, n_bounces, clk);
    return path_shade(p, b, shade, mats, n_bounces, clk);
}

// A path of the render kernel that searches a hierarchy: the segment's nearest-hit search is spread over several
// iterations of the wave's loop, so its state travels with the path.
struct WalkPath : Path {
    int node;      // -1: the segment's search has not begun; otherwise the hierarchy node to go on from
    int sp;        // height of the lane's stack (fat form only)
    Best best;
};

// One iteration of the wave's loop for such a path.  A segment costs a ray anything from a handful to hundreds
// of node visits; were every iteration a whole segment, the wave would wait for its slowest ray each time.
// Instead every lane visits at most walk_budget nodes per iteration, and lanes whose search is over are shaded
// (and start their next segment) once shade_gate of them have gathered or nobody is searching any more, so that
// the shading code does not run for one or two lanes at a time.
template <bool FAT>
PTD bool path_step_walk(WalkPath& p, const SceneRefs& sc, const FShade* shade, const DMat* mats, int n_bounces,
                        int bsdf_model, int walk_budget, int shade_gate, int leaf_gate, uint32_t* stack, PhaseClock& clk) {
    const float dd = dot(p.d, p.d), idd = rcp(dd);
    if (p.node < 0) {
        p.best = best_none();
        scan_sorted<KM_PLANES>(p.o, p.d, dd, idd, sc.sorted, p.best);      // only planes stay outside the hierarchy
        p.node = 0;
        p.sp = 0;
    }
    PHASE(clk, 7);
    bool done;
    if (FAT) {
        walk_fat(p.o, p.d, dd, idd, sc.fat, sc.fat_origin, sc.fat_step, sc.bprims, p.best, p.node, p.sp, stack, walk_budget, leaf_gate);
        done = p.node == WALK_NONE && p.sp == 0;
    } else {
        walk_hierarchy(p.o, p.d, dd, idd, sc.nodes, sc.n_nodes, sc.bprims, p.best, p.node, walk_budget);
        done = p.node >= sc.n_nodes;
    }
    PHASE(clk, 1);
    const unsigned long long done_m = __ballot(done), busy_m = __ballot(!done);
    if (!done || (__popcll(done_m) < shade_gate && busy_m != 0ull)) return true;
    p.node = -1;
    if (bsdf_model == PT_BSDF_CLASSES) return path_shade_alt(p, p.best, shade, mats, n_bounces, clk);
    return path_shade(p, p.best, shade, mats, n_bounces, clk);
}

}  // namespace ptf

namespace {

using namespace ptf;

constexpr int LDS_MAX_PRIMS = 64;   // shading records kept in LDS up to this many primitives
constexpr int LDS_MAX_MATS = 32;

struct FastCtx {
    const FShade* shade;   // LDS or global
    const DMat* mats;      // LDS or global
    float* pxt;            // LDS, this wave's [3][pxs]: i-0.5, H-j-0.5, K_px bits
    uint32_t pxs;          // pixels each of the three rows has room for
    float* stage;          // LDS, FDATA_CHUNK floats shared by the workgroup (LDS-tiled kernel only)
    uint32_t* wstack;      // LDS, this lane's column of the wave's walk stack (fat hierarchy only)
    PhaseClock clk;
};

// BVH: 0 = loop over all primitives, 1 = hierarchy in skip-link form, 2 = hierarchy in fat form.
// SORTED: the loop runs over the kind-sorted records (PT_MATH_FAST) instead of the runs in scene order.
template <int BVH, bool SORTED, int KM = KM_ALL>
struct FastTracer {
    using Path = typename std::conditional<BVH != 0, ptf::WalkPath, ptf::Path>::type;
    // The hierarchy walk counts lanes (shade gate), so there only lanes with a path step.
    static constexpr bool STEP_IDLE_LANES = BVH == 0;
    // Lane px prepares pixel px of the tile: jitter-free sample-plane coordinates
    // (main.cu:62) and the pixel's RNG key.
    static PTD void prologue(const RenderArgs& a, FastCtx& c, uint32_t px, const ptw::PixelRef& ref) {
        c.pxt[px] = (float)ref.i - 0.5f;
        c.pxt[c.pxs + px] = (float)(a.max_y - ref.j) - 0.5f;
        c.pxt[2u * c.pxs + px] = __uint_as_float(rng_hash(ref.index ^ a.seed_hash));
    }
    static PTD void start(Path& p, const RenderArgs& a, FastCtx& c, uint32_t px, uint32_t, uint32_t sample) {
        const uint32_t kpx = __float_as_uint(c.pxt[2u * c.pxs + px]);
        p.key = rng_hash(kpx + 0x9E3779B9u * (sample + 1u));
        float jx, jy;
        rng_pair(p.key, 0u, jx, jy);
        const float sx = c.pxt[px] + jx, sy = c.pxt[c.pxs + px] + jy;
        const V3 cam = v3(a.cam[0], a.cam[1], a.cam[2]);
        const V3 pc = v3(__builtin_fmaf(a.du, sx, a.p00[0]), __builtin_fmaf(a.dv, sy, a.p00[1]), a.p00[2]);
        p.o = cam;
        p.d = unit(pc - cam);
        p.coef = v3(1.0f, 1.0f, 1.0f);
        p.contrib = v3(0.0f, 0.0f, 0.0f);
        p.seg = 0;
        begin_search(p);
    }
    // The same in two halves for the ray stash: what depends on the pool item, and what sets up the lane's path.
    static PTD void make_ray(const RenderArgs& a, FastCtx& c, uint32_t px, uint32_t, uint32_t sample, float4& ray) {
        const uint32_t kpx = __float_as_uint(c.pxt[2u * c.pxs + px]);
        const uint32_t key = rng_hash(kpx + 0x9E3779B9u * (sample + 1u));
        float jx, jy;
        rng_pair(key, 0u, jx, jy);
        const float sx = c.pxt[px] + jx, sy = c.pxt[c.pxs + px] + jy;
        const V3 cam = v3(a.cam[0], a.cam[1], a.cam[2]);
        const V3 pc = v3(__builtin_fmaf(a.du, sx, a.p00[0]), __builtin_fmaf(a.dv, sy, a.p00[1]), a.p00[2]);
        const V3 d = unit(pc - cam);
        ray = make_float4(d.x, d.y, d.z, __uint_as_float(key));
    }
    static PTD void begin(Path& p, const RenderArgs& a, FastCtx&, const float4& ray) {
        p.key = __float_as_uint(ray.w);
        p.o = v3(a.cam[0], a.cam[1], a.cam[2]);
        p.d = v3(ray.x, ray.y, ray.z);
        p.coef = v3(1.0f, 1.0f, 1.0f);
        p.contrib = v3(0.0f, 0.0f, 0.0f);
        p.seg = 0;
        begin_search(p);
    }
    static PTD void begin_search(ptf::Path&) {}
    static PTD void begin_search(ptf::WalkPath& p) { p.node = -1; p.sp = 0; }
    static PTD bool step(Path& p, const RenderArgs& a, FastCtx& c) {
        SceneRefs sc;
        sc.fdata = a.fdata; sc.runs = a.runs; sc.n_runs = a.n_runs;
        sc.nodes = a.bnodes; sc.bprims = a.bprims; sc.n_nodes = a.n_bnodes; sc.fat = a.bfat; sc.sorted = a.sorted;
        sc.fat_origin = v3(a.fat_origin[0], a.fat_origin[1], a.fat_origin[2]); sc.fat_step = v3(a.fat_step[0], a.fat_step[1], a.fat_step[2]);
        return step_path(p, sc, a, c);
    }
    static PTD bool step_path(ptf::Path& p, const SceneRefs& sc, const RenderArgs& a, FastCtx& c) {
        return path_step<false, SORTED, KM>(p, sc, c.shade, c.mats, a.n_bounces, a.bsdf_model, c.clk);
    }
    static PTD bool step_path(ptf::WalkPath& p, const SceneRefs& sc, const RenderArgs& a, FastCtx& c) {
        return path_step_walk<BVH == 2>(p, sc, c.shade, c.mats, a.n_bounces, a.bsdf_model, a.walk_budget, a.shade_gate, a.leaf_gate, c.wstack, c.clk);
    }
    static PTD bool begins_segment(const ptf::Path&) { return true; }
    static PTD bool begins_segment(const ptf::WalkPath& p) { return p.node < 0; }
    static PTD void idle(Path& p) {
        p.o = v3(0.0f, 0.0f, 0.0f); p.d = v3(0.0f, 0.0f, 0.0f);
        p.coef = v3(0.0f, 0.0f, 0.0f); p.contrib = v3(0.0f, 0.0f, 0.0f); p.key = 0; p.seg = 0;
        begin_search(p);
    }
    // getNearestIntersection with the records staged through LDS: EVERY thread of the workgroup calls this
    // (it contains barriers); lanes without a path carry a zero ray, which no test accepts.
    static PTD Best nearest_block(const Path& p, const RenderArgs& a, FastCtx& c) {
        Best b = best_none();
        const float dd = dot(p.d, p.d);
        const float idd = rcp(dd);
        for (int chunk = 0; chunk < a.n_chunks; chunk++) {
            const float4* src = reinterpret_cast<const float4*>(a.fdata + (size_t)chunk * FDATA_CHUNK);
            float4* dst = reinterpret_cast<float4*>(c.stage);
            __syncthreads();                                   // everyone is done with the previous chunk
            for (int w = threadIdx.x; w < FDATA_CHUNK / 4; w += ptw::BLOCK) dst[w] = src[w];
            __syncthreads();
            scan_runs(p.o, p.d, dd, idd, (const float*)c.stage - (size_t)chunk * FDATA_CHUNK, as_constant(a.runs), a.chunk_runs[chunk], a.chunk_runs[chunk + 1], b);
        }
        PHASE(c.clk, 1);
        return b;
    }
    static PTD bool shade(Path& p, const Best& b, const RenderArgs& a, FastCtx& c) {
        if (a.bsdf_model == PT_BSDF_CLASSES) return path_shade_alt(p, b, c.shade, c.mats, a.n_bounces, c.clk);
        return path_shade(p, b, c.shade, c.mats, a.n_bounces, c.clk);
    }
    static PTD void radiance(const Path& p, float rgb[3]) { rgb[0] = p.contrib.x; rgb[1] = p.contrib.y; rgb[2] = p.contrib.z; }
    static PTD void phase(FastCtx& c, int i) { PHASE(c.clk, i); (void)c; (void)i; }
};


// ---- PT_MATH_FAST without a hierarchy: camera rays in batches, hits that need the full shading set aside ---------
// (the loop itself is pt_batched.h's; this is the fast arithmetic it runs on)

// scan_sorted over the records whose bit is set in `mask` (bit s = record s, numbered through the kinds in FP_*
// order, which is the order of this function).  Same tests, same (distance, index) rule: the result is scan_sorted's
// whenever no record that the ray hits is masked out.
template <int KM = KM_ALL>
PTD void scan_sorted_masked(V3 o, V3 d, float dd, float idd, const FSorted& S0, unsigned long long mask, Best& b) {
    const KFloatP base = as_constant(S0.data);
    struct { int off[FP_KINDS], count[FP_KINDS]; } S;
    sorted_header<KM>(S0, base, S.off, S.count);
    const V3 inv = v3(rcp(d.x), rcp(d.y), rcp(d.z));
    int first = 0;
#define PT_KIND_LOOP(KIND, STRIDE, ID_AT, CALL)                                                              \
    if (KM & (1 << KIND)) {                                                                                  \
        const int cnt = S.count[KIND];                                                                       \
        unsigned long long m = cnt > 0 ? (mask >> first) & (cnt >= 64 ? ~0ull : ((1ull << cnt) - 1ull)) : 0ull; \
        first += cnt;                                                                                        \
        for (; m != 0ull; m &= m - 1ull) {                                                                   \
            const KFloatP R = base + S.off[KIND] + __builtin_ctzll(m) * (STRIDE);                            \
            const int id = __float_as_int(R[ID_AT]);                                                         \
            CALL;                                                                                            \
        }                                                                                                    \
    }
    PT_KIND_LOOP(FP_PLANE, 16, 12, test_plane(R, o, d, id, b))
    PT_KIND_LOOP(FP_CUBE, 16, 12, test_cube(R, o, d, id, b))
    PT_KIND_LOOP(FP_CYLINDER, 16, 12, test_cylinder(R, o, d, id, b))
    PT_KIND_LOOP(FP_SPHERE, 16, 12, test_sphere(R, o, d, id, b))
    PT_KIND_LOOP(FP_SPHERE_WS, 8, 6, test_sphere_ws(R, o, d, dd, idd, id, b))
    PT_KIND_LOOP(FP_PLANE_GEN, 20, 16, test_plane_gen(R, o, d, id, b))
    PT_KIND_LOOP(FP_PLANE_X, 8, 5, test_plane_axis<0>(R, o, d, inv, id, b))
    PT_KIND_LOOP(FP_PLANE_Y, 8, 5, test_plane_axis<1>(R, o, d, inv, id, b))
    PT_KIND_LOOP(FP_PLANE_Z, 8, 5, test_plane_axis<2>(R, o, d, inv, id, b))
    PT_KIND_LOOP(FP_CUBE_AA, 8, 6, test_cube_box(R, o, inv, id, b))
#undef PT_KIND_LOOP
}

// The fast kernels' arithmetic for the loop of pt_batched.h.  A set-aside hit keeps its position (the surface record
// needs nothing else of the ray).
template <bool SORTED, int KM>
struct FastBatched {
    using V3 = ptf::V3;
    using Path = ptf::Path;
    using Best = ptf::Best;
    using Surf = ptf::Surf;
    using Ctx = FastCtx;
    using Scene = SceneRefs;
    using T = FastTracer<0, SORTED>;
    static PTD V3 v3(float x, float y, float z) { return ptf::v3(x, y, z); }
    static PTD Scene scene(const RenderArgs& a) {
        SceneRefs sc;
        sc.fdata = a.fdata; sc.runs = a.runs; sc.n_runs = a.n_runs;
        sc.nodes = nullptr; sc.bprims = nullptr; sc.n_nodes = 0; sc.fat = nullptr; sc.sorted = a.sorted;
        return sc;
    }
    static PTD void prologue(const RenderArgs& a, Ctx& c, uint32_t px, const ptw::PixelRef& ref) { T::prologue(a, c, px, ref); }
    static PTD void make_ray(const RenderArgs& a, Ctx& c, uint32_t px, uint32_t sample, float4& ray) { T::make_ray(a, c, px, 0u, sample, ray); }
    static PTD const float* cull_boxes(const RenderArgs& a, int& n) { n = a.n_slots; return a.slot_box; }
    static PTD Best scan(V3 o, V3 d, const Scene& sc) {
        Best b = best_none();
        const float dd = dot(d, d), idd = rcp(dd);
        if (SORTED) scan_sorted<KM>(o, d, dd, idd, sc.sorted, b);
        else scan_runs(o, d, dd, idd, as_constant(sc.fdata), as_constant(sc.runs), 0, sc.n_runs, b);
        return b;
    }
    static PTD Best scan_camera(V3 o, V3 d, const Scene& sc, const RenderArgs& a, unsigned long long cand) {
        Best b = best_none();
        const float dd = dot(d, d), idd = rcp(dd);
        if (SORTED) {
            if (a.slot_box != nullptr && a.n_slots <= 64) scan_sorted_masked<KM>(o, d, dd, idd, sc.sorted, cand, b);
            else scan_sorted<KM>(o, d, dd, idd, sc.sorted, b);
        } else {
            scan_runs(o, d, dd, idd, as_constant(sc.fdata), as_constant(sc.runs), 0, sc.n_runs, b);
        }
        return b;
    }
    static PTD Surf surface(const Best& b, V3 o, V3 d, Ctx& c) { return ptf::surface(b, o, d, c.shade); }
    static PTD Surf surface_parked(const Best& b, V3 pos, V3 d, Ctx& c) { return ptf::surface_at(b, pos, d, c.shade); }
    static PTD V3 park_point(V3, V3 pos) { return pos; }
    static PTD V3 camera_hit_point(V3 cam, V3 d, float t) { return madd(d, t, cam); }
    static PTD void shade_emissive(Path& p, bool hit, const DMat& M) { ptf::shade_emissive(p, hit, M); }
    static PTD bool shade_diffuse(Path& p, const Surf& s, const DMat& M, bool hit, bool keep, int n_bounces) { return ptf::shade_diffuse(p, s, M, hit, keep, n_bounces); }
    static PTD bool shade_scatter(Path& p, const Surf& s, const DMat& M, float t, bool hit, int n_bounces, Ctx& c) { return ptf::shade_scatter(p, s, M, t, hit, n_bounces, c.clk); }
    // albedo / max(albedo): what shade_diffuse leaves of a throughput of 1
    static PTD V3 first_diffuse_coef(const DMat& M) { const V3 alb = ptf::v3(M.albedo[0], M.albedo[1], M.albedo[2]); return alb * rcp(max3(alb.x, alb.y, alb.z)); }
    static PTD void idle(Path& p) { T::idle(p); }
    static PTD void phase(Ctx& c, int i) { PHASE(c.clk, i); (void)c; (void)i; }
};

template <bool LDS_TABLES, bool SORTED, int KM>
__global__ __launch_bounds__(ptw::BLOCK) void render_fast_batched_kernel(RenderArgs a, uint32_t n_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    const uint32_t P = 1u << a.tile_log2;
    const size_t per_wave = ptb::batched_wave_bytes(P, (uint32_t)a.q_cap);
    FastCtx ctx;
    if (LDS_TABLES) {
        // The whole workgroup copies the shading and material tables into LDS once.
        FShade* lshade = reinterpret_cast<FShade*>(dyn_lds + per_wave * ptw::WAVES_PER_BLOCK);
        DMat* lmats = reinterpret_cast<DMat*>(lshade + a.n_prims);
        const uint32_t* gs = reinterpret_cast<const uint32_t*>(a.fshade);
        uint32_t* ls = reinterpret_cast<uint32_t*>(lshade);
        for (uint32_t w = threadIdx.x; w < (uint32_t)a.n_prims * (sizeof(FShade) / 4); w += ptw::BLOCK) ls[w] = gs[w];
        const uint32_t* gm = reinterpret_cast<const uint32_t*>(a.mats);
        uint32_t* lm = reinterpret_cast<uint32_t*>(lmats);
        for (uint32_t w = threadIdx.x; w < (uint32_t)a.n_mats * (sizeof(DMat) / 4); w += ptw::BLOCK) lm[w] = gm[w];
        __syncthreads();
        ctx.shade = lshade; ctx.mats = lmats;
    } else {
        ctx.shade = a.fshade; ctx.mats = a.mats;
    }
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x * ptw::WAVES_PER_BLOCK + wave;
    if (tile >= n_tiles) return;
    ptb::BatchLds L;
    ptb::carve_wave_lds(ptb::lds_off(dyn_lds + per_wave * wave), P, (uint32_t)a.q_cap, L);
    ctx.pxt = L.pxt; ctx.pxs = P;
    ctx.wstack = nullptr;
    ctx.clk.start();
    ptb::render_tile_batched<FastBatched<SORTED, KM>>(a, tile, L, ctx);
    ctx.clk.flush(threadIdx.x & 63);
}

// ---- PT_MATH_FAST with a hierarchy: camera rays in batches against the tile's own short list of primitives -------
//
// The walk of the hierarchy is what a ray of unknown origin and direction needs.  The camera rays of a tile all leave
// one point towards a few adjacent pixels: once per tile the wave descends the tree with the tile's whole view pyramid
// (tile_hierarchy_candidates: breadth first, one node per lane, the box test of pt_batched.h's per-tile culling) and
// keeps the primitives whose boxes the pyramid meets — a few dozen of 10 000 small shapes.  The first segment of 64
// samples is then traced by all 64 lanes at once against the planes and that list (records through scalar loads,
// one wave-uniform branch on the kind per record), shaded together, and what goes on is parked in LDS exactly as
// ptb::camera_batch parks it for scenes without a hierarchy; the lanes of the main loop walk the hierarchy for
// secondary rays only.  A tile whose list does not fit 64 entries parks its camera rays unsearched instead and
// they walk like any other ray.  Arithmetic per path is unchanged, the (distance, index) rule does not care in which
// order or through which path the candidates are tested, and pixel sums are order-free: same image bit for bit.

struct WalkCamScene {
    SceneRefs refs;
    int cand;        // lane l: index into bprims of the l-th primitive of the hierarchy a camera ray of this tile can hit
    int n_cand;      // wave-uniform
};

constexpr uint32_t STASH_CAMERA = 0x40000000u;   // a parked camera ray whose search has not begun (direction, key | -, pixel)

// The primitives of the hierarchy whose boxes the tile's view pyramid meets: their indices into bprims, one per lane
// (lane l < n: the l-th), n returned; -1 if there are more than 64 or a level of the descent is wider than 64 nodes.
// `scratch`: 256 words of the wave's LDS.
PTD int tile_hierarchy_candidates(const RenderArgs& a, const ptb::TileRect& rect, uint32_t* scratch, unsigned lane, int& cand) {
    uint32_t* cur = scratch;
    uint32_t* nxt = scratch + 64;
    uint32_t* out = scratch + 128;
    int n_front = 1, n_out = 0;
    if (lane == 0) cur[0] = 0u;
    // a node's boxes are origin + step * (sum -+ diff) per axis (BFat); relative to the camera, and widened by more than
    // the rounding of this evaluation
    const V3 org = v3(a.fat_origin[0] - a.cam[0], a.fat_origin[1] - a.cam[1], a.fat_origin[2] - a.cam[2]);
    const V3 step = v3(a.fat_step[0], a.fat_step[1], a.fat_step[2]);
    const V3 pad = v3(4e-6f * (__builtin_fabsf(a.fat_origin[0]) + __builtin_fabsf(a.cam[0]) + 65536.0f * step.x),
                      4e-6f * (__builtin_fabsf(a.fat_origin[1]) + __builtin_fabsf(a.cam[1]) + 65536.0f * step.y),
                      4e-6f * (__builtin_fabsf(a.fat_origin[2]) + __builtin_fabsf(a.cam[2]) + 65536.0f * step.z));
    for (int level = 0; level < WALK_STACK_MAX + 8 && n_front > 0; level++) {
        bool ka = false, kb = false;
        int ra = 0, rb = 0;
        if ((int)lane < n_front) {
            const uint4* N = reinterpret_cast<const uint4*>(a.bfat + cur[lane]);
            const uint4 W0 = N[0], W1 = N[1];
#define PT_GRID_LO(w) ((float)((w) & 0xFFFFu))
#define PT_GRID_HI(w) ((float)((w) >> 16))
            const V3 sa = v3(PT_GRID_LO(W0.x), PT_GRID_HI(W0.x), PT_GRID_LO(W0.y)), da = v3(PT_GRID_HI(W0.y), PT_GRID_LO(W0.z), PT_GRID_HI(W0.z));
            const V3 sb = v3(PT_GRID_LO(W0.w), PT_GRID_HI(W0.w), PT_GRID_LO(W1.x)), db = v3(PT_GRID_HI(W1.x), PT_GRID_LO(W1.y), PT_GRID_HI(W1.y));
#undef PT_GRID_LO
#undef PT_GRID_HI
            const V3 la = (sa - da) * step + org - pad, ha = (sa + da) * step + org + pad;
            const V3 lb = (sb - db) * step + org - pad, hb = (sb + db) * step + org + pad;
            ka = !ptb::rect_misses_box(rect, la.x, la.y, la.z, ha.x, ha.y, ha.z);
            kb = !ptb::rect_misses_box(rect, lb.x, lb.y, lb.z, hb.x, hb.y, hb.z);
            ra = (int)W1.z; rb = (int)W1.w;
        }
        const unsigned long long leaf_a = __ballot(ka && ra < 0), leaf_b = __ballot(kb && rb < 0);
        const unsigned long long in_a = __ballot(ka && ra >= 0), in_b = __ballot(kb && rb >= 0);
        const int n_la = __popcll(leaf_a), n_lb = __popcll(leaf_b), n_ia = __popcll(in_a), n_ib = __popcll(in_b);
        if (n_out + n_la + n_lb > 64 || n_ia + n_ib > 64) return -1;
        if (ka && ra < 0) out[n_out + (int)ptw::prefix_count(leaf_a)] = (uint32_t)~ra;
        if (kb && rb < 0) out[n_out + n_la + (int)ptw::prefix_count(leaf_b)] = (uint32_t)~rb;
        if (ka && ra >= 0) nxt[(int)ptw::prefix_count(in_a)] = (uint32_t)ra;
        if (kb && rb >= 0) nxt[n_ia + (int)ptw::prefix_count(in_b)] = (uint32_t)rb;
        n_out += n_la + n_lb;
        n_front = n_ia + n_ib;
        uint32_t* t = cur; cur = nxt; nxt = t;
    }
    if (n_front > 0) return -1;
    cand = (int)lane < n_out ? (int)out[lane] : 0;
    return n_out;
}

// The fast arithmetic of a camera batch (ptb::camera_batch) for a scene with a hierarchy.
struct FastWalkCam {
    using V3 = ptf::V3;
    using Path = ptf::Path;
    using Best = ptf::Best;
    using Surf = ptf::Surf;
    using Ctx = FastCtx;
    using Scene = WalkCamScene;
    using T = FastTracer<2, true>;
    static PTD V3 v3(float x, float y, float z) { return ptf::v3(x, y, z); }
    static PTD void make_ray(const RenderArgs& a, Ctx& c, uint32_t px, uint32_t sample, float4& ray) { T::make_ray(a, c, px, 0u, sample, ray); }
    static PTD Best scan_camera(V3 o, V3 d, const Scene& sc, const RenderArgs&, unsigned long long) {
        Best b = best_none();
        const float dd = dot(d, d), idd = rcp(dd);
        scan_sorted<KM_PLANES>(o, d, dd, idd, sc.refs.sorted, b);          // only planes stay outside the hierarchy
        // (1/d as walk_fat hands it to the leaves' box test)
        const V3 inv = v3(__builtin_amdgcn_fmed3f(rcp(d.x), -1e30f, 1e30f), __builtin_amdgcn_fmed3f(rcp(d.y), -1e30f, 1e30f),
                          __builtin_amdgcn_fmed3f(rcp(d.z), -1e30f, 1e30f));
        for (int i = 0; i < sc.n_cand; i++) {
            const int ref = __builtin_amdgcn_readlane(sc.cand, i);
            const KFloatP R = as_constant(reinterpret_cast<const float*>(sc.refs.bprims + ref));
            const int kind = __float_as_int(R[12]), id = __float_as_int(R[13]);
            switch (kind) {
                case FP_SPHERE_WS: test_sphere_ws(R, o, d, dd, idd, id, b); break;
                case FP_CUBE: test_cube(R, o, d, id, b); break;
                case FP_CUBE_AA: test_cube_box(R, o, inv, id, b); break;
                case FP_SPHERE: test_sphere(R, o, d, id, b); break;
                default: test_cylinder(R, o, d, id, b); break;
            }
        }
        return b;
    }
    static PTD Surf surface(const Best& b, V3 o, V3 d, Ctx& c) { return ptf::surface(b, o, d, c.shade); }
    static PTD void shade_emissive(Path& p, bool hit, const DMat& M) { ptf::shade_emissive(p, hit, M); }
    static PTD bool shade_diffuse(Path& p, const Surf& s, const DMat& M, bool hit, bool keep, int n_bounces) { return ptf::shade_diffuse(p, s, M, hit, keep, n_bounces); }
    static PTD V3 first_diffuse_coef(const DMat& M) { const V3 alb = ptf::v3(M.albedo[0], M.albedo[1], M.albedo[2]); return alb * rcp(max3(alb.x, alb.y, alb.z)); }
};

// 64 camera rays made and parked unsearched (a tile whose candidate list did not fit); returns the number parked.
PTD uint32_t camera_batch_unsearched(const RenderArgs& a, FastCtx& ctx, const ptb::BatchLds& L, uint32_t made, uint32_t total, uint32_t npx, unsigned lane) {
    const uint32_t lg = (uint32_t)a.tile_log2, P = 1u << lg;
    const uint32_t w = made + lane;
    const uint32_t px = w & (P - 1u);
    const bool valid = w < total && px < npx;
    float4 ray;
    FastWalkCam::make_ray(a, ctx, px, (w >> lg) + (uint32_t)a.sample_begin, ray);
    const unsigned long long park = __ballot(valid);
    if (valid) {
        const uint32_t e = ptw::prefix_count(park);
        L.stash[e] = ray;
        L.stash[64u + e] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(px | STASH_CAMERA));
    }
    return (uint32_t)__popcll(park);
}

// One tile through the wave-pool loop of pt_wavepool.h: render_tile, with the camera batches above as the source of
// the lanes' work.
PTD void render_tile_walkcam(const RenderArgs& a, uint32_t tile, const ptb::BatchLds& L, FastCtx& ctx) {
    using X = FastWalkCam;
    const unsigned lane = ptw::lane_id();
    const uint32_t lg = (uint32_t)a.tile_log2, P = 1u << lg;
    const uint32_t lin_base = tile << lg;
    const uint32_t npx = min(P, a.n_lin - lin_base);
    const uint32_t total = (uint32_t)a.n_samples << lg;

    for (uint32_t k = lane; k < 3u * P; k += 64u) L.acc[k] = 0ull;
    WalkCamScene sc;
    sc.refs.fdata = a.fdata; sc.refs.runs = a.runs; sc.refs.n_runs = a.n_runs;
    sc.refs.nodes = a.bnodes; sc.refs.bprims = a.bprims; sc.refs.n_nodes = a.n_bnodes; sc.refs.fat = a.bfat; sc.refs.sorted = a.sorted;
    sc.refs.fat_origin = v3(a.fat_origin[0], a.fat_origin[1], a.fat_origin[2]); sc.refs.fat_step = v3(a.fat_step[0], a.fat_step[1], a.fat_step[2]);
    sc.cand = 0;
    {
        const RenderArgs la = ptb::lazy_args(a);
        if (lane < npx) {
            const ptw::PixelRef ref = ptw::pixel_of(la, lin_base + lane);
            L.out_px[lane] = ref.out;
            X::T::prologue(la, ctx, lane, ref);
        }
        const ptb::TileRect rect = ptb::tile_rect(la, lin_base, npx);
        sc.n_cand = tile_hierarchy_candidates(la, rect, reinterpret_cast<uint32_t*>(L.stash), lane, sc.cand);
    }
    sc.n_cand = __builtin_amdgcn_readfirstlane(sc.n_cand);
    const bool searched = sc.n_cand >= 0;
    CENSUS(8, 1); CENSUS(9, searched ? 1 : 0); CENSUS(10, searched ? sc.n_cand : 0);

    ptf::WalkPath path;
    X::T::idle(path);
    uint32_t my_px = 0;
    bool active = false;
    uint32_t made = 0;               // wave-uniform: pool items the batches have made
    uint32_t parked = 0;             // wave-uniform: entries in the stash
    uint32_t segments = 0;           // wave-uniform: path segments traced
    for (;;) {
        made = ptb::uniform(made); parked = ptb::uniform(parked);
        const unsigned long long want = __ballot(!active);
        uint32_t need = (uint32_t)__popcll(want);
        if (need != 0u) {
            const uint32_t rank = ptw::prefix_count(want);
            uint32_t served = 0;
            for (int pass = 0; pass < 2 && need != 0u; pass++) {   // what is parked, then a fresh batch for the rest
                if (parked == 0u) {
                    if (made >= total) break;
                    const RenderArgs la = ptb::lazy_args(a);
                    if (searched) parked = ptb::camera_batch<X>(la, ctx, L, sc, 0ull, made, total, npx, lane, segments);
                    else parked = camera_batch_unsearched(la, ctx, L, made, total, npx, lane);
                    made = min(made + 64u, total);
                    CENSUS(11, 1); CENSUS(12, parked);
                }
                const uint32_t take = min(need, parked);
                if (!active && rank >= served && rank < served + take) {
                    const uint32_t e = parked - 1u - (rank - served);
                    const float4 A = L.stash[e], B = L.stash[64u + e];
                    const uint32_t meta = __float_as_uint(B.w);
                    my_px = meta & 255u;
                    path.key = __float_as_uint(A.w);
                    path.contrib = v3(0.0f, 0.0f, 0.0f);
                    path.sp = 0;
                    if (meta & (ptb::STASH_FLAG | STASH_CAMERA)) {
                        const RenderArgs la = ptb::lazy_args(a);
                        path.o = v3(la.cam[0], la.cam[1], la.cam[2]);
                        path.coef = v3(1.0f, 1.0f, 1.0f);
                        path.seg = 0;
                        if (meta & STASH_CAMERA) {
                            path.d = v3(A.x, A.y, A.z);
                            path.node = -1;
                        } else {                               // a camera ray with its hit: the search is over, the full shading is due
                            path.d = v3(B.x, B.y, B.z);
                            path.best.t = A.x; path.best.id = __float_as_int(A.y);
                            path.node = WALK_NONE;
                        }
                    } else {
                        // throughput after the first hit, as shade_diffuse left it
                        path.o = v3(A.x, A.y, A.z);
                        path.d = v3(B.x, B.y, B.z);
                        path.coef = X::first_diffuse_coef(ctx.mats[(meta & ~(ptb::STASH_FLAG | STASH_CAMERA)) >> 8]);
                        path.seg = 1;
                        path.node = -1;
                    }
                    active = true;
                }
                parked -= take; served += take; need -= take;
            }
        }
        PHASE(ctx.clk, 0);
        if (__ballot(active) == 0ull) {
            if (made >= total && parked == 0u) break;
            continue;
        }
        segments += (uint32_t)__popcll(__ballot(active && path.node == -1));
        CENSUS(13, 1); CENSUS(14, __popcll(__ballot(active)));
        if (active) {
            const bool go_on = X::T::step_path(path, sc.refs, a, ctx);
            PHASE(ctx.clk, 5);
            if (!go_on) {
                ptb::add_light(L, my_px, path.contrib.x, path.contrib.y, path.contrib.z, a.sample_cap);
                active = false;
            }
        }
        PHASE(ctx.clk, 6);
    }
    const RenderArgs le = ptb::lazy_args(a);
    ptw::add_counters(le, segments, lane);
    ptw::WaveLds l; l.acc = L.acc; l.out_px = L.out_px;
    ptw::store_tile(le, l, npx, lane);
}

// Dynamic LDS of render_fast_walkcam_kernel: the walk stacks ([waves][a.walk_stack][64] words), then per wave the parked
// work (128 float4), the pixel sums, the pixel table and the framebuffer slots, sized by the tile.
__host__ __device__ constexpr uint32_t walkcam_wave_bytes(uint32_t P) { return 2048u + 40u * P; }

template <bool LDS_TABLES>
__global__ __launch_bounds__(ptw::BLOCK) void render_fast_walkcam_kernel(RenderArgs a, uint32_t n_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    __shared__ FShade lshade[LDS_TABLES ? LDS_MAX_PRIMS : 1];
    __shared__ DMat lmats[LDS_TABLES ? LDS_MAX_MATS : 1];
    FastCtx ctx;
    if (LDS_TABLES) {
        const uint32_t* gs = reinterpret_cast<const uint32_t*>(a.fshade);
        uint32_t* ls = reinterpret_cast<uint32_t*>(lshade);
        for (uint32_t w = threadIdx.x; w < (uint32_t)a.n_prims * (sizeof(FShade) / 4); w += ptw::BLOCK) ls[w] = gs[w];
        const uint32_t* gm = reinterpret_cast<const uint32_t*>(a.mats);
        uint32_t* lm = reinterpret_cast<uint32_t*>(lmats);
        for (uint32_t w = threadIdx.x; w < (uint32_t)a.n_mats * (sizeof(DMat) / 4); w += ptw::BLOCK) lm[w] = gm[w];
        __syncthreads();
        ctx.shade = lshade; ctx.mats = lmats;
    } else {
        ctx.shade = a.fshade; ctx.mats = a.mats;
    }
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x * ptw::WAVES_PER_BLOCK + wave;
    if (tile >= n_tiles) return;
    const uint32_t P = 1u << a.tile_log2;
    const uint32_t stack_words = (uint32_t)ptw::WAVES_PER_BLOCK * (uint32_t)a.walk_stack * 64u;
    uint32_t* wstack = reinterpret_cast<uint32_t*>(dyn_lds);
    unsigned char* mine = dyn_lds + 4u * stack_words + wave * walkcam_wave_bytes(P);
    ptb::BatchLds L;
    L.stash = reinterpret_cast<float4*>(mine);
    L.acc = reinterpret_cast<unsigned long long*>(mine + 2048u);
    L.pxt = reinterpret_cast<float*>(mine + 2048u + 24u * P);
    L.out_px = reinterpret_cast<uint32_t*>(mine + 2048u + 36u * P);
    L.gq = nullptr;
    ctx.pxt = L.pxt; ctx.pxs = P;
    ctx.wstack = wstack + wave * ((uint32_t)a.walk_stack * 64u) + (threadIdx.x & 63u);
    ctx.clk.start();
    render_tile_walkcam(a, tile, L, ctx);
    ctx.clk.flush(threadIdx.x & 63);
}

// Dynamic LDS of render_fast_kernel: the walk stacks ([waves][a.walk_stack][64] words, fat hierarchy only), then per
// wave the parked rays (64 float4), the pixel sums (3 x 64 bit per tile pixel), the pixel table (3 floats per tile
// pixel) and the framebuffer slots (one word per tile pixel) — sized by the tile, so that small tiles leave room for
// more workgroups per compute unit.
__host__ __device__ constexpr uint32_t fast_wave_bytes(uint32_t P) { return 1024u + 40u * P; }

template <bool LDS_TABLES, int BVH, bool SORTED, int KM = KM_ALL>
__global__ __launch_bounds__(ptw::BLOCK) void render_fast_kernel(RenderArgs a, uint32_t n_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    __shared__ FShade lshade[LDS_TABLES ? LDS_MAX_PRIMS : 1];
    __shared__ DMat lmats[LDS_TABLES ? LDS_MAX_MATS : 1];
    FastCtx ctx;
    if (LDS_TABLES) {
        // The whole workgroup copies the shading and material tables into LDS once.
        const uint32_t* gs = reinterpret_cast<const uint32_t*>(a.fshade);
        uint32_t* ls = reinterpret_cast<uint32_t*>(lshade);
        for (uint32_t w = threadIdx.x; w < (uint32_t)a.n_prims * (sizeof(FShade) / 4); w += ptw::BLOCK) ls[w] = gs[w];
        const uint32_t* gm = reinterpret_cast<const uint32_t*>(a.mats);
        uint32_t* lm = reinterpret_cast<uint32_t*>(lmats);
        for (uint32_t w = threadIdx.x; w < (uint32_t)a.n_mats * (sizeof(DMat) / 4); w += ptw::BLOCK) lm[w] = gm[w];
        __syncthreads();
        ctx.shade = lshade; ctx.mats = lmats;
    } else {
        ctx.shade = a.fshade; ctx.mats = a.mats;
    }
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x * ptw::WAVES_PER_BLOCK + wave;
    if (tile >= n_tiles) return;
    const uint32_t P = 1u << a.tile_log2;
    const uint32_t stack_words = BVH == 2 ? (uint32_t)ptw::WAVES_PER_BLOCK * (uint32_t)a.walk_stack * 64u : 0u;
    uint32_t* wstack = reinterpret_cast<uint32_t*>(dyn_lds);
    unsigned char* mine = dyn_lds + 4u * stack_words + wave * fast_wave_bytes(P);
    float4* stash = reinterpret_cast<float4*>(mine);
    ptw::WaveLds lds;
    lds.acc = reinterpret_cast<unsigned long long*>(mine + 1024u);
    ctx.pxt = reinterpret_cast<float*>(mine + 1024u + 24u * P);
    ctx.pxs = P;
    lds.out_px = reinterpret_cast<uint32_t*>(mine + 1024u + 36u * P);
    ctx.wstack = BVH == 2 ? wstack + wave * ((uint32_t)a.walk_stack * 64u) + (threadIdx.x & 63u) : nullptr;
    ctx.clk.start();
    ptw::render_tile<FastTracer<BVH, SORTED, KM>>(a, tile, lds, stash, ctx);
    ctx.clk.flush(threadIdx.x & 63);
}

// Large scenes: the workgroup stages the record array through LDS one chunk at a time and all four waves sweep
// each chunk together.  Shading records stay in global memory (one gather per hit), materials go to LDS when
// they fit.
template <bool LDS_MATS>
__global__ __launch_bounds__(ptw::BLOCK) void render_fast_tiled_kernel(RenderArgs a, uint32_t n_tiles) {
    __shared__ ptw::WaveLdsStore lds[ptw::WAVES_PER_BLOCK];
    __shared__ float pxt[ptw::WAVES_PER_BLOCK][192];
    __shared__ __attribute__((aligned(16))) float stage[FDATA_CHUNK];
    __shared__ DMat lmats[LDS_MATS ? LDS_MAX_MATS : 1];
    FastCtx ctx;
    ctx.shade = a.fshade;
    if (LDS_MATS) {
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
    ctx.pxt = pxt[wave];
    ctx.pxs = 64u;
    ctx.stage = stage;
    ctx.clk.start();
    ptw::render_tile_blocksync<FastTracer<0, false>>(a, tile, tile < n_tiles, lds[wave].refs(), ctx);
    ctx.clk.flush(threadIdx.x & 63);
}

template <bool BVH>
__global__ __launch_bounds__(256) void trace_fast_kernel(SceneRefs sc,
                                                         const FShade* __restrict__ shade, const DMat* __restrict__ mats,
                                                         const float* __restrict__ o, const float* __restrict__ d,
                                                         const uint32_t* __restrict__ pixel, const uint32_t* __restrict__ sample,
                                                         int n, uint32_t seed_hash, int n_bounces, int bsdf_model, float* rgb,
                                                         float* end_o, float* end_d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Path p;
    p.key = rng_hash(rng_hash(pixel[i] ^ seed_hash) + 0x9E3779B9u * (sample[i] + 1u));
    p.o = v3(o[3 * i], o[3 * i + 1], o[3 * i + 2]);
    p.d = v3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    p.coef = v3(1.0f, 1.0f, 1.0f); p.contrib = v3(0.0f, 0.0f, 0.0f); p.seg = 0;
    PhaseClock clk;
    // "no depth limit" for the steps, so that false means the path ended; the depth is counted here, and the ray of
    // a segment that ended the path is kept (the reference assigns `r` only when the path goes on, bsdf.h:165)
    for (;;) {
        const V3 o0 = p.o, d0 = p.d;
        if (!path_step<BVH, false>(p, sc, shade, mats, 0x7fffffff, bsdf_model, clk)) { p.o = o0; p.d = d0; break; }
        if (p.seg > n_bounces) break;
    }
    rgb[3 * i] = p.contrib.x; rgb[3 * i + 1] = p.contrib.y; rgb[3 * i + 2] = p.contrib.z;
    if (end_o) { end_o[3 * i] = p.o.x; end_o[3 * i + 1] = p.o.y; end_o[3 * i + 2] = p.o.z; }
    if (end_d) { end_d[3 * i] = p.d.x; end_d[3 * i + 1] = p.d.y; end_d[3 * i + 2] = p.d.z; }
}

// getNearestIntersection as the render kernels run it: the kind-sorted loops (all records, or the ones outside the
// hierarchy followed by the skip-link walk), then the winner's surface.
template <bool BVH>
__global__ __launch_bounds__(256) void nearest_fast_kernel(SceneRefs sc, const FShade* __restrict__ shade,
                                                           const float* __restrict__ o, const float* __restrict__ d, int n,
                                                           int32_t* prim, float* t, float* pos, float* nrm, int32_t* inside) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const V3 ro = v3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), rd = v3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    Best b = best_none();
    const float dd = dot(rd, rd), idd = rcp(dd);
    if (BVH) scan_sorted<KM_PLANES>(ro, rd, dd, idd, sc.sorted, b);
    else scan_sorted(ro, rd, dd, idd, sc.sorted, b);
    if (BVH) { int visits = 0; walk_hierarchy(ro, rd, dd, idd, sc.nodes, sc.n_nodes, sc.bprims, b, visits, 0x7fffffff); }
    const bool hit = b.id >= 0;
    if (prim) prim[i] = hit ? (b.id >> 3) : -1;
    Surf s;
    s.pos = v3(0.0f, 0.0f, 0.0f); s.nrm = s.pos; s.inside = false;
    if (hit) s = surface(b, ro, rd, shade);
    if (t) t[i] = hit ? b.t : 0.0f;
    if (pos) { pos[3 * i] = s.pos.x; pos[3 * i + 1] = s.pos.y; pos[3 * i + 2] = s.pos.z; }
    if (nrm) { nrm[3 * i] = s.nrm.x; nrm[3 * i + 1] = s.nrm.y; nrm[3 * i + 2] = s.nrm.z; }
    if (inside) inside[i] = hit && s.inside ? 1 : 0;
}

}  // namespace

#ifdef PT_PHASE_PROF
extern "C" int pt_dev_phase_cycles(unsigned long long* out, int reset) {
    PT_CHECK(hipDeviceSynchronize());
    PT_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_cycles), 24 * sizeof(unsigned long long)));
    if (reset) { unsigned long long z[24] = {0}; PT_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, sizeof(z))); }
    return 0;
}
#endif

#ifdef PT_WALK_CENSUS
extern "C" int pt_dev_walk_census(unsigned long long* out, int reset) {
    PT_CHECK(hipDeviceSynchronize());
    PT_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_walk_census), 16 * sizeof(unsigned long long)));
    if (reset) { unsigned long long z[16] = {0}; PT_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_walk_census), z, sizeof(z))); }
    return 0;
}
#endif

int ptk_render_fast(const RenderArgs& a, int variant, hipStream_t s) {
    if (a.n_lin == 0) return 0;
    const uint32_t n_tiles = (a.n_lin + (1u << a.tile_log2) - 1u) >> a.tile_log2;
    const uint32_t blocks = (n_tiles + ptw::WAVES_PER_BLOCK - 1) / ptw::WAVES_PER_BLOCK;
    // Measured on MI355X (profiles/r01/lds_vs_scalar.txt): the scalar-load loop beats the LDS-staged loop both on
    // the 11-primitive scene (4.64 vs 6.66 ms) and on the 10 000-primitive scene (223 vs 249 ms), so PT_MATH_FAST
    // always takes the scalar-load loop; the LDS-staged loop runs only when asked for.
    const bool tiled = variant == PT_MATH_FAST_LDS;
    const size_t wave_lds = (size_t)fast_wave_bytes(1u << a.tile_log2) * ptw::WAVES_PER_BLOCK;   // of render_fast_kernel
    if (tiled) {
        if (a.n_mats <= LDS_MAX_MATS)
            hipLaunchKernelGGL(render_fast_tiled_kernel<true>, dim3(blocks), dim3(ptw::BLOCK), 0, s, a, n_tiles);
        else
            hipLaunchKernelGGL(render_fast_tiled_kernel<false>, dim3(blocks), dim3(ptw::BLOCK), 0, s, a, n_tiles);
    } else if (a.n_bnodes > 0) {
        const bool tables = a.n_mats <= LDS_MAX_MATS && a.n_prims <= LDS_MAX_PRIMS;
        if (a.bfat) {
            // the walk stacks: one column of a.walk_stack entries per lane
            const size_t stack_bytes = (size_t)ptw::WAVES_PER_BLOCK * (size_t)a.walk_stack * 64u * sizeof(uint32_t);
            if (a.walk_cam) {
                const size_t cam_lds = stack_bytes + (size_t)walkcam_wave_bytes(1u << a.tile_log2) * ptw::WAVES_PER_BLOCK;
                if (tables) hipLaunchKernelGGL((render_fast_walkcam_kernel<true>), dim3(blocks), dim3(ptw::BLOCK), cam_lds, s, a, n_tiles);
                else hipLaunchKernelGGL((render_fast_walkcam_kernel<false>), dim3(blocks), dim3(ptw::BLOCK), cam_lds, s, a, n_tiles);
            } else
            if (tables) hipLaunchKernelGGL((render_fast_kernel<true, 2, true>), dim3(blocks), dim3(ptw::BLOCK), stack_bytes + wave_lds, s, a, n_tiles);
            else hipLaunchKernelGGL((render_fast_kernel<false, 2, true>), dim3(blocks), dim3(ptw::BLOCK), stack_bytes + wave_lds, s, a, n_tiles);
        } else {
            if (tables) hipLaunchKernelGGL((render_fast_kernel<true, 1, true>), dim3(blocks), dim3(ptw::BLOCK), wave_lds, s, a, n_tiles);
            else hipLaunchKernelGGL((render_fast_kernel<false, 1, true>), dim3(blocks), dim3(ptw::BLOCK), wave_lds, s, a, n_tiles);
        }
    } else {
        const bool tables = a.n_prims <= LDS_MAX_PRIMS && a.n_mats <= LDS_MAX_MATS;
        const bool sorted = variant == PT_MATH_FAST;   // PT_MATH_FAST_SCALAR: the runs, in scene order, plain loop
        // a table that holds short kinds only (or general kinds only) gets the loop without the others' code
        int kinds = 0;
        for (int k = 0; k < FP_KINDS; k++) if (a.sorted.count[k] > 0) kinds |= 1 << k;
        const bool short_only = (kinds & ~KM_SHORT) == 0, general_only = (kinds & ~KM_GENERAL) == 0;
        if (sorted && a.q_cap > 0) {
            // The loop with camera batches and the glass queue.  All of its LDS is dynamic: per-wave arrays sized by the tile, the
            // queue by q_cap, the tables by the scene.
            const size_t table_bytes = tables ? (size_t)a.n_prims * sizeof(FShade) + (size_t)a.n_mats * sizeof(DMat) : 0;
            const size_t lds = ptb::batched_wave_bytes(1u << a.tile_log2, (uint32_t)a.q_cap) * ptw::WAVES_PER_BLOCK + table_bytes;
#define PT_LAUNCH_BATCHED(T, KM) hipLaunchKernelGGL((render_fast_batched_kernel<T, true, KM>), dim3(blocks), dim3(ptw::BLOCK), lds, s, a, n_tiles)
            if (tables) { if (short_only) PT_LAUNCH_BATCHED(true, KM_SHORT); else if (general_only) PT_LAUNCH_BATCHED(true, KM_GENERAL); else PT_LAUNCH_BATCHED(true, KM_ALL); }
            else { if (short_only) PT_LAUNCH_BATCHED(false, KM_SHORT); else if (general_only) PT_LAUNCH_BATCHED(false, KM_GENERAL); else PT_LAUNCH_BATCHED(false, KM_ALL); }
#undef PT_LAUNCH_BATCHED
        } else if (sorted) {
#define PT_LAUNCH_PLAIN(T, KM) hipLaunchKernelGGL((render_fast_kernel<T, 0, true, KM>), dim3(blocks), dim3(ptw::BLOCK), wave_lds, s, a, n_tiles)
            if (tables) { if (short_only) PT_LAUNCH_PLAIN(true, KM_SHORT); else if (general_only) PT_LAUNCH_PLAIN(true, KM_GENERAL); else PT_LAUNCH_PLAIN(true, KM_ALL); }
            else { if (short_only) PT_LAUNCH_PLAIN(false, KM_SHORT); else if (general_only) PT_LAUNCH_PLAIN(false, KM_GENERAL); else PT_LAUNCH_PLAIN(false, KM_ALL); }
#undef PT_LAUNCH_PLAIN
        } else {
            if (tables) hipLaunchKernelGGL((render_fast_kernel<true, 0, false>), dim3(blocks), dim3(ptw::BLOCK), wave_lds, s, a, n_tiles);
            else hipLaunchKernelGGL((render_fast_kernel<false, 0, false>), dim3(blocks), dim3(ptw::BLOCK), wave_lds, s, a, n_tiles);
        }
    }
    PT_CHECK(hipGetLastError());
    return 0;
}

namespace {
__global__ __launch_bounds__(256) void bsdf_fast_kernel(int fn, const float* __restrict__ in, int n, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V3 r = v3(0.0f, 0.0f, 0.0f);
    switch (fn) {
        case PT_FN_SAMPLE_HEMISPHERE: r = sphere_point(in[2 * i], in[2 * i + 1]); break;
        case PT_FN_FRESNEL_SCHLICK: out[i] = ptf::fresnel_schlick(in[3 * i], in[3 * i + 1], in[3 * i + 2]); return;
        case PT_FN_FRESNEL_RATIO: {
            // getFresnelRatio(N, I, ...) only uses N.I, so the argument order of tracePath's call (direction first) does not matter
            const float* a = in + 10 * i;
            out[i] = ptf::fresnel_ratio(v3(a[0], a[1], a[2]), v3(a[3], a[4], a[5]), a[6], a[7], a[8], a[9]);
            return;
        }
        case PT_FN_REFLECT: { const float* a = in + 6 * i; r = ptf::reflect(v3(a[0], a[1], a[2]), v3(a[3], a[4], a[5])); break; }
        case PT_FN_REFRACT: { const float* a = in + 7 * i; r = ptf::refract(v3(a[0], a[1], a[2]), v3(a[3], a[4], a[5]), a[6]); break; }
        case PT_FN_LERP: { const float* a = in + 7 * i; r = ptf::lerp3(v3(a[0], a[1], a[2]), v3(a[3], a[4], a[5]), a[6]); break; }
        default: r = unit_or_zero(v3(in[3 * i], in[3 * i + 1], in[3 * i + 2])); break;
    }
    out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
}
}  // namespace

int ptk_bsdf_batch_fast(int fn, const float* in, int n, float* out, hipStream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(bsdf_fast_kernel, dim3((n + 255) / 256), dim3(256), 0, s, fn, in, n, out);
    PT_CHECK(hipGetLastError());
    return 0;
}

int ptk_nearest_batch_fast(const pt_scene* sc, const float* o, const float* d, int n, int32_t* prim, float* t,
                           float* pos, float* nrm, int32_t* inside, hipStream_t s) {
    if (n <= 0) return 0;
    SceneRefs refs;
    refs.fdata = nullptr; refs.runs = nullptr; refs.n_runs = 0; refs.fat = nullptr;
    if (sc->n_bnodes > 0) {
        refs.sorted = sc->sorted_top; refs.nodes = sc->d_bnodes; refs.bprims = sc->d_bprims; refs.n_nodes = sc->n_bnodes;
        hipLaunchKernelGGL(nearest_fast_kernel<true>, dim3((n + 255) / 256), dim3(256), 0, s, refs, sc->d_fshade, o, d, n, prim, t, pos, nrm, inside);
    } else {
        refs.sorted = sc->sorted; refs.nodes = nullptr; refs.bprims = nullptr; refs.n_nodes = 0;
        hipLaunchKernelGGL(nearest_fast_kernel<false>, dim3((n + 255) / 256), dim3(256), 0, s, refs, sc->d_fshade, o, d, n, prim, t, pos, nrm, inside);
    }
    PT_CHECK(hipGetLastError());
    return 0;
}

int ptk_trace_paths_fast(const pt_scene* sc, const float* o, const float* d, const uint32_t* pixel,
                         const uint32_t* sample, int n, uint32_t seed_hash, int n_bounces, int bsdf_model, float* rgb,
                         float* end_o, float* end_d, hipStream_t s) {
    if (n <= 0) return 0;
    SceneRefs refs;
    if (sc->n_bnodes > 0) {
        refs.fdata = sc->d_fdata_top; refs.runs = sc->d_runs_top; refs.n_runs = sc->n_runs_top;
        refs.nodes = sc->d_bnodes; refs.bprims = sc->d_bprims; refs.n_nodes = sc->n_bnodes; refs.fat = nullptr;
        hipLaunchKernelGGL(trace_fast_kernel<true>, dim3((n + 255) / 256), dim3(256), 0, s, refs,
                           sc->d_fshade, sc->d_mats, o, d, pixel, sample, n, seed_hash, n_bounces, bsdf_model, rgb, end_o, end_d);
    } else {
        refs.fdata = sc->d_fdata; refs.runs = sc->d_runs; refs.n_runs = sc->n_runs;
        refs.nodes = nullptr; refs.bprims = nullptr; refs.n_nodes = 0; refs.fat = nullptr;
        hipLaunchKernelGGL(trace_fast_kernel<false>, dim3((n + 255) / 256), dim3(256), 0, s, refs,
                           sc->d_fshade, sc->d_mats, o, d, pixel, sample, n, seed_hash, n_bounces, bsdf_model, rgb, end_o, end_d);
    }
    PT_CHECK(hipGetLastError());
    return 0;
}
