// Device code that follows the reference's operation order one float operation
// at a time (shapes.h:28-361, bsdf.h:29-168, vec_math.h:10-30).  Translation
// units that include this header are compiled with -ffp-contract=off and the
// default correctly rounded f32 divide/sqrt, so intersection distances come out
// bit for bit as a host restatement computes them; only sinf/cosf/expf differ
// (device libm vs host libm).
#pragma once

#include "pt_internal.h"

#define PTD __device__ __forceinline__

namespace ptp {

struct V3 { float x, y, z; };
PTD V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
PTD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
PTD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
PTD V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
PTD V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
PTD V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }   // cwiseProduct
PTD V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
// Eigen's fixed-size reductions sum as x + (y + z).
PTD float dot(V3 a, V3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
PTD float sqn(V3 a) { return a.x * a.x + (a.y * a.y + a.z * a.z); }
PTD V3 normalized(V3 a) {
    float n = sqn(a);
    if (n > 0.0f) { float s = sqrtf(n); return v3(a.x / s, a.y / s, a.z / s); }
    return a;
}
// Affine3f * point and linear() * vector for a row-major [L | t] block.  (MP: a pointer to floats in any address space.)
template <class MP> PTD V3 xvec(MP m, V3 p) {
    return v3(m[0] * p.x + (m[1] * p.y + m[2] * p.z), m[4] * p.x + (m[5] * p.y + m[6] * p.z),
              m[8] * p.x + (m[9] * p.y + m[10] * p.z));
}
template <class MP> PTD V3 xpoint(MP m, V3 p) {
    V3 l = xvec(m, p);
    return v3(l.x + m[3], l.y + m[7], l.z + m[11]);
}

// ---- counter RNG (stands in for curandState, main.cu:58-60) -----------------
// bits(k) = H(K_smp ^ 0x85EBCA6B*(k+1)), K_smp = H(K_px + 0x9E3779B9*(sample+1)),
// K_px = H(pixel ^ H(seed + 0x9E3779B9)); uniform = bits*2^-32 + 2^-33 in (0,1].
PTD uint32_t rng_hash(uint32_t x) {
    x ^= x >> 16; x *= 0x21f0aaadu; x ^= x >> 15; x *= 0x735a2d97u; x ^= x >> 15;
    return x;
}
PTD uint32_t rng_key(uint32_t seed_hash, uint32_t pixel, uint32_t sample) {
    uint32_t kpx = rng_hash(pixel ^ seed_hash);
    return rng_hash(kpx + 0x9E3779B9u * (sample + 1u));
}
PTD uint32_t rng_bits(uint32_t key, uint32_t slot) { return rng_hash(key ^ (0x85EBCA6Bu * (slot + 1u))); }
PTD float rng_to_uniform(uint32_t bits) { return (float)bits * 2.3283064365386963e-10f + 1.1641532182693481e-10f; }
PTD float rng_u(uint32_t key, uint32_t slot) { return rng_to_uniform(rng_bits(key, slot)); }      // a draw that decides a branch: 32 bits
// The two coordinates of a point (pixel jitter; u1, u2 of a direction) share the bits of the first slot of their pair,
// 16 bits each, cell centres: (0, 1) in 65 536 levels.
PTD float rng_half(uint32_t h16) { return (float)h16 * 1.52587890625e-05f + 7.62939453125e-06f; }
PTD float rng_u_first(uint32_t key, uint32_t first) { return rng_half(rng_bits(key, first) >> 16); }
PTD float rng_u_second(uint32_t key, uint32_t first) { return rng_half(rng_bits(key, first) & 0xffffu); }
// The uniform of slot k under the slot map (0,1 jitter; 2+6s+{0 mode, 1,2 and 3,4 pairs, 5 roulette}).
PTD float rng_slot_uniform(uint32_t key, uint32_t k) {
    const uint32_t j = k < 2u ? k + 1u : (k - 2u) % 6u;
    if (j == 1u || j == 3u) return rng_u_first(key, k);
    if (j == 2u || j == 4u) return rng_u_second(key, k - 1u);
    return rng_u(key, k);
}

// ---- local-space shape tests -------------------------------------------------
// Surface codes let the winner's normal be rebuilt later without carrying it.
enum { SURF_BODY = 0, SURF_DISC_POS = 1, SURF_DISC_NEG = 2 };

struct LHit {
    int n;          // RayPath::n_intersections
    bool h1, h2;    // first/second .intersects
    float t1, t2;   // first/second .distance
    int s1, s2;     // surface codes
};

PTD LHit lhit_none() { LHit h; h.n = 0; h.h1 = false; h.h2 = false; h.t1 = 0.0f; h.t2 = 0.0f; h.s1 = 0; h.s2 = 0; return h; }

// Plane::getRayPath (shapes.h:116-133); out of bounds or parallel => 0 hits.
template <class MP> PTD LHit plane_local(V3 o, V3 d, MP b) {
    LHit h = lhit_none();
    float dz = d.x * 0.0f + (d.y * 0.0f + d.z * 1.0f);
    if (dz != 0.0f) {
        V3 no = -1.0f * o;
        float t = (no.x * 0.0f + (no.y * 0.0f + no.z * 1.0f)) / dz;
        V3 pos = o + d * t;
        if (b[0] < pos.x && b[1] < pos.y && b[2] > pos.x && b[3] > pos.y) {
            h.n = 1; h.h1 = true; h.t1 = t;
        }
    }
    return h;
}

// Cube::getRayPath (shapes.h:171-220).
PTD LHit cube_local(V3 o, V3 d) {
    LHit h = lhit_none();
    float x = 1.0f / d.x, y = 1.0f / d.y, z = 1.0f / d.z;
    float t1 = (-0.5f - o.x) * x, t2 = (0.5f - o.x) * x;
    float t3 = (-0.5f - o.y) * y, t4 = (0.5f - o.y) * y;
    float t5 = (-0.5f - o.z) * z, t6 = (0.5f - o.z) * z;
    float tmin = fmaxf(fmaxf(fminf(t1, t2), fminf(t3, t4)), fminf(t5, t6));
    float tmax = fminf(fminf(fmaxf(t1, t2), fmaxf(t3, t4)), fmaxf(t5, t6));
    if (tmax < 0.0f) return h;
    if (tmin > tmax) return h;
    h.n = 2; h.h1 = true; h.h2 = true; h.t1 = tmin; h.t2 = tmax;
    return h;
}

// Sphere::getRayPath (shapes.h:327-360).  (-b -+ sqrt(disc)) / (2.0*a) is a
// double divide of float operands there; a correctly rounded f32 divide by 2a
// gives the same float.
PTD LHit sphere_local(V3 o, V3 d) {
    LHit h = lhit_none();
    float a = dot(d, d);
    float b = 2.0f * dot(d, o);
    float c = dot(o, o) - 0.25f;
    float disc = b * b - 4 * a * c;
    if (disc < 0.0f) return h;
    float sq = sqrtf(disc);
    float den = 2.0f * a;
    h.n = 2; h.h1 = true; h.h2 = true;
    h.t1 = (-b - sq) / den;
    h.t2 = (-b + sq) / den;
    return h;
}

// Cylinder::getIntersectionDisc (shapes.h:231-255) for disc normal (0,0,nz).
PTD bool cyl_disc(V3 o, V3 d, float nz, float* t_out) {
    V3 n = v3(0.0f, 0.0f, nz);
    V3 p = n * 0.5f;
    float dn = dot(d, n);
    if (dn == 0.0f) return false;
    float t = dot(p - o, n) / dn;
    V3 pos = o + d * t;
    if (sqrtf(sqn(pos - p)) > 0.5f) return false;
    *t_out = t;
    return true;
}

// Cylinder::getRayPath (shapes.h:257-318).
PTD LHit cyl_local(V3 o, V3 d) {
    LHit h = lhit_none();
    float a = d.x * d.x + d.y * d.y;
    float b = 2 * (d.x * o.x + d.y * o.y);
    float c = (o.x * o.x + o.y * o.y) - 0.25f;
    float disc = b * b - 4 * a * c;
    float td1 = 0.0f, td2 = 0.0f;
    bool d1 = cyl_disc(o, d, 1.0f, &td1);
    bool d2 = cyl_disc(o, d, -1.0f, &td2);
    bool c1 = false, c2 = false;
    float tc1 = 0.0f, tc2 = 0.0f;
    if (disc > 0.0f) {
        float sq = sqrtf(disc);
        float den = 2.0f * a;
        tc1 = (-b - sq) / den;
        float z1 = (o + d * tc1).z;
        c1 = (-0.5f < z1 && z1 < 0.5f);
        tc2 = (-b + sq) / den;
        float z2 = (o + d * tc2).z;
        c2 = (-0.5f < z2 && z2 < 0.5f);
    }
    if (d1 && d2) {
        h.n = 2; h.h1 = true; h.t1 = td1; h.s1 = SURF_DISC_POS; h.h2 = true; h.t2 = td2; h.s2 = SURF_DISC_NEG;
    } else if (d1 && disc > 0.0f) {
        h.n = 2; h.h1 = true; h.t1 = td1; h.s1 = SURF_DISC_POS;
        h.h2 = c1 ? c1 : c2; h.t2 = c1 ? tc1 : tc2; h.s2 = SURF_BODY;
    } else if (d2 && disc > 0.0f) {
        h.n = 2; h.h1 = true; h.t1 = td2; h.s1 = SURF_DISC_NEG;
        h.h2 = c1 ? c1 : c2; h.t2 = c1 ? tc1 : tc2; h.s2 = SURF_BODY;
    } else if (disc > 0.0f && c1 && c2) {
        h.n = 2; h.h1 = true; h.t1 = tc1; h.s1 = SURF_BODY; h.h2 = true; h.t2 = tc2; h.s2 = SURF_BODY;
    }
    return h;
}

PTD LHit shape_local(int type, V3 o, V3 d, const float* bounds) {
    switch (type) {
        case PT_PLANE: return plane_local(o, d, bounds);
        case PT_CUBE: return cube_local(o, d);
        case PT_CYLINDER: return cyl_local(o, d);
        default: return sphere_local(o, d);
    }
}

// The epsilon filter of Shape::getIntersections (shapes.h:62-73).  The
// reference compares against the double 0.00001; for a float distance that is
// the same predicate as <= 1e-5f.
PTD void eps_filter(LHit& h) {
    const float EPS = 1e-5f;
    if (h.n == 1 && h.t1 <= EPS) {
        h.n = 0;
    } else if (h.n == 2) {
        bool a = h.t1 <= EPS, b = h.t2 <= EPS;
        if (a && b) {
            h.n = 0;
        } else if (a) {
            h.n = 1; h.h1 = h.h2; h.t1 = h.t2; h.s1 = h.s2;
        } else if (b) {
            h.n = 1;
        }
    }
}

// World position and (unflipped) normal of a local hit at distance t on the
// given surface: the per-class code after `res.first.distance = d`
// (shapes.h:125-129, 158-169, 205-218, 250-252, 276-279, 346-349).
PTD void world_point(const DPrim& P, V3 o, V3 d, float t, int surf, V3* pos_w, V3* nrm_w) {
    V3 wo = v3(P.fwd[3], P.fwd[7], P.fwd[11]);
    V3 pos = o + d * t;
    *pos_w = xpoint(P.fwd, pos);
    V3 nl;
    switch (P.type) {
        case PT_PLANE: {
            float s = 0.0f * d.x + (0.0f * d.y + 1.0f * d.z);
            nl = s > 0.0f ? v3(0.0f, 0.0f, -1.0f) : v3(0.0f, 0.0f, 1.0f);
            break;
        }
        case PT_CUBE: {
            float ax = fabsf(pos.x), ay = fabsf(pos.y), az = fabsf(pos.z);
            float m = fmaxf(ax, fmaxf(ay, az));
            if (ax == m) nl = v3(pos.x, 0.0f, 0.0f);
            else if (ay == m) nl = v3(0.0f, pos.y, 0.0f);
            else nl = v3(0.0f, 0.0f, pos.z);
            break;
        }
        case PT_CYLINDER: {
            if (surf == SURF_DISC_POS) nl = v3(0.0f, 0.0f, 1.0f);
            else if (surf == SURF_DISC_NEG) nl = v3(0.0f, 0.0f, -1.0f);
            else nl = v3(pos.x, pos.y, 0.0f);
            break;
        }
        default: nl = pos; break;
    }
    *nrm_w = normalized(xpoint(P.fwd, nl) - wo);
}

struct Nearest {
    int prim;      // -1 = miss
    float t;
    int surf;
};

// One primitive's part of getNearestIntersection (shapes.h:84-98): its nearest valid intersection replaces the best
// so far if it is nearer.  In scene order that is the whole rule; TIES adds what makes it independent of the order:
// of two hits at the same distance the lower scene index wins, which is what the strict '<' gives in scene order.
template <bool TIES>
PTD void nearest_consider(Nearest& best, V3 o, V3 d, const DPrim& P, int k) {
    V3 ol = xpoint(P.inv, o);
    V3 dl = xvec(P.inv, d);
    LHit h = shape_local(P.type, ol, dl, P.bounds);
    eps_filter(h);
    bool valid; float t; int surf;
    if (h.n == 1) { valid = h.h1; t = h.t1; surf = h.s1; }
    else if (h.n == 2) {
        bool second = h.t2 < h.t1;
        valid = second ? h.h2 : h.h1; t = second ? h.t2 : h.t1; surf = second ? h.s2 : h.s1;
    } else { valid = false; t = 0.0f; surf = 0; }
    if (valid && (best.prim < 0 || t < best.t || (TIES && t == best.t && k < best.prim))) { best.prim = k; best.t = t; best.surf = surf; }
}

// Records that every lane reads at the same, wave-uniform address are read through a pointer to the constant address
// space: such loads are scalar loads (one fetch per wave, operands in scalar registers) whatever else the kernel
// writes to memory.
typedef const DPrim __attribute__((address_space(4)))* KPrimP;

// One record of known shape type for the typed loops: the local test (Plane / Cube / Sphere::getRayPath), the filter
// of Shape::getIntersections (shapes.h:62-73) and the choice between first and second of getNearestIntersection
// (shapes.h:86-92) written without branches — every lane computes every value and the conditions are combined at
// the end, so a record's fields are fetched together and no lane waits for another's branch.  The float operations
// and their order are plane_local's, cube_local's and sphere_local's; values a branch of those would not have
// computed (a quotient by zero, the root of a negative number) are computed here and discarded.
//   two distances t1, t2 that both exist: both <= EPS -> none; only t1 <= EPS -> t2; only t2 <= EPS -> t1;
//   otherwise t2 if t2 < t1, else t1.
PTD float nearer_of_two(float t1, float t2, bool& none) {
    const float EPS = 1e-5f;
    const bool a = t1 <= EPS, b = t2 <= EPS;
    none = a & b;
    return a ? t2 : (b ? t1 : (t2 < t1 ? t2 : t1));
}
template <int TYPE>
PTD void nearest_consider_typed(Nearest& best, V3 o, V3 d, KPrimP P, int k) {
    V3 ol = xpoint(P->inv, o);
    V3 dl = xvec(P->inv, d);
    bool valid; float t; int surf = 0;
    if (TYPE == PT_PLANE) {                                                     // shapes.h:116-133
        // The reference forms d.(0,0,1) and (-o).(0,0,1) as x*0 + (y*0 + z*1).  For finite x and y (the callers send
        // rays with non-finite components down the scene-order loop, which keeps the literal form) the two products
        // are zeros, so the sum is z itself whenever z != 0, and a zero of either sign when z == 0.  A zero dz fails
        // `dz != 0` whatever its sign, and a zero numerator gives t = +-0, which `t <= 1e-5` discards whatever its
        // sign: taking z directly changes no decision and no distance that is used.
        float dz = dl.z;
        t = -ol.z / dz;
        V3 pos = ol + dl * t;
        valid = (dz != 0.0f) & (P->bounds[0] < pos.x) & (P->bounds[1] < pos.y) & (P->bounds[2] > pos.x) & (P->bounds[3] > pos.y) & !(t <= 1e-5f);
    } else if (TYPE == PT_CUBE) {                                               // shapes.h:171-220
        float x = 1.0f / dl.x, y = 1.0f / dl.y, z = 1.0f / dl.z;
        float t1 = (-0.5f - ol.x) * x, t2 = (0.5f - ol.x) * x;
        float t3 = (-0.5f - ol.y) * y, t4 = (0.5f - ol.y) * y;
        float t5 = (-0.5f - ol.z) * z, t6 = (0.5f - ol.z) * z;
        float tmin = fmaxf(fmaxf(fminf(t1, t2), fminf(t3, t4)), fminf(t5, t6));
        float tmax = fminf(fminf(fmaxf(t1, t2), fmaxf(t3, t4)), fmaxf(t5, t6));
        bool none;
        t = nearer_of_two(tmin, tmax, none);
        valid = !(tmax < 0.0f) & !(tmin > tmax) & !none;
    } else if (TYPE == PT_SPHERE) {                                             // shapes.h:327-360
        float a = dot(dl, dl);
        float b = 2.0f * dot(dl, ol);
        float c = dot(ol, ol) - 0.25f;
        float disc = b * b - 4 * a * c;
        // (most waves miss a small sphere with every lane: the root and the two quotients are then skipped)
        if (__ballot(!(disc < 0.0f)) == 0ull) return;
        float sq = sqrtf(disc);
        float den = 2.0f * a;
        bool none;
        t = nearer_of_two((-b - sq) / den, (-b + sq) / den, none);
        valid = !(disc < 0.0f) & !none;
    } else {
        LHit h = cyl_local(ol, dl);
        eps_filter(h);
        if (h.n == 1) { valid = h.h1; t = h.t1; surf = h.s1; }
        else if (h.n == 2) {
            bool second = h.t2 < h.t1;
            valid = second ? h.h2 : h.h1; t = second ? h.t2 : h.t1; surf = second ? h.s2 : h.s1;
        } else { valid = false; t = 0.0f; surf = 0; }
    }
    // shapes.h:93 with the scene index deciding ties: take if no hit yet, or t < best.t, or t == best.t and k lower.
    // Here t is a number above 1e-5 (the filter passed; the callers keep rays that could produce a NaN out of these
    // loops), positive floats order like their bit patterns, and "no hit yet" is held as (+inf, INT_MAX): the three
    // cases are one comparison of (t, k) pairs as 64-bit integers.
    const unsigned long long kc = ((unsigned long long)__float_as_uint(t) << 32) | (unsigned)k;
    const unsigned long long kb = ((unsigned long long)__float_as_uint(best.t) << 32) | (unsigned)best.prim;
    const bool take = valid & (kc < kb);
    best.prim = take ? k : best.prim;
    best.t = take ? t : best.t;
    best.surf = take ? surf : best.surf;
}

// A ray whose distances can come out as NaN: a zero direction (refract's zero vector on total reflection,
// vec_math.h:27, survives .normalized()) or a component that is not finite.  With a NaN in play the strict '<' of
// shapes.h:93 depends on the order of the tests, so such rays are searched in scene order.  The bounds are far
// inside the float range (a direction is a unit vector or zero; origins beyond 1e5 take the scene-order loop), so
// that with the transforms the typed loops are used for (entries below 1e5, checked at upload) every local
// coordinate, square and discriminant is a finite number: no test of the typed loops can produce a NaN distance.
PTD bool degenerate_ray(V3 o, V3 d) {
    const float dd = sqn(d), oo = sqn(o);
    return !(dd > 0.0f && dd < 4.0f && oo < 1e10f);
}

// getNearestIntersection over the table sorted by shape type: one typed loop per type, records [first, last) of each;
// `mask` (bit s = record s of the sorted table) leaves out records no ray of the caller can hit.  For rays whose
// distances are numbers the (distance, scene index) rule gives the loop's result in any order of the tests; a wave
// that holds a degenerate ray takes the loop in scene order.
template <bool MASKED>
PTD Nearest nearest_scan_typed(V3 o, V3 d, const PScene& S, unsigned long long mask) {
    Nearest best; best.prim = -1; best.t = 0.0f; best.surf = 0;
    if (__ballot(degenerate_ray(o, d)) != 0ull) {
        for (int k = 0; k < S.n_prims; k++) nearest_consider<false>(best, o, d, S.prims[k], k);
        return best;
    }
    best.prim = 0x7fffffff; best.t = __builtin_inff();       // "no hit yet" for nearest_consider_typed
#define PT_TYPE_LOOP(TYPE)                                                                                  \
    if (MASKED) {                                                                                            \
        const int first = S.type_first[TYPE], cnt = S.type_first[TYPE + 1] - first;                         \
        unsigned long long m = cnt > 0 ? (mask >> first) & (cnt >= 64 ? ~0ull : ((1ull << cnt) - 1ull)) : 0ull; \
        for (; m != 0ull; m &= m - 1ull) {                                                                   \
            const KPrimP P = (KPrimP)S.by_type + (first + __builtin_ctzll(m));                               \
            nearest_consider_typed<TYPE>(best, o, d, P, P->index);                                           \
        }                                                                                                    \
    } else {                                                                                                 \
        KPrimP P = (KPrimP)S.by_type + S.type_first[TYPE];                                                   \
        for (int n = S.type_first[TYPE + 1] - S.type_first[TYPE]; n > 0; n--, P++)                           \
            nearest_consider_typed<TYPE>(best, o, d, P, P->index);                                           \
    }
    PT_TYPE_LOOP(PT_PLANE)
    PT_TYPE_LOOP(PT_CUBE)
    PT_TYPE_LOOP(PT_CYLINDER)
    PT_TYPE_LOOP(PT_SPHERE)
#undef PT_TYPE_LOOP
    if (best.prim == 0x7fffffff) { best.prim = -1; best.t = 0.0f; }
    return best;
}

// The loop of getNearestIntersection (shapes.h:79-100): k is wave-uniform, so
// prims[k] is fetched with scalar loads and the type switch is a scalar branch.
// A scene with a bounding-volume hierarchy (S.nodes) is searched through it instead: the primitives outside it in
// that same loop, the others per lane through the skip-link tree, every primitive with the same arithmetic as in the
// loop.  A box is entered if the ray meets it no later than the best hit so far, boxes are padded beyond the rounding
// of this test, and ties go to the lower scene index, so the result is the loop's.
PTD Nearest nearest_scan(V3 o, V3 d, const PScene& S) {
    Nearest best; best.prim = -1; best.t = 0.0f; best.surf = 0;
    if (S.nodes == nullptr) {
        for (int k = 0; k < S.n_prims; k++) nearest_consider<false>(best, o, d, S.prims[k], k);
        return best;
    }
    for (int i = 0; i < S.n_top; i++) { const int k = S.top[i]; nearest_consider<false>(best, o, d, S.prims[k], k); }
    // (a lane without a path carries a zero ray, which no test accepts; it would enter every box that holds the origin)
    if (d.x == 0.0f && d.y == 0.0f && d.z == 0.0f) return best;
    const V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    int n = 0;
    while (n < S.n_nodes) {
        const BNode& N = S.nodes[n];
        // slab test; fminf/fmaxf drop a NaN operand (0 * inf when the origin lies on a slab plane of an axis the ray
        // does not move along), which leaves that axis unconstrained on one side: conservative
        const float ax = (N.lo[0] - o.x) * inv.x, bx = (N.hi[0] - o.x) * inv.x;
        const float ay = (N.lo[1] - o.y) * inv.y, by = (N.hi[1] - o.y) * inv.y;
        const float az = (N.lo[2] - o.z) * inv.z, bz = (N.hi[2] - o.z) * inv.z;
        const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
        float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
        if (best.prim >= 0) tf = fminf(tf, best.t);
        const bool hit = tn <= tf;
        const int leaf = N.leaf;
        if (hit && leaf >= 0) {
            const BPrim* q = S.bprims + (leaf >> 4);
            for (int c = leaf & 15; c > 0; c--, q++) { const int k = q->id >> 3; nearest_consider<true>(best, o, d, S.prims[k], k); }
        }
        n = (hit && leaf < 0) ? n + 1 : N.skip;
    }
    return best;
}

struct Hit {
    int prim;      // -1 = miss
    float t;
    V3 pos, nrm;
    bool inside;
};

// The winner's record (shapes.h:93-97) from the search's result.
PTD Hit finish_hit(const Nearest& b, V3 o, V3 d, const DPrim* prims) {
    Hit h; h.prim = b.prim; h.t = b.t; h.pos = v3(0, 0, 0); h.nrm = v3(0, 0, 0); h.inside = false;
    if (b.prim >= 0) {
        const DPrim& P = prims[b.prim];
        V3 ol = xpoint(P.inv, o);
        V3 dl = xvec(P.inv, d);
        world_point(P, ol, dl, b.t, b.surf, &h.pos, &h.nrm);
        h.inside = dot(h.nrm, d) > 0.0f;
        if (h.inside) h.nrm = -1.0f * h.nrm;
    }
    return h;
}
// getNearestIntersection including the winner's record.
PTD Hit nearest_hit(V3 o, V3 d, const PScene& S) { return finish_hit(nearest_scan(o, d, S), o, d, S.prims); }

// ---- bsdf.h ---------------------------------------------------------------------
PTD V3 reflect(V3 v, V3 n) { return v - 2 * (dot(v, n) * n); }                  // vec_math.h:19-21
PTD V3 refract(V3 I, V3 N, float eta) {                                        // vec_math.h:23-30
    float ndi = dot(N, I);
    double k = 1.0 - (double)(eta * eta) * (1.0 - (double)(ndi * ndi));
    float kf = (float)k;
    if (kf < 0.0f) return v3(0.0f, 0.0f, 0.0f);
    return eta * I - (eta * ndi + sqrtf(kf)) * N;
}
PTD V3 lerp3(V3 v0, V3 v1, float t) { return (1 - t) * v0 + t * v1; }           // vec_math.h:10-12
PTD float lerp1(float v0, float v1, float t) { return (1 - t) * v0 + t * v1; }

PTD V3 sample_hemisphere(float u1, float u2) {                                  // bsdf.h:29-36
    float z = u1 * 2.0f - 1.0f;
    float a = (float)((double)(u2 * 2) * 3.14159265358979323846);
    float r = sqrtf(1.0f - z * z);
    return v3(r * cosf(a), r * sinf(a), z);
}

PTD float fresnel_schlick(float n1, float n2, float c) {                        // bsdf.h:48-51
    float q = (n1 - n2) / (n1 + n2);
    float r0 = q * q;
    float m = 1 - c;
    float m2 = m * m;
    float m5 = (m2 * m2) * m;
    return r0 + (1 - r0) * m5;
}

PTD float fresnel_ratio(V3 N, V3 I, float f0, float f90, float n1, float n2) {  // bsdf.h:53-69
    float cX = -dot(N, I);
    if (n1 > n2) {
        float n = n1 / n2;
        float s2 = (float)((double)(n * n) * (1.0 - (double)(cX * cX)));
        if (s2 > 1.0f) return f90;
        cX = (float)sqrt(1.0 - (double)s2);
    }
    return lerp1(f0, f90, fresnel_schlick(n1, n2, cX));
}

struct Path {
    V3 o, d;          // current ray
    V3 coef;          // throughput ("coefficient", bsdf.h:76)
    V3 contrib;       // radiance gathered so far ("contribution", bsdf.h:74)
    uint32_t key;     // K_smp of this sample
    int seg;          // loop index i of bsdf.h:78
};

// One iteration of the tracePath loop (bsdf.h:78-166).  Returns true when the
// path goes on to another segment.  Draw slots of segment s are 2+6s+{0..5};
// draws whose value cannot change the result are not evaluated.
// One exit and unconditional updates of o, d, coef: a lane whose path ends here is refilled before that state is read
// again, so it may hold anything, and the old and new state never have to coexist in registers.  A miss goes through
// the arithmetic with primitive 0's material and is dropped at the end; the arithmetic of a live path is untouched.
PTD void step_emissive(Path& p, bool hit, const DMat& M) {                      // bsdf.h:90-92
    V3 emissive = v3(M.emissive[0], M.emissive[1], M.emissive[2]);
    if (hit && sqn(emissive) > 0.0f) p.contrib = p.contrib + emissive * p.coef;
}

// bsdf.h:95-165 for the hit `h` (h.prim: whether there is one) on material M; true = the path goes on.
PTD bool step_scatter(Path& p, const Hit& h, bool hit, const DMat& M, int n_bounces) {
    if (p.seg > 0 && h.inside) {
        V3 rc = v3(M.refractionColor[0], M.refractionColor[1], M.refractionColor[2]);
        V3 att = (-1.0f * rc) * h.t;
        p.coef = p.coef * v3(expf(att.x), expf(att.y), expf(att.z));
    }
    float f0 = M.f0, transparency = M.transparency, ior = M.IOR;
    float p_specular = f0, p_refract = transparency;
    if (f0 > 0.0f) {
        p_specular = fresnel_ratio(p.d, h.nrm, f0, 1.0f, h.inside ? ior : 1.0f, !h.inside ? ior : 1.0f);
        p_refract = transparency * (1.0f - p_specular) / (1.0f - f0);
    }
    uint32_t slot = 2u + 6u * (uint32_t)p.seg;
    int mode = 3;
    float p_ray;
    {
        bool need_u = p_specular > 0.0f || p_refract > 0.0f;
        float u = need_u ? rng_u(p.key, slot) : 2.0f;
        if (p_specular > 0.0f && u < p_specular) { mode = 1; p_ray = p_specular; }
        else if (p_refract > 0.0f && u < p_specular + p_refract) { mode = 2; p_ray = p_refract; }
        else { mode = 3; p_ray = 1.0f - (p_specular + p_refract); }
    }
    p_ray = fmaxf(p_ray, 0.001f);
    V3 ndir;
    if (mode == 2) {
        V3 rd = normalized(refract(p.d, h.nrm, h.inside ? ior : 1.0f / ior));
        float rr = M.refractionRoughness * M.refractionRoughness;
        if (rr != 0.0f) {
            V3 hd = normalized(h.nrm + sample_hemisphere(rng_u_first(p.key, slot + 3u), rng_u_second(p.key, slot + 3u)));
            rd = lerp3(rd, hd, rr);
        }
        ndir = normalized(rd);
    } else {
        V3 dd = normalized(h.nrm + sample_hemisphere(rng_u_first(p.key, slot + 1u), rng_u_second(p.key, slot + 1u)));
        if (mode == 1) {
            V3 sd = normalized(reflect(p.d, h.nrm));
            float sr = M.specularRoughness * M.specularRoughness;
            sd = lerp3(sd, dd, sr);
            ndir = normalized(sd);
        } else {
            ndir = dd;
        }
    }
    if (mode == 1) p.coef = p.coef * v3(M.specularColor[0], M.specularColor[1], M.specularColor[2]);
    else if (mode == 3) p.coef = p.coef * v3(M.albedo[0], M.albedo[1], M.albedo[2]);
    p.coef = p.coef / p_ray;
    float q = fmaxf(p.coef.x, fmaxf(p.coef.y, p.coef.z));
    const bool dead = !hit || rng_u(p.key, slot + 5u) > q;
    p.coef = (p.coef * 1.0f) / q;
    p.o = h.pos;
    p.d = ndir;
    p.seg++;
    return !dead && p.seg <= n_bounces;
}

// What step_scatter computes for a material with f0 = 0 and transparency = 0 hit from outside (or on the first
// segment): no Beer term, p_specular = p_refract = 0, no mode draw, mode 3, p_ray = max(1 - (0 + 0), 0.001) = 1 and
// coef / 1 = coef, (coef * 1) / q = coef / q exactly.  The operations that remain are step_scatter's, in its order, so
// the path comes out bit for bit the same.  `keep` lanes (hits that need step_scatter) come out unchanged.
PTD bool step_diffuse(Path& p, const Hit& h, bool hit, bool keep, const DMat& M, int n_bounces) {
    const uint32_t slot = 2u + 6u * (uint32_t)p.seg;
    const V3 dd = normalized(h.nrm + sample_hemisphere(rng_u_first(p.key, slot + 1u), rng_u_second(p.key, slot + 1u)));
    V3 coef = p.coef * v3(M.albedo[0], M.albedo[1], M.albedo[2]);
    const float q = fmaxf(coef.x, fmaxf(coef.y, coef.z));
    const bool dead = !hit || rng_u(p.key, slot + 5u) > q;
    coef = coef / q;
    p.coef = v3(keep ? p.coef.x : coef.x, keep ? p.coef.y : coef.y, keep ? p.coef.z : coef.z);
    p.d = v3(keep ? p.d.x : dd.x, keep ? p.d.y : dd.y, keep ? p.d.z : dd.z);
    p.o = v3(keep ? p.o.x : h.pos.x, keep ? p.o.y : h.pos.y, keep ? p.o.z : h.pos.z);
    p.seg += keep ? 0 : 1;
    return !dead && p.seg <= n_bounces;
}

PTD bool path_step(Path& p, const PScene& S, const DMat* __restrict__ mats, int n_bounces) {
    const DPrim* prims = S.prims;
    Hit h = nearest_hit(p.o, p.d, S);
    const bool hit = h.prim >= 0;
    const DMat& M = mats[prims[hit ? h.prim : 0].material];
    step_emissive(p, hit, M);
    return step_scatter(p, h, hit, M, n_bounces);
}

// ---- bsdf_alternative.h: the class-based BSDFs and their tracePath -------------------------
// Draw slots of segment s: 2+6s+{0 dielectric choice, 1,2 Lambertian u1,u2, 5 Russian roulette}.
PTD void construct_onb(V3 N, V3* x, V3* y) {                                    // bsdf_alternative.h:9-18
    float sign = (N.z < 0 ? -1.0f : 1.0f);
    float a = -1 / (sign + N.z);
    float b = N.x * N.y * a;
    *x = v3(1 + sign * (N.x * N.x) * a, sign * b, -sign * N.x);
    *y = v3(b, sign + (N.y * N.y) * a, -N.y);
}

// One iteration of the loop of bsdf_alternative.h:158-217; true = the path goes on.  The switch's `default:` case
// (:195-199, which leaves the outer variables unset) is taken as Lambertian, and a SmoothDielectric with neither
// reflection nor transmission (:131-132, `*transmission` unset) as transmission = false.
PTD bool path_step_alt(Path& p, const PScene& S, const DMat* __restrict__ mats, int n_bounces) {
    const DPrim* prims = S.prims;
    Hit h = nearest_hit(p.o, p.d, S);
    if (h.prim < 0) return false;
    const DMat& M = mats[prims[h.prim].material];
    V3 emissive = v3(M.emissive[0], M.emissive[1], M.emissive[2]);
    if (sqn(emissive) > 0.0f) p.contrib = p.contrib + emissive * p.coef;
    const uint32_t slot = 2u + 6u * (uint32_t)p.seg;
    V3 ndir, f;
    float pdf;
    if (M.bsdfType == 1) {                                                      // SmoothDielectric, :116-149
        float ior = M.IOR;
        float pr = fresnel_ratio(h.nrm, p.d, M.f0, 1.0f, h.inside ? ior : 1.0f, !h.inside ? 1.0f : ior);
        float pt = 1 - pr;
        pr *= (M.f0 > 0) ? 1.0f : 0.0f;
        pt *= (M.transparency > 0) ? 1.0f : 0.0f;
        bool transmission = false;
        if (pr == 0.0f && pt == 0.0f) {
            ndir = v3(0.0f, 0.0f, 0.0f);
        } else {
            float u = rng_u(p.key, slot);
            if (u < pr / (pr + pt)) {
                ndir = normalized(reflect(p.d, h.nrm));
            } else {
                transmission = true;
                ndir = normalized(refract(p.d, h.nrm, h.inside ? ior : 1.0f / ior));
            }
        }
        ndir = normalized(ndir);
        f = v3(M.specularColor[0], M.specularColor[1], M.specularColor[2]);
        pdf = transmission ? pt : 1 - pt;
    } else {                                                                    // Lambertian, :74-114
        float u1 = rng_u_first(p.key, slot + 1u), u2 = rng_u_second(p.key, slot + 1u);
        V3 ox, oy;
        construct_onb(h.nrm, &ox, &oy);
        V3 res = sample_hemisphere(u1, u2);
        ndir = normalized((res.x * ox + res.y * oy) + res.z * h.nrm);
        V3 V = -1.0f * p.d;
        float NDotL = dot(h.nrm, ndir), NDotV = dot(h.nrm, V);
        if (NDotL <= 0.0f || NDotV <= 0.0f) {
            f = v3(0.0f, 0.0f, 0.0f);
        } else {
            V3 out = (float)(1.0f / 3.14159265358979323846) * v3(M.albedo[0], M.albedo[1], M.albedo[2]);
            f = out * fmaxf(fminf(NDotL, 1.0f), 0.0f);
        }
        pdf = (float)((double)fabsf(dot(ndir, h.nrm)) * (1.0f / 3.14159265358979323846));
    }
    if (pdf > 0.0f) p.coef = p.coef * (f / pdf);
    else return false;
    float pm = fmaxf(p.coef.x, fmaxf(p.coef.y, p.coef.z));
    float u = rng_u(p.key, slot + 5u);
    if (pm < u) return false;
    p.coef = p.coef * (1 / pm);
    p.o = h.pos;
    p.d = ndir;
    p.seg++;
    return p.seg <= n_bounces;
}

// Camera ray of (pixel i,j; sample) exactly as render builds it (main.cu:62-64); ci = i - 0.5f and
// cj = max_y - j - 0.5f are the caller's (both exact in float).
PTD void camera_ray_at(const RenderArgs& a, float ci, float cj, uint32_t key, V3* o, V3* d) {
    float jx = rng_u_first(key, 0u), jy = rng_u_second(key, 0u);
    float sx = (ci + jx), sy = (cj + jy);
    V3 pc = v3((a.p00[0] + a.du * sx) + 0.0f * sy, (a.p00[1] + 0.0f * sx) + a.dv * sy, (a.p00[2] + 0.0f * sx) + 0.0f * sy);
    V3 cam = v3(a.cam[0], a.cam[1], a.cam[2]);
    *o = cam;
    *d = normalized(pc - cam);
}
PTD void camera_ray(const RenderArgs& a, int i, int j, uint32_t key, V3* o, V3* d) { camera_ray_at(a, i - 0.5f, a.max_y - j - 0.5f, key, o, d); }

}  // namespace ptp
