// Host side of libpt_hip.so: the C entry points of include/pt_hip.h, the scene
// flattening that replaces constructScene (main.cu:71-198) and the camera set-up
// of render (main.cu:40-52).  Built with -ffp-contract=off: the table values and
// camera constants must round exactly as a plain float evaluation does.
#include "pt_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

namespace {

thread_local char g_err[512] = "";

const double PT_PI = 3.14159265358979323846;   // vec_math.h:5

// ---- small affine helpers (the Eigen calls of shapes.h:36-38, main.cu:73-134) ----
struct Aff {
    float m[3][4];
};
Aff aff_identity() {
    Aff a;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) a.m[i][j] = (i == j) ? 1.0f : 0.0f;
    return a;
}
void aff_translate(Aff& a, float x, float y, float z) { a.m[0][3] = x; a.m[1][3] = y; a.m[2][3] = z; }
// Affine3f::scale: post-multiplies the linear block, translation untouched.
void aff_scale(Aff& a, float sx, float sy, float sz) {
    const float s[3] = {sx, sy, sz};
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) a.m[i][j] *= s[j];
}
// AngleAxisf(angle, axis).toRotationMatrix() written into the linear block.
void aff_rotate(Aff& a, float angle, float ax, float ay, float az) {
    const float s = sinf(angle), c = cosf(angle);
    const float sx = s * ax, sy = s * ay, sz = s * az;
    const float c1 = 1.0f - c;
    const float cx = c1 * ax, cy = c1 * ay, cz = c1 * az;
    float tmp;
    tmp = cx * ay; a.m[0][1] = tmp - sz; a.m[1][0] = tmp + sz;
    tmp = cx * az; a.m[0][2] = tmp + sy; a.m[2][0] = tmp - sy;
    tmp = cy * az; a.m[1][2] = tmp - sx; a.m[2][1] = tmp + sx;
    a.m[0][0] = cx * ax + c; a.m[1][1] = cy * ay + c; a.m[2][2] = cz * az + c;
}
// Affine inverse: cofactor inverse of the 3x3 block, translation = (-Linv) * t.
Aff aff_inverse(const Aff& a) {
    auto cof = [&](int i, int j) {
        const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
        return a.m[i1][j1] * a.m[i2][j2] - a.m[i1][j2] * a.m[i2][j1];
    };
    const float c00 = cof(0, 0), c10 = cof(1, 0), c20 = cof(2, 0);
    const float det = c00 * a.m[0][0] + (c10 * a.m[1][0] + c20 * a.m[2][0]);
    const float invdet = 1.0f / det;
    Aff r;
    for (int row = 0; row < 3; row++)
        for (int col = 0; col < 3; col++) r.m[row][col] = cof(col, row) * invdet;
    for (int row = 0; row < 3; row++)
        r.m[row][3] = (-r.m[row][0]) * a.m[0][3] + ((-r.m[row][1]) * a.m[1][3] + (-r.m[row][2]) * a.m[2][3]);
    return r;
}
void aff_store(const Aff& a, float* out) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) out[i * 4 + j] = a.m[i][j];
}
Aff aff_load(const float* in) {
    Aff a;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) a.m[i][j] = in[i * 4 + j];
    return a;
}

void mat_defaults(pt_material* m) {   // materials.h:12-38
    std::memset(m, 0, sizeof(*m));
    m->IOR = 1.0f;
}
void set3(float* d, float x, float y, float z) { d[0] = x; d[1] = y; d[2] = z; }

void put_prim(pt_prim* p, int type, int mat, const Aff& t, float b) {
    p->type = type; p->material = mat;
    aff_store(t, p->transform);
    p->bounds[0] = -b; p->bounds[1] = -b; p->bounds[2] = b; p->bounds[3] = b;
}

// The eight materials the default scene uses (main.cu:138-183):
// 0 base, 1 red, 2 green, 3 light, 4 glass1, 5 glass2, 6 glass3, 7 diffuse_pure.
void default_materials(pt_material* mats) {
    for (int i = 0; i < 8; i++) mat_defaults(&mats[i]);
    set3(mats[0].albedo, 0.4f, 0.4f, 0.4f);
    set3(mats[1].albedo, 1.0f, 0.0f, 0.0f);
    set3(mats[2].albedo, 0.0f, 1.0f, 0.0f);
    set3(mats[3].emissive, 1.0f * 20.0f, 0.9f * 20.0f, 0.7f * 20.0f);
    for (int g = 4; g <= 6; g++) {
        mats[g].f0 = 0.02f; mats[g].specularRoughness = 0.0f;
        set3(mats[g].specularColor, 1.0f * 0.8f, 1.0f * 0.8f, 1.0f * 0.8f);
        mats[g].IOR = 1.5f; mats[g].transparency = 1.0f; mats[g].refractionRoughness = 0.0f; mats[g].bsdfType = 1;
    }
    set3(mats[5].refractionColor, 0.9f, 0.0f, 0.0f);
    set3(mats[6].refractionColor, 2.0f, 0.0f, 0.0f);
    set3(mats[7].albedo, 0.6f, 0.6f, 0.6f);
}

// The seven walls of the default scene (main.cu:77-108, 185-191); returns 7.
int default_walls(pt_prim* prims) {
    const float HALF_PI = (float)(PT_PI / 2), NEG_HALF_PI = (float)(-PT_PI / 2), FPI = (float)PT_PI;
    Aff floor = aff_identity(); aff_rotate(floor, HALF_PI, 1, 0, 0); aff_translate(floor, 0.0f, -1.0f, 0.0f);
    Aff back = aff_identity(); aff_translate(back, 0.0f, 0.0f, -5.0f);
    Aff left = aff_identity(); aff_rotate(left, HALF_PI, 0, 1, 0); aff_translate(left, -1.0f, 0.0f, -5.0f);
    Aff right = aff_identity(); aff_rotate(right, NEG_HALF_PI, 0, 1, 0); aff_translate(right, 1.0f, 0.0f, -5.0f);
    Aff ceiling = aff_identity(); aff_rotate(ceiling, NEG_HALF_PI, 1, 0, 0); aff_translate(ceiling, 0.0f, 1.0f, 0.0f);
    Aff light = aff_identity(); aff_rotate(light, NEG_HALF_PI, 1, 0, 0); aff_translate(light, 0.0f, 0.97f, -3.25f);
    Aff front = aff_identity(); aff_rotate(front, FPI, 0, 1, 0); aff_translate(front, 0.0f, 0.0f, 7.0f);
    put_prim(&prims[0], PT_PLANE, 0, floor, 15.0f);
    put_prim(&prims[1], PT_PLANE, 1, left, 15.0f);
    put_prim(&prims[2], PT_PLANE, 2, right, 15.0f);
    put_prim(&prims[3], PT_PLANE, 0, back, 15.0f);
    put_prim(&prims[4], PT_PLANE, 0, ceiling, 15.0f);
    put_prim(&prims[5], PT_PLANE, 3, light, 0.5f);
    put_prim(&prims[6], PT_PLANE, 0, front, 15.0f);
    return 7;
}

uint32_t host_hash(uint32_t x) {
    x ^= x >> 16; x *= 0x21f0aaadu; x ^= x >> 15; x *= 0x735a2d97u; x ^= x >> 15;
    return x;
}
float host_uniform(uint32_t key, uint32_t slot) {
    const uint32_t bits = host_hash(key ^ (0x85EBCA6Bu * (slot + 1u)));
    return (float)bits * 2.3283064365386963e-10f + 1.1641532182693481e-10f;
}

float len3(float x, float y, float z) { return sqrtf(x * x + y * y + z * z); }

// Kind, intersection record and shading record of the fast kernels for one
// shape.  Returns the record length in floats.
int bake_fast(const pt_prim& p, const Aff& fwd, const Aff& inv, int general, float* rec, FShade* fs, int32_t* ftype) {
    std::memset(fs, 0, sizeof(*fs));
    fs->type = p.type; fs->material = p.material;
    // Columns of L and the unit normals along them.  The reference forms a normal as normalize(T*axis - wOrigin)
    // (shapes.h:127,161-167): adding the translation and taking it off again swallows components far below it, such as
    // the 4e-8 a rotation by (float)(pi/2) leaves where a zero belongs.  The same arithmetic here gives the same signs,
    // which matters wherever shading branches on the sign of a normal component.
    float col[3][3], ncol[3][3];
    for (int j = 0; j < 3; j++) {
        float n[3];
        for (int i = 0; i < 3; i++) { col[j][i] = fwd.m[i][j]; n[i] = (fwd.m[i][j] + fwd.m[i][3]) - fwd.m[i][3]; }
        const float l = sqrtf(n[0] * n[0] + (n[1] * n[1] + n[2] * n[2]));
        for (int i = 0; i < 3; i++) ncol[j][i] = l > 0.0f ? n[i] / l : n[i];
    }
    auto put_inv = [&]() { for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) rec[i * 4 + j] = inv.m[i][j]; };
    switch (p.type) {
        case PT_PLANE: {
            for (int i = 0; i < 3; i++) fs->a[i] = ncol[2][i];
            // Fold the bounds into rows x and y: x' = (x - cx)/hx, so xMin < x < xMax <=> |x'| < 1.
            // Bounds beyond +-1e30 count as unbounded; a half-bounded axis keeps the explicit test.
            const double INF = 1e30;
            double sc[2], cc[2];
            bool foldable = true;
            for (int ax = 0; ax < 2; ax++) {
                const double lo = p.bounds[ax], hi = p.bounds[2 + ax];
                const bool lo_inf = lo < -INF, hi_inf = hi > INF;
                if (lo_inf && hi_inf) { sc[ax] = 0.0; cc[ax] = 0.0; }
                else if (!lo_inf && !hi_inf && hi > lo) { sc[ax] = 2.0 / (hi - lo); cc[ax] = 0.5 * (hi + lo); }
                else foldable = false;
            }
            for (int j = 0; j < 4; j++) rec[j] = inv.m[2][j];
            if (foldable) {
                *ftype = FP_PLANE;
                for (int ax = 0; ax < 2; ax++) {
                    for (int j = 0; j < 3; j++) rec[4 + 4 * ax + j] = (float)(inv.m[ax][j] * sc[ax]);
                    rec[4 + 4 * ax + 3] = (float)((inv.m[ax][3] - cc[ax]) * sc[ax]);
                }
                // A plane whose normal and edges run along world axes (every row above has one entry that matters)
                // gets the short test: t = (k - o_n)/d_n, then two interval tests on single coordinates.  What a
                // rotation by (float)(pi/2) or (float)pi leaves where a zero belongs, 4e-8 to 9e-8 of the row, counts
                // as zero; the limit is 2e-7, the size of a float rounding.
                int axis[3] = {-1, -1, -1};
                bool aligned = (general & PT_GENERAL_PLANES) == 0;
                for (int r = 0; r < 3 && aligned; r++) {
                    const float* row = rec + 4 * r;
                    int a = 0;
                    for (int j = 1; j < 3; j++) if (fabsf(row[j]) > fabsf(row[a])) a = j;
                    if (row[a] == 0.0f) { if (r == 0) aligned = false; continue; }   // an unbounded edge direction: no entry at all
                    for (int j = 0; j < 3; j++) if (j != a && fabsf(row[j]) > 2e-7f * fabsf(row[a])) aligned = false;
                    axis[r] = a;
                }
                if (aligned) {
                    // edge rows without an entry take the axes that are left
                    for (int r = 1; r < 3; r++) if (axis[r] < 0) for (int a = 0; a < 3; a++) if (a != axis[0] && a != axis[1] && a != axis[2]) { axis[r] = a; break; }
                    aligned = axis[0] != axis[1] && axis[0] != axis[2] && axis[1] != axis[2];
                }
                if (aligned) {
                    const int ru = axis[1] < axis[2] ? 1 : 2, rv = 3 - ru;   // u < v
                    float out[5];
                    out[0] = (float)(-(double)rec[3] / (double)rec[axis[0]]);
                    const int rows[2] = {ru, rv};
                    for (int e = 0; e < 2; e++) {
                        const float* row = rec + 4 * rows[e];
                        const float a = row[axis[rows[e]]];
                        if (a == 0.0f) { out[1 + 2 * e] = 0.0f; out[2 + 2 * e] = INFINITY; }
                        else { out[1 + 2 * e] = (float)(-(double)row[3] / (double)a); out[2 + 2 * e] = (float)(1.0 / fabs((double)a)); }
                    }
                    *ftype = FP_PLANE_X + axis[0];
                    for (int j = 0; j < 5; j++) rec[j] = out[j];
                    for (int j = 5; j < 16; j++) rec[j] = 0.0f;
                    return 5;
                }
                return 12;
            }
            *ftype = FP_PLANE_GEN;
            for (int j = 0; j < 4; j++) { rec[4 + j] = inv.m[0][j]; rec[8 + j] = inv.m[1][j]; rec[12 + j] = p.bounds[j]; }
            return 16;
        }
        case PT_SPHERE: {
            fs->a[0] = fwd.m[0][3]; fs->a[1] = fwd.m[1][3]; fs->a[2] = fwd.m[2][3];
            // L = s * I exactly (what Affine3f::scale(s) on an unrotated transform gives): the local frame is a scale
            // and an offset, read off T^-1 as the reference computes it
            const bool scaled_identity = fwd.m[0][0] != 0.0f && fwd.m[0][0] == fwd.m[1][1] && fwd.m[0][0] == fwd.m[2][2] &&
                                         fwd.m[0][1] == 0.0f && fwd.m[0][2] == 0.0f && fwd.m[1][0] == 0.0f && fwd.m[1][2] == 0.0f &&
                                         fwd.m[2][0] == 0.0f && fwd.m[2][1] == 0.0f &&
                                         inv.m[0][0] == inv.m[1][1] && inv.m[0][0] == inv.m[2][2];
            // A sphere whose centre lies many diameters from the origin (|q| = |centre| / s large) has its discriminant
            // cancel to a few bits: what the test returns there is mostly rounding, and the short form above rounds
            // differently from the reference's d_l = T^-1 d, a = d_l.d_l (measured on 10 000 spheres of diameter 0.01-0.03,
            // |q| up to 500: mean radiance +0.27 % against the parity-order kernel with the short form, +0.05 % with the
            // general one).  Such spheres keep the general record, which follows the reference's operations.
            const float q_max = fmaxf(fabsf(inv.m[0][3]), fmaxf(fabsf(inv.m[1][3]), fabsf(inv.m[2][3])));
            if (scaled_identity && q_max <= 32.0f && (general & PT_GENERAL_SPHERES) == 0) {
                *ftype = FP_SPHERE_WS;
                const float k = inv.m[0][0];
                rec[0] = k; rec[1] = inv.m[0][3]; rec[2] = inv.m[1][3]; rec[3] = inv.m[2][3];
                rec[4] = k * k; rec[5] = 1.0f / (k * k);
                return 6;
            }
            *ftype = FP_SPHERE;
            put_inv();
            return 12;
        }
        case PT_CUBE: {
            for (int j = 0; j < 3; j++) for (int i = 0; i < 3; i++) fs->a[j * 3 + i] = ncol[j][i];
            // Diagonal L (off-diagonal entries below 2e-7 of their column's diagonal entry): the cube is a box along
            // the world axes and is tested there, local axis i being world axis i.
            bool diagonal = (general & PT_GENERAL_CUBES) == 0;
            for (int j = 0; j < 3 && diagonal; j++) {
                if (fwd.m[j][j] == 0.0f) diagonal = false;
                for (int i = 0; i < 3; i++) if (i != j && fabsf(fwd.m[i][j]) > 2e-7f * fabsf(fwd.m[j][j])) diagonal = false;
            }
            if (diagonal) {
                *ftype = FP_CUBE_AA;
                for (int i = 0; i < 3; i++) {
                    const float h = 0.5f * fabsf(fwd.m[i][i]);
                    rec[i] = fwd.m[i][3] - h; rec[3 + i] = fwd.m[i][3] + h;
                }
                return 6;
            }
            *ftype = FP_CUBE;
            put_inv();
            return 12;
        }
        default: {   // PT_CYLINDER
            *ftype = FP_CYLINDER;
            put_inv();
            for (int i = 0; i < 3; i++) { fs->a[i] = col[0][i]; fs->a[3 + i] = col[1][i]; fs->a[6 + i] = ncol[2][i]; }
            for (int j = 0; j < 4; j++) { fs->a[9 + j] = inv.m[0][j]; fs->a[13 + j] = inv.m[1][j]; }
            return 12;
        }
    }
}

// One primitive after baking: its fast-kernel record and kind, and its world-space bounding box.
struct Baked {
    float rec[16];
    int len;
    int32_t kind;
    float lo[3], hi[3];
};

// World-space box of a finite primitive: the unit cube (which also bounds the unit cylinder) maps to
// centre +- 0.5 * sum_j |L_ij|, the sphere of radius 0.5 to centre +- 0.5 * |row i of L|.
void world_box(const pt_prim& p, const Aff& fwd, float* lo, float* hi) {
    for (int i = 0; i < 3; i++) {
        float h;
        if (p.type == PT_SPHERE) h = 0.5f * len3(fwd.m[i][0], fwd.m[i][1], fwd.m[i][2]);
        else h = 0.5f * (fabsf(fwd.m[i][0]) + fabsf(fwd.m[i][1]) + fabsf(fwd.m[i][2]));
        lo[i] = fwd.m[i][3] - h; hi[i] = fwd.m[i][3] + h;
    }
}

// World-space box of a plane's rectangle (shapes.h:105-112): the corners (xMin|xMax, yMin|yMax, 0) under T.  A
// rectangle without an edge on some side (bounds beyond +-1e30) gets an infinite box.
void plane_box(const pt_prim& p, const Aff& fwd, float* lo, float* hi) {
    bool finite = true;
    for (int j = 0; j < 4; j++) finite = finite && fabsf(p.bounds[j]) <= 1e30f;
    for (int i = 0; i < 3; i++) { lo[i] = finite ? INFINITY : -INFINITY; hi[i] = finite ? -INFINITY : INFINITY; }
    if (!finite) return;
    for (int cx = 0; cx < 2; cx++)
        for (int cy = 0; cy < 2; cy++) {
            const float x = p.bounds[cx ? 2 : 0], y = p.bounds[cy ? 3 : 1];
            for (int i = 0; i < 3; i++) {
                const float w = fwd.m[i][0] * x + fwd.m[i][1] * y + fwd.m[i][3];
                lo[i] = fminf(lo[i], w); hi[i] = fmaxf(hi[i], w);
            }
        }
}

// Packs the records of primitives `ids` (ascending scene indices) into one float array cut into runs of
// consecutive same-kind primitives.  Runs start on a 64-byte line; a record never straddles a chunk boundary and a
// run never spans two chunks.  Returns the number of chunks.
int pack_runs(const std::vector<Baked>& baked, const std::vector<int>& ids, std::vector<float>& fdata,
              std::vector<FRun>& runs, std::vector<int32_t>& chunk_runs) {
    int prev = -2;
    for (int k : ids) {
        const Baked& b = baked[k];
        bool new_run = runs.empty() || runs.back().type != b.kind || k != prev + 1;
        size_t start = fdata.size();
        if (new_run) start = (start + 15) / 16 * 16;
        if (start / FDATA_CHUNK != (start + b.len - 1) / FDATA_CHUNK) start = (start / FDATA_CHUNK + 1) * FDATA_CHUNK;
        if (start % FDATA_CHUNK == 0) new_run = true;
        fdata.resize(start, 0.0f);
        if (new_run) {
            FRun r; r.type = b.kind; r.first = k; r.count = 0; r.offset = (int32_t)start;
            runs.push_back(r);
        }
        runs.back().count++;
        fdata.insert(fdata.end(), b.rec, b.rec + b.len);
        prev = k;
    }
    for (int pad = 0; pad < 16; pad++) fdata.push_back(0.0f);   // scalar loads may read past the last record
    const int n_chunks = (int)((runs.back().offset + FDATA_CHUNK) / FDATA_CHUNK);
    fdata.resize((size_t)n_chunks * FDATA_CHUNK + 16, 0.0f);     // whole chunks are staged
    chunk_runs.assign(n_chunks + 1, 0);
    size_t r = 0;
    for (int c = 0; c < n_chunks; c++) {
        chunk_runs[c] = (int32_t)r;
        while (r < runs.size() && runs[r].offset / FDATA_CHUNK == c) r++;
    }
    chunk_runs[n_chunks] = (int32_t)runs.size();
    return n_chunks;
}

// Packs the records of primitives `ids` (ascending scene indices) sorted by kind, each with its scene index << 3
// behind it (as int bits), at the kind's stride.  Offsets are 16-float aligned; 32 floats of padding follow the last
// record because the loops fetch records two at a time.
void pack_sorted(const std::vector<Baked>& baked, const std::vector<int>& ids, std::vector<float>& data, FSorted& hdr,
                 std::vector<float>* boxes = nullptr) {
    std::memset(&hdr, 0, sizeof(hdr));
    data.assign(32, 0.0f);      // head of the buffer: a copy of the offsets and counts (filled in below)
    for (int kind = 0; kind < FP_KINDS; kind++) {
        data.resize((data.size() + 15) / 16 * 16, 0.0f);
        hdr.off[kind] = (int32_t)data.size();
        const int stride = FSORTED_STRIDE[kind];
        for (int k : ids) {
            const Baked& b = baked[k];
            if (b.kind != kind) continue;
            const size_t at = data.size();
            data.resize(at + (size_t)stride, 0.0f);
            std::memcpy(&data[at], b.rec, (size_t)b.len * sizeof(float));
            const int32_t id = (int32_t)(k << 3);
            std::memcpy(&data[at + (size_t)b.len], &id, sizeof(id));
            hdr.count[kind]++;
            if (boxes) {
                // padded like the hierarchy's boxes: the culling test's rounding must never drop a record its own test accepts
                for (int i = 0; i < 3; i++) boxes->push_back(b.lo[i] - 1e-4f * (1.0f + fabsf(b.lo[i])));
                for (int i = 0; i < 3; i++) boxes->push_back(b.hi[i] + 1e-4f * (1.0f + fabsf(b.hi[i])));
            }
        }
    }
    data.resize(data.size() + 32, 0.0f);
    static_assert(2 * FP_KINDS <= 32, "the header has 32 words");
    std::memcpy(&data[0], hdr.off, sizeof(hdr.off));
    std::memcpy(&data[FP_KINDS], hdr.count, sizeof(hdr.count));
}

// Hierarchy over ids[lo, hi), appended to `nodes` in depth-first order.  Boxes are padded so that the rounding of
// the kernels' slab test can never cull a primitive its own test would accept.
// `levels` is the depth the leaves below this node may reach (hi - lo <= leaf_prims << levels).  With `sah` the
// split is the one of least surface-area cost (area(left) * count(left) + area(right) * count(right)) over the three
// axes and every position of the centroid order that keeps both sides within levels - 1; without it, the median
// of the widest centroid axis.
struct BvhBuild {
    const std::vector<Baked>& baked;
    std::vector<BNode>& nodes;
    std::vector<BPrim>& bprims;
    int leaf_prims;
    bool sah;
    std::vector<float> suffix;   // scratch: area of the box of the sorted range's tail
};

float box_area(const float* lo, const float* hi) {
    const float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
    return x * y + y * z + z * x;
}

void build_bvh(BvhBuild& B, std::vector<int>& ids, int lo, int hi, int levels) {
    const std::vector<Baked>& baked = B.baked;
    std::vector<BNode>& nodes = B.nodes;
    const size_t me = nodes.size();
    nodes.push_back(BNode());
    float blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};
    float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = lo; i < hi; i++) {
        const Baked& b = baked[ids[i]];
        for (int a = 0; a < 3; a++) {
            blo[a] = fminf(blo[a], b.lo[a]); bhi[a] = fmaxf(bhi[a], b.hi[a]);
            const float c = 0.5f * (b.lo[a] + b.hi[a]);
            clo[a] = fminf(clo[a], c); chi[a] = fmaxf(chi[a], c);
        }
    }
    for (int a = 0; a < 3; a++) {
        const float pad = 1e-4f * (1.0f + fmaxf(fabsf(blo[a]), fabsf(bhi[a])));
        nodes[me].lo[a] = blo[a] - pad; nodes[me].hi[a] = bhi[a] + pad;
    }
    const int n = hi - lo;
    if (n <= B.leaf_prims) {
        nodes[me].leaf = (int32_t)((B.bprims.size() << 4) | (size_t)n);
        std::sort(ids.begin() + lo, ids.begin() + hi);
        for (int i = lo; i < hi; i++) {
            const Baked& b = baked[ids[i]];
            BPrim q;
            std::memset(&q, 0, sizeof(q));
            std::memcpy(q.rec, b.rec, 12 * sizeof(float));
            q.kind = b.kind; q.id = ids[i] << 3;
            B.bprims.push_back(q);
        }
    } else {
        auto by_centroid = [&](int axis) {
            return [&baked, axis](int x, int y) {
                const float cx = baked[x].lo[axis] + baked[x].hi[axis], cy = baked[y].lo[axis] + baked[y].hi[axis];
                return cx < cy || (cx == cy && x < y);
            };
        };
        int axis = 0;
        for (int a = 1; a < 3; a++) if (chi[a] - clo[a] > chi[axis] - clo[axis]) axis = a;
        int mid = (lo + hi) / 2;
        // each side may hold at most `cap` primitives if its leaves are to stay within levels - 1
        const long long cap = levels >= 1 ? (long long)B.leaf_prims << std::min(levels - 1, 40) : 1;
        const int k_min = (int)std::max<long long>(1, n - cap), k_max = (int)std::min<long long>(n - 1, cap);
        if (B.sah && k_min < k_max) {
            float best_cost = INFINITY;
            int best_axis = axis, best_k = mid - lo;
            B.suffix.resize((size_t)n + 1);
            for (int a = 0; a < 3; a++) {
                std::sort(ids.begin() + lo, ids.begin() + hi, by_centroid(a));
                float l[3] = {INFINITY, INFINITY, INFINITY}, h[3] = {-INFINITY, -INFINITY, -INFINITY};
                for (int i = n - 1; i >= 1; i--) {
                    const Baked& b = baked[ids[lo + i]];
                    for (int c = 0; c < 3; c++) { l[c] = fminf(l[c], b.lo[c]); h[c] = fmaxf(h[c], b.hi[c]); }
                    B.suffix[(size_t)i] = box_area(l, h);
                }
                for (int c = 0; c < 3; c++) { l[c] = INFINITY; h[c] = -INFINITY; }
                for (int k = 1; k <= k_max; k++) {        // k primitives on the left
                    const Baked& b = baked[ids[lo + k - 1]];
                    for (int c = 0; c < 3; c++) { l[c] = fminf(l[c], b.lo[c]); h[c] = fmaxf(h[c], b.hi[c]); }
                    if (k < k_min) continue;
                    const float cost = box_area(l, h) * (float)k + B.suffix[(size_t)k] * (float)(n - k);
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_k = k; }
                }
            }
            axis = best_axis; mid = lo + best_k;
            if (axis != 2) std::sort(ids.begin() + lo, ids.begin() + hi, by_centroid(axis));
        } else {
            std::nth_element(ids.begin() + lo, ids.begin() + mid, ids.begin() + hi, by_centroid(axis));
        }
        nodes[me].leaf = -1;
        build_bvh(B, ids, lo, mid, levels - 1);
        build_bvh(B, ids, mid, hi, levels - 1);
    }
    nodes[me].skip = (int32_t)nodes.size();
}

// Levels a tree over n primitives needs when every split is as even as it can be.
int bvh_min_levels(int n, int leaf_prims) {
    int levels = 0;
    while (((long long)leaf_prims << levels) < n) levels++;
    return levels;
}

// The fat form of a skip-link tree whose leaves hold one primitive each: one entry per inner node, in the same
// depth-first order, each with the boxes and references of its two children (left child = next node, right child =
// the left child's skip link), the boxes on a 15-bit grid over the root's box (see BFat; `step` is half a grid cell).
// Returns the depth of the deepest leaf (the root's children are at depth 1): the walk parks at most one reference
// per level, so this is the stack height a lane can need.
int build_fat(const std::vector<BNode>& nodes, std::vector<BFat>& fat, float origin[3], float step[3]) {
    float cell[3];
    for (int k = 0; k < 3; k++) {
        origin[k] = nodes[0].lo[k];
        cell[k] = (nodes[0].hi[k] - nodes[0].lo[k]) / 32767.0f * (1.0f + 1e-6f);
        if (!(cell[k] > 0.0f)) cell[k] = 1e-30f;       // a flat box: every coordinate is the origin
        step[k] = 0.5f * cell[k];
    }
    // the largest q with origin + cell * q <= x, and the smallest with origin + cell * q >= x, as the float
    // arithmetic has it
    auto below = [&](float x, int k) {
        int q = (int)std::floor((x - origin[k]) / cell[k]);
        q = std::min(std::max(q, 0), 32767);
        while (q > 0 && std::fmaf(cell[k], (float)q, origin[k]) > x) q--;
        return q;
    };
    auto above = [&](float x, int k) {
        int q = (int)std::ceil((x - origin[k]) / cell[k]);
        q = std::min(std::max(q, 0), 32767);
        while (q < 32767 && std::fmaf(cell[k], (float)q, origin[k]) < x) q++;
        return q;
    };
    std::vector<int32_t> fat_index(nodes.size(), -1);
    int32_t n_inner = 0;
    for (size_t i = 0; i < nodes.size(); i++) if (nodes[i].leaf < 0) fat_index[i] = n_inner++;
    fat.assign((size_t)n_inner, BFat());
    auto ref_of = [&](size_t i) { return nodes[i].leaf < 0 ? fat_index[i] : ~(int32_t)(nodes[i].leaf >> 4); };
    for (size_t i = 0; i < nodes.size(); i++) {
        if (nodes[i].leaf >= 0) continue;
        const size_t a = i + 1, b = (size_t)nodes[i + 1].skip;
        BFat& f = fat[(size_t)fat_index[i]];
        std::memset(&f, 0, sizeof(f));
        for (int k = 0; k < 3; k++) {
            const int la = below(nodes[a].lo[k], k), ha = std::max(la, above(nodes[a].hi[k], k));
            const int lb = below(nodes[b].lo[k], k), hb = std::max(lb, above(nodes[b].hi[k], k));
            f.sum_a[k] = (uint16_t)(la + ha); f.diff_a[k] = (uint16_t)(ha - la);
            f.sum_b[k] = (uint16_t)(lb + hb); f.diff_b[k] = (uint16_t)(hb - lb);
        }
        f.ref_a = ref_of(a); f.ref_b = ref_of(b);
    }
    // depth-first order: a node's depth is one more than that of the nearest earlier node whose subtree holds it
    int depth = 0;
    std::vector<std::pair<int32_t, int>> open;   // (skip link, depth) of the enclosing inner nodes
    for (size_t i = 0; i < nodes.size(); i++) {
        while (!open.empty() && (size_t)open.back().first <= i) open.pop_back();
        const int d = open.empty() ? 0 : open.back().second + 1;
        if (nodes[i].leaf < 0) open.push_back({nodes[i].skip, d});
        else depth = std::max(depth, d);
    }
    return depth;
}

}  // namespace

int pt_set_error(int code, const char* expr, const char* file, int line) {
    std::snprintf(g_err, sizeof(g_err), "HIP error = %u at %s:%d '%s' ", (unsigned)code, file, line, expr);
    return code ? code : -1;
}
#define PT_FAIL(msg) pt_set_error(-1, msg, __FILE__, __LINE__)

extern "C" {

const char* pt_last_error(void) { return g_err; }
const char* pt_version(void) { return "pathtracer_amd 0.1 (gfx950)"; }

float pt_focal_length(int nx, int ny, float v_height, float fov_deg) {   // main.cu:229-232
    if (ny == 0) return 0.0f;
    const float v_width = (float)(nx / ny) * v_height;
    return (float)((double)(v_width / 2) / tan(((double)(fov_deg / 2) * PT_PI) / 180));
}

int pt_default_scene(pt_prim* prims, pt_material* mats, int* n_mats) {   // main.cu:73-195
    default_materials(mats);
    *n_mats = 8;
    int n = default_walls(prims);
    Aff t7 = aff_identity(); aff_translate(t7, -0.6f, -0.5f, -3.0f); aff_scale(t7, 0.4f, 0.4f, 0.4f);
    Aff t5 = aff_identity(); aff_translate(t5, 0.0f, -0.5f, -3.0f); aff_scale(t5, 0.4f, 0.4f, 0.4f);
    Aff t6 = aff_identity(); aff_translate(t6, 0.6f, -0.5f, -3.0f); aff_scale(t6, 0.4f, 0.4f, 0.4f);
    Aff t8 = aff_identity(); aff_translate(t8, 0.0f, -0.5f, -4.2f); aff_scale(t8, 2.5f, 0.3f, 0.5f);
    put_prim(&prims[n++], PT_SPHERE, 4, t7, 0.0f);
    put_prim(&prims[n++], PT_SPHERE, 5, t5, 0.0f);
    put_prim(&prims[n++], PT_SPHERE, 6, t6, 0.0f);
    put_prim(&prims[n++], PT_CUBE, 7, t8, 0.0f);
    return n;
}

int pt_glass_grid_scene(int n, pt_prim* prims, pt_material* mats, int* n_mats) {
    if (n < 1) return 0;
    default_materials(mats);
    *n_mats = 8;
    int k = default_walls(prims);
    Aff t8 = aff_identity(); aff_translate(t8, 0.0f, -0.5f, -4.2f); aff_scale(t8, 2.5f, 0.3f, 0.5f);
    put_prim(&prims[k++], PT_CUBE, 7, t8, 0.0f);
    // n x n glass spheres in the vertical plane z = -3, x in [-0.8,0.8], y in [-0.8,0.6];
    // materials cycle glass1/glass2/glass3.
    const float x0 = -0.8f, x1 = 0.8f, y0 = -0.8f, y1 = 0.6f;
    const float pitch = n > 1 ? (x1 - x0) / (float)(n - 1) : 0.0f;
    const float pitch_y = n > 1 ? (y1 - y0) / (float)(n - 1) : 0.0f;
    const float diam = n > 1 ? 0.8f * fminf(pitch, pitch_y) : 0.4f;
    for (int r = 0; r < n; r++)
        for (int c = 0; c < n; c++) {
            Aff t = aff_identity();
            aff_translate(t, n > 1 ? x0 + pitch * (float)c : 0.0f, n > 1 ? y0 + pitch_y * (float)r : -0.5f, -3.0f);
            aff_scale(t, diam, diam, diam);
            put_prim(&prims[k++], PT_SPHERE, 4 + (r * n + c) % 3, t, 0.0f);
        }
    return k;
}

int pt_synthetic_scene(int n_spheres, int n_cubes, uint32_t seed, pt_prim* prims, pt_material* mats, int* n_mats) {
    if (n_spheres < 0 || n_cubes < 0) return 0;
    int k = pt_default_scene(prims, mats, n_mats);
    // centres uniform in [-0.9,0.9] x [-0.9,0.9] x [-4.9,-1.5], uniform scale in
    // [0.01,0.03]; materials cycle base, red, green, diffuse_pure, glass1.
    const int cycle[5] = {0, 1, 2, 7, 4};
    const uint32_t key = host_hash(seed ^ 0x5bd1e995u);
    const int total = n_spheres + n_cubes;
    for (int i = 0; i < total; i++) {
        const uint32_t b = (uint32_t)i * 4u;
        const float ux = host_uniform(key, b), uy = host_uniform(key, b + 1), uz = host_uniform(key, b + 2), us = host_uniform(key, b + 3);
        const float s = 0.01f + 0.02f * us;
        Aff t = aff_identity();
        aff_translate(t, -0.9f + 1.8f * ux, -0.9f + 1.8f * uy, -4.9f + 3.4f * uz);
        aff_scale(t, s, s, s);
        put_prim(&prims[k++], i < n_spheres ? PT_SPHERE : PT_CUBE, cycle[i % 5], t, 0.0f);
    }
    return k;
}

int pt_scene_upload(const pt_prim* prims, int n_prims, const pt_material* mats, int n_mats, int device, pt_scene** out) {
    return pt_scene_upload_opt(prims, n_prims, mats, n_mats, device, nullptr, out);
}

static_assert(sizeof(pt_upload_options) == 16 && sizeof(pt_config) == 72 && sizeof(pt_prim) == 72 && sizeof(pt_material) == 72,
              "the layouts the bindings of other languages assume");

int pt_scene_upload_opt(const pt_prim* prims, int n_prims, const pt_material* mats, int n_mats, int device,
                        const pt_upload_options* opt, pt_scene** out) {
    if (!prims || !mats || !out || n_prims <= 0 || n_mats <= 0) return PT_FAIL("pt_scene_upload: bad arguments");
    pt_upload_options o = {};
    if (opt) o = *opt;
    if (o.bvh_split < 0 || o.bvh_split > PT_BVH_SPLIT_AREA + 5) return PT_FAIL("pt_scene_upload_opt: bvh_split must be 0..PT_BVH_SPLIT_AREA + 5");
    if (o.bvh_min_prims != 0 && o.bvh_min_prims < 2) return PT_FAIL("pt_scene_upload_opt: bvh_min_prims must be 0 or >= 2");
    if (o.bvh_leaf_prims < 0 || o.bvh_leaf_prims > 15) return PT_FAIL("pt_scene_upload_opt: bvh_leaf_prims must be 0..15");
    for (int k = 0; k < n_prims; k++) {
        if (prims[k].type < PT_PLANE || prims[k].type > PT_SPHERE) return PT_FAIL("pt_scene_upload: unknown shape type");
        if (prims[k].material < 0 || prims[k].material >= n_mats) return PT_FAIL("pt_scene_upload: material index out of range");
        // A transform with a NaN or infinite entry, or one that cannot be inverted (shapes.h:37 inverts it), has no
        // usable T^-1: every record built from it would be NaN or infinite.  Bounds may be infinite (an unbounded
        // plane, shapes.h:111-112) but not NaN.
        for (int j = 0; j < 12; j++) if (!std::isfinite(prims[k].transform[j])) return PT_FAIL("pt_scene_upload: a transform entry is not finite");
        for (int j = 0; j < 4; j++) if (std::isnan(prims[k].bounds[j])) return PT_FAIL("pt_scene_upload: a plane bound is NaN");
        const Aff inv = aff_inverse(aff_load(prims[k].transform));
        for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++)
            if (!std::isfinite(inv.m[i][j])) return PT_FAIL("pt_scene_upload: a transform is singular (no finite inverse)");
    }
    PT_CHECK(hipSetDevice(device));
    std::vector<DPrim> dp(n_prims);
    std::vector<FShade> fs(n_prims);
    std::vector<Baked> baked(n_prims);
    std::vector<int> all(n_prims), top, finite;
    for (int k = 0; k < n_prims; k++) {
        const Aff fwd = aff_load(prims[k].transform);
        const Aff inv = aff_inverse(fwd);
        aff_store(inv, dp[k].inv);
        aff_store(fwd, dp[k].fwd);
        for (int j = 0; j < 4; j++) dp[k].bounds[j] = prims[k].bounds[j];
        dp[k].type = prims[k].type; dp[k].material = prims[k].material; dp[k].index = k; dp[k].pad = 0;
        std::memset(&baked[k], 0, sizeof(Baked));
        baked[k].len = bake_fast(prims[k], fwd, inv, o.general_kinds, baked[k].rec, &fs[k], &baked[k].kind);
        if (prims[k].type == PT_PLANE) plane_box(prims[k], fwd, baked[k].lo, baked[k].hi);
        else world_box(prims[k], fwd, baked[k].lo, baked[k].hi);
        all[k] = k;
        (prims[k].type == PT_PLANE ? top : finite).push_back(k);
    }
    // A sphere or cylinder is tested in its own frame (shapes.h:42-48), where a ray origin at distance D from a shape of
    // scale s has coordinates of magnitude D/s: the c = |o_l|^2 - 0.25 of the quadratic (shapes.h:258-261,328-331) is
    // then known to a few units in the last place of (D/s)^2 only, which moves the shape's apparent surface by up to
    // about s * 2^-21 * (D/s)^2 — for a shape of scale 0.01 seen from 4 units away, a tenth of its radius.  The loop
    // over all primitives hits that apparent surface wherever it lies; whatever culls by boxes (the hierarchy, the
    // per-tile lists of the camera batches) must not drop such a hit, so the boxes of these shapes grow by that much,
    // for origins anywhere in the box of the finite shapes or at the world origin (the reference's camera).  Rays from
    // farther away than that can still see the loop and the hierarchy disagree on a grazing hit of a tiny shape.
    {
        float flo[3] = {0.0f, 0.0f, 0.0f}, fhi[3] = {0.0f, 0.0f, 0.0f};
        for (int k : finite) for (int i = 0; i < 3; i++) { flo[i] = fminf(flo[i], baked[k].lo[i]); fhi[i] = fmaxf(fhi[i], baked[k].hi[i]); }
        for (int k : finite) {
            if (prims[k].type != PT_SPHERE && prims[k].type != PT_CYLINDER) continue;
            const Aff fwd = aff_load(prims[k].transform);
            float s_min = INFINITY, s_max = 0.0f, far2 = 0.0f;
            for (int j = 0; j < 3; j++) {
                const float c = len3(fwd.m[0][j], fwd.m[1][j], fwd.m[2][j]);
                s_min = fminf(s_min, c); s_max = fmaxf(s_max, c);
                const float centre = fwd.m[j][3];
                const float away = fmaxf(fabsf(centre - flo[j]), fabsf(centre - fhi[j]));
                far2 += away * away;
            }
            if (!(s_min > 0.0f) || !std::isfinite(far2)) continue;
            const float m = sqrtf(far2) / s_min + 1.0f;
            const float pad = fminf(s_max * 4.76837158e-7f * m * m, 0.5f * s_max);     // 2^-21; at most the shape's own radius
            for (int i = 0; i < 3; i++) { baked[k].lo[i] -= pad; baked[k].hi[i] += pad; }
        }
    }
    // the parity-order kernels' table sorted by shape type, with the boxes the camera batches cull by (padded like
    // the hierarchy's boxes: the culling test's rounding must never drop a record its own test accepts)
    std::vector<DPrim> dp_by_type;
    std::vector<float> type_boxes;
    int type_first[5] = {0, 0, 0, 0, 0};
    for (int t = PT_PLANE; t <= PT_SPHERE; t++) {
        type_first[t] = (int)dp_by_type.size();
        for (int k = 0; k < n_prims; k++) {
            if (dp[k].type != t) continue;
            dp_by_type.push_back(dp[k]);
            for (int i = 0; i < 3; i++) type_boxes.push_back(baked[k].lo[i] - 1e-4f * (1.0f + fabsf(baked[k].lo[i])));
            for (int i = 0; i < 3; i++) type_boxes.push_back(baked[k].hi[i] + 1e-4f * (1.0f + fabsf(baked[k].hi[i])));
        }
    }
    type_first[4] = (int)dp_by_type.size();
    std::vector<float> fdata, fdata_top;
    std::vector<FRun> runs, runs_top;
    std::vector<int32_t> chunk_runs, chunk_runs_top;
    const int n_chunks = pack_runs(baked, all, fdata, runs, chunk_runs);
    std::vector<float> sorted_all, sorted_top;
    FSorted hdr_all, hdr_top;
    std::vector<float> slot_boxes;
    pack_sorted(baked, all, sorted_all, hdr_all, &slot_boxes);
    pack_sorted(baked, top, sorted_top, hdr_top);
    std::vector<BNode> bnodes;
    std::vector<BPrim> bprims;
    std::vector<BFat> bfat;
    int fat_depth = 0;
    float fat_origin[3] = {0.0f, 0.0f, 0.0f}, fat_step[3] = {1.0f, 1.0f, 1.0f};
    const int bvh_min = o.bvh_min_prims ? o.bvh_min_prims : BVH_MIN_PRIMS;
    if ((int)finite.size() >= bvh_min) {
        const int leaf_prims = o.bvh_leaf_prims ? o.bvh_leaf_prims : BVH_LEAF_PRIMS;
        const int split = o.bvh_split ? o.bvh_split : BVH_SPLIT;
        BvhBuild build = {baked, bnodes, bprims, leaf_prims, split >= PT_BVH_SPLIT_AREA, {}};
        const int slack = split >= PT_BVH_SPLIT_AREA ? split - PT_BVH_SPLIT_AREA : 0;
        build_bvh(build, finite, 0, (int)finite.size(), bvh_min_levels((int)finite.size(), leaf_prims) + slack);
        if (leaf_prims == 1 && finite.size() >= 2 && (int)finite.size() <= BFAT_MAX_PRIMS) {
            fat_depth = build_fat(bnodes, bfat, fat_origin, fat_step);
            if (fat_depth > WALK_STACK_MAX) bfat.clear();   // cannot happen: the builder keeps <= 2^19 leaves within 19 + 5 levels
        }
        if (top.empty()) {   // the loop over the other primitives still wants one (empty) chunk to read
            fdata_top.assign(FDATA_CHUNK + 16, 0.0f);
        } else {
            pack_runs(baked, top, fdata_top, runs_top, chunk_runs_top);
        }
    }
    std::vector<DMat> dm(n_mats);
    float max_emissive = 0.0f;
    for (int i = 0; i < n_mats; i++) {
        std::memset(&dm[i], 0, sizeof(DMat));
        std::memcpy(&dm[i], &mats[i], sizeof(pt_material));
        const float* em = mats[i].emissive;
        // emissive.norm() > 0 (bsdf.h:90) as float arithmetic has it: false for a NaN component, false when the squares underflow
        dm[i].emits = (em[0] * em[0] + (em[1] * em[1] + em[2] * em[2])) > 0.0f ? 1 : 0;
        for (int c = 0; c < 3; c++) if (std::isfinite(em[c])) max_emissive = std::max(max_emissive, fabsf(em[c]));
    }
    pt_scene* sc = new (std::nothrow) pt_scene();
    if (!sc) return PT_FAIL("pt_scene_upload: out of host memory");
    std::memset(sc, 0, sizeof(*sc));
    sc->device = device; sc->n_prims = n_prims; sc->n_mats = n_mats; sc->max_emissive = max_emissive;
    hipError_t e;
#define UP(dst, vec, T)                                                                        \
    e = hipMalloc((void**)&sc->dst, vec.size() * sizeof(T));                                   \
    if (e == hipSuccess) e = hipMemcpy(sc->dst, vec.data(), vec.size() * sizeof(T), hipMemcpyHostToDevice); \
    if (e != hipSuccess) { pt_scene_free(sc); return pt_set_error((int)e, "scene upload " #dst, __FILE__, __LINE__); }
    UP(d_prims, dp, DPrim)
    UP(d_prims_by_type, dp_by_type, DPrim)
    UP(d_type_box, type_boxes, float)
    for (int t = 0; t < 5; t++) sc->type_first[t] = type_first[t];
    sc->typed_loops_ok = 1;
    for (const DPrim& q : dp)
        for (int j = 0; j < 12; j++) if (!(fabsf(q.inv[j]) < 1e5f) || !(fabsf(q.fwd[j]) < 1e5f)) sc->typed_loops_ok = 0;
    UP(d_mats, dm, DMat)
    UP(d_fdata, fdata, float)
    UP(d_runs, runs, FRun)
    UP(d_chunk_runs, chunk_runs, int32_t)
    sc->n_chunks = n_chunks;
    UP(d_fshade, fs, FShade)
    UP(d_sorted, sorted_all, float)
    sc->sorted = hdr_all; sc->sorted.data = sc->d_sorted;
    UP(d_slot_box, slot_boxes, float)
    sc->n_slots = n_prims;
    sc->n_runs = (int)runs.size();
    if (!bnodes.empty()) {
        UP(d_bnodes, bnodes, BNode)
        UP(d_bprims, bprims, BPrim)
        if (!bfat.empty()) {
            UP(d_bfat, bfat, BFat)
            sc->fat_depth = fat_depth;
            for (int k = 0; k < 3; k++) { sc->fat_origin[k] = fat_origin[k]; sc->fat_step[k] = fat_step[k]; }
        }
        if (!top.empty()) { UP(d_top_ids, top, int32_t) }
        sc->n_top_ids = (int)top.size();
        UP(d_sorted_top, sorted_top, float)
        sc->sorted_top = hdr_top; sc->sorted_top.data = sc->d_sorted_top;
        UP(d_fdata_top, fdata_top, float)
        if (!runs_top.empty()) { UP(d_runs_top, runs_top, FRun) }
        sc->n_bnodes = (int)bnodes.size();
        sc->n_runs_top = (int)runs_top.size();
    }
#undef UP
    *out = sc;
    return 0;
}

int pt_scene_free(pt_scene* sc) {
    if (!sc) return 0;
    (void)hipSetDevice(sc->device);
    if (sc->d_prims) (void)hipFree(sc->d_prims);
    if (sc->d_prims_by_type) (void)hipFree(sc->d_prims_by_type);
    if (sc->d_type_box) (void)hipFree(sc->d_type_box);
    if (sc->d_mats) (void)hipFree(sc->d_mats);
    if (sc->d_fdata) (void)hipFree(sc->d_fdata);
    if (sc->d_runs) (void)hipFree(sc->d_runs);
    if (sc->d_chunk_runs) (void)hipFree(sc->d_chunk_runs);
    if (sc->d_fshade) (void)hipFree(sc->d_fshade);
    if (sc->d_bnodes) (void)hipFree(sc->d_bnodes);
    if (sc->d_bprims) (void)hipFree(sc->d_bprims);
    if (sc->d_bfat) (void)hipFree(sc->d_bfat);
    if (sc->d_top_ids) (void)hipFree(sc->d_top_ids);
    if (sc->d_sorted) (void)hipFree(sc->d_sorted);
    if (sc->d_slot_box) (void)hipFree(sc->d_slot_box);
    if (sc->d_sorted_top) (void)hipFree(sc->d_sorted_top);
    if (sc->d_fdata_top) (void)hipFree(sc->d_fdata_top);
    if (sc->d_runs_top) (void)hipFree(sc->d_runs_top);
    delete sc;
    return 0;
}

}  // extern "C"

PScene pt_parity_scene(const pt_scene* sc) {
    PScene p;
    p.prims = sc->d_prims; p.n_prims = sc->n_prims;
    p.by_type = sc->d_prims_by_type; p.boxes = sc->n_prims <= 64 ? sc->d_type_box : nullptr;
    for (int t = 0; t < 5; t++) p.type_first[t] = sc->type_first[t];
    p.nodes = sc->d_bnodes; p.n_nodes = sc->n_bnodes; p.bprims = sc->d_bprims;
    p.top = sc->d_top_ids; p.n_top = sc->n_top_ids;
    if (sc->tune[PT_TUNE_BVH_WALK] == 3) { p.nodes = nullptr; p.n_nodes = 0; }    // the loop over all primitives is asked for
    return p;
}

extern "C" {

int pt_scene_tune(pt_scene* sc, int knob, int value) {
    if (!sc || knob < 0 || knob >= PT_TUNE_COUNT) return PT_FAIL("pt_scene_tune: bad scene or knob");
    bool ok = value == 0;
    switch (knob) {
        case PT_TUNE_TILE_LOG2: ok = ok || (value >= 2 && value <= 6); break;
        case PT_TUNE_WALK_BUDGET: ok = ok || value >= 1; break;
        case PT_TUNE_SHADE_GATE: ok = ok || (value >= 1 && value <= 64); break;
        case PT_TUNE_BVH_WALK: ok = ok || value == 1 || value == 2 || value == 3; break;
        case PT_TUNE_GLASS_QUEUE: ok = ok || value == -1 || (value >= 8 && value <= 64); break;
        case PT_TUNE_GLASS_GATE: ok = ok || (value >= 1 && value <= 64); break;
        case PT_TUNE_LEAF_GATE: ok = ok || (value >= 1 && value <= 64); break;
    }
    if (!ok) return PT_FAIL("pt_scene_tune: value out of range for this knob");
    sc->tune[knob] = value;
    return 0;
}

int pt_scene_device(const pt_scene* sc) { return sc ? sc->device : -1; }
int pt_scene_bvh_nodes(const pt_scene* sc) { return sc ? sc->n_bnodes : -1; }

}  // extern "C"

// pt_render (sums == null: all of cfg->n_samples, floats to d_fb) and pt_render_samples (samples [sample_begin,
// sample_begin + sample_count) of them, 32.32 sums added to `sums`).
static int render_launch(const pt_scene* sc, const pt_config* cfg, float* d_fb, int sample_begin, int sample_count,
                         unsigned long long* sums, unsigned long long* counters, void* stream) {
    if (!sc || !cfg || (!d_fb && !sums)) return PT_FAIL("pt_render: null argument");
    if (cfg->max_x <= 0 || cfg->max_y <= 0 || cfg->n_samples <= 0 || cfg->n_bounces < 0)
        return PT_FAIL("pt_render: bad image size, n_samples or n_bounces");
    if (cfg->n_bounces >= (1 << 24)) return PT_FAIL("pt_render: n_bounces above 2^24");   // (a set-aside hit packs the segment number into 24 bits)
    if (sample_begin < 0 || sample_count <= 0 || (long long)sample_begin + sample_count > cfg->n_samples)
        return PT_FAIL("pt_render_samples: the sample range must lie inside [0, n_samples)");
    if (cfg->n_samples > (1 << 24)) return PT_FAIL("pt_render: n_samples above 2^24");
    // Pixel sums are 32.32 fixed point and a sample counts up to 2^30 / n_samples (ptw::to_fixed), so a sum cannot
    // leave the range; a light whose own emission is beyond that cap would be dimmed silently, so it is refused.
    if ((double)sc->max_emissive * (double)cfg->n_samples > 1073741824.0)
        return PT_FAIL("pt_render: n_samples x the largest emissive component exceeds 2^30, the range of a pixel's sum");
    if ((uint64_t)cfg->max_x * (uint64_t)cfg->max_y > 0x7fffffffull) return PT_FAIL("pt_render: image above 2^31 pixels");
    if (cfg->row_begin < 0 || cfg->row_end > cfg->max_y || cfg->row_begin > cfg->row_end)
        return PT_FAIL("pt_render: row tile outside the image");
    if (cfg->fb_layout != PT_FB_AOS && cfg->fb_layout != PT_FB_SOA) return PT_FAIL("pt_render: unknown fb_layout");
    if (cfg->row_step < 0) return PT_FAIL("pt_render: negative row_step");
    if (cfg->fb_compact != 0 && cfg->fb_compact != 1) return PT_FAIL("pt_render: fb_compact must be 0 or 1");
    if (cfg->bsdf_model != PT_BSDF_REFERENCE && cfg->bsdf_model != PT_BSDF_CLASSES) return PT_FAIL("pt_render: unknown bsdf_model");
    if (cfg->tile_log2 != 0 && (cfg->tile_log2 < 2 || cfg->tile_log2 > 6)) return PT_FAIL("pt_render: tile_log2 must be 0 or 2..6");
    PT_CHECK(hipSetDevice(sc->device));
    RenderArgs a;
    std::memset(&a, 0, sizeof(a));
    // main.cu:40-52, one float operation at a time.
    const float vh = cfg->view_port_height;
    const float vw = vh * ((float)cfg->max_x / cfg->max_y);
    const float vu[3] = {vw, 0.0f, 0.0f}, vv[3] = {0.0f, -vh, 0.0f};
    float du[3], dv[3], ul[3];
    const float foc[3] = {0.0f, 0.0f, cfg->focal_length};
    for (int c = 0; c < 3; c++) {
        du[c] = vu[c] / (float)cfg->max_x;
        dv[c] = vv[c] / (float)cfg->max_y;
        ul[c] = ((cfg->camera_pos[c] - foc[c]) - vu[c] / 2.0f) - vv[c] / 2.0f;
        a.p00[c] = ul[c] + (du[c] + dv[c]) * 0.5f;
        a.cam[c] = cfg->camera_pos[c];
    }
    a.du = du[0]; a.dv = dv[1];
    a.max_x = cfg->max_x; a.max_y = cfg->max_y;
    a.n_samples = sample_count; a.sample_begin = sample_begin; a.sums = sums; a.counters = counters;
    a.n_bounces = cfg->n_bounces; a.bsdf_model = cfg->bsdf_model;
    a.sample_cap = 1073741824.0f / (float)cfg->n_samples;
    a.seed_hash = host_hash(cfg->seed + 0x9E3779B9u);
    const int row_step = cfg->row_step == 0 ? 1 : cfg->row_step;
    a.row_begin = cfg->row_begin; a.row_step = row_step;
    a.n_rows = (cfg->row_end - cfg->row_begin + row_step - 1) / row_step;
    a.n_lin = (uint32_t)a.n_rows * (uint32_t)cfg->max_x;
    a.fb_compact = cfg->fb_compact;
    // Pixels per wave.  Small tiles even out the end of the launch (a wave lives for about P*spp*2.4/64 loop
    // iterations and the chip holds 8192 waves), large tiles amortise each wave's drain phase; measured on the
    // 1080p frame and its interleaved row sets: 32 is best from 1.5 Mpixel, 16 from 0.2 Mpixel, 8 below, 64 only
    // for frames several times 1080p.  The image does not depend on this choice (pixel sums are exact
    // fixed-point sums).
    // The loop with camera batches keeps 32 up to the largest frames (3840x2160 @ 1024: 181 ms against 202 with 64) and
    // takes it from 0.9 Mpixel (half a 1080p frame: 1.479 ms against 1.508 with 16; a third: 1.040 against 1.024).
    // (the loop with camera batches and the queue of set-aside hits shades with bsdf.h's tracePath only and searches
    // no hierarchy; the class-based model and scenes with a hierarchy keep the plain loop)
    const bool fast_batched = cfg->math_mode == PT_MATH_FAST && sc->n_bnodes == 0;
    const bool parity_batched = cfg->math_mode == PT_MATH_PARITY && sc->typed_loops_ok && (sc->n_bnodes == 0 || sc->tune[PT_TUNE_BVH_WALK] == 3);
    const bool batched = (fast_batched || parity_batched) && cfg->bsdf_model == PT_BSDF_REFERENCE && sc->tune[PT_TUNE_GLASS_QUEUE] >= 0;
    a.tile_log2 = (a.n_lin >= 6000000u && !batched) ? 6 : (a.n_lin >= (batched ? 900000u : 1500000u) ? 5 : (a.n_lin >= 200000u ? 4 : 3));
    // ... and a pool of at least ~2048 samples per wave, or the drain phase dominates at low sample counts (10 000
    // primitives, 1080p, 4 spp: 263 ms with 32 pixels per wave, 216 ms with 64) — as long as that leaves two waves
    // for every wave slot of the chip.
    // (the hierarchy kernel with camera batches, whose per-wave LDS grows with the tile: 512 samples — 10 000 primitives,
    // 1080p: 4 spp 3.79 ms with 64 pixels per wave, 4.24 with 32; 16 spp 11.36 against 10.61)
    const bool walk_cam = cfg->math_mode == PT_MATH_FAST && sc->n_bnodes > 0 && sc->d_bfat != nullptr && cfg->bsdf_model == PT_BSDF_REFERENCE &&
                          sc->tune[PT_TUNE_BVH_WALK] != 1 && sc->tune[PT_TUNE_BVH_WALK] != 2;
    const uint32_t pool_min = walk_cam ? 512u : 2048u;
    while (a.tile_log2 < 6 && ((uint32_t)sample_count << a.tile_log2) < pool_min && (a.n_lin >> (a.tile_log2 + 1)) >= 16384u) a.tile_log2++;
    if (sc->tune[PT_TUNE_TILE_LOG2] != 0) a.tile_log2 = sc->tune[PT_TUNE_TILE_LOG2];
    if (cfg->tile_log2 != 0) a.tile_log2 = cfg->tile_log2;
    a.fb_layout = cfg->fb_layout;
    a.n_prims = sc->n_prims; a.n_mats = sc->n_mats;
    a.prims = sc->d_prims; a.mats = sc->d_mats;
    a.pscene = pt_parity_scene(sc);
    a.fdata = sc->d_fdata; a.runs = sc->d_runs; a.n_runs = sc->n_runs; a.fshade = sc->d_fshade;
    a.chunk_runs = sc->d_chunk_runs; a.n_chunks = sc->n_chunks;
    a.sorted = sc->sorted;
    if (sc->n_bnodes > 0 && cfg->math_mode == PT_MATH_FAST) {
        // PT_MATH_FAST walks the hierarchy for the finite primitives and loops over the rest;
        // PT_MATH_FAST_SCALAR and PT_MATH_FAST_LDS keep testing every primitive.
        a.bnodes = sc->d_bnodes; a.bprims = sc->d_bprims; a.n_bnodes = sc->n_bnodes;
        a.bfat = sc->d_bfat; a.walk_stack = sc->fat_depth;
        for (int k = 0; k < 3; k++) { a.fat_origin[k] = sc->fat_origin[k]; a.fat_step[k] = sc->fat_step[k]; }
        if (sc->tune[PT_TUNE_BVH_WALK] == 1) a.bfat = nullptr;
        // camera rays in batches against each tile's own list of hierarchy primitives (bsdf.h shading only, pixels of a
        // tile in one byte); PT_TUNE_BVH_WALK = 2 keeps the loop in which camera rays walk like every other ray
        a.walk_cam = walk_cam;
        a.fdata = sc->d_fdata_top; a.runs = sc->d_runs_top; a.n_runs = sc->n_runs_top;
        a.sorted = sc->sorted_top;
        // measured optimum on the 10 000-primitive scene is flat around these (a visit of the fat form decides
        // about two boxes, so its budget is half the skip-link form's)
        // (with camera batches the walkers are secondary rays only, and a later leaf phase pays: 1080p @ 64 spp 36.9 ms
        // with a leaf gate of 8, 36.4 with 16)
        a.walk_budget = a.bfat ? 16 : 32; a.shade_gate = 8; a.leaf_gate = walk_cam ? 16 : 8;
        if (sc->tune[PT_TUNE_LEAF_GATE] != 0) a.leaf_gate = sc->tune[PT_TUNE_LEAF_GATE];
        if (sc->tune[PT_TUNE_WALK_BUDGET] != 0) a.walk_budget = sc->tune[PT_TUNE_WALK_BUDGET];
        if (sc->tune[PT_TUNE_SHADE_GATE] != 0) a.shade_gate = sc->tune[PT_TUNE_SHADE_GATE];
    } else if (batched) {
        // The queue of set-aside glass hits: as many entries as fit beside the per-wave arrays and the tables while
        // six workgroups still share a compute unit's 160 KiB of LDS.  Measured on the 1080p frame of the default scene
        // (tools/q_sweep.py): 40 entries at six workgroups 3.12 ms, 24 entries at seven 3.19, 48 entries at five 3.37;
        // free lanes take entries back from three quarters full (half full: +1.5 %, a quarter: +5 %).
        const uint32_t P = 1u << a.tile_log2;
        const bool tables = sc->n_prims <= 64 && sc->n_mats <= 32;
        const size_t table_bytes = parity_batched ? (sc->n_mats <= 32 ? (size_t)sc->n_mats * sizeof(DMat) : 0u)
                                                  : (tables ? (size_t)sc->n_prims * sizeof(FShade) + (size_t)sc->n_mats * sizeof(DMat) : 0u);
        const size_t fixed = 4u * ((size_t)P * 40u + 2048u) + table_bytes;
        const size_t budget = 163840u / 6u;
        int cap = fixed < budget ? (int)((budget - fixed) / 256u) & ~3 : 0;
        // (the parity-order loop, whose full shading is several times the fast one's, gains a little from a longer queue:
        // 1080p frame 7.70 ms with 40 entries, 7.61 with 48, 7.81 with 56 at five workgroups)
        cap = std::max(16, std::min(parity_batched ? 48 : 40, cap));
        if (sc->tune[PT_TUNE_GLASS_QUEUE] != 0) cap = sc->tune[PT_TUNE_GLASS_QUEUE];
        int gate = (3 * cap) / 4;
        if (sc->tune[PT_TUNE_GLASS_GATE] != 0) gate = std::min(sc->tune[PT_TUNE_GLASS_GATE], cap);
        a.q_cap = cap; a.q_gate = gate; a.q_inline = cap - gate + 2;   // (q_gate - 1) + (q_inline - 1) = q_cap
        a.slot_box = sc->n_slots <= 64 ? sc->d_slot_box : nullptr; a.n_slots = sc->n_slots;
    }
    a.fb = d_fb;
    if (cfg->math_mode == PT_MATH_PARITY) return ptk_render_parity(a, (hipStream_t)stream);
    if (cfg->math_mode == PT_MATH_FAST || cfg->math_mode == PT_MATH_FAST_SCALAR || cfg->math_mode == PT_MATH_FAST_LDS)
        return ptk_render_fast(a, cfg->math_mode, (hipStream_t)stream);
    return PT_FAIL("pt_render: unknown math_mode");
}

namespace {
// p_color = totalLight / n_samples (main.cu:67) from the 32.32 sums, with the arithmetic of the render kernels' own
// last step (ptw::from_fixed, then one float division), for the pixels of the rendered rows: one thread per pixel.
__global__ void resolve_kernel(const unsigned long long* __restrict__ sums, float* __restrict__ fb, uint32_t n_lin, int max_x, int max_y,
                               int row_begin, int row_step, int compact, int layout, float ns) {
    const uint32_t lin = blockIdx.x * blockDim.x + threadIdx.x;
    if (lin >= n_lin) return;
    const uint32_t k = lin / (uint32_t)max_x, i = lin - k * (uint32_t)max_x;
    const size_t out = compact ? (size_t)lin : (size_t)(row_begin + (int)k * row_step) * (size_t)max_x + i;
    const size_t plane = compact ? (size_t)n_lin : (size_t)max_x * (size_t)max_y;
    for (int c = 0; c < 3; c++) {
        const size_t at = layout == PT_FB_AOS ? out * 3u + (size_t)c : (size_t)c * plane + out;
        const unsigned long long v = sums[at];
        fb[at] = ((float)(int)(v >> 32) + (float)(unsigned)v * 2.3283064365386963e-10f) / ns;
    }
}
}  // namespace

extern "C" {

int pt_render(const pt_scene* sc, const pt_config* cfg, float* d_fb, void* stream) {
    if (!d_fb) return PT_FAIL("pt_render: null argument");
    return render_launch(sc, cfg, d_fb, 0, cfg ? cfg->n_samples : 0, nullptr, nullptr, stream);
}

int pt_render_counted(const pt_scene* sc, const pt_config* cfg, float* d_fb, unsigned long long* d_counters, void* stream) {
    if (!d_fb || !d_counters) return PT_FAIL("pt_render_counted: null argument");
    return render_launch(sc, cfg, d_fb, 0, cfg ? cfg->n_samples : 0, nullptr, d_counters, stream);
}

int pt_render_samples(const pt_scene* sc, const pt_config* cfg, int sample_begin, int sample_count,
                      unsigned long long* d_sums, void* stream) {
    if (!d_sums) return PT_FAIL("pt_render_samples: null argument");
    return render_launch(sc, cfg, nullptr, sample_begin, sample_count, d_sums, nullptr, stream);
}

long long pt_sums_count(const pt_config* cfg) {
    if (!cfg || cfg->max_x <= 0 || cfg->max_y <= 0 || cfg->row_begin < 0 || cfg->row_end > cfg->max_y || cfg->row_begin > cfg->row_end || cfg->row_step < 0) return -1;
    const int step = cfg->row_step == 0 ? 1 : cfg->row_step;
    const long long rows = cfg->fb_compact ? (cfg->row_end - cfg->row_begin + step - 1) / step : cfg->max_y;
    return rows * cfg->max_x * 3;
}

int pt_resolve(const pt_scene* sc, const pt_config* cfg, const unsigned long long* d_sums, float* d_fb, void* stream) {
    if (!sc || !cfg || !d_sums || !d_fb) return PT_FAIL("pt_resolve: null argument");
    if (pt_sums_count(cfg) < 0 || cfg->n_samples <= 0) return PT_FAIL("pt_resolve: bad image size, rows or n_samples");
    if (cfg->fb_layout != PT_FB_AOS && cfg->fb_layout != PT_FB_SOA) return PT_FAIL("pt_resolve: unknown fb_layout");
    const int step = cfg->row_step == 0 ? 1 : cfg->row_step;
    const uint32_t n_lin = (uint32_t)((cfg->row_end - cfg->row_begin + step - 1) / step) * (uint32_t)cfg->max_x;
    if (n_lin == 0) return 0;
    PT_CHECK(hipSetDevice(sc->device));
    hipLaunchKernelGGL(resolve_kernel, dim3((n_lin + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, d_sums, d_fb, n_lin, cfg->max_x, cfg->max_y,
                       cfg->row_begin, step, cfg->fb_compact, cfg->fb_layout, (float)cfg->n_samples);
    PT_CHECK(hipGetLastError());
    return 0;
}

int pt_raypath_batch(const pt_scene* sc, int prim, const float* d_o, const float* d_d, int n_rays, int filter,
                     int32_t* d_n, int32_t* d_hit1, float* d_t1, float* d_pos1, float* d_nrm1,
                     int32_t* d_hit2, float* d_t2, float* d_pos2, float* d_nrm2, void* stream) {
    if (!sc || !d_o || !d_d || n_rays < 0) return PT_FAIL("pt_raypath_batch: bad arguments");
    if (prim < 0 || prim >= sc->n_prims) return PT_FAIL("pt_raypath_batch: primitive index out of range");
    PT_CHECK(hipSetDevice(sc->device));
    return ptk_raypath_batch(sc, prim, d_o, d_d, n_rays, filter, d_n, d_hit1, d_t1, d_pos1, d_nrm1,
                             d_hit2, d_t2, d_pos2, d_nrm2, (hipStream_t)stream);
}

int pt_nearest_batch(const pt_scene* sc, const float* d_o, const float* d_d, int n_rays,
                     int32_t* d_prim, float* d_t, float* d_pos, float* d_nrm, int32_t* d_inside, void* stream) {
    if (!sc || !d_o || !d_d || n_rays < 0) return PT_FAIL("pt_nearest_batch: bad arguments");
    PT_CHECK(hipSetDevice(sc->device));
    return ptk_nearest_batch(sc, d_o, d_d, n_rays, d_prim, d_t, d_pos, d_nrm, d_inside, (hipStream_t)stream);
}

int pt_nearest_batch_mode(const pt_scene* sc, const float* d_o, const float* d_d, int n_rays, int math_mode,
                          int32_t* d_prim, float* d_t, float* d_pos, float* d_nrm, int32_t* d_inside, void* stream) {
    if (!sc || !d_o || !d_d || n_rays < 0) return PT_FAIL("pt_nearest_batch_mode: bad arguments");
    if (math_mode != PT_MATH_PARITY && math_mode != PT_MATH_FAST) return PT_FAIL("pt_nearest_batch_mode: unknown math_mode");
    PT_CHECK(hipSetDevice(sc->device));
    if (math_mode == PT_MATH_PARITY) return ptk_nearest_batch(sc, d_o, d_d, n_rays, d_prim, d_t, d_pos, d_nrm, d_inside, (hipStream_t)stream);
    return ptk_nearest_batch_fast(sc, d_o, d_d, n_rays, d_prim, d_t, d_pos, d_nrm, d_inside, (hipStream_t)stream);
}

static int trace_paths(const char* who, const pt_scene* sc, const float* d_o, const float* d_d, const uint32_t* d_pixel,
                       const uint32_t* d_sample, int n_rays, uint32_t seed, int n_bounces, int math_mode, int bsdf_model,
                       float* d_rgb, float* d_end_o, float* d_end_d, void* stream) {
    if (!sc || !d_o || !d_d || !d_pixel || !d_sample || !d_rgb || n_rays < 0 || n_bounces < 0)
        return pt_set_error(-1, who, __FILE__, __LINE__);
    if (bsdf_model != PT_BSDF_REFERENCE && bsdf_model != PT_BSDF_CLASSES) return pt_set_error(-1, who, __FILE__, __LINE__);
    PT_CHECK(hipSetDevice(sc->device));
    const uint32_t sh = host_hash(seed + 0x9E3779B9u);
    if (math_mode == PT_MATH_PARITY)
        return ptk_trace_paths_parity(sc, d_o, d_d, d_pixel, d_sample, n_rays, sh, n_bounces, bsdf_model, d_rgb, d_end_o, d_end_d, (hipStream_t)stream);
    if (math_mode == PT_MATH_FAST)
        return ptk_trace_paths_fast(sc, d_o, d_d, d_pixel, d_sample, n_rays, sh, n_bounces, bsdf_model, d_rgb, d_end_o, d_end_d, (hipStream_t)stream);
    return pt_set_error(-1, who, __FILE__, __LINE__);
}

int pt_trace_paths(const pt_scene* sc, const float* d_o, const float* d_d, const uint32_t* d_pixel,
                   const uint32_t* d_sample, int n_rays, uint32_t seed, int n_bounces, int math_mode,
                   float* d_rgb, void* stream) {
    return trace_paths("pt_trace_paths: bad arguments or unknown math_mode", sc, d_o, d_d, d_pixel, d_sample, n_rays, seed,
                       n_bounces, math_mode, PT_BSDF_REFERENCE, d_rgb, nullptr, nullptr, stream);
}

int pt_trace_paths_alt(const pt_scene* sc, const float* d_o, const float* d_d, const uint32_t* d_pixel,
                       const uint32_t* d_sample, int n_rays, uint32_t seed, int n_bounces, int math_mode,
                       float* d_rgb, void* stream) {
    return trace_paths("pt_trace_paths_alt: bad arguments or unknown math_mode", sc, d_o, d_d, d_pixel, d_sample, n_rays, seed,
                       n_bounces, math_mode, PT_BSDF_CLASSES, d_rgb, nullptr, nullptr, stream);
}

int pt_trace_paths_end(const pt_scene* sc, const float* d_o, const float* d_d, const uint32_t* d_pixel,
                       const uint32_t* d_sample, int n_rays, uint32_t seed, int n_bounces, int math_mode, int bsdf_model,
                       float* d_rgb, float* d_end_o, float* d_end_d, void* stream) {
    return trace_paths("pt_trace_paths_end: bad arguments, unknown math_mode or bsdf_model", sc, d_o, d_d, d_pixel, d_sample, n_rays, seed,
                       n_bounces, math_mode, bsdf_model, d_rgb, d_end_o, d_end_d, stream);
}

int pt_bsdf_arity(int fn, int* n_in, int* n_out) {
    static const int in[PT_FN_COUNT] = {2, 3, 10, 6, 7, 7, 3}, out[PT_FN_COUNT] = {3, 1, 1, 3, 3, 3, 3};
    if (fn < 0 || fn >= PT_FN_COUNT) return PT_FAIL("pt_bsdf_arity: unknown function");
    if (n_in) *n_in = in[fn];
    if (n_out) *n_out = out[fn];
    return 0;
}

int pt_bsdf_batch(int device, int fn, int math_mode, const float* d_in, int n, float* d_out, void* stream) {
    if (fn < 0 || fn >= PT_FN_COUNT) return PT_FAIL("pt_bsdf_batch: unknown function");
    if (n < 0 || (n > 0 && (!d_in || !d_out))) return PT_FAIL("pt_bsdf_batch: bad arguments");
    if (math_mode != PT_MATH_PARITY && math_mode != PT_MATH_FAST) return PT_FAIL("pt_bsdf_batch: math_mode must be PT_MATH_PARITY or PT_MATH_FAST");
    PT_CHECK(hipSetDevice(device));
    if (math_mode == PT_MATH_PARITY) return ptk_bsdf_batch_parity(fn, d_in, n, d_out, (hipStream_t)stream);
    return ptk_bsdf_batch_fast(fn, d_in, n, d_out, (hipStream_t)stream);
}

int pt_rng_draws(int device, uint32_t seed, uint32_t pixel, uint32_t sample, int n_slots,
                 uint32_t* d_bits, float* d_uniforms, void* stream) {
    if (n_slots < 0) return PT_FAIL("pt_rng_draws: bad arguments");
    PT_CHECK(hipSetDevice(device));
    return ptk_rng_draws(host_hash(seed + 0x9E3779B9u), pixel, sample, n_slots, d_bits, d_uniforms, (hipStream_t)stream);
}

long pt_ppm_p3(const float* fb, int nx, int ny, char* out, long cap) {   // main.cu:240-253
    long n = 0;
    int w = std::snprintf(out, (size_t)cap, "P3\n%d %d\n255\n", nx, ny);
    if (w < 0 || w >= cap) return -1;
    n += w;
    for (int j = ny - 1; j >= 0; j--) {
        for (int i = 0; i < nx; i++) {
            const size_t pixel_index = (size_t)j * nx + i;
            const float r = fb[pixel_index * 3], g = fb[pixel_index * 3 + 1], b = fb[pixel_index * 3 + 2];
            const int ir = int(255.99 * r), ig = int(255.99 * g), ib = int(255.99 * b);
            w = std::snprintf(out + n, (size_t)(cap - n), "%d %d %d\n", ir, ig, ib);
            if (w < 0 || w >= cap - n) return -1;
            n += w;
        }
    }
    return n;
}

}  // extern "C"
