// Shared between the host side of libpt_hip and its kernels: the device
// primitive table, the launch argument blocks and the error plumbing.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pt_hip.h"

// One primitive of the tagged-union table used by the parity-order kernels.
// The record is wave-uniform data: kernels index it with a uniform k so the
// compiler fetches it with scalar loads.  128 bytes.
struct DPrim {
    float inv[12];    // rows of transform_inv  (shapes.h:37)
    float fwd[12];    // rows of transform; wOrigin = (fwd[3], fwd[7], fwd[11]) (shapes.h:36,38)
    float bounds[4];  // Plane xMin, yMin, xMax, yMax (shapes.h:105-109)
    int32_t type;
    int32_t material;
    int32_t index;    // the primitive's index in the scene (the copies sorted by type need it; in scene order it is the position)
    int32_t pad;
};

// Material padded to 80 bytes so a lane can fetch it with five 16-byte reads.
struct DMat {
    float albedo[3];
    float emissive[3];
    float f0;
    float specularRoughness;
    float specularColor[3];
    float IOR;
    float transparency;
    float refractionRoughness;
    float refractionColor[3];
    int32_t bsdfType;
    int32_t emits;    // (fast kernels) some emissive component is > 0 or < 0, i.e. emissive.norm() > 0 (bsdf.h:90): the hit adds light
    float pad;
};

// Intersection data of the fast kernels: one float array, cut into RUNS of
// consecutive primitives of the same kind so that the primitive loop is a tight
// per-kind loop in the scene's own order (ties keep the lowest index, as in
// shapes.h:93).  Record layouts (floats):
//   FP_PLANE (12)     : row z of T^-1 (4) | row x scaled (4) | row y scaled (4)
//                       — the scaled rows map the bounds to |x'| < 1, |y'| < 1
//   FP_PLANE_GEN (16) : rows z, x, y of T^-1 (12) | xMin, yMin, xMax, yMax
//   FP_PLANE_X/Y/Z (5): a bounded plane whose normal and edges run along world axes (normal along x/y/z, edges
//                       along the other two, u < v): plane coordinate | centre, half-width along u | the same along v
//   FP_SPHERE_WS (6)  : a sphere under L = s*I: k = 1/s | q = -centre/s | k^2 | 1/k^2  (the reference's local frame
//                       o_l = k*o + q without the matrix; the name is historical: it used to be a world-space test)
//   FP_SPHERE, FP_CUBE, FP_CYLINDER (12): rows of T^-1
//   FP_CUBE_AA (6)    : a cube under a diagonal L (no rotation): world-space box, low corner | high corner
// The float array is cut into chunks of FDATA_CHUNK floats; no record and no run crosses a chunk boundary, so a
// workgroup can stage one chunk at a time in LDS (the LDS-tiled primitive loop for large scenes).
constexpr int FDATA_CHUNK = 4096;

enum { FP_PLANE = 0, FP_CUBE = 1, FP_CYLINDER = 2, FP_SPHERE = 3, FP_SPHERE_WS = 4, FP_PLANE_GEN = 5,
       FP_PLANE_X = 6, FP_PLANE_Y = 7, FP_PLANE_Z = 8, FP_CUBE_AA = 9, FP_KINDS = 10 };

// The same records sorted by kind (scene order within a kind), each followed by its primitive's scene index << 3 and
// padded to a fixed stride per kind: what PT_MATH_FAST loops over.  No run table: six typed loops whose offsets and
// counts travel in the kernel arguments.  The (distance, index) rule of the tests makes the result independent of
// the order in which primitives are tested.
struct FSorted {
    const float* data;
    int32_t off[FP_KINDS];     // by FP_*: first float of the kind's records
    int32_t count[FP_KINDS];
};
constexpr int FSORTED_STRIDE[FP_KINDS] = {16, 16, 16, 16, 8, 20, 8, 8, 8, 8};   // floats per record by FP_*; the index sits after the record

struct FRun {
    int32_t type;     // FP_*
    int32_t first;    // index of the run's first primitive in the scene
    int32_t count;
    int32_t offset;   // of the run's first record in the float array
};
// Shading record of the fast kernels, fetched per lane for the winning primitive.
//   plane   : a[0..2] = unit world normal (third column of L)
//   sphere  : a[0..2] = centre (translation)
//   cube    : a[3j..3j+2] = unit column j of L
//   cylinder: a[0..2], a[3..5] = columns 0,1 of L, a[6..8] = unit column 2,
//             a[9..12], a[13..16] = rows x,y of T^-1
struct FShade {
    float a[20];
    int32_t type;
    int32_t material;
    int32_t pad[2];
};

// Bounding-volume hierarchy over the finite primitives (everything but planes) of scenes with many of them.
// Nodes are stored in depth-first order: the first child of an inner node is the next node, `skip` is the node
// to go to when the box is missed or its subtree is finished (the node count ends the walk), so the walk needs no
// stack.  A leaf owns `leaf & 15` consecutive BPrim records starting at `leaf >> 4`; inner nodes have leaf = -1.
struct BNode {
    float lo[3];
    int32_t skip;
    float hi[3];
    int32_t leaf;
};
// One primitive as the hierarchy's leaves hold it: the same record the primitive loop reads (see above), its kind
// and its scene index << 3.  64 bytes, fetched per lane.
struct BPrim {
    float rec[12];
    int32_t kind;     // FP_*
    int32_t id;
    int32_t pad[2];
};
// The same tree with both children's boxes in the parent, for the ordered walk with a per-lane stack: one visit
// fetches 32 bytes and decides about two subtrees.  A box is held per axis as the sum and the difference of its two
// planes' coordinates on the 15-bit grid that spans the box of all finite primitives (plane = origin + step *
// (sum -+ diff), step = half a grid cell; pt_scene::fat_origin / fat_step; lo rounded down to the grid, hi rounded up,
// so a box only grows): the walk is bound by the vector memory path and then by instruction issue, a node of floats
// would be twice the fetches, and from centre and half width the slab distances come as three multiply-adds per
// axis, without the minimum / maximum that picks the near plane.  A child reference is an index into this array
// (>= 0) or, for a one-primitive leaf, ~(index into the BPrim array).  Built when every leaf holds one primitive
// and the tree has at most 2^19 of them (references are packed into 20 bits on the stack).
struct BFat {
    uint16_t sum_a[3], diff_a[3];
    uint16_t sum_b[3], diff_b[3];
    int32_t ref_a, ref_b;
};
static_assert(sizeof(BFat) == 32, "two 16-byte fetches per node");
constexpr int BFAT_MAX_PRIMS = 1 << 19;
constexpr int WALK_STACK_MAX = 24;    // stack entries per lane: the builder keeps 2^19 leaves within 19 + 5 levels
constexpr int WALK_NONE = 0x7fffffff; // "no node in hand": the next one comes off the stack
constexpr int BVH_MIN_PRIMS = 60;    // fewer finite primitives than this: no hierarchy, the loop tests them all (measured
                                     // at 1080p: 48 small scattered shapes 3.06 ms through the hierarchy, 2.43 through the
                                     // loop; 64: 3.24 vs 3.92; 8x8 large glass spheres at depth 31: 3.55 vs 4.09)
constexpr int BVH_SPLIT = 4;          // PT_BVH_SPLIT_AREA + 2: 10 000 small shapes 18.35 ms (median split 20.66, + 0: 18.93,
                                      // + 3: 18.36), 10x10 large glass spheres at depth 31 3.87 ms (5.44, 4.45, 3.87)
constexpr int BVH_LEAF_PRIMS = 1;     // measured on the 10 000-primitive scene: 1 is fastest (24.4 ms), 4: 27.7, 8: 29.5



// What a parity-order nearest-hit search reads of a scene: the primitive table and, when the scene has a hierarchy
// (nodes != null), its skip-link form and the scene indices, ascending, of the primitives outside it (the planes).
struct PScene {
    const DPrim* prims;
    int32_t n_prims;
    // The same records sorted by shape type (PT_PLANE, PT_CUBE, PT_CYLINDER, PT_SPHERE; scene order within a type):
    // records [type_first[t], type_first[t + 1]) are of type t.  The loop with camera batches runs one typed loop per
    // shape type over them, which the (distance, scene index) rule allows.  `boxes` holds a padded world-space box
    // (lo.xyz, hi.xyz) per record of `by_type`, for the per-tile culling of camera rays, or is null.
    const DPrim* by_type;
    int32_t type_first[5];
    const float* boxes;
    const BNode* nodes;
    int32_t n_nodes;
    const BPrim* bprims;
    const int32_t* top;
    int32_t n_top;
};

struct pt_scene {
    int device;
    int n_prims;
    int n_mats;
    DPrim* d_prims;
    DPrim* d_prims_by_type;                // the same sorted by shape type, and a world-space box for each of them
    int type_first[5];
    float* d_type_box;
    int typed_loops_ok;                    // every entry of every T and T^-1 is below 1e5 in magnitude (see ptp::degenerate_ray)
    DMat* d_mats;
    float* d_fdata;        // fast-kernel intersection records
    FRun* d_runs;
    int n_runs;
    int32_t* d_chunk_runs; // runs [d_chunk_runs[c], d_chunk_runs[c+1]) lie in chunk c
    int n_chunks;
    FShade* d_fshade;
    // With a hierarchy (n_bnodes > 0): the primitives outside it (planes) as their own record array and runs.
    BNode* d_bnodes;
    int n_bnodes;
    BPrim* d_bprims;
    float* d_sorted;                       // all primitives sorted by kind
    FSorted sorted;
    float* d_sorted_top;                   // the primitives outside the hierarchy sorted by kind
    FSorted sorted_top;
    BFat* d_bfat;                          // or null: only the skip-link form exists
    int fat_depth;                         // depth of its deepest leaf = stack entries a lane can need
    int32_t* d_top_ids;                    // scene indices of the primitives outside the hierarchy, ascending
    int n_top_ids;
    float fat_origin[3], fat_step[3];      // the grid of its box coordinates
    float* d_fdata_top;
    FRun* d_runs_top;
    int n_runs_top;
    float max_emissive;                    // largest finite |emissive component| of the materials
    int tune[PT_TUNE_COUNT];               // pt_scene_tune: 0 = the library's choice
    float* d_slot_box;                     // world-space box (lo.xyz, hi.xyz) of every record of `sorted`, in its order
    int n_slots;
};

// Camera constants computed on the host in the reference's operation order
// (main.cu:40-52) and the per-launch parameters of the render kernels.
struct RenderArgs {
    float cam[3];
    float p00[3];          // pixel00_loc
    float du, dv;          // pixel_delta_u.x, pixel_delta_v.y
    int32_t max_x, max_y;
    int32_t n_samples, n_bounces;   // n_samples: samples per pixel THIS launch traces, numbered sample_begin ... (RNG streams go by that number)
    int32_t sample_begin;
    unsigned long long* counters;   // null, or where every wave adds what it traced: [0] path segments (nearest-hit searches begun)
    unsigned long long* sums;       // null: the launch writes fb = sum / n_samples; else it ADDS its pixels' 32.32 sums here (pt_render_samples)
    float sample_cap;      // 2^30 / (samples of the whole frame): what one sample's radiance counts as at most (pixel sums are 32.32 fixed point)
    int32_t bsdf_model;    // PT_BSDF_*
    uint32_t seed_hash;    // H(seed + 0x9E3779B9)
    int32_t row_begin, row_step, n_rows;   // rendered rows: row_begin + k*row_step, k in [0, n_rows)
    uint32_t n_lin;                        // n_rows * max_x: pixels of this launch, numbered row by row
    int32_t fb_compact;                    // rendered row k is stored as row k (1) or at its image row (0)
    int32_t tile_log2;                     // pixels per wave tile = 1 << tile_log2 (2..6)
    int32_t fb_layout;
    int32_t n_prims, n_mats;
    PScene pscene;                         // the same scene for the parity-order search
    const DPrim* prims;
    const DMat* mats;
    const float* fdata;
    const FRun* runs;
    int32_t n_runs;
    const int32_t* chunk_runs;
    int32_t n_chunks;
    const FShade* fshade;
    const BNode* bnodes;                   // hierarchy over the finite primitives, or n_bnodes = 0
    const BPrim* bprims;
    FSorted sorted;                        // PT_MATH_FAST: the primitives the typed loops test (all, or those outside the hierarchy)
    const BFat* bfat;                      // non-null: walk this form (ordered, per-lane stack in LDS)
    float fat_origin[3], fat_step[3];
    int32_t walk_stack;                    // stack entries per lane the launch provides (dynamic LDS)
    int32_t n_bnodes;
    int32_t walk_budget;                   // hierarchy nodes a lane visits per loop iteration
    int32_t shade_gate;                    // lanes whose search is over before they are shaded together
    int32_t leaf_gate;                     // lanes holding a leaf before the wave leaves the node loop to test them
    int32_t walk_cam;                      // 1: camera rays in batches against the tile's own list of hierarchy primitives
    // PT_MATH_FAST without a hierarchy: hits that need the full shading (glass, paths inside a shape) are set aside
    // in a per-wave queue in LDS and shaded together.  q_cap = 0: the plain loop.
    int32_t q_cap;                         // queue entries per wave (64 bytes each)
    int32_t q_gate;                        // entries from which free lanes take them back
    int32_t q_inline;                      // such hits in one iteration from which they stay in registers instead; (q_gate-1)+(q_inline-1) <= q_cap
    const float* slot_box;                 // boxes of the records of `sorted` (6 floats each), or null: camera rays test every record
    int32_t n_slots;
    float* fb;
};

// Kernel launchers implemented in the .hip translation units.
PScene pt_parity_scene(const pt_scene* sc);
int ptk_render_parity(const RenderArgs& a, hipStream_t s);
int ptk_render_fast(const RenderArgs& a, int variant, hipStream_t s);   // variant: PT_MATH_FAST (auto), _SCALAR, _LDS
int ptk_raypath_batch(const pt_scene* sc, int prim, const float* o, const float* d, int n, int filter,
                      int32_t* dn, int32_t* h1, float* t1, float* p1, float* n1,
                      int32_t* h2, float* t2, float* p2, float* n2, hipStream_t s);
int ptk_nearest_batch(const pt_scene* sc, const float* o, const float* d, int n, int32_t* prim, float* t,
                      float* pos, float* nrm, int32_t* inside, hipStream_t s);
int ptk_trace_paths_parity(const pt_scene* sc, const float* o, const float* d, const uint32_t* pixel,
                           const uint32_t* sample, int n, uint32_t seed_hash, int n_bounces, int bsdf_model, float* rgb,
                           float* end_o, float* end_d, hipStream_t s);
int ptk_trace_paths_fast(const pt_scene* sc, const float* o, const float* d, const uint32_t* pixel,
                         const uint32_t* sample, int n, uint32_t seed_hash, int n_bounces, int bsdf_model, float* rgb,
                         float* end_o, float* end_d, hipStream_t s);
int ptk_bsdf_batch_parity(int fn, const float* in, int n, float* out, hipStream_t s);
int ptk_nearest_batch_fast(const pt_scene* sc, const float* o, const float* d, int n, int32_t* prim, float* t,
                           float* pos, float* nrm, int32_t* inside, hipStream_t s);
int ptk_bsdf_batch_fast(int fn, const float* in, int n, float* out, hipStream_t s);
int ptk_rng_draws(uint32_t seed_hash, uint32_t pixel, uint32_t sample, int n_slots, uint32_t* bits, float* u, hipStream_t s);

// Records "HIP error = <n> at <file>:<line> '<expr>'" for pt_last_error().
int pt_set_error(int code, const char* expr, const char* file, int line);
#define PT_CHECK(expr)                                                        \
    do {                                                                      \
        hipError_t _e = (expr);                                               \
        if (_e != hipSuccess) return pt_set_error((int)_e, #expr, __FILE__, __LINE__); \
    } while (0)
