/* pt_hip.h — C ABI of libpt_hip.so, the MI355X (gfx950) path-tracing kernels.
 *
 * The reference has no FFI: its hot path is header-inline __device__ C++ called
 * from one __global__ kernel (main.cu:6-8,65).  This ABI is the boundary a host
 * program binds instead of launching that kernel: plain pointers and sizes, no
 * C++ types.  Every entry point returns 0 on success or a non-zero HIP/PT status
 * and never exits the process; pt_last_error() gives the text in the reference's
 * check_cuda shape (main.cu:11-19): "HIP error = <n> at <file>:<line> '<expr>'".
 *
 * Pointers named d_* are DEVICE pointers on the scene's device; everything else
 * is host memory.  Calls are stream-ordered on `stream` (a hipStream_t passed as
 * void*, NULL = the default stream) and do not synchronise the device.
 */
#ifndef PT_HIP_H
#define PT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Shape tags (class hierarchy of shapes.h:103,137,225,323). */
enum { PT_PLANE = 0, PT_CUBE = 1, PT_CYLINDER = 2, PT_SPHERE = 3 };

/* Material, field for field (materials.h:12-38); 72 bytes like the reference's. */
typedef struct pt_material {
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
} pt_material;

/* One Shape as its constructor arguments (shapes.h:111,147,229,325):
 * transform = rows of the affine [L | t] (Eigen::Affine3f without its last row),
 * bounds = Plane xMin,yMin,xMax,yMax (ignored by the other types). */
typedef struct pt_prim {
    int32_t type;
    int32_t material;
    float transform[12];
    float bounds[4];
} pt_prim;

/* Math mode of pt_render. */
enum {
    PT_MATH_PARITY = 0, /* reference operation order, IEEE div/sqrt, no FMA contraction; a scene with a bounding-volume
                         * hierarchy (below) is searched through it, with the same result as the reference's loop */
    PT_MATH_FAST = 1,   /* same algorithm, FMA + v_rcp/v_rsq/v_sin/v_cos, restructured tests; scenes with
                         * 60 or more finite primitives are searched through a bounding-volume hierarchy
                         * (same nearest hit as the loop of shapes.h:82-98, lowest index on ties) */
    /* The loop over ALL primitives, in its two forms (same results bit for bit): */
    PT_MATH_FAST_SCALAR = 2, /* primitive records fetched by scalar loads */
    PT_MATH_FAST_LDS = 3     /* primitive records staged chunk by chunk in LDS by the workgroup */
};
/* Shading model: which tracePath the kernels run. */
enum {
    PT_BSDF_REFERENCE = 0, /* tracePath of bsdf.h:73-168 — what main.cu compiles */
    PT_BSDF_CLASSES = 1    /* tracePath of bsdf_alternative.h:153-219: Lambertian / SmoothDielectric classes chosen by
                            * Material::bsdfType (materials.h:38); that file is not compiled by the reference */
};
/* Framebuffer layout of pt_render. */
enum {
    PT_FB_AOS = 0, /* fb[(j*max_x+i)*3 + c]  — the reference's vec3* fb (main.cu:68) */
    PT_FB_SOA = 1  /* fb[c*max_x*max_y + j*max_x + i] — three planes */
};

/* config_t (main.cu:22-28) followed by the values the reference hard-codes:
 * n_samples (main.cu:54), n_bounces (bsdf.h:75), the RNG seed (main.cu:60 seeds
 * by pixel index only; seed selects an independent family of streams), and the
 * rows this call renders: row_begin, row_begin + row_step, ... below row_end
 * (row 0 = bottom of the image, main.cu:38; row_step 0 means 1).  With
 * fb_compact = 0 a rendered row is written at its place in the full framebuffer;
 * with fb_compact = 1 the k-th rendered row is written as row k of d_fb, which
 * then only needs room for the rendered rows (interleaved multi-GPU tiles).
 * tile_log2: log2 of the pixels one wave64 renders (2..6), 0 = chosen from the
 * launch size.  The image does not depend on it.  The automatic choice suits a
 * launch that has the GPU to itself; a host that keeps several launches in flight
 * on different streams (they fill each other's tails) does better with 5 or 6. */
typedef struct pt_config {
    float camera_pos[3];
    int32_t max_x;
    int32_t max_y;
    float focal_length;
    float view_port_height;
    int32_t n_samples;
    int32_t n_bounces;
    uint32_t seed;
    int32_t row_begin;
    int32_t row_end;
    int32_t fb_layout;
    int32_t math_mode;
    int32_t row_step;
    int32_t fb_compact;
    int32_t bsdf_model;
    int32_t tile_log2;
} pt_config;

typedef struct pt_scene pt_scene; /* opaque; owns the device copy of the primitive table */

/* Replaces cudaMalloc + constructScene<<<1,1>>> (main.cu:213-220): flattens the
 * shapes into the device primitive table on `device`. */
int pt_scene_upload(const pt_prim* prims, int n_prims, const pt_material* mats, int n_mats, int device, pt_scene** out);
/* The same with explicit build options (NULL = the defaults, which is what pt_scene_upload uses).  No counterpart
 * in the reference.  They change how PT_MATH_FAST searches the scene, never which hit it finds. */
enum { PT_GENERAL_PLANES = 1, PT_GENERAL_CUBES = 2, PT_GENERAL_SPHERES = 4, PT_GENERAL_ALL = 7 };
enum { PT_BVH_SPLIT_MEDIAN = 1, PT_BVH_SPLIT_AREA = 2 };
typedef struct pt_upload_options {
    int32_t bvh_min_prims;  /* finite primitives (spheres, cubes, cylinders) from which PT_MATH_FAST builds a
                             * bounding-volume hierarchy; 0 = the default (60), values >= 2 */
    int32_t bvh_leaf_prims; /* primitives per leaf, 1..15; 0 = the default (1) */
    int32_t general_kinds;  /* PT_GENERAL_* bits: these shapes keep the general record kinds, tested in local space
                             * as shapes.h does, instead of the shorter world-space tests that axis-aligned planes,
                             * unrotated cubes and uniformly scaled spheres otherwise get */
    int32_t bvh_split;      /* how the hierarchy divides a set of primitives: PT_BVH_SPLIT_MEDIAN = in halves along the
                             * widest axis; PT_BVH_SPLIT_AREA + s (s = 0..5) = where the surface-area cost is least,
                             * leaves at most s levels deeper than the even tree's; 0 = the default */
} pt_upload_options;
int pt_scene_upload_opt(const pt_prim* prims, int n_prims, const pt_material* mats, int n_mats, int device,
                        const pt_upload_options* opt, pt_scene** out);
int pt_scene_free(pt_scene* scene);
/* Launch parameters of pt_render for this scene, by knob; the image never depends on them.  value 0 = the library's
 * own (measured) choice.  No counterpart in the reference.  Not to be called while a pt_render call on the same
 * scene runs in another thread. */
enum {
    PT_TUNE_TILE_LOG2 = 0,   /* log2 of the pixels per wave (2..6) when pt_config.tile_log2 is 0 */
    PT_TUNE_WALK_BUDGET = 1, /* hierarchy nodes a lane visits (or takes off its stack) per loop iteration (>= 1) */
    PT_TUNE_SHADE_GATE = 2,  /* lanes whose hierarchy search is over before they are shaded together (1..64) */
    PT_TUNE_BVH_WALK = 3,    /* 1: PT_MATH_FAST walks the hierarchy in its skip-link form instead of the two-children-per-node
                              * form; 2: the two-children form, but camera rays walk it like every other ray instead of being
                              * traced in batches against their tile's own list of primitives; 3: PT_MATH_PARITY loops over
                              * all primitives although the scene has a hierarchy */
    PT_TUNE_GLASS_QUEUE = 4, /* PT_MATH_FAST, no hierarchy: entries per wave of the LDS queue in which hits on glass
                              * wait to be shaded together (8..64); -1: no queue, every hit is shaded where it is found */
    PT_TUNE_GLASS_GATE = 5,  /* queue entries from which free lanes take them back (1..queue size) */
    PT_TUNE_LEAF_GATE = 6,   /* lanes holding a leaf of the hierarchy from which the wave stops descending and tests
                              * them together (1..64; 64 = only when no lane can descend any further) */
    PT_TUNE_COUNT = 7
};
int pt_scene_tune(pt_scene* scene, int knob, int value);
int pt_scene_device(const pt_scene* scene);
/* Nodes of the scene's bounding-volume hierarchy (0 = the scene has none: every
 * primitive is tested for every ray).  No counterpart in the reference, whose
 * getNearestIntersection always loops over all objects (shapes.h:82). */
int pt_scene_bvh_nodes(const pt_scene* scene);

/* The reference's default scene as data (main.cu:73-195): writes 11 primitives
 * and 8 materials; returns the primitive count. */
int pt_default_scene(pt_prim* prims, pt_material* mats, int* n_mats);
/* Stress scene: the default box, light and cube plus an n x n grid of glass
 * spheres (material of main.cu:150-157).  Returns the primitive count
 * (8 + n*n); prims/mats must hold that many / 8 entries. */
int pt_glass_grid_scene(int n, pt_prim* prims, pt_material* mats, int* n_mats);
/* Synthetic scene: the whole default scene (11 primitives) plus n_spheres spheres
 * and n_cubes cubes placed by the counter RNG with `seed`.  Returns the
 * primitive count (11 + n_spheres + n_cubes). */
int pt_synthetic_scene(int n_spheres, int n_cubes, uint32_t seed, pt_prim* prims, pt_material* mats, int* n_mats);

/* Scene description text -> the same tables the builders above fill (grammar in
 * pathtracer_amd/csrc/pt_io.cpp; scenes/default.scene is main.cu:73-195 written
 * in it).  Returns the primitive count, or -1 with the reason in pt_last_error().
 * No counterpart in the reference, whose scene is compiled in (main.cu:71-198). */
int pt_scene_parse(const char* text, pt_prim* prims, int cap_prims, pt_material* mats, int cap_mats, int* n_mats);

/* Host-side focal length exactly as main.cu:229-232 computes it (integer nx/ny). */
float pt_focal_length(int nx, int ny, float v_height, float fov_deg);

/* Replaces the render<<<blocks,threads>>> launch (main.cu:236).  d_fb is the
 * max_x*max_y*3 float framebuffer in device memory (or, with fb_compact, room
 * for the rendered rows); only the rendered rows are written.  The image is
 * bitwise the same however the rows are split over calls or GPUs.  Pixel sums
 * (totalLight, main.cu:61-67) are kept as signed 32.32 fixed point: a sample's
 * radiance is resolved to 2^-32 and counts up to +-2^30 / n_samples per channel
 * (an infinite radiance included), so a sum never leaves the range; a NaN sample
 * adds nothing where the reference would turn the pixel NaN.  A scene whose own
 * lights exceed that cap (n_samples x largest emissive component > 2^30) is
 * refused.  n_samples is at most 2^24, n_bounces below 2^24. */
int pt_render(const pt_scene* scene, const pt_config* cfg, float* d_fb, void* stream);

/* pt_render that also counts what it traced: every wave ADDS to d_counters (PT_COUNTER_COUNT 64-bit integers on the
 * scene's device, zeroed by the caller): [PT_COUNTER_SEGMENTS] path segments, i.e. calls of getNearestIntersection
 * (bsdf.h:81).  Same image.  For run statistics (segments per sample, an estimate of the arithmetic done); the
 * reference has no counterpart. */
enum { PT_COUNTER_SEGMENTS = 0, PT_COUNTER_COUNT = 4 };
int pt_render_counted(const pt_scene* scene, const pt_config* cfg, float* d_fb, unsigned long long* d_counters, void* stream);

/* The sample loop of main.cu:61-67 in pieces.  pt_render_samples traces samples [sample_begin, sample_begin +
 * sample_count) of the cfg->n_samples that make up the frame (sample s of a pixel draws from the same random stream
 * whichever call traces it) and ADDS each pixel's sum of radiance, as signed 32.32 fixed point, to d_sums: 64-bit
 * integers laid out like the framebuffer (pt_sums_count(cfg) of them: rows x max_x x 3, the rows being all of the image
 * or, with fb_compact, the rendered ones), zeroed by the caller before the first piece (hipMemsetAsync).  The additions
 * are atomic, so pieces may run on different streams.  pt_resolve writes d_fb = sum / n_samples for the rendered rows.
 * Integer sums do not depend on how the samples were cut up: the frame is pt_render's bit for bit.  What a long render
 * needs to be resumed is d_sums and the number of samples done.  The reference has no counterpart (its loop runs inside
 * one launch). */
int pt_render_samples(const pt_scene* scene, const pt_config* cfg, int sample_begin, int sample_count,
                      unsigned long long* d_sums, void* stream);
long long pt_sums_count(const pt_config* cfg);   /* -1: bad size or rows */
int pt_resolve(const pt_scene* scene, const pt_config* cfg, const unsigned long long* d_sums, float* d_fb, void* stream);

/* Shape::getIntersections (filter != 0; shapes.h:60-75) or Shape::getRayPath
 * (filter == 0, rays taken as already local; shapes.h:58) of primitive `prim`
 * for n_rays rays.  d_o, d_d: [n_rays][3].  Outputs (all device, any may be
 * NULL): d_n[n_rays] = n_intersections; for first (1) and second (2):
 * d_hit, d_t [n_rays]; d_pos, d_nrm [n_rays][3]. */
int pt_raypath_batch(const pt_scene* scene, int prim, const float* d_o, const float* d_d, int n_rays, int filter,
                     int32_t* d_n, int32_t* d_hit1, float* d_t1, float* d_pos1, float* d_nrm1,
                     int32_t* d_hit2, float* d_t2, float* d_pos2, float* d_nrm2, void* stream);

/* getNearestIntersection (shapes.h:79-100) for n_rays rays: winning primitive
 * index (-1 = miss), distance, world position, facing normal, inside flag. */
int pt_nearest_batch(const pt_scene* scene, const float* d_o, const float* d_d, int n_rays,
                     int32_t* d_prim, float* d_t, float* d_pos, float* d_nrm, int32_t* d_inside, void* stream);

/* The same through the PT_MATH_FAST kernels' search (kind-sorted records, and the hierarchy when the scene has
 * one): what pt_render(PT_MATH_FAST) finds per segment, reachable on its own so that its distances, positions and
 * normals can be compared with pt_nearest_batch's.  math_mode: PT_MATH_PARITY (= pt_nearest_batch) or PT_MATH_FAST. */
int pt_nearest_batch_mode(const pt_scene* scene, const float* d_o, const float* d_d, int n_rays, int math_mode,
                          int32_t* d_prim, float* d_t, float* d_pos, float* d_nrm, int32_t* d_inside, void* stream);

/* tracePath (bsdf.h:73-168) for n_rays explicit rays; ray i draws from the
 * stream (seed, d_pixel[i], d_sample[i]) starting at slot 2.  d_rgb: [n_rays][3]. */
int pt_trace_paths(const pt_scene* scene, const float* d_o, const float* d_d, const uint32_t* d_pixel,
                   const uint32_t* d_sample, int n_rays, uint32_t seed, int n_bounces, int math_mode,
                   float* d_rgb, void* stream);

/* The same for the tracePath of bsdf_alternative.h:153-219 (PT_BSDF_CLASSES). */
int pt_trace_paths_alt(const pt_scene* scene, const float* d_o, const float* d_d, const uint32_t* d_pixel,
                       const uint32_t* d_sample, int n_rays, uint32_t seed, int n_bounces, int math_mode,
                       float* d_rgb, void* stream);

/* tracePath of either file with its side effect: the reference takes `ray& r` and leaves in it the last ray it
 * assigned (bsdf.h:165 / bsdf_alternative.h:215) — the ray of the segment on which the path ended (miss, roulette),
 * or the ray after the last bounce when the path ran out of bounces.  d_end_o, d_end_d: [n_rays][3], either may be
 * NULL.  bsdf_model: PT_BSDF_REFERENCE or PT_BSDF_CLASSES. */
int pt_trace_paths_end(const pt_scene* scene, const float* d_o, const float* d_d, const uint32_t* d_pixel,
                       const uint32_t* d_sample, int n_rays, uint32_t seed, int n_bounces, int math_mode, int bsdf_model,
                       float* d_rgb, float* d_end_o, float* d_end_d, void* stream);

/* The helper functions of bsdf.h and vec_math.h on arrays, one element per thread, in either family of kernels:
 * what tracePath calls per segment, reachable on its own (the reference has no such entry point: they are
 * __device__ functions).  d_in: [n][n_in] floats, d_out: [n][n_out] floats, both on `device`.
 *   fn                        n_in                               n_out   reference
 *   PT_FN_SAMPLE_HEMISPHERE   2: u1 u2                           3       sampleHemisphere, bsdf.h:29-36
 *   PT_FN_FRESNEL_SCHLICK     3: n1 n2 cosTheta                  1       FresnelSchlick, bsdf.h:48-51
 *   PT_FN_FRESNEL_RATIO       10: N(3) I(3) f0 f90 n1 n2         1       getFresnelRatio, bsdf.h:53-69
 *   PT_FN_REFLECT             6: v(3) n(3)                       3       reflect, vec_math.h:19-21
 *   PT_FN_REFRACT             7: I(3) N(3) eta                   3       refract, vec_math.h:23-30 (0 vector on total reflection)
 *   PT_FN_LERP                7: v0(3) v1(3) t                   3       lerp<vec3>, vec_math.h:10-12
 *   PT_FN_NORMALIZED          3: v                               3       Eigen's .normalized() (0 vector stays 0)
 * math_mode: PT_MATH_PARITY (the reference's operation order) or PT_MATH_FAST (what the fast kernels evaluate:
 * v_sin/v_cos in revolutions, v_rcp, v_rsq, fused multiply-adds). */
enum { PT_FN_SAMPLE_HEMISPHERE = 0, PT_FN_FRESNEL_SCHLICK = 1, PT_FN_FRESNEL_RATIO = 2, PT_FN_REFLECT = 3, PT_FN_REFRACT = 4,
       PT_FN_LERP = 5, PT_FN_NORMALIZED = 6, PT_FN_COUNT = 7 };
int pt_bsdf_batch(int device, int fn, int math_mode, const float* d_in, int n, float* d_out, void* stream);
/* Floats per element going in and coming out of `fn` (0 on success). */
int pt_bsdf_arity(int fn, int* n_in, int* n_out);

/* Draws of the counter RNG for slots [0,n_slots) of stream (seed, pixel, sample), computed on the device: the hash
 * bits of each slot, and the uniform the kernels use there.  Slots 0,1 are the pixel jitter; segment s uses slots
 * 2+6s+{0 mode, 1,2 diffuse u1,u2, 3,4 roughness u1,u2, 5 roulette}.  A draw that decides a branch (mode, roulette)
 * takes all 32 bits of its slot: bits * 2^-32 + 2^-33, in (0,1].  The two coordinates of a point (jitter; u1,u2) share
 * the bits of the pair's first slot, 16 bits each: half * 2^-16 + 2^-17, in (0,1). */
int pt_rng_draws(int device, uint32_t seed, uint32_t pixel, uint32_t sample, int n_slots,
                 uint32_t* d_bits, float* d_uniforms, void* stream);

/* ---- Several GPUs on one frame ------------------------------------------------
 * The reference launches render once on one GPU (main.cu:236); nothing below has
 * a counterpart there.  Rank r of `world` renders image rows r, r+world, ... with
 * pt_render (row_begin = r, row_step = world, fb_compact = 1) into a tile of
 * pt_tile_rows(height, world) rows; pt_gather_framebuffer all-gathers the tiles
 * with RCCL (over xGMI inside a node) and writes the rows in image order, so every
 * rank ends up with the whole AoS frame, bit for bit the single-GPU one.
 * RCCL is loaded on the first pt_comm_* call, not when this library is. */
typedef struct pt_comm pt_comm;
#define PT_COMM_ID_BYTES 128

/* ceil(height / world): rows per rank in the gathered layout (the last ranks'
 * tiles may hold one row fewer; their last row is padding). */
int pt_tile_rows(int height, int world);

/* One process per GPU: rank 0 makes the id, the host program hands a copy to
 * every rank (its own launcher, a file, MPI ...), all ranks then call
 * pt_comm_init_rank, which blocks until all have.  (ncclGetUniqueId /
 * ncclCommInitRank.) */
int pt_comm_unique_id(char id[PT_COMM_ID_BYTES]);
int pt_comm_init_rank(const char id[PT_COMM_ID_BYTES], int rank, int world, int device, pt_comm** out);
/* One process driving `world` GPUs, one host thread per GPU: fills out[0..world)
 * (ncclCommInitAll).  The threads then call pt_gather_framebuffer concurrently,
 * each with its own communicator. */
int pt_comm_init_all(int world, const int* devices, pt_comm** out);
int pt_comm_free(pt_comm* comm);
int pt_comm_rank(const pt_comm* comm);
int pt_comm_world(const pt_comm* comm);

/* d_tile: this rank's compact tile, pt_tile_rows x width x 3 floats;
 * d_gathered: world x pt_tile_rows x width x 3 floats of scratch owned by the
 * caller; d_full: height x width x 3, the assembled frame.  All on the
 * communicator's device, ordered on `stream`; every rank must call it. */
int pt_gather_framebuffer(pt_comm* comm, const float* d_tile, float* d_gathered, float* d_full,
                          int height, int width, void* stream);
/* pt_gather_framebuffer for callers that cannot allocate or bring a workspace
 * (code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc).  No counterpart in the reference.  The gathered tiles are staged
 * in a static device array inside the library instead of d_gathered, so the
 * call itself allocates nothing; the array's per-device address is resolved
 * when the communicator is made (pt_comm_init_rank / pt_comm_init_all).
 * Limit: world x pt_tile_rows(height, world) x width x 3 floats must not exceed
 * PT_GATHER_NOALLOC_MAX_FLOATS (32 MiB: a 1920x1080 frame for any world <= 8);
 * larger frames are refused with an error and need pt_gather_framebuffer.
 * All calls on one GPU share that GPU's array: calls that overlap in time on
 * one GPU (different streams or communicators) race; order them.
 * Cost: the array is part of the library's code object, so its 32 MiB are
 * resident on every GPU a process uses libpt_hip.so on, from the moment the
 * code object is loaded there until the process ends, whether the process ever
 * gathers or not. */
#define PT_GATHER_NOALLOC_MAX_FLOATS (8u << 20)
int pt_gather_framebuffer_noalloc(pt_comm* comm, const float* d_tile, float* d_full, int height, int width, void* stream);
/* The reordering step alone: gathered[r][k] -> image row k*world + r, launched on
 * `stream` (the null stream: the calling thread's current device). */
int pt_assemble_rows(const float* d_gathered, float* d_full, int height, int width, int world, void* stream);

/* PPM P3 text of a host AoS framebuffer exactly as main.cu:240-253 prints it.
 * Returns the byte count, or -1 if cap is too small. */
long pt_ppm_p3(const float* fb, int nx, int ny, char* out, long cap);

/* Binary writers beside the reference's ASCII PPM: P6 (8-bit, int(255.99*c)
 * clamped to [0,255], rows top to bottom) and PFM (float32, rows bottom to top).
 * Return the byte count, or -1 if cap is too small. */
long pt_ppm_p6(const float* fb, int nx, int ny, unsigned char* out, long cap);
long pt_pfm(const float* fb, int nx, int ny, unsigned char* out, long cap);

const char* pt_last_error(void);
const char* pt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PT_HIP_H */
