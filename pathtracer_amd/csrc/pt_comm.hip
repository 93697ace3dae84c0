// Multi-GPU frame assembly behind the C ABI: an RCCL communicator, the all-gather of the ranks' row tiles over
// xGMI and the kernel that puts the gathered rows back in image order.  The reference renders on one GPU
// (main.cu:236) and has no counterpart.
//
// RCCL is opened with dlopen on first use, so single-GPU users of libpt_hip.so neither link nor load it; a
// process that already holds an RCCL (one that imported torch, say) gets that same copy.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "pt_internal.h"

#define PT_FAIL(msg) pt_set_error(-1, msg, __FILE__, __LINE__)

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

void open_rccl() {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);   // a copy the process already holds
    for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    g_rccl.handle = h;
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_rccl.CommInitAll = (decltype(g_rccl.CommInitAll))dlsym(h, "ncclCommInitAll");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(h, "ncclAllGather");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommInitAll && g_rccl.CommDestroy &&
                g_rccl.AllGather && g_rccl.GetErrorString;
}

const Rccl* rccl() {
    std::call_once(g_rccl_once, open_rccl);
    return g_rccl.ok ? &g_rccl : nullptr;
}

int rccl_error(ncclResult_t r, const char* expr, const char* file, int line) {
    static thread_local char text[256];
    std::snprintf(text, sizeof(text), "%s: %s", expr, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
    return pt_set_error((int)r, text, file, line);
}
#define PT_RCCL(expr)                                                          \
    do {                                                                       \
        ncclResult_t _r = (expr);                                              \
        if (_r != ncclSuccess) return rccl_error(_r, #expr, __FILE__, __LINE__); \
    } while (0)

// Image row j comes from rank j % world, where it is row j / world of that rank's tile.  One thread moves one
// 16-byte (VEC = 4) or 4-byte (VEC = 1) piece; consecutive threads move consecutive pieces of a row.
template <int VEC>
__global__ __launch_bounds__(256) void assemble_rows_kernel(const float* __restrict__ gathered, float* __restrict__ full,
                                                            int height, int row_pieces, int world, int tile_rows) {
    const size_t total = (size_t)height * (size_t)row_pieces;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i / (size_t)row_pieces);
        const int c = (int)(i - (size_t)j * (size_t)row_pieces);
        const size_t src = ((size_t)(j % world) * (size_t)tile_rows + (size_t)(j / world)) * (size_t)row_pieces + (size_t)c;
        if (VEC == 4) reinterpret_cast<float4*>(full)[i] = reinterpret_cast<const float4*>(gathered)[src];
        else full[i] = gathered[src];
    }
}

// Staging area of pt_gather_framebuffer_noalloc: the gathered tiles of one frame, world x pt_tile_rows x width x 3
// floats, at most PT_GATHER_NOALLOC_MAX_FLOATS (32 MiB: a 1920x1080 frame over any world size up to 8).  Every GPU
// the library's code object is loaded on holds its own copy until the process ends.  In host code the array's name is
// only a handle for hipGetSymbolAddress; the address that call returns, kept per communicator, is what the all-gather
// and the reordering kernel are given.
static __device__ __align__(256) float scratch[PT_GATHER_NOALLOC_MAX_FLOATS];

// Device address of `scratch` on the CURRENT device (each GPU has its own).  Called from the communicator set-up
// calls, never from the gather itself: the first use of a code object on a device may load it, which allocates.
int scratch_address(float** out) {
    void* p = nullptr;
    PT_CHECK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) return PT_FAIL("hipGetSymbolAddress(scratch) returned a null address");
    *out = static_cast<float*>(p);
    return 0;
}

}  // namespace

struct pt_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    float* scratch = nullptr;   // this device's copy of the static staging array
};

extern "C" {

int pt_tile_rows(int height, int world) { return (height <= 0 || world <= 0) ? 0 : (height + world - 1) / world; }

int pt_comm_unique_id(char* id) {
    if (!id) return PT_FAIL("pt_comm_unique_id: null argument");
    const Rccl* R = rccl();
    if (!R) return PT_FAIL("pt_comm_unique_id: librccl.so.1 could not be loaded");
    ncclUniqueId u;
    PT_RCCL(R->GetUniqueId(&u));
    static_assert(sizeof(u.internal) == PT_COMM_ID_BYTES, "PT_COMM_ID_BYTES must be RCCL's unique-id size");
    std::memcpy(id, u.internal, PT_COMM_ID_BYTES);
    return 0;
}

int pt_comm_init_rank(const char* id, int rank, int world, int device, pt_comm** out) {
    if (!id || !out || world <= 0 || rank < 0 || rank >= world) return PT_FAIL("pt_comm_init_rank: bad arguments");
    const Rccl* R = rccl();
    if (!R) return PT_FAIL("pt_comm_init_rank: librccl.so.1 could not be loaded");
    PT_CHECK(hipSetDevice(device));
    ncclUniqueId u;
    std::memcpy(u.internal, id, PT_COMM_ID_BYTES);
    pt_comm* c = new (std::nothrow) pt_comm;
    if (!c) return PT_FAIL("pt_comm_init_rank: out of host memory");
    c->rank = rank; c->world = world; c->device = device;
    if (const int rc = scratch_address(&c->scratch)) { delete c; return rc; }
    const ncclResult_t r = R->CommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) { delete c; return rccl_error(r, "ncclCommInitRank", __FILE__, __LINE__); }
    *out = c;
    return 0;
}

int pt_comm_init_all(int world, const int* devices, pt_comm** out) {
    if (!devices || !out || world <= 0) return PT_FAIL("pt_comm_init_all: bad arguments");
    const Rccl* R = rccl();
    if (!R) return PT_FAIL("pt_comm_init_all: librccl.so.1 could not be loaded");
    int before = 0;
    PT_CHECK(hipGetDevice(&before));                  // (before anything exists that an error return would have to undo)
    std::vector<ncclComm_t> comms((size_t)world, nullptr);
    PT_RCCL(R->CommInitAll(comms.data(), world, devices));
    for (int r = 0; r < world; r++) {
        pt_comm* c = new (std::nothrow) pt_comm;
        if (!c) {
            for (int k = 0; k < r; k++) { delete out[k]; out[k] = nullptr; }
            for (int k = 0; k < world; k++) (void)R->CommDestroy(comms[(size_t)k]);
            return PT_FAIL("pt_comm_init_all: out of host memory");
        }
        c->comm = comms[(size_t)r]; c->rank = r; c->world = world; c->device = devices[r];
        out[r] = c;
    }
    // each device's own address of the static staging array, resolved with that device current
    for (int r = 0; r < world; r++) {
        hipError_t e = hipSetDevice(devices[r]);
        int rc = e == hipSuccess ? scratch_address(&out[r]->scratch) : pt_set_error((int)e, "hipSetDevice(devices[r])", __FILE__, __LINE__);
        if (rc) {
            for (int k = 0; k < world; k++) { (void)R->CommDestroy(comms[(size_t)k]); delete out[k]; out[k] = nullptr; }
            (void)hipSetDevice(before);
            return rc;
        }
    }
    // on success the caller owns out[0..world); failing to go back to the device that was current is not worth
    // taking them away again
    (void)hipSetDevice(before);
    return 0;
}

int pt_comm_free(pt_comm* c) {
    if (!c) return 0;
    int rc = 0;
    if (c->comm && g_rccl.ok) {
        (void)hipSetDevice(c->device);
        const ncclResult_t r = g_rccl.CommDestroy(c->comm);
        if (r != ncclSuccess) rc = rccl_error(r, "ncclCommDestroy", __FILE__, __LINE__);
    }
    delete c;
    return rc;
}

int pt_comm_rank(const pt_comm* c) { return c ? c->rank : -1; }
int pt_comm_world(const pt_comm* c) { return c ? c->world : -1; }

int pt_assemble_rows(const float* d_gathered, float* d_full, int height, int width, int world, void* stream) {
    if (!d_gathered || !d_full || height <= 0 || width <= 0 || world <= 0) return PT_FAIL("pt_assemble_rows: bad arguments");
    const int tile_rows = pt_tile_rows(height, world);
    const size_t row_floats = (size_t)width * 3u;
    const bool vec = (row_floats % 4u == 0u) && (((uintptr_t)d_gathered | (uintptr_t)d_full) % 16u == 0u);
    const size_t row_pieces = vec ? row_floats / 4u : row_floats;
    if (row_pieces > 0x7fffffffull) return PT_FAIL("pt_assemble_rows: row too long");
    const size_t total = (size_t)height * row_pieces;
    const unsigned blocks = (unsigned)std::min<size_t>((total + 255u) / 256u, 256u * 32u);
    if (vec) hipLaunchKernelGGL(assemble_rows_kernel<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_gathered, d_full, height, (int)row_pieces, world, tile_rows);
    else hipLaunchKernelGGL(assemble_rows_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_gathered, d_full, height, (int)row_pieces, world, tile_rows);
    PT_CHECK(hipGetLastError());
    return 0;
}

int pt_gather_framebuffer(pt_comm* c, const float* d_tile, float* d_gathered, float* d_full, int height, int width, void* stream) {
    if (!c || !d_tile || !d_gathered || !d_full || height <= 0 || width <= 0) return PT_FAIL("pt_gather_framebuffer: bad arguments");
    if (!g_rccl.ok) return PT_FAIL("pt_gather_framebuffer: librccl.so.1 could not be loaded");
    PT_CHECK(hipSetDevice(c->device));
    const size_t tile_floats = (size_t)pt_tile_rows(height, c->world) * (size_t)width * 3u;
    PT_RCCL(g_rccl.AllGather(d_tile, d_gathered, tile_floats, ncclFloat, c->comm, (hipStream_t)stream));
    return pt_assemble_rows(d_gathered, d_full, height, width, c->world, stream);
}

int pt_gather_framebuffer_noalloc(pt_comm* c, const float* d_tile, float* d_full, int height, int width, void* stream) {
    if (!c || !d_tile || !d_full || height <= 0 || width <= 0) return PT_FAIL("pt_gather_framebuffer_noalloc: bad arguments");
    if (!g_rccl.ok) return PT_FAIL("pt_gather_framebuffer_noalloc: librccl.so.1 could not be loaded");
    if (!c->scratch) return PT_FAIL("pt_gather_framebuffer_noalloc: the communicator holds no staging address");
    const size_t tile_floats = (size_t)pt_tile_rows(height, c->world) * (size_t)width * 3u;
    if (tile_floats * (size_t)c->world > (size_t)PT_GATHER_NOALLOC_MAX_FLOATS) {
        static thread_local char text[200];
        std::snprintf(text, sizeof(text), "pt_gather_framebuffer_noalloc: %zu gathered floats exceed the static staging area of %zu (PT_GATHER_NOALLOC_MAX_FLOATS); use pt_gather_framebuffer",
                      tile_floats * (size_t)c->world, (size_t)PT_GATHER_NOALLOC_MAX_FLOATS);
        return PT_FAIL(text);
    }
    PT_CHECK(hipSetDevice(c->device));
    // Every float of the staging area that the reordering reads is written by the all-gather first, so nothing
    // has to be cleared between calls.
    PT_RCCL(g_rccl.AllGather(d_tile, c->scratch, tile_floats, ncclFloat, c->comm, (hipStream_t)stream));
    return pt_assemble_rows(c->scratch, d_full, height, width, c->world, stream);
}

}  // extern "C"
