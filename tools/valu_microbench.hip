// Instruction-throughput microbenchmark for gfx950 (development aid, not part of
// the library).  For each VALU instruction kind it runs a long unrolled stream
// of that instruction on W waves per SIMD (W = 1, 2, 4, 8) and reports shader
// cycles per wave-instruction per SIMD, from s_memtime stamps inside the kernel.
//
//   hipcc --offload-arch=gfx950 -O2 tools/valu_microbench.hip -o /tmp/valu_mb && /tmp/valu_mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

constexpr int ITERS = 10000;        // 10000 * 64 = 640000 instructions per wave (milliseconds per launch: clocks settle)

#define KERNEL(name, body, pre)                                                                   \
    __global__ __launch_bounds__(256) void k_##name(unsigned long long* out, float seed) {        \
        float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;               \
        float a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;                         \
        float b = seed * 0.5f + 1.0f, c = seed + 0.25f;                                           \
        unsigned long long m = 0; (void)m;                                                        \
        pre;                                                                                      \
        unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                 \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                        \
        for (int it = 0; it < ITERS; it++) {                                                      \
            REP8(body)                                                                            \
        }                                                                                         \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                     \
        unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                 \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                        \
        if (blockIdx.x == 0 && threadIdx.x == 0) out[1 + 8 * 256 * 4] = r1 - r0;                  \
        float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                          \
        if (s == 12345.678f) out[0] = 1;                                                          \
        if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;      \
    }

// eight independent accumulators per group of 8 instructions
#define OP3(ins) \
    asm volatile(ins " %0, %0, %8, %9\n\t" ins " %1, %1, %8, %9\n\t" ins " %2, %2, %8, %9\n\t" ins " %3, %3, %8, %9\n\t" \
                 ins " %4, %4, %8, %9\n\t" ins " %5, %5, %8, %9\n\t" ins " %6, %6, %8, %9\n\t" ins " %7, %7, %8, %9"     \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
#define OP2(ins) \
    asm volatile(ins " %0, %0, %8\n\t" ins " %1, %1, %8\n\t" ins " %2, %2, %8\n\t" ins " %3, %3, %8\n\t" \
                 ins " %4, %4, %8\n\t" ins " %5, %5, %8\n\t" ins " %6, %6, %8\n\t" ins " %7, %7, %8"     \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
#define OP1(ins) \
    asm volatile(ins " %0, %0\n\t" ins " %1, %1\n\t" ins " %2, %2\n\t" ins " %3, %3\n\t" \
                 ins " %4, %4\n\t" ins " %5, %5\n\t" ins " %6, %6\n\t" ins " %7, %7"     \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
// compare into vcc (VOPC) and into an SGPR pair (VOP3)
#define OPCMP_VCC \
    asm volatile("v_cmp_lt_f32 vcc, %0, %8\n\tv_cmp_lt_f32 vcc, %1, %8\n\tv_cmp_lt_f32 vcc, %2, %8\n\tv_cmp_lt_f32 vcc, %3, %8\n\t" \
                 "v_cmp_lt_f32 vcc, %4, %8\n\tv_cmp_lt_f32 vcc, %5, %8\n\tv_cmp_lt_f32 vcc, %6, %8\n\tv_cmp_lt_f32 vcc, %7, %8"     \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
#define OPCMP_SGPR \
    asm volatile("v_cmp_lt_f32_e64 %8, %0, %9\n\tv_cmp_lt_f32_e64 %8, %1, %9\n\tv_cmp_lt_f32_e64 %8, %2, %9\n\tv_cmp_lt_f32_e64 %8, %3, %9\n\t" \
                 "v_cmp_lt_f32_e64 %8, %4, %9\n\tv_cmp_lt_f32_e64 %8, %5, %9\n\tv_cmp_lt_f32_e64 %8, %6, %9\n\tv_cmp_lt_f32_e64 %8, %7, %9"     \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=s"(m) : "v"(b));
#define OPCNDMASK \
    asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n\tv_cndmask_b32 %1, %1, %8, vcc\n\tv_cndmask_b32 %2, %2, %8, vcc\n\tv_cndmask_b32 %3, %3, %8, vcc\n\t" \
                 "v_cndmask_b32 %4, %4, %8, vcc\n\tv_cndmask_b32 %5, %5, %8, vcc\n\tv_cndmask_b32 %6, %6, %8, vcc\n\tv_cndmask_b32 %7, %7, %8, vcc"     \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
#define OPSALU \
    asm volatile("s_and_b64 %0, %0, exec\n\ts_or_b64 %0, %0, vcc\n\ts_and_b64 %0, %0, exec\n\ts_or_b64 %0, %0, vcc\n\t" \
                 "s_and_b64 %0, %0, exec\n\ts_or_b64 %0, %0, vcc\n\ts_and_b64 %0, %0, exec\n\ts_or_b64 %0, %0, vcc" : "+s"(m) :: "scc");
// mixed: 6 VALU fma + 2 SALU per group (do SALU instructions steal VALU issue slots?)
#define OPMIX \
    asm volatile("v_fma_f32 %0, %0, %8, %9\n\ts_and_b64 %10, %10, exec\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\t" \
                 "v_fma_f32 %3, %3, %8, %9\n\ts_or_b64 %10, %10, vcc\n\tv_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9"     \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(m) : "scc");
// packed f32: operands are 64-bit register pairs
#define OPPK(ins) \
    asm volatile(ins " %0, %0, %4, %5\n\t" ins " %1, %1, %4, %5\n\t" ins " %2, %2, %4, %5\n\t" ins " %3, %3, %4, %5\n\t" \
                 ins " %0, %0, %4, %5\n\t" ins " %1, %1, %4, %5\n\t" ins " %2, %2, %4, %5\n\t" ins " %3, %3, %4, %5"     \
                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));


// VALU with one SGPR source operand (the render kernels read primitive records from SGPRs)
#define OP3S(ins) \
    asm volatile(ins " %0, %0, %8, %9\n\t" ins " %1, %1, %8, %9\n\t" ins " %2, %2, %8, %9\n\t" ins " %3, %3, %8, %9\n\t" \
                 ins " %4, %4, %8, %9\n\t" ins " %5, %5, %8, %9\n\t" ins " %6, %6, %8, %9\n\t" ins " %7, %7, %8, %9"     \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(b), "v"(c));
#define OP2S(ins) \
    asm volatile(ins " %0, %8, %0\n\t" ins " %1, %8, %1\n\t" ins " %2, %8, %2\n\t" ins " %3, %8, %3\n\t" \
                 ins " %4, %8, %4\n\t" ins " %5, %8, %5\n\t" ins " %6, %8, %6\n\t" ins " %7, %8, %7"     \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(b));
#define OPCNDMASK_S \
    asm volatile("v_cndmask_b32_e64 %0, %0, %8, %9\n\tv_cndmask_b32_e64 %1, %1, %8, %9\n\tv_cndmask_b32_e64 %2, %2, %8, %9\n\tv_cndmask_b32_e64 %3, %3, %8, %9\n\t" \
                 "v_cndmask_b32_e64 %4, %4, %8, %9\n\tv_cndmask_b32_e64 %5, %5, %8, %9\n\tv_cndmask_b32_e64 %6, %6, %8, %9\n\tv_cndmask_b32_e64 %7, %7, %8, %9"     \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(m));
// cndmask whose two value operands differ from the destination (no read-modify-write of the accumulator)
#define OPCNDMASK_2 \
    asm volatile("v_cndmask_b32 %0, %8, %9, vcc\n\tv_cndmask_b32 %1, %8, %9, vcc\n\tv_cndmask_b32 %2, %8, %9, vcc\n\tv_cndmask_b32 %3, %8, %9, vcc\n\t" \
                 "v_cndmask_b32 %4, %8, %9, vcc\n\tv_cndmask_b32 %5, %8, %9, vcc\n\tv_cndmask_b32 %6, %8, %9, vcc\n\tv_cndmask_b32 %7, %8, %9, vcc"     \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
// the select pattern of the primitive loop: compare then two selects on its mask
#define OPCMPSEL \
    asm volatile("v_cmp_lt_f32 vcc, %0, %8\n\tv_cndmask_b32 %1, %1, %8, vcc\n\tv_cndmask_b32 %2, %2, %9, vcc\n\tv_cmp_lt_f32 vcc, %3, %8\n\t" \
                 "v_cndmask_b32 %4, %4, %8, vcc\n\tv_cndmask_b32 %5, %5, %9, vcc\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"     \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");

// 64-bit compare into an SGPR pair: the (distance, index) rule of the primitive tests
#define OPCMP64 \
    asm volatile("v_cmp_lt_u64_e64 %4, %0, %5\n\tv_cmp_lt_u64_e64 %4, %1, %5\n\tv_cmp_lt_u64_e64 %4, %2, %5\n\tv_cmp_lt_u64_e64 %4, %3, %5\n\t" \
                 "v_cmp_lt_u64_e64 %4, %0, %5\n\tv_cmp_lt_u64_e64 %4, %1, %5\n\tv_cmp_lt_u64_e64 %4, %2, %5\n\tv_cmp_lt_u64_e64 %4, %3, %5"     \
                 : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "=s"(m) : "v"(qb));
#define Q64PRE unsigned long long q0 = threadIdx.x, q1 = q0 + 1, q2 = q0 + 2, q3 = q0 + 3, qb = (unsigned long long)(seed * 3.0f);
#define OPREADLANE \
    asm volatile("v_readlane_b32 %8, %0, 3\n\tv_readlane_b32 %8, %1, 5\n\tv_readlane_b32 %8, %2, 7\n\tv_readlane_b32 %8, %3, 9\n\t" \
                 "v_readlane_b32 %8, %4, 11\n\tv_readlane_b32 %8, %5, 13\n\tv_readlane_b32 %8, %6, 15\n\tv_readlane_b32 %8, %7, 17"     \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=s"(mi));
#define OPREADFIRST \
    asm volatile("v_readfirstlane_b32 %8, %0\n\tv_readfirstlane_b32 %8, %1\n\tv_readfirstlane_b32 %8, %2\n\tv_readfirstlane_b32 %8, %3\n\t" \
                 "v_readfirstlane_b32 %8, %4\n\tv_readfirstlane_b32 %8, %5\n\tv_readfirstlane_b32 %8, %6\n\tv_readfirstlane_b32 %8, %7"     \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=s"(mi));

// the select pattern as the compiler emits it: the mask combined by the scalar unit into vcc, or into an SGPR pair
#define OPSELVCC \
    asm volatile("s_and_b64 vcc, %8, exec\n\tv_cndmask_b32 %0, %0, %9, vcc\n\tv_cndmask_b32 %1, %1, %9, vcc\n\tv_fma_f32 %2, %2, %9, %9\n\tv_fma_f32 %3, %3, %9, %9\n\t" \
                 "s_and_b64 vcc, %8, exec\n\tv_cndmask_b32 %4, %4, %9, vcc\n\tv_cndmask_b32 %5, %5, %9, vcc\n\tv_fma_f32 %6, %6, %9, %9\n\tv_fma_f32 %7, %7, %9, %9"     \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(m), "v"(b) : "vcc", "scc");
#define OPSELSGPR \
    asm volatile("s_and_b64 %10, %8, exec\n\tv_cndmask_b32_e64 %0, %0, %9, %10\n\tv_cndmask_b32_e64 %1, %1, %9, %10\n\tv_fma_f32 %2, %2, %9, %9\n\tv_fma_f32 %3, %3, %9, %9\n\t" \
                 "s_and_b64 %10, %8, exec\n\tv_cndmask_b32_e64 %4, %4, %9, %10\n\tv_cndmask_b32_e64 %5, %5, %9, %10\n\tv_fma_f32 %6, %6, %9, %9\n\tv_fma_f32 %7, %7, %9, %9"     \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(m), "v"(b), "s"(m2) : "scc");

typedef float float2v __attribute__((ext_vector_type(2)));
#define PKPRE float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};

KERNEL(fma, OP3("v_fma_f32"), )
KERNEL(mul, OP2("v_mul_f32"), )
KERNEL(add, OP2("v_add_f32"), )
KERNEL(max, OP2("v_max_f32"), )
KERNEL(max3, OP3("v_max3_f32"), )
KERNEL(med3, OP3("v_med3_f32"), )
KERNEL(cmp_vcc, OPCMP_VCC, )
KERNEL(cmp_sgpr, OPCMP_SGPR, )
KERNEL(cndmask, OPCNDMASK, )
KERNEL(mul_lo_u32, OP2("v_mul_lo_u32"), )
KERNEL(mul_hi_u32, OP2("v_mul_hi_u32"), )
KERNEL(mad_u32_u24, OP3("v_mad_u32_u24"), )
KERNEL(mul_u32_u24, OP2("v_mul_u32_u24"), )
KERNEL(xor, OP2("v_xor_b32"), )
KERNEL(lshrrev, OP2("v_lshrrev_b32"), )
KERNEL(add_u32, OP2("v_add_u32"), )
KERNEL(xad_u32, OP3("v_xad_u32"), )
KERNEL(lshl_add_u32, OP3("v_lshl_add_u32"), )
KERNEL(alignbit, OP3("v_alignbit_b32"), )
KERNEL(rcp, OP1("v_rcp_f32"), )
KERNEL(rsq, OP1("v_rsq_f32"), )
KERNEL(sqrt, OP1("v_sqrt_f32"), )
KERNEL(sin, OP1("v_sin_f32"), )
KERNEL(exp, OP1("v_exp_f32"), )
KERNEL(cvt_f32_u32, OP1("v_cvt_f32_u32"), )
KERNEL(fma_sgpr, OP3S("v_fma_f32"), )
KERNEL(mul_sgpr, OP2S("v_mul_f32"), )
KERNEL(fmac, OP2("v_fmac_f32"), )
KERNEL(min, OP2("v_min_f32"), )
KERNEL(sub, OP2("v_sub_f32"), )
KERNEL(mov, OP1("v_mov_b32"), )
KERNEL(and, OP2("v_and_b32"), )
KERNEL(bfe_u32, OP3("v_bfe_u32"), )
KERNEL(cndmask_sgpr, OPCNDMASK_S, )
KERNEL(cndmask_2src, OPCNDMASK_2, )
KERNEL(cmp_sel_mix, OPCMPSEL, )
KERNEL(cmp_lt_u64, OPCMP64, Q64PRE)
KERNEL(readlane, OPREADLANE, int mi = 0; (void)mi)
KERNEL(readfirstlane, OPREADFIRST, int mi = 0; (void)mi)
KERNEL(cvt_u32_f32, OP1("v_cvt_u32_f32"), )
KERNEL(floor, OP1("v_floor_f32"), )
KERNEL(mbcnt, OP2("v_mbcnt_lo_u32_b32"), )
KERNEL(sel_salu_vcc, OPSELVCC, m = 0x5555555555555555ull)
KERNEL(sel_salu_sgpr, OPSELSGPR, m = 0x5555555555555555ull; unsigned long long m2 = 1; asm volatile("" : "+s"(m2)))
KERNEL(salu, OPSALU, )
KERNEL(mix_fma_salu, OPMIX, )

__global__ __launch_bounds__(256) void k_pk_fma(unsigned long long* out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    float b = seed * 0.5f + 1.0f, c = seed + 0.25f;
    PKPRE
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < ITERS; it++) { REP8(OPPK("v_pk_fma_f32")) }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (blockIdx.x == 0 && threadIdx.x == 0) out[1 + 8 * 256 * 4] = r1 - r0;
    float s = p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if (s == 12345.678f) out[0] = 1;
    if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

typedef void (*kern_t)(unsigned long long*, float);
struct Entry { const char* name; kern_t k; };

int main() {
    std::vector<Entry> ks = {
#define E(n) {#n, k_##n}
        E(fma), E(mul), E(add), E(sub), E(fmac), E(mov), E(xor), E(and), E(lshrrev), E(add_u32), E(fma_sgpr), E(mul_sgpr),
        E(max), E(min), E(max3), E(med3), E(cmp_vcc), E(cmp_sgpr), E(cndmask_sgpr), E(cndmask), E(cndmask_2src), E(cmp_sel_mix), E(sel_salu_vcc), E(sel_salu_sgpr),
        E(mul_lo_u32), E(mul_hi_u32), E(mad_u32_u24), E(mul_u32_u24), E(xad_u32), E(lshl_add_u32), E(alignbit), E(bfe_u32),
        E(cvt_f32_u32), E(cvt_u32_f32), E(floor), E(cmp_lt_u64), E(readlane), E(readfirstlane), E(mbcnt), E(pk_fma), E(rcp), E(rsq), E(sqrt), E(sin), E(exp), E(salu), E(mix_fma_salu)};
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs; %d instructions per wave per launch\n", prop.gcnArchName, cus, ITERS * 64);
    unsigned long long* d;
    const int max_blocks = cus * 8;
    hipMalloc(&d, sizeof(unsigned long long) * (2 + max_blocks * 4));
    std::vector<unsigned long long> h(2 + max_blocks * 4);
    printf("%-14s | wave-instructions per ns per SIMD from wall time (W x 256-thread blocks per CU = W waves per SIMD) | one wave alone: shader cycles per instruction\n", "instruction");
    printf("%-14s %8s %8s %8s %8s | %8s\n", "", "W=1", "W=2", "W=4", "W=8", "cycles");
    const double n_instr = (double)ITERS * 64.0;
    for (auto& e : ks) {
        printf("%-14s", e.name);
        double cyc1 = 0;
        for (int w : {1, 2, 4, 8}) {
            const int blocks = cus * w;
            hipMemset(d, 0, sizeof(unsigned long long) * (2 + max_blocks * 4));
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d, 1.0f);   // warm-up
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            hipEventDestroy(e0); hipEventDestroy(e1);
            if (w == 1) {
                hipMemcpy(h.data(), d, sizeof(unsigned long long) * (2 + max_blocks * 4), hipMemcpyDeviceToHost);
                std::vector<unsigned long long> v(h.begin() + 1, h.begin() + 1 + blocks * 4);
                std::sort(v.begin(), v.end());
                cyc1 = (double)v[v.size() / 2] / n_instr;
            }
            printf(" %8.3f", n_instr * w / (ms * 1e6));
        }
        printf(" | %8.2f\n", cyc1);
    }
    hipFree(d);
    return 0;
}
