// valu_rate.hip -- gfx950 micro-benchmark in THROUGHPUT form: cycles per wave64 VALU instruction per SIMD for the op classes an int8
// min-sum decoder is built from (integer vs fp32 vs packed forms). 4 independent accumulator chains per lane; 1, 2, 3, 4 and 8 waves
// per SIMD (one block per CU; 3 = the occupancy of ldpc_decode2_kernel). Two clocks: wall time x the nominal 2.4 GHz, and the shader
// clock itself (s_memtime, first start to last end over all waves of CU 0: independent of what the clock actually was).
// Build: hipcc --offload-arch=gfx950 -O2 valu_rate.hip -o valu_rate.   Output: one line per op, and `json` lines for tools.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP 64
#define ITER 256

#define KERNEL(NAME, DECL, BODY)                                                        \
    __global__ void k_##NAME(int *out, int seed, long long *clk)                          \
    {                                                                                     \
        DECL;                                                                             \
        const long long t0 = __builtin_amdgcn_s_memtime();                                \
        for (int it = 0; it < ITER; ++it) {                                               \
            _Pragma("unroll") for (int r = 0; r < REP / 4; ++r) { BODY; }                 \
        }                                                                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = FIN;                                 \
        const long long t1 = __builtin_amdgcn_s_memtime();                                \
        if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { clk[2 * (threadIdx.x >> 6)] = t0; clk[2 * (threadIdx.x >> 6) + 1] = t1; } \
    }

#define IDECL int a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, b = seed + 11, c = seed + 5
#define FIN (a0 ^ a1 ^ a2 ^ a3)
#define ASM4(OP) asm volatile(OP : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c))

KERNEL(add_u32, IDECL, ASM4("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4"))
KERNEL(min_i32, IDECL, ASM4("v_min_i32 %0, %0, %4\n v_min_i32 %1, %1, %4\n v_min_i32 %2, %2, %4\n v_min_i32 %3, %3, %4"))
KERNEL(med3_i32, IDECL, ASM4("v_med3_i32 %0, %0, %4, %5\n v_med3_i32 %1, %1, %4, %5\n v_med3_i32 %2, %2, %4, %5\n v_med3_i32 %3, %3, %4, %5"))
KERNEL(xor_b32, IDECL, ASM4("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4"))
KERNEL(cndmask, IDECL, ASM4("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc"))
KERNEL(cmp_cnd, IDECL, ASM4("v_cmp_lt_i32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %5, vcc\n v_cmp_lt_i32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %5, vcc"))
KERNEL(sad_u32, IDECL, ASM4("v_sad_u32 %0, %0, %4, %5\n v_sad_u32 %1, %1, %4, %5\n v_sad_u32 %2, %2, %4, %5\n v_sad_u32 %3, %3, %4, %5"))
KERNEL(add3_u32, IDECL, ASM4("v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %4, %5\n v_add3_u32 %2, %2, %4, %5\n v_add3_u32 %3, %3, %4, %5"))
KERNEL(bfe_i32, IDECL, ASM4("v_bfe_i32 %0, %0, %4, %5\n v_bfe_i32 %1, %1, %4, %5\n v_bfe_i32 %2, %2, %4, %5\n v_bfe_i32 %3, %3, %4, %5"))
KERNEL(perm_b32, IDECL, ASM4("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5"))
KERNEL(lshl_or, IDECL, ASM4("v_lshl_or_b32 %0, %0, %4, %5\n v_lshl_or_b32 %1, %1, %4, %5\n v_lshl_or_b32 %2, %2, %4, %5\n v_lshl_or_b32 %3, %3, %4, %5"))
KERNEL(pk_add_i16, IDECL, ASM4("v_pk_add_i16 %0, %0, %4\n v_pk_add_i16 %1, %1, %4\n v_pk_add_i16 %2, %2, %4\n v_pk_add_i16 %3, %3, %4"))
KERNEL(pk_min_i16, IDECL, ASM4("v_pk_min_i16 %0, %0, %4\n v_pk_min_i16 %1, %1, %4\n v_pk_min_i16 %2, %2, %4\n v_pk_min_i16 %3, %3, %4"))
KERNEL(pk_sub_i16_clamp, IDECL, ASM4("v_pk_sub_i16 %0, %0, %4 clamp\n v_pk_sub_i16 %1, %1, %4 clamp\n v_pk_sub_i16 %2, %2, %4 clamp\n v_pk_sub_i16 %3, %3, %4 clamp"))
KERNEL(add_f32, IDECL, ASM4("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4"))
KERNEL(min_f32, IDECL, ASM4("v_min_f32 %0, %0, %4\n v_min_f32 %1, %1, %4\n v_min_f32 %2, %2, %4\n v_min_f32 %3, %3, %4"))
KERNEL(min_f32_abs, IDECL, ASM4("v_min_f32 %0, |%0|, %4\n v_min_f32 %1, |%1|, %4\n v_min_f32 %2, |%2|, %4\n v_min_f32 %3, |%3|, %4"))
KERNEL(med3_f32, IDECL, ASM4("v_med3_f32 %0, %0, %4, %5\n v_med3_f32 %1, %1, %4, %5\n v_med3_f32 %2, %2, %4, %5\n v_med3_f32 %3, %3, %4, %5"))
KERNEL(fma_f32, IDECL, ASM4("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"))
KERNEL(mul_f32, IDECL, ASM4("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4"))
KERNEL(cmp_f32_cnd, IDECL, ASM4("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %5, vcc\n v_cmp_lt_f32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %5, vcc"))
KERNEL(cvt_f32_i32, IDECL, ASM4("v_cvt_f32_i32 %0, %0\n v_cvt_f32_i32 %1, %1\n v_cvt_f32_i32 %2, %2\n v_cvt_f32_i32 %3, %3"))
KERNEL(cvt_f32_ubyte0, IDECL, ASM4("v_cvt_f32_ubyte0 %0, %0\n v_cvt_f32_ubyte0 %1, %1\n v_cvt_f32_ubyte0 %2, %2\n v_cvt_f32_ubyte0 %3, %3"))

struct A2 { unsigned long long x, y; };
__global__ void k_pk_add_f32(int *out, int seed, long long *clk)
{
    const long long t0 = __builtin_amdgcn_s_memtime();
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 * 3.f, a2 = a0 * 5.f, a3 = a0 * 7.f, b = {(float)seed, 2.f};
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 4; ++r)
            asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (int)(a0.x + a1.x + a2.x + a3.y);
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { clk[2 * (threadIdx.x >> 6)] = t0; clk[2 * (threadIdx.x >> 6) + 1] = t1; }
}

#define PK_KERNEL(NAME, OPS)                                                                                          \
    __global__ void k_##NAME(int *out, int seed, long long *clk)                                                       \
    {                                                                                                                  \
        const long long t0 = __builtin_amdgcn_s_memtime();                                                             \
        typedef float f2 __attribute__((ext_vector_type(2)));                                                          \
        f2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 * 3.f, a2 = a0 * 5.f, a3 = a0 * 7.f, b = {(float)seed, 2.f};          \
        float s0 = (float)threadIdx.x, s1 = s0 * 3.f, s2 = s0 * 5.f, s3 = s0 * 7.f, sb = (float)seed;                    \
        for (int it = 0; it < ITER; ++it) {                                                                            \
            _Pragma("unroll") for (int r = 0; r < REP / 4; ++r)                                                        \
                asm volatile(OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(b), "v"(sb)); \
        }                                                                                                              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (int)(a0.x + a1.x + a2.x + a3.y + s0 + s1 + s2 + s3);              \
        const long long t1 = __builtin_amdgcn_s_memtime();                                                             \
        if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { clk[2 * (threadIdx.x >> 6)] = t0; clk[2 * (threadIdx.x >> 6) + 1] = t1; } \
    }
// the decimator's own mix, and packed beside plain fp32
PK_KERNEL(mix_pkmul_pkadd, "v_pk_mul_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8")
PK_KERNEL(mix_pkadd_add, "v_pk_add_f32 %0, %0, %8\n v_add_f32 %4, %4, %9\n v_pk_add_f32 %2, %2, %8\n v_add_f32 %6, %6, %9")
PK_KERNEL(mix_pkmul_mul_add, "v_pk_mul_f32 %0, %0, %8\n v_mul_f32 %4, %4, %9\n v_add_f32 %5, %5, %9\n v_pk_add_f32 %1, %1, %8")
PK_KERNEL(mix_mul_add, "v_mul_f32 %4, %4, %9\n v_add_f32 %5, %5, %9\n v_mul_f32 %6, %6, %9\n v_add_f32 %7, %7, %9")

// further classes the two-frame kernel uses
KERNEL(and_b32, IDECL, ASM4("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4"))
KERNEL(lshlrev_b32, IDECL, ASM4("v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3"))
KERNEL(mov_dpp, IDECL, ASM4("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"))
KERNEL(pk_max_i16, IDECL, ASM4("v_pk_max_i16 %0, %0, %4\n v_pk_max_i16 %1, %1, %4\n v_pk_max_i16 %2, %2, %4\n v_pk_max_i16 %3, %3, %4"))
KERNEL(pk_mad_i16, IDECL, ASM4("v_pk_mad_i16 %0, %0, %4, %5\n v_pk_mad_i16 %1, %1, %4, %5\n v_pk_mad_i16 %2, %2, %4, %5\n v_pk_mad_i16 %3, %3, %4, %5"))
KERNEL(pk_ashr_i16, IDECL, ASM4("v_pk_ashrrev_i16 %0, 15, %0\n v_pk_ashrrev_i16 %1, 15, %1\n v_pk_ashrrev_i16 %2, 15, %2\n v_pk_ashrrev_i16 %3, 15, %3"))
KERNEL(sub_u16_sdwa, IDECL, ASM4("v_sub_u16_sdwa %0, %0, %4 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_1\n v_sub_u16_sdwa %1, %1, %4 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_1\n v_sub_u16_sdwa %2, %2, %4 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_1\n v_sub_u16_sdwa %3, %3, %4 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_1"))
KERNEL(bfi_b32, IDECL, ASM4("v_bfi_b32 %0, %4, %0, %5\n v_bfi_b32 %1, %4, %1, %5\n v_bfi_b32 %2, %4, %2, %5\n v_bfi_b32 %3, %4, %3, %5"))
KERNEL(cmp_only, IDECL, ASM4("v_cmp_lt_i32 vcc, %0, %4\n v_cmp_lt_i32 vcc, %1, %4\n v_cmp_lt_i32 vcc, %2, %4\n v_cmp_lt_i32 vcc, %3, %4"))

// mixes of a fast-class and a slow-class op (two of each per group of four, independent chains): additive or overlapped?
KERNEL(mix_xor_pkmin, IDECL, ASM4("v_xor_b32 %0, %0, %4\n v_pk_min_i16 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_pk_min_i16 %3, %3, %4"))
KERNEL(mix_add_perm, IDECL, ASM4("v_add_u32 %0, %0, %4\n v_perm_b32 %1, %1, %4, %5\n v_add_u32 %2, %2, %4\n v_perm_b32 %3, %3, %4, %5"))
KERNEL(mix_addf_minf, IDECL, ASM4("v_add_f32 %0, %0, %4\n v_min_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_min_f32 %3, %3, %4"))
// the same 1:1 mix in runs of two per class (4.13 cycles per instruction at 3 waves: alternating, 3.85, is the better order)
KERNEL(mix_xor_pkmin_r2, IDECL, ASM4("v_xor_b32 %0, %0, %4\n v_xor_b32 %2, %2, %4\n v_pk_min_i16 %1, %1, %4\n v_pk_min_i16 %3, %3, %4"))
typedef void (*kfn)(int *, int, long long *);
struct Entry { const char *name; kfn fn; };
#define E(N) {#N, k_##N}

int main()
{
    Entry tests[] = {E(add_u32), E(and_b32), E(xor_b32), E(lshlrev_b32), E(min_i32), E(med3_i32), E(cmp_only), E(cmp_cnd), E(sad_u32), E(add3_u32), E(bfe_i32),
                     E(bfi_b32), E(perm_b32), E(lshl_or), E(mov_dpp), E(sub_u16_sdwa), E(pk_add_i16), E(pk_min_i16), E(pk_max_i16), E(pk_sub_i16_clamp),
                     E(pk_mad_i16), E(pk_ashr_i16), E(add_f32), E(min_f32), E(med3_f32), E(fma_f32), E(mul_f32), E(pk_add_f32), E(mix_xor_pkmin), E(mix_add_perm), E(mix_addf_minf), E(mix_pkmul_pkadd), E(mix_pkadd_add), E(mix_pkmul_mul_add), E(mix_mul_add), E(mix_xor_pkmin_r2)};
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    double clk = prop.clockRate * 1e3;   // Hz (nominal max)
    int *out; hipMalloc(&out, (size_t)cus * 8 * 1024 * sizeof(int));
    long long *d_clk; hipMalloc(&d_clk, 64 * sizeof(long long));
    printf("device %s, %d CUs, nominal clock %.0f MHz\n", prop.name, cus, clk / 1e6);
    const int WPS[] = {1, 2, 3, 4};
    printf("%-18s", "op");
    for (int w : WPS) printf("  %dw wall  %dw shdr", w, w);
    printf("   (cycles per wave64 instruction per SIMD; wall = at the nominal clock, shdr = s_memtime span of CU 0)\n");
    for (auto &t : tests) {
        printf("%-18s", t.name);
        double shader[4];
        int k = 0;
        for (int wps : WPS) {
            int threads = 256 * wps;     // wps waves on each of the 4 SIMDs
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            t.fn<<<cus, threads>>>(out, 1, d_clk);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < 5; ++r) t.fn<<<cus, threads>>>(out, r, d_clk);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double insts_per_simd = 5.0 * (double)ITER * REP * wps;       // wave-instructions issued on each SIMD
            double cyc = (ms * 1e-3) * clk / insts_per_simd;
            long long h[64];
            hipMemcpy(h, d_clk, sizeof(long long) * 2 * 4 * wps, hipMemcpyDeviceToHost);
            long long lo = h[0], hi = h[1];
            for (int w = 0; w < 4 * wps; ++w) { lo = h[2 * w] < lo ? h[2 * w] : lo; hi = h[2 * w + 1] > hi ? h[2 * w + 1] : hi; }
            shader[k] = (double)(hi - lo) / ((double)ITER * REP * wps);
            printf("  %7.2f  %7.2f", cyc, shader[k]);
            ++k;
        }
        printf("\n");
        printf("json {\"op\": \"%s\", \"cycles_1w\": %.3f, \"cycles_2w\": %.3f, \"cycles_3w\": %.3f, \"cycles_4w\": %.3f}\n", t.name, shader[0], shader[1], shader[2], shader[3]);
    }
    return 0;
}
