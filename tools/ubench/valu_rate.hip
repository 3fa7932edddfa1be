// valu_rate.hip -- gfx950 micro-benchmark: cycles per wave64 VALU instruction for the op mix an int8 min-sum decoder
// can be built from (integer vs fp32 vs packed forms). 4 independent accumulator chains per lane, 1 wave per SIMD
// (256 threads per block, one block per CU) and again with 2/4 waves per SIMD.  Build: hipcc --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP 64
#define ITER 256

#define KERNEL(NAME, DECL, BODY)                                                        \
    __global__ void k_##NAME(int *out, int seed)                                          \
    {                                                                                     \
        DECL;                                                                             \
        for (int it = 0; it < ITER; ++it) {                                               \
            _Pragma("unroll") for (int r = 0; r < REP / 4; ++r) { BODY; }                 \
        }                                                                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = FIN;                                 \
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
__global__ void k_pk_add_f32(int *out, int seed)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 * 3.f, a2 = a0 * 5.f, a3 = a0 * 7.f, b = {(float)seed, 2.f};
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 4; ++r)
            asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (int)(a0.x + a1.x + a2.x + a3.y);
}

typedef void (*kfn)(int *, int);
struct Entry { const char *name; kfn fn; };
#define E(N) {#N, k_##N}

int main()
{
    Entry tests[] = {E(add_u32), E(min_i32), E(med3_i32), E(xor_b32), E(cndmask), E(cmp_cnd), E(sad_u32), E(add3_u32), E(bfe_i32),
                     E(perm_b32), E(lshl_or), E(pk_add_i16), E(pk_min_i16), E(pk_sub_i16_clamp), E(add_f32), E(min_f32), E(min_f32_abs),
                     E(med3_f32), E(fma_f32), E(mul_f32), E(cmp_f32_cnd), E(cvt_f32_i32), E(cvt_f32_ubyte0), E(pk_add_f32)};
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    double clk = prop.clockRate * 1e3;   // Hz (nominal max)
    int *out; hipMalloc(&out, (size_t)cus * 8 * 1024 * sizeof(int));
    printf("device %s, %d CUs, clock %.0f MHz\n", prop.name, cus, clk / 1e6);
    printf("%-20s %12s %12s %12s   (cycles per wave64 instruction per SIMD at nominal clock; wave occupancy 1/2/4 per SIMD)\n", "op", "1w", "2w", "4w");
    for (auto &t : tests) {
        printf("%-20s", t.name);
        for (int wps : {1, 2, 4}) {
            int threads = 256 * wps;     // wps waves on each of the 4 SIMDs
            if (threads > 1024) threads = 1024;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            t.fn<<<cus, threads>>>(out, 1);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < 5; ++r) t.fn<<<cus, threads>>>(out, r);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double insts_per_simd = 5.0 * (double)ITER * REP * wps;       // wave-instructions issued on each SIMD
            double cyc = (ms * 1e-3) * clk / insts_per_simd;
            printf(" %12.2f", cyc);
        }
        printf("\n");
    }
    return 0;
}
