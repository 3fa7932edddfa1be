// cp_device.h -- device code shared by front_kernels.hip (cp_correlate_kernel) and ofdm_kernels.hip (sym_sync_kernel): the guard-interval
// correlation of symbol_acquisition (/root/reference/src/DVB_T2/dvbt2_demodulator.cpp:321-327) for one buffered symbol by one workgroup
// of 256 lanes, and the reference's atan2 approximation. Both including files are compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>

namespace t2gpu {

__device__ __forceinline__ float atan2_approx_ref(float y, float x)              // DSP/fast_math.h:61-81
{
    const float PI_F = 3.14159274101257324219f, PI_2 = 1.57079637050628662109f;
    if (x == 0.0f) return y > 0.0f ? PI_2 : -PI_2;
    if (y == 0.0f) return x > 0.0f ? 0.0f : -PI_F;
    const float abs_x = fabsf(x), abs_y = fabsf(y);
    const bool min_x = abs_x < abs_y;
    const float a = min_x ? abs_x / abs_y : abs_y / abs_x;
    const float s = a * a;
    float r = ((-4.6496475e-2f * s + 1.5931422e-1f) * s - 3.2762276e-1f) * s * a + a;
    if (min_x) r = PI_2 - r;
    if (x < 0.0f) r = PI_F - r;
    if (y < 0.0f) r = -r;
    return r;
}

// s: the symbol's guard + fft_size cells (guard first). All NL lanes of the workgroup call it (NL = 256, or 128 lanes that each stand for
// two of the 256: the same per-lane sums and the same tree, so the same value); the value is valid in lane 0:
// {sum.re, sum.im, frequency_est, 0}. Per-lane double sums of the float products (stride 256), folded by a tree over red[2][256].
template <int NL = 256>
__device__ __forceinline__ float4 cp_correlate_body(const float2 *__restrict__ s, int fft_size, int guard, double (*red)[256])
{
    static_assert(NL == 256 || NL == 128, "256 lanes, or 128 standing for 256");
    const float2 *cp = s + fft_size;
    for (int v = (int)threadIdx.x; v < 256; v += NL) {
        double sr = 0.0, si = 0.0;
        for (int i = 4 + v; i < guard - 4; i += 256) {
            const float2 a = cp[i], b = s[i];
            sr += (double)(a.x * b.x + a.y * b.y);                              // cp[i] * conj(sym[i])
            si += (double)(a.y * b.x - a.x * b.y);
        }
        red[0][v] = sr; red[1][v] = si;
    }
    __syncthreads();
    for (int t = 128; t > 0; t >>= 1) {
        if ((int)threadIdx.x < t) { red[0][threadIdx.x] += red[0][threadIdx.x + t]; red[1][threadIdx.x] += red[1][threadIdx.x + t]; }
        __syncthreads();
    }
    const float re = (float)red[0][0], im = (float)red[1][0];
    return make_float4(re, im, atan2_approx_ref(im, re) / (float)(fft_size << 1), 0.0f);
}

}  // namespace t2gpu
