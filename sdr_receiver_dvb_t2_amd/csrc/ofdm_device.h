// ofdm_device.h -- device code of the OFDM side that more than one translation unit needs: the in-register FFT pieces and the two halves of
// a ONE-symbol transform (ofdm_kernels.hip: fft_stage_a_kernel / fft_stage_bc_kernel / fft_one_sync_kernel), the reference's atan2
// approximation and per-pilot estimate, and a symbol's synchronisation floats from its pilots (sym_sync_body). front_kernels.hip includes it
// for the launch that runs a chunk's front end AND the symbol's transform (front_fft_one_kernel). Both including files are compiled with
// -ffp-contract=off; the arithmetic is the reference's, operation for operation.
#pragma once
#include <hip/hip_runtime.h>
#include "ofdm_kernels.h"
#include "cp_device.h"
#include "loop_device.h"

#pragma clang fp contract(off)

namespace t2gpu {

struct cf { float x, y; };
__device__ __forceinline__ cf cmul(cf a, cf b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }

// in-register radix-2 decimation-in-frequency FFT of R points (R = 32 or 16); output in bit-reversed order
template <int R>
__device__ __forceinline__ void fft_reg(cf (&v)[32])
{
    constexpr float TW_C[16] = {1.0f, 0.98078528040323f, 0.92387953251129f, 0.83146961230255f, 0.70710678118655f,
                                0.55557023301960f, 0.38268343236509f, 0.19509032201613f, 0.0f, -0.19509032201613f,
                                -0.38268343236509f, -0.55557023301960f, -0.70710678118655f, -0.83146961230255f,
                                -0.92387953251129f, -0.98078528040323f};
    constexpr float TW_S[16] = {0.0f, 0.19509032201613f, 0.38268343236509f, 0.55557023301960f, 0.70710678118655f,
                                0.83146961230255f, 0.92387953251129f, 0.98078528040323f, 1.0f, 0.98078528040323f,
                                0.92387953251129f, 0.83146961230255f, 0.70710678118655f, 0.55557023301960f,
                                0.38268343236509f, 0.19509032201613f};          // W_32^m = C[m] - j S[m]
#pragma unroll
    for (int half = R / 2; half >= 1; half >>= 1) {
#pragma unroll
        for (int base = 0; base < R; base += 2 * half) {
#pragma unroll
            for (int i = 0; i < half; ++i) {
                const cf a = v[base + i], b = v[base + i + half];
                v[base + i] = {a.x + b.x, a.y + b.y};
                const cf d = {a.x - b.x, a.y - b.y};
                const int m = i * (16 / half);                                   // exponent of W_32
                v[base + i + half] = (m == 0) ? d : cmul(d, cf{TW_C[m], -TW_S[m]});
            }
        }
    }
}
// w^k for k = 1 .. 31 from w, w^2, w^4, w^8, w^16 (pw[0..4], table values): the product over the set bits of k, at most four
// multiplications deep.
// The stage twiddles W_N^(t k) are powers of the lane's own W_N^t: read per (lane, k) from the table they were 62 eight-byte
// gathers per lane and symbol -- as many bytes as the symbol itself, 30 to 64 distinct lines per load instruction -- and the kernel
// ran at the address rate of the CU; as powers they cost five reads and ~50 complex multiplications per stage.
template <int K>
__device__ __forceinline__ cf cpow_bits(const cf (&pw)[5])
{
    static_assert(K >= 1 && K < 32, "exponent");
    constexpr int lo = K & -K;                                   // lowest set bit
    constexpr int idx = lo == 1 ? 0 : lo == 2 ? 1 : lo == 4 ? 2 : lo == 8 ? 3 : 4;
    if constexpr ((K & (K - 1)) == 0) return pw[idx];
    else return cmul(cpow_bits<(K & (K - 1))>(pw), pw[idx]);
}
// pw[b] = W_N^(m 2^b), each a table value (squaring the first would double its angle error every time), from five planes laid out
// so that a wavefront's lanes read consecutive entries (t2gpu_ofdm.cpp)
__device__ __forceinline__ void cpow_table(const float2 *__restrict__ planes, int plane_len, int lane_index, cf (&pw)[5])
{
#pragma unroll
    for (int b = 0; b < 5; ++b) { const float2 w = planes[b * plane_len + lane_index]; pw[b] = cf{w.x, w.y}; }
}
template <int R, int I = 1>
__device__ __forceinline__ void twiddle_powers(cf (&v)[32], const cf (&pw)[5]);

template <int R> __device__ __forceinline__ constexpr int bitrev(int i)
{
    int r = 0;
    for (int b = 1, s = R >> 1; s >= 1; b <<= 1, s >>= 1) if (i & b) r |= s;
    return r;
}

// v[r] *= w^bitrev(r) for r = 1 .. 31
template <int R, int I>
__device__ __forceinline__ void twiddle_powers(cf (&v)[32], const cf (&pw)[5])
{
    if constexpr (I < 32) {
        constexpr int k = bitrev<R>(I);
        v[I] = cmul(v[I], cpow_bits<k>(pw));
        twiddle_powers<R, I + 1>(v, pw);
    }
}


// (one lane's share of stage A: tid = the lane's number among the T of a symbol)
template <int T2>
__device__ __forceinline__ void fft_stage_a_body(const float2 *__restrict__ x, float2 *__restrict__ sc, const float2 *__restrict__ twiddle, int tid)
{
    constexpr int T = 32 * T2, N = 32 * T;
    cf v[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) { const float2 a = x[tid + T * j]; v[j] = {a.x, a.y}; }
    fft_reg<32>(v);
    {
        cf pw[5];
        cpow_table(twiddle + N, T, tid, pw);
        twiddle_powers<32>(v, pw);
    }
#pragma unroll
    for (int r = 0; r < 32; ++r) sc[bitrev<32>(r) * T + tid] = make_float2(v[r].x, v[r].y);
}


constexpr int FFT_BC_LDS_FLOATS = 32 * 8 * 33;
template <int T2>
__device__ __forceinline__ void fft_stage_bc_body(const float2 *__restrict__ scratch, float2 *__restrict__ out,
                                                  const float2 *__restrict__ twiddle, float *lds /* [FFT_BC_LDS_FLOATS]: rows (q1, k1 of this workgroup) of T2 values over t1 */,
                                                  int block = -1)
{
    constexpr int T = 32 * T2, N = 32 * T, PITCH = 33;
    if (block < 0) block = (int)blockIdx.x;
    const int sym = block / 4, kb = (block % 4) * 8;
    const int l = (int)threadIdx.x, k1l = l / T2, t1n = l % T2, k1n = kb + k1l;
    const float2 *sc = scratch + (size_t)sym * N;
    float2 *y = out + (size_t)sym * N;
    cf u[32];
#pragma unroll
    for (int t2 = 0; t2 < 32; ++t2) { const float2 a = sc[k1n * T + t1n + T2 * t2]; u[t2] = {a.x, a.y}; }
    fft_reg<32>(u);
    {
        cf pw[5];
        cpow_table(twiddle + N + 5 * T, T2, t1n, pw);
        twiddle_powers<32>(u, pw);
    }
    cf w2[32];
#pragma unroll
    for (int plane = 0; plane < 2; ++plane) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 32; ++r) lds[(bitrev<32>(r) * 8 + k1l) * PITCH + t1n] = plane ? u[r].y : u[r].x;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            const int pair = (T2 == 32) ? l : l + 8 * T2 * (e / T2);               // local (q1, k1) pair = q1 * 8 + k1l
            const float f = lds[pair * PITCH + (e % T2)];
            if (plane) w2[e].y = f; else w2[e].x = f;
        }
    }
    // the lane's pair(s) in the numbering of fft_fwd_shift_kernel: id = q1 * 32 + k1
    const int q1 = l / 8, k1 = kb + l % 8;
    if (T2 == 32) {
        fft_reg<32>(w2);
        const int id = q1 * 32 + k1;
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const int k = id + 1024 * bitrev<32>(r);
            y[(k + N / 2) & (N - 1)] = make_float2(w2[r].x, w2[r].y);
        }
    } else {
        cf a[32], b[32];
#pragma unroll
        for (int e = 0; e < 16; ++e) { a[e] = w2[e]; b[e] = w2[16 + e]; }
        fft_reg<16>(a);
        fft_reg<16>(b);
        const int ida = q1 * 32 + k1, idb = (q1 + 16) * 32 + k1;                   // second pair: local index + 128 = q1 + 16
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q2 = bitrev<16>(r);
            y[(ida + 1024 * q2 + N / 2) & (N - 1)] = make_float2(a[r].x, a[r].y);
            y[(idb + 1024 * q2 + N / 2) & (N - 1)] = make_float2(b[r].x, b[r].y);
        }
    }
}


__device__ __forceinline__ float atan2_approx_dev(float y, float x)         // DSP/fast_math.h:61-81
{
    const float PI = 3.14159274101257324219f, PI_2 = 1.57079637050628662109f;
    if (x == 0.0f) return y > 0.0f ? PI_2 : -PI_2;
    if (y == 0.0f) return x > 0.0f ? 0.0f : -PI;
    const float abs_x = fabsf(x), abs_y = fabsf(y);
    const bool min_x = abs_x < abs_y;
    const float a = min_x ? abs_x / abs_y : abs_y / abs_x;
    const float s = a * a;
    float r = ((-4.6496475e-2f * s + 1.5931422e-1f) * s - 3.2762276e-1f) * s * a + a;
    if (min_x) r = PI_2 - r;
    if (x < 0.0f) r = PI - r;
    if (y < 0.0f) r = -r;
    return r;
}

struct PilotEst { float angle, amp, er, ei; };

__device__ __forceinline__ PilotEst pilot_estimate(float2 cell, float refer, float amp_pilot, int recip)
{
    PilotEst p;
    p.er = cell.x * refer; p.ei = cell.y * refer;                              // est_pilot = cell * pilot_refer
    p.angle = atan2_approx_dev(p.ei, p.er);
    const float mag = sqrtf(cell.x * cell.x + cell.y * cell.y);
    p.amp = recip ? mag * (1.0f / amp_pilot) : mag / amp_pilot;                // sqrt(norm(cell)) / amp_pilot (EqParams::recip_amp)
    return p;
}

template <int NL>
__device__ __forceinline__ void sym_sync_body(const EqParams &p, const float2 *__restrict__ symbol, int idx_symbol,
                                              const float2 *__restrict__ buffered, int guard, float4 *cp_out,
                                              float2 *__restrict__ sync, float *h_small, unsigned *h_flag, unsigned seq, float *sy_lds, T2DevLoop *loop,
                                              const float4 *cp_ready = nullptr /* the guard correlation, formed already (fft_one_sync_body) */)
{
    __shared__ int sh_lower;
    const int tid = threadIdx.x;
    const int row = idx_symbol - p.n_p2;
    const int nseg = p.seg_count[row];
    const float2 *cell = symbol + p.l_nulls;
    const float *refer = p.refer + (size_t)row * p.k_total;
    const int4 *segs = p.segs + (size_t)row * p.max_seg;
    float4 *l4 = reinterpret_cast<float4 *>(sy_lds);
    if (tid == 0) sh_lower = 0;
    __syncthreads();
    int lower = 0;
    for (int k = tid; k <= nseg; k += NL) {
        const int pc = k == 0 ? segs[0].x : segs[k - 1].y;                     // entry 0: the symbol's first pilot; entry k: segment k - 1's right pilot
        const PilotEst e = pilot_estimate(cell[pc], refer[pc], 1.0f, 0);
        const bool upper = pc > p.k_total / 2;
        l4[k] = k == 0 ? make_float4(e.er, e.ei, 0.0f, 0.0f) : make_float4(e.er, e.ei, e.angle, upper ? 1.0f : 0.0f);
        lower += (k >= 1 && !upper) ? 1 : 0;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) lower += __shfl_xor(lower, d, 64);
    if ((tid & 63) == 0 && lower) atomicAdd(&sh_lower, lower);
    float4 cp = make_float4(0.f, 0.f, 0.f, 0.f);
    if (buffered && !cp_ready) {
        double (*red)[256] = reinterpret_cast<double (*)[256]>(sy_lds + 4 * (size_t)(p.max_seg + 2));
        cp = cp_correlate_body<NL>(buffered, p.fft_size, guard, red);               // (its barriers also publish l4 / sh_lower)
    } else {
        if (buffered && tid == 0) cp = *cp_ready;                                   // (lane 0 is the one that uses it)
        __syncthreads();
    }
    lower = sh_lower;
    const int lane = tid;
    const int comp = lane % 3, second = lane / 3;                               // eq_sync_kernel's chains: lanes 0..2 first set, 3..5 second
    float acc = 0.0f;
    if (lane < 6) {
        int k = second ? lower + 1 : (comp == 2 ? 1 : 0);
        const int kend = second ? nseg : lower;
        const float *q = sy_lds + comp;
        for (; k + 8 <= kend + 1; k += 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = q[4 * (k + u)];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += t[u];
        }
        for (; k <= kend; ++k) acc += q[4 * k];
    }
    if (tid >= 64) return;
    const float s1r = __shfl(acc, 0, 64), s1i = __shfl(acc, 1, 64), a1 = __shfl(acc, 2, 64);
    const float s2r = __shfl(acc, 3, 64), s2i = __shfl(acc, 4, 64), a2 = __shfl(acc, 5, 64);
    if (lane == 0) {
        const float2 sv = make_float2(atan2_approx_dev(s2i, s2r) + atan2_approx_dev(s1i, s1r), a2 - a1);
        if (sync) *sync = sv;
        if (cp_out && buffered) *cp_out = cp;
        if (loop) {
            // the tracking loops on the device (loop_device.h): t2gpu_sync_frequency (dvbt2_demodulator.cpp:328-330) when the guard correlation
            // was formed, then t2gpu_sync_symbol's phase filter (:429) -- PiFilter::step's operations in its order
            if (buffered) {
                float integral = loop->f_int + loop->f_ki * cp.z;
                const float out = integral + loop->f_kp * cp.z;
                const float mx = 1.0f / (float)p.fft_size;
                if (integral > mx) integral = mx; else if (integral < -mx) integral = -mx;
                loop->f_int = integral;
                loop->frequency_est_filtered += out;
            }
            const float err = sv.x * 0.5f, mx2 = 3.14159274101257324219f * 2;
            float integral = loop->p_int + loop->p_ki * err;
            const float out = integral + loop->p_kp * err;
            if (integral > mx2) integral = mx2; else if (integral < -mx2) integral = -mx2;
            loop->p_int = integral;
            loop->pe = out;
            loop->fe = loop->frequency_est_filtered + loop->tuner;
        }
        if (h_small) {
            if (buffered) { h_small[0] = cp.x; h_small[1] = cp.y; h_small[2] = cp.z; h_small[3] = cp.w; }
            h_small[4] = sv.x; h_small[5] = sv.y;
            if (loop) { h_small[6] = loop->pe; h_small[7] = loop->fe; }
            __threadfence_system();
            __hip_atomic_store(h_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}


// the body of fft_one_sync_kernel for workgroup b of its eight (ofdm_kernels.hip); lds: FFT_BC_LDS_FLOATS floats
template <int T2>
__device__ __forceinline__ void fft_one_sync_body(const float2 *__restrict__ in, float2 *__restrict__ scratch, float2 *__restrict__ out,
                                                  const float2 *__restrict__ twiddle, unsigned *count, const EqParams &p, int idx_symbol,
                                                  const float2 *__restrict__ buffered, int guard, float4 *cp_out,
                                                  float2 *__restrict__ sync, float *h_small, unsigned *h_flag, unsigned seq, T2DevLoop *loop, const int b, float *lds,
                                                  long long *stamps = nullptr /* development: 16 wall-clock stamps of this workgroup */)
{
    __shared__ int sh_last;
    if (b < 4) {
        fft_stage_a_body<T2>(in, scratch, twiddle, b * (8 * T2) + (int)threadIdx.x);
        __syncthreads();
        if (stamps && threadIdx.x == 0) stamps[2] = wall_clock64();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(count + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    // The guard correlation needs the buffered symbol only, not its transform: the first of these four workgroups forms it NOW, while stage A
    // runs (round 6; the last workgroup formed it behind the transform before: ~2.5 us of every symbol's chain). Same function, same lanes,
    // same tree: the same floats. It reaches the last workgroup through memory, behind this workgroup's release below.
    float4 *cp_slot = reinterpret_cast<float4 *>(count + 8);
    if (b == 4 && buffered) {
        const float4 cp = cp_correlate_body<8 * T2>(buffered, p.fft_size, guard, reinterpret_cast<double (*)[256]>(lds));
        if (threadIdx.x == 0) *cp_slot = cp;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        for (; __hip_atomic_load(count + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 4u && spins < (1u << 24); ++spins) __builtin_amdgcn_s_sleep(1);
        if (spins == (1u << 24)) __builtin_trap();            // (~0.5 s: stage A's workgroups never arrived -- the launch fails loudly rather than hand on a spectrum of leftovers)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (stamps && threadIdx.x == 0) stamps[3] = wall_clock64();
    fft_stage_bc_body<T2>(scratch, out, twiddle, lds, b - 4);
    __syncthreads();
    if (stamps && threadIdx.x == 0) stamps[4] = wall_clock64();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned before = __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sh_last = before == 3u;
        if (sh_last) {
            __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(count + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (all four have passed their wait)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
    __syncthreads();
    if (!sh_last) return;
    if (stamps && threadIdx.x == 0) stamps[5] = wall_clock64();
    sym_sync_body<8 * T2>(p, out, idx_symbol, buffered, guard, cp_out, sync, h_small, h_flag, seq, lds, loop, cp_slot);
    if (stamps && threadIdx.x == 0) stamps[6] = wall_clock64();
}

}  // namespace t2gpu
