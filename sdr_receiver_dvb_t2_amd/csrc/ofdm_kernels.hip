// ofdm_kernels.hip -- OFDM-side kernels for gfx950: batched 32K / 16K forward FFT with fftshift, and the data-symbol
// channel estimator / equaliser fused with the frequency de-interleaver. HBM-bound streaming work; no matrix cores.
#include "ofdm_kernels.h"
#include "cp_device.h"
#include "loop_device.h"
#include "ofdm_device.h"
#include "t2gpu_common.h"
#include <algorithm>

// data_symbol.cpp arithmetic is restated operation for operation (the reference is built without FMA)
#pragma clang fp contract(off)

namespace t2gpu {

// ====================================================================================================== FFT
// Replaces fast_fourier_transform::execute (/root/reference/src/DSP/fast_fourier_transform.h:62-70): forward complex DFT
// (FFTW3f there), unnormalised, followed by the two half-buffer memcpys that put DC at index N/2.
//
// One workgroup transforms one symbol in a single pass over HBM (read N, write N complex floats): the N = 32*32*R points
// live in registers (32 per lane), three in-register radix-32 / radix-R stages are separated by two all-to-all
// exchanges through LDS (real and imaginary planes one after the other: 128 KiB each for 32K), twiddles come from an
// L2-resident table of W_N^m. Lane-to-address maps are chosen so that every global access of a wavefront is one
// contiguous 512-byte run and every LDS access is bank-conflict free (row pitch 33).
// T2 = 32: N = 32768, 1024 threads. T2 = 16: N = 16384, 512 threads.
template <int T2>
__global__ __launch_bounds__(32 * T2) void fft_fwd_shift_kernel(const float2 *__restrict__ in, float2 *__restrict__ out,
                                                               const float2 *__restrict__ twiddle, int n_symbols, FftLayout lay)
{
    constexpr int T = 32 * T2, N = 32 * T, PITCH = 33;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    // 32K: one workgroup per symbol. Inside a grid-stride loop the compiler carried state across the back edge of the fully unrolled
    // body and spilled 142 of the 128 registers a 1024-lane workgroup may have (572 B of scratch per lane, 1.6x the algorithmic HBM
    // bytes, profiles/r01_rx_pmc.txt); straight-line it spills 20. 16K (512 lanes, 256 registers, no spills) keeps its persistent loop.
    for (int sym = blockIdx.x; sym < n_symbols; sym += gridDim.x) {
        // symbol `sym` of the batch starts at first + (sym / per_frame) * frame_stride + (sym % per_frame) * sym_stride cells
        const float2 *x = in + lay.first + (long)(sym / lay.per_frame) * lay.frame_stride + (long)(sym % lay.per_frame) * lay.sym_stride;
        float2 *y = out + (size_t)sym * N;
        cf v[32];
        // ---- stage A: thread t holds x[t + T*j]; 32-point DFT over j -> index k1 (bit-reversed in registers)
#pragma unroll
        for (int j = 0; j < 32; ++j) { const float2 a = x[tid + T * j]; v[j] = {a.x, a.y}; }
        fft_reg<32>(v);
        {                                                    // twiddle W_N^(t*k1), k1 = bitrev(r): powers of W_N^t
            cf pw[5];
            cpow_table(twiddle + N, T, tid, pw);
            twiddle_powers<32>(v, pw);
        }
        // ---- exchange 1: thread (k1, t1) := id k1*T2 + t1 collects y_k1[t1 + T2*t2], t2 = 0..31
        cf u[32];
        const int k1n = tid / T2, t1n = tid % T2;
#pragma unroll
        for (int plane = 0; plane < 2; ++plane) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 32; ++r) lds[bitrev<32>(r) * T + tid] = plane ? v[r].y : v[r].x;
            __syncthreads();
#pragma unroll
            for (int t2 = 0; t2 < 32; ++t2) {
                const float f = lds[k1n * T + t1n + T2 * t2];
                if (plane) u[t2].y = f; else u[t2].x = f;
            }
        }
        // ---- stage B: 32-point DFT over t2 -> q1; twiddle W_T^(t1*q1) = W_N^(32*t1*q1)
        fft_reg<32>(u);
        {                                                    // powers of W_N^(32*t1)
            cf pw[5];
            cpow_table(twiddle + N + 5 * T, T2, t1n, pw);
            twiddle_powers<32>(u, pw);
        }
        // ---- exchange 2: rows (q1, k1) of T2 values over t1, pitch 33. New thread id' -> pairs (q1, k1) with k1 fastest:
        //      32K: one pair per thread (q1 = id'/32, k1 = id'%32), 32 values; 16K: two pairs per thread, 16 values each.
        cf w2[32];
#pragma unroll
        for (int plane = 0; plane < 2; ++plane) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 32; ++r) lds[(bitrev<32>(r) * 32 + k1n) * PITCH + t1n] = plane ? u[r].y : u[r].x;
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 32; ++e) {
                const int pair = (T2 == 32) ? tid : tid + T * (e / T2);          // (q1, k1) pair index = q1*32 + k1
                const float f = lds[pair * PITCH + (e % T2)];
                if (plane) w2[e].y = f; else w2[e].x = f;
            }
        }
        // ---- stage C: T2-point DFT over t1 -> q2; output bin k = k1 + 32*q1 + 1024*q2, stored fft-shifted
        if (T2 == 32) {
            fft_reg<32>(w2);
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                const int k = tid + 1024 * bitrev<32>(r);                         // tid = q1*32 + k1 = k1 + 32*q1
                const int ks = (k + N / 2) & (N - 1);
                y[ks] = make_float2(w2[r].x, w2[r].y);
            }
        } else {
            cf a[32], b[32];
#pragma unroll
            for (int e = 0; e < 16; ++e) { a[e] = w2[e]; b[e] = w2[16 + e]; }
            fft_reg<16>(a);
            fft_reg<16>(b);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q2 = bitrev<16>(r);
                const int ka = tid + 1024 * q2, kb = tid + T + 1024 * q2;        // second pair index = tid + 512
                y[(ka + N / 2) & (N - 1)] = make_float2(a[r].x, a[r].y);
                y[(kb + N / 2) & (N - 1)] = make_float2(b[r].x, b[r].y);
            }
        }
        if constexpr (T2 == 32) break;                        // 32K: a single trip, no back edge (see above)
    }
}

// ---- the same transform of ONE symbol (or a few) on many CUs: the slot-shaped path hands the library one OFDM symbol at a time and
// waits for it, and one workgroup on one CU took 19 us for a 32K symbol (load, three register stages, two exchanges, store -- all of
// it latency of a single CU). After stage A the 32 values of k1 never meet again: the rest of the transform is 32 independent
// 32 x T2 transforms. So: launch 1 = stage A, 64 lanes per workgroup (T / 64 workgroups per symbol), the first exchange through a
// scratch array in memory (L2) instead of LDS; launch 2 = stages B and C, one workgroup per 8 values of k1 (8 * T2 lanes, 4
// workgroups per symbol), the second exchange in its own LDS, and the lanes of a store instruction cover 8 consecutive bins (whole
// 64-byte lines). Every value goes through the same operations in the same order as in fft_fwd_shift_kernel: bit-identical output
// (tests/test_ofdm_gpu.py holds the two against each other).
template <int T2>
__global__ __launch_bounds__(64) void fft_stage_a_kernel(const float2 *__restrict__ in, float2 *__restrict__ scratch,
                                                         const float2 *__restrict__ twiddle, FftLayout lay)
{
    constexpr int T = 32 * T2, N = 32 * T, WGS = T / 64;
    const int sym = (int)blockIdx.x / WGS, tid = ((int)blockIdx.x % WGS) * 64 + (int)threadIdx.x;
    const float2 *x = in + lay.first + (long)(sym / lay.per_frame) * lay.frame_stride + (long)(sym % lay.per_frame) * lay.sym_stride;
    fft_stage_a_body<T2>(x, scratch + (size_t)sym * N, twiddle, tid);
}

template <int T2>
__global__ __launch_bounds__(8 * T2) void fft_stage_bc_kernel(const float2 *__restrict__ scratch, float2 *__restrict__ out,
                                                              const float2 *__restrict__ twiddle)
{
    __shared__ float lds[FFT_BC_LDS_FLOATS];
    fft_stage_bc_body<T2>(scratch, out, twiddle, lds);
}

hipError_t launch_fft(int fft_size, const float2 *in, float2 *out, const float2 *twiddle, int n_symbols, int max_blocks,
                      hipStream_t s, const FftLayout *layout, float2 *scratch, int scratch_symbols)
{
    const FftLayout lay = layout ? *layout : FftLayout{0, 0, n_symbols > 0 ? n_symbols : 1, fft_size};
    if (scratch && n_symbols >= 1 && n_symbols <= scratch_symbols) {
        if (fft_size == 32768) {
            hipLaunchKernelGGL(fft_stage_a_kernel<32>, dim3(16 * n_symbols), dim3(64), 0, s, in, scratch, twiddle, lay);
            hipLaunchKernelGGL(fft_stage_bc_kernel<32>, dim3(4 * n_symbols), dim3(256), 0, s, scratch, out, twiddle);
            return hipGetLastError();
        }
        if (fft_size == 16384) {
            hipLaunchKernelGGL(fft_stage_a_kernel<16>, dim3(8 * n_symbols), dim3(64), 0, s, in, scratch, twiddle, lay);
            hipLaunchKernelGGL(fft_stage_bc_kernel<16>, dim3(4 * n_symbols), dim3(128), 0, s, scratch, out, twiddle);
            return hipGetLastError();
        }
    }
    const int blocks = (fft_size == 32768 || n_symbols < max_blocks) ? n_symbols : max_blocks;   // 32K: one workgroup per symbol
    if (fft_size == 32768) {
        const int lds_bytes = 32 * 1024 * 4 > 32 * 32 * 33 * 4 ? 32 * 1024 * 4 : 32 * 32 * 33 * 4;
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(fft_fwd_shift_kernel<32>), lds_bytes)) return e;
        hipLaunchKernelGGL(fft_fwd_shift_kernel<32>, dim3(blocks), dim3(1024), lds_bytes, s, in, out, twiddle, n_symbols, lay);
    } else if (fft_size == 16384) {
        const int lds_bytes = 32 * 32 * 33 * 4;      // >= 32*512*4
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(fft_fwd_shift_kernel<16>), lds_bytes)) return e;
        hipLaunchKernelGGL(fft_fwd_shift_kernel<16>, dim3(blocks), dim3(512), lds_bytes, s, in, out, twiddle, n_symbols, lay);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ====================================================================================================== equaliser
// Replaces data_symbol::execute (/root/reference/src/DVB_T2/data_symbol.cpp:108-335) and the equaliser part of
// p2_symbol::execute (/root/reference/src/DVB_T2/p2_symbol.cpp:89-262), which differ only in the pilot amplitudes. Between two consecutive pilots the
// reference interpolates phase and amplitude linearly by repeated float addition and de-rotates every data cell with a
// LUT cos/sin divided by the amplitude; segments between pilots are independent, so one lane walks one segment with
// exactly the reference's sequence of float operations. The frequency de-interleaver (out[h[d]]) is fused in.

// ---- output-range form ----------------------------------------------------------------------------------------------------------
// (Round 2's kernel walked groups of segments and stored every cell with an 8-byte write somewhere in the symbol's 219 KB of output,
// depending on L2 to merge them into lines: 1.2 TB/s. Retired in round 6 -- its A/B against this form is settled, profiles/HISTORY.md.)
// The de-interleaver's scatter lands in LDS. One workgroup = one symbol x one RANGE of output positions
// [q0, q1) (a third of a 32K symbol: 73 KB as (re, im) pairs, two workgroups of 512 lanes per CU):
//   phase 1: one lane per pilot-to-pilot segment of the WHOLE symbol runs the reference's angle / amplitude recurrences (serial
//            float additions, in the reference's order) and leaves (angle, amplitude) of the cells that land in the range AT THEIR
//            OUTPUT POSITION in LDS. Every range of the symbol repeats the recurrences; they are ~290 chains of ~95 steps.
//   phase 2: all lanes over the range's cells in CELL order (host list: ascending carriers, so the spectrum is read in ascending,
//            half-dense runs): table read, two divisions, rotation -- the reference's per-cell operations -- result back to the
//            same LDS slot.
//   phase 3: the range leaves LDS as one contiguous run.
// Bit-identical for every number of ranges (tests/test_ofdm_gpu.py runs three against the oracle).
// Measured (config 3, 2832 data symbols; skipping phases one at a time): reads 0.18 ms, phase 1 0.12, phase 2 0.11, phase 3 0.12 --
// they add up, two workgroups per CU overlap little of it: 0.52 ms (0.90 ms for the retired segment-group kernel). More, smaller ranges
// repeat phase 1 more often (+0.09 ms per range); a half symbol per workgroup of 1024 lanes leaves one workgroup per CU (0.55 ms).
// IT = cells per lane (range <= IT * THREADS): every global input of phase 2 (list entry, spectrum cell) is in registers before
// phase 1 starts, so the recurrences run under their latency and phase 2 waits for LDS and the cos / sin table only.
template <int IT, int THREADS>
__global__ __launch_bounds__(THREADS) void eq_split_kernel(EqParams p, const float2 *__restrict__ symbols,
                                                           const int32_t *__restrict__ symbol_index, float2 *__restrict__ out,
                                                           float4 *__restrict__ pilot_scratch, int n_symbols)
{
    extern __shared__ __attribute__((aligned(16))) float2 eqs_buf[];
    const float K_TABLE = 32767.0f / (2.0f * 3.14159274101257324219f);
    const float PI = 3.14159274101257324219f;
    const int NS = p.n_splits;
    // workgroup w -> XCD w % 8 (round-robin dispatch). All ranges of a symbol run on one XCD back to back (they read the same
    // spectrum), and in the frame layout an XCD walks the frames of one table ROW before the next row (the row's lists: 110 KB of
    // `sel`, 65 KB of `cellq`, enter each L2 once).
    const int wg = (int)blockIdx.x, xcd = wg & 7, i = wg >> 3;
    int b, s;
    if (p.per_frame > 1) {
        const int frames = n_symbols / p.per_frame, fx = (frames + 7) >> 3;
        const int row = i / (fx * NS), rem = i - row * fx * NS;
        const int fr_x = rem / NS;
        s = rem - fr_x * NS;
        const int frame = fr_x * 8 + xcd;
        if (row >= p.per_frame || frame >= frames) return;
        b = frame * p.per_frame + row;
    } else {
        b = 8 * (i / NS) + xcd;
        s = i % NS;
    }
    if (b >= n_symbols) return;
    const int fr = p.per_frame ? b / p.per_frame : 0, lo = p.per_frame ? b - fr * p.per_frame : 0;
    const int idx_symbol = p.per_frame ? p.first + lo : symbol_index[b];
    const int row = idx_symbol - p.n_p2;
    const int nseg = p.seg_count[row];
    const float2 *cell = symbols + (p.per_frame ? (size_t)(fr * p.in_syms_per_frame + idx_symbol) : (size_t)b) * p.fft_size + p.l_nulls;
    const uint8_t *map = p.map + (size_t)row * p.k_total;
    const float *refer = p.refer + (size_t)row * p.k_total;
    const int4 *segs = p.segs + (size_t)row * p.max_seg;
    const uint2 *cq = reinterpret_cast<const uint2 *>(p.cellq) + (size_t)row * (p.cq_steps / 4) * p.max_seg;
    const int q0 = eq_split_q(p.c_data, NS, s), q1 = eq_split_q(p.c_data, NS, s + 1);
    const unsigned range = (unsigned)(q1 - q0);
    const int tid = (int)threadIdx.x;
    // ---- inputs of phase 2: lane cell u is entry tid + u * THREADS of the range's list (lanes past the end repeat entry 0's slot
    // harmlessly: carrier 0, slot 0, nothing stored)
    const uint32_t *sel = p.sel + (size_t)row * p.c_data + q0;
    uint32_t e[IT];
    float2 v[IT];
#pragma unroll
    for (int u = 0; u < IT; ++u) {
        const int k = tid + u * THREADS;
        e[u] = k < (int)range ? sel[k] : (uint32_t)q0;
    }
#pragma unroll
    for (int u = 0; u < IT; ++u) v[u] = cell[e[u] >> 16];
    // ---- phase 1
    for (int seg = tid; seg < nseg; seg += THREADS) {
        const int4 sg = segs[seg];
        const int pl = sg.x, pr = sg.y, n = sg.w;
        const uint2 *cqs = cq + seg;
        const float2 cl = cell[pl], cr = cell[pr];
        const float refer_l = refer[pl], refer_r = refer[pr];
        const uint8_t tl = map[pl], tr = map[pr];
        const PilotEst L = pilot_estimate(cl, refer_l, tl == T2_P2PILOT ? p.amp_p2 : (tl == T2_CONTINUAL ? p.amp_cp : p.amp_sp), p.recip_amp);
        const PilotEst R = pilot_estimate(cr, refer_r, tr == T2_P2PILOT ? p.amp_p2 : (tr == T2_CONTINUAL ? p.amp_cp : p.amp_sp), p.recip_amp);
        float dif_angle = R.angle - L.angle;
        if (dif_angle > PI) dif_angle = PI * 2.0f - dif_angle;                  // as written in the reference (:189-191)
        else if (dif_angle < -PI) dif_angle = PI * 2.0f + dif_angle;
        const float delta_angle = dif_angle / (float)(n + 1);
        const float delta_amp = (R.amp - L.amp) / (float)(n + 1);
        float angle_est = L.angle, amp_est = L.amp;
        // destinations: four steps per 8-byte read, EQS_PU steps ahead of the chain; 0xffff past the segment's end
        static_assert(EQS_PU == 8, "two 8-byte reads per trip");
        for (int k0 = 0; k0 < n; k0 += 8) {
            const uint2 w0 = cqs[(size_t)(k0 >> 2) * p.max_seg], w1 = cqs[(size_t)((k0 >> 2) + 1) * p.max_seg];
            const unsigned q[8] = {w0.x & 0xffffu, w0.x >> 16, w0.y & 0xffffu, w0.y >> 16, w1.x & 0xffffu, w1.x >> 16, w1.y & 0xffffu, w1.y >> 16};
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                angle_est += delta_angle; amp_est += delta_amp;
                eqs_buf[min(q[u] - (unsigned)q0, range)] = make_float2(angle_est, amp_est);   // slot `range`: other ranges' cells, steps past n
            }
        }
        if (s == 0) {
            float4 *ps = pilot_scratch + (size_t)b * (p.max_seg + 1);
            if (seg == 0) ps[0] = make_float4(L.er, L.ei, 0.0f, 0.0f);
            ps[seg + 1] = make_float4(R.er, R.ei, R.angle, pr > p.k_total / 2 ? 1.0f : 0.0f);
        }
    }
    __syncthreads();
    // ---- phase 2
    const float2 *__restrict__ lut = p.lut_cs;
    constexpr int U = 4;
#pragma unroll
    for (int u0 = 0; u0 < IT; u0 += U) {
        float2 cs[U], aa[U];
#pragma unroll
        for (int u = u0; u < u0 + U && u < IT; ++u) {
            aa[u - u0] = eqs_buf[(e[u] & 0xffffu) - (unsigned)q0];
            cs[u - u0] = lut[(int)(aa[u - u0].x * K_TABLE + 32767) & 65535];
        }
#pragma unroll
        for (int u = u0; u < u0 + U && u < IT; ++u) {
            if (tid + u * THREADS >= (int)range) continue;
            const float amp = aa[u - u0].y;
            const float dr = cs[u - u0].x / amp, di = cs[u - u0].y / amp;
            eqs_buf[(e[u] & 0xffffu) - (unsigned)q0] = make_float2(v[u].x * dr + v[u].y * di, v[u].y * dr - v[u].x * di);
        }
    }
    __syncthreads();
    // ---- phase 3
    float2 *o = p.per_frame ? out + (size_t)fr * p.out_frame_stride + p.out_offset + (size_t)lo * p.c_data : out + (size_t)b * p.c_data;
    for (int k = tid; k < (int)range; k += THREADS) {
        const int q = q0 + k, at = q - p.out_skip;
        const float2 eq = eqs_buf[k];
        if (at >= 0) o[at] = eq;
        else if (p.skip_out) p.skip_out[(size_t)fr * p.out_skip + q] = eq;
        if (p.pub_cells) p.pub_cells[q] = eq;
    }
    if (p.pub_cells) {                                      // (one symbol: NS workgroups get here)
        __threadfence_system();
        __syncthreads();
        if (tid == 0) {
            const unsigned before = __hip_atomic_fetch_add(p.pub_count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (before == (unsigned)NS - 1) {
                __hip_atomic_store(p.pub_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __threadfence_system();
                __hip_atomic_store(p.pub_flag, p.pub_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

template <int IT, int THREADS>
static hipError_t launch_eq_split(const EqParams &p, const float2 *symbols, const int32_t *symbol_index, int n_symbols, float2 *out,
                                  float4 *pilot_scratch, int bytes, hipStream_t s)
{
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(eq_split_kernel<IT, THREADS>), bytes)) return e;
    unsigned grid = (unsigned)(((n_symbols + 7) / 8) * 8 * p.n_splits);
    if (p.per_frame > 1) grid = (unsigned)(8 * p.per_frame * ((n_symbols / p.per_frame + 7) / 8) * p.n_splits);
    hipLaunchKernelGGL((eq_split_kernel<IT, THREADS>), dim3(grid), dim3(THREADS), bytes, s, p, symbols, symbol_index, out, pilot_scratch, n_symbols);
    return hipGetLastError();
}

// phase_offset = atan2(sum_pilot_2) + atan2(sum_pilot_1), sample_rate_offset = sum_angle_2 - sum_angle_1 (:319-324)
// The sums are float additions in carrier order (the reference's loop order): serial chains per symbol -- six of them, (re, im, angle)
// over the pilots of the lower half of the spectrum and the same over the upper half, and they do not depend on each other. One
// wavefront per symbol: all lanes stage the symbol's per-pilot terms in LDS (coalesced 16-byte reads, eight in flight per lane), then
// lanes 0..5 take one chain each and add their component pilot by pilot. A chain step is one LDS read and one add. (One thread per
// symbol doing everything: 0.47 ms per 2280 symbols; one wavefront with v_readlane broadcasts into identical accumulators: 0.03 ms
// for the data symbols but 0.18 ms for the 38 P2 symbols, whose 4640 pilots each were ~90 cycles apiece in one chain of three.)
__global__ __launch_bounds__(64) void eq_sync_kernel(EqParams p, const int32_t *__restrict__ symbol_index, const float4 *__restrict__ pilot_scratch,
                                                     float2 *__restrict__ sync, int n_symbols)
{
    extern __shared__ __attribute__((aligned(16))) float sy_lds[];             // [nseg + 1][4]
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= n_symbols) return;
    const int nseg = p.seg_count[(p.per_frame ? p.first + b % p.per_frame : symbol_index[b]) - p.n_p2];
    const float4 *ps = pilot_scratch + (size_t)b * (p.max_seg + 1);
    float4 *l4 = reinterpret_cast<float4 *>(sy_lds);
    int lower = 0;                                                              // entries 1..nseg whose pilot lies in the lower half
    constexpr int U = 8;
    for (int k0 = lane; k0 <= nseg; k0 += 64 * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int k = k0 + 64 * u; v[u] = k <= nseg ? ps[k] : make_float4(0.f, 0.f, 0.f, 1.f); }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + 64 * u;
            if (k <= nseg) { l4[k] = v[u]; lower += (k >= 1 && v[u].w == 0.0f) ? 1 : 0; }
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) lower += __shfl_xor(lower, d, 64);
    __syncthreads();
    // pilots are in carrier order: entries 1..lower feed the first set of sums, lower+1..nseg the second; entry 0 is the first pilot
    // of the symbol, which has no angle term (:153-162) and opens the first set's (re, im)
    const int comp = lane % 3, second = lane / 3;                               // lanes 0..2: first set, 3..5: second
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
    const float s1r = __shfl(acc, 0, 64), s1i = __shfl(acc, 1, 64), a1 = __shfl(acc, 2, 64);
    const float s2r = __shfl(acc, 3, 64), s2i = __shfl(acc, 4, 64), a2 = __shfl(acc, 5, 64);
    if (lane == 0) sync[b] = make_float2(atan2_approx_dev(s2i, s2r) + atan2_approx_dev(s1i, s1r), a2 - a1);
}

// ---- the synchronisation floats of ONE symbol straight from its spectrum (the slot-shaped path, t2gpu_demod.cpp) ----------------
// phase_offset and sample_rate_offset (data_symbol.cpp:319-324, p2_symbol.cpp / fc_symbol.cpp likewise) depend on the symbol's pilots
// only: nothing of the equaliser's cell work. In the reference's per-symbol loop they are what the NEXT chunk's NCO / resampler values
// wait for, so here they are formed by a launch of their own right behind the FFT -- the per-pilot terms by pilot_estimate exactly as
// eq_split_kernel / eq_data_kernel form them, folded in carrier order exactly as eq_sync_kernel folds them: the same floats, bit for
// bit (tests/test_ofdm_gpu.py) -- while the equaliser runs beside the next chunk's front end. The same workgroup also forms the guard
// correlation of the buffered symbol (symbol_acquisition, dvbt2_demodulator.cpp:321-327; cp_device.h: cp_correlate_kernel's body) and
// stores all six floats to page-locked host memory with the sequence word behind them.
// NL lanes (256, or the 128 of fft_stage_bc_kernel<16>); sy_lds: (max_seg + 2) x 16 bytes, then the correlation's 2 x 256 doubles
__global__ __launch_bounds__(256) void sym_sync_kernel(EqParams p, const float2 *__restrict__ symbol, int idx_symbol,
                                                       const float2 *__restrict__ buffered, int guard, float4 *cp_out,
                                                       float2 *__restrict__ sync, float *h_small, unsigned *h_flag, unsigned seq, T2DevLoop *loop)
{
    extern __shared__ __attribute__((aligned(16))) float sy_lds_dyn[];
    sym_sync_body<256>(p, symbol, idx_symbol, buffered, guard, cp_out, sync, h_small, h_flag, seq, sy_lds_dyn, loop);
}

// ---- the second launch of a ONE-symbol FFT with the symbol's synchronisation floats behind it in the same launch: the last of its four
// workgroups to finish (a counter; its stores released, the others' acquired) runs sym_sync_body on the spectrum the four have just
// written. One launch and one launch gap less on the slot-shaped path's per-symbol critical path. The body's LDS is the FFT's exchange
// buffer (33 KB: pilot tables of up to FFT_SYNC_MAX_SEG segments; denser tables -- P2 -- take sym_sync_kernel).
template <int T2>
__global__ __launch_bounds__(8 * T2) void fft_stage_bc_sync_kernel(const float2 *__restrict__ scratch, float2 *__restrict__ out,
                                                                   const float2 *__restrict__ twiddle, unsigned *count, EqParams p, int idx_symbol,
                                                                   const float2 *__restrict__ buffered, int guard, float4 *cp_out,
                                                                   float2 *__restrict__ sync, float *h_small, unsigned *h_flag, unsigned seq, T2DevLoop *loop)
{
    __shared__ __attribute__((aligned(16))) float lds[FFT_BC_LDS_FLOATS];
    __shared__ int sh_last;
    fft_stage_bc_body<T2>(scratch, out, twiddle, lds);
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned before = __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sh_last = before == gridDim.x - 1;
        if (sh_last) { __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
    }
    __syncthreads();
    if (!sh_last) return;
    sym_sync_body<8 * T2>(p, out, idx_symbol, buffered, guard, cp_out, sync, h_small, h_flag, seq, lds, loop);
}

// ---- ... and the FIRST launch in it as well (round 5): ONE launch per symbol's transform. Workgroups 0-3 run stage A (the T lanes of the
// symbol, 8 * T2 of them each; no LDS, no barrier), raise a counter behind their stores; workgroups 4-7 wait for the four (a workgroup
// only ever waits for lower-numbered ones, which were dispatched before it; bounded), run stages B and C, and the last of them runs
// sym_sync_body as above. Every value goes through the operations of the two launches in their order: bit-identical spectrum and floats.
// While decodes are resident every launch of the chain costs tens of microseconds more than its body (profiles/HISTORY.md, round 5):
// the per-symbol critical path has one launch less.
template <int T2>
__global__ __launch_bounds__(8 * T2) void fft_one_sync_kernel(const float2 *__restrict__ in, float2 *__restrict__ scratch, float2 *__restrict__ out,
                                                              const float2 *__restrict__ twiddle, unsigned *count, EqParams p, int idx_symbol,
                                                              const float2 *__restrict__ buffered, int guard, float4 *cp_out,
                                                              float2 *__restrict__ sync, float *h_small, unsigned *h_flag, unsigned seq, T2DevLoop *loop)
{
    __shared__ __attribute__((aligned(16))) float lds[FFT_BC_LDS_FLOATS];
    fft_one_sync_body<T2>(in, scratch, out, twiddle, count, p, idx_symbol, buffered, guard, cp_out, sync, h_small, h_flag, seq, loop, (int)blockIdx.x, lds);
}

hipError_t launch_fft_sym_sync(int fft_size, const float2 *in, float2 *out, const float2 *twiddle, const FftLayout &lay, float2 *scratch, unsigned *count,
                               const EqParams &p, int idx_symbol, const float2 *buffered, int guard, float4 *cp_out, float2 *sync, float *h_small,
                               unsigned *h_flag, unsigned seq, hipStream_t s, T2DevLoop *loop, bool one_launch)
{
    if ((p.max_seg + 2) * 16 + 2 * 256 * 8 > FFT_BC_LDS_FLOATS * 4 || !scratch || !count) return hipErrorInvalidValue;
    if (one_launch && (fft_size == 32768 || fft_size == 16384)) {
        const float2 *x = in + lay.first;                     // (one symbol: symbol 0 of the layout)
        if (fft_size == 32768)
            hipLaunchKernelGGL(fft_one_sync_kernel<32>, dim3(8), dim3(256), 0, s, x, scratch, out, twiddle, count, p, idx_symbol, buffered, guard, cp_out, sync,
                               h_small, h_flag, seq, loop);
        else
            hipLaunchKernelGGL(fft_one_sync_kernel<16>, dim3(8), dim3(128), 0, s, x, scratch, out, twiddle, count, p, idx_symbol, buffered, guard, cp_out, sync,
                               h_small, h_flag, seq, loop);
    } else if (fft_size == 32768) {
        hipLaunchKernelGGL(fft_stage_a_kernel<32>, dim3(16), dim3(64), 0, s, in, scratch, twiddle, lay);
        hipLaunchKernelGGL(fft_stage_bc_sync_kernel<32>, dim3(4), dim3(256), 0, s, scratch, out, twiddle, count, p, idx_symbol, buffered, guard, cp_out, sync,
                           h_small, h_flag, seq, loop);
    } else if (fft_size == 16384) {
        hipLaunchKernelGGL(fft_stage_a_kernel<16>, dim3(8), dim3(64), 0, s, in, scratch, twiddle, lay);
        hipLaunchKernelGGL(fft_stage_bc_sync_kernel<16>, dim3(4), dim3(128), 0, s, scratch, out, twiddle, count, p, idx_symbol, buffered, guard, cp_out, sync,
                           h_small, h_flag, seq, loop);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_sym_sync(const EqParams &p, const float2 *symbol, int idx_symbol, const float2 *buffered, int guard, float4 *cp_out,
                           float2 *sync, float *h_small, unsigned *h_flag, unsigned seq, hipStream_t s, T2DevLoop *loop)
{
    const int bytes = (p.max_seg + 2) * 16 + 2 * 256 * 8;
    if (bytes > 64 * 1024)
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(sym_sync_kernel), bytes)) return e;
    hipLaunchKernelGGL(sym_sync_kernel, dim3(1), dim3(256), bytes, s, p, symbol, idx_symbol, buffered, guard, cp_out, sync, h_small, h_flag, seq, loop);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void publish_symbol_kernel(const float2 *__restrict__ cells, int n_cells, const float *__restrict__ cp4,
                                                            const float *__restrict__ sync2, float2 *h_cells, float *h_small, unsigned *h_flag,
                                                            unsigned seq, unsigned *d_count)
{
    const int tid = threadIdx.x;
    const int pairs = n_cells >> 1;
    const float4 *src = reinterpret_cast<const float4 *>(cells);
    float4 *dst = reinterpret_cast<float4 *>(h_cells);
    for (int i = (int)blockIdx.x * 256 + tid; i < pairs; i += (int)gridDim.x * 256) dst[i] = src[i];
    if ((n_cells & 1) && blockIdx.x == 0 && tid == 0) h_cells[n_cells - 1] = cells[n_cells - 1];
    __threadfence_system();                                 // this lane's stores have reached the host before the workgroup is counted
    __syncthreads();
    if (tid == 0) {
        const unsigned before = __hip_atomic_fetch_add(d_count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (before == gridDim.x - 1) {                      // the last workgroup: everything is over there
            __hip_atomic_store(d_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cp4) { h_small[0] = cp4[0]; h_small[1] = cp4[1]; h_small[2] = cp4[2]; h_small[3] = cp4[3]; }
            if (sync2) { h_small[4] = sync2[0]; h_small[5] = sync2[1]; }
            __threadfence_system();
            __hip_atomic_store(h_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

hipError_t launch_publish_symbol(const float2 *cells, int n_cells, const float *cp4, const float *sync2, float2 *h_cells, float *h_small,
                                 unsigned *h_flag, unsigned seq, unsigned *d_count, hipStream_t s)
{
    int grid = (n_cells / 2 + 1023) / 1024;                 // four 16-byte stores per lane
    grid = grid < 1 ? 1 : (grid > 64 ? 64 : grid);
    hipLaunchKernelGGL(publish_symbol_kernel, dim3((unsigned)grid), dim3(256), 0, s, cells, n_cells, cp4, sync2, h_cells, h_small, h_flag, seq, d_count);
    return hipGetLastError();
}

hipError_t launch_eq_data(const EqParams &p, const float2 *symbols, const int32_t *symbol_index, int n_symbols, float2 *out,
                          float4 *pilot_scratch, float2 *sync, hipStream_t s)
{
    if (p.n_splits < 1) return hipErrorInvalidValue;
    {
        int range = 0;
        for (int k = 0; k < p.n_splits; ++k) range = std::max(range, eq_split_q(p.c_data, p.n_splits, k + 1) - eq_split_q(p.c_data, p.n_splits, k));
        const int bytes = (range + 1) * 8;
        const int threads = 512;                                          // (1024-lane workgroups, one per CU, measured slower: profiles/HISTORY.md)
        const int it = (range + threads - 1) / threads;
        hipError_t e = hipErrorInvalidValue;
#define T2_EQS(IT_, TH_) e = launch_eq_split<IT_, TH_>(p, symbols, symbol_index, n_symbols, out, pilot_scratch, bytes, s)
        if (it <= 8) T2_EQS(8, 512); else if (it <= 14) T2_EQS(14, 512); else if (it <= 18) T2_EQS(18, 512); else if (it <= 28) T2_EQS(28, 512);
#undef T2_EQS
        if (e != hipSuccess) return e;
    }
    if (sync) {
        const int sy_bytes = (p.max_seg + 1) * 16;
        if (sy_bytes > 64 * 1024)
            if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(eq_sync_kernel), sy_bytes)) return e;
        hipLaunchKernelGGL(eq_sync_kernel, dim3(n_symbols), dim3(64), sy_bytes, s, p, symbol_index, pilot_scratch, sync, n_symbols);
    }
    return hipGetLastError();
}

}  // namespace t2gpu
