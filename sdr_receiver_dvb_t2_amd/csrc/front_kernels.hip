// front_kernels.hip -- sample-rate front end of the reference on gfx950.
//
// Replaces the per-sample loop of dvbt2_demodulator::execute (/root/reference/src/DVB_T2/dvbt2_demodulator.cpp:145-254), the cubic
// Farrow resampler (/root/reference/src/DSP/interpolator_farrow.hh:41-68), the 64-tap /2 decimator
// (/root/reference/src/DSP/filter_decimator.h:72-131) and the guard-interval correlation of symbol_acquisition
// (dvbt2_demodulator.cpp:321-327). All of it is streaming HBM work: int16 IQ in, complex float out.
//
// What is sequential in the reference and how it is made parallel here:
//   * dc averager  out += 1e-6 * (in - out)  -- a linear recurrence; composed as (a, b) pairs (x -> a*x + b) in a three-level
//     scan (thread, workgroup, call). Evaluated in double; the reference's float rounding noise is not reproducible in any
//     parallel order, parity for this stage is by tolerance (stated in tests/test_front_gpu.py).
//   * NCO phase and Farrow position accumulators -- exact, through the run tables of front_plan.h.
//   * decimation phase, delay lines -- carried in the buffers' prefixes (3 de-rotated samples, 63 interpolated samples).
// Everything else is the reference's float arithmetic in its source order (no FMA contraction: -ffp-contract=off and the
// *_r helpers), so given the same de-rotated samples the Farrow and decimator outputs are bit-identical to the oracle.
#include "front_kernels.h"
#include "tables/dsp_tables_data.h"
#include "cp_device.h"
#include "ofdm_device.h"

#pragma clang fp contract(off)

#ifndef T2_FRONT_FUSED
#define T2_FRONT_FUSED 1
#endif

namespace {

__device__ __forceinline__ float mul_r(float a, float b) { return a * b; }
__device__ __forceinline__ float add_r(float a, float b) { return a + b; }
__device__ __forceinline__ float sub_r(float a, float b) { return a - b; }

constexpr double DC_ALPHA = (double)1.0e-6f;                 // dc_ratio, dvbt2_demodulator.h:94
constexpr float PI_F = 3.14159274101257324219f;
constexpr float PI_X_2 = PI_F * 2.0f;
constexpr float K_TABLE = 32767.0f / (2.0f * PI_F);          // fast_math.h:29

struct Lin { double a, re, im; };                            // x -> a*x + (re, im)
__device__ __forceinline__ Lin compose(const Lin &l, const Lin &r) { return Lin{l.a * r.a, r.a * l.re + r.re, r.a * l.im + r.im}; }

__device__ __forceinline__ Lin shfl_up_lin(const Lin &v, int d)
{
    return Lin{__shfl_up(v.a, d, 64), __shfl_up(v.re, d, 64), __shfl_up(v.im, d, 64)};
}
// Inclusive scan of `v` (composition in thread order) over the 256 threads of a workgroup: wavefront scan by shuffles, the four
// wavefront totals through 96 bytes of LDS. Returns the EXCLUSIVE prefix of this thread; *total receives the workgroup's
// composition. (Round 1 kept every front-end kernel under 12 KB of LDS so that it could run beside the LDPC decoder's persistent
// workgroups of another stream, receiver.pipeline_step with T2GPU_PIPE_SERIAL=0. That schedule measured slower than the serial one
// -- a decoder workgroup slowed by a neighbour stalls its whole SIMD batch -- and is not supported by these kernels any more: the
// de-rotation kernel stages its outputs in 35 KB, the Farrow + decimator kernel its window in 19 KB. Serial is the default.)
template <int NW = 4>
__device__ __forceinline__ Lin block_scan_exclusive(Lin v, Lin *wave_tot /* LDS[NW] */, Lin *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const Lin o = shfl_up_lin(v, d);
        if (lane >= d) v = compose(o, v);
    }
    if (lane == 63) wave_tot[wave] = v;
    __syncthreads();
    Lin ex = shfl_up_lin(v, 1);
    if (lane == 0) ex = Lin{1.0, 0.0, 0.0};
    Lin pre{1.0, 0.0, 0.0};
    for (int w = 0; w < wave; ++w) pre = compose(pre, wave_tot[w]);
    if (total) { Lin t{1.0, 0.0, 0.0}; for (int w = 0; w < NW; ++w) t = compose(t, wave_tot[w]); *total = t; }
    return compose(pre, ex);
}

// a lane's FRONT_PER consecutive samples (four at a time as two 8-byte reads where the layout allows)
__device__ __forceinline__ void load_samples(const FrontParams &p, long s0, float (&xr)[FRONT_PER], float (&xi)[FRONT_PER], int &valid)
{
    valid = (int)min((long)FRONT_PER, max(0L, (long)p.n - s0));
    if (p.stride == 1 && valid == FRONT_PER && ((((uintptr_t)p.i_in | (uintptr_t)p.q_in) & 7) == 0)) {
#pragma unroll
        for (int g = 0; g < FRONT_PER / 4; ++g) {
            const short4 vi = *reinterpret_cast<const short4 *>(p.i_in + s0 + 4 * g);
            const short4 vq = *reinterpret_cast<const short4 *>(p.q_in + s0 + 4 * g);
            xr[4 * g] = (float)vi.x * p.short_to_float; xr[4 * g + 1] = (float)vi.y * p.short_to_float;
            xr[4 * g + 2] = (float)vi.z * p.short_to_float; xr[4 * g + 3] = (float)vi.w * p.short_to_float;
            xi[4 * g] = (float)vq.x * p.short_to_float; xi[4 * g + 1] = (float)vq.y * p.short_to_float;
            xi[4 * g + 2] = (float)vq.z * p.short_to_float; xi[4 * g + 3] = (float)vq.w * p.short_to_float;
        }
    } else {
        for (int k = 0; k < FRONT_PER; ++k) {
            const long j = (s0 + k) * p.stride;                                  // dvbt2_demodulator.cpp:176-178
            xr[k] = k < valid ? (float)p.i_in[j] * p.short_to_float : 0.0f;
            xi[k] = k < valid ? (float)p.q_in[j] * p.short_to_float : 0.0f;
        }
    }
}

__device__ __forceinline__ Lin thread_lin(const float (&xr)[FRONT_PER], const float (&xi)[FRONT_PER], int valid)
{
    Lin l{1.0, 0.0, 0.0};
    for (int k = 0; k < FRONT_PER; ++k)
        if (k < valid) { l.a *= (1.0 - DC_ALPHA); l.re = (1.0 - DC_ALPHA) * l.re + DC_ALPHA * (double)xr[k]; l.im = (1.0 - DC_ALPHA) * l.im + DC_ALPHA * (double)xi[k]; }
    return l;
}

// Where block b's aggregate (and later the averager value before block b) lives in p.blk: the level-2 scan is ONE workgroup whose
// lane t owns the contiguous blocks [t * per, t * per + per); stored in block order, a load instruction of that kernel touched 64
// lines 9 KB apart and the kernel ran at the address rate of one CU (190 us per 73 600 blocks). Stored lane-interleaved -- block
// t * per + k at slot k * 256 + t -- the same loads are contiguous across the wavefront.
#ifndef T2_DC_LANES
#define T2_DC_LANES 512
#endif
constexpr int DC_LANES = T2_DC_LANES;      // lanes of the level-2 scan's one workgroup
__device__ __forceinline__ long dc_slot(int b, int per) { return (long)(b % per) * DC_LANES + b / per; }
__device__ __forceinline__ int dc_per(int n_blocks) { return (n_blocks + DC_LANES - 1) / DC_LANES; }

// ---- level 1: per-workgroup aggregate of the dc recurrence
__device__ __forceinline__ void front_dc_block_body(const FrontParams &p, const int bid)
{
    __shared__ Lin sh[256];
    const int tid = threadIdx.x;
    float xr[FRONT_PER], xi[FRONT_PER]; int valid;
    load_samples(p, (long)bid * FRONT_BLOCK + tid * FRONT_PER, xr, xi, valid);
    sh[tid] = thread_lin(xr, xi, valid);
    __syncthreads();
    for (int s = 1; s < 256; s <<= 1) {
        if ((tid & (2 * s - 1)) == 0) sh[tid] = compose(sh[tid], sh[tid + s]);
        __syncthreads();
    }
    if (tid == 0) { double *o = p.blk + 4 * dc_slot(bid, dc_per(p.n_blocks)); o[0] = sh[0].a; o[1] = sh[0].re; o[2] = sh[0].im; }
}
__global__ __launch_bounds__(256) void front_dc_block_kernel(FrontParams p) { front_dc_block_body(p, (int)blockIdx.x); }

// ---- level 2: scan over the workgroup aggregates (one workgroup); blk[b] becomes the averager value BEFORE block b
__global__ __launch_bounds__(DC_LANES) void front_dc_scan_kernel(FrontParams p)
{
    __shared__ Lin wave_tot[DC_LANES / 64];
    const int tid = threadIdx.x, nb = p.n_blocks;
    const int per = dc_per(nb), b0 = min(nb, tid * per), b1 = min(nb, b0 + per);
    // several loads in flight per lane instead of one (same operations in the same order); 1024 lanes leave 64 VGPRs per lane
    constexpr int U = DC_LANES > 512 ? 3 : 8;
    Lin l{1.0, 0.0, 0.0};
    for (int b = b0; b < b1; b += U) {
        Lin w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const double4 v = *reinterpret_cast<const double4 *>(p.blk + 4 * dc_slot(min(b + u, b1 - 1), per)); w[u] = Lin{v.x, v.y, v.z}; }
#pragma unroll
        for (int u = 0; u < U; ++u) if (b + u < b1) l = compose(l, w[u]);
    }
    Lin total;
    const Lin ex = block_scan_exclusive<DC_LANES / 64>(l, wave_tot, &total);
    const double s_re = p.state->dc_re, s_im = p.state->dc_im;
    double re = ex.a * s_re + ex.re, im = ex.a * s_im + ex.im;
    for (int b = b0; b < b1; b += U) {
        Lin w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const double4 v = *reinterpret_cast<const double4 *>(p.blk + 4 * dc_slot(min(b + u, b1 - 1), per)); w[u] = Lin{v.x, v.y, v.z}; }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (b + u < b1) {
                *reinterpret_cast<double2 *>(p.blk + 4 * dc_slot(b + u, per)) = make_double2(re, im);
                re = w[u].a * re + w[u].re; im = w[u].a * im + w[u].im;
            }
    }
    __syncthreads();
    if (tid == 0) { p.state->dc_re = total.a * s_re + total.re; p.state->dc_im = total.a * s_im + total.im; }
}

__device__ __forceinline__ float wrap_2pi(float a)
{
    while (a > PI_X_2) a -= PI_X_2;
    while (a < -PI_X_2) a += PI_X_2;
    return a;
}

// ---- level 3: dc removal, sign statistics, IQ-imbalance correction, NCO de-rotation (dvbt2_demodulator.cpp:175-205)
// last run whose first sample is <= i (runs are sorted by i0, the first starts at the call's first sample). The planner emits
// one run for whole buffers when the loops stand still and never more than one per sample; a search costs log2(runs) cached loads.
__device__ __forceinline__ int find_run(const FrontRun *__restrict__ runs, int n, long i)
{
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (runs[mid].i0 <= i) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ void front_derotate_body(const FrontParams &p, const int bid, const double *start /* averager value before this block: re, im */)
{
    __shared__ Lin wave_tot[4];
    __shared__ double red[3][4];
    // a lane's FRONT_PER results are FRONT_PER x 8 bytes apart from its neighbour's: stored from the lanes they were 32 lines per
    // store instruction (the kernel doubled its time at 8 samples per lane); through LDS (rows padded by one cell) they leave as
    // whole 512-byte runs per wavefront
    __shared__ float2 sh_out[256 * (FRONT_PER + 1)];
    const int tid = threadIdx.x;
    const long s0 = (long)bid * FRONT_BLOCK + tid * FRONT_PER;
    float xr[FRONT_PER], xi[FRONT_PER]; int valid;
    load_samples(p, s0, xr, xi, valid);
    const Lin ex = block_scan_exclusive(thread_lin(xr, xi, valid), wave_tot, nullptr);
    double dre = ex.a * start[0] + ex.re, dim = ex.a * start[1] + ex.im;
    const float c1 = p.state->c1, c2 = p.state->c2;
    double t1 = 0.0, t2 = 0.0, t3 = 0.0;
    int r = valid ? find_run(p.nco_runs, p.n_nco_runs, s0) : 0;
    // (one pass per sample. Round 4 split it in two -- everything up to the table index, then the 2 x FRONT_PER table reads together --
    // for the one-launch chain of the slot-shaped path, where it bought nothing: an in-kernel clock showed the phase's time is the
    // 16-step chain of double-precision operations per lane, not the reads; on 48-frame calls the arrays it kept in registers cost
    // 357 -> 421 us. Back to the round-3 loop.)
    for (int k = 0; k < valid; ++k) {
        const long i = s0 + k;
        dre = dre + DC_ALPHA * ((double)xr[k] - dre);                           // exponential_averager, loop_filters.hh:63-67
        dim = dim + DC_ALPHA * ((double)xi[k] - dim);
        float real = sub_r(xr[k], (float)dre), imag = sub_r(xi[k], (float)dim);
        float sgn = real < 0 ? -1.0f : 1.0f;                                    // est_1_bit_quantization, :256-265
        t1 -= (double)mul_r(imag, sgn);
        t2 += (double)mul_r(real, sgn);
        sgn = imag < 0 ? -1.0f : 1.0f;
        t3 += (double)mul_r(imag, sgn);
        real = mul_r(real, c2);                                                 // :184-185
        imag = add_r(imag, mul_r(c1, real));
        while (r + 1 < p.n_nco_runs && p.nco_runs[r + 1].i0 <= i) ++r;
        const FrontRun run = p.nco_runs[r];
        const float fnco = (float)(run.base + (double)(i - run.i0) * run.step); // frequency_nco for this sample (exact)
        const float off = wrap_2pi(sub_r(fnco, run.aux));                       // :194-200
        const int li = (int)(off * K_TABLE + 32767) & 65535;                    // fast_math.h:47-58
        const float nr = p.lut_cos[li], ni = p.lut_sin[li];
        sh_out[tid * (FRONT_PER + 1) + k] = make_float2(sub_r(mul_r(real, nr), mul_r(imag, ni)), add_r(mul_r(imag, nr), mul_r(real, ni)));
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { t1 += __shfl_down(t1, d, 64); t2 += __shfl_down(t2, d, 64); t3 += __shfl_down(t3, d, 64); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = t1; red[1][tid >> 6] = t2; red[2][tid >> 6] = t3; }
    __syncthreads();
    if (tid == 0) {
        double *o = p.theta_part + 4 * (long)bid;
        for (int c = 0; c < 3; ++c) o[c] = (red[c][0] + red[c][1]) + (red[c][2] + red[c][3]);
    }
    const long b0 = (long)bid * FRONT_BLOCK;
#pragma unroll
    for (int g = 0; g < FRONT_PER; ++g) {
        const int e = g * 256 + tid;
        if (b0 + e < p.n) p.derot[3 + b0 + e] = sh_out[(e / FRONT_PER) * (FRONT_PER + 1) + e % FRONT_PER];
    }
}
__global__ __launch_bounds__(256) void front_derotate_kernel(FrontParams p)
{
    front_derotate_body(p, (int)blockIdx.x, p.blk + 4 * dc_slot((int)blockIdx.x, dc_per(p.n_blocks)));
}

// ---- Farrow resampler: one lane per input sample (interpolator_farrow.hh:47-66); positions from the run table
__global__ __launch_bounds__(256) void front_farrow_kernel(FrontParams p)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= p.n) return;
    const FrontRun run = p.far_runs[find_run(p.far_runs, p.n_far_runs, i)];
    const long k = i - run.i0;
    float x1 = (float)(run.base + (double)k * run.step);
    long o = (long)run.o0 + k * run.cnt;
    const float delay_x = run.aux;
    const float2 in = p.derot[3 + i], d1 = p.derot[2 + i], d2 = p.derot[1 + i], d3 = p.derot[i];
    float a0[2], a1[2], a2[2], a3[2];
    const float vin[2] = {in.x, in.y}, v1[2] = {d1.x, d1.y}, v2[2] = {d2.x, d2.y}, v3[2] = {d3.x, d3.y};
    for (int c = 0; c < 2; ++c) {
        const float even1 = add_r(v3[c], vin[c]), even2 = add_r(v2[c], v1[c]);
        const float odd1 = sub_r(v3[c], vin[c]), odd2 = sub_r(v2[c], v1[c]);
        a0[c] = sub_r(mul_r(9.0f / 16.0f, even2), mul_r(1.0f / 16.0f, even1));
        a1[c] = sub_r(mul_r(1.0f / 8.0f, odd1), mul_r(11.0f / 8.0f, odd2));
        a2[c] = mul_r(1.0f / 4.0f, sub_r(even1, even2));
        a3[c] = sub_r(mul_r(3.0f / 2.0f, odd2), mul_r(1.0f / 2.0f, odd1));
    }
    float2 *out = p.interp + 63;
    while (x1 < 0.5f) {
        const float x2 = mul_r(x1, x1), x3 = mul_r(x2, x1);
        float v[2];
        for (int c = 0; c < 2; ++c) v[c] = add_r(add_r(add_r(mul_r(a3[c], x3), mul_r(a2[c], x2)), mul_r(a1[c], x1)), a0[c]);
        out[o++] = make_float2(v[0], v[1]);
        x1 = add_r(x1, delay_x);
    }
}

// ---- /2 decimator: one lane per output cell; the window and the summation order of filter_decimator.h:83-115
__constant__ float c_taps[64];

__global__ __launch_bounds__(256) void front_decimate_kernel(FrontParams p)
{
    __shared__ float2 w[2 * 256 + 64];
    const long k0 = (long)blockIdx.x * 256;
    const long m0 = 2 * k0 + (1 - p.decim_phase);            // buffer index (63-cell prefix included) of the oldest cell of output k0
    const long avail = 63 + p.n_interp;
    for (int t = threadIdx.x; t < 2 * 256 + 62; t += 256) w[t] = m0 + t < avail ? p.interp[m0 + t] : make_float2(0.f, 0.f);
    __syncthreads();
    const long k = k0 + threadIdx.x;
    if (k >= p.n_out) return;
    const float2 *x = w + 2 * threadIdx.x;
    float lane_r[4], lane_i[4];
    for (int q = 0; q < 4; ++q) {
        float ar = 0.0f, ai = 0.0f;
        for (int blk = 0; blk < 4; ++blk) {
            const int c = 16 * blk + q;
            const float2 x0 = x[c], x1 = x[c + 4], x2 = x[c + 8], x3 = x[c + 12];
            const float h0 = c_taps[c], h1 = c_taps[c + 4], h2 = c_taps[c + 8], h3 = c_taps[c + 12];
            ar = add_r(ar, add_r(add_r(mul_r(x0.x, h0), mul_r(x1.x, h1)), add_r(mul_r(x2.x, h2), mul_r(x3.x, h3))));
            ai = add_r(ai, add_r(add_r(mul_r(x0.y, h0), mul_r(x1.y, h1)), add_r(mul_r(x2.y, h2), mul_r(x3.y, h3))));
        }
        lane_r[q] = ar; lane_i[q] = ai;
    }
    p.out[k] = make_float2(add_r(add_r(add_r(lane_r[0], lane_r[1]), lane_r[2]), lane_r[3]),
                           add_r(add_r(add_r(lane_i[0], lane_i[1]), lane_i[2]), lane_i[3]));
}

// ---- Farrow + /2 decimator in one pass: the resampled stream (twice the input rate, 8 B per cell written and read again by the
// two kernels above: half of the front end's HBM traffic) never leaves the CU. A workgroup needs cells m0 .. m0 + 573 of the
// resampled stream for its 256 outputs; its lanes take the input samples that own those cells (from the owner of the first to the owner of the
// last: about 290) and run front_farrow_kernel's arithmetic on each, storing into LDS. Same values bit
// for bit (tests/test_front_gpu.py runs both forms). The last 63 cells still go to p.interp: front_finish_kernel carries them over.
__device__ __forceinline__ int find_run_out(const FrontRun *__restrict__ runs, int n, long o)
{
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (runs[mid].o0 <= o) lo = mid; else hi = mid - 1;
    }
    return lo;
}

#ifndef T2_FD_OUT
#define T2_FD_OUT 1024
#endif
#ifndef T2_FD_WAVES
#define T2_FD_WAVES 5
#endif
// outputs per workgroup: FD_OUT need about FD_OUT + 34 input samples, so the Farrow passes of 256 lanes end with one mostly empty pass
// whatever FD_OUT is -- the larger, the less that costs. Measured with the one-run path (tools/front_shape_probe.sh, us per 75 M
// samples, round 2's one-output-per-lane decimator): 256 -> 840, 512 -> 795, 768 -> 738, 1024 -> 701, 1536 -> 915, 2048 -> 848. With four
// consecutive outputs per lane (below) the kernel holds 22 cells + 32 accumulators: 82 VGPRs, five wavefronts per SIMD, 19 KB of LDS.
constexpr int FD_THREADS = 256, FD_OUT = T2_FD_OUT;
// The decimator stage: a lane computes FD_R = FD_OUT / FD_THREADS CONSECUTIVE outputs, whose 64-cell windows overlap in all but two
// cells per step: per block of 16 taps it reads 16 + 2 (FD_R - 1) cells once and uses each for every output it belongs to -- 22 LDS
// reads per 4 outputs and tap block instead of 64 (with one output per lane and pass the kernel spent its time on ds_read_b64: 64
// reads of 8 bytes per output; the arithmetic is 252 float operations per output either way, in the reference's order). A lane's
// window starts 2 FD_R cells after its neighbour's: the window is stored with one pad cell after every 8 (index t -> t + t / 8), so
// that the 64 lanes of a ds_read_b64 fall on 32 distinct bank pairs, two lanes each -- the two-cycle minimum of a 512-byte read.
constexpr int FD_R = FD_OUT / FD_THREADS;
static_assert(FD_R == 4 && FD_OUT % FD_THREADS == 0, "the decimator stage is written for four consecutive outputs per lane");
__device__ __forceinline__ int fd_pad(int t) { return t + (t >> 3); }
constexpr int FD_W = 2 * FD_OUT + 62;                                           // cells of a workgroup's window
__device__ __forceinline__ void front_farrow_decimate_body(const FrontParams &p, const int bid)
{
    __shared__ float2 w[FD_W + FD_W / 8 + 8];
    const long k0 = (long)bid * FD_OUT;
    const long m0 = 2 * k0 + (1 - p.decim_phase);
    const long avail = 63 + p.n_interp;
    constexpr int W = 2 * FD_OUT + 62;
    for (int t = threadIdx.x; t < W; t += FD_THREADS) w[fd_pad(t)] = m0 + t < 63 ? p.interp[m0 + t] : make_float2(0.f, 0.f);   // carried cells; zeros behind the end
    // input samples whose outputs fall into the window: from the owner of its first cell to the owner of its last
    const long oA = m0 > 63 ? m0 - 63 : 0, oB = (m0 + W - 1 < avail ? m0 + W - 1 : avail - 1) - 63;
    __syncthreads();
    const int run_a = oA <= oB ? find_run_out(p.far_runs, p.n_far_runs, oA) : 0, run_b = oA <= oB ? find_run_out(p.far_runs, p.n_far_runs, oB) : 0;
    if (oA <= oB && run_a == run_b) {
        // The whole window inside one run (runs are up to 2^24 samples long while the loops stand still): position, output index and
        // the outputs-per-sample count come from that one record through scalar registers, in 32-bit arithmetic relative to the
        // window; the reference's `while (x1 < 0.5f)` makes exactly cnt trips for every sample of a run (front_plan.cpp). Same values.
        const FrontRun run = p.far_runs[run_a];
        const int cnt = run.cnt;
        const long i_first = (long)run.i0 + (uint32_t)(oA - run.o0) / (uint32_t)cnt, i_last = (long)run.i0 + (uint32_t)(oB - run.o0) / (uint32_t)cnt;
        const int nin = (int)(i_last - i_first) + 1, kf = (int)(i_first - run.i0);
        const uint32_t tb = (uint32_t)(63 + (long)run.o0 - m0);                 // window index of output o: 63 + o - m0 = tb + k * cnt (mod 2^32)
        const long tail = (long)p.n_interp - m0;                                // window index from which cells are also the next call's prefix
        const int tail_t = tail > W + 8 ? 0x7fffffff : (tail < -(1 << 30) ? -(1 << 30) : (int)tail);
        const float2 *src = p.derot + i_first;
        float2 *keep = p.interp + m0;
        const float delay_x = run.aux;
        for (int il = threadIdx.x; il < nin; il += FD_THREADS) {
            const float2 in = src[3 + il], d1 = src[2 + il], d2 = src[1 + il], d3 = src[il];
            const int k = kf + il;
            float x1 = (float)(run.base + (double)k * run.step);
            int t = (int)(tb + (uint32_t)k * (uint32_t)cnt);
            float a0[2], a1[2], a2[2], a3[2];
            const float vin[2] = {in.x, in.y}, v1[2] = {d1.x, d1.y}, v2[2] = {d2.x, d2.y}, v3[2] = {d3.x, d3.y};
            for (int c = 0; c < 2; ++c) {
                const float even1 = add_r(v3[c], vin[c]), even2 = add_r(v2[c], v1[c]);
                const float odd1 = sub_r(v3[c], vin[c]), odd2 = sub_r(v2[c], v1[c]);
                a0[c] = sub_r(mul_r(9.0f / 16.0f, even2), mul_r(1.0f / 16.0f, even1));
                a1[c] = sub_r(mul_r(1.0f / 8.0f, odd1), mul_r(11.0f / 8.0f, odd2));
                a2[c] = mul_r(1.0f / 4.0f, sub_r(even1, even2));
                a3[c] = sub_r(mul_r(3.0f / 2.0f, odd2), mul_r(1.0f / 2.0f, odd1));
            }
            for (int c = 0; c < cnt; ++c) {
                const float x2 = mul_r(x1, x1), x3 = mul_r(x2, x1);
                float v[2];
                for (int q = 0; q < 2; ++q) v[q] = add_r(add_r(add_r(mul_r(a3[q], x3), mul_r(a2[q], x2)), mul_r(a1[q], x1)), a0[q]);
                if (t >= 0 && t < W) w[fd_pad(t)] = make_float2(v[0], v[1]);
                if (t >= tail_t) keep[t] = make_float2(v[0], v[1]);             // the tail the next call starts from
                ++t;
                x1 = add_r(x1, delay_x);
            }
        }
    } else if (oA <= oB) {
        const FrontRun ra = p.far_runs[run_a], rb = p.far_runs[run_b];
        const long i_first = (long)ra.i0 + (uint32_t)(oA - ra.o0) / (uint32_t)ra.cnt, i_last = (long)rb.i0 + (uint32_t)(oB - rb.o0) / (uint32_t)rb.cnt;
        for (long i = i_first + threadIdx.x; i <= i_last; i += FD_THREADS) {
            const FrontRun run = p.far_runs[find_run(p.far_runs, p.n_far_runs, i)];
            const long k = i - run.i0;
            float x1 = (float)(run.base + (double)k * run.step);
            long o = (long)run.o0 + k * run.cnt;
            const float delay_x = run.aux;
            const float2 in = p.derot[3 + i], d1 = p.derot[2 + i], d2 = p.derot[1 + i], d3 = p.derot[i];
            float a0[2], a1[2], a2[2], a3[2];
            const float vin[2] = {in.x, in.y}, v1[2] = {d1.x, d1.y}, v2[2] = {d2.x, d2.y}, v3[2] = {d3.x, d3.y};
            for (int c = 0; c < 2; ++c) {
                const float even1 = add_r(v3[c], vin[c]), even2 = add_r(v2[c], v1[c]);
                const float odd1 = sub_r(v3[c], vin[c]), odd2 = sub_r(v2[c], v1[c]);
                a0[c] = sub_r(mul_r(9.0f / 16.0f, even2), mul_r(1.0f / 16.0f, even1));
                a1[c] = sub_r(mul_r(1.0f / 8.0f, odd1), mul_r(11.0f / 8.0f, odd2));
                a2[c] = mul_r(1.0f / 4.0f, sub_r(even1, even2));
                a3[c] = sub_r(mul_r(3.0f / 2.0f, odd2), mul_r(1.0f / 2.0f, odd1));
            }
            while (x1 < 0.5f) {
                const float x2 = mul_r(x1, x1), x3 = mul_r(x2, x1);
                float v[2];
                for (int c = 0; c < 2; ++c) v[c] = add_r(add_r(add_r(mul_r(a3[c], x3), mul_r(a2[c], x2)), mul_r(a1[c], x1)), a0[c]);
                const long t = 63 + o - m0;
                if (t >= 0 && t < W) w[fd_pad((int)t)] = make_float2(v[0], v[1]);
                if (o >= p.n_interp - 63) p.interp[63 + o] = make_float2(v[0], v[1]);   // the tail the next call starts from
                ++o;
                x1 = add_r(x1, delay_x);
            }
        }
    }
    __syncthreads();
    {
        const int ko = FD_R * (int)threadIdx.x;                                 // first of the lane's outputs; its window starts at cell 2 ko
        const long k = k0 + ko;
        if (k >= p.n_out) return;
        const float2 *x = w + 2 * ko + (2 * ko >> 3);                           // fd_pad(2 ko + m) = fd_pad(2 ko) + m + m / 8: 2 ko is a multiple of 8
        float ar[FD_R][4], ai[FD_R][4];
#pragma unroll
        for (int r = 0; r < FD_R; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) { ar[r][q] = 0.0f; ai[r][q] = 0.0f; }
#pragma unroll 1
        for (int blk = 0; blk < 4; ++blk) {                                     // not unrolled: 16 taps in scalar registers and 22 cells at a time
            constexpr int NC = 16 + 2 * (FD_R - 1);
            float2 c[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) { const int m = 16 * blk + j; c[j] = x[m + (m >> 3)]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float h0 = c_taps[16 * blk + q], h1 = c_taps[16 * blk + q + 4], h2 = c_taps[16 * blk + q + 8], h3 = c_taps[16 * blk + q + 12];
#pragma unroll
                for (int r = 0; r < FD_R; ++r) {                                // output ko + r: cells shifted by 2 r (filter_decimator.h:83-115)
                    const float2 x0 = c[q + 2 * r], x1 = c[q + 2 * r + 4], x2 = c[q + 2 * r + 8], x3 = c[q + 2 * r + 12];
                    ar[r][q] = add_r(ar[r][q], add_r(add_r(mul_r(x0.x, h0), mul_r(x1.x, h1)), add_r(mul_r(x2.x, h2), mul_r(x3.x, h3))));
                    ai[r][q] = add_r(ai[r][q], add_r(add_r(mul_r(x0.y, h0), mul_r(x1.y, h1)), add_r(mul_r(x2.y, h2), mul_r(x3.y, h3))));
                }
            }
        }
        float2 o[FD_R];
#pragma unroll
        for (int r = 0; r < FD_R; ++r)
            o[r] = make_float2(add_r(add_r(add_r(ar[r][0], ar[r][1]), ar[r][2]), ar[r][3]), add_r(add_r(add_r(ai[r][0], ai[r][1]), ai[r][2]), ai[r][3]));
        if (k + FD_R <= p.n_out && (((uintptr_t)(p.out + k)) & 15) == 0) {
            float4 *dst = reinterpret_cast<float4 *>(p.out + k);
            dst[0] = make_float4(o[0].x, o[0].y, o[1].x, o[1].y);
            dst[1] = make_float4(o[2].x, o[2].y, o[3].x, o[3].y);
        } else {
#pragma unroll
            for (int r = 0; r < FD_R; ++r) if (k + r < p.n_out) p.out[k + r] = o[r];
        }
    }
}
__global__ __launch_bounds__(FD_THREADS, T2_FD_WAVES) void front_farrow_decimate_kernel(FrontParams p) { front_farrow_decimate_body(p, (int)blockIdx.x); }

// c1, c2, level_detect from the sign statistics of `len` samples (dvbt2_demodulator.cpp:227-235)
__device__ __forceinline__ void front_iq_estimate(FrontState &s, double t1, double t2, double t3, float len)
{
    const float avg1 = (float)t1 / len, avg2 = (float)t2 / len, avg3 = (float)t3 / len;
    s.c1 = avg1 / avg2;
    const float c_temp = avg3 / avg2;
    s.c2 = sqrtf(sub_r(mul_r(c_temp, c_temp), mul_r(s.c1, s.c1)));
    s.level_detect = mul_r(avg2, avg3);
}

// ---- end of execute(): statistics -> c1, c2, level (:228-235); carry the delay lines and the decimation phase
__device__ __forceinline__ void front_finish_body(const FrontParams &p)
{
    __shared__ double red[3][256];
    const int tid = threadIdx.x;
    double t[3] = {0.0, 0.0, 0.0};
    const int nbk = (p.stages & FRONT_STAGE_DEROTATE) ? p.n_blocks : 0;
    for (int b = tid; b < nbk; b += 256 * 8) {                 // eight loads in flight per lane; the additions keep their order
        double v[8][3];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double *q = p.theta_part + 4 * (long)min(b + 256 * u, nbk - 1);
            v[u][0] = q[0]; v[u][1] = q[1]; v[u][2] = q[2];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) if (b + 256 * u < nbk) { t[0] += v[u][0]; t[1] += v[u][1]; t[2] += v[u][2]; }
    }
    for (int c = 0; c < 3; ++c) red[c][tid] = t[c];
    const bool carry_d = (p.stages & FRONT_STAGE_FARROW) && tid < 3;
    const bool carry_i = (p.stages & FRONT_STAGE_DECIMATE) && tid >= 64 && tid < 64 + 63;
    float2 keep = make_float2(0.f, 0.f);
    if (carry_d) keep = p.derot[p.n + tid];
    else if (carry_i) keep = p.interp[p.n_interp + (tid - 64)];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) for (int c = 0; c < 3; ++c) red[c][tid] += red[c][tid + s];
        __syncthreads();
    }
    if (carry_d) p.derot[tid] = keep;
    else if (carry_i) p.interp[tid - 64] = keep;
    if (tid == 0 && (p.stages & FRONT_STAGE_DECIMATE)) p.state->decim_phase = (int)((p.decim_phase + p.n_interp) & 1);
    if (tid == 0 && (p.stages & FRONT_STAGE_DEROTATE) && p.n > 0) {
        FrontState &s = *p.state;
        for (int c = 0; c < 3; ++c) s.theta[c] = red[c][0];
        if (p.stages & FRONT_STAGE_HOLD_IQ) {
            for (int c = 0; c < 3; ++c) s.theta_acc[c] += red[c][0];
            s.n_acc += (double)p.n;
        } else {
            front_iq_estimate(s, red[0][0], red[1][0], red[2][0], (float)p.n);
        }
    }
}
__global__ __launch_bounds__(256) void front_finish_kernel(FrontParams p) { front_finish_body(p); }

// ---- a SHORT call in ONE launch and ONE pass per workgroup (a symbol's worth of samples: the slot-shaped path hands the front end one
// chunk per OFDM symbol and waits for what follows from it, so this kernel's duration is on that path's critical path once per symbol).
// Round 4 ran the five kernels' bodies phase after phase with three barriers across the grid (36 us per 32K symbol: the barriers, a
// 16-step chain of double-precision operations per lane in the de-rotation, the de-rotated stream through memory). Here a workgroup
// takes F1_B = 1024 input samples (four per lane) and does everything for them: dc aggregate -> the averager's value at its first sample
// from the workgroups before it (their aggregates arrive through agent-scope flags: a look-back, no barrier; a workgroup only ever waits
// for LOWER-numbered ones, which were dispatched before it) -> de-rotation into LDS, with the F1_H samples before its own re-derived
// (they feed the Farrow / decimator windows that reach back across the workgroup's border) -> Farrow into the decimator's window in LDS
// -> the decimator's outputs that COMPLETE inside its samples. The last workgroup to finish sums the sign statistics and carries the
// state. NCO phases, Farrow positions and every Farrow / decimator operation are those of the five launches (the same functions on the
// same values); the dc averager's linear recurrence is composed in a different order (per 1024 samples here, per 4096 there), so the
// de-rotated samples agree with the five launches to the last bits of a double -- and both with the reference's float IIR to 2e-6
// (tests/test_front_gpu.py).
__device__ __forceinline__ Lin lin_of(const float *xr, const float *xi, int valid)
{
    Lin l{1.0, 0.0, 0.0};
    for (int k = 0; k < F1_PER; ++k)
        if (k < valid) { l.a *= (1.0 - DC_ALPHA); l.re = (1.0 - DC_ALPHA) * l.re + DC_ALPHA * (double)xr[k]; l.im = (1.0 - DC_ALPHA) * l.im + DC_ALPHA * (double)xi[k]; }
    return l;
}
__device__ __forceinline__ void load4(const FrontParams &p, long s, int valid, float *xr, float *xi)
{
    if (p.stride == 1 && valid == F1_PER && ((((uintptr_t)p.i_in | (uintptr_t)p.q_in) & 7) == 0)) {
        const short4 vi = *reinterpret_cast<const short4 *>(p.i_in + s), vq = *reinterpret_cast<const short4 *>(p.q_in + s);
        xr[0] = (float)vi.x * p.short_to_float; xr[1] = (float)vi.y * p.short_to_float; xr[2] = (float)vi.z * p.short_to_float; xr[3] = (float)vi.w * p.short_to_float;
        xi[0] = (float)vq.x * p.short_to_float; xi[1] = (float)vq.y * p.short_to_float; xi[2] = (float)vq.z * p.short_to_float; xi[3] = (float)vq.w * p.short_to_float;
    } else {
        for (int k = 0; k < F1_PER; ++k) {
            const long j = (s + k) * p.stride;
            xr[k] = k < valid ? (float)p.i_in[j] * p.short_to_float : 0.0f;
            xi[k] = k < valid ? (float)p.q_in[j] * p.short_to_float : 0.0f;
        }
    }
}
// front_derotate_body's loop for a lane's (up to) four consecutive samples from sample s on; (dre, dim): the averager's value before s
__device__ __forceinline__ void derot4(const FrontParams &p, float c1, float c2, long s, int valid, const float *xr, const float *xi, double dre, double dim,
                                       float2 *dst, double *t, float2 *pre_out)
{
    // (in three steps, so that the lane's table reads are in flight together: with four samples per lane the arrays stay in registers, and a
    // read per sample in the loop was four memory latencies in a row -- 3 of the kernel's 15 us)
    int r = valid ? find_run(p.nco_runs, p.n_nco_runs, s) : 0;
    float re_[F1_PER], im_[F1_PER];
    int li_[F1_PER];
#pragma unroll
    for (int k = 0; k < F1_PER; ++k) {
        re_[k] = 0.0f; im_[k] = 0.0f; li_[k] = 0;
        if (k >= valid) continue;
        const long i = s + k;
        dre = dre + DC_ALPHA * ((double)xr[k] - dre);                           // exponential_averager, loop_filters.hh:63-67
        dim = dim + DC_ALPHA * ((double)xi[k] - dim);
        float real = sub_r(xr[k], (float)dre), imag = sub_r(xi[k], (float)dim);
        if (t) {
            float sgn = real < 0 ? -1.0f : 1.0f;                                // est_1_bit_quantization, :256-265
            t[0] -= (double)mul_r(imag, sgn);
            t[1] += (double)mul_r(real, sgn);
            sgn = imag < 0 ? -1.0f : 1.0f;
            t[2] += (double)mul_r(imag, sgn);
        }
        real = mul_r(real, c2);                                                 // :184-185
        imag = add_r(imag, mul_r(c1, real));
        while (r + 1 < p.n_nco_runs && p.nco_runs[r + 1].i0 <= i) ++r;
        const FrontRun run = p.nco_runs[r];
        const float fnco = (float)(run.base + (double)(i - run.i0) * run.step); // frequency_nco for this sample (exact)
        const float off = wrap_2pi(sub_r(fnco, run.aux));                       // :194-200
        li_[k] = (int)(off * K_TABLE + 32767) & 65535;                          // fast_math.h:47-58
        re_[k] = real; im_[k] = imag;
    }
    float nr_[F1_PER], ni_[F1_PER];
#pragma unroll
    for (int k = 0; k < F1_PER; ++k) { nr_[k] = p.lut_cos[li_[k]]; ni_[k] = p.lut_sin[li_[k]]; }
#pragma unroll
    for (int k = 0; k < F1_PER; ++k) {
        if (k >= valid) continue;
        const long i = s + k;
        const float2 v = make_float2(sub_r(mul_r(re_[k], nr_[k]), mul_r(im_[k], ni_[k])), add_r(mul_r(im_[k], nr_[k]), mul_r(re_[k], ni_[k])));
        dst[k] = v;
        if (pre_out && i >= (long)p.n - 3) pre_out[i - ((long)p.n - 3)] = v;     // delay_data_3,2,1 of the next call
    }
}
// index of the first resampled cell input sample i produces (i == n: one past the call's last)
__device__ __forceinline__ long first_out_of(const FrontParams &p, long i)
{
    if (i >= p.n) return p.n_interp;
    const FrontRun run = p.far_runs[p.n_far_runs == 1 ? 0 : find_run(p.far_runs, p.n_far_runs, i)];
    return (long)run.o0 + (i - run.i0) * run.cnt;
}

// ---- t2_plan_nco (front_plan.cpp) on the device: the same walk with the same float / double operations, one lane. Any decomposition
// into exact runs gives the same sequence of float values; this one is the host's. Returns the number of runs, or -1 when `cap` is too small.
__device__ __forceinline__ float nco_next_dev(float v, float fe, bool *wrapped)
{
    float t = sub_r(v, fe);
    *wrapped = false;
    while (t > PI_X_2) { t = sub_r(t, PI_X_2); *wrapped = true; }
    while (t < -PI_X_2) { t = add_r(t, PI_X_2); *wrapped = true; }
    return t;
}
__device__ __forceinline__ int expo_dev(double v) { return v == 0.0 ? -(1 << 30) : ilogb(v); }
__device__ __forceinline__ long span_dev(double V, double D, double lo, double hi, long cap)
{
    if (!(V >= lo && V < hi)) return -1;
    if (D == 0.0) return cap;
    const double lim = D > 0.0 ? (hi - V) / D : (V - lo) / -D;
    long t = lim >= (double)cap ? cap : (long)lim;
    if (t > cap) t = cap;
    while (t > 0 && !(V + (double)t * D >= lo && V + (double)t * D < hi)) --t;
    return t;
}
__device__ __noinline__ int plan_nco_dev(float *acc, int n, float fe, float phase_nco, FrontRun *runs, int cap)
{
    constexpr int MAX_RUN = 1 << 24;
    int i = 0, nr = 0;
    float prev = *acc;
    bool w;
    while (i < n) {
        const float v0 = nco_next_dev(prev, fe, &w);
        long len = 1;
        double step = 0.0;
        bool w0;
        if (nco_next_dev(v0, fe, &w0) == v0 && !w0) {
            len = n - i;
        } else if (n - i >= 3 && v0 != 0.0f) {
            bool w1, w2;
            const float v1 = nco_next_dev(v0, fe, &w1), v2 = nco_next_dev(v1, fe, &w2);
            const double s1 = (double)v1 - (double)v0, s2 = (double)v2 - (double)v1;
            const int e = expo_dev(v0);
            const bool same = !w1 && !w2 && s1 == s2 && v1 != 0.0f && v2 != 0.0f && expo_dev(v1) == e && expo_dev(v2) == e &&
                              signbit(v0) == signbit(v1) && signbit(v0) == signbit(v2);
            if (same) {
                const double u = ldexp(1.0, e - 23);
                const double lo = ldexp(1.0, e) + u;
                double hi = ldexp(1.0, e + 1) - u;
                if (hi > (double)PI_X_2 - u) hi = (double)PI_X_2 - u;
                const double mag = fabs((double)v0);
                const double dmag = signbit(v0) ? -s1 : s1;
                long t = span_dev(mag, dmag, lo, hi, MAX_RUN);
                if (t > n - i - 1) t = n - i - 1;
                if (t >= 2) { len = t + 1; step = s1; }
            }
        }
        if (nr >= cap) return -1;
        runs[nr++] = FrontRun{i, 0, (double)v0, step, 0, phase_nco};
        prev = (float)((double)v0 + (double)(len - 1) * step);
        i += (int)len;
    }
    *acc = prev;
    return nr;
}

// (development) phase stamps of a launch: workgroup w, phase k, by one lane
#define T2_STAMP(a_, w_, k_) do { if ((a_).stamps && threadIdx.x == 0 && (w_) < 64) (a_).stamps[16 * (w_) + (k_)] = wall_clock64(); } while (0)

__device__ __forceinline__ void front_one_body(FrontOneArgs &a, const int b, const int nb)
{
    __shared__ FrontRun sh_runs[FRONT_CHAIN_RUNS], sh_nco[FRONT_CHAIN_RUNS];
    __shared__ Lin wave_tot[4];
    __shared__ Lin sh_a1;
    __shared__ int sh_nr, sh_nco_far;
    __shared__ double sh_rec[F1_MAX_GRID][4];                  // the aggregates of the workgroups before this one: a, re, im
    __shared__ double sh_v[8];                                 // S (re, im), M (re, im), c1, c2, the last workgroup's part of the new dc
    __shared__ double red[3][4];
    __shared__ int sh_last;
    __shared__ float2 D[F1_H + F1_B];                          // de-rotated samples s0 - F1_H .. s0 + F1_B - 1
    __shared__ float2 W[F1_WCAP + F1_WCAP / 8 + 8];            // the decimator's window, padded as front_farrow_decimate_body pads it
    const int tid = threadIdx.x;
    T2_STAMP(a, b, 0);
    FrontParams p = a.p;
    for (int t = tid; t < p.n_nco_runs + p.n_far_runs; t += 256) sh_runs[t] = a.runs[t];
    p.nco_runs = sh_runs; p.far_runs = sh_runs + p.n_nco_runs;
    const long s0 = (long)b * F1_B;
    // ---- own samples, their aggregate, the look-back
    float xr[F1_PER], xi[F1_PER];
    const long sl = s0 + (long)tid * F1_PER;
    const int valid = (int)min((long)F1_PER, max(0L, (long)p.n - sl));
    load4(p, sl, valid, xr, xi);
    Lin total;
    const Lin ex = block_scan_exclusive<4>(lin_of(xr, xi, valid), wave_tot, &total);
    if (tid == (F1_B - F1_H) / F1_PER) sh_a1 = ex;             // the aggregate of the samples before the last F1_H: what the next workgroup starts its halo from
    if (b == 0 && tid == 0) { sh_v[0] = p.state->dc_re; sh_v[1] = p.state->dc_im; sh_v[4] = (double)p.state->c1; sh_v[5] = (double)p.state->c2; }
    __syncthreads();
    T2_STAMP(a, b, 1);
    if (tid == 0) {
        double *rec = a.rec + 16 * (size_t)b;
        rec[0] = total.a; rec[1] = total.re; rec[2] = total.im; rec[3] = sh_a1.a; rec[4] = sh_a1.re; rec[5] = sh_a1.im;
        if (b == 0) { rec[6] = sh_v[0]; rec[7] = sh_v[1]; rec[8] = sh_v[4]; rec[9] = sh_v[5]; }   // the state as block 0 found it: nobody else reads it
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_store(a.flags + b, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    T2_STAMP(a, b, 2);
    if (a.loop && tid == 255) {
        // The loop on the device: this chunk's NCO runs from the loop values the last sym_sync_kernel left (dvbt2_demodulator.cpp:165-171,
        // 187-193). EVERY workgroup plans them for itself, by a lane that polls no flag, while the look-back below waits for the other
        // workgroups' aggregates (round 6; before, workgroup 0 planned ahead of its flag and all the others waited for it: ~5 us of every
        // symbol's chain). The plan is a function of four floats and the chunk's length: the same runs everywhere. Into LDS; a chunk of
        // more than FRONT_CHAIN_RUNS runs (a residual offset of hundreds of Hz) into the workgroup's own slice of loop_runs.
        const float phase_new = wrap_2pi(add_r(a.loop->phase_nco, a.loop->pe));
        float acc = a.loop->frequency_nco;
        int far = 0;
        int nr = plan_nco_dev(&acc, p.n, a.loop->fe, phase_new, sh_nco, FRONT_CHAIN_RUNS);
        if (nr < 0) {
            far = 1;
            acc = a.loop->frequency_nco;
            nr = plan_nco_dev(&acc, p.n, a.loop->fe, phase_new, a.loop_runs + (size_t)b * T2_LOOP_RUNS_CAP, T2_LOOP_RUNS_CAP);
            if (nr < 0) { nr = 0; __hip_atomic_store(a.error, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        }
        sh_nr = nr; sh_nco_far = far;
        if (b == 0) { double *rec = a.rec; rec[11] = (double)phase_new; rec[12] = (double)acc; }   // the accumulators' new values: the last workgroup leaves them in the loop state
    }
    if (b > 0) {
        if (tid < b) {
            long long t0 = 0;
            for (unsigned spins = 1; __hip_atomic_load(a.flags + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.seq; ++spins) {
                if ((spins & 0xfffu) == 0) {
                    const long long now = wall_clock64();
                    if (!t0) t0 = now;
                    else if (now - t0 > 200000000LL) { __hip_atomic_store(a.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }   // 2 s at 100 MHz
                }
                __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const double *rec = a.rec + 16 * (size_t)tid;
            sh_rec[tid][0] = rec[0]; sh_rec[tid][1] = rec[1]; sh_rec[tid][2] = rec[2];
            if (tid == b - 1) { sh_v[2] = rec[3]; sh_v[3] = rec[4]; sh_v[6] = rec[5]; }
            if (tid == 0) { sh_v[0] = rec[6]; sh_v[1] = rec[7]; sh_v[4] = rec[8]; sh_v[5] = rec[9]; }
        }
        __syncthreads();
        // the aggregates of the workgroups before this one composed by a scan over the lanes (workgroup t on lane t): lane b - 1 ends up
        // with everything before workgroup b - 1 and finishes the two values this workgroup starts from
        {
            Lin l{1.0, 0.0, 0.0};
            if (tid < b) l = Lin{sh_rec[tid][0], sh_rec[tid][1], sh_rec[tid][2]};
            const Lin exb = block_scan_exclusive<4>(l, wave_tot, nullptr);
            if (tid == b - 1) {
                const double re = exb.a * sh_v[0] + exb.re, im = exb.a * sh_v[1] + exb.im;      // the averager before workgroup b - 1's first sample
                const double a1 = sh_v[2], a1re = sh_v[3], a1im = sh_v[6];
                sh_v[2] = a1 * re + a1re; sh_v[3] = a1 * im + a1im;             // ... before sample s0 - F1_H
                sh_v[0] = l.a * re + l.re; sh_v[1] = l.a * im + l.im;           // ... before sample s0
            }
        }
        __syncthreads();
    }
    T2_STAMP(a, b, 3);
    if (a.loop) {
        __syncthreads();                                       // (the planning lane's runs and their count)
        p.n_nco_runs = sh_nr;
        p.nco_runs = sh_nco_far ? a.loop_runs + (size_t)b * T2_LOOP_RUNS_CAP : sh_nco;
    }
    const float c1 = (float)sh_v[4], c2 = (float)sh_v[5];
    T2_STAMP(a, b, 4);
    // ---- de-rotation: own samples (sign statistics, the next call's delay line), then the halo by the first half of wavefront 0
    double t[3] = {0.0, 0.0, 0.0};
    derot4(p, c1, c2, sl, valid, xr, xi, ex.a * sh_v[0] + ex.re, ex.a * sh_v[1] + ex.im, D + F1_H + tid * F1_PER, t, a.pre_out);
    if (tid < 64) {
        if (b > 0) {
            const long hs = s0 - F1_H + (long)tid * F1_PER;
            const int hv = tid < F1_H / F1_PER ? F1_PER : 0;
            float hr[F1_PER], hi[F1_PER];
            load4(p, hv ? hs : 0, hv, hr, hi);
            Lin v = lin_of(hr, hi, hv);
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const Lin o = shfl_up_lin(v, d);
                if (tid >= d) v = compose(o, v);
            }
            Lin e = shfl_up_lin(v, 1);
            if (tid == 0) e = Lin{1.0, 0.0, 0.0};
            if (hv) derot4(p, c1, c2, hs, hv, hr, hi, e.a * sh_v[2] + e.re, e.a * sh_v[3] + e.im, D + tid * F1_PER, nullptr, nullptr);
        } else if (tid < 3) {
            D[F1_H - 3 + tid] = p.derot[tid];                                   // the delay line the call before left
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { t[0] += __shfl_down(t[0], d, 64); t[1] += __shfl_down(t[1], d, 64); t[2] += __shfl_down(t[2], d, 64); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = t[0]; red[1][tid >> 6] = t[1]; red[2][tid >> 6] = t[2]; }
    // ---- the decimator outputs this workgroup completes, their window
    const long o_b = first_out_of(p, s0), o_e = first_out_of(p, min(s0 + (long)F1_B, (long)p.n));
    const long k_lo = (o_b + p.decim_phase) >> 1, k_hi = (o_e + p.decim_phase) >> 1;
    const int n_k = (int)(k_hi - k_lo);
    const long m0 = 2 * k_lo + (1 - p.decim_phase);                             // buffer index (63-cell prefix included) of the window's first cell
    for (int w = tid; w < 63 && m0 + w < 63; w += 256) W[fd_pad(w)] = p.interp[m0 + w];   // cells the call before left
    __syncthreads();
    T2_STAMP(a, b, 5);
    if (tid == 0) {
        double *o = p.theta_part + 4 * (long)b;
        for (int c = 0; c < 3; ++c) o[c] = (red[c][0] + red[c][1]) + (red[c][2] + red[c][3]);
    }
    // ---- Farrow (front_farrow_kernel's arithmetic) for the inputs from the owner of the window's first cell to the workgroup's last sample
    {
        const long oA = m0 > 63 ? m0 - 63 : 0;
        const long i_end = min(s0 + (long)F1_B, (long)p.n);
        long i_first = s0;
        if (oA < o_b) {
            const FrontRun ra = p.far_runs[p.n_far_runs == 1 ? 0 : find_run_out(p.far_runs, p.n_far_runs, oA)];
            i_first = (long)ra.i0 + (uint32_t)(oA - ra.o0) / (uint32_t)ra.cnt;
        }
        const long tail_o = (long)p.n_interp - 63;
        for (long i = i_first + tid; i < i_end; i += 256) {
            const FrontRun run = p.far_runs[p.n_far_runs == 1 ? 0 : find_run(p.far_runs, p.n_far_runs, i)];
            const long kk = i - run.i0;
            float x1 = (float)(run.base + (double)kk * run.step);
            long o = (long)run.o0 + kk * run.cnt;
            const float delay_x = run.aux;
            const float2 *dd = D + F1_H + (i - s0);
            const float2 in = dd[0], d1 = dd[-1], d2 = dd[-2], d3 = dd[-3];
            float a0[2], a1[2], a2[2], a3[2];
            const float vin[2] = {in.x, in.y}, v1[2] = {d1.x, d1.y}, v2[2] = {d2.x, d2.y}, v3[2] = {d3.x, d3.y};
            for (int c = 0; c < 2; ++c) {
                const float even1 = add_r(v3[c], vin[c]), even2 = add_r(v2[c], v1[c]);
                const float odd1 = sub_r(v3[c], vin[c]), odd2 = sub_r(v2[c], v1[c]);
                a0[c] = sub_r(mul_r(9.0f / 16.0f, even2), mul_r(1.0f / 16.0f, even1));
                a1[c] = sub_r(mul_r(1.0f / 8.0f, odd1), mul_r(11.0f / 8.0f, odd2));
                a2[c] = mul_r(1.0f / 4.0f, sub_r(even1, even2));
                a3[c] = sub_r(mul_r(3.0f / 2.0f, odd2), mul_r(1.0f / 2.0f, odd1));
            }
            while (x1 < 0.5f) {
                const float x2 = mul_r(x1, x1), x3 = mul_r(x2, x1);
                float v[2];
                for (int c = 0; c < 2; ++c) v[c] = add_r(add_r(add_r(mul_r(a3[c], x3), mul_r(a2[c], x2)), mul_r(a1[c], x1)), a0[c]);
                const long tw = 63 + o - m0;
                if (tw >= 0 && tw < F1_WCAP) W[fd_pad((int)tw)] = make_float2(v[0], v[1]);
                if (i >= s0 && o >= tail_o) a.pre_out[3 + (o - tail_o)] = make_float2(v[0], v[1]);   // the tail the next call starts from (its owner stores it)
                ++o;
                x1 = add_r(x1, delay_x);
            }
        }
    }
    __syncthreads();
    T2_STAMP(a, b, 6);
    // ---- /2 decimator: front_farrow_decimate_body's stage, four consecutive outputs per lane
    for (int g = tid; 4 * g < n_k; g += 256) {
        const int ko = FD_R * g;
        const long k = k_lo + ko;
        const float2 *x = W + 2 * ko + (2 * ko >> 3);
        float ar[FD_R][4], ai[FD_R][4];
#pragma unroll
        for (int r = 0; r < FD_R; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) { ar[r][q] = 0.0f; ai[r][q] = 0.0f; }
#pragma unroll 1
        for (int blk = 0; blk < 4; ++blk) {
            constexpr int NC = 16 + 2 * (FD_R - 1);
            float2 c[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) { const int m = 16 * blk + j; c[j] = x[m + (m >> 3)]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float h0 = c_taps[16 * blk + q], h1 = c_taps[16 * blk + q + 4], h2 = c_taps[16 * blk + q + 8], h3 = c_taps[16 * blk + q + 12];
#pragma unroll
                for (int r = 0; r < FD_R; ++r) {
                    const float2 x0 = c[q + 2 * r], x1 = c[q + 2 * r + 4], x2 = c[q + 2 * r + 8], x3 = c[q + 2 * r + 12];
                    ar[r][q] = add_r(ar[r][q], add_r(add_r(mul_r(x0.x, h0), mul_r(x1.x, h1)), add_r(mul_r(x2.x, h2), mul_r(x3.x, h3))));
                    ai[r][q] = add_r(ai[r][q], add_r(add_r(mul_r(x0.y, h0), mul_r(x1.y, h1)), add_r(mul_r(x2.y, h2), mul_r(x3.y, h3))));
                }
            }
        }
#pragma unroll
        for (int r = 0; r < FD_R; ++r)
            if (ko + r < n_k)
                p.out[k + r] = make_float2(add_r(add_r(add_r(ar[r][0], ar[r][1]), ar[r][2]), ar[r][3]), add_r(add_r(add_r(ai[r][0], ai[r][1]), ai[r][2]), ai[r][3]));
    }
    // ---- the last workgroup to get here finishes the call (front_finish_kernel's work)
    __syncthreads();
    T2_STAMP(a, b, 7);
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned long long before = __hip_atomic_fetch_add(a.done, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sh_last = before == a.done_target + (unsigned long long)nb - 1;
        if (sh_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    T2_STAMP(a, b, 8);
    if (!sh_last) return;
    // the delay lines: the last 3 de-rotated samples, the last 63 resampled cells -- from this call where it had that many, else moved up
    float2 keep = make_float2(0.f, 0.f);
    if (tid < 3) { const long i = (long)p.n - 3 + tid; keep = i >= 0 ? a.pre_out[tid] : p.derot[tid + p.n]; }
    else if (tid >= 64 && tid < 64 + 63) { const long o = (long)p.n_interp - 63 + (tid - 64); keep = o >= 0 ? a.pre_out[3 + (tid - 64)] : p.interp[(tid - 64) + p.n_interp]; }
    __syncthreads();
    if (tid < 3) p.derot[tid] = keep;
    else if (tid >= 64 && tid < 64 + 63) p.interp[tid - 64] = keep;
    // every workgroup's aggregate and sign statistics into LDS by a lane each (one round of memory latency, not one per workgroup), then
    // folded in workgroup order by one lane
    double *fin = reinterpret_cast<double *>(W);                               // [nb][6]: a, re, im, theta 1..3 (the window is done with)
    if (tid < nb) {
        const double *q = p.theta_part + 4 * (long)tid, *rec = a.rec + 16 * (size_t)tid;
        double *f = fin + 6 * tid;
        f[0] = rec[0]; f[1] = rec[1]; f[2] = rec[2]; f[3] = q[0]; f[4] = q[1]; f[5] = q[2];
    }
    if (tid == 0) { sh_v[0] = a.rec[6]; sh_v[1] = a.rec[7]; }
    __syncthreads();
    if (tid == 0) {
        FrontState &s = *p.state;
        double th[3] = {0.0, 0.0, 0.0};
        double re = sh_v[0], im = sh_v[1];
        for (int k = 0; k < nb; ++k) {
            const double *f = fin + 6 * k;
            th[0] += f[3]; th[1] += f[4]; th[2] += f[5];
            re = f[0] * re + f[1]; im = f[0] * im + f[2];
        }
        s.dc_re = re; s.dc_im = im;
        if (a.loop) { a.loop->phase_nco = (float)a.rec[11]; a.loop->frequency_nco = (float)a.rec[12]; }
        s.decim_phase = (int)((p.decim_phase + p.n_interp) & 1);
        for (int c = 0; c < 3; ++c) s.theta[c] = th[c];
        if (p.stages & FRONT_STAGE_HOLD_IQ) {
            for (int c = 0; c < 3; ++c) s.theta_acc[c] += th[c];
            s.n_acc += (double)p.n;
        } else {
            front_iq_estimate(s, th[0], th[1], th[2], (float)p.n);
        }
    }
}

__device__ __forceinline__ void front_copy_body(const int16_t *__restrict__ hi, const int16_t *__restrict__ hq, int16_t *__restrict__ di,
                                                int16_t *__restrict__ dq, size_t n, const size_t t, const size_t stride);
// (the workgroups behind the chunk's own bring the next chunk's I/Q over: FrontOneArgs::cp_*)
__device__ __forceinline__ void front_copy_ahead(const FrontOneArgs &a, int wg)
{
    front_copy_body(a.cp_si, a.cp_sq, a.cp_di, a.cp_dq, (size_t)a.cp_n, (size_t)wg * 256 + threadIdx.x, (size_t)a.cp_wgs * 256);
}
__global__ __launch_bounds__(256) void front_one_kernel(FrontOneArgs a)
{
    const int nb = (int)gridDim.x - a.cp_wgs, b = (int)blockIdx.x;
    if (b >= nb) { front_copy_ahead(a, b - nb); return; }
    front_one_body(a, b, nb);
}

// ---- a chunk's front end AND the transform + synchronisation floats of the 32K / 16K symbol it completes, in ONE launch (round 5): workgroups
// 0 .. nb_front - 1 are front_one_kernel's, the eight behind them fft_one_sync_kernel<T2>'s (ofdm_device.h) -- stage A's four wait until every
// front-end workgroup has counted itself done (its cells stored and released; a workgroup only ever waits for lower-numbered ones, which were
// dispatched before it), the other four wait for stage A as before. The per-symbol critical path of the slot-shaped path is then one launch
// and no launch gap; every value goes through the operations of the two launches in their order.
template <int T2>
__global__ __launch_bounds__(256) void front_fft_one_kernel(FrontOneArgs a, t2gpu::FftOneArgs f, int nb_front)
{
    __shared__ __attribute__((aligned(16))) float fft_lds[t2gpu::FFT_BC_LDS_FLOATS];
    const int b = (int)blockIdx.x;
    if (b < nb_front) { front_one_body(a, b, nb_front); return; }
    if (b >= nb_front + 8) { front_copy_ahead(a, b - nb_front - 8); return; }
    const int fb = b - nb_front;
    T2_STAMP(a, 48 + fb, 0);
    if (fb <= 4) {                                                // stage A's four, and the workgroup that forms the guard correlation beside them: the buffered symbol must be complete
        if (threadIdx.x == 0) {
            const unsigned long long want = a.done_target + (unsigned long long)nb_front;
            long long t0 = 0;
            for (unsigned spins = 1; __hip_atomic_load(a.done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want; ++spins) {
                if ((spins & 0xfffu) == 0) {
                    const long long now = wall_clock64();
                    if (!t0) t0 = now;
                    else if (now - t0 > 200000000LL) __builtin_trap();          // 2 s at 100 MHz: the front end never finished -- fail loudly
                }
                __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    // (a 16K transform's workgroups have 128 lanes: the upper two wavefronts of these leave here -- a workgroup's barriers count the
    // wavefronts that have not ended)
    T2_STAMP(a, 48 + fb, 1);
    if constexpr (T2 == 16) { if (threadIdx.x >= 128) return; }
    t2gpu::fft_one_sync_body<T2>(f.in, f.scratch, f.out, f.twiddle, f.count, f.p, f.idx_symbol, f.buffered, f.guard, f.cp_out, f.sync, f.h_small, f.h_flag, f.seq,
                                 f.loop, fb, fft_lds, a.stamps ? a.stamps + 16 * (48 + fb) : nullptr);
    T2_STAMP(a, 48 + fb, 9);
}

// ---- a buffer of int16 I and Q from PAGE-LOCKED host memory into the device's staging by a kernel of the caller's stream (the
// slot-shaped path: two copy-engine transfers per execute() cost 12 us each and, worse, ~20 + 9 + 30 us of hand-over between the stream's
// kernels and the copy engine on either side of them -- 85 us per 172 032-sample buffer, a whole OFDM symbol's worth)
__device__ __forceinline__ void front_copy_body(const int16_t *__restrict__ hi, const int16_t *__restrict__ hq, int16_t *__restrict__ di,
                                                int16_t *__restrict__ dq, size_t n, const size_t t, const size_t stride);
__global__ __launch_bounds__(256) void front_copy_in_kernel(const int16_t *__restrict__ hi, const int16_t *__restrict__ hq, int16_t *__restrict__ di,
                                                            int16_t *__restrict__ dq, size_t n)
{
    front_copy_body(hi, hq, di, dq, n, (size_t)blockIdx.x * blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x);
}
__device__ __forceinline__ void front_copy_body(const int16_t *__restrict__ hi, const int16_t *__restrict__ hq, int16_t *__restrict__ di,
                                                int16_t *__restrict__ dq, size_t n, const size_t t, const size_t stride)
{
    if (((((uintptr_t)hi | (uintptr_t)hq | (uintptr_t)di | (uintptr_t)dq) & 15) == 0)) {
        const size_t n8 = n / 8;
        const uint4 *a = reinterpret_cast<const uint4 *>(hi), *b = reinterpret_cast<const uint4 *>(hq);
        uint4 *da = reinterpret_cast<uint4 *>(di), *db = reinterpret_cast<uint4 *>(dq);
        for (size_t i = t; i < n8; i += stride) { const uint4 x = a[i], y = b[i]; da[i] = x; db[i] = y; }
        for (size_t i = 8 * n8 + t; i < n; i += stride) { di[i] = hi[i]; dq[i] = hq[i]; }
    } else {
        for (size_t i = t; i < n; i += stride) { di[i] = hi[i]; dq[i] = hq[i]; }
    }
}

// end of an execute() whose chunks ran with FRONT_STAGE_HOLD_IQ (:227-235)
__global__ void front_commit_iq_kernel(FrontState *state, FrontState *h_copy, unsigned *h_flag, unsigned seq, const int *error)
{
    FrontState &s = *state;
    if (s.n_acc > 0.0) front_iq_estimate(s, s.theta_acc[0], s.theta_acc[1], s.theta_acc[2], (float)s.n_acc);
    s.theta_acc[0] = s.theta_acc[1] = s.theta_acc[2] = 0.0;
    s.n_acc = 0.0;
    if (h_copy) {                                           // the state as it stands now, to page-locked host memory, the sequence word behind it
        *h_copy = s;
        h_copy->error_ = error ? *error : 0;                // a look-back wait of the one-launch form that gave up: reported with the state
        __threadfence_system();
        __hip_atomic_store(h_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}


__global__ __launch_bounds__(256) void cp_correlate_kernel(const float2 *sym, long first, long frame_stride, int per_frame, int fft_size,
                                                           int guard, float4 *out)
{
    __shared__ double red[2][256];
    const int i = blockIdx.x;
    const float2 *s = sym + first + (long)(i / per_frame) * frame_stride + (long)(i % per_frame) * (fft_size + guard);
    const float4 r = t2gpu::cp_correlate_body(s, fft_size, guard, red);
    if (threadIdx.x == 0) out[blockIdx.x] = r;
}

// ---- the loop state up and down by launches of the chain's own stream (a 64-byte copy through the copy engine stands in its queue behind
// whatever else is there -- the 13 MB of a TI block coming down: 250 us per T2 frame -- and costs a hand-over on either side in the stream)
__global__ void front_loop_set_kernel(T2DevLoop *d, T2DevLoop v) { *d = v; }
__global__ void front_loop_get_kernel(const T2DevLoop *d, T2DevLoop *h_out, unsigned *h_flag, unsigned seq)
{
    *h_out = *d;
    __threadfence_system();
    __hip_atomic_store(h_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

bool g_taps_loaded[16] = {};

}  // namespace

void launch_front_loop_set(T2DevLoop *d, const T2DevLoop &v, hipStream_t stream) { hipLaunchKernelGGL(front_loop_set_kernel, dim3(1), dim3(1), 0, stream, d, v); }
void launch_front_loop_get(const T2DevLoop *d, T2DevLoop *h_out, unsigned *h_flag, unsigned seq, hipStream_t stream)
{
    hipLaunchKernelGGL(front_loop_get_kernel, dim3(1), dim3(1), 0, stream, d, h_out, h_flag, seq);
}

void launch_front_copy_in(const int16_t *hi, const int16_t *hq, int16_t *di, int16_t *dq, size_t n, hipStream_t stream)
{
    int grid = (int)((n / 8 + 255) / 256);
    grid = grid < 1 ? 1 : (grid > 256 ? 256 : grid);
    hipLaunchKernelGGL(front_copy_in_kernel, dim3(grid), dim3(256), 0, stream, hi, hq, di, dq, n);
}

void launch_front_commit_iq(FrontState *state, FrontState *h_copy, unsigned *h_flag, unsigned seq, const int *error, hipStream_t stream)
{
    hipLaunchKernelGGL(front_commit_iq_kernel, dim3(1), dim3(1), 0, stream, state, h_copy, h_flag, seq, error);
}

static void load_taps()
{
    int dev = 0;
    hipGetDevice(&dev);
    if (dev >= 0 && dev < 16 && !g_taps_loaded[dev]) {
        hipMemcpyToSymbol(HIP_SYMBOL(c_taps), T2_DECIM_TAPS, sizeof(T2_DECIM_TAPS));
        g_taps_loaded[dev] = true;
    }
}

static long fd_blocks_of(const FrontParams &p)
{
    const long by_out = (p.n_out + FD_OUT - 1) / FD_OUT, by_cells = p.n_interp / (2 * FD_OUT) + 1;
    return by_out > by_cells ? by_out : by_cells;
}

// the grid of the one-launch form, or 0 when the call does not qualify: the whole chain, at most F1_MAX_GRID x F1_B samples, run tables
// that fit the kernel's arguments, and between one and two resampled cells per input sample (the window a workgroup keeps in LDS and the
// F1_H samples of halo are sized for that: ratios from 1/2 to ~1, the reference's devices)
int front_one_grid(const FrontParams &p, const FrontRun *far_runs, size_t n_nco_runs, size_t n_far_runs)
{
    const int all = FRONT_STAGE_DEROTATE | FRONT_STAGE_FARROW | FRONT_STAGE_DECIMATE;
    if (!T2_FRONT_FUSED || p.n <= 0 || (p.stages & all) != all || n_nco_runs + n_far_runs > (size_t)FRONT_CHAIN_RUNS || n_far_runs < 1) return 0;
    // (a ratio a hair under 1/2 -- the sample-rate tracker's first steps -- gives three cells for a sample now and then: the window has room
    // for 128 of those per call)
    for (size_t r = 0; r < n_far_runs; ++r) if (far_runs[r].cnt < 1 || far_runs[r].cnt > 3) return 0;
    if (p.n_interp > 2 * (long)p.n + 128) return 0;
    const long g = ((long)p.n + F1_B - 1) / F1_B;
    return g <= F1_MAX_GRID ? (int)g : 0;
}

void launch_front_one(FrontOneArgs &a, int grid, hipStream_t stream)
{
    load_taps();
    hipLaunchKernelGGL(front_one_kernel, dim3((unsigned)(grid + a.cp_wgs)), dim3(256), 0, stream, a);
}

void launch_front_fft_one(FrontOneArgs &a, int grid, const t2gpu::FftOneArgs &f, hipStream_t stream)
{
    load_taps();
    if (f.fft_size == 32768) hipLaunchKernelGGL(front_fft_one_kernel<32>, dim3((unsigned)(grid + 8 + a.cp_wgs)), dim3(256), 0, stream, a, f, grid);
    else hipLaunchKernelGGL(front_fft_one_kernel<16>, dim3((unsigned)(grid + 8 + a.cp_wgs)), dim3(256), 0, stream, a, f, grid);
}

void launch_front(const FrontParams &p, hipStream_t stream)
{
    load_taps();
    if (p.n > 0 && (p.stages & FRONT_STAGE_DEROTATE)) {
        hipLaunchKernelGGL(front_dc_block_kernel, dim3(p.n_blocks), dim3(256), 0, stream, p);
        hipLaunchKernelGGL(front_dc_scan_kernel, dim3(1), dim3(DC_LANES), 0, stream, p);
        hipLaunchKernelGGL(front_derotate_kernel, dim3(p.n_blocks), dim3(256), 0, stream, p);
    }
    const bool both = (p.stages & FRONT_STAGE_FARROW) && (p.stages & FRONT_STAGE_DECIMATE);
    if (T2_FRONT_FUSED && both && p.n > 0) {
        // one pass, nothing of the resampled stream in HBM but its last 63 cells; enough workgroups that their windows
        // also cover the cells behind the last complete output (they are part of what the next call starts from)
        hipLaunchKernelGGL(front_farrow_decimate_kernel, dim3((unsigned)fd_blocks_of(p)), dim3(FD_THREADS), 0, stream, p);
    } else {
        if (p.n > 0 && (p.stages & FRONT_STAGE_FARROW))
            hipLaunchKernelGGL(front_farrow_kernel, dim3((p.n + 255) / 256), dim3(256), 0, stream, p);
        if (p.n_out > 0 && (p.stages & FRONT_STAGE_DECIMATE)) hipLaunchKernelGGL(front_decimate_kernel, dim3((unsigned)((p.n_out + 255) / 256)), dim3(256), 0, stream, p);
    }
    hipLaunchKernelGGL(front_finish_kernel, dim3(1), dim3(256), 0, stream, p);
}

void launch_cp_correlate(const float2 *sym, long first, long frame_stride, int per_frame, int n_symbols, int fft_size, int guard,
                         float4 *out, hipStream_t stream)
{
    if (n_symbols > 0)
        hipLaunchKernelGGL(cp_correlate_kernel, dim3(n_symbols), dim3(256), 0, stream, sym, first, frame_stride, per_frame, fft_size, guard, out);
}
