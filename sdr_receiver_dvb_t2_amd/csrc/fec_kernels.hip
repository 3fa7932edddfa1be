// fec_kernels.hip -- FEC-side streaming kernels for gfx950: BB descrambler (BCH stub), LLR demapper + bit de-interleaver,
// time / cell de-interleaver with cyclic Q-delay removal. All HBM-bound gather/scatter work; no matrix cores involved.
#include "fec_kernels.h"

// The reference is built without FMA (-mavx2 only): every product and sum rounds on its own. HIP's mul_r/add_r
// are header-defined plain operators carrying the default "contract" flag, i.e. the compiler still fuses them; the
// helpers below are defined under contract(off) instead (and the Makefile adds -ffp-contract=off for this file).
#pragma clang fp contract(off)

namespace t2gpu {

__device__ __forceinline__ float mul_r(float a, float b) { return a * b; }
__device__ __forceinline__ float add_r(float a, float b) { return a + b; }
__device__ __forceinline__ float sub_r(float a, float b) { return a - b; }
__device__ __forceinline__ float div_r(float a, float b) { return a / b; }

// ---------------------------------------------------------------------------------------------------- K-descramble
// Replaces bch_decoder::execute (/root/reference/src/DVB_T2/bch_decoder.cpp:63-164): the reference does no BCH decoding
// (":136 // TODO BCH decode"); it drops the parity tail of each k_ldpc block and XORs the BB scrambling sequence.
__global__ __launch_bounds__(256) void bch_descramble_kernel(const uint8_t *__restrict__ bits, int n_frames, int k_ldpc,
                                                            int k_bch, const uint8_t *__restrict__ prbs, uint8_t *__restrict__ out)
{
    const int words = k_bch / 4;             // every k_bch and k_ldpc of DVB-T2 is a multiple of 8
    const int f = blockIdx.y;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(bits + (size_t)f * k_ldpc);
    const uint32_t *pr = reinterpret_cast<const uint32_t *>(prbs);
    uint32_t *dst = reinterpret_cast<uint32_t *>(out + (size_t)f * k_bch);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < words; i += gridDim.x * blockDim.x) dst[i] = src[i] ^ pr[i];
}

hipError_t launch_bch_descramble(const uint8_t *bits, int n_frames, int k_ldpc, int k_bch, const uint8_t *prbs, uint8_t *out,
                                 hipStream_t s)
{
    dim3 grid((k_bch / 4 + 255) / 256 > 16 ? 16 : (k_bch / 4 + 255) / 256, n_frames);
    hipLaunchKernelGGL(bch_descramble_kernel, grid, dim3(256), 0, s, bits, n_frames, k_ldpc, k_bch, prbs, out);
    return hipGetLastError();
}

// K-descramble-pack: the same XOR, with the bit -> byte packing of bb_de_header::execute (bb_de_header.cpp:84-448: every byte it
// emits is assembled MSB first from 8 bit-bytes) done where the bits already are: out[f][j] = pack(in[f][8j .. 8j+7] ^ prbs), k_bch / 8
// bytes per BBFRAME. HBM-bound: k_ldpc-strided bit-bytes in (8 per lane and load: a wavefront reads 512 contiguous bytes), packed
// bytes out (rows of k_bch / 8 bytes are not word aligned for most codes, so a lane stores its byte: 64 contiguous bytes per
// wavefront store, one ninth of the traffic). The PRBS comes packed too (one byte per output byte, 6.7 KB, cache resident).
// A bit-byte is 0 or 1 (LDPC hard decisions); the multiply gathers the 8 low bits of a little-endian 64-bit word, first byte
// into the top bit of the result.
constexpr int PACK_UNROLL = 4;
__global__ __launch_bounds__(256) void bch_descramble_pack_kernel(const uint8_t *__restrict__ bits, long total_bytes, int k_ldpc, int row_bytes,
                                                                 const uint8_t *__restrict__ prbs_packed, uint8_t *__restrict__ out)
{
    const long stride = (long)gridDim.x * blockDim.x;
    long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i0 < total_bytes; i0 += PACK_UNROLL * stride) {
        uint2 v[PACK_UNROLL];
        int j[PACK_UNROLL];
#pragma unroll
        for (int u = 0; u < PACK_UNROLL; ++u) {
            const long i = i0 + u * stride;
            v[u] = make_uint2(0u, 0u); j[u] = 0;
            if (i < total_bytes) {
                const long f = i / row_bytes;
                j[u] = (int)(i - f * row_bytes);
                v[u] = *reinterpret_cast<const uint2 *>(bits + f * k_ldpc + 8L * j[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < PACK_UNROLL; ++u) {
            const long i = i0 + u * stride;
            if (i < total_bytes) {
                // bytes b0..b3 of a word w (each 0 / 1): (w * 0x08040201) >> 24 gives b0 << 3 | b1 << 2 | b2 << 1 | b3 in its low nibble
                const uint32_t hi = ((v[u].x & 0x01010101u) * 0x08040201u) >> 24, lo = ((v[u].y & 0x01010101u) * 0x08040201u) >> 24;
                out[i] = (uint8_t)((((hi & 15u) << 4) | (lo & 15u)) ^ prbs_packed[j[u]]);
            }
        }
    }
}

hipError_t launch_bch_descramble_pack(const uint8_t *bits, int n_frames, int k_ldpc, int k_bch, const uint8_t *prbs_packed, uint8_t *out,
                                      hipStream_t s)
{
    const long total = (long)n_frames * (k_bch / 8);
    long blocks = (total + 256L * PACK_UNROLL - 1) / (256L * PACK_UNROLL);
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(bch_descramble_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, s, bits, total, k_ldpc, k_bch / 8, prbs_packed, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------- demapper
// Replaces llr_demapper::execute and qpsk/qam16/qam64/qam256 (/root/reference/src/DVB_T2/llr_demapper.cpp:132-158,
// 160-228, 230-364, 366-535, 537-768). Floating-point products and sums are written with the non-contracting
// intrinsics so that the LLR arithmetic (x*p, (|x|-c)*p, round-to-nearest-even, truncating int8 cast) is the
// reference's operation for operation; only the order of the sum_s / sum_e reduction differs (tolerance in DESIGN.md).
__device__ __forceinline__ float2 derotate(float2 v, float c, float s)
{
    // (re + j im) * (c + j s), products and sums rounded separately as std::complex<float> operator*= does without FMA
    return make_float2(sub_r(mul_r(v.x, c), mul_r(v.y, s)), add_r(mul_r(v.x, s), mul_r(v.y, c)));
}

__device__ __forceinline__ float slice_axis(int mod, float x, float d)
{
    const float x2 = d * 2.0f, x4 = d * 4.0f, x6 = d * 6.0f;
    if (mod == 0) return x > 0 ? d : -d;
    if (mod == 1) {
        if (x > 0) return (x > x2) ? d * 3.0f : d;
        return (x < -x2) ? -(d * 3.0f) : -d;
    }
    if (mod == 2) {
        if (x > 0) {
            if (x > x4) return (x > x6) ? d * 7.0f : d * 5.0f;
            return (x > x2) ? d * 3.0f : d;
        }
        if (x < -x4) return (x > x6) ? -(d * 7.0f) : -(d * 5.0f);     // llr_demapper.cpp:407,427: '>' as written
        return (x < -x2) ? -(d * 3.0f) : -d;
    }
    const float x8 = d * 8.0f, x10 = d * 10.0f, x12 = d * 12.0f, x14 = d * 14.0f;
    if (x > 0) {
        if (x > x8) {
            if (x > x12) return (x > x14) ? d * 15.0f : d * 13.0f;
            return (x > x10) ? d * 11.0f : d * 9.0f;
        }
        if (x > x4) return (x > x6) ? d * 7.0f : d * 5.0f;
        return (x > x2) ? d * 3.0f : d;
    }
    if (x < -x8) {
        if (x < -x12) return (x < -x14) ? -(d * 15.0f) : -(d * 13.0f);
        return (x < -x10) ? -(d * 11.0f) : -(d * 9.0f);
    }
    if (x < -x4) return (x < -x6) ? -(d * 7.0f) : -(d * 5.0f);
    return (x < -x2) ? -(d * 3.0f) : -d;
}

// blockIdx.y = TI block of a batch: its cells at cells + y * cells_stride, its partial sums at partial + y * 2 * gridDim.x
__global__ __launch_bounds__(256) void demap_stats_kernel(DemapParams p, const float2 *__restrict__ cells, int n_snr,
                                                         double *__restrict__ partial, long cells_stride)
{
    cells += (long)blockIdx.y * cells_stride;
    partial += (long)blockIdx.y * 2 * gridDim.x;
    double ss = 0.0, se = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_snr; i += gridDim.x * blockDim.x) {
        float2 v = cells[i];
        if (p.rotate) v = derotate(v, p.rot_c, p.rot_s);
        const float sr = slice_axis(p.mod, v.x, p.d), si = slice_axis(p.mod, v.y, p.d);
        const float er = sub_r(v.x, sr), ei = sub_r(v.y, si);
        ss += (double)add_r(mul_r(sr, sr), mul_r(si, si));      // std::norm(s), float as in the reference
        se += (double)add_r(mul_r(er, er), mul_r(ei, ei));
    }
    __shared__ double sh[2][256];
    sh[0][threadIdx.x] = ss; sh[1][threadIdx.x] = se;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { sh[0][threadIdx.x] += sh[0][threadIdx.x + o]; sh[1][threadIdx.x] += sh[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = sh[0][0]; partial[2 * blockIdx.x + 1] = sh[1][0]; }
}

__global__ __launch_bounds__(64) void demap_stats_final_kernel(const double *__restrict__ partial, int blocks, float d,
                                                               float precision_override, float *__restrict__ sums, int sums_stride)
{
    partial += (long)blockIdx.x * 2 * blocks;
    sums += (long)blockIdx.x * sums_stride;
    // one wavefront folds the per-block partial sums; lane-strided accumulation then a fixed butterfly: deterministic
    double ss = 0.0, se = 0.0;
    for (int b = threadIdx.x; b < blocks; b += 64) { ss += partial[2 * b]; se += partial[2 * b + 1]; }
    for (int o = 32; o > 0; o >>= 1) { ss += __shfl_down(ss, o, 64); se += __shfl_down(se, o, 64); }
    if (threadIdx.x != 0) return;
    const float fs = (float)ss, fe = (float)se;
    float precision = div_r(mul_r(mul_r(8.0f, d), fs), fe);        // 8.0f * NORM * sum_s / sum_e
    if (precision_override > 0.0f) precision = precision_override;
    sums[0] = fs; sums[1] = fe; sums[2] = precision;
}

hipError_t launch_demap_stats(const DemapParams &p, const float2 *cells, int n_snr, double *partial, int blocks, float *sums,
                              float precision_override, hipStream_t s)
{
    hipLaunchKernelGGL(demap_stats_kernel, dim3(blocks), dim3(256), 0, s, p, cells, n_snr, partial, 0L);
    hipLaunchKernelGGL(demap_stats_final_kernel, dim3(1), dim3(64), 0, s, partial, blocks, p.d, precision_override, sums, 0);
    return hipGetLastError();
}

// n_batch TI blocks in one launch pair; partial holds n_batch * blocks pairs. Per block the additions happen in the order of
// launch_demap_stats, so the sums are the same bit for bit.
hipError_t launch_demap_stats_batch(const DemapParams &p, const float2 *cells, long cells_stride, int n_snr, int n_batch, double *partial,
                                    int blocks, float *sums, int sums_stride, float precision_override, hipStream_t s)
{
    hipLaunchKernelGGL(demap_stats_kernel, dim3(blocks, n_batch), dim3(256), 0, s, p, cells, n_snr, partial, cells_stride);
    hipLaunchKernelGGL(demap_stats_final_kernel, dim3(n_batch), dim3(64), 0, s, partial, blocks, p.d, precision_override, sums, sums_stride);
    return hipGetLastError();
}

__device__ __forceinline__ int8_t cast_i8_trunc(float v)
{
    return (int8_t)(uint8_t)((int)v & 0xff);      // cvttss2si + low byte, no saturation (16/64/256-QAM paths)
}

// frames_per_sums: FEC frames that share one statistics triple (one TI block); sums of TI block t at sums + t * sums_stride
// One workgroup stages a whole FEC frame (up to 64 800 B of LDS, so two workgroups per CU): the more wavefronts it has, the more
// cell loads are in flight per CU -- 1024 lanes instead of 256 (measured: see DESIGN.md K-demap)
#ifndef T2_DEMAP_THREADS
#define T2_DEMAP_THREADS 1024
#endif
__global__ __launch_bounds__(T2_DEMAP_THREADS) void demap_llr_kernel(DemapParams p, const float2 *__restrict__ cells,
                                                       const float *__restrict__ sums, int8_t *__restrict__ out, int frames_per_sums,
                                                       int sums_stride)
{
    extern __shared__ __attribute__((aligned(16))) int8_t stage[];        // one FEC frame in LDPC input order
    const int f = blockIdx.x;
    const float precision = sums[(size_t)(f / frames_per_sums) * sums_stride + 2];
    const float2 *src = cells + (size_t)f * p.cells_per_fec;
    const int levels = p.mod + 1;
    // a lane's cells of a pass and their LDS positions (an L2-resident table) are all read before the first is used: with two
    // workgroups per CU -- a frame each in LDS -- nothing else covers a round trip per cell (387 -> 325 us per 7676 frames)
    constexpr int U = 4;                       // 8 (a whole 256-QAM frame in one pass) needs more than the 64 VGPRs two 1024-lane workgroups leave: 405 us
    const uint32_t *addr32 = reinterpret_cast<const uint32_t *>(p.address);   // one dword per cell and level: the two LDS positions
    for (int c0 = threadIdx.x; c0 < p.cells_per_fec; c0 += U * blockDim.x) {
        float2 cell[U];
        uint32_t pos[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + u * blockDim.x;
            const bool in = c < p.cells_per_fec;
            cell[u] = in ? src[c] : make_float2(0.0f, 0.0f);
#pragma unroll
            for (int l = 0; l < 4; ++l) pos[u][l] = (in && l < levels && p.mod != 0) ? addr32[c * levels + l] : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + u * blockDim.x;
            if (c >= p.cells_per_fec) break;
            float2 v = cell[u];
            if (p.rotate) v = derotate(v, p.rot_c, p.rot_s);
            if (p.mod == 0) {                                              // quantize(): clamps (llr_demapper.cpp:770-776)
                float a = rintf(mul_r(v.x, precision)), b = rintf(mul_r(v.y, precision));
                a = fminf(fmaxf(a, -128.0f), 127.0f); b = fminf(fmaxf(b, -128.0f), 127.0f);
                stage[2 * c] = (int8_t)a; stage[2 * c + 1] = (int8_t)b;
                continue;
            }
            float thr = p.d * (float)(1 << p.mod);
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                if (l >= levels) break;
                float lx = rintf(mul_r(v.x, precision)), ly = rintf(mul_r(v.y, precision));
                if (p.saturate) { lx = fminf(fmaxf(lx, -128.0f), 127.0f); ly = fminf(fmaxf(ly, -128.0f), 127.0f); }
                stage[pos[u][l] & 0xffffu] = cast_i8_trunc(lx);
                stage[pos[u][l] >> 16] = cast_i8_trunc(ly);
                v.x = sub_r(fabsf(v.x), thr);
                v.y = sub_r(fabsf(v.y), thr);
                thr *= 0.5f;
            }
        }
    }
    __syncthreads();
    const uint2 *s2 = reinterpret_cast<const uint2 *>(stage);
    uint2 *dst = reinterpret_cast<uint2 *>(out + (size_t)f * p.fec_size);
    for (int i = threadIdx.x; i < p.fec_size / 8; i += blockDim.x) dst[i] = s2[i];
}

hipError_t launch_demap_llr(const DemapParams &p, const float2 *cells, int n_frames, const float *sums, int8_t *out, hipStream_t s,
                            int frames_per_sums, int sums_stride)
{
    if (frames_per_sums < 1) frames_per_sums = n_frames > 0 ? n_frames : 1;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(demap_llr_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 64800);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(demap_llr_kernel, dim3(n_frames), dim3(T2_DEMAP_THREADS), p.fec_size, s, p, cells, sums, out, frames_per_sums, sums_stride);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------- time de-interleaver
// Replaces the per-cell loop of time_deinterleaver::execute (/root/reference/src/DVB_T2/time_deinterleaver.cpp:316-345).
// Cell n of the TI block sits at row-column address d = (n mod cols)*rows + n div cols of the interleaver memory;
// perm[d] is its cell-de-interleaved position. I goes there, Q goes one cell earlier (cyclic inside the FEC block):
// the Q of a block's first cell is parked in first_q[block] and placed on the block's last cell by the fix-up kernel.
__global__ __launch_bounds__(256) void ti_scatter_kernel(TiParams p, const float2 *__restrict__ cells, int n0, int n,
                                                        float2 *__restrict__ out, float *__restrict__ first_q)
{
    float *o = reinterpret_cast<float *>(out);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int nn = n0 + i;
        const int row = nn / p.cols, col = nn - row * p.cols;
        const int ia = p.perm[col * p.rows + row];
        const float2 v = cells[i];
        o[2 * (size_t)ia] = v.x;
        const int blk = ia / p.cells_per_fec;
        if (ia - blk * p.cells_per_fec == 0) first_q[blk] = v.y;
        else o[2 * (size_t)(ia - 1) + 1] = v.y;
    }
}

hipError_t launch_ti_scatter(const TiParams &p, const float2 *cells, int n0, int n, float2 *out, float *first_q, hipStream_t s)
{
    int blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(ti_scatter_kernel, dim3(blocks), dim3(256), 0, s, p, cells, n0, n, out, first_q);
    return hipGetLastError();
}

__global__ void ti_fixup_kernel(TiParams p, const int32_t *__restrict__ order, const uint8_t *__restrict__ lost, int num_blocks,
                                const float *__restrict__ first_q, float2 *__restrict__ out)
{
    float *o = reinterpret_cast<float *>(out);
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < num_blocks; k += gridDim.x * blockDim.x) {
        if (lost[k]) continue;            // the reference overwrites this held value before ever storing it
        const int blk = order[k];
        o[2 * ((size_t)blk * p.cells_per_fec + p.cells_per_fec - 1) + 1] = first_q[blk];
    }
}

hipError_t launch_ti_fixup(const TiParams &p, const int32_t *order, const uint8_t *lost, int num_blocks, const float *first_q,
                           float2 *out, hipStream_t s)
{
    hipLaunchKernelGGL(ti_fixup_kernel, dim3((num_blocks + 255) / 256), dim3(256), 0, s, p, order, lost, num_blocks, first_q, out);
    return hipGetLastError();
}

// ---- the same for complete TI blocks, staged through LDS -------------------------------------------------------------------
// Cell n = row * cols + col of the TI block sits at interleaver address d = col * rows + row; FEC block b owns columns
// 5b .. 5b+4 and perm maps its d range onto its own output range (time_deinterleaver.cpp:174-266: the cell permutation is
// block-local), so a workgroup needs nothing but its block's cells.
#ifndef T2_TI_THREADS
#define T2_TI_THREADS 1024
#endif
// STATS: the demapper's hard-decision statistics (sum |s|^2, sum |e|^2 over the TI block, llr_demapper.cpp:564-676) formed here,
// where every de-interleaved cell is in a register on its way out -- one double pair per FEC block, folded per TI block by
// demap_stats_final_kernel -- instead of a second pass over the cells in HBM (demap_stats_kernel: 8 B per cell read again).
template <bool STATS>
__global__ __launch_bounds__(T2_TI_THREADS) void ti_block_kernel(TiParams p, const uint8_t *__restrict__ lost_by_block, int num_blocks,
                                                      const float2 *__restrict__ cells, long in_stride, float2 *__restrict__ out,
                                                      long out_stride, DemapParams dp, int n_snr, double *__restrict__ partial)
{
    extern __shared__ float ti_lds[];                    // [cells_per_fec][2]
    // FEC block b reads the 40-byte runs of columns 5b .. 5b+4 in every row: a 128-byte line holds the runs of three neighbouring
    // blocks. Workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so neighbours go to ONE XCD, one after the
    // other: XCD x takes blocks [x * per, x * per + per) of every frame (2.2x -> ~1.1x of the algorithmic bytes fetched).
    const int per = (num_blocks + 7) >> 3, xcd = (int)blockIdx.x & 7, k = (int)blockIdx.x >> 3;
    const int f = k / per, b = xcd * per + (k - f * per);
    if (b >= num_blocks) return;
    const int C = p.cells_per_fec, rows = p.rows;
    const float2 *in = cells + (long)f * in_stride;
    float2 *o = out + (long)f * out_stride + (long)b * C;
    const int32_t *perm = p.perm + (long)b * C;
    const int base = b * C;
    for (int t = threadIdx.x; t < C; t += blockDim.x) {
        const int row = t / 5, c5 = t - row * 5;
        const float2 v = in[(long)row * p.cols + 5 * b + c5];
        const int ia = perm[c5 * rows + row] - base;     // position inside the FEC block
        ti_lds[2 * ia] = v.x;
        ti_lds[2 * (ia == 0 ? C - 1 : ia - 1) + 1] = v.y; // Q travels one cell behind its I, cyclically (:321-336)
    }
    __syncthreads();
    const bool lost = lost_by_block[b] != 0;
    double ss = 0.0, se = 0.0;
    for (int t = threadIdx.x; t < C; t += blockDim.x) {
        float2 v = make_float2(ti_lds[2 * t], ti_lds[2 * t + 1]);
        if (t == C - 1 && lost) {
            reinterpret_cast<float *>(o)[2 * t] = v.x;
            if (STATS) v.y = reinterpret_cast<const float *>(o)[2 * t + 1];      // the Q this block never received: what the buffer holds
        } else o[t] = v;
        if (STATS && base + t < n_snr) {                                       // the arithmetic of demap_stats_kernel, cell by cell
            if (dp.rotate) v = derotate(v, dp.rot_c, dp.rot_s);
            const float sr = slice_axis(dp.mod, v.x, dp.d), si = slice_axis(dp.mod, v.y, dp.d);
            const float er = sub_r(v.x, sr), ei = sub_r(v.y, si);
            ss += (double)add_r(mul_r(sr, sr), mul_r(si, si));
            se += (double)add_r(mul_r(er, er), mul_r(ei, ei));
        }
    }
    if (STATS) {
        __shared__ double red[2][T2_TI_THREADS / 64];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { ss += __shfl_down(ss, d, 64); se += __shfl_down(se, d, 64); }
        if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ss; red[1][threadIdx.x >> 6] = se; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double a = 0.0, c = 0.0;
            for (int w = 0; w < (int)blockDim.x / 64; ++w) { a += red[0][w]; c += red[1][w]; }
            double *q = partial + 2 * ((long)f * num_blocks + b);
            q[0] = a; q[1] = c;
        }
    }
}

hipError_t launch_ti_blocks(const TiParams &p, const uint8_t *lost_by_block, int num_blocks, const float2 *cells, long in_stride,
                            float2 *out, long out_stride, int frames, hipStream_t s)
{
    const size_t lds = (size_t)p.cells_per_fec * 8;
    if (lds > 150 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ti_block_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(ti_block_kernel<false>, dim3((unsigned)(8 * ((num_blocks + 7) / 8) * frames)), dim3(T2_TI_THREADS), lds, s, p, lost_by_block, num_blocks,
                       cells, in_stride, out, out_stride, DemapParams{}, 0, (double *)nullptr);
    return hipGetLastError();
}

// The same with the demapper's statistics: partial = [frames][num_blocks] double pairs (scratch), sums + f * sums_stride receives
// (sum_s, sum_e, precision) of frame f's TI block. n_snr: cells of the TI block that count (all, or the first 2048 for QPSK).
hipError_t launch_ti_blocks_stats(const TiParams &p, const uint8_t *lost_by_block, int num_blocks, const float2 *cells, long in_stride,
                                  float2 *out, long out_stride, int frames, const DemapParams &dp, int n_snr, double *partial, float *sums,
                                  int sums_stride, float precision_override, hipStream_t s)
{
    const size_t lds = (size_t)p.cells_per_fec * 8;
    if (lds > 150 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ti_block_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(ti_block_kernel<true>, dim3((unsigned)(8 * ((num_blocks + 7) / 8) * frames)), dim3(T2_TI_THREADS), lds, s, p, lost_by_block, num_blocks,
                       cells, in_stride, out, out_stride, dp, n_snr, partial);
    hipLaunchKernelGGL(demap_stats_final_kernel, dim3(frames), dim3(64), 0, s, partial, num_blocks, dp.d, precision_override, sums, sums_stride);
    return hipGetLastError();
}

}  // namespace t2gpu
