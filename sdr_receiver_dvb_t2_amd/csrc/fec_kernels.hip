// fec_kernels.hip -- FEC-side streaming kernels for gfx950: BB descrambler (BCH stub), LLR demapper + bit de-interleaver,
// time / cell de-interleaver with cyclic Q-delay removal. All HBM-bound gather/scatter work; no matrix cores involved.
#include "fec_kernels.h"
#include "t2gpu_common.h"
#include <cstdlib>

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

// K-rows-to-host: a plain copy, 16 bytes per lane, for results that leave for page-locked host memory ON THE STREAM THAT MADE THEM
// (t2gpu_rx's host end in the overlap mode: a copy on a stream of its own is one more queue with work that waits for a decode, and
// whatever shares that hardware queue waits with it). The tail and unaligned buffers go byte by byte.
__global__ __launch_bounds__(256) void copy_bytes_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, long n16, long bytes)
{
    const long stride = (long)gridDim.x * blockDim.x;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
    uint4 *d4 = reinterpret_cast<uint4 *>(dst);
    for (long i = t; i < n16; i += stride) d4[i] = s4[i];
    for (long i = 16 * n16 + t; i < bytes; i += stride) dst[i] = src[i];
}
hipError_t launch_copy_bytes(const void *src, void *dst, long bytes, hipStream_t s)
{
    if (bytes <= 0) return hipSuccess;
    const bool aligned = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0;
    const long n16 = aligned ? bytes / 16 : 0;
    long blocks = ((aligned ? n16 : bytes) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(copy_bytes_kernel, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const uint8_t *>(src), static_cast<uint8_t *>(dst), n16, bytes);
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

// Hard decision on one axis: the comparison trees of llr_demapper.cpp:257-276 (16-QAM), :395-436 (64-QAM), :567-654 (256-QAM) without
// their branches -- a wavefront's lanes take all sixteen paths of the 256-QAM tree otherwise. The trees compare x > t on the positive
// side and x < -t on the negative one with t = d * 2, d * 4, ...: the level is the number of thresholds |x| exceeds (strictly), the
// amplitude d * (2 level + 1) the same float product the tree returns, x = 0 goes with the negatives. 64-QAM, negative side: the
// reference tests `x > x6` there (:407,427), which a negative x never passes: the outermost point is never decided.
__device__ __forceinline__ float slice_axis(int mod, float x, float d)
{
    const float a = fabsf(x);
    int lvl = 0;
    if (mod >= 1) lvl += a > d * 2.0f ? 1 : 0;
    if (mod >= 2) { lvl += a > d * 4.0f ? 1 : 0; lvl += a > d * 6.0f ? 1 : 0; }
    if (mod >= 3) { lvl += a > d * 8.0f ? 1 : 0; lvl += a > d * 10.0f ? 1 : 0; lvl += a > d * 12.0f ? 1 : 0; lvl += a > d * 14.0f ? 1 : 0; }
    const bool pos = x > 0;
    if (mod == 2 && !pos) lvl = min(lvl, 2);
    const float amp = d * (float)(2 * lvl + 1);
    return pos ? amp : -amp;
}

// ---- K-snr, exact: the reference's statistics are two SEQUENTIAL float sums over the TI block (llr_demapper.cpp:564-676; in the
// reference binary one vaddss per cell and sum, in cell order) -- float addition is not associative, and a tree or double-precision sum
// (the round 1-2 form; removed in round 5) differs from it by ~1e-4 relative, i.e. one LLR step on ~2 % of
// the positions, enough to flip a SIMD batch at the decoding threshold. This kernel reproduces the sequential sum bit for bit, in
// parallel: while the running sum s stays inside one binade [2^E, 2^(E+1)), fl(s + x) = s + round(x / ulp) ulp with ulp = 2^(E-23) --
// an INTEGER addition of the terms quantised at that ulp, which is associative -- except where x / ulp sits exactly on a half (the
// tie is broken by the parity of s / ulp) and at the addition that carries s into the next binade. So: quantise a chunk's terms at the
// current ulp, reduce them per wavefront (2048 consecutive cells each); if no total carries the sum out of its binade and nothing ties,
// the chunk is one integer addition. Otherwise the first wavefront with such an event has its terms prefix-summed over the workgroup,
// s jumps to just before the event in one step, that one addition is performed as a real float add, and the walk goes on from
// there (with the new ulp if it changed). A TI block of 1.6 M cells has about 25 events
// per sum (one per binade, plus ties while s is still small). One workgroup per TI block; the chunks of a block are its cells in
// order, SEQ_U consecutive cells per lane. Checked against a sequential float loop on tie-heavy data in tests/test_fec_gpu.py.
constexpr int SEQ_THREADS = 512, SEQ_U = 32, SEQ_CHUNK = SEQ_THREADS * SEQ_U, SEQ_WAVES = SEQ_THREADS / 64;
constexpr int SEQ_HEAD = 2048;                     // cells summed by the plain sequential loop before the chunked walk starts
constexpr int SEQ_CAP = 1 << 25;
constexpr int SEQ_LDS_BYTES = SEQ_THREADS * 4 * 4;                 // one wavefront's run of a chunk (2048 terms) in cell order; the head before that
static_assert(SEQ_U * 64 == SEQ_THREADS * 4 && SEQ_HEAD * 4 <= SEQ_LDS_BYTES && SEQ_HEAD <= SEQ_CHUNK, "the head is staged in the same LDS");                    // quantised terms and their sums saturate here: anything >= 2^24 ends the binade anyway

struct SeqShared {
    int4 tot[2][1][SEQ_WAVES / 4];                  // [buffer][sum][wavefront]: saturated totals of the quantised terms, bit 30 = a rounding tie among them
    int scan[SEQ_WAVES];                            // event path
    int before, first;
    float x;
};

__device__ __forceinline__ int sat_add(int a, int b) { return min(a + b, SEQ_CAP); }   // associative on [0, SEQ_CAP]
// x / ulp rounded to an integer as fl(s + x) rounds it while s stays in its binade; tie: the fraction is exactly one half
// (then the parity of s / ulp decides). All float operations here are exact (scaling by a power of two, floor, difference).
__device__ __forceinline__ int seq_quant(float x, float inv_ulp, bool &tie)
{
    const float y = fminf(mul_r(x, inv_ulp), (float)SEQ_CAP), m = floorf(y), f = sub_r(y, m);
    tie = f == 0.5f;
    return (int)m + (f > 0.5f ? 1 : 0);
}
struct SeqScale { float inv_ulp, ulp; int S; };
__device__ __forceinline__ SeqScale seq_scale(float s)
{
    const int E = (int)((__float_as_uint(s) >> 23) & 0xffu) - 127;      // s in [2^E, 2^(E+1)) (the sums are far from the denormals)
    SeqScale r;
    r.inv_ulp = __uint_as_float((uint32_t)(127 + 23 - E) << 23);
    r.ulp = __uint_as_float((uint32_t)(127 + E - 23) << 23);
    r.S = (int)mul_r(s, r.inv_ulp);                                    // s / ulp: an integer in [2^23, 2^24)
    return r;
}

// Event path: one step of the walk over a run of SEQ_THREADS * U terms x in cell order (this lane: run indices tid * U + u) from index
// a on; returns the new a (CH = run consumed). s is uniform over the workgroup. Three barriers; taken ~25 times per sum and TI block.
template <int U>
__device__ __forceinline__ int seq_step(const float (&x)[U], int a, float &s, SeqShared &sh)
{
    constexpr int CH = SEQ_THREADS * U;
    const int tid = (int)threadIdx.x, k0 = tid * U, lane = tid & 63, wave = tid >> 6;
    __syncthreads();                                                   // whoever still reads the shared words of the previous step
    if (tid == 0) sh.first = CH;
    if (s == 0.0f) {                                                   // 0 + x = x exactly: take the first non-zero term as it is
        int cand = CH;
#pragma unroll
        for (int u = U - 1; u >= 0; --u) if (k0 + u >= a && x[u] > 0.0f) cand = k0 + u;
        __syncthreads();
        if (cand < CH) atomicMin(&sh.first, cand);
        __syncthreads();
        const int p = sh.first;
        if (p >= CH) return CH;
        if (tid == p / U) {
#pragma unroll
            for (int u = 0; u < U; ++u) if (u == p % U) sh.x = x[u];       // (no dynamic register index: that would put x[] in scratch)
        }
        __syncthreads();
        s = sh.x;
        return p + 1;
    }
    const SeqScale sc = seq_scale(s);
    int pre[U], loc = 0, tie_at = CH;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        int q = 0;
        if (k0 + u >= a) {
            bool tie;
            q = seq_quant(x[u], sc.inv_ulp, tie);
            if (tie && tie_at == CH) tie_at = k0 + u;
        }
        loc = sat_add(loc, q);
        pre[u] = loc;                                                  // inclusive prefix inside the lane
    }
    int inc = loc;                                                     // inclusive prefix over the lanes (lane order = cell order)
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d, 64); if (lane >= d) inc = sat_add(inc, o); }
    if (lane == 63) sh.scan[wave] = inc;
    __syncthreads();
    int off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < SEQ_WAVES; ++w) { const int t = sh.scan[w]; if (w < wave) off = sat_add(off, t); total = sat_add(total, t); }
    off = sat_add(off, __shfl_up(inc, 1, 64) * (lane > 0));            // exclusive prefix of this lane
    // first term that carries s out of its binade (s / ulp reaches 2^24), or whose rounding is a tie
    int cand = tie_at;
#pragma unroll
    for (int u = U - 1; u >= 0; --u) if (k0 + u >= a && sc.S + sat_add(off, pre[u]) >= (1 << 24) && k0 + u < cand) cand = k0 + u;
    if (cand < CH) atomicMin(&sh.first, cand);
    __syncthreads();
    const int p = sh.first;
    if (p >= CH) {                                              // no event: the whole rest of the chunk in one integer addition
        s = mul_r((float)(sc.S + total), sc.ulp);
        return CH;
    }
    if (tid == p / U) {
        sh.before = off;                                               // quantised terms a .. p - 1 (no event among them: below 2^24)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (u == p % U) sh.x = x[u];
            if (u + 1 == p % U) sh.before = sat_add(off, pre[u]);
        }
    }
    __syncthreads();
    s = add_r(mul_r((float)(sc.S + sh.before), sc.ulp), sh.x);         // the event's own addition, as the float addition it is
    return p + 1;
}

// (|s|^2, |e|^2) of one cell: de-rotate, slice, error, both norms as float (llr_demapper.cpp:564-676)
__device__ __forceinline__ float2 demap_term(const DemapParams &p, float2 v)
{
    if (p.rotate) v = derotate(v, p.rot_c, p.rot_s);
    const float sr = slice_axis(p.mod, v.x, p.d), si = slice_axis(p.mod, v.y, p.d);
    const float er = sub_r(v.x, sr), ei = sub_r(v.y, si);
    return make_float2(add_r(mul_r(sr, sr), mul_r(si, si)), add_r(mul_r(er, er), mul_r(ei, ei)));
}

// The terms of the two sums, one pair per cell, by the whole device (the per-cell arithmetic of llr_demapper.cpp:564-676: de-rotate,
// slice, |s|^2 and |e|^2 as float): the sequential walk below is one workgroup per TI block and must not carry this work.
__global__ __launch_bounds__(256) void demap_terms_kernel(DemapParams p, const float2 *__restrict__ cells, int n_snr, long cells_stride,
                                                         float2 *__restrict__ terms, long terms_stride)
{
    cells += (long)blockIdx.y * cells_stride;
    // two planes per TI block (all |s|^2, then all |e|^2): a walk reads its own sum's terms only
    float *ts = reinterpret_cast<float *>(terms + (long)blockIdx.y * terms_stride), *te = ts + terms_stride;
    const int padded = (n_snr + SEQ_CHUNK - 1) / SEQ_CHUNK * SEQ_CHUNK;        // zeros behind the block: the walk reads whole chunks unguarded
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < padded; i += gridDim.x * blockDim.x) {
        float2 t = make_float2(0.0f, 0.0f);
        if (i < n_snr) t = demap_term(p, cells[i]);
        ts[i] = t.x;
        te[i] = t.y;
    }
}

// One workgroup per (TI block, sum) walks the sum's terms in cell order, SEQ_CHUNK at a time. Lane l of wavefront w holds the chunk's
// terms w * 64 * SEQ_U + r * 256 + 4 l + {0..3}, r < SEQ_U / 4 (16-byte loads, 1 KB per wavefront and instruction); the common case
// needs only the chunk's total, which no order affects: without a tie, round-to-nearest-even of x / ulp IS the rounding of the addition.
// A chunk with an event is transposed through LDS so that every lane holds SEQ_U consecutive terms, and walked by seq_step.
// The loop is a latency chain (load a chunk, reduce it, next): the terms of the next TWO chunks are in flight while one is summed.
__device__ __forceinline__ int seq_pos(int wave, int lane, int u) { return wave * 64 * SEQ_U + (u >> 2) * 256 + 4 * lane + (u & 3); }
constexpr int SEQ_R = SEQ_U / 4;

__device__ __forceinline__ void seq_load(float4 (&c)[SEQ_R], const float4 *src, int base)
{
#pragma unroll
    for (int r = 0; r < SEQ_R; ++r) c[r] = src[base / 4 + 64 * r];
}

// one chunk: x = this lane's terms (seq_pos order), s = the running sum on entry and exit (uniform); it = count of total exchanges so
// far (picks the buffer of sh.tot). Common case: no term of the chunk ties and the chunk does not leave the binade -> one reduction,
// one barrier; without a tie, round-to-nearest-even of x / ulp is what the addition's own rounding adds. Otherwise the event is
// narrowed down to the first WAVEFRONT whose 2048 consecutive terms contain it (every wavefront's total and tie flag are known from
// the reduction): the wavefronts before it are one integer addition, its own terms go through LDS to all lanes, four consecutive
// terms each, and are walked event by event (seq_step<4>); the wavefronts behind it are reduced again at the scale the sum has then.
// nf (per lane): bit 0 = a term of this lane was +-inf, bit 1 = NaN. Such terms (a dead equaliser symbol: 0 * inf) are taken out of
// the walk -- quantised they would all be "events", tens of thousands of serial steps -- and the result is what the reference's float
// loop gives for them: NaN once a NaN was added, +inf after an inf (the terms are norms: no inf - inf). One float add per term finds
// them (fast-class op; the terms are >= 0, so a non-finite one makes the lane's plain sum non-finite).
__device__ __forceinline__ void seq_chunk(float (&x)[SEQ_U], int base, float &s, SeqShared &sh, float *seq_tr, int &it, int &nf)
{
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (base < SEQ_HEAD) {                                             // (first chunk only) the head's terms are in the sum already
#pragma unroll
        for (int u = 0; u < SEQ_U; ++u) if (base + seq_pos(wave, lane, u) < SEQ_HEAD) x[u] = 0.0f;
    }
    for (int w_from = 0; w_from < SEQ_WAVES;) {                        // uniform
        SeqScale sc{0.0f, 0.0f, 0};
        int word = 1 << 30;
        float plain = 0.0f;
#pragma unroll
        for (int u = 0; u < SEQ_U; ++u) plain += x[u];
        const int bad_wave = __ballot(!(plain <= 3.402823466e+38f)) != 0ull ? 1 << 29 : 0;
        if (s != 0.0f) {                                               // uniform (a block of zeros stays on the event path)
            sc = seq_scale(s);
            int loc = 0;
            float far = 0.0f;                                          // largest |y - rint(y)| among the terms: 1/2 = a tie
#pragma unroll
            for (int u = 0; u < SEQ_U; ++u) {
                const float y = fminf(mul_r(x[u], sc.inv_ulp), (float)SEQ_CAP), r = rintf(y);
                loc += (int)r;                                         // <= SEQ_U * 2^25: no overflow
                far = fmaxf(far, fabsf(sub_r(y, r)));
            }
            word = min(loc, SEQ_CAP) | (far == 0.5f ? 1 << 30 : 0);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            const int o = __shfl_xor(word, d, 64);
            word = sat_add(word & (SEQ_CAP * 2 - 1), o & (SEQ_CAP * 2 - 1)) | ((word | o) & (1 << 30));
        }
        word |= bad_wave;
        const int buf = it & 1;
        ++it;
        if (lane == 0) reinterpret_cast<int *>(sh.tot[buf][0])[wave] = word;
        __syncthreads();
        int tot[SEQ_WAVES];
#pragma unroll
        for (int w = 0; w < SEQ_WAVES / 4; ++w) {
            const int4 t = sh.tot[buf][0][w];
            tot[4 * w] = t.x; tot[4 * w + 1] = t.y; tot[4 * w + 2] = t.z; tot[4 * w + 3] = t.w;
        }
        int anybad = 0;
#pragma unroll
        for (int w = 0; w < SEQ_WAVES; ++w) anybad |= tot[w];
        if (anybad & (1 << 29)) {                                      // uniform, rare: drop the non-finite terms, remember them, same chunk again
#pragma unroll
            for (int u = 0; u < SEQ_U; ++u) {
                const bool fin = fabsf(x[u]) <= 3.402823466e+38f;
                nf |= fin ? 0 : (x[u] != x[u] ? 2 : 1);
                x[u] = fin ? x[u] : 0.0f;
            }
            continue;
        }
        int S = sc.S, wev = SEQ_WAVES;                                 // first wavefront >= w_from with an event among its terms
#pragma unroll
        for (int w = 0; w < SEQ_WAVES; ++w) {
            if (w >= w_from && wev == SEQ_WAVES) {
                const int v = tot[w] & (SEQ_CAP * 2 - 1);
                if (((tot[w] >> 30) & 1) || S + v >= (1 << 24)) wev = w;
                else S += v;
            }
        }
        if (s != 0.0f) s = mul_r((float)S, sc.ulp);                    // the sum in front of wavefront wev (exact: an integer below 2^24)
        if (wev == SEQ_WAVES) return;
        if (wave == wev) {
#pragma unroll
            for (int u = 0; u < SEQ_U; ++u) seq_tr[(u >> 2) * 256 + 4 * lane + (u & 3)] = x[u];      // cell order inside the wavefront's run
        }
        __syncthreads();
        const float4 y4 = reinterpret_cast<const float4 *>(seq_tr)[tid];
        const float xt[4] = {y4.x, y4.y, y4.z, y4.w};
        for (int a = 0; a < SEQ_THREADS * 4;) a = seq_step<4>(xt, a, s, sh);
        __syncthreads();                                               // seq_tr is written again by the next event
        w_from = wev + 1;
    }
}

// blockIdx.x = TI block, blockIdx.y = which sum (0: sum_s, 1: sum_e -- two independent chains, a workgroup each); the scale is formed
// by demap_scale_kernel from the two results.
__global__ __launch_bounds__(SEQ_THREADS) void demap_stats_exact_kernel(const float2 *__restrict__ terms, int n_snr, long terms_stride,
                                                                       float *__restrict__ sums, int sums_stride)
{
    __shared__ SeqShared sh;
    extern __shared__ float seq_tr[];                                  // [SEQ_CHUNK + pad]: a chunk's terms in cell order
    const int which = (int)blockIdx.y;
    const float *tp = reinterpret_cast<const float *>(terms + (long)blockIdx.x * terms_stride) + (long)which * terms_stride;   // this sum's plane
    sums += (long)blockIdx.x * sums_stride;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float4 *src = reinterpret_cast<const float4 *>(tp) + (wave * 64 * SEQ_U) / 4 + lane;        // + 64 r: this lane's quads inside a chunk
    float s = 0.0f;                                                    // the running sum: uniform over the workgroup
    int nf = 0;                                                        // non-finite terms met by this lane (seq_chunk)
    float4 c0[SEQ_R], c1[SEQ_R];
    seq_load(c0, src, 0);
    if (SEQ_CHUNK < n_snr) seq_load(c1, src, SEQ_CHUNK);
    // The head of the block as the plain loop it is: while the sum is small every few additions change the binade (two thirds of a
    // block's events fall into its first couple of thousand cells). Staged through LDS; every lane reads the same word (a broadcast)
    // and runs the same additions: uniform, no exchange.
    {
        for (int k = tid; k < SEQ_HEAD; k += SEQ_THREADS) {                         // (zeros behind the block's end)
            const float v = tp[k];
            const bool fin = fabsf(v) <= 3.402823466e+38f;
            nf |= fin ? 0 : (v != v ? 2 : 1);
            seq_tr[k] = fin ? v : 0.0f;
        }
        __syncthreads();
        if (wave == 0) {                                               // one wavefront: eight of them reading every word kept the LDS pipe busy for 2/3 of the loop
            const float4 *h4 = reinterpret_cast<const float4 *>(seq_tr);
#pragma unroll 4
            for (int k = 0; k < SEQ_HEAD / 4; ++k) { const float4 t = h4[k]; s = add_r(add_r(add_r(add_r(s, t.x), t.y), t.z), t.w); }
            if (lane == 0) sh.x = s;
        }
        __syncthreads();                                               // seq_tr is the event path's from here on
        s = sh.x;
    }
#ifdef T2_SEQ_PROF
    long long t_head = __builtin_amdgcn_s_memtime(); long long hb[6] = {0,0,0,0,0,0}; int hn[6] = {0,0,0,0,0,0};
#define PROF_CHUNK(B) { const long long t0 = __builtin_amdgcn_s_memtime(); seq_chunk(x, B, s, sh, seq_tr, it, nf); const long long dt = __builtin_amdgcn_s_memtime() - t0; const int bk = dt < 4000 ? 0 : dt < 8000 ? 1 : dt < 16000 ? 2 : dt < 32000 ? 3 : dt < 64000 ? 4 : 5; _Pragma("unroll") for (int z = 0; z < 6; ++z) if (z == bk) { hb[z] += dt; ++hn[z]; } }
#else
#define PROF_CHUNK(B) seq_chunk(x, B, s, sh, seq_tr, it, nf)
#endif
    int it = 0;
    for (int base = 0; base < n_snr; base += 2 * SEQ_CHUNK) {
        float x[SEQ_U];
#pragma unroll
        for (int r = 0; r < SEQ_R; ++r) { x[4 * r] = c0[r].x; x[4 * r + 1] = c0[r].y; x[4 * r + 2] = c0[r].z; x[4 * r + 3] = c0[r].w; }
        if (base + 2 * SEQ_CHUNK < n_snr) seq_load(c0, src, base + 2 * SEQ_CHUNK);
        PROF_CHUNK(base);
        if (base + SEQ_CHUNK >= n_snr) break;
#pragma unroll
        for (int r = 0; r < SEQ_R; ++r) { x[4 * r] = c1[r].x; x[4 * r + 1] = c1[r].y; x[4 * r + 2] = c1[r].z; x[4 * r + 3] = c1[r].w; }
        if (base + 3 * SEQ_CHUNK < n_snr) seq_load(c1, src, base + 3 * SEQ_CHUNK);
        PROF_CHUNK(base + SEQ_CHUNK);
    }
#ifdef T2_SEQ_PROF
    if (threadIdx.x == 0 && blockIdx.x < 2) printf("blk %d sum %d: total %lld cycles; chunks <4k %d (%lld) <8k %d (%lld) <16k %d (%lld) <32k %d (%lld) <64k %d (%lld) more %d (%lld)\n", (int)blockIdx.x, which, __builtin_amdgcn_s_memtime() - t_head, hn[0], hb[0], hn[1], hb[1], hn[2], hb[2], hn[3], hb[3], hn[4], hb[4], hn[5], hb[5]);
#endif
    const int any_nan = __syncthreads_or(nf & 2), any_inf = __syncthreads_or(nf & 1);
    if (any_nan) s = __int_as_float(0x7fc00000);
    else if (any_inf) s = __int_as_float(0x7f800000);
    if (threadIdx.x == 0) sums[which] = s;
}

// precision = 8.0f * NORM * sum_s / sum_e (or the caller's), block by block
__global__ void demap_scale_kernel(float d, float precision_override, float *__restrict__ sums, int sums_stride, int n_batch)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_batch) return;
    float *q = sums + (long)i * sums_stride;
    float precision = div_r(mul_r(mul_r(8.0f, d), q[0]), q[1]);
    if (precision_override > 0.0f) precision = precision_override;
    q[2] = precision;
}


long demap_terms_padded(int n_snr) { return ((long)n_snr + SEQ_CHUNK - 1) / SEQ_CHUNK * SEQ_CHUNK; }   // term pairs of scratch per TI block

static hipError_t launch_exact(const DemapParams &p, const float2 *cells, long cells_stride, int n_snr, int n_batch, float2 *terms, float *sums,
                               int sums_stride, float precision_override, hipStream_t s)
{
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(demap_stats_exact_kernel), SEQ_LDS_BYTES)) return e;
    const long padded = demap_terms_padded(n_snr);
    if (cells) {                                                      // (nullptr: the terms are there already, launch_ti_blocks formed them)
        int bx = (int)((padded + 256 * 4 - 1) / (256 * 4));
        bx = bx < 1 ? 1 : (bx > 2048 ? 2048 : bx);
        hipLaunchKernelGGL(demap_terms_kernel, dim3(bx, n_batch), dim3(256), 0, s, p, cells, n_snr, cells_stride, terms, padded);
    }
    hipLaunchKernelGGL(demap_stats_exact_kernel, dim3(n_batch, 2), dim3(SEQ_THREADS), SEQ_LDS_BYTES, s, terms, n_snr, padded, sums, sums_stride);
    hipLaunchKernelGGL(demap_scale_kernel, dim3((n_batch + 63) / 64), dim3(64), 0, s, p.d, precision_override, sums, sums_stride, n_batch);
    return hipGetLastError();
}

bool demap_stats_exact_form() { return true; }              // (rounds 1-2 summed as a tree in double: 2e-4 off the reference's sequential float sums)
hipError_t launch_demap_stats_from_terms(const DemapParams &p, int n_snr, int n_batch, float2 *terms, float *sums, int sums_stride,
                                         float precision_override, hipStream_t s)
{
    return launch_exact(p, nullptr, 0L, n_snr, n_batch, terms, sums, sums_stride, precision_override, s);
}

// terms: scratch of n_snr float pairs (exact form); partial / blocks: scratch of the tree form
hipError_t launch_demap_stats(const DemapParams &p, const float2 *cells, int n_snr, double *partial, int blocks, float2 *terms, float *sums,
                              float precision_override, hipStream_t s)
{
    (void)partial; (void)blocks;
    return launch_exact(p, cells, 0L, n_snr, 1, terms, sums, 0, precision_override, s);
}

// n_batch TI blocks in one launch pair: every block summed on its own exactly as launch_demap_stats does (terms: n_batch * n_snr pairs)
hipError_t launch_demap_stats_batch(const DemapParams &p, const float2 *cells, long cells_stride, int n_snr, int n_batch, double *partial,
                                    int blocks, float2 *terms, float *sums, int sums_stride, float precision_override, hipStream_t s)
{
    (void)partial; (void)blocks;
    return launch_exact(p, cells, cells_stride, n_snr, n_batch, terms, sums, sums_stride, precision_override, s);
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
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(demap_llr_kernel), 64800)) return e;
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
// TERMS: the demapper's statistics terms of the cells are formed on the way out (the cells are read once for both)
template <bool TERMS>
__global__ __launch_bounds__(T2_TI_THREADS) void ti_block_kernel(TiParams p, const uint8_t *__restrict__ lost_by_block, int num_blocks,
                                                      const float2 *__restrict__ cells, long in_stride, float2 *__restrict__ out,
                                                      long out_stride, DemapParams dp, float *__restrict__ terms, long plane, int n_snr)
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
    float *ts = TERMS ? terms + (long)f * 2 * plane + base : nullptr;      // plane of |s|^2, the one of |e|^2 `plane` floats on
    for (int t = threadIdx.x; t < C; t += blockDim.x) {
        float2 v = make_float2(ti_lds[2 * t], ti_lds[2 * t + 1]);
        if (t == C - 1 && lost) {
            reinterpret_cast<float *>(o)[2 * t] = v.x;
            if (TERMS) v.y = o[t].y;                                      // the Q that stays in the caller's buffer (the parked-Q loss)
        } else o[t] = v;
        if (TERMS && base + t < n_snr) {
            const float2 tm = demap_term(dp, v);
            ts[t] = tm.x;
            ts[plane + t] = tm.y;
        }
    }
}

hipError_t launch_ti_blocks(const TiParams &p, const uint8_t *lost_by_block, int num_blocks, const float2 *cells, long in_stride,
                            float2 *out, long out_stride, int frames, hipStream_t s, const TiTerms *tt)
{
    const size_t lds = (size_t)p.cells_per_fec * 8;
    if (lds > 150 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(ti_block_kernel<false>), (int)lds);
        if (e == hipSuccess) e = ensure_dynamic_lds(reinterpret_cast<const void *>(ti_block_kernel<true>), (int)lds);
        if (e != hipSuccess) return e;
    }
    const dim3 grid((unsigned)(8 * ((num_blocks + 7) / 8) * frames));
    if (tt && tt->terms) {
        const long plane = demap_terms_padded(tt->n_snr);
        const long key = ((long)tt->n_snr * 4099 + plane) * 8191 + frames;
        if (plane > tt->n_snr && !(tt->pad_key && *tt->pad_key == key)) {   // zeros behind every plane's n_snr terms: the walk reads whole chunks
            if (tt->pad_key) *tt->pad_key = key;                         // (written once per geometry: the kernel below never touches them)
            hipError_t e = hipMemset2DAsync(reinterpret_cast<float *>(tt->terms) + tt->n_snr, (size_t)plane * 4, 0, (size_t)(plane - tt->n_snr) * 4,
                                            (size_t)2 * frames, s);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(ti_block_kernel<true>, grid, dim3(T2_TI_THREADS), lds, s, p, lost_by_block, num_blocks, cells, in_stride, out, out_stride,
                           *tt->dp, reinterpret_cast<float *>(tt->terms), plane, tt->n_snr);
    } else {
        hipLaunchKernelGGL(ti_block_kernel<false>, grid, dim3(T2_TI_THREADS), lds, s, p, lost_by_block, num_blocks, cells, in_stride, out, out_stride,
                           DemapParams{}, (float *)nullptr, 0L, 0);
    }
    return hipGetLastError();
}

}  // namespace t2gpu
