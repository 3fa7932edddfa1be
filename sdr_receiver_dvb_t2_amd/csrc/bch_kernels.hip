// bch_kernels.hip -- K-bch for gfx950: the outer code of DVB-T2 (EN 302 755 clause 6.1.1) checked and corrected on the LDPC
// decoder's output, one workgroup per FEC frame. The reference stage has no decoder to restate (bch_decoder.cpp:136
// "// TODO BCH decode": it keeps the first k_bch bits and descrambles); this is SURVEY.md 8(f)-2, opt-in, in front of
// K-descramble, checked against oracle/bch_oracle.c (textbook arithmetic, nothing shared with this file).
//
// HBM sees every bit once (64.8 KB per normal frame at one bit per byte, the LDPC kernel's output format). Syndromes are not
// evaluated bit by bit: the frame is packed to bytes in LDS, each of the 256 lanes takes a contiguous run of bytes and pushes
// it through t byte-wide remainder registers (r(x) mod m_j(x), 256-entry tables in LDS, as a CRC would), turns each remainder
// into the field element it stands for (sum over its bits of alpha^(j b)), weights it by alpha^(j * bits that follow the run)
// and the workgroup XORs the lanes' shares. A clean frame -- every frame once the LDPC has converged -- ends there. Otherwise
// lane 0 runs Berlekamp-Massey on the 2t syndromes (the even ones are squares) and all lanes search the n positions for roots
// of the locator; the frame is corrected only when the locator has as many roots inside the frame as its degree.
#include "bch_kernels.h"

namespace t2gpu {

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ uint32_t pack4(uint32_t w)      // four one-bit bytes, first bit -> bit 3
{
    return ((w & 0x01010101u) * 0x08040201u) >> 24;
}

template <int M, int T>
__global__ __launch_bounds__(kThreads) void bch_decode_kernel(uint8_t *__restrict__ bits, int n_frames, BchDev p,
                                                              int32_t *__restrict__ status)
{
    constexpr int ORDER = (1 << M) - 1;
    constexpr uint32_t MASK = (1u << M) - 1u;
    constexpr int MAXB = 28;                               // bytes per lane: ceil(54000 / 8 / 256) = 27
    __shared__ uint8_t pk[kThreads * (MAXB | 1)];
    __shared__ uint16_t rem[T * 256], basis[T * 16];
    __shared__ uint32_t part[kThreads / 64][T / 2];
    __shared__ uint16_t S[2 * T], C[2 * T + 2], B[2 * T + 2], Tm[2 * T + 2], logC[2 * T + 2];
    __shared__ int ctl[2 + T];                             // locator degree, roots found, root positions

    const int tid = threadIdx.x, n = p.n_bits, nbytes = n >> 3;
    const int run = (nbytes + kThreads - 1) / kThreads, stride = run | 1;     // odd stride: lanes' runs start in different banks
    for (int i = tid; i < T * 256; i += kThreads) rem[i] = p.rem[i];
    if (tid < T * 16) basis[tid] = p.basis[tid];

    for (int f = blockIdx.x; f < n_frames; f += gridDim.x) {
        uint8_t *row = bits + (size_t)f * n;
        __syncthreads();
        for (int b = tid; b < nbytes; b += kThreads) {
            const uint2 w = reinterpret_cast<const uint2 *>(row)[b];
            pk[(b / run) * stride + b % run] = (uint8_t)(pack4(w.x) << 4 | pack4(w.y));
        }
        __syncthreads();

        // remainders of this lane's run modulo the t minimal polynomials
        uint32_t r[T];
#pragma unroll
        for (int i = 0; i < T; ++i) r[i] = 0;
        const int first = tid * run, last = min(first + run, nbytes);
        for (int k = first; k < last; ++k) {
            const uint32_t byte = pk[tid * stride + (k - first)];
#pragma unroll
            for (int i = 0; i < T; ++i) {
                const uint32_t idx = ((r[i] >> (M - 8)) ^ byte) & 0xffu;
                r[i] = ((r[i] << 8) & MASK) ^ rem[i * 256 + idx];
            }
        }
        // -> field elements, weighted by the position of the run's last bit. The byte-wide step leaves (run(x) x^M) mod m_j, as a
        // CRC register does: M is taken off the exponent again.
        const int tail = 8 * (nbytes - last) + ORDER - M;
        uint32_t el[T];
#pragma unroll
        for (int i = 0; i < T; ++i) {
            uint32_t v = 0;
#pragma unroll
            for (int b = 0; b < M; ++b) v ^= (0u - (r[i] >> b & 1u)) & basis[i * 16 + b];
            if (v) v = p.exp[(p.log[v] + (2 * i + 1) * tail) % ORDER];
            el[i] = v;
        }
        // XOR over the workgroup, two syndromes per register
        uint32_t w2[T / 2];
#pragma unroll
        for (int i = 0; i < T / 2; ++i) {
            uint32_t v = el[2 * i] | el[2 * i + 1] << 16;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v ^= __shfl_xor(v, d, 64);
            w2[i] = v;
        }
        if ((tid & 63) == 0)
#pragma unroll
            for (int i = 0; i < T / 2; ++i) part[tid >> 6][i] = w2[i];
        __syncthreads();
        uint32_t any = 0;
#pragma unroll
        for (int i = 0; i < T / 2; ++i) {
            uint32_t v = 0;
#pragma unroll
            for (int wv = 0; wv < kThreads / 64; ++wv) v ^= part[wv][i];
            w2[i] = v;
            any |= v;
        }
        if (!any) {                                         // a codeword: the common case
            if (tid == 0) status[f] = 0;
            continue;
        }

        // ---- the rare path: locator by Berlekamp-Massey (lane 0), roots by all lanes
        auto mul = [&](uint32_t a, uint32_t b) -> uint32_t { return (a && b) ? p.exp[(p.log[a] + p.log[b]) % ORDER] : 0u; };
        if (tid == 0) {
#pragma unroll
            for (int i = 0; i < T / 2; ++i) {
                S[4 * i] = (uint16_t)(w2[i] & 0xffffu);        // S_(4i+1)
                S[4 * i + 2] = (uint16_t)(w2[i] >> 16);        // S_(4i+3)
            }
            for (int j = 2; j <= 2 * T; j += 2) {              // S_2j = S_j^2
                const uint32_t h = S[j / 2 - 1];
                S[j - 1] = (uint16_t)(h ? p.exp[(2 * p.log[h]) % ORDER] : 0u);
            }
            for (int i = 0; i < 2 * T + 2; ++i) C[i] = B[i] = 0;
            C[0] = B[0] = 1;
            int L = 0, sh = 1;
            uint32_t bb = 1;
            for (int k = 0; k < 2 * T; ++k) {
                uint32_t d = S[k];
                for (int i = 1; i <= L; ++i) d ^= mul(C[i], S[k - i]);
                if (d == 0) { ++sh; continue; }
                const uint32_t coef = p.exp[(p.log[d] + ORDER - p.log[bb]) % ORDER];
                if (2 * L <= k) {
                    for (int i = 0; i < 2 * T + 2; ++i) Tm[i] = C[i];
                    for (int i = 0; i + sh < 2 * T + 2; ++i) C[i + sh] ^= (uint16_t)mul(coef, B[i]);
                    L = k + 1 - L;
                    for (int i = 0; i < 2 * T + 2; ++i) B[i] = Tm[i];
                    bb = d;
                    sh = 1;
                } else {
                    for (int i = 0; i + sh < 2 * T + 2; ++i) C[i + sh] ^= (uint16_t)mul(coef, B[i]);
                    ++sh;
                }
            }
            for (int i = 0; i < 2 * T + 2; ++i) logC[i] = C[i] ? p.log[C[i]] : (uint16_t)0xffffu;
            ctl[0] = L;
            ctl[1] = 0;
        }
        __syncthreads();
        const int L = ctl[0];
        if (L <= T) {
            for (int pos = tid; pos < n; pos += kThreads) {
                const int e = n - 1 - pos, ne = e ? ORDER - e : 0;      // alpha^(-e); n <= ORDER for every T2 code
                uint32_t v = 1;                                         // C[0]
                for (int k = 1; k <= L; ++k)
                    if (logC[k] != 0xffffu) v ^= p.exp[(logC[k] + ne * k) % ORDER];
                if (v == 0) {
                    const int slot = atomicAdd(&ctl[1], 1);
                    if (slot < T) ctl[2 + slot] = pos;
                }
            }
        }
        __syncthreads();
        const bool ok = L <= T && ctl[1] == L;
        if (ok && tid < L) row[ctl[2 + tid]] ^= 1u;
        if (tid == 0) status[f] = ok ? L : -1;
    }
}

}  // namespace

hipError_t launch_bch_decode(uint8_t *bits, int n_frames, const BchDev &p, int32_t *status, hipStream_t s)
{
    if (n_frames < 1) return hipSuccess;
    const int grid = n_frames < 8192 ? n_frames : 8192;
    if (p.m == 16 && p.t == 12) hipLaunchKernelGGL((bch_decode_kernel<16, 12>), dim3(grid), dim3(kThreads), 0, s, bits, n_frames, p, status);
    else if (p.m == 16 && p.t == 10) hipLaunchKernelGGL((bch_decode_kernel<16, 10>), dim3(grid), dim3(kThreads), 0, s, bits, n_frames, p, status);
    else if (p.m == 14 && p.t == 12) hipLaunchKernelGGL((bch_decode_kernel<14, 12>), dim3(grid), dim3(kThreads), 0, s, bits, n_frames, p, status);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace t2gpu
