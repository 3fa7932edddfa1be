// ldpc_kernel2.hip -- the LDPC decoder: TWO FEC frames per workgroup (ldpc_cn3.h): frames 2m and 2m + 1 of a
// SIMD batch share a workgroup, their LLR bytes interleaved in LDS, every address / LDS access / DPP exchange serving both and the
// int8 arithmetic running in the 16-bit halves of the registers. Same mapping otherwise: two lanes per check node (720 of 768
// lanes), LLRs resident in LDS for the whole decode, per-link messages (one byte per link and frame, as in the reference) streaming
// through L2 one layer ahead, the reference's ascending-j order kept exactly in PAIR / GENERIC layers, SIMD batches of `group`
// frames stopping together through one atomic word per (batch, trial).
//
// Replaces the compute of ldpc_decoder::execute (/root/reference/src/DVB_T2/ldpc_decoder.cpp:157-301) and
// LDPCDecoder::{bad,update,operator()} (/root/reference/src/DVB_T2/LDPC/layered_decoder.hh:65-110,168-180).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ldpc_kernel.h"
#include <cstdlib>
#include "ldpc_cn3.h"

namespace t2gpu {
namespace {

typedef __attribute__((address_space(3))) int8_t lds2_i8;
struct LdsMem2 {
    uint32_t base;
    __device__ __forceinline__ int off() const { return (int)base; }
    __device__ __forceinline__ int8_t ld(int a) const { return *reinterpret_cast<const lds2_i8 *>((uint32_t)a); }
    __device__ __forceinline__ void st(int a, int8_t v) { *reinterpret_cast<lds2_i8 *>((uint32_t)a) = v; }
    // the LLR pair as ds_read_u16 delivers it; typed 16 bit so that no zero-extension is materialised (only bytes 0, 1 are ever read)
    __device__ __forceinline__ uint16_t ld16(int a) const
    {
        return *reinterpret_cast<const __attribute__((address_space(3))) uint16_t *>((uint32_t)a);
    }
    __device__ __forceinline__ uint16_t ld16_volatile(int a) const
    {
        return *reinterpret_cast<const volatile __attribute__((address_space(3))) uint16_t *>((uint32_t)a);
    }
    __device__ __forceinline__ void st16(int a, uint32_t v) { *reinterpret_cast<__attribute__((address_space(3))) uint16_t *>((uint32_t)a) = (uint16_t)v; }
    __device__ __forceinline__ uint32_t ld32(int a) const { return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>((uint32_t)a); }
    __device__ __forceinline__ void st32(int a, uint32_t v) { *reinterpret_cast<__attribute__((address_space(3))) uint32_t *>((uint32_t)a) = v; }
    __device__ __forceinline__ void st_pair(int a, uint32_t x, uint32_t y)
    {
        __attribute__((address_space(3))) uint32_t *q = reinterpret_cast<__attribute__((address_space(3))) uint32_t *>((uint32_t)a);
        q[0] = x; q[1] = y;
    }
    __device__ __forceinline__ uint2 ld_pair(int a) const
    {
        const __attribute__((address_space(3))) uint32_t *q = reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>((uint32_t)a);
        return make_uint2(q[0], q[1]);
    }
};

constexpr int kThreads2 = 768;

__device__ __forceinline__ void lds_barrier2() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- bit-parallel parity check of both frames (LDPCDecoder::bad, layered_decoder.hh:65-82) -----------------
// Sign words as there: 13 dwords per 360-bit group (bits 0..359 + a copy of bits 0..55), one array per frame. A dword of the
// interleaved LLR bytes holds (A_k, B_k, A_k+1, B_k+1).
__device__ __forceinline__ uint32_t has_zero_byte2(uint32_t v) { return (v - 0x01010101u) & ~v & 0x80808080u; }
__device__ __forceinline__ void sign_bits2(uint32_t v, uint32_t &a2, uint32_t &b2)
{
    const uint32_t t = (v >> 7) & 0x01010101u;
    const uint32_t a = t & 0x00010001u, b = (t >> 8) & 0x00010001u;
    a2 = (a | (a >> 15)) & 3u;
    b2 = (b | (b >> 15)) & 3u;
}
__device__ __forceinline__ uint32_t sign_window2(const uint32_t *S, int g, int m)
{
    const uint32_t *w = S + g * 13 + (m >> 5);
    return __builtin_amdgcn_alignbit(w[1], w[0], (uint32_t)(m & 31));
}
// sign words (bits 32 kk .. 32 kk + 31 of group g) of both frames; zero bits 7/23 = frame A has an exactly-zero LLR, 15/31 = frame B
__device__ __forceinline__ void sign_word2(const int8_t *Lm, int g, int kk, uint32_t &wa, uint32_t &wb, uint32_t *zero)
{
    const uint4 *src = reinterpret_cast<const uint4 *>(Lm + 2 * (g * 360 + 32 * kk));     // 64 bytes = 32 bits of either frame
    const int nq = (kk == 11) ? 1 : 4;          // dword 11 holds only bits 352..359: 16 bytes
    wa = 0; wb = 0;
#pragma unroll
    for (int x = 0; x < 4; ++x)
        if (x < nq) {
            const uint4 v = src[x];
            const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                uint32_t a2, b2;
                *zero |= has_zero_byte2(d[y]);
                sign_bits2(d[y], a2, b2);
                wa |= a2 << (8 * x + 2 * y);
                wb |= b2 << (8 * x + 2 * y);
            }
        }
}
__device__ __forceinline__ uint32_t layer_syndrome2(const uint32_t *S, const LdpcLayerDev &ly, const uint32_t *__restrict__ ent, int gp0, int q,
                                                    int i, int tj)
{
    const int j0 = 32 * tj;
    uint32_t syn = S[(gp0 + i) * 13 + tj];
    if (i > 0) syn ^= S[(gp0 + i - 1) * 13 + tj];
    else syn ^= (tj == 0) ? (S[(gp0 + q - 1) * 13] << 1) : sign_window2(S, gp0 + q - 1, j0 - 1);
    for (int c = 0; c < ly.cnt; ++c) {
        const uint32_t e = ent[c];
        const int g = (int)__umulhi(e & 0xffffu, 11930465u);                 // base / 360
        int m = j0 - (int)(e >> 16);
        m += (m < 0) ? 360 : 0;
        syn ^= sign_window2(S, g, m);
    }
    return syn & ((tj == 11) ? 0xffu : 0xffffffffu);
}
// the same with the layer's table entries taken from their LDS copy ((lds_base + 2 (bit base - shift), shift) pairs): no memory
// round trip inside the check
__device__ __forceinline__ int ent_group_lds(const uint32_t *lds_ent, int lds_base, int c, int *shift)
{
    const int x = (int)lds_ent[2 * c], sh = (int)lds_ent[2 * c + 1];
    *shift = sh;
    return (int)__umulhi((uint32_t)(((x - lds_base) >> 1) + sh), 11930465u);   // bit base / 360
}
__device__ __forceinline__ uint32_t layer_syndrome2_lds(const uint32_t *S, int cnt, const uint32_t *lds_ent, int lds_base, int gp0, int q, int i, int tj)
{
    const int j0 = 32 * tj;
    uint32_t syn = S[(gp0 + i) * 13 + tj];
    if (i > 0) syn ^= S[(gp0 + i - 1) * 13 + tj];
    else syn ^= (tj == 0) ? (S[(gp0 + q - 1) * 13] << 1) : sign_window2(S, gp0 + q - 1, j0 - 1);
    for (int c = 0; c < cnt; ++c) {
        int sh;
        const int g = ent_group_lds(lds_ent, lds_base, c, &sh);
        int m = j0 - sh;
        m += (m < 0) ? 360 : 0;
        syn ^= sign_window2(S, g, m);
    }
    return syn & ((tj == 11) ? 0xffu : 0xffffffffu);
}
__device__ __forceinline__ void finish_group2(uint32_t *S, int g)
{
    const uint32_t d0 = S[g * 13], d1 = S[g * 13 + 1];
    S[g * 13 + 11] = (S[g * 13 + 11] & 0xffu) | (d0 << 8);
    S[g * 13 + 12] = (d0 >> 24) | (d1 << 8);
}

// per thread: bit 0 = frame A saw a zero LLR or a failing check, bit 1 = frame B; bit 2 = bits 0 / 1 are already the workgroup's
// verdict (the probe failed for both frames). probe_first / probe_cnt: first table entry and link count of layer 1; lds_ent: the
// LDS copy of the table entries, lds_base: the LDS address of the LLR array they were made for.
__device__ __forceinline__ int frames_parity_bad(const int8_t *Lm, uint32_t *SA, uint32_t *SB, const LdpcLayerDev *__restrict__ layers,
                                                 const uint32_t *__restrict__ entries, int n, int k, int q, int tid, int *s_ctl,
                                                 int probe_first, int probe_cnt, const uint32_t *lds_ent, int lds_base, int trial)
{
    const int ngroups = n / 360, gp0 = k / 360;
    uint32_t zero = 0;
    int probe = 0;
    // Before the probe, its first 32 checks alone (nodes 0..31 of layer 1): one lane per (group, node) reads the LLR pair that node
    // sees in that group, a wavefront ballot IS the 32-bit sign window of two groups, the windows are XOR-ed into one LDS word per
    // frame -- one LDS read per lane and one barrier. A frame pair that is still far from a codeword fails here in both frames and
    // is done; only otherwise the probe proper (three barriers) and the full check run. Two accumulator pairs, used alternately:
    // the one for the next sweep is cleared behind this sweep's barrier.
    if (q > 1 && probe_cnt + 2 <= kThreads2 / 32) {
        const uint32_t *ent = lds_ent + 2 * probe_first;
        int *acc = s_ctl + 4 + 2 * (trial & 1);
        const int gi = tid >> 5, i = tid & 31;
        uint32_t v = 0;
        if (gi < probe_cnt + 2) {
            int sh = 0;
            const int g = gi < probe_cnt ? ent_group_lds(ent, lds_base, gi, &sh) : gp0 + (gi - probe_cnt);
            int b = (sh ? 360 - sh : 0) + i;                         // node i sees bit (i - shift) mod 360 of the group
            b -= b >= 360 ? 360 : 0;
            v = *reinterpret_cast<const uint16_t *>(Lm + 2 * (360 * g + b));
        }
        const unsigned long long ba = __ballot(v & 0x0080u), bb = __ballot(v & 0x8000u);
        if ((tid & 63) == 0) {
            const uint32_t wa = (uint32_t)ba ^ (uint32_t)(ba >> 32), wb = (uint32_t)bb ^ (uint32_t)(bb >> 32);
            if (wa) atomicXor(reinterpret_cast<unsigned *>(acc), wa);
            if (wb) atomicXor(reinterpret_cast<unsigned *>(acc) + 1, wb);
        }
        lds_barrier2();
        const int fail_a = acc[0] != 0, fail_b = acc[1] != 0;
        if (tid == 0) { s_ctl[4 + 2 * ((trial + 1) & 1)] = 0; s_ctl[5 + 2 * ((trial + 1) & 1)] = 0; }
        if (fail_a && fail_b) return 3 | 4;
    }
    // Probe: the 360 checks of layer 1 need the sign words of ~14 groups only, and a frame that has not converged
    // almost always fails there. When BOTH frames fail the probe the workgroup is done; otherwise the full check runs for both.
    // The probe runs once per sweep for every frame pair that is still decoding: it reads nothing from memory (table entries from
    // their LDS copy) and ends in ONE barrier (the wavefronts' verdicts are OR-ed into an LDS word).
    if (q > 1) {
        const uint32_t *ent = lds_ent + 2 * probe_first;
        const int ng = probe_cnt + 2;
        if (tid == 0) s_ctl[2] = 0;
        if (tid < ng * 12) {
            const int gi = tid / 12, kk = tid - gi * 12;
            int sh;
            const int g = gi < probe_cnt ? ent_group_lds(ent, lds_base, gi, &sh) : gp0 + (gi - probe_cnt);
            uint32_t wa, wb;
            sign_word2(Lm, g, kk, wa, wb, &zero);
            SA[g * 13 + kk] = wa; SB[g * 13 + kk] = wb;
        }
        lds_barrier2();
        if (tid < ng) {
            int sh;
            const int g = tid < probe_cnt ? ent_group_lds(ent, lds_base, tid, &sh) : gp0 + (tid - probe_cnt);
            finish_group2(SA, g); finish_group2(SB, g);
        }
        lds_barrier2();
        int bad = ((zero & 0x00800080u) ? 1 : 0) | ((zero & 0x80008000u) ? 2 : 0);
        if (tid < 12) {
            if (layer_syndrome2_lds(SA, probe_cnt, ent, lds_base, gp0, q, 1, tid)) bad |= 1;
            if (layer_syndrome2_lds(SB, probe_cnt, ent, lds_base, gp0, q, 1, tid)) bad |= 2;
        }
        const int wave_bad = (__ballot(bad & 1) ? 1 : 0) | (__ballot(bad & 2) ? 2 : 0);
        if ((tid & 63) == 0 && wave_bad) atomicOr(&s_ctl[2], wave_bad);
        lds_barrier2();
        probe = s_ctl[2];
        if (probe == 3) return 3 | 4;
    }
    (void)s_ctl;
    for (int task = tid; task < ngroups * 12; task += kThreads2) {
        const int g = task / 12, kk = task - g * 12;
        uint32_t wa, wb;
        sign_word2(Lm, g, kk, wa, wb, &zero);
        SA[g * 13 + kk] = wa; SB[g * 13 + kk] = wb;
    }
    lds_barrier2();
    for (int g = tid; g < ngroups; g += kThreads2) { finish_group2(SA, g); finish_group2(SB, g); }
    lds_barrier2();
    int bad = probe | ((zero & 0x00800080u) ? 1 : 0) | ((zero & 0x80008000u) ? 2 : 0);
    for (int task = tid; task < q * 12; task += kThreads2) {
        const int i = task / 12, tj = task - i * 12;
        const LdpcLayerDev ly = layers[i];
        if (layer_syndrome2(SA, ly, entries + ly.first_entry, gp0, q, i, tj)) bad |= 1;
        if (layer_syndrome2(SB, ly, entries + ly.first_entry, gp0, q, i, tj)) bad |= 2;
    }
    return bad;
}

#ifndef T2_PAIR_SEG_MIN
#define T2_PAIR_SEG_MIN 12                 // chains at least this long are walked in segments (cut where a node's output ignores its input)
#endif
template <int CNT, int NCMAX, bool PF>
__device__ __forceinline__ void layer_update2(LdsMem2 &L, const LayerDesc &d, uint2 (&e)[(CNT + 3) / 2], int j, int h, bool active, int a_own, int a_prev,
                                              bool prev_absent, uint32_t info, P2Regs<CNT> &r, uint32_t *pair_rec, int next_ent_lds)
{
    // the one parity bit this lane reads (ldpc_cn3.h, p2_load): the previous layer's on the odd lane of an even link count / the even
    // lane of an odd one, the node's own on the other
    const bool prev_lane = (CNT % 2 == 0) == (h == 1);
    if (active) p2_phase_a<CNT, PF>(L, d, e, j, h, prev_lane ? a_prev : a_own, prev_lane && prev_absent, r, pair_rec, next_ent_lds);
    if (d.kind == T2_LAYER_PAIR) {
        lds_barrier2();
        __builtin_amdgcn_s_setprio(3);
        // frame A's chains on lanes 0..359, frame B's on lanes 384..743 (whole wavefronts apart), side by side
        const int t = (int)threadIdx.x, frame = t >= 384 ? 1 : 0, node = t - 384 * frame;
        if (node < 360) {
            if (d.lmax >= T2_PAIR_SEG_MIN) p2_pair_walk_segments(L, d, node, frame, pair_rec + 360 * frame);
            else if (node < d.step) p2_pair_walk(L, d, node, frame, pair_rec + 360 * frame);
        }
        __builtin_amdgcn_s_setprio(0);
        if (active) p2_pair_finish<CNT>(L, d, j, r);       // polls the walks' ready flags: no barrier here
    } else if (d.kind == T2_LAYER_GENERIC && d.band) {
        if constexpr (NCMAX >= 3) {
            lds_barrier2();
            __builtin_amdgcn_s_setprio(3);
            // frame A's bands on the first lanes of wavefront 0, frame B's on those of wavefront 6 (another SIMD)
            const int t = (int)threadIdx.x, frame = t >= 384 ? 1 : 0, lane = t - 384 * frame;
#define T2_BAND_(NC_, LPN_)                                                          \
    do {                                                                             \
        if (lane < LPN_ * d.band) {                                                  \
            if (d.band_prefetch) p2_band_walk<NC_, LPN_, true>(L, d, lane, frame);   \
            else p2_band_walk<NC_, LPN_, false>(L, d, lane, frame);                  \
        }                                                                            \
    } while (0)
            if (d.nc == 3) {
                if (d.band <= 16) T2_BAND_(3, 4);
                else T2_BAND_(3, 2);
            } else if constexpr (NCMAX >= 4) {
                if (d.band <= 16) T2_BAND_(4, 4);
                else T2_BAND_(4, 2);
            }
#undef T2_BAND_
            __builtin_amdgcn_s_setprio(0);
            lds_barrier2();
            if (active) {
                if (d.nc == 3) p2_band_finish<CNT, 3>(L, d, j, r);
                else if constexpr (NCMAX >= 4) p2_band_finish<CNT, 4>(L, d, j, r);
            }
        }
    } else if (d.kind == T2_LAYER_GENERIC) {
        __builtin_amdgcn_s_setprio(3);
        for (int lv = 1; lv <= d.lmax; ++lv) {
            if (active) p2_generic_level<CNT, NCMAX>(L, d, lv, info, r);
            lds_barrier2();
        }
        __builtin_amdgcn_s_setprio(0);
        if (active) p2_generic_finish<CNT>(L, d, r);
    }
    if (!d.no_close) lds_barrier2();
}

// one layer for a compile-time link count: record in, update, record out. RW = record dwords per lane in memory (>= P2Regs::W).
// The new record is handed back in registers (rec_new): the caller stores it one layer later, so that whatever the vector-memory
// counter still holds at the head of a layer was issued a whole layer ago.
// UNI (every layer of the code has CNT links): epf holds this layer's table entries on entry and the NEXT layer's on return -- their
// LDS reads are issued right behind this layer's LLR reads (p2_load) and so cost no round trip of their own (with one workgroup per
// CU there is no neighbour to fill a wavefront's waits).
template <int CNT, int NCMAX, bool UNI, int RW>
__device__ __forceinline__ void layer_step2(LdsMem2 &L, const LayerDesc &d, int j, int h, bool active, int a_own, int a_prev, bool prev_absent, uint32_t info,
                                            const uint32_t *rec_in, uint32_t (&rec_new)[RW], uint32_t *pair_rec,
                                            uint2 (&epf)[(CNT + 3) / 2], int next_ent_lds)
{
    P2Regs<CNT> r;
    constexpr int W = P2Regs<CNT>::W, H = P2Regs<CNT>::H;
    static_assert(W <= RW, "record stride");
#pragma unroll
    for (int w = 0; w < W; ++w) r.mo[w] = rec_in[w];
#pragma unroll
    for (int w = 0; w < W; ++w) r.mn[w] = 0u;
    if constexpr (UNI) {
        layer_update2<CNT, NCMAX, true>(L, d, epf, j, h, active, a_own, a_prev, prev_absent, info, r, pair_rec, next_ent_lds);
    } else {
        uint2 e[H];
        if (active) p2_entries<CNT>(L, d.ent_lds, h, e);
        layer_update2<CNT, NCMAX, false>(L, d, e, j, h, active, a_own, a_prev, prev_absent, info, r, pair_rec, 0);
    }
#pragma unroll
    for (int w = 0; w < RW; ++w) rec_new[w] = w < W ? r.mn[w] : 0u;
}

// The layer table in registers: lane l of every wavefront holds the four packed dwords of layer base + l (LdpcKernelParams::layer_words);
// a layer's description is four v_readlane_b32 with the (uniform) layer number as the lane select.
__device__ __forceinline__ uint4 layer_words_load(const uint4 *__restrict__ words, int q, int base)
{
    const int l = base + (int)(threadIdx.x & 63u);
    return l < q ? words[l] : make_uint4(0u, 0u, 0u, 0u);
}
__device__ __forceinline__ void layer_words_get(const uint4 &dsc, int lane, uint32_t (&w)[4])
{
    w[0] = (uint32_t)__builtin_amdgcn_readlane((int)dsc.x, lane);
    w[1] = (uint32_t)__builtin_amdgcn_readlane((int)dsc.y, lane);
    w[2] = (uint32_t)__builtin_amdgcn_readlane((int)dsc.z, lane);
    w[3] = (uint32_t)__builtin_amdgcn_readlane((int)dsc.w, lane);
}
template <int RW>
__device__ __forceinline__ void store_record(uint32_t *rec_out, const uint32_t (&v)[RW])
{
    *reinterpret_cast<uint4 *>(rec_out) = make_uint4(v[0], v[1], v[2], v[3]);
    if constexpr (RW > 4) *reinterpret_cast<uint4 *>(rec_out + 4) = make_uint4(v[4], v[5], v[6], v[7]);
}

#define T2_PROF2_T(var) long long var = p.prof ? (long long)__builtin_readcyclecounter() : 0
#define T2_PROF2_ADD(slot, t0)                                                                                  \
    do {                                                                                                        \
        if (p.prof && threadIdx.x == 0) p.prof[blockIdx.x * 8 + (slot)] += (long long)__builtin_readcyclecounter() - (t0); \
    } while (0)

template <int LO, int HI, int NCMAX = T2_LDPC_NC_MAX>
__global__ __launch_bounds__(kThreads2, 3) void ldpc_decode2_kernel(const LdpcLayerDev *__restrict__ layers, const uint32_t *__restrict__ entries,
                                                                   const uint32_t *__restrict__ cninfo, const uint32_t *__restrict__ entries2,
                                                                   LdpcKernelParams p)
{
    extern __shared__ __attribute__((aligned(16))) int8_t lds[];
    int8_t *Lm = lds;                                                // [n][2] interleaved LLR bytes
    int *s_ctl = reinterpret_cast<int *>(lds + p.lds_ctl_offset);
    uint32_t *pair_rec = reinterpret_cast<uint32_t *>(lds + p.lds_rec_offset);     // [2][360]
    uint32_t *SA = reinterpret_cast<uint32_t *>(lds + p.lds_sign_offset), *SB = SA + (p.n / 360) * 13;
    uint32_t *lds_ent = reinterpret_cast<uint32_t *>(lds + p.lds_ent_offset);
    for (int x = threadIdx.x; x < 2 * p.n_entries; x += kThreads2) lds_ent[x] = entries2[x];      // made visible by the first barrier below
    LdsMem2 L{(uint32_t)(uintptr_t)(lds2_i8 *)Lm};
    const int probe_first = p.q > 1 ? layers[1].first_entry : 0, probe_cnt = p.q > 1 ? layers[1].cnt : 0;   // the parity probe's layer
    uint4 dsc = layer_words_load(p.layer_words, p.q, 0);
    const int ent_lds0 = L.off() + p.lds_ent_offset;

    const int tid = threadIdx.x;
    const int j = tid >> 1, h = tid & 1;
    const bool active = j < 360;
    const uint32_t rec_lane_off = (uint32_t)threadIdx.x * (uint32_t)(((HI + 2 + 1) / 2 > 8 ? 8 : 4) * 4);   // byte offset of the lane's record inside a layer's block
    const int par0 = L.off() + 2 * (p.k + j);                        // LDS address of the parity bit pair of node j in layer 0
    if (L.off() != p.lds_base) {                     // the split table was built for another LDS layout: refuse, loudly
        if (tid == 0) *p.error = 2;
        return;
    }
    constexpr int RW = (HI + 2 + 1) / 2 > 8 ? 8 : 4;                 // record dwords per lane in memory: W <= 4 -> 4, else 8
    // an odd group (group = 1: independent frames) gives its last workgroup ONE frame: the B halves then carry a copy of frame A whose
    // results are not stored (rounds 1-4 kept a second, one-frame-per-workgroup kernel for such groups)
    const int group = p.group, wg_per_batch = (group + 1) >> 1;
    const int slot = blockIdx.x / wg_per_batch, member = blockIdx.x % wg_per_batch;
    const int nslots = gridDim.x / wg_per_batch;
    const int nbatches = (p.n_frames + group - 1) / group;
    uint32_t *state = reinterpret_cast<uint32_t *>(p.state) + (size_t)blockIdx.x * p.q * 720 * RW;

    if (p.prof && tid == 0) p.prof[blockIdx.x * 8 + 6] = wall_clock64();
    if (tid == 0) __hip_atomic_fetch_add(p.resident, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    for (int round = 0;; ++round) {
        // Which batch next: by ticket -- the slot's first workgroup draws the next unclaimed batch and posts it, its siblings read
        // the post -- so that slots whose batches stop early take more of them; without tickets, batch = slot + round * slots.
        int batch = slot + round * nslots;
        if (p.ticket) {
            if (tid == 0) {
                unsigned *w = p.ticket + 1 + (size_t)slot * p.ticket_rounds + round;
                unsigned v = 0;
                if (round >= p.ticket_rounds) v = 0x7fffffffu;
                else if (member == 0) {
                    v = atomicAdd(p.ticket, 1u) + 1u;
                    __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    const long long t0 = wall_clock64();
                    for (;;) {
                        v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (v) break;
                        if (__hip_atomic_load(p.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { v = 0x7fffffffu; break; }
                        if (wall_clock64() - t0 > p.spin_timeout_ticks) {
                            __hip_atomic_store(p.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            v = 0x7fffffffu;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(16);
                    }
                }
                s_ctl[1] = (int)(v - 1u);
            }
            __syncthreads();
            batch = s_ctl[1];
            __syncthreads();
        }
        if (batch >= nbatches) break;
        const int frame_a = batch * group + 2 * member, frame_b = frame_a + 1;
        const int batch_end = (batch + 1) * group < p.n_frames ? (batch + 1) * group : p.n_frames;
        const bool have_a = frame_a < batch_end, have_b = frame_b < batch_end;
        const int nhave = (have_a ? 1 : 0) + (have_b ? 1 : 0);
        int members = p.n_frames - batch * group;
        members = members > group ? group : members;

        if (have_a) {
            // interleave the two frames' LLR bytes: (a0 a1 a2 a3), (b0 b1 b2 b3) -> (a0 b0 a1 b1), (a2 b2 a3 b3)
            const uint32_t *sa = reinterpret_cast<const uint32_t *>(p.llr + (size_t)frame_a * p.n);
            const uint32_t *sb = reinterpret_cast<const uint32_t *>(p.llr + (size_t)(have_b ? frame_b : frame_a) * p.n);
            uint2 *dst = reinterpret_cast<uint2 *>(Lm);
            for (int x = tid; x < p.n / 4; x += kThreads2) {
                const uint32_t a = sa[x], b = sb[x];
                dst[x] = make_uint2(__builtin_amdgcn_perm(b, a, 0x05010400u), __builtin_amdgcn_perm(b, a, 0x07030602u));
            }
            // (no zeroing of the message records: the first sweep of a batch takes every old message as 0 without reading them -- 518 KB
            // per workgroup and batch neither written nor read back)
        }
        if (tid < 4) s_ctl[4 + tid] = 0;                 // the parity check's 32-check accumulators
        __syncthreads();

        int trials = p.max_trials;
        int result;
        for (int t = 0;; ++t) {
            // the first layer's records of the coming sweep are fetched ahead of the parity check (they were written a sweep ago;
            // read at the head of the sweep, all twelve wavefronts sat out the L2 round trip together)
            uint32_t first_rec[RW];
            if (have_a && active && t > 0) {
                const uint4 *q4 = reinterpret_cast<const uint4 *>(state + (size_t)tid * RW);
#pragma unroll
                for (int w = 0; w < RW / 4; ++w) { const uint4 v = q4[w]; first_rec[4 * w] = v.x; first_rec[4 * w + 1] = v.y; first_rec[4 * w + 2] = v.z; first_rec[4 * w + 3] = v.w; }
            } else {
#pragma unroll
                for (int w = 0; w < RW; ++w) first_rec[w] = 0u;
            }
            // ---- parity check of both frames (LDPCDecoder::bad)
            T2_PROF2_T(tp0);
            const int bad = have_a ? frames_parity_bad(Lm, SA, SB, layers, entries, p.n, p.k, p.q, tid, s_ctl, probe_first, probe_cnt, lds_ent,
                                                       L.off(), t) : 0;
            int bad_a, bad_b;
            if (bad & 4) { bad_a = bad & 1; bad_b = bad & 2; }       // uniform: the probe's verdict, already the workgroup's
            else { bad_a = __syncthreads_or(bad & 1); bad_b = __syncthreads_or(bad & 2); }
            const int clean = (have_a && !bad_a ? 1 : 0) + (have_b && !bad_b ? 1 : 0);
            int all_ok = clean == nhave;
            T2_PROF2_ADD(0, tp0);
            T2_PROF2_T(tp1);
            if (wg_per_batch > 1 && have_a) {
                // one word per (batch, trial): high half counts frames arrived, low half counts parity-clean frames
                if (tid == 0) {
                    unsigned *w = p.sync + (size_t)batch * (p.max_trials + 1) + t;
                    atomicAdd(w, ((unsigned)nhave << 16) | (unsigned)clean);
                    int verdict = -1;
                    // a workgroup with a frame that still fails knows the verdict without waiting: the batch goes on
                    if (!all_ok) verdict = 0;
                    else {
                        const long long t0 = wall_clock64();
                        for (;;) {
                            const unsigned v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if ((int)(v >> 16) >= members) { verdict = ((int)(v & 0xffffu) == members); break; }
                            if (__hip_atomic_load(p.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                            if (wall_clock64() - t0 > p.spin_timeout_ticks) {
                                __hip_atomic_store(p.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                break;
                            }
                            __builtin_amdgcn_s_sleep(8);
                        }
                    }
                    s_ctl[0] = verdict;
                }
                __syncthreads();
                const int verdict = s_ctl[0];
                __syncthreads();
                if (verdict < 0) { result = -3; break; }
                all_ok = verdict;
            }
            T2_PROF2_ADD(1, tp1);
            if (all_ok) { result = trials; break; }
            if (--trials < 0) { result = -1; break; }

            // ---- one layered update sweep (LDPCDecoder::update)
            if (have_a) {
                uint32_t nxt[RW];
#pragma unroll
                for (int w = 0; w < RW; ++w) nxt[w] = first_rec[w];
                uint32_t lw[4];
                layer_words_get(dsc, 0, lw);
                uint32_t info_nxt = (active && (lw[0] & 3u) == T2_LAYER_GENERIC) ? cninfo[j] : 0u;
                constexpr bool UNI = LO == HI;
                uint2 epf[(HI + 3) / 2];
                if constexpr (UNI) {
                    if (active) p2_entries<HI>(L, ent_lds0 + 8 * (int)(lw[1] >> 16), h, epf);
                }
                uint32_t held[RW];                                          // layer i - 1's new record, stored at the head of layer i
#pragma unroll
                for (int w = 0; w < RW; ++w) held[w] = 0u;
                for (int i0 = 0; i0 < p.q; i0 += 64) {
                if (i0) dsc = layer_words_load(p.layer_words, p.q, i0);       // codes of more than 64 layers: the table 64 layers at a time
                const int i1 = i0 + 64 < p.q ? i0 + 64 : p.q;
                for (int i = i0; i < i1; ++i) {
                    // the layer's description: four dwords out of registers (v_readlane), nothing from memory (the scalar loads this
                    // replaces -- layers[i], then entries[first_entry], then layers[i + 1] -- were three dependent round trips that all
                    // twelve wavefronts sat out together right behind the barrier, every layer)
                    layer_words_get(dsc, i - i0, lw);
                    const int kind = (int)(lw[0] & 3u), cnt = (int)((lw[0] >> 2) & 31u), nc = (int)((lw[0] >> 7) & 31u), lmax = (int)((lw[0] >> 12) & 511u);
                    const int first_entry = (int)(lw[1] >> 16);
                    LayerDesc d{entries + first_entry, cnt, lmax, nc, kind, (int)(lw[1] & 0xffffu), L.off() + p.lds_ctl_offset + 32, lw[2],
                                entries2 + 2 * first_entry, ent_lds0 + 8 * first_entry};
                    d.band = (nc <= NCMAX && nc <= cnt) ? (int)((lw[0] >> 23) & 63u) : 0; d.band_prefetch = (int)((lw[0] >> 21) & 1u); d.no_close = (int)((lw[0] >> 22) & 1u);
                    d.band_rec_lds = L.off() + p.lds_sign_offset; d.band_in_lds = L.off() + p.lds_rec_offset;
                    d.pair_flag_lds = L.off() + p.lds_sign_offset;
                    // this layer's record and node number were fetched a layer ago: consume them HERE, ahead of this head's own loads and
                    // store -- a wait placed behind those would wait for them too
                    uint32_t cur[RW];
#pragma unroll
                    for (int w = 0; w < RW; ++w) cur[w] = nxt[w];
                    uint32_t info = info_nxt;
                    if constexpr (RW == 4) asm volatile("" :: "v"(cur[0]), "v"(cur[1]), "v"(cur[2]), "v"(cur[3]), "v"(info) : "memory");
                    else asm volatile("" :: "v"(cur[0]), "v"(cur[1]), "v"(cur[2]), "v"(cur[3]), "v"(cur[4]), "v"(cur[5]), "v"(cur[6]), "v"(cur[7]), "v"(info) : "memory");
                    // the node this lane pair takes and its two parity bits: lane constants plus 720 i, except in the layers that deal
                    // their nodes by dependency level (GENERIC without a band walk)
                    // (the two rare cases are real branches -- both conditions are uniform -- with an empty asm inside so that they are not
                    // turned into selects every layer pays for)
                    int jn = j, a_own = par0 + 720 * i;
                    if (kind == T2_LAYER_GENERIC && !d.band) { asm volatile(""); jn = (int)(info >> 20); a_own += 2 * (jn - j); }
                    int a_prev = a_own - 720;
                    bool prev_absent = false;
                    if (i == 0) {                                                    // the wrap: the last layer's bit of node j - 1; none for node 0
                        asm volatile("");
                        prev_absent = jn == 0;
                        a_prev = prev_absent ? d.dummy : a_own + 720 * (p.q - 1) - 2;
                    }
                    if (active) {
                        if (i + 1 < p.q) {                                           // prefetch the next layer's record (and node)
                            if (t > 0) {
                                // (a uniform base + the lane's 32-bit offset: the scalar-base addressing form, no 64-bit vector arithmetic per layer)
                                const uint4 *q4 = reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(state) + (size_t)(i + 1) * (720 * RW * 4) + rec_lane_off);
#pragma unroll
                                for (int w = 0; w < RW / 4; ++w) { const uint4 v = q4[w]; nxt[4 * w] = v.x; nxt[4 * w + 1] = v.y; nxt[4 * w + 2] = v.z; nxt[4 * w + 3] = v.w; }
                            } else {
#pragma unroll
                                for (int w = 0; w < RW; ++w) nxt[w] = 0u;                 // first sweep of a batch: every old message is 0
                            }
                            info_nxt = (lw[0] >> 29) & 1u ? cninfo[(i + 1) * 360 + j] : 0u;
                        }
                        if (i > 0) store_record<RW>(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(state) + (size_t)(i - 1) * (720 * RW * 4) + rec_lane_off), held);
                    }
                    T2_PROF2_T(tp2);
                    const int next_ent = ent_lds0 + 8 * (int)lw[3];                  // (the last layer names layer 0: the next sweep starts there)
                    if constexpr (UNI) {
                        layer_step2<HI, NCMAX, true, RW>(L, d, jn, h, active, a_own, a_prev, prev_absent, info, cur, held, pair_rec, epf, next_ent);
                    } else {
                        T2_LDPC_DISPATCH_RANGE(cnt, LO, HI, ({ uint2 none[(CNT + 3) / 2]; layer_step2<CNT, NCMAX, false, RW>(L, d, jn, h, active, a_own, a_prev, prev_absent, info, cur, held, pair_rec, none, 0); }));
                    }
                    T2_PROF2_ADD(2 + kind, tp2);
                    if (p.prof && blockIdx.x == 0 && tid == 0 && i < 64) p.prof[(size_t)p.prof_blocks * 8 + i] += (long long)__builtin_readcyclecounter() - tp2;
                }
                }
                if (active) store_record<RW>(state + ((size_t)(p.q - 1) * 720 + tid) * RW, held);
                if (p.q > 64) dsc = layer_words_load(p.layer_words, p.q, 0);
            }
            __syncthreads();   // once per sweep: the records written above are re-read by the same thread next sweep
        }

        // ---- outputs: hard decision of the information bits (ldpc_decoder.cpp:270-277), one bit per byte, frame by frame
        if (have_a) {
            const uint2 *src = reinterpret_cast<const uint2 *>(Lm);
            if (p.bits) {
                uint32_t *oa = reinterpret_cast<uint32_t *>(p.bits + (size_t)frame_a * p.k);
                uint32_t *ob = reinterpret_cast<uint32_t *>(p.bits + (size_t)(have_b ? frame_b : frame_a) * p.k);
                for (int x = tid; x < p.k / 4; x += kThreads2) {
                    const uint2 v = src[x];                          // (a0 b0 a1 b1), (a2 b2 a3 b3)
                    const uint32_t a = __builtin_amdgcn_perm(v.y, v.x, 0x06040200u), b = __builtin_amdgcn_perm(v.y, v.x, 0x07050301u);
                    oa[x] = (a >> 7) & 0x01010101u;
                    if (have_b) ob[x] = (b >> 7) & 0x01010101u;
                }
            }
            if (p.llr_out) {
                uint32_t *oa = reinterpret_cast<uint32_t *>(p.llr_out + (size_t)frame_a * p.n);
                uint32_t *ob = reinterpret_cast<uint32_t *>(p.llr_out + (size_t)(have_b ? frame_b : frame_a) * p.n);
                for (int x = tid; x < p.n / 4; x += kThreads2) {
                    const uint2 v = src[x];
                    oa[x] = __builtin_amdgcn_perm(v.y, v.x, 0x06040200u);
                    if (have_b) ob[x] = __builtin_amdgcn_perm(v.y, v.x, 0x07050301u);
                }
            }
            if (tid == 0 && member == 0) p.trials_left[batch] = result;
        }
        __syncthreads();
    }
    if (p.prof && tid == 0) p.prof[blockIdx.x * 8 + 7] = wall_clock64();
}

typedef void (*ldpc2_kernel_fn)(const LdpcLayerDev *, const uint32_t *, const uint32_t *, const uint32_t *, LdpcKernelParams);
ldpc2_kernel_fn pick_kernel2(int min_cnt, int max_cnt)
{
    if (min_cnt == max_cnt) {
        switch (max_cnt) {
        case 5: return ldpc_decode2_kernel<5, 5, 2>;      // N 1/2
        case 8: return ldpc_decode2_kernel<8, 8, 4>;      // N 2/3
        case 9: return ldpc_decode2_kernel<9, 9, 4>;      // N 3/5
        case 12: return ldpc_decode2_kernel<12, 12, 4>;   // N 3/4
        case 16: return ldpc_decode2_kernel<16, 16, 7>;   // N 4/5
        case 20: return ldpc_decode2_kernel<20, 20, 6>;   // N 5/6
        default: break;
        }
    }
    if (max_cnt <= 8) return ldpc_decode2_kernel<1, 8>;
    if (min_cnt >= 7 && max_cnt <= 12) return ldpc_decode2_kernel<7, 12>;
    if (min_cnt >= 13 && max_cnt <= 17) return ldpc_decode2_kernel<13, 17>;
    if (min_cnt >= 16 && max_cnt <= 20) return ldpc_decode2_kernel<16, 20>;
    return ldpc_decode2_kernel<1, 20>;
}

}  // namespace

// link-count ceiling of the variant pick_kernel2 chooses (its HI): fixes the record stride in memory
static int picked_hi(int min_cnt, int max_cnt)
{
    if (min_cnt == max_cnt && (max_cnt == 5 || max_cnt == 8 || max_cnt == 9 || max_cnt == 12 || max_cnt == 16 || max_cnt == 20)) return max_cnt;
    if (max_cnt <= 8) return 8;
    if (min_cnt >= 7 && max_cnt <= 12) return 12;
    if (min_cnt >= 13 && max_cnt <= 17) return 17;
    return 20;
}
int ldpc_kernel2_record_dwords(int min_cnt, int max_cnt) { return (picked_hi(min_cnt, max_cnt) + 2 + 1) / 2 > 8 ? 8 : 4; }

hipError_t ldpc_kernel2_attributes(int min_cnt, int max_cnt, int lds_bytes, int *blocks_per_cu, int *static_lds_bytes)
{
    ldpc2_kernel_fn fn = pick_kernel2(min_cnt, max_cnt);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return e;
    hipFuncAttributes attr;
    if ((e = hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(fn))) != hipSuccess) return e;
    *static_lds_bytes = (int)attr.sharedSizeBytes;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, fn, kThreads2, lds_bytes);
}

// The results of a decode into page-locked host memory by the device's own stores, in the decode's stream (t2gpu_ldpc_submit). A copy
// engine would do the same -- but the runtime's copy queue is ONE in-order queue for all streams: a decode's copy-out waits in it for
// its kernel, and the copy-in of the next handle's decode waits behind that copy-out, so decodes submitted on eight streams ran one
// after the other (measured: 8 x 2.3 ms = 14 ms for what takes 4.6 ms side by side; profiles/HISTORY.md, round 5).
__global__ __launch_bounds__(256) void ldpc_results_to_host_kernel(const uint4 *__restrict__ bits, size_t n16, const uint8_t *__restrict__ bits_tail, int n_tail,
                                                                  const int *__restrict__ trials, int n_trials, const int *__restrict__ error,
                                                                  uint4 *h_bits, uint8_t *h_bits_tail, int *h_trials, int *h_error)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) h_bits[i] = bits[i];
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < n_tail; i += blockDim.x) h_bits_tail[i] = bits_tail[i];
        for (int i = threadIdx.x; i < n_trials; i += blockDim.x) h_trials[i] = trials[i];
        if (threadIdx.x == 0) *h_error = *error;
    }
}

// zeroes the rendezvous words and the error word of a decode in the decode's own stream (two hipMemsetAsync before: the runtime's fill
// kernels -- see t2gpu_ldpc_execute_dev)
__global__ __launch_bounds__(256) void ldpc_clear_kernel(unsigned *a, size_t na, unsigned *b, size_t nb, unsigned *c, size_t nc)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < na; i += stride) a[i] = 0u;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += stride) b[i] = 0u;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nc; i += stride) c[i] = 0u;
}
hipError_t ldpc_clear(unsigned *a, size_t na, unsigned *b, size_t nb, hipStream_t stream, unsigned *c, size_t nc)
{
    size_t n = na > nb ? na : nb;
    n = n > nc ? n : nc;
    int grid = (int)((n + 255) / 256);
    grid = grid < 1 ? 1 : (grid > 64 ? 64 : grid);
    hipLaunchKernelGGL(ldpc_clear_kernel, dim3(grid), dim3(256), 0, stream, a, na, b, nb, c, nc);
    return hipGetLastError();
}

hipError_t ldpc_results_to_host(const uint8_t *d_bits, size_t bytes, const int *d_trials, int n_trials, const int *d_error, uint8_t *h_bits, int *h_trials,
                                int *h_error, hipStream_t stream)
{
    const size_t n16 = bytes / 16;
    int grid = (int)((n16 + 255) / 256);
    grid = grid < 1 ? 1 : (grid > 64 ? 64 : grid);
    hipLaunchKernelGGL(ldpc_results_to_host_kernel, dim3(grid), dim3(256), 0, stream, reinterpret_cast<const uint4 *>(d_bits), n16, d_bits + 16 * n16,
                       (int)(bytes - 16 * n16), d_trials, n_trials, d_error, reinterpret_cast<uint4 *>(h_bits), h_bits + 16 * n16, h_trials, h_error);
    return hipGetLastError();
}

hipError_t ldpc_kernel2_launch(int min_cnt, int max_cnt, const LdpcKernelParams &p, int grid, int lds_bytes, hipStream_t stream, bool allow_cooperative)
{
    ldpc2_kernel_fn fn = pick_kernel2(min_cnt, max_cnt);
    // The workgroups of a SIMD batch meet at every sweep: the grid must be resident as a whole. A cooperative launch makes that the
    // runtime's promise (it refuses a grid that does not fit and does not start it beside work that would keep part of it out)
    // instead of an assumption about what else is on the device. T2GPU_LDPC_COOPERATIVE=0: the plain launch (A/B measurements).
    const char *coop_env = std::getenv("T2GPU_LDPC_COOPERATIVE");
    // allow_cooperative false: the caller keeps several small launches in flight on streams of their own (t2gpu_ldpc_submit) and has
    // bounded their total size by the device's resident capacity itself -- cooperative launches of different streams run one after the
    // other (measured: union of 18 launches = their sum), plain ones side by side
    const bool cooperative = allow_cooperative && !(coop_env && std::atoi(coop_env) == 0);
    if (cooperative) {
        const LdpcLayerDev *layers = p.layers;
        const uint32_t *entries = p.entries, *cninfo = p.cninfo, *entries2 = p.entries2;
        LdpcKernelParams q = p;
        void *args[] = {&layers, &entries, &cninfo, &entries2, &q};
        const hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void *>(fn), dim3(grid), dim3(kThreads2), args, (unsigned)lds_bytes, stream);
        if (e == hipSuccess) return e;
        (void)hipGetLastError();
        // only a device / runtime WITHOUT cooperative launches falls back to the plain launch; "too large" is exactly what the
        // cooperative launch is there to catch (a grid that is not co-resident would hang at the first batch rendezvous), and any
        // other failure is a failure of this call too
        if (e != hipErrorNotSupported && e != hipErrorInvalidDeviceFunction) return e;
    }
    hipLaunchKernelGGL(fn, dim3(grid), dim3(kThreads2), lds_bytes, stream, p.layers, p.entries, p.cninfo, p.entries2, p);
    return hipGetLastError();
}

}  // namespace t2gpu
