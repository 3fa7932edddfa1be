// ldpc_cn.h -- check-node arithmetic of the layered int8 offset-min-sum decoder, written once for the HIP kernel and
// for the host-side schedule emulator in tests/ (it is arithmetic only: no memory model, no parallelism).
//
// Semantics follow the reference bit for bit:
//   LDPCDecoder::update   /root/reference/src/DVB_T2/LDPC/layered_decoder.hh:83-110
//   OffsetMinSumAlgorithm<SIMD<int8_t,W>,NormalUpdate,2>  LDPC/algorithms.hh:221-292 (finalp :250-276, update :288-291)
//   int8 primitives       LDPC/avx2.hh:379-385 (adds), :443-449 (subs), :459-465 (subs_epu8), :491-497 (vqabs),
//                         :535-541 (vsign)
//
// The reference stores one int8 message per link (bnl[], 226 799 bytes for N=64800 r=3/4). Here a check node keeps
// only what is needed to regenerate those messages exactly: the two smallest magnitudes clamped to 32, the slot of
// the smallest, and one sign bit per link:  msg_c = sign_c ? -m : min(m, 31),  m = (c == idx ? min1 : min0),
// which equals clamp(out_c, -32, 31) of algorithms.hh:290 because out_c = +-(mag_c == min0 ? min1 : min0) and equal
// minima make the slot choice immaterial. The a-posteriori update uses the unclamped value, as the reference does.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define T2_HD __host__ __device__ __forceinline__
#else
#define T2_HD inline
#endif

namespace t2gpu {

struct CnState {
    uint32_t w0;   // min0c (bits 0..7) | min1c (8..15) | idx (16..23)
    uint32_t w1;   // sign bit per link slot
};

T2_HD int t2_clamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Address of the information bit that node j reaches through table entry `e` (base | shift<<16).
T2_HD int t2_link_addr(uint32_t e, int j)
{
    int m = j - (int)(e >> 16);
    m += (m < 0) ? 360 : 0;
    return (int)(e & 0xffffu) + m;
}

// One check-node update. L: byte-addressable LLR store with ld(addr)/st(addr, v).
//   CNT     information-bit links (link slots 0..CNT-1 in entry order)
//   slot CNT   = own parity bit      pty[360*i + j]                     (always present)
//   slot CNT+1 = previous parity bit pty[360*(i-1) + j], or pty[360*(q-1) + j - 1] for i == 0, absent for (0,0)
template <int CNT, class LMEM>
T2_HD void t2_cn_update(LMEM &L, const uint32_t *__restrict__ ent, int j, int a_p0, int a_p1, CnState &st)
{
    constexpr int DEG = CNT + 2;
    int addr[DEG], in[DEG], mag[DEG];
    const int min0c = (int)(st.w0 & 0xff), min1c = (int)((st.w0 >> 8) & 0xff), idx = (int)((st.w0 >> 16) & 0xff);
    const uint32_t signs = st.w1;
    int m0 = 255, m1 = 255, mi = 0, sx = 0;
#pragma unroll
    for (int c = 0; c < DEG; ++c) {
        int a = (c < CNT) ? t2_link_addr(ent[c < CNT ? c : 0], j) : (c == CNT ? a_p0 : a_p1);
        addr[c] = a;
        const bool present = (c <= CNT) || (a >= 0);
        int lc = present ? (int)L.ld(a) : 0;
        int mm = (c == idx) ? min1c : min0c;
        int msg = ((signs >> c) & 1u) ? -mm : (mm > 31 ? 31 : mm);
        int v = present ? t2_clamp(lc - msg, -128, 127) : 0;      // alg.sub: saturating
        in[c] = v;
        int av = v < 0 ? -v : v;                                   // vqabs(max(v,-127)) then uint8 sat-sub of beta=1
        av = av > 127 ? 127 : av;
        av = av > 0 ? av - 1 : 0;
        av = present ? av : 255;
        mag[c] = av;
        if (av < m0) { m1 = m0; m0 = av; mi = c; }
        else if (av < m1) { m1 = av; }
        sx ^= v;
    }
    uint32_t nsigns = 0;
#pragma unroll
    for (int c = 0; c < DEG; ++c) {
        const bool present = (c <= CNT) || (addr[c] >= 0);
        int other = (mag[c] == m0) ? m1 : m0;
        bool neg = ((sx ^ in[c]) < 0);
        int out = neg ? -other : other;
        int ln = t2_clamp(in[c] + out, -128, 127);                 // alg.add: saturating
        if (present) L.st(addr[c], (int8_t)ln);
        nsigns |= ((neg && other != 0 && present) ? 1u : 0u) << c;
    }
    st.w0 = (uint32_t)(m0 > 32 ? 32 : m0) | ((uint32_t)(m1 > 32 ? 32 : m1) << 8) | ((uint32_t)mi << 16);
    st.w1 = nsigns;
}

// Parity check of node j on the current LLRs (LDPCDecoder::bad, layered_decoder.hh:65-82): the node is bad when
// the sign product of its neighbours is not strictly positive -- any zero LLR counts as bad (vsign zeroes).
template <int CNT, class LMEM>
T2_HD bool t2_cn_bad(const LMEM &L, const uint32_t *__restrict__ ent, int j, int a_p0, int a_p1)
{
    int sx = 0;
    bool zero = false;
#pragma unroll
    for (int c = 0; c < CNT; ++c) {
        int v = (int)L.ld(t2_link_addr(ent[c], j));
        sx ^= v;
        zero |= (v == 0);
    }
    int v = (int)L.ld(a_p0);
    sx ^= v; zero |= (v == 0);
    if (a_p1 >= 0) { v = (int)L.ld(a_p1); sx ^= v; zero |= (v == 0); }
    return zero || (sx < 0);
}

// Dispatch a runtime per-layer link count to the unrolled instance. Counts present in the twelve T2 codes: 2..20.
#define T2_LDPC_DISPATCH_CNT(cnt, CALL)                                                                         \
    switch (cnt) {                                                                                              \
    case 1: { constexpr int CNT = 1; CALL; } break;   case 2: { constexpr int CNT = 2; CALL; } break;           \
    case 3: { constexpr int CNT = 3; CALL; } break;   case 4: { constexpr int CNT = 4; CALL; } break;           \
    case 5: { constexpr int CNT = 5; CALL; } break;   case 6: { constexpr int CNT = 6; CALL; } break;           \
    case 7: { constexpr int CNT = 7; CALL; } break;   case 8: { constexpr int CNT = 8; CALL; } break;           \
    case 9: { constexpr int CNT = 9; CALL; } break;   case 10: { constexpr int CNT = 10; CALL; } break;         \
    case 11: { constexpr int CNT = 11; CALL; } break; case 12: { constexpr int CNT = 12; CALL; } break;         \
    case 13: { constexpr int CNT = 13; CALL; } break; case 14: { constexpr int CNT = 14; CALL; } break;         \
    case 15: { constexpr int CNT = 15; CALL; } break; case 16: { constexpr int CNT = 16; CALL; } break;         \
    case 17: { constexpr int CNT = 17; CALL; } break; case 18: { constexpr int CNT = 18; CALL; } break;         \
    case 19: { constexpr int CNT = 19; CALL; } break; case 20: { constexpr int CNT = 20; CALL; } break;         \
    default: break;                                                                                             \
    }

}  // namespace t2gpu
