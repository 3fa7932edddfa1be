// ldpc_cn.h -- check-node arithmetic and per-layer schedule of the layered int8 offset-min-sum decoder, written once
// for the HIP kernel (ldpc_kernel.hip) and for the host-side schedule emulator in tests/emu (which replays the same
// phases thread by thread, in adversarial thread orders, to prove the schedule is race-free and LLR-exact).
//
// Semantics follow the reference bit for bit:
//   LDPCDecoder::update   /root/reference/src/DVB_T2/LDPC/layered_decoder.hh:83-110
//   OffsetMinSumAlgorithm<SIMD<int8_t,W>,NormalUpdate,2>  LDPC/algorithms.hh:221-292 (finalp :250-276, update :288-291)
//   int8 primitives       LDPC/avx2.hh:379-385 (adds), :443-449 (subs), :459-465 (subs_epu8), :491-497 (vqabs),
//                         :535-541 (vsign)
//
// Message storage. The reference keeps one int8 message per link (bnl[], 226 799 bytes for N=64800 r=3/4). Here a
// check node keeps only what regenerates those messages exactly: the two smallest magnitudes clamped to 32 (A, B) and a
// 2-bit code per link ("held the minimum", sign):   msg_c = min(31, sign_c ? -m : m),  m = ismin_c ? B : A,
// which equals clamp(out_c, -32, 31) of algorithms.hh:290 because out_c = +-(mag_c == min0 ? min1 : min0). (A sign bit
// set on a zero magnitude is harmless: -0 == 0.) The a-posteriori update uses the unclamped value, as the reference does.
//
// Order. The reference visits the 360 nodes of a layer in ascending j. Nodes sharing a bit inside a layer (ldpc_graph.h)
// must observe that order; everything else is free. Three layer kinds:
//   PLAIN   no shared bits: every node reads, computes, writes; one barrier.
//   PAIR    one 360-bit group entered through two shifts s0, s1 with (s1 - s0) mod 360 = step <= 180. Node j's slot 1
//           bit is node (j - step)'s slot 0 bit: chains j, j+step, j+2*step, ... The value handed down a chain is a
//           scalar recurrence  X' = sat(in0 + sgn * min(E, mag(sat(X - msg1))))  -- one lane walks one chain with X in a
//           register (no LDS round trip, no barrier per step), then all nodes finish in parallel.
//   GENERIC anything else: dependency levels (ldpc_graph.h); per level only the conflict slots are recomputed and
//           written, the remaining slots of every node are finished in parallel afterwards.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define T2_HD __host__ __device__ __forceinline__
#else
#define T2_HD inline
#endif

#ifndef T2_CN_HOOK_AFTER_LOAD
#define T2_CN_HOOK_AFTER_LOAD ((void)0)      // diagnostics builds of the kernel time-stamp this point (ldpc_kernel.hip, T2_PROF_DETAIL 3)
#endif

namespace t2gpu {

enum { T2_LAYER_PLAIN = 0, T2_LAYER_PAIR = 1, T2_LAYER_GENERIC = 2 };
enum { T2_LDPC_NC_MAX = 10 };   // most conflict slots of any layer of the twelve T2 codes: 9 (short 5/6, layer 4)

struct CnState {
    uint32_t w0;   // 2-bit code per link slot 0..15: bit 2c = sign of the stored message, bit 2c+1 = "slot held the minimum"
    uint32_t w1;   // codes of slots 16..21 (bits 0..11) | A = min(f(min0), 32) << 16 | B = min(f(min1), 32) << 24
};

// median of three. On the GPU this is pinned to one v_med3_i32: left to itself the compiler turns the int8 saturation
// idiom into shift / 16-bit saturating op / shift (4 instructions instead of 2 per saturating add).
T2_HD int t2_med3(int a, int b, int c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
#else
    int lo = a < b ? a : b, hi = a < b ? b : a;
    return c < lo ? lo : (c > hi ? hi : c);
#endif
}
T2_HD int t2_clamp(int v, int lo, int hi) { return t2_med3(v, lo, hi); }

// Address of the information bit that node j reaches through table entry `e` (base | shift<<16).
T2_HD int t2_link_addr(uint32_t e, int j)
{
    int m = j - (int)(e >> 16);
    m += (m < 0) ? 360 : 0;
    return (int)(e & 0xffffu) + m;
}

// f(|v|): vqabs (cap at 127) then uint8 saturating subtraction of beta = 1. Monotone, so the two smallest of f(|in_c|)
// are f of the two smallest |in_c|: the decoder tracks raw magnitudes per link and applies f twice per node.
T2_HD int t2_f(int a)
{
    a = a > 127 ? 127 : a;
    return a > 0 ? a - 1 : 0;
}

// Per-thread registers of one check node (link slots 0..CNT-1 information bits in entry order, slot CNT own parity
// bit pty[360*i+j], slot CNT+1 previous parity bit -- absent for node (0,0), address < 0). Addresses are what the memory
// object L takes in ld()/st(): LLR index + L.off(); callers pass a_p0 / a_p1 (and LayerDesc::dummy) in that form.
template <int CNT>
struct CnRegs {
    static constexpr int DEG = CNT + 2;
    int addr[DEG], in[DEG];     // in = sat(L - old message); the absent slot holds 0
    uint32_t lut;               // old messages by code: byte0 = min(A,31), byte1 = -A, byte2 = min(B,31), byte3 = -B
    uint32_t c0, c1;            // old codes
    int p0, p1, psx;            // two smallest raw magnitudes / sign xor over the non-conflict slots
    int m0, m0f, m1f, sx;       // over all slots: raw minimum, f(min0), f(min1), sign xor
    uint32_t n0, n1;            // new codes
};

template <int CNT>
T2_HD bool t2_present(const CnRegs<CNT> &r, int c) { return c <= CNT || r.addr[c] >= 0; }

// byte `code` (0..3) of lut, sign-extended. One v_perm_b32 on the GPU.
T2_HD int t2_lut_byte(uint32_t lut, uint32_t code)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (int)(int8_t)__builtin_amdgcn_perm(0u, lut, code | 0x0c0c0c00u);
#else
    return (int)(int8_t)((lut >> (8 * code)) & 0xffu);
#endif
}

// message stored for slot c by the previous sweep: clamp(out_c, -32, 31) (algorithms.hh:288-291)
template <int CNT>
T2_HD int t2_old_msg(const CnRegs<CNT> &r, int c)
{
    uint32_t code = (c < 16) ? ((r.c0 >> (2 * (c & 15))) & 3u) : ((r.c1 >> (2 * (c & 15))) & 3u);
    return t2_lut_byte(r.lut, code);
}

// (re)read slot c: in = sat(L - old message) (alg.sub), raw magnitude
template <int CNT, class LMEM>
T2_HD void t2_read_slot(const LMEM &L, CnRegs<CNT> &r, int c)
{
    const bool present = t2_present(r, c);
    int lc = present ? (int)L.ld(r.addr[c]) : 0;
    int v = present ? t2_clamp(lc - t2_old_msg(r, c), -128, 127) : 0;
    r.in[c] = v;
}

// raw magnitude |in| (0..128) of slot c; 255 for the absent slot so that it never wins a minimum
template <int CNT>
T2_HD int t2_rawmag(const CnRegs<CNT> &r, int c)
{
    int v = r.in[c], nv = -v;
    int a = v > nv ? v : nv;
    return t2_present(r, c) ? a : 255;
}

template <int CNT, class LMEM>
T2_HD void t2_cn_load(const LMEM &L, const uint32_t *__restrict__ ent, int j, int a_p0, int a_p1, const CnState &st,
                      CnRegs<CNT> &r, const uint32_t *__restrict__ ent2 = nullptr)
{
    r.c0 = st.w0; r.c1 = st.w1;
    {
        const int A = (int)((st.w1 >> 16) & 0xffu), B = (int)(st.w1 >> 24);
        r.lut = (uint32_t)(A > 31 ? 31 : A) | ((uint32_t)((-A) & 0xff) << 8) | ((uint32_t)(B > 31 ? 31 : B) << 16) |
                ((uint32_t)((-B) & 0xff) << 24);
    }
    r.n0 = 0; r.n1 = 0;
    // three passes -- addresses, loads, arithmetic -- so that the CNT + 2 LDS reads are in flight together
#pragma unroll
    for (int c = 0; c < CNT + 2; ++c) {
        // L.off(): where the LLR array starts in the memory L addresses (the LDS offset on the GPU; folded into the table entry's
        // base by scalar arithmetic, so the per-lane address needs no further add at the load or the store)
#if defined(__HIP_DEVICE_COMPILE__)
        if (c < CNT) {
            r.addr[c] = (int)ent2[2 * c] + (j >= (int)ent2[2 * c + 1] ? j : j + 360);
        } else {
            r.addr[c] = (c == CNT ? a_p0 : a_p1);
        }
#else
        (void)ent2;
        r.addr[c] = (c < CNT) ? t2_link_addr(ent[c < CNT ? c : 0] + (uint32_t)L.off(), j) : (c == CNT ? a_p0 : a_p1);
#endif
    }
#pragma unroll
    for (int c = 0; c < CNT + 2; ++c) r.in[c] = t2_present(r, c) ? (int)L.ld(r.addr[c]) : 0;
#pragma unroll
    for (int c = 0; c < CNT + 2; ++c) r.in[c] = t2_present(r, c) ? t2_clamp(r.in[c] - t2_old_msg(r, c), -128, 127) : 0;
}

// second smallest of {m0 <= m1, a} is the median; smallest is the min
T2_HD void t2_min2(int a, int &m0, int &m1)
{
    m1 = t2_med3(m0, m1, a);       // second smallest of (m0 <= m1, a) is their median
    m0 = m0 < a ? m0 : a;
}

// two smallest raw magnitudes / sign xor over slots >= nc
template <int CNT>
T2_HD void t2_cn_partial(CnRegs<CNT> &r, int nc)
{
    int m0 = 255, m1 = 255, sx = 0;
#pragma unroll
    for (int c = 0; c < CNT + 2; ++c)
        if (c >= nc) { t2_min2(t2_rawmag<CNT>(r, c), m0, m1); sx ^= r.in[c]; }
    r.p0 = m0; r.p1 = m1; r.psx = sx;
}

// fold the conflict slots (< nc) into the partial result and apply f
template <int CNT>
T2_HD void t2_cn_merge(CnRegs<CNT> &r, int nc)
{
    int m0 = r.p0, m1 = r.p1, sx = r.psx;
#pragma unroll
    for (int c = 0; c < T2_LDPC_NC_MAX; ++c)
        if (c < CNT && c < nc) { t2_min2(t2_rawmag<CNT>(r, c), m0, m1); sx ^= r.in[c]; }
    r.m0 = m0; r.m0f = t2_f(m0); r.m1f = t2_f(m1); r.sx = sx;
}

// output of slot c, a-posteriori update L = sat(in + out) (alg.add), sign / is-min bits of the new message.
// `other` = (mag_c == min0 ? min1 : min0) on f-values equals the same selection on raw magnitudes (see t2_f).
template <int CNT, class LMEM>
T2_HD void t2_write_slot(LMEM &L, CnRegs<CNT> &r, int c, bool store)
{
    const bool present = t2_present(r, c);
    const bool eq = present && (r.in[c] == r.m0 || r.in[c] == -r.m0);    // |in_c| == raw minimum
    int other = eq ? r.m1f : r.m0f;
    int sm = (r.sx ^ r.in[c]) >> 31;                 // 0 / -1: sign of the product of the other inputs
    int out = (other ^ sm) - sm;
    int ln = t2_clamp(r.in[c] + out, -128, 127);
    if (present && store) L.st(r.addr[c], (int8_t)ln);
    if (c < 16) r.n0 |= ((uint32_t)sm & (1u << (2 * (c & 15)))) | (eq ? (2u << (2 * (c & 15))) : 0u);
    else r.n1 |= ((uint32_t)sm & (1u << (2 * (c & 15)))) | (eq ? (2u << (2 * (c & 15))) : 0u);
}

template <int CNT>
T2_HD void t2_cn_pack(const CnRegs<CNT> &r, CnState &st)
{
    st.w0 = r.n0;
    st.w1 = (r.n1 & 0xfffu) | ((uint32_t)(r.m0f > 32 ? 32 : r.m0f) << 16) | ((uint32_t)(r.m1f > 32 ? 32 : r.m1f) << 24);
}

// ---- PAIR layers -------------------------------------------------------------------------------------------------
// Record a chain walker needs from a node that has both a predecessor and a successor, in the form the walk consumes:
//   bits 0-7   k1 = -s * msg1 - 1 (signed), msg1 = old message of slot 1, s = +1 / -1 the sign parity of the slots other than 0, 1
//   bits 8-14  cap = min(126, f(E)), E the smallest raw magnitude of those other slots
//   bits 16-23 in0 (input of slot 0, signed)          bits 24-25 s as a 2-bit signed field
T2_HD uint32_t t2_pair_pack(int msg1, int cap, int in0, bool neg)
{
    const int k1 = (neg ? msg1 : -msg1) - 1;
    return (uint32_t)(k1 & 0xff) | ((uint32_t)cap << 8) | ((uint32_t)(in0 & 0xff) << 16) | (neg ? (3u << 24) : (1u << 24));
}
template <int CNT>
T2_HD uint32_t t2_pair_record(const CnRegs<CNT> &r)
{
    return t2_pair_pack(t2_old_msg(r, 1), t2_f(r.p0), r.in[0], r.psx < 0);
}

// One step down a chain. X = LLR of the shared bit after the predecessor; returns it after this node:
//   X' = sat(in0 + s * sign(u) * min(f(E), f(|sat(u)|))),  u = X - msg1.
// f(|sat(u)|) = med3(|u| - 1, 0, 126) also without the saturation (|u| <= 160), so the middle term is the odd function
//   g(w) = sign(w) * med3(|w| - 1, 0, cap) = med3(w - 1, 0, cap) + med3(w + 1, -cap, 0)   of   w = s * u = s * X + (k1 + 1),
// which needs no absolute value and no sign bookkeeping: one multiply-add, one add, two med3, one three-operand add, the final
// saturation -- four operations deep, because this recurrence IS the critical path of a PAIR layer.
struct PairRec { int k1, cap, ncap, in0, sg; };
T2_HD PairRec t2_pair_unpack(uint32_t rec)
{
    PairRec r;
    r.k1 = (int)(int8_t)(rec & 0xff);
    r.cap = (int)((rec >> 8) & 0x7f);
    r.ncap = -r.cap;
    r.in0 = (int)(int8_t)((rec >> 16) & 0xff);
    r.sg = ((int)(rec << 6)) >> 30;                      // bits 24-25, sign-extended: +1 / -1
    return r;
}
T2_HD int t2_pair_step(const PairRec &r, int X)
{
#if defined(__HIP_DEVICE_COMPILE__)
    int t1;                                              // w - 1
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(t1) : "v"(X), "v"(r.sg), "v"(r.k1));
#else
    const int t1 = X * r.sg + r.k1;
#endif
    const int t2 = t1 + 2;                               // w + 1
    return t2_clamp(t2_clamp(t1, 0, r.cap) + t2_clamp(t2, r.ncap, 0) + r.in0, -128, 127);
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
        int v = (int)L.ld(t2_link_addr(ent[c] + (uint32_t)L.off(), j));
        sx ^= v;
        zero |= (v == 0);
    }
    int v = (int)L.ld(a_p0);
    sx ^= v; zero |= (v == 0);
    if (a_p1 >= 0) { v = (int)L.ld(a_p1); sx ^= v; zero |= (v == 0); }
    return zero || (sx < 0);
}

// ---- one layer, phase by phase ------------------------------------------------------------------------------------
// `SYNC` provides barrier(): the HIP kernel passes the workgroup barrier, the emulator records an epoch boundary and is
// driven phase by phase instead (see tests/emu). Every thread of the workgroup calls these in the same order.
struct LayerDesc {
    const uint32_t *ent;
    int cnt, lmax, nc, kind, step;
    int dummy;      // address of a scratch byte in the LLR memory: target of the stores a chain walker predicates away
    uint32_t e0;    // ent[0], fetched when the descriptor is made (the chain walker needs it right behind a barrier)
    const uint32_t *ent2 = nullptr;   // GPU only: the same entries as (base + L.off() - shift, shift) dword pairs -- saves the scalar
                                      // add / shift / mask per link that unpacking costs every wavefront in every layer
    int ent_lds = 0;                  // pair-lane kernel: LDS address of a copy of those pairs (ldpc_cn2.h)
    int pair_flag_lds = 0;            // two-frame kernel, PAIR layers: LDS address of the per-node ready flags [360][2 frames] (ldpc_cn3.h)
    int no_close = 0;                 // two-frame kernel: the layer ends without the workgroup barrier (ldpc_graph.h)
    int band = 0, band_prefetch = 0;  // two-frame kernel: band walk of a GENERIC layer (ldpc_graph.h), with the LDS addresses of its
    int band_rec_lds = 0, band_in_lds = 0;   // per-node records [2][360] x 8 B and of the inputs it leaves behind [2][360] x 4 B
};

// phase A: every node loads; PLAIN nodes and chain-start / level-free nodes finish at once
template <int CNT, class LMEM>
T2_HD void t2_layer_phase_a(LMEM &L, const LayerDesc &d, int j, int a_p0, int a_p1, CnState &st, CnRegs<CNT> &r,
                            uint32_t *pair_rec)
{
    t2_cn_load<CNT>(L, d.ent, j, a_p0, a_p1, st, r, d.ent2);
    T2_CN_HOOK_AFTER_LOAD;
    if (d.kind == T2_LAYER_PLAIN) {
        t2_cn_partial<CNT>(r, 0);
        r.m0 = r.p0; r.m0f = t2_f(r.p0); r.m1f = t2_f(r.p1); r.sx = r.psx;
#pragma unroll
        for (int c = 0; c < CNT + 2; ++c) t2_write_slot<CNT>(L, r, c, true);
        t2_cn_pack<CNT>(r, st);
    } else if (d.kind == T2_LAYER_PAIR) {
        t2_cn_partial<CNT>(r, 2);
        if (j < d.step) {                       // chain start: nothing earlier touches its bits. Its two group bits must be final
            t2_cn_merge<CNT>(r, 2);             // early -- slot 0 is handed down the chain, slot 1 is the slot-0 bit of the chain END
            t2_write_slot<CNT>(L, r, 0, true);  // j + 360 - step, which re-reads it in phase C; the private slots wait for phase C
            t2_write_slot<CNT>(L, r, 1, true);  // so that every wavefront has the same amount of work in both phases
        } else {
            pair_rec[j] = t2_pair_record<CNT>(r);
        }
    } else {
        t2_cn_partial<CNT>(r, d.nc);
    }
}

// PAIR phase B: lane `lane` < step walks chain lane, lane+step, ... (all nodes that have a successor). Records are fetched
// and unpacked a batch ahead so that only the scalar recurrence (X in a register) is on the critical path. A walking
// wavefront is issue-bound (one lane's worth of work per instruction), so the body is kept short and branch-free: all chains
// of a layer have n or n + 1 such nodes (n = (360 - step) / step - 1, the same for every lane), the first n steps run
// unconditionally in batches of four, at most four predicated steps finish (record index clamped, X kept, store redirected
// to the scratch byte d.dummy for a lane that is already done).
template <class LMEM>
T2_HD void t2_pair_walk(LMEM &L, const LayerDesc &d, int lane, const uint32_t *pair_rec)
{
#ifdef T2_PAIR_WALK_BATCH
    constexpr int B = T2_PAIR_WALK_BATCH;
#else
    constexpr int B = 4;
#endif
    const uint32_t e0 = d.e0;
    const int base = (int)(e0 & 0xffffu) + L.off(), s0 = (int)(e0 >> 16), step = d.step;
    int m = lane - s0;                                   // position of the shared bit inside its 360-bit group
    m += (m < 0) ? 360 : 0;
    int X = (int)L.ld(base + m);
    const int n_common = (360 - step) / step - 1;        // nodes with a successor that every chain of this layer has
    int jj = lane + step;
    int k = 0;
    if (n_common >= B) {
        uint32_t nxt[B];
#pragma unroll
        for (int u = 0; u < B; ++u) nxt[u] = pair_rec[jj + u * step];
        for (; k + B <= n_common; k += B) {
            PairRec cur[B];
#pragma unroll
            for (int u = 0; u < B; ++u) cur[u] = t2_pair_unpack(nxt[u]);
            jj += B * step;
#pragma unroll
            for (int u = 0; u < B; ++u) { const int q = jj + u * step; nxt[u] = pair_rec[q < 359 ? q : 359]; }
#pragma unroll
            for (int u = 0; u < B; ++u) {
                const unsigned t = (unsigned)(m + step);
                m = (int)(t < t - 360u ? t : t - 360u);  // (m + step) mod 360
                X = t2_pair_step(cur[u], X);
                L.st(base + m, (int8_t)X);
            }
        }
    }
    // the rest of the common part (fewer than B nodes, same count for every lane) ...
    for (; k < n_common; ++k) {
        const PairRec r = t2_pair_unpack(pair_rec[jj]);
        const unsigned t = (unsigned)(m + step);
        m = (int)(t < t - 360u ? t : t - 360u);
        X = t2_pair_step(r, X);
        L.st(base + m, (int8_t)X);
        jj += step;
    }
    // ... and the one node more that the chains of the low lanes have
    {
        const bool live = jj + step < 360;
        const PairRec r = t2_pair_unpack(pair_rec[jj < 359 ? jj : 359]);
        const unsigned t = (unsigned)(m + step);
        m = (int)(t < t - 360u ? t : t - 360u);
        const int Xn = t2_pair_step(r, X);
        X = live ? Xn : X;
        L.st(live ? base + m : d.dummy, (int8_t)X);
    }
}

// PAIR phase C: every node finishes. Chain starts wrote slots 0 and 1 in phase A; the others re-read slot 1 (always) and slot 0
// (when they are a chain end) first.
template <int CNT, class LMEM>
T2_HD void t2_pair_finish(LMEM &L, const LayerDesc &d, int j, CnState &st, CnRegs<CNT> &r)
{
    if (j < d.step) {
#pragma unroll
        for (int c = 2; c < CNT + 2; ++c) t2_write_slot<CNT>(L, r, c, true);
        t2_cn_pack<CNT>(r, st);
        return;
    }
    const bool has_succ = j + d.step < 360;
    t2_read_slot<CNT>(L, r, 1);
    if (!has_succ) t2_read_slot<CNT>(L, r, 0);
    t2_cn_merge<CNT>(r, 2);
#pragma unroll
    for (int c = 0; c < CNT + 2; ++c) t2_write_slot<CNT>(L, r, c, !(c == 0 && has_succ));
    t2_cn_pack<CNT>(r, st);
}

// GENERIC level step lv (1-based): nodes of that level settle their conflict slots. All conflict slots are re-read
// (batched, branch-free): a slot nobody touched earlier still holds the value read in phase A. NC (the layer's number of
// conflict slots) is a template argument so that the loads of a step are issued back to back and waited for once -- with a
// run-time bound every slot became its own load / wait / branch and a level step paid one LDS round trip per slot.
template <int CNT, int NC, class LMEM>
T2_HD void t2_generic_level_nc(LMEM &L, int lv, uint32_t info, CnRegs<CNT> &r)
{
    if ((int)(info & 0xff) != lv) return;
    if (lv > 1) {
#pragma unroll
        for (int c = 0; c < NC; ++c) t2_read_slot<CNT>(L, r, c);
    }
    int m0 = r.p0, m1 = r.p1, sx = r.psx;
#pragma unroll
    for (int c = 0; c < NC; ++c) { t2_min2(t2_rawmag<CNT>(r, c), m0, m1); sx ^= r.in[c]; }
    r.m0 = m0; r.m0f = t2_f(m0); r.m1f = t2_f(m1); r.sx = sx;
#pragma unroll
    for (int c = 0; c < NC; ++c) t2_write_slot<CNT>(L, r, c, true);
}

// NCMAX: the largest conflict-slot count the code family of the calling kernel has (bounds the instances carried)
template <int CNT, int NCMAX = T2_LDPC_NC_MAX, class LMEM>
T2_HD void t2_generic_level(LMEM &L, const LayerDesc &d, int lv, uint32_t info, CnRegs<CNT> &r)
{
    switch (d.nc) {
#define T2_NC_(n) case n: if constexpr (n <= CNT && n <= NCMAX) t2_generic_level_nc<CNT, n>(L, lv, info, r); break;
        T2_NC_(2) T2_NC_(3) T2_NC_(4) T2_NC_(5) T2_NC_(6) T2_NC_(7) T2_NC_(8) T2_NC_(9) T2_NC_(10)
#undef T2_NC_
    default: break;
    }
}

// GENERIC last phase: the private slots
template <int CNT, class LMEM>
T2_HD void t2_generic_finish(LMEM &L, const LayerDesc &d, CnState &st, CnRegs<CNT> &r)
{
#pragma unroll
    for (int c = 0; c < CNT + 2; ++c)
        if (c >= d.nc) t2_write_slot<CNT>(L, r, c, true);
    t2_cn_pack<CNT>(r, st);
}

// Dispatch a runtime per-layer link count to the unrolled instance. Counts present in the twelve T2 codes: 2..20.
// LO/HI (compile-time) restrict the instances a kernel variant carries, so register allocation and code size follow
// the code family actually being decoded.
#define T2_CASE_(n, LO, HI, CALL) case n: if constexpr ((LO) <= n && n <= (HI)) { constexpr int CNT = n; CALL; } break;
#define T2_LDPC_DISPATCH_RANGE(cnt, LO, HI, CALL)                                                               \
    switch (cnt) {                                                                                              \
        T2_CASE_(1, LO, HI, CALL) T2_CASE_(2, LO, HI, CALL) T2_CASE_(3, LO, HI, CALL) T2_CASE_(4, LO, HI, CALL)     \
        T2_CASE_(5, LO, HI, CALL) T2_CASE_(6, LO, HI, CALL) T2_CASE_(7, LO, HI, CALL) T2_CASE_(8, LO, HI, CALL)     \
        T2_CASE_(9, LO, HI, CALL) T2_CASE_(10, LO, HI, CALL) T2_CASE_(11, LO, HI, CALL) T2_CASE_(12, LO, HI, CALL)  \
        T2_CASE_(13, LO, HI, CALL) T2_CASE_(14, LO, HI, CALL) T2_CASE_(15, LO, HI, CALL) T2_CASE_(16, LO, HI, CALL) \
        T2_CASE_(17, LO, HI, CALL) T2_CASE_(18, LO, HI, CALL) T2_CASE_(19, LO, HI, CALL) T2_CASE_(20, LO, HI, CALL) \
    default: break;                                                                                             \
    }
#define T2_LDPC_DISPATCH_CNT(cnt, CALL) T2_LDPC_DISPATCH_RANGE(cnt, 1, 20, CALL)

}  // namespace t2gpu
