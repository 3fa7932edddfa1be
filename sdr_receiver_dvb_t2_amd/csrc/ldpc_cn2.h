// ldpc_cn2.h -- the check-node update of ldpc_cn.h with one check node spread over TWO adjacent lanes (GPU only).
//
// Lane pair (2j, 2j+1) owns node j: lane half h = 0 takes the even link slots c = 0, 2, 4, ..., h = 1 the odd ones, slot numbering,
// record format, layer kinds and phase structure exactly as in ldpc_cn.h (which stays the reference form: the schedule emulator in
// tests/emu runs it, and every function below names the one it splits). What a node needs as a whole -- the two smallest
// magnitudes, the sign parity, the code bits of its record -- meets through one DPP move between the two lanes
// (quad_perm [1,0,3,2]); nothing else changes hands, and the two lanes never touch the same LLR.
//
// Why: 360 nodes are 6 wavefronts on 4 SIMDs, two SIMDs carry two of them and the layer waits for those (a third of a PLAIN
// layer was barrier wait). 720 half-nodes are 12 wavefronts, three per SIMD, each with half the
// instruction stream.
#pragma once
#include "ldpc_cn.h"

namespace t2gpu {

// f of ldpc_cn.h for a raw magnitude 0..255 in two instructions
__device__ __forceinline__ int pl_f(int a) { return t2_clamp(a - 1, 0, 126); }

// The record of a node with at most 16 link slots keeps all its codes in w0, so w1 can carry the message table of the NEXT sweep
// ready-made (bytes min(A,31), -A, min(B,31), -B with A = min(f(min0), 32), B = min(f(min1), 32)) instead of A and B:
// six instructions at the pack (two packed 16-bit ones) replace five there plus eight at every load.
typedef short pl_short2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pl_lut_word(int m0f, int m1f)
{
    const int A = m0f > 32 ? 32 : m0f, B = m1f > 32 ? 32 : m1f;
    const uint32_t ab = (uint32_t)A | ((uint32_t)B << 16);
    pl_short2 x = __builtin_bit_cast(pl_short2, ab);
    const pl_short2 neg = -x;
    const pl_short2 lim = {31, 31};
    const pl_short2 cap = __builtin_elementwise_min(x, lim);
    return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, neg), __builtin_bit_cast(uint32_t, cap), 0x06020400u);
}

// value of v in the partner lane (lane ^ 1)
__device__ __forceinline__ int pl_x(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t pl_xu(uint32_t v) { return (uint32_t)pl_x((int)v); }

// the two smallest of the union of two sorted pairs
__device__ __forceinline__ void pl_merge2(int a0, int a1, int b0, int b1, int &m0, int &m1)
{
    const int lo = a0 < b0 ? a0 : b0, hi = a0 < b0 ? b0 : a0;
    const int s = a1 < b1 ? a1 : b1;
    m0 = lo;
    m1 = hi < s ? hi : s;
}

template <int CNT>
struct PlRegs {
    static constexpr int DEG = CNT + 2, H = (DEG + 1) / 2;     // link slots of the node / of one lane (slot c = 2v + h)
    int addr[H], in[H];         // absent slots: addr < 0, in = 0
    int mag[H];                 // raw magnitude |in| (255 for an absent slot), kept for the "held the minimum" test of the write phase
    uint32_t lut;               // as CnRegs::lut
    uint32_t c0s, c1s;          // old code words shifted right by 2h: slot v's code sits at bits 4v (word 0: c < 16, word 1: the rest)
    int p0, p1, psx;            // node-wide: two smallest raw magnitudes / sign xor over the non-conflict slots
    int m0, m0f, m1f, sx;       // node-wide over all slots
    uint32_t n0, n1;            // new codes of this lane's slots at bits 4v (as if h were 0); shifted and joined in pl_pack
    int h;
};

// compile-time classification of slot c of a node with CNT information links
template <int CNT> __device__ __forceinline__ constexpr bool pl_is_info(int c) { return c < CNT; }

template <int CNT>
__device__ __forceinline__ bool pl_present(const PlRegs<CNT> &r, int v)
{
    return (2 * v + 1 <= CNT) || r.addr[v] >= 0;      // both candidates are information or own-parity slots: always there
}

template <int CNT>
__device__ __forceinline__ int pl_old_msg(const PlRegs<CNT> &r, int v)
{
    const uint32_t w = v < 8 ? r.c0s : r.c1s;
    return t2_lut_byte(r.lut, (w >> (4 * (v & 7))) & 3u);
}

// sat-free part of alg.sub for slot v: (LLR byte, sign-extended) - (old message). The message is byte `code` of lut; a selector
// with zero upper bytes leaves lut's byte 0 in the upper bytes of the permute result, which the byte-selecting subtract never
// reads -- so the code needs no "| 0x0c0c0c00": bit-field extract, permute, byte-selecting subtract (three instructions, four before).
template <int CNT>
__device__ __forceinline__ int pl_llr_minus_msg(const PlRegs<CNT> &r, int v, uint8_t raw)
{
    const uint32_t w = v < 8 ? r.c0s : r.c1s;
    const uint32_t code = __builtin_amdgcn_ubfe(w, 4u * (uint32_t)(v & 7), 2u);
    const uint32_t m = __builtin_amdgcn_perm(0u, r.lut, code);
    return (int)(int8_t)raw - (int)(int8_t)(m & 0xffu);   // both sign extensions ride on the subtract (SDWA byte selects)
}

template <int CNT> __device__ __forceinline__ int pl_rawmag_of(const PlRegs<CNT> &r, int v);
template <int CNT, class LMEM>
__device__ __forceinline__ void pl_read_slot(const LMEM &L, PlRegs<CNT> &r, int v)
{
    const bool present = pl_present(r, v);
    const uint8_t raw = present ? L.ld_raw(r.addr[v]) : (uint8_t)0;
    r.in[v] = present ? t2_clamp(pl_llr_minus_msg(r, v, raw), -128, 127) : 0;
    r.mag[v] = pl_rawmag_of(r, v);
}

template <int CNT>
__device__ __forceinline__ int pl_rawmag_of(const PlRegs<CNT> &r, int v)
{
    const int x = r.in[v], nx = -x;
    const int a = x > nx ? x : nx;
    return pl_present(r, v) ? a : 255;
}
template <int CNT>
__device__ __forceinline__ int pl_rawmag(const PlRegs<CNT> &r, int v) { return r.mag[v]; }

// t2_cn_load for one lane. ent_lds: LDS address of the layer's entries as (base + L.off() - shift, shift) pairs.
template <int CNT, class LMEM>
__device__ __forceinline__ void pl_load(const LMEM &L, int ent_lds, int j, int h, int a_p0, int a_p1, const CnState &st, PlRegs<CNT> &r)
{
    constexpr int H = PlRegs<CNT>::H;
    r.h = h;
    if constexpr (PlRegs<CNT>::DEG <= 16) {
        r.lut = st.w1;                                             // see pl_lut_word
        r.c1s = 0;
    } else {
        const int A = (int)((st.w1 >> 16) & 0xffu), B = (int)(st.w1 >> 24);
        r.lut = (uint32_t)(A > 31 ? 31 : A) | ((uint32_t)((-A) & 0xff) << 8) | ((uint32_t)(B > 31 ? 31 : B) << 16) |
                ((uint32_t)((-B) & 0xff) << 24);
        r.c1s = (st.w1 & 0xfffu) >> (2 * h);
    }
    r.c0s = st.w0 >> (2 * h);
    r.n0 = 0; r.n1 = 0;
    uint2 e[H];
    const int jw = j + 360;
    const int mine = ent_lds + 8 * h;                              // the odd lane reads the odd entries
#pragma unroll
    for (int v = 0; v < H; ++v) {                                  // the lane's table entries from the LDS copy, issued together
        const int c0 = 2 * v, c1 = 2 * v + 1;
        if (c1 < CNT) e[v] = L.ld_pair(mine + 8 * c0);
        else if (c0 < CNT) e[v] = L.ld_pair(ent_lds + 8 * c0);
        else e[v] = make_uint2(0u, 0u);
    }
#pragma unroll
    for (int v = 0; v < H; ++v) {
        const int c0 = 2 * v, c1 = 2 * v + 1;
        const int link = (int)e[v].x + (j >= (int)e[v].y ? j : jw);  // (base - shift) + j, + 360 where the shift wraps
        if (c1 < CNT) r.addr[v] = link;                             // information slots on both lanes
        else if (c0 < CNT) r.addr[v] = h ? a_p0 : link;             // c1 == CNT: own parity bit on the odd lane
        else if (c0 == CNT) r.addr[v] = h ? a_p1 : a_p0;            // own parity / previous parity
        else r.addr[v] = h ? -1 : a_p1;                             // c0 == CNT + 1: previous parity, nothing on the odd lane
    }
    uint8_t raw[H];
#pragma unroll
    for (int v = 0; v < H; ++v) raw[v] = pl_present(r, v) ? L.ld_raw(r.addr[v]) : (uint8_t)0;
#pragma unroll
    for (int v = 0; v < H; ++v) {
        r.in[v] = pl_present(r, v) ? t2_clamp(pl_llr_minus_msg(r, v, raw[v]), -128, 127) : 0;
        r.mag[v] = pl_rawmag_of(r, v);
    }
}

// t2_cn_partial: over the node's slots c >= nc (both lanes), result node-wide in p0 / p1 / psx
template <int CNT>
__device__ __forceinline__ void pl_partial(PlRegs<CNT> &r, int nc)
{
    constexpr int H = PlRegs<CNT>::H;
    int m0 = 255, m1 = 255, sx = 0;
#pragma unroll
    for (int v = 0; v < H; ++v) {
        if (2 * v + 1 >= nc) {                                      // uniform: at least the odd lane's slot counts
            const bool mine = (2 * v >= nc) || r.h;
            t2_min2(mine ? pl_rawmag(r, v) : 255, m0, m1);
            sx ^= mine ? r.in[v] : 0;
        }
    }
    pl_merge2(m0, m1, pl_x(m0), pl_x(m1), r.p0, r.p1);
    r.psx = sx ^ pl_x(sx);
}

// t2_cn_merge: fold the conflict slots c < nc into the partial result, apply f. NV = the most slots v a lane can have below nc.
template <int CNT, int NV>
__device__ __forceinline__ void pl_merge(PlRegs<CNT> &r, int nc)
{
    int m0 = 255, m1 = 255, sx = 0;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        if (v < PlRegs<CNT>::H && 2 * v < nc) {                     // uniform: at least the even lane's slot is a conflict slot
            const bool mine = (2 * v + 1 < nc) || !r.h;
            t2_min2(mine ? pl_rawmag(r, v) : 255, m0, m1);
            sx ^= mine ? r.in[v] : 0;
        }
    }
    int c0, c1;
    pl_merge2(m0, m1, pl_x(m0), pl_x(m1), c0, c1);
    sx ^= pl_x(sx);
    int a0, a1;
    pl_merge2(c0, c1, r.p0, r.p1, a0, a1);
    r.m0 = a0; r.m0f = pl_f(a0); r.m1f = pl_f(a1); r.sx = sx ^ r.psx;
}

// t2_write_slot for lane slot v
template <int CNT, class LMEM>
__device__ __forceinline__ void pl_write_slot(LMEM &L, PlRegs<CNT> &r, int v, bool store)
{
    const bool present = pl_present(r, v);
    const bool eq = r.mag[v] == r.m0;                    // an absent slot carries 255 and the node's minimum is at most 128
    const int other = eq ? r.m1f : r.m0f;
    const int sm = (r.sx ^ r.in[v]) >> 31;               // 0 / -1: sign of the product of the other inputs
    int upd;                                             // in + sign * other in one multiply-add (sign = sm | 1 = +1 / -1)
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(upd) : "v"(other), "v"(sm | 1), "v"(r.in[v]));
    const int ln = t2_clamp(upd, -128, 127);
    if (present && store) L.st(r.addr[v], (int8_t)ln);
    const uint32_t bits = ((uint32_t)sm & (1u << (4 * (v & 7)))) | (eq ? (2u << (4 * (v & 7))) : 0u);
    if (v < 8) r.n0 |= bits; else r.n1 |= bits;
}

// t2_cn_pack: both lanes end up with the node's whole record
template <int CNT>
__device__ __forceinline__ void pl_pack(const PlRegs<CNT> &r, CnState &st)
{
    const uint32_t x0 = r.n0 << (2 * r.h);
    st.w0 = x0 | pl_xu(x0);
    if constexpr (PlRegs<CNT>::DEG <= 16) {
        st.w1 = pl_lut_word(r.m0f, r.m1f);
    } else {
        const uint32_t x1 = r.n1 << (2 * r.h);
        st.w1 = ((x1 | pl_xu(x1)) & 0xfffu) | ((uint32_t)(r.m0f > 32 ? 32 : r.m0f) << 16) | ((uint32_t)(r.m1f > 32 ? 32 : r.m1f) << 24);
    }
}

// t2_layer_phase_a
template <int CNT, class LMEM>
__device__ __forceinline__ void pl_phase_a(LMEM &L, const LayerDesc &d, int j, int h, int a_p0, int a_p1, CnState &st, PlRegs<CNT> &r,
                                           uint32_t *pair_rec)
{
    constexpr int H = PlRegs<CNT>::H;
    pl_load<CNT>(L, d.ent_lds, j, h, a_p0, a_p1, st, r);
    T2_CN_HOOK_AFTER_LOAD;
    if (d.kind == T2_LAYER_PLAIN) {
        pl_partial<CNT>(r, 0);
        r.m0 = r.p0; r.m0f = pl_f(r.p0); r.m1f = pl_f(r.p1); r.sx = r.psx;
#pragma unroll
        for (int v = 0; v < H; ++v) pl_write_slot<CNT>(L, r, v, true);
        pl_pack<CNT>(r, st);
    } else if (d.kind == T2_LAYER_PAIR) {
        pl_partial<CNT>(r, 2);
        if (j < d.step) {                       // chain start: its two group bits (slot 0 on the even lane, slot 1 on the odd one) now
            pl_merge<CNT, 1>(r, 2);
            pl_write_slot<CNT>(L, r, 0, true);
        } else if (h == 0) {                    // t2_pair_record: msg1 = old message of slot 1, from the record's code bits 2..3
            pair_rec[j] = t2_pair_pack(t2_lut_byte(r.lut, (st.w0 >> 2) & 3u), t2_f(r.p0), r.in[0], r.psx < 0);
        }
    } else {
        pl_partial<CNT>(r, d.nc);
    }
}

// PAIR phase B for long chains: the walk cut into segments. A node whose other slots leave no room (cap = f(E) = 0: its smallest
// other magnitude is 0 or 1) hands down sat(in0) whatever it receives -- the recurrence forgets its past there. Every such node
// and every chain start walks its own segment up to the next such node; the results are those of the single walk, the length
// is that of the longest segment instead of the chain (N 3/4 has a layer with two chains of 180 nodes: two thirds of all chain
// steps of a sweep, walked by two lanes). Data dependent: with confident LLRs there are no such nodes and nothing changes.
// `node` = 0..359, one lane each; nodes without a successor (>= 360 - step) are not walked, as in t2_pair_walk.
template <class LMEM>
__device__ __forceinline__ void pl_pair_walk_segments(LMEM &L, const LayerDesc &d, int node, const uint32_t *pair_rec)
{
    const uint32_t e0 = d.e0;
    const int base = (int)(e0 & 0xffffu) + L.off(), s0 = (int)(e0 >> 16), step = d.step;
    const int last = 360 - step;                          // nodes below `last` have a successor
    int m = node - s0;                                    // position of the node's slot-0 bit inside its 360-bit group
    m += (m < 0) ? 360 : 0;
    bool go = false;
    int X = 0;
    if (node < step) {                                    // chain start: its bit was finished in phase A
        go = true;
        X = (int)L.ld(base + m);
    } else if (node < last) {
        const PairRec r = t2_pair_unpack(pair_rec[node]);
        if (r.cap == 0) {                                 // segment head: output independent of the input
            go = true;
            X = r.in0;
            L.st(base + m, (int8_t)X);
        }
    }
    int jj = node + step;
    uint32_t nxt = (go && jj < last) ? pair_rec[jj] : 0u;
    while (go && jj < last) {
        const PairRec r = t2_pair_unpack(nxt);
        if (r.cap == 0) break;                            // the next node heads a segment of its own
        jj += step;
        nxt = pair_rec[jj < 359 ? jj : 359];
        const unsigned t = (unsigned)(m + step);
        m = (int)(t < t - 360u ? t : t - 360u);           // (m + step) mod 360
        X = t2_pair_step(r, X);
        L.st(base + m, (int8_t)X);
    }
}

// t2_pair_finish
template <int CNT, class LMEM>
__device__ __forceinline__ void pl_pair_finish(LMEM &L, const LayerDesc &d, int j, CnState &st, PlRegs<CNT> &r)
{
    constexpr int H = PlRegs<CNT>::H;
    if (j < d.step) {
#pragma unroll
        for (int v = 1; v < H; ++v) pl_write_slot<CNT>(L, r, v, true);
        pl_pack<CNT>(r, st);
        return;
    }
    const bool has_succ = j + d.step < 360;
    if (r.h || !has_succ) pl_read_slot<CNT>(L, r, 0);     // slot 1 always, slot 0 when the node ends its chain
    pl_merge<CNT, 1>(r, 2);
#pragma unroll
    for (int v = 0; v < H; ++v) pl_write_slot<CNT>(L, r, v, !(v == 0 && !r.h && has_succ));
    pl_pack<CNT>(r, st);
}

// t2_generic_level_nc
template <int CNT, int NC, class LMEM>
__device__ __forceinline__ void pl_generic_level_nc(LMEM &L, int lv, uint32_t info, PlRegs<CNT> &r)
{
    if ((int)(info & 0xff) != lv) return;
    constexpr int NV = (NC + 1) / 2;
    if (lv > 1) {
#pragma unroll
        for (int v = 0; v < NV; ++v)
            if ((2 * v + 1 < NC) || !r.h) pl_read_slot<CNT>(L, r, v);
    }
    pl_merge<CNT, NV>(r, NC);
#pragma unroll
    for (int v = 0; v < NV; ++v)
        if ((2 * v + 1 < NC) || !r.h) pl_write_slot<CNT>(L, r, v, true);
}

template <int CNT, int NCMAX = T2_LDPC_NC_MAX, class LMEM>
__device__ __forceinline__ void pl_generic_level(LMEM &L, const LayerDesc &d, int lv, uint32_t info, PlRegs<CNT> &r)
{
    switch (d.nc) {
#define T2_NC_(n) case n: if constexpr (n <= CNT && n <= NCMAX) pl_generic_level_nc<CNT, n>(L, lv, info, r); break;
        T2_NC_(2) T2_NC_(3) T2_NC_(4) T2_NC_(5) T2_NC_(6) T2_NC_(7) T2_NC_(8) T2_NC_(9) T2_NC_(10)
#undef T2_NC_
    default: break;
    }
}

// t2_generic_finish
template <int CNT, class LMEM>
__device__ __forceinline__ void pl_generic_finish(LMEM &L, const LayerDesc &d, CnState &st, PlRegs<CNT> &r)
{
    constexpr int H = PlRegs<CNT>::H;
#pragma unroll
    for (int v = 0; v < H; ++v) {
        if (2 * v + 1 >= d.nc) {
            if ((2 * v >= d.nc) || r.h) pl_write_slot<CNT>(L, r, v, true);
        }
    }
    pl_pack<CNT>(r, st);
}

}  // namespace t2gpu
