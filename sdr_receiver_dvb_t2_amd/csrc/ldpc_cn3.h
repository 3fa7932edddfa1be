// ldpc_cn3.h -- the check-node update of ldpc_cn2.h (one check node on two adjacent lanes) for TWO FEC frames at once: the two
// frames of a workgroup share the graph, so every address, every LDS access and every DPP exchange serves both, and the int8
// arithmetic runs in the two 16-bit halves of a register (v_pk_*_i16: two values per lane per issue slot). GPU only.
//
// Layout. LLR byte of bit a of frame f at LDS address base + 2a + f: one ds_read_u16 / ds_write_b16 moves both frames' values.
// Messages are kept as the reference keeps them -- one int8 per link and frame, clamp(out, -32, 31) (algorithms.hh:288-291) --
// in the lane's record: slot v of the lane -> bytes (2(v&1), 2(v&1)+1) = (frame A, frame B) of record dword v >> 1. The 2-bit codes
// + two minima of ldpc_cn.h / ldpc_cn2.h are the compact form for one frame per register; with two frames per register the
// message bytes are what the packed subtract consumes directly (two SDWA subtracts per slot for both frames).
//
// Scale. Inside a node the int8 quantities are carried times 256, i.e. in the HIGH byte of their 16-bit half: the reference's two
// saturations to int8 (L - message, input + output) then ARE the 16-bit saturations of v_pk_sub_i16 / v_pk_mad_i16 with the clamp
// bit -- one instruction each instead of three -- and bytes move between memory and halves with one v_perm_b32 either way. The one
// value that is not a multiple of 256 is the positive saturation 32767 (for 127 * 256 = 32512): every consumer either takes the
// high byte (127) or applies f, which caps at 126 * 256 far below, or compares it with values of which it is the largest anyway.
// Raw magnitudes are unsigned halves: 128 * 256 = 0x8000 (from -128) and the absent slot's 255 * 256 order correctly.
//
// Semantics, slot numbering, layer kinds and phase structure are those of ldpc_cn.h / ldpc_cn2.h, function by function (each names
// the one it restates); results are LLR-exact against the same goldens (tests/test_ldpc_gpu.py).
#pragma once
#include "ldpc_cn.h"


namespace t2gpu {

typedef short p16 __attribute__((ext_vector_type(2)));     // (frame A, frame B)
typedef unsigned short q16 __attribute__((ext_vector_type(2)));   // the same halves read as unsigned (raw magnitudes)
__device__ __forceinline__ p16 p_of(uint32_t v) { return __builtin_bit_cast(p16, v); }
__device__ __forceinline__ uint32_t u_of(p16 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ q16 q_of(p16 v) { return __builtin_bit_cast(q16, v); }
__device__ __forceinline__ p16 p_ofq(q16 v) { return __builtin_bit_cast(p16, v); }
__device__ __forceinline__ p16 p_set(int a) { return (p16){(short)a, (short)a}; }
__device__ __forceinline__ p16 p_min(p16 a, p16 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ p16 p_max(p16 a, p16 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ p16 p_clamp(p16 v, int lo, int hi) { return p_min(p_max(v, p_set(lo)), p_set(hi)); }
// unsigned minimum / maximum of the halves (raw magnitudes)
__device__ __forceinline__ p16 u_min(p16 a, p16 b) { return p_ofq(__builtin_elementwise_min(q_of(a), q_of(b))); }
__device__ __forceinline__ p16 u_max(p16 a, p16 b) { return p_ofq(__builtin_elementwise_max(q_of(a), q_of(b))); }
__device__ __forceinline__ int p_a(p16 v) { return (int)v.x; }      // frame A / frame B value, sign-extended
__device__ __forceinline__ int p_b(p16 v) { return (int)v.y; }
constexpr int P2_ONE = 256;                                  // the scale
constexpr int P2_ABSENT = 0xff00;                            // raw magnitude of a slot the node does not have (255)

// value of v in the partner lane (lane ^ 1)
__device__ __forceinline__ uint32_t p2_xu(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ p16 p2_x(p16 v) { return p_of(p2_xu(u_of(v))); }

// f of ldpc_cn.h on raw magnitudes, both frames: max(a - 1, 0) capped at 126, times 256
__device__ __forceinline__ p16 p2_f(p16 a)
{
    const q16 d = __builtin_elementwise_sub_sat(q_of(a), (q16){(unsigned short)P2_ONE, (unsigned short)P2_ONE});
    return p_ofq(__builtin_elementwise_min(d, (q16){(unsigned short)(126 * P2_ONE), (unsigned short)(126 * P2_ONE)}));
}

// sat((LLR bytes A, B in bytes 0, 1 of raw) - (message bytes A, B in bytes 2*odd, 2*odd + 1 of msg)), times 256, one frame per
// half: either pair of bytes goes to the high bytes of the halves with one v_perm_b32, the subtract saturates (v_pk_sub_i16 clamp)
__device__ __forceinline__ p16 p2_llr_minus_msg(uint32_t raw, uint32_t msg, int odd)       // raw: the zero-extended 16-bit load (kept 32 bits
{                                                                                            // wide: as uint16_t it cost a v_and per slot)
    const p16 l = p_of(__builtin_amdgcn_perm(0u, raw, 0x010c000cu));
    const p16 m = p_of(__builtin_amdgcn_perm(0u, msg, odd ? 0x030c020cu : 0x010c000cu));
    return __builtin_elementwise_sub_sat(l, m);
}
__device__ __forceinline__ p16 p2_abs(p16 x) { return p_max(x, p_set(0) - x); }   // |-128 * 256| = 0x8000: right as an unsigned half

// the two smallest of the union of two sorted pairs (pl_merge2)
__device__ __forceinline__ void p2_merge2(p16 a0, p16 a1, p16 b0, p16 b1, p16 &m0, p16 &m1)
{
    m0 = u_min(a0, b0);
    m1 = u_min(u_max(a0, b0), u_min(a1, b1));
}
// t2_min2: fold a into the sorted pair (m0 <= m1)
__device__ __forceinline__ void p2_min2(p16 a, p16 &m0, p16 &m1)
{
    m1 = u_min(m1, u_max(m0, a));
    m0 = u_min(m0, a);
}

template <int CNT>
struct P2Regs {
    static constexpr int DEG = CNT + 2, H = (DEG + 1) / 2, W = (H + 1) / 2;   // slots of the node / of one lane / record dwords of one lane
    int addr[H];                // LDS address of the slot's LLR pair (absent slot: < 0)
    p16 in[H], mag[H];          // in = sat(L - old message) and its raw magnitude, times 256 (P2_ABSENT for an absent slot)
    uint32_t mo[W], mn[W];      // old / new message bytes
    p16 p0, p1;                 // node-wide over the non-conflict slots: two smallest raw magnitudes
    uint32_t psx;               //   ... and the sign xor (bits 15 / 31)
    p16 m0, m1f, dm;            // node-wide over all slots: raw minimum, f(min1), f(min0) - f(min1)
    uint32_t sx;
    int h;
    bool last_present;          // the lane's last slot (the parity slot beyond the information links) holds a bit
};

template <int CNT>
__device__ __forceinline__ bool p2_present(const P2Regs<CNT> &r, int v) { return (2 * v + 1 <= CNT) || r.last_present; }

// pl_read_slot
template <int CNT, class LMEM>
__device__ __forceinline__ void p2_read_slot(const LMEM &L, P2Regs<CNT> &r, int v)
{
    const bool present = p2_present(r, v);
    const uint32_t raw = present ? (uint32_t)L.ld16(r.addr[v]) : 0u;
    const p16 x = p2_llr_minus_msg(raw, r.mo[v >> 1], v & 1);
    r.in[v] = present ? x : p_set(0);
    r.mag[v] = present ? p2_abs(x) : p_set(P2_ABSENT);
}

// the lane's table entries of a layer: slot v <- entry 2v + h (information slots), from the LDS copy, issued together
template <int CNT, class LMEM>
__device__ __forceinline__ void p2_entries(const LMEM &L, int ent_lds, int h, uint2 (&e)[(CNT + 3) / 2])
{
    const int mine = ent_lds + 8 * h;                              // the odd lane reads the odd entries
#pragma unroll
    for (int v = 0; v < (CNT + 3) / 2; ++v) {
        const int c0 = 2 * v, c1 = 2 * v + 1;
        if (c1 < CNT) e[v] = L.ld_pair(mine + 8 * c0);
        else if (c0 < CNT) e[v] = L.ld_pair(ent_lds + 8 * c0);
        else e[v] = make_uint2(0u, 0u);
    }
}

// pl_load. e: the lane's entries as (base + 2 * (bit base - shift), shift) pairs (p2_entries); j2 = 2 j.
// a_par: the ONE parity-bit address this lane needs -- with an even link count the even lane holds the node's own parity bit and the
// odd lane the previous one in their last slot; with an odd count the odd lane holds the own bit (beside its last information link)
// and the even lane the previous one -- always a valid address: par_absent marks the one lane whose previous parity bit does not
// exist (node (0, 0)), it reads the scratch pair at `dummy` and discards it. (Rounds 2-4 passed both addresses with -1 for "absent" and
// paid a compare, two selects and a max per layer to find out again what the caller knew.)
// PF: e is overwritten with the NEXT layer's entries (at next_ent_lds) once this layer's addresses are formed -- the reads are issued
// behind this layer's LLR reads (LDS returns in order: nothing waits for them before the next layer) and land in the registers the
// next layer reads them from, so the layer loop carries one set of entry registers and no copies. Compile-time, so that the loads and
// their consumers stay in one basic block (the zero-extension of a 16-bit load is folded into it only there).
template <int CNT, bool PF, class LMEM>
__device__ __forceinline__ void p2_load(const LMEM &L, uint2 (&e)[(CNT + 3) / 2], int j, int h, int a_par, bool par_absent, int dummy, P2Regs<CNT> &r,
                                        int next_ent_lds)
{
    constexpr int H = P2Regs<CNT>::H, W = P2Regs<CNT>::W;
    r.h = h;
    r.last_present = (CNT % 2 == 0) ? !par_absent : (h ? false : !par_absent);
#pragma unroll
    for (int w = 0; w < W; ++w) r.mn[w] = 0u;
    const int j2 = 2 * j, jw2 = j2 + 720;
    uint32_t raw[H];
#pragma unroll
    for (int v = 0; v < H; ++v) {
        const int c0 = 2 * v, c1 = 2 * v + 1;
        const int link = (int)e[v].x + (j >= (int)e[v].y ? j2 : jw2);
        if (c1 < CNT) r.addr[v] = link;                             // information slots on both lanes
        else if (c0 < CNT) r.addr[v] = h ? a_par : link;            // c1 == CNT: own parity bit on the odd lane
        else if (c0 == CNT) r.addr[v] = a_par;                      // own parity (even lane) / previous parity (odd lane)
        else r.addr[v] = h ? dummy : a_par;                         // c0 == CNT + 1: previous parity, nothing on the odd lane
        raw[v] = (uint32_t)L.ld16(r.addr[v]);
    }
    if constexpr (PF) p2_entries<CNT>(L, next_ent_lds, h, e);
    // the old messages' bytes into the halves (nothing here depends on the reads: the compiler fills the address arithmetic's hazard
    // slots with these; explicit scheduling groups -- read after every three instructions -- measured 2 % slower)
    p16 m[H];
#pragma unroll
    for (int v = 0; v < H; ++v) m[v] = p_of(__builtin_amdgcn_perm(0u, r.mo[v >> 1], (v & 1) ? 0x030c020cu : 0x010c000cu));
#pragma unroll
    for (int v = 0; v < H; ++v) {
        const bool present = p2_present(r, v);
        const p16 l = p_of(__builtin_amdgcn_perm(0u, raw[v], 0x010c000cu));
        const p16 x = __builtin_elementwise_sub_sat(l, m[v]);
        r.in[v] = present ? x : p_set(0);
        r.mag[v] = present ? p2_abs(x) : p_set(P2_ABSENT);
    }
}

// pl_partial: over the node's slots c >= nc (both lanes), result node-wide in p0 / p1 / psx
template <int CNT>
__device__ __forceinline__ void p2_partial(P2Regs<CNT> &r, int nc)
{
    constexpr int H = P2Regs<CNT>::H;
    p16 m0 = p_set(P2_ABSENT), m1 = p_set(P2_ABSENT);
    uint32_t sx = 0;
#pragma unroll
    for (int v = 0; v < H; ++v) {
        if (2 * v + 1 >= nc) {                                      // uniform: at least the odd lane's slot counts
            const bool mine = (2 * v >= nc) || r.h;
            p2_min2(mine ? r.mag[v] : p_set(P2_ABSENT), m0, m1);
            sx ^= mine ? u_of(r.in[v]) : 0u;
        }
    }
    p2_merge2(m0, m1, p2_x(m0), p2_x(m1), r.p0, r.p1);
    r.psx = sx ^ p2_xu(sx);
}

__device__ __forceinline__ void p2_set_minima(p16 a0, p16 a1, p16 &m0, p16 &m1f, p16 &dm)
{
    m0 = a0;
    m1f = p2_f(a1);
    dm = p2_f(a0) - m1f;
}

// pl_merge: fold the conflict slots c < nc into the partial result, apply f. NV = the most slots v a lane can have below nc.
template <int CNT, int NV>
__device__ __forceinline__ void p2_merge(P2Regs<CNT> &r, int nc)
{
    p16 m0 = p_set(P2_ABSENT), m1 = p_set(P2_ABSENT);
    uint32_t sx = 0;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        if (v < P2Regs<CNT>::H && 2 * v < nc) {                     // uniform: at least the even lane's slot is a conflict slot
            const bool mine = (2 * v + 1 < nc) || !r.h;
            p2_min2(mine ? r.mag[v] : p_set(P2_ABSENT), m0, m1);
            sx ^= mine ? u_of(r.in[v]) : 0u;
        }
    }
    p16 c0, c1, a0, a1;
    p2_merge2(m0, m1, p2_x(m0), p2_x(m1), c0, c1);
    sx ^= p2_xu(sx);
    p2_merge2(c0, c1, r.p0, r.p1, a0, a1);
    p2_set_minima(a0, a1, r.m0, r.m1f, r.dm);
    r.sx = sx ^ r.psx;
}

// pl_write_slot for lane slot v: a-posteriori update L = sat(in + out) and the new message clamp(out, -32, 31), both frames
template <int CNT, class LMEM>
__device__ __forceinline__ void p2_write_slot(LMEM &L, P2Regs<CNT> &r, int v, bool store)
{
    const bool present = p2_present(r, v);
    // 0 where the slot holds the raw minimum (an absent slot carries 255), else 1; then f(min1) for the minimum's slot, f(min0) for
    // the others. As plain C the compiler turns min(d, 1) * dm + m1f into a compare and a select PER HALF plus a v_perm to join them
    // (five instructions and a hazard nop); these are the three it should be.
    uint32_t s_, other_;
    asm("v_pk_sub_u16 %0, %1, %2" : "=v"(s_) : "v"(u_of(r.mag[v])), "v"(u_of(r.m0)));
    asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(s_) : "v"(s_));
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(other_) : "v"(s_), "v"(u_of(r.dm)), "v"(u_of(r.m1f)));
    const p16 other = p_of(other_);
    const p16 sm = p_of(r.sx ^ u_of(r.in[v])) >> p_set(15);   // 0 / -1: sign of the product of the other inputs
    const p16 sgn = p_of(u_of(sm) | 0x00010001u);        // +1 / -1
    const p16 out = other * sgn;
    const p16 ln = __builtin_elementwise_add_sat(out, r.in[v]);   // sat(in + out): the 16-bit saturation
    if (present && store) L.st16(r.addr[v], __builtin_amdgcn_perm(0u, u_of(ln), 0x0c0c0301u));
    const p16 msg = p_clamp(out, -32 * P2_ONE, 31 * P2_ONE);
    // bytes 1, 3 of msg -> bytes (0, 1) or (2, 3) of the record dword
    if (v & 1) r.mn[v >> 1] = __builtin_amdgcn_perm(u_of(msg), r.mn[v >> 1], 0x07050100u);
    else r.mn[v >> 1] = __builtin_amdgcn_perm(u_of(msg), r.mn[v >> 1], 0x03020705u);
}

// pl_phase_a. pair_rec: [2][360] chain-walk records, one array per frame
template <int CNT, bool PF, class LMEM>
__device__ __forceinline__ void p2_phase_a(LMEM &L, const LayerDesc &d, uint2 (&e)[(CNT + 3) / 2], int j, int h, int a_par, bool par_absent,
                                           P2Regs<CNT> &r, uint32_t *pair_rec, int next_ent_lds)
{
    constexpr int H = P2Regs<CNT>::H;
    p2_load<CNT, PF>(L, e, j, h, a_par, par_absent, d.dummy, r, next_ent_lds);
    if (d.kind == T2_LAYER_PLAIN) {
        p2_partial<CNT>(r, 0);
        p2_set_minima(r.p0, r.p1, r.m0, r.m1f, r.dm);
        r.sx = r.psx;
#pragma unroll
        for (int v = 0; v < H; ++v) p2_write_slot<CNT>(L, r, v, true);
    } else if (d.kind == T2_LAYER_PAIR) {
        p2_partial<CNT>(r, 2);
        const uint32_t partner_m0 = p2_xu(r.mo[0]);      // the odd lane's slot v = 0 is the node's slot 1: its old message is msg1
        if (j < d.step) {                       // chain start: its two group bits (slot 0 on the even lane, slot 1 on the odd one) now
            p2_merge<CNT, 1>(r, 2);
            p2_write_slot<CNT>(L, r, 0, true);
        } else if (h == 0) {
            L.st16(d.pair_flag_lds + 2 * j, 0u);          // "the bit this node shares with its predecessor is final": not yet (both frames)
            // t2_pair_record / t2_pair_pack for both frames at once, in the halves: k1 = (neg ? msg1 : -msg1) - 1, cap = f(p0), in0 and
            // the sign field (neg ? 3 : 1) are formed times 256 (the sign field plain) and their bytes gathered with four v_perm_b32
            const p16 m1 = p_of(__builtin_amdgcn_perm(0u, partner_m0, 0x010c000cu));      // slot 1's old messages, times 256
            const p16 pos = p_of(~r.psx) >> p_set(15);                                    // -1 where the other slots' sign product is +
            const p16 k1 = p_of(u_of(m1) ^ u_of(pos)) - pos - p_set(P2_ONE);
            const uint32_t sg = ((r.psx >> 14) & 0x00020002u) | 0x00010001u;
            const uint32_t t = __builtin_amdgcn_perm(u_of(p2_f(r.p0)), u_of(k1), 0x07030501u);   // (k1 A, cap A, k1 B, cap B)
            const uint32_t u = __builtin_amdgcn_perm(sg, u_of(r.in[0]), 0x06030401u);            // (in0 A, sign A, in0 B, sign B)
            pair_rec[j] = __builtin_amdgcn_perm(u, t, 0x05040100u);
            pair_rec[360 + j] = __builtin_amdgcn_perm(u, t, 0x07060302u);
        }
    } else {
        p2_partial<CNT>(r, d.nc);
        if (d.band) {
            // t2 band record of the node, one per frame: dword 0 = old messages of conflict slots 0..3, dword 1 = the partial result
            // (two smallest raw magnitudes of the other slots, their sign product in bit 16). Slots 0, 2 live on the even lane (bytes
            // 0/1 and 2/3 of its first record dword = frame A/B), slots 1, 3 on the odd one.
            const uint32_t po = p2_xu(r.mo[0]);
            if (h == 0) {
                const uint32_t ma = __builtin_amdgcn_perm(po, r.mo[0], 0x06020400u), mb = __builtin_amdgcn_perm(po, r.mo[0], 0x07030501u);
                const uint32_t p0 = u_of(r.p0), p1 = u_of(r.p1);
                const uint32_t qa = ((p0 >> 8) & 0xffu) | (p1 & 0xff00u) | (((r.psx >> 15) & 1u) << 16);
                const uint32_t qb = (p0 >> 24) | ((p1 >> 16) & 0xff00u) | ((r.psx >> 31) << 16);
                L.st_pair(d.band_rec_lds + 8 * j, ma, qa);
                L.st_pair(d.band_rec_lds + 2880 + 8 * j, mb, qb);
            }
        }
    }
}

// Chain walks run one frame per lane on the unpacked recurrence of ldpc_cn.h (t2_pair_step): the walk is a latency chain, so the two
// frames' chains walking side by side on different lanes cost the time of one. base2 = LDS address of bit 0 of the group for this
// frame (lds base + 2 * bit base + frame); positions advance by 2 bytes.
template <class LMEM>
__device__ __forceinline__ void p2_pair_walk_segments(LMEM &L, const LayerDesc &d, int node, int frame, const uint32_t *pair_rec)
{
    const uint32_t e0 = d.e0;
    const int base2 = 2 * (int)(e0 & 0xffffu) + L.off() + frame, s0 = (int)(e0 >> 16), step = d.step;
    const int last = 360 - step;                          // nodes below `last` have a successor
    int m = node - s0;                                    // position of the node's slot-0 bit inside its 360-bit group
    m += (m < 0) ? 360 : 0;
    bool go = false;
    int X = 0;
    const int flag = d.pair_flag_lds + frame;             // flag of node n at flag + 2 n: set once the bit n shares with n - step is final
    if (node < step) {                                    // chain start: its bit was finished in phase A
        go = true;
        X = (int)L.ld(base2 + 2 * m);
    } else if (node < last) {
        const PairRec r = t2_pair_unpack(pair_rec[node]);
        if (r.cap == 0) {                                 // segment head: output independent of the input (ldpc_cn2.h)
            go = true;
            X = r.in0;
            L.st(base2 + 2 * m, (int8_t)X);
        }
    }
    int jj = node + step;
    if (go && jj < 360) L.st(flag + 2 * jj, (int8_t)1);   // after the read / store above: LDS accesses of a wavefront execute in order
    uint32_t nxt = (go && jj < last) ? pair_rec[jj] : 0u;
    while (go && jj < last) {
        const PairRec r = t2_pair_unpack(nxt);
        if (r.cap == 0) break;                            // the next node heads a segment of its own
        jj += step;
        nxt = pair_rec[jj < 359 ? jj : 359];
        const unsigned t = (unsigned)(m + step);
        m = (int)(t < t - 360u ? t : t - 360u);           // (m + step) mod 360
        X = t2_pair_step(r, X);
        L.st(base2 + 2 * m, (int8_t)X);
        L.st(flag + 2 * jj, (int8_t)1);
    }
}

// t2_pair_walk for short chains (lane < step walks chain lane, lane + step, ...), one frame
template <class LMEM>
__device__ __forceinline__ void p2_pair_walk(LMEM &L, const LayerDesc &d, int lane, int frame, const uint32_t *pair_rec)
{
    const uint32_t e0 = d.e0;
    const int base2 = 2 * (int)(e0 & 0xffffu) + L.off() + frame, s0 = (int)(e0 >> 16), step = d.step;
    int m = lane - s0;
    m += (m < 0) ? 360 : 0;
    int X = (int)L.ld(base2 + 2 * m);
    const int flag = d.pair_flag_lds + frame;
    // the records are read one node ahead of the recurrence (they do not depend on it): a step then costs its arithmetic, not
    // an LDS round trip on top
    int jj = lane + step;
    L.st(flag + 2 * jj, (int8_t)1);                               // node lane + step may read its bit (after the read above)
    uint32_t nxt = pair_rec[jj < 359 ? jj : 359];
    for (; jj + step < 360; jj += step) {                         // nodes that have a successor
        const PairRec r = t2_pair_unpack(nxt);
        const int jn = jj + step;
        nxt = pair_rec[jn < 359 ? jn : 359];
        const unsigned t = (unsigned)(m + step);
        m = (int)(t < t - 360u ? t : t - 360u);
        X = t2_pair_step(r, X);
        L.st(base2 + 2 * m, (int8_t)X);
        L.st(flag + 2 * jn, (int8_t)1);
    }
}

// pl_pair_finish
template <int CNT, class LMEM>
__device__ __forceinline__ void p2_pair_finish(LMEM &L, const LayerDesc &d, int j, P2Regs<CNT> &r)
{
    constexpr int H = P2Regs<CNT>::H;
    if (j < d.step) {
#pragma unroll
        for (int v = 1; v < H; ++v) p2_write_slot<CNT>(L, r, v, true);
        return;
    }
    // wait until the walks of both frames have passed this node's predecessor (no barrier behind the walks: a wavefront goes on as
    // soon as its own nodes are served, so the finishes of the early nodes run beside the walks of the late ones)
#ifndef T2_PAIR_SLEEP
#define T2_PAIR_SLEEP 1
#endif
    while (L.ld16_volatile(d.pair_flag_lds + 2 * j) != 0x0101u) { if (T2_PAIR_SLEEP) __builtin_amdgcn_s_sleep(T2_PAIR_SLEEP); }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const bool has_succ = j + d.step < 360;
    if (r.h || !has_succ) p2_read_slot<CNT>(L, r, 0);     // slot 1 always, slot 0 when the node ends its chain
    p2_merge<CNT, 1>(r, 2);
#pragma unroll
    for (int v = 0; v < H; ++v) p2_write_slot<CNT>(L, r, v, !(v == 0 && !r.h && has_succ));
}

// pl_generic_level_nc
template <int CNT, int NC, class LMEM>
__device__ __forceinline__ void p2_generic_level_nc(LMEM &L, int lv, uint32_t info, P2Regs<CNT> &r)
{
    if ((int)(info & 0xff) != lv) return;
    constexpr int NV = (NC + 1) / 2;
    if (lv > 1) {
#pragma unroll
        for (int v = 0; v < NV; ++v)
            if ((2 * v + 1 < NC) || !r.h) p2_read_slot<CNT>(L, r, v);
    }
    p2_merge<CNT, NV>(r, NC);
#pragma unroll
    for (int v = 0; v < NV; ++v)
        if ((2 * v + 1 < NC) || !r.h) p2_write_slot<CNT>(L, r, v, true);
}

template <int CNT, int NCMAX = T2_LDPC_NC_MAX, class LMEM>
__device__ __forceinline__ void p2_generic_level(LMEM &L, const LayerDesc &d, int lv, uint32_t info, P2Regs<CNT> &r)
{
    switch (d.nc) {
#define T2_NC_(n) case n: if constexpr (n <= CNT && n <= NCMAX) p2_generic_level_nc<CNT, n>(L, lv, info, r); break;
        T2_NC_(2) T2_NC_(3) T2_NC_(4) T2_NC_(5) T2_NC_(6) T2_NC_(7) T2_NC_(8) T2_NC_(9) T2_NC_(10)
#undef T2_NC_
    default: break;
    }
}

// pl_generic_finish
template <int CNT, class LMEM>
__device__ __forceinline__ void p2_generic_finish(LMEM &L, const LayerDesc &d, P2Regs<CNT> &r)
{
    constexpr int H = P2Regs<CNT>::H;
#pragma unroll
    for (int v = 0; v < H; ++v) {
        if (2 * v + 1 >= d.nc) {
            if ((2 * v >= d.nc) || r.h) p2_write_slot<CNT>(L, r, v, true);
        }
    }
}

// ---- band walk (GENERIC layers with at most four conflict slots, ldpc_graph.h) ---------------------------------------------------
// Lane group c < D walks nodes c, c + D, c + 2D, ... : band t = nodes [D t, D t + D) at step t, in the reference's ascending-j order
// band by band (nodes of a band share no bit). A step is the conflict-slot part of the check-node update on plain 32-bit integers,
// one frame per lane -- inputs L - old message of the conflict slots, merged with the node's partial result from phase A, new
// a-posteriori values of those slots stored to the LLR memory -- and it leaves the inputs behind for the finish (p2_band_finish),
// which builds the messages. A node is spread over LPN = 4, 2 or 1 adjacent lanes (slot c on sub-lane c mod LPN; minima and sign
// product meet through quad DPP): the walk is one wavefront issuing for one band at a time, so the step costs its instruction
// count, and LPN is the most that keeps D x LPN lanes inside the wavefront. The value slot 0 hands to node j + D (its slot 1) stays
// in registers: no LDS round trip on the recurrence. pf: everything else a node reads was written two bands earlier or more, so
// its loads are issued one step ahead. LDS accesses of a wavefront execute in order: no barrier between bands.
template <int CTRL>
__device__ __forceinline__ int p2_quad(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }

template <int NC, int LPN, bool PF, class LMEM>
__device__ __forceinline__ void p2_band_walk(LMEM &L, const LayerDesc &d, int lane, int frame)
{
    constexpr int K = 4 / LPN;                                      // slots a lane serves (slot c = sub + LPN k; absent when c >= NC)
    const int D = d.band, sub = lane & (LPN - 1);
    const int rec_lds = d.band_rec_lds + 2880 * frame;
    int in_at = d.band_in_lds + 1440 * frame + sub + 4 * (lane / LPN);
    int at[K], es[K], sh[K];
    bool has[K];
    int j = lane / LPN;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int c = sub + LPN * k;
        has[k] = NC == 4 || c < NC;
        const uint2 e = L.ld_pair(d.ent_lds + 8 * (has[k] ? c : 0));
        es[k] = (int)e.y;
        at[k] = (int)e.x + frame + 2 * j + (j >= es[k] ? 0 : 720);  // LLR byte of the slot for node j
        sh[k] = 8 * c;
    }
    uint2 rec = L.ld_pair(rec_lds + 8 * j);
    int Lv[K];
#pragma unroll
    for (int k = 0; k < K; ++k) Lv[k] = (int)L.ld(at[k]);
    // every load above has landed before the loop: a load still pending at the loop head costs an lgkmcnt(0) there in EVERY step,
    // which also drains the stores the step before has just issued
    __builtin_amdgcn_s_waitcnt(0xc07f);
    while (j < 360) {
        // the next node of this lane group (the last step repeats its own: nothing reads what that loads)
        const int jn = j + D < 360 ? j + D : j, dj2 = 2 * (jn - j);
        int an[K];
#pragma unroll
        for (int k = 0; k < K; ++k) an[k] = at[k] + dj2 - ((j < es[k] && jn >= es[k]) ? 720 : 0);
        uint2 recn;
        int Ln[K];
        if constexpr (PF) {
            recn = L.ld_pair(rec_lds + 8 * jn);
#pragma unroll
            for (int k = 0; k < K; ++k) Ln[k] = (int)L.ld(an[k]);
        }
        int x[K], mg[K];
        int m0 = 0, m1 = 255, sx = (int)(rec.y << 15);              // bit 31 = sign product of the other slots
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int mo = __builtin_amdgcn_sbfe((int)rec.x, sh[k], 8);
            const int v = t2_clamp(Lv[k] - mo, -128, 127);
            x[k] = has[k] ? v : 0;
            mg[k] = has[k] ? (v < 0 ? -v : v) : 255;
            sx ^= x[k];
            if (k == 0) m0 = mg[0];
            else {
                m1 = min(m1, max(m0, mg[k]));
                m0 = min(m0, mg[k]);
            }
        }
        if constexpr (LPN >= 2) {
            const int b0 = p2_quad<0xB1>(m0), b1 = p2_quad<0xB1>(m1);
            m1 = min(max(m0, b0), min(m1, b1));
            m0 = min(m0, b0);
            sx ^= p2_quad<0xB1>(sx);
        }
        if constexpr (LPN >= 4) {
            const int b0 = p2_quad<0x4E>(m0), b1 = p2_quad<0x4E>(m1);
            m1 = min(max(m0, b0), min(m1, b1));
            m0 = min(m0, b0);
            sx ^= p2_quad<0x4E>(sx);
        }
        if constexpr (LPN >= 2) sx ^= (int)(rec.y << 15);           // every lane started from the partial sign: an even count cancels
        {
            const int p0 = (int)(rec.y & 0xffu), p1 = (int)((rec.y >> 8) & 0xffu);
            m1 = min(max(m0, p0), min(m1, p1));
            m0 = min(m0, p0);
        }
        int hand = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int other = t2_clamp((mg[k] == m0 ? m1 : m0) - 1, 0, 126);
            const int sgn = ((sx ^ x[k]) >> 31) | 1;
            int sum;                                                // +-other + x
            asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(sum) : "v"(other), "v"(sgn), "v"(x[k]));
            const int ln = t2_clamp(sum, -128, 127);
            L.st(has[k] ? at[k] : d.dummy, (int8_t)ln);
            L.st(has[k] ? in_at + LPN * k : d.dummy, (int8_t)x[k]);
            if (k == 0) hand = ln;
        }
        if constexpr (!PF) {
            recn = L.ld_pair(rec_lds + 8 * jn);
#pragma unroll
            for (int k = 0; k < K; ++k) Ln[k] = (int)L.ld(an[k]);
        }
        // slot 1 of node j + D is the bit slot 0 of node j just wrote
        if constexpr (LPN == 1) Ln[1] = hand;
        else {
            const int from0 = p2_quad<(LPN == 2 ? 0xA0 : 0x00)>(hand);
            Ln[0] = sub == 1 ? from0 : Ln[0];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) { Lv[k] = Ln[k]; at[k] = an[k]; }
        rec = recn;
        in_at += 4 * D;
        j += D;
    }
}

// what p2_generic_level_nc + p2_generic_finish do, after a band walk: the walk has stored the conflict slots' new LLRs and left their
// inputs; here the node forms its minima over all slots and the new messages of every slot, and stores the other slots' LLRs.
template <int CNT, int NC, class LMEM>
__device__ __forceinline__ void p2_band_finish(LMEM &L, const LayerDesc &d, int j, P2Regs<CNT> &r)
{
    constexpr int H = P2Regs<CNT>::H, NV = (NC + 1) / 2;
    const uint32_t wa = L.ld32(d.band_in_lds + 4 * j), wb = L.ld32(d.band_in_lds + 1440 + 4 * j);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        if ((2 * v + 1 < NC) || !r.h) {
            // byte 2v + h of either word into the high bytes of the two halves: (frame A, frame B) times 256
            const uint32_t sel = r.h ? (v ? 0x070c030cu : 0x050c010cu) : (v ? 0x060c020cu : 0x040c000cu);
            const p16 x = p_of(__builtin_amdgcn_perm(wb, wa, sel));
            r.in[v] = x;
            r.mag[v] = p2_abs(x);
        }
    }
    p2_merge<CNT, NV>(r, NC);
#pragma unroll
    for (int v = 0; v < H; ++v) {
        p2_write_slot<CNT>(L, r, v, 2 * v + r.h >= NC);             // the conflict slots' LLRs were stored by the walk
    }
}

}  // namespace t2gpu
