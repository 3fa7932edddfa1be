// ldpc_kernel.hip -- layered int8 offset-min-sum LDPC decoder for DVB-T2 on gfx950 (MI355X).
//
// Replaces the compute of ldpc_decoder::execute (/root/reference/src/DVB_T2/ldpc_decoder.cpp:157-301) and
// LDPCDecoder::{bad,update,operator()} (/root/reference/src/DVB_T2/LDPC/layered_decoder.hh:65-110,168-180).
//
// Mapping (see DESIGN.md "K-ldpc"):
//   * one workgroup decodes one FEC frame; the frame's 64 800 (16 200) a-posteriori LLR bytes live in LDS for the
//     whole decode, HBM is touched once on the way in (LLRs) and once on the way out (hard bits);
//   * thread j owns check node (i, j) of every layer i: a wavefront of consecutive j reads/writes consecutive LLR
//     bytes of a 360-bit group (quasi-cyclic shift) -> conflict-free ds_read_u8 / ds_write_b8;
//   * the per-link messages of the reference (226 799 bytes) are replaced by an 8-byte record per check node
//     (ldpc_cn.h) that streams through L2, coalesced, prefetched one layer ahead and never waited for at a barrier;
//   * layers whose nodes share bits keep the reference's ascending-j order exactly (ldpc_cn.h: PAIR chains walked in
//     registers, GENERIC levels): results are LLR-exact, not merely codeword-exact;
//   * the reference decodes SIMD batches of 32 frames that stop together (all 32 parity-clean, or 25 updates). The
//     32 workgroups of such a batch are co-resident (persistent grid) and agree on every stop decision through one
//     atomic word per (batch, trial).
#include <hip/hip_runtime.h>
#include <stdint.h>
#ifndef T2_PROF_DETAIL
#define T2_PROF_DETAIL 0      // diagnostics builds: 1 splits the PAIR layers, 2 the GENERIC layers, 3 the PLAIN layers into slots 3 / 4 / 5
#endif
#if T2_PROF_DETAIL == 3
// PLAIN layers: slot 3 = table entries + addresses + LDS loads + input arithmetic, slot 4 = minima, outputs, stores, slot 5 = barrier
namespace t2gpu { __device__ __forceinline__ void hook_after_load(); }
#define T2_CN_HOOK_AFTER_LOAD t2gpu::hook_after_load()
#endif
#include "ldpc_kernel.h"
#include "ldpc_cn.h"
#if T2_LDPC_PAIRLANE
#include "ldpc_cn2.h"
#endif

namespace t2gpu {

// LLR bytes in LDS, addressed by their 32-bit LDS address: the array's LDS offset is folded into the scalar part of every
// address computation (ldpc_cn.h, L.off()), so a load or store is one ds instruction without a per-lane add.
typedef __attribute__((address_space(3))) int8_t lds_i8;
struct LdsMem {
    uint32_t base;
    __device__ __forceinline__ int off() const { return (int)base; }
    __device__ __forceinline__ int8_t ld(int a) const { return *reinterpret_cast<const lds_i8 *>((uint32_t)a); }
    __device__ __forceinline__ void st(int a, int8_t v) { *reinterpret_cast<lds_i8 *>((uint32_t)a) = v; }
    __device__ __forceinline__ uint8_t ld_raw(int a) const       // the LLR byte as ds_read_u8 delivers it; the upper bits are nobody's business
    {
        return *reinterpret_cast<const __attribute__((address_space(3))) uint8_t *>((uint32_t)a);
    }
    __device__ __forceinline__ uint2 ld_pair(int a) const
    {
        const __attribute__((address_space(3))) uint32_t *q = reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>((uint32_t)a);
        return make_uint2(q[0], q[1]);
    }
};

static constexpr int kThreads = T2GPU_LDPC_THREADS;   // 6 wavefronts, 360 of 384 lanes own a check node -- or 12 and two lanes per node

// Workgroup barrier that orders LDS traffic only. __syncthreads() would also drain the vector-memory queue
// (s_waitcnt vmcnt(0)) and so expose the L2 round trip of the record loads/stores at every layer; those records are
// private to one thread, nothing another wave reads, so they may stay in flight across the barrier.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ int parity_prev_addr(int k, int q, int i, int j)
{
    if (i > 0) return k + 360 * (i - 1) + j;
    return j > 0 ? k + 360 * (q - 1) + j - 1 : -1;
}

// ---- bit-parallel parity check (LDPCDecoder::bad, layered_decoder.hh:65-82) ------------------------------------------
// The sign bits of all N LLRs are packed, 360-bit group by 360-bit group, into 13 dwords per group: bits 0..359 of the
// group, followed by a copy of bits 0..55 (so any 32-bit window starting at bit m < 360 can be read without wrapping).
// The 360 parity checks of a layer are then the XOR of 32-bit windows -- one per table entry, shifted by the entry's
// shift -- plus the two parity-bit windows: 32 check nodes per lane per XOR instead of one. A frame is also bad when
// any LLR is exactly zero (vsign zeroes the product), detected on the packed bytes while building the sign words.
__device__ __forceinline__ uint32_t sign_nibble(uint32_t v)
{
    uint32_t t = (v >> 7) & 0x01010101u;
    t |= t >> 7;
    t |= t >> 14;
    return t & 0xfu;
}
__device__ __forceinline__ uint32_t has_zero_byte(uint32_t v) { return (v - 0x01010101u) & ~v & 0x80808080u; }

__device__ __forceinline__ uint32_t sign_window(const uint32_t *S2, int g, int m)
{
    const uint32_t *w = S2 + g * 13 + (m >> 5);
    return __builtin_amdgcn_alignbit(w[1], w[0], (uint32_t)(m & 31));
}

// one 32-bit sign word (bits 32*kk .. 32*kk+31 of group g) from the LLR bytes; *zero collects "some LLR is exactly 0"
__device__ __forceinline__ uint32_t sign_word(const int8_t *Lm, int g, int kk, uint32_t *zero)
{
    const uint2 *src = reinterpret_cast<const uint2 *>(Lm + g * 360 + 32 * kk);
    const int nb = (kk == 11) ? 1 : 4;          // dword 11 holds only bits 352..359
    uint32_t word = 0;
#pragma unroll
    for (int x = 0; x < 4; ++x)
        if (x < nb) {
            uint2 v = src[x];
            *zero |= has_zero_byte(v.x) | has_zero_byte(v.y);
            word |= (sign_nibble(v.x) | (sign_nibble(v.y) << 4)) << (8 * x);
        }
    return word;
}

// 32 checks of layer i (nodes 32*tj ..) as one XOR of sign windows; nonzero bits = failing checks
__device__ __forceinline__ uint32_t layer_syndrome(const uint32_t *S2, const LdpcLayerDev &ly, const uint32_t *__restrict__ ent, int gp0, int q,
                                                   int i, int tj)
{
    const int j0 = 32 * tj;
    uint32_t syn = S2[(gp0 + i) * 13 + tj];                                   // own parity bits pty[360*i + j]
    if (i > 0) syn ^= S2[(gp0 + i - 1) * 13 + tj];                            // pty[360*(i-1) + j]
    else syn ^= (tj == 0) ? (S2[(gp0 + q - 1) * 13] << 1)                     // pty[360*(q-1) + j - 1], none for j = 0
                          : sign_window(S2, gp0 + q - 1, j0 - 1);
    for (int c = 0; c < ly.cnt; ++c) {
        const uint32_t e = ent[c];
        const int g = (int)__umulhi(e & 0xffffu, 11930465u);                 // base / 360 (base is a multiple of 360)
        int m = j0 - (int)(e >> 16);
        m += (m < 0) ? 360 : 0;
        syn ^= sign_window(S2, g, m);
    }
    return syn & ((tj == 11) ? 0xffu : 0xffffffffu);
}

// returns nonzero (per thread) if this thread saw a zero LLR or a failing check
__device__ __forceinline__ int frame_parity_bad(const int8_t *Lm, uint32_t *S2, const LdpcLayerDev *__restrict__ layers,
                                                const uint32_t *__restrict__ entries, int n, int k, int q, int tid)
{
    const int ngroups = n / 360;
    const int gp0 = k / 360;
    uint32_t zero = 0;
    // Probe: a frame that has not converged almost always fails already among the 360 checks of one layer, and those need the sign
    // words of ~14 of the 180 groups only. Layer 1 is taken (its two parity groups are ordinary ones). If it fails the frame is
    // bad -- the answer the full check would give; only when it passes is the full check run.
    if (q > 1) {
        const LdpcLayerDev ly = layers[1];
        const uint32_t *ent = entries + ly.first_entry;
        const int ng = ly.cnt + 2;
        if (tid < ng * 12) {
            const int gi = tid / 12, kk = tid - gi * 12;
            const int g = gi < ly.cnt ? (int)__umulhi(ent[gi] & 0xffffu, 11930465u) : gp0 + (gi - ly.cnt);
            S2[g * 13 + kk] = sign_word(Lm, g, kk, &zero);
        }
        lds_barrier();
        if (tid < ng) {
            const int g = tid < ly.cnt ? (int)__umulhi(ent[tid] & 0xffffu, 11930465u) : gp0 + (tid - ly.cnt);
            const uint32_t d0 = S2[g * 13], d1 = S2[g * 13 + 1];
            S2[g * 13 + 11] = (S2[g * 13 + 11] & 0xffu) | (d0 << 8);         // two entries may name the same group: idempotent
            S2[g * 13 + 12] = (d0 >> 24) | (d1 << 8);
        }
        lds_barrier();
        uint32_t bad = zero;
        if (tid < 12) bad |= layer_syndrome(S2, ly, ent, gp0, q, 1, tid);
        if (__syncthreads_or(bad != 0)) return 1;
    }
    for (int task = tid; task < ngroups * 12; task += kThreads) {
        const int g = task / 12, kk = task - g * 12;
        S2[g * 13 + kk] = sign_word(Lm, g, kk, &zero);
    }
    lds_barrier();
    for (int g = tid; g < ngroups; g += kThreads) {
        const uint32_t d0 = S2[g * 13], d1 = S2[g * 13 + 1];
        S2[g * 13 + 11] |= d0 << 8;
        S2[g * 13 + 12] = (d0 >> 24) | (d1 << 8);
    }
    lds_barrier();
    uint32_t bad = zero;
    for (int task = tid; task < q * 12; task += kThreads) {
        const int i = task / 12, tj = task - i * 12;
        const LdpcLayerDev ly = layers[i];
        bad |= layer_syndrome(S2, ly, entries + ly.first_entry, gp0, q, i, tj);
    }
    return bad != 0;
}

#if T2_PROF_DETAIL == 3
__shared__ long long s_after_load;
__device__ __forceinline__ void hook_after_load()
{
    // wait for the loads and their arithmetic to be done before reading the clock: the stamp is only as good as the fence
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) s_after_load = (long long)__builtin_readcyclecounter();
}
#endif
#define T2_DTL(kind_, slot_)                                                                                        \
    do {                                                                                                            \
        if (T2_PROF_DETAIL == (kind_) && prof && threadIdx.x == 0) {                                                \
            const long long now_ = (long long)__builtin_readcyclecounter();                                        \
            prof[blockIdx.x * 8 + (slot_)] += now_ - tdt;                                                           \
            tdt = now_;                                                                                             \
        }                                                                                                           \
    } while (0)

#if T2_LDPC_PAIRLANE
template <int CNT, int NCMAX>
__device__ __forceinline__ void layer_update(LdsMem &L, const LayerDesc &d, int j, int h, bool active, int a0, int a1,
                                             CnState &st, uint32_t info, uint32_t *pair_rec, long long *prof)
{
    PlRegs<CNT> r;
    long long tdt = (T2_PROF_DETAIL && prof) ? (long long)__builtin_readcyclecounter() : 0;
    if (active) pl_phase_a<CNT>(L, d, j, h, a0, a1, st, r, pair_rec);
#if T2_PROF_DETAIL == 3
    if (d.kind == T2_LAYER_PLAIN && prof && threadIdx.x == 0) {
        const long long now_ = (long long)__builtin_readcyclecounter();
        prof[blockIdx.x * 8 + 3] += s_after_load - tdt;
        prof[blockIdx.x * 8 + 4] += now_ - s_after_load;
        tdt = now_;
    }
#endif
    if (d.kind == T2_LAYER_PAIR) {
        lds_barrier();
        T2_DTL(1, 3);
        __builtin_amdgcn_s_setprio(3);
#ifndef T2_SEG_WALK_MIN
#define T2_SEG_WALK_MIN 12         // chains at least this long are walked in segments (ldpc_cn2.h)
#endif
        if (d.lmax >= T2_SEG_WALK_MIN) {
            if ((int)threadIdx.x < 360) pl_pair_walk_segments(L, d, (int)threadIdx.x, pair_rec);
        } else if ((int)threadIdx.x < d.step) {
            t2_pair_walk(L, d, (int)threadIdx.x, pair_rec);   // any lane can walk any chain: the first `step` threads do
        }
        __builtin_amdgcn_s_setprio(0);
        lds_barrier();
        T2_DTL(1, 4);
        if (active) pl_pair_finish<CNT>(L, d, j, st, r);
    } else if (d.kind == T2_LAYER_GENERIC) {
        T2_DTL(2, 3);
        __builtin_amdgcn_s_setprio(3);
        for (int lv = 1; lv <= d.lmax; ++lv) {
            if (active) pl_generic_level<CNT, NCMAX>(L, d, lv, info, r);
            lds_barrier();
        }
        __builtin_amdgcn_s_setprio(0);
        T2_DTL(2, 4);
        if (active) pl_generic_finish<CNT>(L, d, st, r);
    }
    lds_barrier();
#if T2_PROF_DETAIL == 3
    if (d.kind == T2_LAYER_PLAIN && prof && threadIdx.x == 0) prof[blockIdx.x * 8 + 5] += (long long)__builtin_readcyclecounter() - tdt;
#endif
    if (d.kind == T2_LAYER_PAIR) T2_DTL(1, 5);
    if (d.kind == T2_LAYER_GENERIC) T2_DTL(2, 5);
}
#else
template <int CNT, int NCMAX>
__device__ __forceinline__ void layer_update(LdsMem &L, const LayerDesc &d, int j, bool active, int a0, int a1,
                                             CnState &st, uint32_t info, uint32_t *pair_rec, long long *prof)
{
    CnRegs<CNT> r;
    long long tdt = (T2_PROF_DETAIL && prof) ? (long long)__builtin_readcyclecounter() : 0;
    if (active) t2_layer_phase_a<CNT>(L, d, j, a0, a1, st, r, pair_rec);
#if T2_PROF_DETAIL == 3
    if (d.kind == T2_LAYER_PLAIN && prof && threadIdx.x == 0) {
        const long long now_ = (long long)__builtin_readcyclecounter();
        prof[blockIdx.x * 8 + 3] += s_after_load - tdt;
        prof[blockIdx.x * 8 + 4] += now_ - s_after_load;
        tdt = now_;
    }
#endif
    // The sequential parts (chain walks, level steps) keep only a few lanes busy and sit on the workgroup's critical
    // path while the co-resident workgroup is usually in a throughput phase: give them issue priority.
    if (d.kind == T2_LAYER_PAIR) {
        lds_barrier();
        T2_DTL(1, 3);
        __builtin_amdgcn_s_setprio(3);
        if (j < d.step) t2_pair_walk(L, d, j, pair_rec);
        __builtin_amdgcn_s_setprio(0);
        lds_barrier();
        T2_DTL(1, 4);
        if (active) t2_pair_finish<CNT>(L, d, j, st, r);
    } else if (d.kind == T2_LAYER_GENERIC) {
        T2_DTL(2, 3);
        __builtin_amdgcn_s_setprio(3);
        for (int lv = 1; lv <= d.lmax; ++lv) {
            if (active) t2_generic_level<CNT, NCMAX>(L, d, lv, info, r);
            lds_barrier();
        }
        __builtin_amdgcn_s_setprio(0);
        T2_DTL(2, 4);
        if (active) t2_generic_finish<CNT>(L, d, st, r);
    }
    lds_barrier();
#if T2_PROF_DETAIL == 3
    if (d.kind == T2_LAYER_PLAIN && prof && threadIdx.x == 0) prof[blockIdx.x * 8 + 5] += (long long)__builtin_readcyclecounter() - tdt;
#endif
    if (d.kind == T2_LAYER_PAIR) T2_DTL(1, 5);
    if (d.kind == T2_LAYER_GENERIC) T2_DTL(2, 5);
}
#endif

#define T2_PROF_T(var) long long var = p.prof ? (long long)__builtin_readcyclecounter() : 0
#define T2_PROF_ADD(slot, t0)                                                                                   \
    do {                                                                                                        \
        if (p.prof && threadIdx.x == 0) p.prof[blockIdx.x * 8 + (slot)] += (long long)__builtin_readcyclecounter() - (t0); \
    } while (0)

template <int LO, int HI, int NCMAX = T2_LDPC_NC_MAX>
#ifndef T2_LDPC_MIN_WAVES
#define T2_LDPC_MIN_WAVES (T2_LDPC_PAIRLANE ? 6 : 4)   // waves per SIMD the register allocation leaves room for (two workgroups per CU): 4 -> 128 VGPRs, 6 -> 80
#endif
__global__ __launch_bounds__(kThreads, T2_LDPC_MIN_WAVES) void ldpc_decode_kernel(const LdpcLayerDev *__restrict__ layers, const uint32_t *__restrict__ entries,
                                                                  const uint32_t *__restrict__ cninfo, const uint32_t *__restrict__ entries2,
                                                                  LdpcKernelParams p)
{
    extern __shared__ __attribute__((aligned(16))) int8_t lds[];
    int8_t *Lm = lds;
    int *s_ctl = reinterpret_cast<int *>(lds + p.lds_ctl_offset);
    uint32_t *pair_rec = reinterpret_cast<uint32_t *>(lds + p.lds_rec_offset);
    uint32_t *S2 = reinterpret_cast<uint32_t *>(lds + p.lds_sign_offset);
#if T2_LDPC_PAIRLANE
    uint32_t *lds_ent = reinterpret_cast<uint32_t *>(lds + p.lds_ent_offset);
    for (int x = threadIdx.x; x < 2 * p.n_entries; x += kThreads) lds_ent[x] = entries2[x];      // made visible by the first barrier below
#endif
    LdsMem L{(uint32_t)(uintptr_t)(lds_i8 *)Lm};     // generic -> LDS address space cast: the 32-bit LDS offset of the array

    const int tid = threadIdx.x;
#if T2_LDPC_PAIRLANE
    const int j = tid >> 1, h = tid & 1;             // node, and which half of its link slots this lane takes
#else
    const int j = tid;
#endif
    const bool active = j < 360;
    if (L.off() != p.lds_base) {                     // the split table was built for another LDS layout: refuse, loudly
        if (tid == 0) *p.error = 2;
        return;
    }
    const int group = p.group;
    const int slot = blockIdx.x / group, member = blockIdx.x % group;
    const int nslots = gridDim.x / group;
    const int nbatches = (p.n_frames + group - 1) / group;
    uint2 *state = p.state + (size_t)blockIdx.x * p.q * 360;

    if (p.prof && tid == 0) p.prof[blockIdx.x * 8 + 6] = wall_clock64();
    if (tid == 0) __hip_atomic_fetch_add(p.resident, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // "this workgroup is placed"
    for (int batch = slot; batch < nbatches; batch += nslots) {
        const int frame = batch * group + member;
        const bool have = frame < p.n_frames;
        int members = p.n_frames - batch * group;
        members = members > group ? group : members;

        if (have) {
            const uint2 *src = reinterpret_cast<const uint2 *>(p.llr + (size_t)frame * p.n);
            uint2 *dst = reinterpret_cast<uint2 *>(Lm);
            for (int x = tid; x < p.n / 8; x += kThreads) dst[x] = src[x];
            for (int x = tid; x < p.q * 360; x += kThreads) state[x] = make_uint2(0u, 0u);
        }
        __syncthreads();

        int trials = p.max_trials;
        int result;
        for (int t = 0;; ++t) {
            // ---- parity check of the whole frame (LDPCDecoder::bad)
            T2_PROF_T(tp0);
            int bad = have ? frame_parity_bad(Lm, S2, layers, entries, p.n, p.k, p.q, tid) : 0;
            int frame_bad = __syncthreads_or(bad);
            int all_ok = !frame_bad;
            T2_PROF_ADD(0, tp0);
            T2_PROF_T(tp1);
            if (group > 1 && have) {
                // one word per (batch, trial): high half counts arrivals, low half counts parity-clean frames
                if (tid == 0) {
                    unsigned *w = p.sync + (size_t)batch * (p.max_trials + 1) + t;
                    atomicAdd(w, 0x10000u | (all_ok ? 1u : 0u));
                    unsigned v;
                    long long t0 = wall_clock64();
                    int verdict = -1;
                    // A frame that still fails its parity check knows the verdict without waiting: the batch goes on. Only
                    // parity-clean frames have to learn whether all the others are clean too.
                    if (!all_ok) verdict = 0;
                    else for (;;) {
                        v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((int)(v >> 16) >= members) { verdict = ((int)(v & 0xffffu) == members); break; }
                        if (__hip_atomic_load(p.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                        if (wall_clock64() - t0 > p.spin_timeout_ticks) {
                            __hip_atomic_store(p.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(8);
                    }
                    s_ctl[0] = verdict;
                }
                __syncthreads();
                int verdict = s_ctl[0];
                __syncthreads();
                if (verdict < 0) { result = -3; break; }   // sync failure: reported through p.error
                all_ok = verdict;
            }
            T2_PROF_ADD(1, tp1);
            if (all_ok) { result = trials; break; }
            if (--trials < 0) { result = -1; break; }

            // ---- one layered update sweep (LDPCDecoder::update)
            if (have) {
                // records belong to (layer, thread); in GENERIC layers thread j takes node cninfo >> 20 (nodes sorted by dependency
                // level, ldpc_graph.cpp), elsewhere node j
                uint2 nxt = active ? state[j] : make_uint2(0u, 0u);
                uint32_t info_nxt = (active && layers[0].kind == T2_LAYER_GENERIC) ? cninfo[j] : 0u;
                for (int i = 0; i < p.q; ++i) {
                    const LdpcLayerDev ly = layers[i];
                    LayerDesc d{entries + ly.first_entry, ly.cnt, ly.lmax, ly.nc, ly.kind, ly.step, L.off() + p.lds_ctl_offset + 32, entries[ly.first_entry],
                                entries2 + 2 * ly.first_entry, L.off() + p.lds_ent_offset + 8 * ly.first_entry};
                    const uint32_t info = info_nxt;
                    const int jn = ly.kind == T2_LAYER_GENERIC ? (int)(info >> 20) : j;
                    const int a0 = L.off() + p.k + 360 * i + jn, a1r = parity_prev_addr(p.k, p.q, i, jn);
                    const int a1 = a1r >= 0 ? a1r + L.off() : -1;
                    CnState st{nxt.x, nxt.y};
                    if (active && i + 1 < p.q) {                                     // prefetch the next layer's record (and node)
                        nxt = state[(i + 1) * 360 + j];
                        info_nxt = layers[i + 1].kind == T2_LAYER_GENERIC ? cninfo[(i + 1) * 360 + j] : 0u;
                    }
                    T2_PROF_T(tp2);
#if T2_LDPC_PAIRLANE
                    T2_LDPC_DISPATCH_RANGE(ly.cnt, LO, HI, (layer_update<CNT, NCMAX>(L, d, jn, h, active, a0, a1, st, info, pair_rec, p.prof)));
#else
                    T2_LDPC_DISPATCH_RANGE(ly.cnt, LO, HI, (layer_update<CNT, NCMAX>(L, d, jn, active, a0, a1, st, info, pair_rec, p.prof)));
#endif
                    if (!(T2_PROF_DETAIL && (ly.kind == T2_PROF_DETAIL || T2_PROF_DETAIL == 3))) T2_PROF_ADD(2 + ly.kind, tp2);   // mode 3: slots 3-5 are the PLAIN split only
#if T2_LDPC_PAIRLANE
                    if (active && h == 0) state[i * 360 + j] = make_uint2(st.w0, st.w1);
#else
                    if (active) state[i * 360 + j] = make_uint2(st.w0, st.w1);
#endif
                }
            }
            __syncthreads();   // once per sweep: the records written above are re-read by the same thread next sweep
        }

        // ---- outputs: hard decision of the information bits (ldpc_decoder.cpp:270-277), one bit per byte
        if (have) {
            if (p.bits) {
                uint8_t *o = p.bits + (size_t)frame * p.k;
                for (int x = tid; x < p.k / 4; x += kThreads) {
                    uint32_t v = reinterpret_cast<const uint32_t *>(Lm)[x];
                    reinterpret_cast<uint32_t *>(o)[x] = (v >> 7) & 0x01010101u;
                }
            }
            if (p.llr_out) {
                uint2 *o = reinterpret_cast<uint2 *>(p.llr_out + (size_t)frame * p.n);
                for (int x = tid; x < p.n / 8; x += kThreads) o[x] = reinterpret_cast<const uint2 *>(Lm)[x];
            }
            if (tid == 0 && member == 0) p.trials_left[batch] = result;
        }
        __syncthreads();
    }
    if (p.prof && tid == 0) p.prof[blockIdx.x * 8 + 7] = wall_clock64();
}

// Kernel variants by range of information-bit links per check node: the twelve T2 codes fall into four families.
typedef void (*ldpc_kernel_fn)(const LdpcLayerDev *, const uint32_t *, const uint32_t *, const uint32_t *, LdpcKernelParams);
static ldpc_kernel_fn pick_kernel(int min_cnt, int max_cnt)
{
    if (min_cnt == max_cnt) {      // the six normal-frame codes have one link count each; third argument = the most conflict
        switch (max_cnt) {         // slots any of that code's layers has (ldpc_graph.cpp; checked by tests/test_capi_symbols.py)
        case 5: return ldpc_decode_kernel<5, 5, 2>;      // N 1/2: no GENERIC layer
        case 8: return ldpc_decode_kernel<8, 8, 4>;      // N 2/3
        case 9: return ldpc_decode_kernel<9, 9, 4>;      // N 3/5
        case 12: return ldpc_decode_kernel<12, 12, 4>;   // N 3/4
        case 16: return ldpc_decode_kernel<16, 16, 7>;   // N 4/5
        case 20: return ldpc_decode_kernel<20, 20, 6>;   // N 5/6
        default: break;
        }
    }
    if (max_cnt <= 8) return ldpc_decode_kernel<1, 8>;
    if (min_cnt >= 7 && max_cnt <= 12) return ldpc_decode_kernel<7, 12>;
    if (min_cnt >= 13 && max_cnt <= 17) return ldpc_decode_kernel<13, 17>;
    if (min_cnt >= 16 && max_cnt <= 20) return ldpc_decode_kernel<16, 20>;
    return ldpc_decode_kernel<1, 20>;
}

hipError_t ldpc_kernel_attributes(int min_cnt, int max_cnt, int lds_bytes, int *blocks_per_cu, int *static_lds_bytes)
{
    ldpc_kernel_fn fn = pick_kernel(min_cnt, max_cnt);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return e;
    hipFuncAttributes attr;
    if ((e = hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(fn))) != hipSuccess) return e;
    *static_lds_bytes = (int)attr.sharedSizeBytes;        // the dynamic array (the LLRs) starts behind the static LDS
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, fn, kThreads, lds_bytes);
}

hipError_t ldpc_kernel_launch(int min_cnt, int max_cnt, const LdpcKernelParams &p, int grid, int lds_bytes, hipStream_t stream)
{
    ldpc_kernel_fn fn = pick_kernel(min_cnt, max_cnt);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(kThreads), lds_bytes, stream, p.layers, p.entries, p.cninfo, p.entries2, p);
    return hipGetLastError();
}

}  // namespace t2gpu
