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
//     (ldpc_cn.h) that streams through L2, coalesced;
//   * layers whose nodes share bits are executed level by level (ldpc_graph.h) so every node sees exactly the LLRs
//     the reference's sequential j loop would show it: results are LLR-exact, not merely codeword-exact;
//   * the reference decodes SIMD batches of 32 frames that stop together (all 32 parity-clean, or 25 updates). The
//     32 workgroups of such a batch are co-resident (persistent grid) and agree on every stop decision through one
//     atomic word per (batch, trial).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ldpc_cn.h"
#include "ldpc_kernel.h"

namespace t2gpu {

struct LdsMem {
    int8_t *p;
    __device__ __forceinline__ int8_t ld(int a) const { return p[a]; }
    __device__ __forceinline__ void st(int a, int8_t v) { p[a] = v; }
};

static constexpr int kThreads = T2GPU_LDPC_THREADS;   // 6 wavefronts; 360 of 384 lanes own a check node

__device__ __forceinline__ int parity_prev_addr(int k, int q, int i, int j)
{
    if (i > 0) return k + 360 * (i - 1) + j;
    return j > 0 ? k + 360 * (q - 1) + j - 1 : -1;
}

__global__ __launch_bounds__(kThreads) void ldpc_decode_kernel(LdpcKernelParams p)
{
    extern __shared__ __attribute__((aligned(16))) int8_t lds[];
    int8_t *Lm = lds;
    int *s_ctl = reinterpret_cast<int *>(lds + p.lds_ctl_offset);
    LdsMem L{Lm};

    const int tid = threadIdx.x;
    const int j = tid;
    const bool active = j < 360;
    const int group = p.group;
    const int slot = blockIdx.x / group, member = blockIdx.x % group;
    const int nslots = gridDim.x / group;
    const int nbatches = (p.n_frames + group - 1) / group;
    uint2 *state = p.state + (size_t)blockIdx.x * p.q * 360;

    for (int batch = slot; batch < nbatches; batch += nslots) {
        const int frame = batch * group + member;
        const bool have = frame < p.n_frames;
        int members = p.n_frames - batch * group;
        members = members > group ? group : members;

        if (have) {
            const uint2 *src = reinterpret_cast<const uint2 *>(p.llr + (size_t)frame * p.n);
            uint2 *dst = reinterpret_cast<uint2 *>(Lm);
            for (int x = tid; x < p.n / 8; x += kThreads) dst[x] = src[x];
            if (active)
                for (int i = 0; i < p.q; ++i) state[i * 360 + j] = make_uint2(0u, 0u);
        }
        __syncthreads();

        int trials = p.max_trials;
        int result;
        for (int t = 0;; ++t) {
            // ---- parity check of the whole frame (LDPCDecoder::bad)
            int bad = 0;
            if (have && active) {
                for (int i = 0; i < p.q && !bad; ++i) {
                    const LdpcLayerDev ly = p.layers[i];
                    const uint32_t *ent = p.entries + ly.first_entry;
                    const int a0 = p.k + 360 * i + j, a1 = parity_prev_addr(p.k, p.q, i, j);
                    T2_LDPC_DISPATCH_CNT(ly.cnt, bad = t2_cn_bad<CNT>(L, ent, j, a0, a1));
                }
            }
            int frame_bad = __syncthreads_or(bad);
            int all_ok = !frame_bad;
            if (group > 1 && have) {
                // one word per (batch, trial): high half counts arrivals, low half counts parity-clean frames
                if (tid == 0) {
                    unsigned *w = p.sync + (size_t)batch * (p.max_trials + 1) + t;
                    atomicAdd(w, 0x10000u | (all_ok ? 1u : 0u));
                    unsigned v;
                    long long t0 = wall_clock64();
                    int verdict = -1;
                    for (;;) {
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
            if (all_ok) { result = trials; break; }
            if (--trials < 0) { result = -1; break; }

            // ---- one layered update sweep (LDPCDecoder::update)
            if (have) {
                for (int i = 0; i < p.q; ++i) {
                    const LdpcLayerDev ly = p.layers[i];
                    const uint32_t *ent = p.entries + ly.first_entry;
                    const int a0 = p.k + 360 * i + j, a1 = parity_prev_addr(p.k, p.q, i, j);
                    const int lvl = active ? (ly.lmax > 1 ? (int)p.levels[i * 360 + j] : 1) : 0;
                    for (int lv = 1; lv <= ly.lmax; ++lv) {
                        if (lvl == lv) {
                            uint2 raw = state[i * 360 + j];
                            CnState st{raw.x, raw.y};
                            T2_LDPC_DISPATCH_CNT(ly.cnt, t2_cn_update<CNT>(L, ent, j, a0, a1, st));
                            state[i * 360 + j] = make_uint2(st.w0, st.w1);
                        }
                        __syncthreads();
                    }
                }
            }
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
}

hipError_t ldpc_kernel_attributes(int lds_bytes, int *blocks_per_cu)
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ldpc_decode_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return e;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, ldpc_decode_kernel, kThreads, lds_bytes);
}

hipError_t ldpc_kernel_launch(const LdpcKernelParams &p, int grid, int lds_bytes, hipStream_t stream)
{
    hipLaunchKernelGGL(ldpc_decode_kernel, dim3(grid), dim3(kThreads), lds_bytes, stream, p);
    return hipGetLastError();
}

}  // namespace t2gpu
