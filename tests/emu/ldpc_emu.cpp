// tests/emu/ldpc_emu.cpp -- TEST HELPER (not part of the product library, never shipped in libt2gpu.so).
//
// Replays, sequentially on the host, exactly the schedule the HIP kernel (csrc/ldpc_kernel.hip) executes: same graph
// (csrc/ldpc_graph.cpp), same check-node arithmetic and state compression (csrc/ldpc_cn.h), same level order. It lets
// the CPU-only test tier compare "what the kernel is going to compute" with the oracle before a GPU is involved, and
// it flags any two nodes of one level that touch the same LLR byte (which on the GPU would be a data race).
#include <cstdint>
#include <cstring>
#include <vector>
#include "../../sdr_receiver_dvb_t2_amd/csrc/ldpc_cn.h"
#include "../../sdr_receiver_dvb_t2_amd/csrc/ldpc_graph.h"

using namespace t2gpu;

namespace {
struct TrackMem {
    int8_t *p;
    int *owner;      // last node that touched each byte in the current level step (-1 = none)
    int node;
    int *races;
    int8_t ld(int a) const { touch(a); return p[a]; }
    void st(int a, int8_t v) { touch(a); p[a] = v; }
    void touch(int a) const
    {
        if (!owner) return;
        if (owner[a] >= 0 && owner[a] != node) ++*races;
        owner[a] = node;
    }
};
int prev_addr(int k, int q, int i, int j) { return i > 0 ? k + 360 * (i - 1) + j : (j > 0 ? k + 360 * (q - 1) + j - 1 : -1); }
}

extern "C" int emu_ldpc_decode(int code_id, const int8_t *llr_in, int blocks, int max_trials, uint8_t *bits_out,
                               int8_t *llr_out, int *races_out)
{
    LdpcGraph g;
    if (!ldpc_build_graph(code_id, g)) return -2;
    std::vector<std::vector<int8_t>> L(blocks, std::vector<int8_t>(g.n));
    std::vector<std::vector<CnState>> S(blocks, std::vector<CnState>((size_t)g.q * 360, CnState{0, 0}));
    for (int b = 0; b < blocks; ++b) memcpy(L[b].data(), llr_in + (size_t)b * g.n, g.n);
    std::vector<int> owner(g.n);
    int races = 0;
    int trials = max_trials;
    for (;;) {
        bool bad = false;
        for (int b = 0; b < blocks && !bad; ++b) {
            TrackMem M{L[b].data(), nullptr, 0, &races};
            for (int i = 0; i < g.q && !bad; ++i) {
                const LdpcLayer &ly = g.layers[i];
                const uint32_t *ent = &g.entries[ly.first_entry];
                for (int j = 0; j < 360 && !bad; ++j) {
                    int a0 = g.k + 360 * i + j, a1 = prev_addr(g.k, g.q, i, j);
                    T2_LDPC_DISPATCH_CNT(ly.cnt, bad = t2_cn_bad<CNT>(M, ent, j, a0, a1));
                }
            }
        }
        if (!bad) break;
        if (--trials < 0) break;
        for (int b = 0; b < blocks; ++b) {
            for (int i = 0; i < g.q; ++i) {
                const LdpcLayer &ly = g.layers[i];
                const uint32_t *ent = &g.entries[ly.first_entry];
                for (int lv = 1; lv <= ly.lmax; ++lv) {
                    std::fill(owner.begin(), owner.end(), -1);
                    // descending j inside a level: if the level assignment were wrong, this order would expose it
                    for (int j = 359; j >= 0; --j) {
                        if (g.levels[(size_t)i * 360 + j] != lv) continue;
                        TrackMem M{L[b].data(), b == 0 ? owner.data() : nullptr, j, &races};
                        int a0 = g.k + 360 * i + j, a1 = prev_addr(g.k, g.q, i, j);
                        CnState &st = S[b][(size_t)i * 360 + j];
                        T2_LDPC_DISPATCH_CNT(ly.cnt, t2_cn_update<CNT>(M, ent, j, a0, a1, st));
                    }
                }
            }
        }
    }
    for (int b = 0; b < blocks; ++b) {
        if (llr_out) memcpy(llr_out + (size_t)b * g.n, L[b].data(), g.n);
        if (bits_out)
            for (int i = 0; i < g.k; ++i) bits_out[(size_t)b * g.k + i] = L[b][i] < 0;
    }
    if (races_out) *races_out = races;
    return trials;
}
