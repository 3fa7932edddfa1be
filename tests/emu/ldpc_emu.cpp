// tests/emu/ldpc_emu.cpp -- TEST HELPER (not part of the product library, never shipped in libt2gpu.so).
//
// Replays on the host exactly the per-layer schedule the HIP kernel (csrc/ldpc_kernel.hip) executes: same graph
// (csrc/ldpc_graph.cpp), same phase functions and state compression (csrc/ldpc_cn.h). Everything between two workgroup
// barriers of the kernel is an "epoch": inside an epoch the 360 node-threads are run to completion one after the
// other in a caller-chosen order (ascending, descending or a seeded shuffle). If the result is identical to the oracle
// for every order, the schedule neither depends on intra-epoch timing (= no data race the GPU could expose) nor
// deviates from the reference's sequential order. This lets the CPU-only test tier validate the kernel's logic.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <random>
#include <map>
#include <vector>
#include "../../sdr_receiver_dvb_t2_amd/csrc/ldpc_cn.h"
#include "../../sdr_receiver_dvb_t2_amd/csrc/ldpc_graph.h"

using namespace t2gpu;

namespace {
struct Mem {
    int8_t *p;
    int off() const { return 0; }
    int8_t ld(int a) const { return p[a]; }
    void st(int a, int8_t v) { p[a] = v; }
};
int prev_addr(int k, int q, int i, int j) { return i > 0 ? k + 360 * (i - 1) + j : (j > 0 ? k + 360 * (q - 1) + j - 1 : -1); }

struct Order {
    int mode; std::mt19937 rng;
    std::vector<int> make()
    {
        std::vector<int> o(360);
        std::iota(o.begin(), o.end(), 0);
        if (mode == 1) std::reverse(o.begin(), o.end());
        else if (mode >= 2) std::shuffle(o.begin(), o.end(), rng);
        return o;
    }
};

template <int CNT>
void emu_layer(Mem &L, const LdpcGraph &g, int i, CnState *st, Order &ord)
{
    const LdpcLayer &ly = g.layers[i];
    LayerDesc d{&g.entries[ly.first_entry], ly.cnt, ly.lmax, ly.n_conflict, ly.kind, ly.step, g.n, g.entries[ly.first_entry]};   // scratch byte behind the LLRs
    std::vector<CnRegs<CNT>> regs(360);
    std::vector<uint32_t> rec(360, 0xdeadbeefu);
    auto a0 = [&](int j) { return g.k + 360 * i + j; };
    auto a1 = [&](int j) { return prev_addr(g.k, g.q, i, j); };
    if (d.kind == T2_LAYER_GENERIC) {
        // cninfo entry t describes the node thread t takes (bits 20..28); index it by node here
        std::vector<uint32_t> info(360, 0);
        for (int t = 0; t < 360; ++t) { const uint32_t v = g.cninfo[(size_t)i * 360 + t]; info[v >> 20] = v; }
        for (int j : ord.make()) {       // epoch 1: phase A immediately followed by level step 1 (no barrier between)
            t2_layer_phase_a<CNT>(L, d, j, a0(j), a1(j), st[j], regs[j], rec.data());
            t2_generic_level<CNT>(L, d, 1, info[j], regs[j]);
        }
        for (int lv = 2; lv <= d.lmax; ++lv)
            for (int j : ord.make()) t2_generic_level<CNT>(L, d, lv, info[j], regs[j]);
        for (int j : ord.make()) t2_generic_finish<CNT>(L, d, st[j], regs[j]);
        return;
    }
    for (int j : ord.make()) t2_layer_phase_a<CNT>(L, d, j, a0(j), a1(j), st[j], regs[j], rec.data());
    if (d.kind == T2_LAYER_PAIR) {
        for (int j : ord.make())
            if (j < d.step) t2_pair_walk(L, d, j, rec.data());
        for (int j : ord.make()) t2_pair_finish<CNT>(L, d, j, st[j], regs[j]);
    }
}
}  // namespace

// order_mode: 0 ascending thread order inside every epoch, 1 descending, >= 2 seeded shuffles
extern "C" int emu_ldpc_decode(int code_id, const int8_t *llr_in, int blocks, int max_trials, uint8_t *bits_out,
                               int8_t *llr_out, int order_mode)
{
    LdpcGraph g;
    if (!ldpc_build_graph(code_id, g)) return -2;
    std::vector<std::vector<int8_t>> Lv(blocks, std::vector<int8_t>(g.n + 1));
    std::vector<std::vector<CnState>> S(blocks, std::vector<CnState>((size_t)g.q * 360, CnState{0, 0}));
    for (int b = 0; b < blocks; ++b) memcpy(Lv[b].data(), llr_in + (size_t)b * g.n, g.n);
    Order ord{order_mode, std::mt19937(1234u + order_mode)};
    int trials = max_trials;
    for (;;) {
        bool bad = false;
        for (int b = 0; b < blocks && !bad; ++b) {
            Mem M{Lv[b].data()};
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
            Mem M{Lv[b].data()};
            for (int i = 0; i < g.q; ++i) {
                CnState *st = &S[b][(size_t)i * 360];
                T2_LDPC_DISPATCH_CNT(g.layers[i].cnt, emu_layer<CNT>(M, g, i, st, ord));
            }
        }
    }
    for (int b = 0; b < blocks; ++b) {
        if (llr_out) memcpy(llr_out + (size_t)b * g.n, Lv[b].data(), g.n);
        if (bits_out)
            for (int i = 0; i < g.k; ++i) bits_out[(size_t)b * g.k + i] = Lv[b][i] < 0;
    }
    return trials;
}

extern "C" int emu_ldpc_graph_summary(int code_id, int *n_plain, int *n_pair, int *n_generic, int *serial_steps)
{
    LdpcGraph g;
    if (!ldpc_build_graph(code_id, g)) return -1;
    int c[3] = {0, 0, 0};
    for (const LdpcLayer &l : g.layers) c[l.kind]++;
    *n_plain = c[0]; *n_pair = c[1]; *n_generic = c[2]; *serial_steps = g.serial_steps;
    return 0;
}

// largest number of conflict slots of any GENERIC layer of the code (0 when it has none)
extern "C" int emu_ldpc_max_conflict(int code_id)
{
    LdpcGraph g;
    if (!ldpc_build_graph(code_id, g)) return -1;
    int m = 0;
    for (const LdpcLayer &l : g.layers) if (l.kind == T2_LAYER_GENERIC) m = std::max(m, l.n_conflict);
    return m;
}

// Independent check of the band-walk description of GENERIC layers (ldpc_graph.h: band, band_prefetch; the two-frame kernel's
// p2_band_walk relies on it). For every band layer, by brute force over the 360 nodes and their conflict entries:
//   1: two nodes that share a bit never lie in the same band, and the earlier node lies in the earlier band (ascending-j order
//      band by band is then the reference's order);
//   2: the bit node j reaches through entry 0 is the bit node j + D reaches through entry 1, and no node between them touches it
//      (the value handed down in a register is the value the reference would read);
//   3: with band_prefetch, whatever else a node of band t reads was last written in band t - 2 or earlier (loads one band ahead).
// Returns 0, or 100 * layer + rule on the first violation; *n_band = number of band layers of the code.
extern "C" int emu_ldpc_band_check(int code_id, int *n_band)
{
    LdpcGraph g;
    if (!ldpc_build_graph(code_id, g)) return -1;
    *n_band = 0;
    for (int i = 0; i < g.q; ++i) {
        const LdpcLayer &L = g.layers[i];
        if (!L.band) continue;
        ++*n_band;
        if (L.kind != T2_LAYER_GENERIC || L.n_conflict > 4 || L.band > 32) return 100 * i + 9;
        const int D = L.band, nc = L.n_conflict;
        std::vector<int> grp(nc), sh(nc);
        for (int c = 0; c < nc; ++c) { const uint32_t e = g.entries[L.first_entry + c]; grp[c] = (int)(e & 0xffffu) / 360; sh[c] = (int)(e >> 16); }
        auto bit = [&](int j, int c) { return grp[c] * 360 + ((j - sh[c]) % 360 + 360) % 360; };
        // touchers of every bit in node order
        std::map<int, std::vector<std::pair<int, int>>> touch;       // bit -> (node, entry), ascending node
        for (int j = 0; j < 360; ++j)
            for (int c = 0; c < nc; ++c) touch[bit(j, c)].push_back({j, c});
        for (auto &kv : touch) {
            auto &v = kv.second;
            std::sort(v.begin(), v.end());
            for (size_t a = 0; a + 1 < v.size(); ++a)
                if (v[a].first / D >= v[a + 1].first / D) return 100 * i + 1;
        }
        for (int j = 0; j + D < 360; ++j) {
            if (bit(j, 0) != bit(j + D, 1)) return 100 * i + 2;
            const auto &v = touch[bit(j, 0)];
            for (size_t a = 0; a + 1 < v.size(); ++a)
                if (v[a].first == j && v[a + 1].first != j + D) return 100 * i + 2;
        }
        if (L.band_prefetch)
            for (int j = D; j < 360; ++j)
                for (int c = 0; c < nc; ++c) {
                    if (c == 1) continue;
                    const auto &v = touch[bit(j, c)];
                    for (size_t a = 1; a < v.size(); ++a)
                        if (v[a].first == j && v[a].second == c && v[a - 1].first / D > j / D - 2) return 100 * i + 3;
                }
    }
    return 0;
}

// Independent check of LdpcLayer::no_close (closing barriers the two-frame kernel leaves out): walking the layers, every layer that
// starts while earlier ones are still open (no barrier since) must be PLAIN or PAIR like them, share no information-bit group with
// any of them, and not be a second PAIR layer among them; a PAIR layer's inner barrier closes all layers before it. Returns 0 or
// 100 * layer + rule; *n_open = number of barriers left out.
extern "C" int emu_ldpc_open_check(int code_id, int *n_open)
{
    LdpcGraph g;
    if (!ldpc_build_graph(code_id, g)) return -1;
    *n_open = 0;
    std::vector<int> open;                                      // layers not yet closed by a barrier
    for (int i = 0; i < g.q; ++i) {
        const LdpcLayer &L = g.layers[i];
        const bool simple = L.kind == T2_LAYER_PLAIN || L.kind == T2_LAYER_PAIR;
        if (!open.empty()) {
            if (!simple) return 100 * i + 1;
            for (int o : open) {
                const LdpcLayer &O = g.layers[o];
                if (!(O.kind == T2_LAYER_PLAIN || O.kind == T2_LAYER_PAIR)) return 100 * i + 1;
                if (O.kind == T2_LAYER_PAIR && L.kind == T2_LAYER_PAIR) return 100 * i + 2;
                for (int a = 0; a < L.cnt; ++a)
                    for (int b = 0; b < O.cnt; ++b)
                        if ((g.entries[L.first_entry + a] & 0xffffu) / 360 == (g.entries[O.first_entry + b] & 0xffffu) / 360) return 100 * i + 3;
            }
        }
        if (L.kind == T2_LAYER_PAIR) open.clear();               // its inner barrier
        if (i + 1 == g.q && L.no_close) return 100 * i + 4;      // the last layer always closes
        if (L.no_close) { open.push_back(i); ++*n_open; }
        else open.clear();
    }
    return 0;
}
