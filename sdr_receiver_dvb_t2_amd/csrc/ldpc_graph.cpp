// ldpc_graph.cpp -- see ldpc_graph.h. Pure host code (no HIP), also compiled into the schedule emulator of tests/.
#include "ldpc_graph.h"
#include "ldpc_cn.h"
#include "tables/ldpc_tables_data.h"
#include <algorithm>
#include <map>

namespace t2gpu {

static const int GROUP = 360;

bool ldpc_build_graph(int code_id, LdpcGraph &g)
{
    if (code_id < 0 || code_id >= T2_LDPC_NUM_CODES) return false;
    const t2_ldpc_table_t &t = T2_LDPC_TABLES[code_id];
    g = LdpcGraph();
    g.id = code_id; g.n = t.n; g.k = t.k; g.r = t.n - t.k; g.q = g.r / GROUP;

    struct Ent { int group, shift; };
    std::vector<std::vector<Ent>> per_layer(g.q);
    const uint16_t *a = t.addr;
    for (int grp = 0; grp < t.n_groups; ++grp) {
        for (int e = 0; e < t.group_deg[grp]; ++e) {
            int x = a[e];
            per_layer[x % g.q].push_back({grp, x / g.q});
        }
        a += t.group_deg[grp];
    }

    g.layers.resize(g.q);
    g.levels.assign((size_t)g.q * GROUP, 1);
    g.cninfo.assign((size_t)g.q * GROUP, 1);
    for (int i = 0; i < g.q; ++i) {
        std::vector<Ent> &ents = per_layer[i];
        std::map<int, int> mult;
        for (const Ent &e : ents) mult[e.group]++;
        // conflict entries first (stable), so the kernel finds them in fixed link slots
        std::stable_sort(ents.begin(), ents.end(), [&](const Ent &x, const Ent &y) {
            bool cx = mult[x.group] > 1, cy = mult[y.group] > 1;
            if (cx != cy) return cx;
            if (x.group != y.group) return x.group < y.group;
            return x.shift < y.shift;
        });
        LdpcLayer &L = g.layers[i];
        L.first_entry = (int)g.entries.size();
        L.cnt = (int)ents.size();
        L.n_conflict = 0;
        for (const Ent &e : ents)
            if (mult[e.group] > 1) L.n_conflict++;
        L.kind = L.n_conflict == 0 ? T2_LAYER_PLAIN : (L.n_conflict == 2 ? T2_LAYER_PAIR : T2_LAYER_GENERIC);
        L.step = 0;
        if (L.kind == T2_LAYER_PAIR) {
            // slot 0 = the shift whose bit is shared with the LATER node j+step, slot 1 = with the earlier j-step
            int d = ents[1].shift - ents[0].shift;          // sorted: > 0
            if (d > 180) { std::swap(ents[0], ents[1]); d = 360 - d; }
            L.step = d;
        }
        if (L.kind == T2_LAYER_GENERIC && L.n_conflict <= 4) {
            // band walk: D = the smallest (shift_y - shift_x) mod 360 over ordered pairs of conflict entries of one group
            int D = GROUP, bx = -1, by = -1, at_d = 0;
            for (int x = 0; x < L.n_conflict; ++x)
                for (int y = 0; y < L.n_conflict; ++y) {
                    if (x == y || ents[x].group != ents[y].group) continue;
                    const int d = ((ents[y].shift - ents[x].shift) % GROUP + GROUP) % GROUP;
                    if (d < D) { D = d; bx = x; by = y; at_d = 1; }
                    else if (d == D) ++at_d;
                }
            bool far = at_d == 1;
            for (int x = 0; x < L.n_conflict; ++x)
                for (int y = 0; y < L.n_conflict; ++y) {
                    if (x == y || ents[x].group != ents[y].group || (x == bx && y == by)) continue;
                    const int d = ((ents[y].shift - ents[x].shift) % GROUP + GROUP) % GROUP;
                    if (d < 2 * D) far = false;
                }
            if (D >= 2 && D <= 32 && bx >= 0) {      // two lanes per node at least: D x 2 lanes of one wavefront
                std::vector<Ent> re;
                re.push_back(ents[bx]); re.push_back(ents[by]);
                for (int x = 0; x < L.n_conflict; ++x)
                    if (x != bx && x != by) re.push_back(ents[x]);
                for (int x = 0; x < L.n_conflict; ++x) ents[x] = re[x];
                L.band = D; L.band_prefetch = far ? 1 : 0;
            }
        }
        for (const Ent &e : ents) g.entries.push_back((uint32_t)(e.group * GROUP) | ((uint32_t)e.shift << 16));
        g.max_cnt = std::max(g.max_cnt, L.cnt);
        g.min_cnt = std::min(g.min_cnt, L.cnt);
        // dependency levels: node j depends on every earlier node sharing one of its bits
        uint8_t *lev = &g.levels[(size_t)i * GROUP];
        int lmax = 1;
        if (L.n_conflict) {
            for (int j = 0; j < GROUP; ++j) {
                int lv = 1;
                uint32_t dep = 0;
                for (int x = 0; x < L.n_conflict; ++x)
                    for (int y = 0; y < L.n_conflict; ++y) {
                        if (x == y || ents[x].group != ents[y].group) continue;
                        // node j reaches bit m through entry x; the same bit reaches node j2 through entry y
                        int m = ((j - ents[x].shift) % GROUP + GROUP) % GROUP;
                        int j2 = (ents[y].shift + m) % GROUP;
                        if (j2 < j) { lv = std::max(lv, lev[j2] + 1); dep |= 1u << x; }
                    }
                lev[j] = (uint8_t)lv;
                g.cninfo[(size_t)i * GROUP + j] = (uint32_t)lv | (dep << 8) | ((uint32_t)j << 20);
                lmax = std::max(lmax, lv);
            }
            if (L.band) {
                // every earlier node sharing a bit lies in an earlier band (else the layer keeps the level schedule only)
                for (int j = 0; j < GROUP && L.band; ++j)
                    for (int x = 0; x < L.n_conflict && L.band; ++x)
                        for (int y = 0; y < L.n_conflict; ++y) {
                            if (x == y || ents[x].group != ents[y].group) continue;
                            const int m = ((j - ents[x].shift) % GROUP + GROUP) % GROUP, j2 = (ents[y].shift + m) % GROUP;
                            if (j2 < j && j2 / L.band >= j / L.band) { L.band = 0; L.band_prefetch = 0; break; }
                        }
            }
            if (L.kind == T2_LAYER_GENERIC) {
                // thread t of the workgroup takes node order[t]: nodes sorted by level, so that a level step keeps one or two
                // wavefronts busy instead of a few lanes in every wavefront
                std::vector<int> order(GROUP);
                for (int j = 0; j < GROUP; ++j) order[j] = j;
                std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return lev[a] < lev[b]; });
                std::vector<uint32_t> byt(GROUP);
                for (int t = 0; t < GROUP; ++t) byt[t] = g.cninfo[(size_t)i * GROUP + order[t]];
                for (int t = 0; t < GROUP; ++t) g.cninfo[(size_t)i * GROUP + t] = byt[t];
            }
        }
        L.lmax = lmax;
        g.total_levels += lmax;
        if (L.kind == T2_LAYER_PAIR) g.serial_steps += std::max(0, (GROUP - 1) / L.step - 1);
        else if (L.kind == T2_LAYER_GENERIC) g.serial_steps += lmax;
        g.links_total += GROUP * (L.cnt + 2) - (i == 0 ? 1 : 0);
    }
    // closing barriers the two-frame kernel may leave out (LdpcLayer::no_close)
    {
        auto groups_of = [&](int i) {
            std::vector<int> v;
            for (int c = 0; c < g.layers[i].cnt; ++c) v.push_back((int)(g.entries[g.layers[i].first_entry + c] & 0xffffu) / GROUP);
            return v;
        };
        auto simple = [&](int i) { return g.layers[i].kind == T2_LAYER_PLAIN || g.layers[i].kind == T2_LAYER_PAIR; };
        std::vector<int> open = groups_of(0);
        bool open_pair = g.layers[0].kind == T2_LAYER_PAIR;
        for (int i = 0; i + 1 < g.q; ++i) {
            const std::vector<int> nxt = groups_of(i + 1);
            const bool next_pair = g.layers[i + 1].kind == T2_LAYER_PAIR;
            bool ok = simple(i) && simple(i + 1) && !(next_pair && open_pair);
            for (int a : nxt)
                for (int b : open) ok = ok && a != b;
            g.layers[i].no_close = ok ? 1 : 0;
            if (ok && !next_pair) { open.insert(open.end(), nxt.begin(), nxt.end()); }
            else { open = nxt; open_pair = false; }               // a closing barrier, or the next layer's inner one, closes all before
            open_pair = open_pair || next_pair;
        }
    }
    return true;
}

int ldpc_k_bch(int code_id)
{
    static const int kb[12] = {7032, 9552, 10632, 11712, 12432, 13152, 32208, 38688, 43040, 48408, 51648, 53840};
    return (code_id >= 0 && code_id < 12) ? kb[code_id] : -1;
}

}  // namespace t2gpu
