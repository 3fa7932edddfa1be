// ldpc_graph.h -- host-side construction of the layered Tanner-graph description the LDPC kernel walks.
//
// What the reference computes at init time (LDPCDecoder::init, /root/reference/src/DVB_T2/LDPC/layered_decoder.hh:115-167,
// from the table walk of LDPC<TABLE>, LDPC/ldpc.hh:39-123) is a per-check-node list of bit indices. Here the same
// graph is kept in its quasi-cyclic form instead: check node (i, j) -- layer i in [0,q), j in [0,360) -- is
// connected, for every table address x with x mod q == i, to information bit 360*g + (j - x/q) mod 360 of group g.
// One (base = 360*g, shift = x/q) pair per layer therefore describes 360 links, and a wavefront of consecutive j
// reads consecutive LLR bytes.
//
// The reference visits the 360 nodes of a layer strictly in ascending j (layered_decoder.hh:86-108). Two nodes of
// a layer that share a bit (two table addresses of one group falling into the same layer) are therefore ordered,
// and the later one sees the earlier one's update. `level` encodes that order as the longest dependency chain
// ending in each node: nodes of equal level are independent and may run concurrently, levels run in sequence.
#pragma once
#include <stdint.h>
#include <vector>

namespace t2gpu {

struct LdpcLayer {
    int first_entry;   // index into entries[]
    int cnt;           // information-bit links per node in this layer (reference: cnc[i])
    int lmax;          // number of dependency levels (1 = conflict-free layer)
    int n_conflict;    // leading entries that belong to a group appearing more than once in this layer
    int kind;          // T2_LAYER_PLAIN / PAIR / GENERIC (ldpc_cn.h)
    int step;          // PAIR: (shift of slot 1 - shift of slot 0) mod 360, 1..180
    // GENERIC layers with at most four conflict slots can also be walked in BANDS (ldpc_cn3.h, p2_band_walk): `band` = D, the smallest
    // index distance of two nodes that share a bit. Nodes [D t, D t + D) are then independent of each other and depend only on bands
    // before t, which is the reference's ascending-j order again. Slots 0 / 1 are the pair whose bit node j hands to node j + D
    // (through slot 0 of j, slot 1 of j + D). band_prefetch: every other dependency reaches back two bands or more. 0 = not eligible
    // (more than four conflict slots, or D > 32: those layers are few-level ones and keep the level schedule).
    int band = 0, band_prefetch = 0;
    // The two-frame kernel may run on into the next layer without the workgroup barrier that closes this one: both layers are PLAIN or
    // PAIR (node j on the same lane pair in both, so the parity bits they share go through one wavefront's in-order LDS), not both
    // PAIR (they would share the chain-record scratch), and the next layer's information-bit groups are disjoint from those of
    // every layer still open since the last barrier (a PAIR layer's inner barrier closes everything before it).
    int no_close = 0;
};

struct LdpcGraph {
    int id = -1, n = 0, k = 0, r = 0, q = 0;
    int max_cnt = 0, min_cnt = 1 << 30;
    int links_total = 0;
    int total_levels = 0;
    std::vector<LdpcLayer> layers;     // [q]
    std::vector<uint32_t> entries;     // packed: base (bits 0..15) | shift (bits 16..24)
    std::vector<uint8_t> levels;       // [q*360], values 1..lmax
    std::vector<uint32_t> cninfo;      // [q*360]: level (bits 0..7) | dependent-conflict-slot mask << 8 | node j << 20. In GENERIC
                                       // layers entry t describes the node THREAD t takes (nodes sorted by level), else node t
    int serial_steps = 0;              // sum over layers of sequential steps a sweep needs (chain walks + level steps)
};

// fec_type / code_rate use the reference enums (dvbt2_definition.h:60-67,85-88): fec_type 0 = short (16200),
// 1 = normal (64800); code_rate 0..5 = 1/2, 3/5, 2/3, 3/4, 4/5, 5/6.
inline int ldpc_code_id(int fec_type, int code_rate)
{
    if (fec_type < 0 || fec_type > 1 || code_rate < 0 || code_rate > 5) return -1;
    return fec_type * 6 + code_rate;
}

bool ldpc_build_graph(int code_id, LdpcGraph &g);

// k_bch for the BCH stub / BB descrambler (bch_decoder.cpp:79-134)
int ldpc_k_bch(int code_id);

}  // namespace t2gpu
