#include "fec_tables.h"
#include "tables/bitint_tables_data.h"

namespace t2gpu {

bool bitdeint_address(int mod, int fec_type, int code_rate, std::vector<uint16_t> &address)
{
    const int bpc = 2 * (mod + 1);
    int rate_sel = -1;
    if (fec_type == 1) {
        if (code_rate == 1) rate_sel = 1;
        else if (code_rate == 2 && mod == 3) rate_sel = 2;
    }
    const t2_bitint_cfg_t *cfg = nullptr;
    for (int i = 0; i < T2_BITINT_NUM_CFG; ++i) {
        const t2_bitint_cfg_t &c = T2_BITINT_CFG[i];
        if (c.bits_per_cell == bpc && c.fec_normal == fec_type && c.rate_sel == rate_sel) cfg = &c;
    }
    if (!cfg) return false;
    const int cols = cfg->columns, rows = cfg->rows;
    address.resize((size_t)cols * rows);
    // LLR k sits in demux sub-stream n = k mod rows of cell-group c = k div rows; the demultiplexer maps it to
    // interleaver column r = demux[n]; the column-twist memory is written column-wise with a start offset of twist[r].
    for (int c = 0; c < cols; ++c)
        for (int n = 0; n < rows; ++n) {
            const int r = cfg->demux[n];
            address[(size_t)c * rows + n] = (uint16_t)(cols * r + (c + cols - cfg->twist[r]) % cols);
        }
    return true;
}

void cell_deint_permutation(int num_blocks, int cells_per_fec, std::vector<int32_t> &perm)
{
    int nd = 0;
    while ((1 << nd) < cells_per_fec) ++nd;                 // ceil(log2(Ncells))
    // taps of the (nd-1)-bit shift register R', EN 302 755 table 2 (cell interleaver)
    uint32_t taps = 0;
    switch (nd) {
    case 11: taps = (1u << 0) | (1u << 3); break;
    case 12: taps = (1u << 0) | (1u << 2); break;
    case 13: taps = (1u << 0) | (1u << 1) | (1u << 4) | (1u << 6); break;
    case 15: taps = (1u << 0) | (1u << 1) | (1u << 2) | (1u << 12); break;
    default: taps = (1u << 0) | (1u << 1) | (1u << 4) | (1u << 5) | (1u << 9) | (1u << 11); break;   // 14
    }
    const uint32_t mask = (1u << (nd - 1)) - 1;
    std::vector<int32_t> L0;
    L0.reserve(cells_per_fec);
    uint32_t reg = 0;
    for (int i = 0; i < (1 << nd); ++i) {
        if (i < 2) reg = 0;
        else if (i == 2) reg = 1;
        else {
            uint32_t fb = __builtin_popcount(reg & taps) & 1u;
            reg = ((reg & mask) >> 1) | (fb << (nd - 2));
        }
        uint32_t v = (reg & mask) | ((uint32_t)(i & 1) << (nd - 1));
        if ((int)v < cells_per_fec) L0.push_back((int32_t)v);
    }
    perm.assign((size_t)num_blocks * cells_per_fec, 0);
    int counter = 0;
    for (int r = 0; r < num_blocks; ++r) {
        int shift;
        do {                                                  // bit-reversed nd-bit counter, values >= Ncells skipped
            shift = 0;
            for (int p = 0; p < nd; ++p) shift |= ((counter >> p) & 1) << (nd - 1 - p);
            // the reference's loop shifts once more than it ORs (time_deinterleaver.cpp:250-256): value << 1
            shift <<= 1;
            ++counter;
        } while (shift >= cells_per_fec);
        for (int w = 0; w < cells_per_fec; ++w)
            perm[(size_t)r * cells_per_fec + (L0[w] + shift) % cells_per_fec] = r * cells_per_fec + w;
    }
}

void bb_prbs(std::vector<uint8_t> &bits, int n)
{
    bits.resize(n);
    uint32_t sr = 0x4A80;                                     // 100101010000000, bit 0 leaves first
    for (int i = 0; i < n; ++i) {
        uint32_t b = (sr ^ (sr >> 1)) & 1u;
        bits[i] = (uint8_t)b;
        sr = (sr >> 1) | (b << 14);
    }
}

}  // namespace t2gpu
