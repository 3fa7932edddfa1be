#include "ofdm_tables.h"
#include "tables/ofdm_tables_data.h"
#include <cmath>

namespace t2gpu {

bool t2_mode_init(T2Mode &m)
{
    if (m.fft_mode == 4 || m.fft_mode == 11) m.is32k = 0;
    else if (m.fft_mode == 5 || m.fft_mode == 7) m.is32k = 1;
    else return false;                                     // 1K..8K: tables not carried (reference: 16K/32K only, README:17-23)
    if (m.pilot_pattern < 0 || m.pilot_pattern > 7 || m.carrier_mode < 0 || m.carrier_mode > 1) return false;
    m.fft_size = m.is32k ? 32768 : 16384;
    const int knorm = m.is32k ? 27265 : 13633, kext = m.is32k ? 288 : 144;
    if (m.carrier_mode) { m.k_total = knorm + 2 * kext; m.k_ext = kext; m.k_offset = 0; }
    else { m.k_total = knorm; m.k_ext = 0; m.k_offset = kext; }
    m.l_nulls = (m.fft_size - m.k_total) / 2 + 1;
    m.n_p2 = 1;
    m.c_p2 = m.is32k ? 22432 : 8944;
    static const int DX[8] = {3, 6, 6, 12, 12, 24, 24, 6}, DY[8] = {4, 2, 4, 2, 4, 2, 4, 16};
    m.dx = DX[m.pilot_pattern]; m.dy = DY[m.pilot_pattern];
    m.amp_sp = m.pilot_pattern < 2 ? 4.0f / 3.0f : (m.pilot_pattern < 4 ? 7.0f / 4.0f : 7.0f / 3.0f);
    m.amp_cp = 8.0f / 3.0f;
    m.amp_p2 = (m.is32k ? std::sqrt(37.0f) : std::sqrt(31.0f)) / 5.0f;
    // cells per symbol: (c_data, n_fc, c_fc) for normal then extended carrier mode
    static const int32_t *const CELLS[2][8] = {
        {T2_CELLS_16K_PP1, T2_CELLS_16K_PP2, T2_CELLS_16K_PP3, T2_CELLS_16K_PP4, T2_CELLS_16K_PP5, T2_CELLS_16K_PP6, T2_CELLS_16K_PP7, T2_CELLS_16K_PP8},
        {T2_CELLS_32K_PP1, T2_CELLS_32K_PP2, T2_CELLS_32K_PP3, T2_CELLS_32K_PP4, T2_CELLS_32K_PP5, T2_CELLS_32K_PP6, T2_CELLS_32K_PP7, T2_CELLS_32K_PP8}};
    const int32_t *c = CELLS[m.is32k][m.pilot_pattern] + (m.carrier_mode ? 6 : 0);
    m.c_data = c[1]; m.n_fc = c[3]; m.c_fc = c[5];
    if (m.c_data == 0) return false;                       // combination not allowed for this FFT size
    if (m.papr_mode == 2 || m.papr_mode == 3) {
        const int tr = m.is32k ? 288 : 144;
        if (m.c_data) m.c_data -= tr;
        if (m.n_fc) m.n_fc -= tr;
        if (m.c_fc) m.c_fc -= tr;
    }
    // guard-interval / pilot-pattern pairs without frame-closing symbol (dvbt2_definition.cpp:601-618)
    const int gi = m.guard_interval_mode, pp = m.pilot_pattern;
    if ((gi == 4 && pp == 6) || (gi == 0 && pp == 3) || (gi == 1 && pp == 1) || (gi == 6 && pp == 1)) { m.n_fc = 0; m.c_fc = 0; }
    switch (gi) {
    case 3: m.guard_interval_size = m.fft_size / 4; break;
    case 2: m.guard_interval_size = m.fft_size / 8; break;
    case 1: m.guard_interval_size = m.fft_size / 16; break;
    case 0: m.guard_interval_size = m.fft_size / 32; break;
    case 4: m.guard_interval_size = m.fft_size / 128; break;
    case 5: m.guard_interval_size = (m.fft_size / 128) * 19; break;
    case 6: m.guard_interval_size = (m.fft_size / 256) * 19; break;
    default: return false;
    }
    m.l_fc = m.n_fc ? 1 : 0;
    m.len_frame = m.n_p2 + m.n_data;
    return true;
}

static int pilot_prbs(int i)        // w_i of x^11 + x^2 + 1, all-ones start (pilot_generator.cpp:54-60)
{
    static std::vector<uint8_t> seq;
    if (seq.empty()) {
        seq.resize(32768 + 512);
        uint32_t sr = 0x7ff;
        for (size_t n = 0; n < seq.size(); ++n) {
            seq[n] = sr & 1u;
            uint32_t b = (sr ^ (sr >> 2)) & 1u;
            sr = (sr >> 1) | (b << 10);
        }
    }
    return seq[i];
}
static int frame_pn(int l) { return (T2_PN_SEQUENCE_BYTES[l >> 3] >> (7 - (l & 7))) & 1; }

void t2_symbol_carriers(const T2Mode &m, int idx_symbol, std::vector<uint8_t> &map, std::vector<float> &refer)
{
    const int K = m.k_total;
    map.assign(K, T2_DATA);
    refer.assign(K, 0.0f);
    const int pn = frame_pn(idx_symbol);
    auto sign = [&](int k, float amp) { return (pilot_prbs(k + m.k_offset) ^ pn) ? -amp : amp; };
    if (idx_symbol < m.n_p2) {
        // P2: a pilot every 6th (32K SISO) or 3rd carrier, every carrier of the extension bands, reserved tones empty
        const int step = m.is32k ? 6 : 3;
        for (int k = 0; k < K; ++k)
            if (k % step == 0 || k < m.k_ext || k >= K - m.k_ext) map[k] = T2_P2PILOT;
        const uint16_t *pp = m.is32k ? T2_P2_PAPR_32K : T2_P2_PAPR_16K;
        for (int i = 0; i < (m.is32k ? 288 : 144); ++i) map[pp[i] + m.k_ext] = T2_P2PAPR;
        for (int k = 0; k < K; ++k)
            if (map[k] == T2_P2PILOT) refer[k] = sign(k, m.amp_p2);
        return;
    }
    const bool tr = m.papr_mode == 2 || m.papr_mode == 3;
    if (m.l_fc && idx_symbol == m.len_frame - 1) {
        // frame-closing symbol: a pilot every dx carriers + the edges (pilot_generator.cpp:2011-2091)
        for (int k = 0; k < K; ++k)
            if (k % m.dx == 0) map[k] = T2_SCATTERED;
        map[0] = map[K - 1] = T2_SCATTERED;
        if (tr) {
            const uint16_t *pp = m.is32k ? T2_P2_PAPR_32K : T2_P2_PAPR_16K;
            for (int i = 0; i < (m.is32k ? 288 : 144); ++i) map[pp[i] + m.k_ext] = T2_TRPAPR;
        }
    } else {
        const t2_cp_set_t &cp = T2_CP_SETS[m.is32k][m.pilot_pattern];
        for (int i = 0; i < cp.n_cp; ++i) map[cp.cp[i]] = T2_CONTINUAL;
        if (m.carrier_mode)
            for (int i = 0; i < cp.n_cpx; ++i) map[cp.cpx[i]] = T2_CONTINUAL;
        const int period = m.dx * m.dy, phase = m.dx * (idx_symbol % m.dy);
        for (int k = 0; k < K; ++k) {
            int r = (k - m.k_ext) % period;
            if (r < 0) r += period;
            if (r == phase) map[k] = T2_SCATTERED;
        }
        map[0] = map[K - 1] = T2_SCATTERED;
        if (tr) {
            const int shift = m.carrier_mode ? m.dx * ((idx_symbol + m.k_ext / m.dx) % m.dy) : phase;
            const uint16_t *pp = m.is32k ? T2_TR_PAPR_32K : T2_TR_PAPR_16K;
            for (int i = 0; i < (m.is32k ? 288 : 144); ++i) map[pp[i] + shift] = T2_TRPAPR;
        }
    }
    for (int k = 0; k < K; ++k) {
        if (map[k] == T2_SCATTERED) refer[k] = sign(k, m.amp_sp);
        else if (map[k] == T2_CONTINUAL) refer[k] = sign(k, m.amp_cp);
    }
}

void t2_freq_deint(const T2Mode &m, int kind, std::vector<int32_t> &h_even, std::vector<int32_t> &h_odd)
{
    const int cells = kind == 0 ? m.c_p2 : (kind == 1 ? m.c_data : m.n_fc);
    const int nr = m.is32k ? 15 : 14;                      // log2(M_max)
    const int mmax = 1 << nr;
    const uint8_t *pe = m.is32k ? T2_FI_PERM_32K : T2_FI_PERM_16K_EVEN;
    const uint8_t *po = m.is32k ? T2_FI_PERM_32K : T2_FI_PERM_16K_ODD;
    const uint32_t taps = m.is32k ? ((1u << 0) | (1u << 1) | (1u << 2) | (1u << 12))
                                  : ((1u << 0) | (1u << 1) | (1u << 4) | (1u << 5) | (1u << 9) | (1u << 11));
    // H(q): addresses < cells in generation order (EN 302 755 8.5); R' has nr-1 bits, the MSB toggles with i
    std::vector<int32_t> He, Ho;
    uint32_t reg = 0;
    const uint32_t mask = (1u << (nr - 1)) - 1;
    for (int i = 0; i < mmax; ++i) {
        if (i < 2) reg = 0;
        else if (i == 2) reg = 1;
        else {
            uint32_t fb = __builtin_popcount(reg & taps) & 1u;
            reg = ((reg & mask) >> 1) | (fb << (nr - 2));
        }
        uint32_t e = 0, o = 0;
        for (int b = 0; b < nr - 1; ++b) {
            e |= ((reg >> b) & 1u) << pe[b];
            o |= ((reg >> b) & 1u) << po[b];
        }
        e += (i & 1) * (mmax / 2);
        o += (i & 1) * (mmax / 2);
        if ((int)e < cells) He.push_back((int32_t)e);
        if ((int)o < cells) Ho.push_back((int32_t)o);
    }
    h_even.assign(cells, 0);
    h_odd.assign(cells, 0);
    for (int q = 0; q < cells; ++q) h_odd[Ho[q]] = q;
    if (m.is32k) {
        // 32K uses one permutation and its inverse (address_freq_deinterleaver.cpp:149-155,185-196): the "even" receive
        // table is the odd transmit permutation itself
        for (int q = 0; q < cells; ++q) h_even[q] = Ho[q];
    } else {
        for (int q = 0; q < cells; ++q) h_even[He[q]] = q;
    }
}

}  // namespace t2gpu
