/*
 * oracle/ofdm_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
 *
 * CPU restatement of the reference's OFDM-side stages for the FFT sizes it supports end to end (16K / 32K, SISO):
 *   mode arithmetic            dvbt2_p2_parameters_init / dvbt2_bwt_ext_parameters_init / dvbt2_data_parameters_init
 *                              (/root/reference/src/DVB_T2/dvbt2_definition.cpp:20-91,93-159,161-648)
 *   pilot generator            pilot_generator::{init_prbs,p2_generator,data_generator,...} (/root/reference/src/DVB_T2/pilot_generator.cpp:48-2166)
 *   frequency de-interleaver   address_freq_deinterleaver::{init,p2_...,data_...} (/root/reference/src/DVB_T2/address_freq_deinterleaver.cpp:28-209)
 *   LUT trig / atan2           /root/reference/src/DSP/fast_math.h:27-42,61-81
 *   data-symbol equaliser      data_symbol::execute (/root/reference/src/DVB_T2/data_symbol.cpp:108-335)
 *
 * Parity status: fast_math PINNED (tests compare ora_atan2_approx / ora_cos_lut / ora_sin_lut with oracle/_ref/libref_dsp.so,
 * the reference header compiled unmodified). Everything else UNPINNED: those reference files need Qt, which this image
 * lacks (see oracle/fec_oracle.c header for what stands in: transmitter-model cross-checks and end-to-end decoding).
 * The continual-pilot / reserved-tone / PN tables are the generated data of csrc/tables/ofdm_tables_data.h.
 *
 * Floating point: built -O2 -ffp-contract=off, every operation in source order (the reference is -Ofast without FMA).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../sdr_receiver_dvb_t2_amd/csrc/tables/ofdm_tables_data.h"

enum { DATA_CARRIER = 1, P2CARRIER, P2PAPR_CARRIER, TRPAPR_CARRIER, SCATTERED_CARRIER, CONTINUAL_CARRIER };

typedef struct {
    int fft_mode, carrier_mode, pilot_pattern, guard_interval_mode, papr_mode, n_data;   /* inputs */
    int is32k, fft_size, k_total, k_ext, k_offset, l_nulls, n_p2, c_p2, c_data, n_fc, c_fc, l_fc, len_frame;
    int dx, dy;
    float amp_sp, amp_cp, amp_p2;
} ora_mode;

/* dvbt2_definition.cpp:20-91 (SISO rows), :93-159, :161-648 */
int ora_mode_init(ora_mode *m)
{
    if (m->fft_mode == 4 || m->fft_mode == 11) m->is32k = 0;
    else if (m->fft_mode == 5 || m->fft_mode == 7) m->is32k = 1;
    else return -1;
    m->n_p2 = 1;
    m->c_p2 = m->is32k ? 22432 : 8944;
    m->fft_size = m->is32k ? 32768 : 16384;
    if (m->is32k) {
        if (m->carrier_mode == 0) { m->k_total = 27265; m->k_ext = 0; m->k_offset = 288; }
        else { m->k_total = 27841; m->k_ext = 288; m->k_offset = 0; }
    } else {
        if (m->carrier_mode == 0) { m->k_total = 13633; m->k_ext = 0; m->k_offset = 144; }
        else { m->k_total = 13921; m->k_ext = 144; m->k_offset = 0; }
    }
    m->l_nulls = ((m->fft_size - m->k_total) / 2) + 1;
    static const int32_t *const CELLS[2][8] = {
        {T2_CELLS_16K_PP1, T2_CELLS_16K_PP2, T2_CELLS_16K_PP3, T2_CELLS_16K_PP4, T2_CELLS_16K_PP5, T2_CELLS_16K_PP6, T2_CELLS_16K_PP7, T2_CELLS_16K_PP8},
        {T2_CELLS_32K_PP1, T2_CELLS_32K_PP2, T2_CELLS_32K_PP3, T2_CELLS_32K_PP4, T2_CELLS_32K_PP5, T2_CELLS_32K_PP6, T2_CELLS_32K_PP7, T2_CELLS_32K_PP8}};
    const int32_t *c = CELLS[m->is32k][m->pilot_pattern] + (m->carrier_mode ? 6 : 0);
    m->c_data = c[1]; m->n_fc = c[3]; m->c_fc = c[5];
    if (m->papr_mode == 2 || m->papr_mode == 3) {
        int tr = m->is32k ? 288 : 144;
        if (m->c_data != 0) m->c_data -= tr;
        if (m->n_fc != 0) m->n_fc -= tr;
        if (m->c_fc != 0) m->c_fc -= tr;
    }
    int gi = m->guard_interval_mode, pp = m->pilot_pattern;
    if (gi == 4 && pp == 6) { m->n_fc = 0; m->c_fc = 0; }
    if (gi == 0 && pp == 3) { m->n_fc = 0; m->c_fc = 0; }
    if (gi == 1 && pp == 1) { m->n_fc = 0; m->c_fc = 0; }
    if (gi == 6 && pp == 1) { m->n_fc = 0; m->c_fc = 0; }
    m->l_fc = m->n_fc == 0 ? 0 : 1;
    m->len_frame = m->n_p2 + m->n_data;
    /* pilot_generator::sp_amplitudes (:430-506), data_symbol::init amp_cp (:48-66), p2_pilot_amplitudes (:367-382) */
    static const int DX[8] = {3, 6, 6, 12, 12, 24, 24, 6}, DY[8] = {4, 2, 4, 2, 4, 2, 4, 16};
    m->dx = DX[pp]; m->dy = DY[pp];
    m->amp_sp = pp < 2 ? 4.0f / 3.0f : (pp < 4 ? 7.0f / 4.0f : 7.0f / 3.0f);
    m->amp_cp = 8.0f / 3.0f;
    m->amp_p2 = (m->is32k ? sqrtf(37.0f) : sqrtf(31.0f)) / 5.0f;
    return m->c_data ? 0 : -1;
}

/* ---- pilot generator ------------------------------------------------------------------------------------------- */
static void make_prbs(int *prbs, int len)          /* pilot_generator.cpp:48-60 */
{
    int sr = 0x7ff;
    for (int i = 0; i < len; ++i) {
        int b = ((sr) ^ (sr >> 2)) & 1;
        prbs[i] = sr & 1;
        sr >>= 1;
        if (b) sr |= 0x400;
    }
}
static int pn_bit(int j) { return (T2_PN_SEQUENCE_BYTES[j / 8] >> (7 - j % 8)) & 0x1; }   /* :61-66 */

/* carrier map (int) and pilot reference (float) of symbol idx_symbol (0 = P2), k_total entries each */
int ora_symbol_carriers(const ora_mode *m, int idx_symbol, int *map, float *refer)
{
    const int K = m->k_total;
    int *prbs = (int *)malloc(sizeof(int) * (K + m->k_offset));
    make_prbs(prbs, K + m->k_offset);
    if (idx_symbol < m->n_p2) {
        /* p2_carrier_mapping (:133-365, SISO), p2_modulation (:2093-2112) */
        for (int i = 0; i < K; ++i) map[i] = DATA_CARRIER;
        int step = m->is32k ? 6 : 3;
        for (int i = 0; i < K; i += step) map[i] = P2CARRIER;
        if (m->carrier_mode == 1)
            for (int i = 0; i < m->k_ext; ++i) { map[i] = P2CARRIER; map[i + (K - m->k_ext)] = P2CARRIER; }
        const uint16_t *pp = m->is32k ? T2_P2_PAPR_32K : T2_P2_PAPR_16K;
        for (int i = 0; i < (m->is32k ? 288 : 144); ++i) map[pp[i] + m->k_ext] = P2PAPR_CARRIER;
        float bpsk[2] = {m->amp_p2, -m->amp_p2};
        for (int n = 0; n < K; ++n) refer[n] = map[n] == P2CARRIER ? bpsk[prbs[n + m->k_offset] ^ pn_bit(idx_symbol)] : 0.0f;
        free(prbs);
        return 0;
    }
    const int tr = m->papr_mode == 2 || m->papr_mode == 3;
    const float sp_bpsk[2] = {m->amp_sp, -m->amp_sp}, cp_bpsk[2] = {m->amp_cp, -m->amp_cp};
    if (m->l_fc && idx_symbol == m->len_frame - m->l_fc) {
        /* fc_carrier_mapping (:2011-2091) */
        for (int i = 0; i < K; ++i) map[i] = DATA_CARRIER;
        for (int i = 0; i < K; ++i) if (i % m->dx == 0) map[i] = SCATTERED_CARRIER;
        map[0] = SCATTERED_CARRIER; map[K - 1] = SCATTERED_CARRIER;
        if (tr) {
            const uint16_t *pp = m->is32k ? T2_P2_PAPR_32K : T2_P2_PAPR_16K;
            for (int i = 0; i < (m->is32k ? 288 : 144); ++i) map[pp[i] + m->k_ext] = TRPAPR_CARRIER;
        }
    } else {
        /* data_carries_mapping, cp_mappinng (:516-1932 via the generated group lists), sp_mappinng (:1934-1960),
         * tr_papr_carriers_mapping (:1962-2009) -- in that order (data_generator :119-124) */
        for (int i = 0; i < K; ++i) map[i] = DATA_CARRIER;
        const t2_cp_set_t *cp = &T2_CP_SETS[m->is32k][m->pilot_pattern];
        for (int i = 0; i < cp->n_cp; ++i) map[cp->cp[i]] = CONTINUAL_CARRIER;
        if (m->carrier_mode == 1) for (int i = 0; i < cp->n_cpx; ++i) map[cp->cpx[i]] = CONTINUAL_CARRIER;
        for (int i = 0; i < K; ++i) {
            int remainder = (i - m->k_ext) % (m->dx * m->dy);
            if (remainder < 0) remainder += (m->dx * m->dy);
            if (remainder == (m->dx * (idx_symbol % m->dy))) map[i] = SCATTERED_CARRIER;
        }
        map[0] = SCATTERED_CARRIER; map[K - 1] = SCATTERED_CARRIER;
        if (tr) {
            int shift = m->carrier_mode == 0 ? m->dx * (idx_symbol % m->dy) : m->dx * ((idx_symbol + (m->k_ext / m->dx)) % m->dy);
            const uint16_t *pp = m->is32k ? T2_TR_PAPR_32K : T2_TR_PAPR_16K;
            for (int i = 0; i < (m->is32k ? 288 : 144); ++i) map[pp[i] + shift] = TRPAPR_CARRIER;
        }
    }
    /* modulation (:2114-2166) */
    for (int n = 0; n < K; ++n) {
        switch (map[n]) {
        case SCATTERED_CARRIER: refer[n] = sp_bpsk[prbs[n + m->k_offset] ^ pn_bit(idx_symbol)]; break;
        case CONTINUAL_CARRIER: refer[n] = cp_bpsk[prbs[n + m->k_offset] ^ pn_bit(idx_symbol)]; break;
        default: refer[n] = 0.0f; break;
        }
    }
    free(prbs);
    return 0;
}

/* ---- frequency de-interleaver (address_freq_deinterleaver.cpp:28-209) ------------------------------------------ */
int ora_freq_deint(const ora_mode *m, int kind, int *h_even, int *h_odd)
{
    static const int logic16k[6] = {0, 1, 4, 5, 9, 11}, logic32k[4] = {0, 1, 2, 12};
    const int *logic = m->is32k ? logic32k : logic16k;
    const int xor_size = m->is32k ? 4 : 6, pn_degree = m->is32k ? 14 : 13, pn_mask = m->is32k ? 0x3fff : 0x1fff;
    const int max_states = m->is32k ? 32768 : 16384;
    const uint8_t *bitpermeven = m->is32k ? T2_FI_PERM_32K : T2_FI_PERM_16K_EVEN;
    const uint8_t *bitpermodd = m->is32k ? T2_FI_PERM_32K : T2_FI_PERM_16K_ODD;
    int *max_even = (int *)malloc(sizeof(int) * max_states), *max_odd = (int *)malloc(sizeof(int) * max_states);
    int lfsr = 0;
    for (int i = 0; i < max_states; i++) {
        if (i == 0 || i == 1) lfsr = 0;
        else if (i == 2) lfsr = 1;
        else {
            int result = 0;
            for (int k = 0; k < xor_size; k++) result ^= (lfsr >> logic[k]) & 1;
            lfsr &= pn_mask;
            lfsr >>= 1;
            lfsr |= result << (pn_degree - 1);
        }
        int even = 0, odd = 0;
        for (int n = 0; n < pn_degree; n++) even |= ((lfsr >> n) & 0x1) << bitpermeven[n];
        for (int n = 0; n < pn_degree; n++) odd |= ((lfsr >> n) & 0x1) << bitpermodd[n];
        max_even[i] = even + ((i % 2) * (max_states / 2));
        max_odd[i] = odd + ((i % 2) * (max_states / 2));
    }
    const int cells = kind == 0 ? m->c_p2 : (kind == 1 ? m->c_data : m->n_fc);
    int *even_tx = (int *)calloc(32768, sizeof(int)), *odd_tx = (int *)calloc(32768, sizeof(int));
    int q_even = 0, q_odd = 0;
    for (int i = 0; i < max_states; i++) {
        if (max_even[i] < cells) even_tx[q_even++] = max_even[i];
        if (max_odd[i] < cells) odd_tx[q_odd++] = max_odd[i];
    }
    if (m->is32k)
        for (int j = 0; j < q_odd; j++) { int a = odd_tx[j]; even_tx[a] = j; }
    for (int i = 0; i < q_even; i++) h_even[even_tx[i]] = i;
    for (int i = 0; i < q_odd; i++) h_odd[odd_tx[i]] = i;
    free(max_even); free(max_odd); free(even_tx); free(odd_tx);
    return cells;
}

/* ---- fast_math.h:27-42,61-81 ------------------------------------------------------------------------------------ */
#undef M_PIf
#undef M_PI_2f
#define M_PIf 3.14159274101257324219f          /* M_PIf32 of glibc */
#define M_PI_2f 1.57079637050628662109f
static float lut_sin[65536], lut_cos[65536];
static int lut_ready = 0;
static const float k_table = 32767.0f / (2.0f * M_PIf);
void ora_lut_init(void)
{
    if (lut_ready) return;
    /* As the reference BINARY computes it (g++ -Ofast): i / k_table becomes i * (1 / k_table) in float
     * (-freciprocal-math) and sin(x), cos(x) of a float argument stored to float become one sincosf(x)
     * (-funsafe-math-optimizations). Verified entry by entry against oracle/_ref/libref_dsp.so. */
    const float rk = 1.0f / k_table;
    for (int i = -32767; i < 32768; i++) sincosf((float)i * rk, &lut_sin[i + 32767], &lut_cos[i + 32767]);
    lut_ready = 1;
}
float ora_sin_lut(float x) { ora_lut_init(); return lut_sin[(int)(x * k_table + 32767) & 65535]; }
float ora_cos_lut(float x) { ora_lut_init(); return lut_cos[(int)(x * k_table + 32767) & 65535]; }
const float *ora_lut_table(int which) { ora_lut_init(); return which ? lut_cos : lut_sin; }
float ora_atan2_approx(float y, float x)
{
    if (x == 0.0f) return y > 0.0f ? M_PI_2f : -M_PI_2f;
    if (y == 0.0f) return x > 0.0f ? 0.0f : -M_PIf;
    float abs_x = fabsf(x), abs_y = fabsf(y);
    int min_x = abs_x < abs_y;
    float a = min_x ? abs_x / abs_y : abs_y / abs_x;
    float s = a * a;
    float r = ((-4.6496475e-2f * s + 1.5931422e-1f) * s - 3.2762276e-1f) * s * a + a;
    if (min_x) r = M_PI_2f - r;
    if (x < 0.0f) r = M_PIf - r;
    if (y < 0.0f) r = -r;
    return r;
}

/* ---- data_symbol::execute (data_symbol.cpp:108-335) ------------------------------------------------------------- */
/* ofdm_cell: the fft-shifted symbol (fft_size complex, re/im interleaved); map/refer: this symbol's tables; h: h_odd when
 * idx_symbol is even, h_even when odd (caller picks, :148-149). out: c_data cells (c_p2 for a P2 symbol's tables). sync[0] = phase_offset,
 * sync[1] = sample_rate_offset. Returns the number of cells written. */
/* kind: 0 = P2 symbol (p2_symbol.cpp:89-262), 1 = data symbol, 2 = frame-closing symbol (fc_symbol.cpp:82-271). The three
 * functions are the same estimator; what differs in the reference BINARY (built -Ofast, sdr_receiver_dvb_t2.pro:33-39) is the
 * pilot amplitude: in p2_symbol / fc_symbol amp_pilot never changes inside the function, so gcc's -freciprocal-math turns
 * `sqrt(norm(cell)) / amp_pilot` into a product with 1/amp_pilot computed once; in data_symbol amp_pilot alternates between the
 * continual and scattered value and the division stays. Pinned against oracle/_ref/libref_t2sym.so (tests/test_ref_pins.py):
 * data symbols bit-exact; P2 / FC bit-exact except cells whose table index int(angle * k + 32767) sits on a rounding boundary
 * (2 of 22 432 in the CFG-A fixture, |difference| < 3e-4) -- with a true division 40 % of the P2 cells differ in the last bits. */
int ora_symbol_equalise(const ora_mode *m, int kind, const float *ofdm_cell_full, const int *map, const float *refer, const int *h,
                        float *out, float *sync)
{
    const int recip = kind != 1;
    ora_lut_init();
    const float *oc = ofdm_cell_full + 2 * (size_t)m->l_nulls;
    const int k_total = m->k_total, half_total = k_total / 2;
    const float amp_sp = m->amp_sp, amp_cp = m->amp_cp;
    float angle = 0, delta_angle, angle_est, amp = 0, delta_amp, amp_est, dif_angle;
    float sum_angle_1 = 0, sum_angle_2 = 0, sp1r = 0, sp1i = 0, sp2r = 0, sp2i = 0;
    float amp_pilot = map[0] == P2CARRIER ? m->amp_p2 : amp_sp;
    float *buf = (float *)malloc(sizeof(float) * 2 * (size_t)m->k_total + 16);
    int idx_data = 0, d = 0;
    {   /* first pilot */
        float cr = oc[0], ci = oc[1], pr = refer[0];
        float er = cr * pr, ei = ci * pr;
        sp1r += er; sp1i += ei;
        angle_est = ora_atan2_approx(ei, er);
        amp_est = recip ? sqrtf(cr * cr + ci * ci) * (1.0f / amp_pilot) : sqrtf(cr * cr + ci * ci) / amp_pilot;
    }
    for (int pass = 0; pass < 2; ++pass) {
        int lo = pass == 0 ? 1 : half_total + 1, hi = pass == 0 ? half_total : k_total;
        for (int i = lo; i < hi; ++i) {
            float cr = oc[2 * i], ci = oc[2 * i + 1], pr = refer[i];
            switch (map[i]) {
            case DATA_CARRIER:
                buf[2 * idx_data] = cr; buf[2 * idx_data + 1] = ci; ++idx_data;
                break;
            case P2CARRIER:                 /* p2_symbol.cpp:89-262: same estimator, every pilot has amplitude amp_p2 */
                amp_pilot = m->amp_p2;
                /* fallthrough */
            case CONTINUAL_CARRIER:
                if (map[i] == CONTINUAL_CARRIER) amp_pilot = amp_cp;
                /* fallthrough */
            case SCATTERED_CARRIER: {
                float er = cr * pr, ei = ci * pr;
                if (pass == 0) { sp1r += er; sp1i += ei; } else { sp2r += er; sp2i += ei; }
                angle = ora_atan2_approx(ei, er);
                dif_angle = angle - angle_est;
                if (dif_angle > M_PIf) dif_angle = M_PIf * 2.0f - dif_angle;
                else if (dif_angle < -M_PIf) dif_angle = M_PIf * 2.0f + dif_angle;
                if (pass == 0) sum_angle_1 += angle; else sum_angle_2 += angle;
                delta_angle = (dif_angle) / (idx_data + 1);
                amp = recip ? sqrtf(cr * cr + ci * ci) * (1.0f / amp_pilot) : sqrtf(cr * cr + ci * ci) / amp_pilot;
                amp_pilot = amp_sp;
                delta_amp = (amp - amp_est) / (idx_data + 1);
                for (int j = 0; j < idx_data; ++j) {
                    angle_est += delta_angle;
                    amp_est += delta_amp;
                    float dr = ora_cos_lut(angle_est) / amp_est, di = ora_sin_lut(angle_est) / amp_est;
                    float br = buf[2 * j], bi = buf[2 * j + 1];
                    /* buffer_cell[j] * conj(derotate) */
                    out[2 * h[d]] = br * dr + bi * di;
                    out[2 * h[d] + 1] = bi * dr - br * di;
                    ++d;
                }
                idx_data = 0;
                angle_est = angle;
                amp_est = amp;
                break;
            }
            default: break;     /* TRPAPR_CARRIER: skipped (":215 //TODO") */
            }
        }
        if (pass == 0) {        /* the centre carrier (:224-256): data is buffered, a pilot there is not used */
            int i = half_total;
            if (map[i] == DATA_CARRIER) { buf[2 * idx_data] = oc[2 * i]; buf[2 * idx_data + 1] = oc[2 * i + 1]; ++idx_data; }
            else if (map[i] == CONTINUAL_CARRIER || map[i] == SCATTERED_CARRIER) amp_pilot = amp_sp;
        }
    }
    if (sync) {
        sync[0] = ora_atan2_approx(sp2i, sp2r) + ora_atan2_approx(sp1i, sp1r);
        sync[1] = sum_angle_2 - sum_angle_1;
    }
    free(buf);
    return d;
}

/* kind inferred for callers that only equalise P2 / data symbols */
int ora_data_symbol(const ora_mode *m, const float *ofdm_cell_full, const int *map, const float *refer, const int *h,
                    float *out, float *sync)
{
    return ora_symbol_equalise(m, map[0] == P2CARRIER ? 0 : 1, ofdm_cell_full, map, refer, h, out, sync);
}
