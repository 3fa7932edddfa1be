/*
 * oracle/ref_t2sym.cpp -- TEST INFRASTRUCTURE ONLY. Thin extern "C" driver that INSTANTIATES AND CALLS the reference's own
 * symbol-level classes where they lie under /root/reference/src/DVB_T2 (compiled unmodified by oracle/Makefile with the
 * reference's flags, real Qt 5.9.7 headers + moc from /opt/conda, the FFTW binary the reference ships):
 *     dvbt2_definition.cpp, pilot_generator.cpp, address_freq_deinterleaver.cpp, p2_symbol.cpp, data_symbol.cpp, fc_symbol.cpp,
 *     p1_symbol.cpp.
 * No reference source is copied and no stand-in header is written. Output: oracle/_ref/libref_t2sym.so (git-ignored).
 * It exists to (i) generate tests/golden/t2sym_golden.npz (tests/golden/make_t2sym_golden.py) and (ii) let the CPU tier compare
 * oracle/*.c against the reference live in this container. Nothing in the product links or loads it.
 *
 * The call order is the one dvbt2_demodulator follows: P1 gives preamble + FFT mode; init_dvbt2 (dvbt2_demodulator.cpp:129-143)
 * = dvbt2_p2_parameters_init, address init, p2_symbol::init; the first P2 symbol's L1-pre (p2_symbol.cpp:493-500) gives carrier
 * mode, guard interval, PAPR, pilot pattern and the data-symbol count; then data_symbol::init / fc_symbol::init (:386-395).
 */
#include <QtCore/QObject>
#include <QtCore/QString>
#include <cstring>
#include <complex>
#include <vector>

#define private public                /* read-only access to the classes' tables for the fixtures; class layout is unchanged */
#include "dvbt2_definition.h"
#include "pilot_generator.h"
#include "address_freq_deinterleaver.h"
#include "p2_symbol.h"
#include "data_symbol.h"
#include "fc_symbol.h"
#include "p1_symbol.h"
#undef private

namespace {
struct ref_sym {
    dvbt2_parameters dvbt2{};
    pilot_generator *pilot = new pilot_generator;
    address_freq_deinterleaver *fq = new address_freq_deinterleaver;
    p2_symbol *p2 = new p2_symbol;
    data_symbol *data = new data_symbol;
    fc_symbol *fc = new fc_symbol;
    l1_presignalling l1_pre;
    l1_postsignalling l1_post;
    bool data_ready = false;
};

void put_dvbt2(const dvbt2_parameters &d, int *o)
{
    const int v[] = { d.preamble, d.bandwidth, d.miso, d.miso_group, d.fft_mode, d.fft_size, d.guard_interval_mode,
                      d.guard_interval_size, d.carrier_mode, d.l_nulls, d.pilot_pattern, d.papr_mode, d.l1_mod, d.l1_cod,
                      d.l1_fec_type, d.l1_post_size, d.l1_post_info_size, d.c_p2, d.n_p2, d.c_data, d.c_fc, d.n_fc, d.k_total,
                      d.k_ext, d.k_offset, d.len_frame, d.n_data, d.n_t2, d.l_fc, d.t2_version };
    std::memcpy(o, v, sizeof v);
}
}  // namespace

extern "C" {

/* what P1 decoding leaves in dvbt2 (p1_symbol.cpp:180-298) followed by init_dvbt2 (dvbt2_demodulator.cpp:129-143) */
void *ref_sym_new(int preamble, int fft_mode)
{
    ref_sym *h = new ref_sym;
    h->dvbt2.preamble = preamble;
    h->dvbt2.fft_mode = fft_mode;
    h->dvbt2.bandwidth = BANDWIDTH_8_0_MHZ;
    h->dvbt2.miso_group = MISO_TX1;
    dvbt2_p2_parameters_init(h->dvbt2);
    h->fq->init(h->dvbt2);
    h->p2->init(h->dvbt2, h->pilot, h->fq);
    h->dvbt2.guard_interval_size = h->dvbt2.fft_size / 4;
    return h;
}

/* 30 ints, order of put_dvbt2 */
void ref_sym_params(void *hv, int *out) { put_dvbt2(static_cast<ref_sym *>(hv)->dvbt2, out); }

/* p2_symbol::execute on one fft-shifted P2 symbol. out_cells: c_p2 complex. l1_pre_out: 29 ints (struct order).
 * l1_post_out: see pack below. flags[0]=crc32_l1_pre, flags[1]=crc32_l1_post. sync[0]=sample_rate_offset, sync[1]=phase_offset. */
int ref_sym_p2(void *hv, int demod_init, const float *ofdm_cell, float *out_cells, int *l1_pre_out, int *l1_post_out, int *flags,
               float *sync)
{
    ref_sym *h = static_cast<ref_sym *>(hv);
    int idx_symbol = 0;
    bool crc_pre = false, crc_post = false;
    float sro = 0.0f, ph = 0.0f;
    std::vector<complex> in(h->dvbt2.fft_size);
    std::memcpy(in.data(), ofdm_cell, sizeof(complex) * in.size());
    complex *cells = h->p2->execute(h->dvbt2, demod_init != 0, idx_symbol, in.data(), h->l1_pre, h->l1_post, crc_pre, crc_post,
                                    sro, ph);
    std::memcpy(out_cells, cells, sizeof(complex) * h->p2->c_p2);
    flags[0] = crc_pre;
    flags[1] = crc_post;
    sync[0] = sro;
    sync[1] = ph;
    if (crc_pre && l1_pre_out) std::memcpy(l1_pre_out, &h->l1_pre, sizeof(int) * 29);
    if (crc_pre && crc_post && l1_post_out) {
        const l1_postsignalling &p = h->l1_post;
        int *o = l1_post_out;
        *o++ = p.sub_slices_per_frame; *o++ = p.num_plp; *o++ = p.num_aux; *o++ = p.aux_config_rfu;
        *o++ = p.fef_type; *o++ = p.fef_length; *o++ = p.fef_interval; *o++ = p.fef_length_msb; *o++ = p.reserved_2;
        *o++ = p.dyn.frame_idx; *o++ = p.dyn.sub_slice_interval; *o++ = p.dyn.type_2_start; *o++ = p.dyn.l1_change_counter;
        *o++ = p.dyn.start_rf_idx; *o++ = p.dyn.reserved_1; *o++ = p.dyn.reserved_3;
        *o++ = h->l1_pre.num_rf;
        for (int i = 0; i < h->l1_pre.num_rf; ++i) { *o++ = p.rf[i].rf_idx; *o++ = p.rf[i].frequency; }
        for (int i = 0; i < p.num_plp; ++i) {
            std::memcpy(o, &p.plp[i], sizeof(int) * 21);
            o += 21;
            std::memcpy(o, &p.dyn.plp[i], sizeof(int) * 4);
            o += 4;
        }
        for (int i = 0; i < p.num_aux; ++i) { *o++ = p.aux[i].aux_stream_type; *o++ = p.aux[i].aux_private_config; }
        return static_cast<int>(o - l1_post_out);
    }
    return 0;
}

/* the assignments p2_symbol::l1_pre_info makes to dvbt2 (p2_symbol.cpp:493-500), for table fixtures that need no P2 signal */
void ref_sym_set_l1_pre(void *hv, int bwt_ext, int guard_interval, int papr, int pilot_pattern, int num_data_symbols)
{
    ref_sym *h = static_cast<ref_sym *>(hv);
    if (h->dvbt2.carrier_mode != bwt_ext) {
        h->dvbt2.carrier_mode = bwt_ext;
        dvbt2_bwt_ext_parameters_init(h->dvbt2);
    }
    h->dvbt2.guard_interval_mode = guard_interval;
    h->dvbt2.papr_mode = papr;
    h->dvbt2.pilot_pattern = pilot_pattern;
    h->dvbt2.n_data = num_data_symbols;
}

/* dvbt2_demodulator.cpp:386-395 */
int ref_sym_data_init(void *hv)
{
    ref_sym *h = static_cast<ref_sym *>(hv);
    h->data->init(h->dvbt2, h->pilot, h->fq);
    if (h->dvbt2.l_fc) h->fc->init(h->dvbt2, h->pilot, h->fq);
    h->data_ready = true;
    return h->dvbt2.l_fc;
}

int ref_sym_data(void *hv, int idx_symbol, const float *ofdm_cell, float *out_cells, float *sync)
{
    ref_sym *h = static_cast<ref_sym *>(hv);
    std::vector<complex> in(h->dvbt2.fft_size);
    std::memcpy(in.data(), ofdm_cell, sizeof(complex) * in.size());
    float sro = 0.0f, ph = 0.0f;
    complex *cells = h->data->execute(idx_symbol, in.data(), sro, ph);
    std::memcpy(out_cells, cells, sizeof(complex) * h->dvbt2.c_data);
    sync[0] = sro;
    sync[1] = ph;
    return h->dvbt2.c_data;
}

int ref_sym_fc(void *hv, const float *ofdm_cell, float *out_cells, float *sync)
{
    ref_sym *h = static_cast<ref_sym *>(hv);
    std::vector<complex> in(h->dvbt2.fft_size);
    std::memcpy(in.data(), ofdm_cell, sizeof(complex) * in.size());
    float sro = 0.0f, ph = 0.0f;
    complex *cells = h->fc->execute(in.data(), sro, ph);
    std::memcpy(out_cells, cells, sizeof(complex) * h->dvbt2.n_fc);
    sync[0] = sro;
    sync[1] = ph;
    return h->dvbt2.n_fc;
}

/* carrier-type map and pilot reference of symbol idx_symbol: kind 0 = P2 (idx < n_p2), 1 = data (idx_symbol counted as in
 * data_symbol::execute, i.e. frame index incl. P2), 2 = frame closing. k_total entries each. */
int ref_sym_carriers(void *hv, int kind, int idx_symbol, int *map, float *refer)
{
    ref_sym *h = static_cast<ref_sym *>(hv);
    const int k = h->dvbt2.k_total;
    if (kind == 0) {
        std::memcpy(map, h->pilot->p2_carrier_map, sizeof(int) * k);
        std::memcpy(refer, h->pilot->p2_pilot_refer[idx_symbol], sizeof(float) * k);
    } else if (kind == 1) {
        const int idx_data_symbol = idx_symbol - h->dvbt2.n_p2;                     /* data_symbol.cpp:127 */
        std::memcpy(map, h->pilot->data_carrier_map[idx_data_symbol], sizeof(int) * k);
        std::memcpy(refer, h->pilot->data_pilot_refer[idx_data_symbol], sizeof(float) * k);
    } else {
        std::memcpy(map, h->pilot->fc_carrier_map, sizeof(int) * k);
        std::memcpy(refer, h->pilot->fc_pilot_refer, sizeof(float) * k);
    }
    return k;
}

int ref_sym_freq_deint(void *hv, int kind, int *h_even, int *h_odd)
{
    ref_sym *h = static_cast<ref_sym *>(hv);
    const int n = kind == 0 ? h->dvbt2.c_p2 : (kind == 1 ? h->dvbt2.c_data : h->dvbt2.n_fc);
    const int *e = kind == 0 ? h->fq->h_even_p2 : (kind == 1 ? h->fq->h_even_data : h->fq->h_even_fc);
    const int *o = kind == 0 ? h->fq->h_odd_p2 : (kind == 1 ? h->fq->h_odd_data : h->fq->h_odd_fc);
    std::memcpy(h_even, e, sizeof(int) * n);
    std::memcpy(h_odd, o, sizeof(int) * n);
    return n;
}

/* ---- p1_symbol: one instance, fed in caller-chosen pieces ------------------------------------------------------------ */
struct ref_p1 {
    p1_symbol *p1 = new p1_symbol;
    dvbt2_parameters dvbt2{};
    std::vector<complex> buffer_sym = std::vector<complex>(FFT_32K + FFT_32K / 4 + P1_LEN);
};
void *ref_p1_new(void) { return new ref_p1; }

/* res: detected, consume, idx_buffer_sym, p1_decoded, preamble, fft_mode, reset(out); returns detected */
int ref_p1_execute(void *hv, int gain_changed, float level_detect, int len_in, const float *in, int consume, int reset, int *res,
                   double *coarse_freq_offset, float *buffer_sym_out, int buffer_sym_len)
{
    ref_p1 *h = static_cast<ref_p1 *>(hv);
    std::vector<complex> x(len_in);
    std::memcpy(x.data(), in, sizeof(complex) * len_in);
    int idx_buffer_sym = 0;
    bool p1_decoded = false, rst = reset != 0;
    double cfo = *coarse_freq_offset;
    const bool det = h->p1->execute(gain_changed != 0, level_detect, len_in, x.data(), consume, h->buffer_sym.data(), idx_buffer_sym,
                                    h->dvbt2, cfo, p1_decoded, rst);
    res[0] = det; res[1] = consume; res[2] = idx_buffer_sym; res[3] = p1_decoded; res[4] = h->dvbt2.preamble;
    res[5] = h->dvbt2.fft_mode; res[6] = rst;
    *coarse_freq_offset = cfo;
    if (buffer_sym_out) std::memcpy(buffer_sym_out, h->buffer_sym.data(), sizeof(complex) * buffer_sym_len);
    return det;
}

}  // extern "C"
