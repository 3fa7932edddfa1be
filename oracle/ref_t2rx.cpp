/*
 * oracle/ref_t2rx.cpp -- TEST INFRASTRUCTURE ONLY. Thin extern "C" driver that INSTANTIATES AND RUNS the reference's own
 * receiver objects where they lie under /root/reference/src/DVB_T2 (all of *.cpp + LDPC/*.hh compiled unmodified by
 * oracle/Makefile: real Qt 5.9.7 headers, moc and libraries from /opt/conda, the FFTW binary the reference ships):
 *   - the FEC chain exactly as the reference wires it: time_deinterleaver -> llr_demapper -> ldpc_decoder -> bch_decoder ->
 *     bb_de_header, each on its own QThread, hand-shaking through the reference's mutex / wait-condition pairs
 *     (time_deinterleaver.cpp:20-37, llr_demapper.cpp:81-94, ldpc_decoder.cpp ctor, bch_decoder.cpp:29-42);
 *   - the whole receiver: dvbt2_demodulator::execute(len, i, q, signal_estimate*) from int16 I/Q to the TS file.
 * The driver only (i) calls public slots / methods, (ii) listens to the stages' own signals with direct connections to copy
 * what they emit, (iii) reads (never writes, except where a comment says so for a test set-up) private state through
 * `#define private public` in this one translation unit. No reference source is copied, no stand-in header is written.
 * Output: oracle/_ref/libref_t2rx.so (git-ignored). Used by tests/golden/make_t2rx_golden.py to produce fixtures and by the CPU
 * tier for live comparisons in this container. Nothing in the product links or loads it.
 */
#include <QtCore/QCoreApplication>
#include <QtCore/QObject>
#include <QtCore/QThread>
#include <QtCore/QMutex>
#include <QtCore/QWaitCondition>
#include <QtCore/QString>
#include <QtCore/QFile>
#include <QtNetwork/QUdpSocket>
#include <cstring>
#include <complex>
#include <deque>
#include <mutex>
#include <string>
#include <vector>

#define private public
#include "dvbt2_definition.h"
#include "dvbt2_demodulator.h"
#undef private

namespace {

QCoreApplication *g_app = nullptr;
void qt_once()
{
    if (!QCoreApplication::instance()) {
        static int argc = 1;
        static char arg0[] = "ref_t2rx";
        static char *argv[] = { arg0, nullptr };
        g_app = new QCoreApplication(argc, argv);
    }
    /* the reference registers l1_postsignalling for its queued connections inside dvbt2_p2_parameters_init
     * (dvbt2_definition.cpp:22), which init_dvbt2 runs before any stage emits; a stand-alone FEC chain needs the same call */
    static bool registered = false;
    if (!registered) {
        dvbt2_parameters d{};
        d.preamble = T2_SISO;
        d.fft_mode = FFTSIZE_32K;
        dvbt2_p2_parameters_init(d);
        registered = true;
    }
}

struct tap_item {
    int meta[4];
    std::vector<unsigned char> bytes;
};

/* what the stages emitted, in emission order. 0: ti_block cells, 1: 32-frame LLR batch, 2: LDPC hard bits of a batch,
 * 3: descrambled BBFRAME, 4: cells handed to the time de-interleaver (data / l1_dyn_execute), 5: ts_stage messages */
struct taps {
    std::mutex m;
    std::deque<tap_item> q[6];
    bool keep[6] = { true, true, true, true, true, true };
    void push(int which, int m0, int m1, int m2, int m3, const void *p, size_t n)
    {
        if (!keep[which]) return;
        std::lock_guard<std::mutex> g(m);
        q[which].emplace_back();
        tap_item &t = q[which].back();
        t.meta[0] = m0; t.meta[1] = m1; t.meta[2] = m2; t.meta[3] = m3;
        t.bytes.assign(static_cast<const unsigned char *>(p), static_cast<const unsigned char *>(p) + n);
    }
};

void wire_fec_taps(time_deinterleaver *ti, taps *t)
{
    QObject::connect(ti, &time_deinterleaver::ti_block, ti,
                     [t](int size, complex *cells, int plp, l1_postsignalling) {
                         t->push(0, size, plp, 0, 0, cells, sizeof(complex) * size);
                     }, Qt::DirectConnection);
    llr_demapper *qam = ti->qam;
    QObject::connect(qam, &llr_demapper::soft_multiplexer_de_twist, qam,
                     [t](int *plp_simd, l1_postsignalling, int len, int8_t *out) {
                         t->push(1, len, plp_simd[0], plp_simd[SIZEOF_SIMD - 1], 0, out, len);
                     }, Qt::DirectConnection);
    ldpc_decoder *ld = qam->decoder;
    QObject::connect(ld, &ldpc_decoder::bit_bch, ld,
                     [t](int *plp_simd, l1_postsignalling, int len, uint8_t *out) {
                         t->push(2, len, plp_simd[0], 0, 0, out, len);
                     }, Qt::DirectConnection);
    bch_decoder *bd = ld->decoder;
    QObject::connect(bd, &bch_decoder::bit_descramble, bd,
                     [t](int plp, l1_postsignalling, int len, uint8_t *out) { t->push(3, len, plp, 0, 0, out, len); },
                     Qt::DirectConnection);
    QObject::connect(bd->deheader, &bb_de_header::ts_stage, bd->deheader,
                     [t](QString s) {
                         const QByteArray b = s.toUtf8();
                         t->push(5, b.size(), 0, 0, 0, b.constData(), b.size());
                     }, Qt::DirectConnection);
}

/* l1_postsignalling from ints: [num_plp, then per PLP 21 configurable ints (struct order) + 4 dynamic ints (id, start,
 * num_blocks, reserved_2)], dyn.frame_idx last. The arrays are kept alive by the holder. */
struct l1_holder {
    l1_postsignalling post;
    std::vector<l1_postsignalling_plp> plp;
    std::vector<dynamic_plp> dyn;
    void set(const int *v)
    {
        const int n = v[0];
        plp.resize(n);
        dyn.resize(n);
        const int *p = v + 1;
        for (int i = 0; i < n; ++i) {
            std::memcpy(&plp[i], p, sizeof(int) * 21);
            p += 21;
            std::memcpy(&dyn[i], p, sizeof(int) * 4);
            p += 4;
        }
        post = l1_postsignalling();
        post.num_plp = n;
        post.plp = plp.data();
        post.dyn.plp = dyn.data();
        post.dyn.frame_idx = *p;
    }
};

struct ref_fec {
    QMutex *mutex = new QMutex;
    QWaitCondition *cond = new QWaitCondition;
    time_deinterleaver *ti = nullptr;
    taps t;
    l1_holder l1;
    std::vector<complex> cells;
};

struct ref_rx {
    dvbt2_demodulator *dem = nullptr;
    taps t;
    signal_estimate sig;
    std::vector<double> traj;          /* per symbol that reaches the tracking loops (dvbt2_demodulator.cpp:429-444): TRAJ_W doubles */
};
constexpr int TRAJ_W = 16;

int tap_get(taps &t, int which, int *meta, void *out, int cap)
{
    std::lock_guard<std::mutex> g(t.m);
    if (t.q[which].empty()) return -1;
    tap_item &it = t.q[which].front();
    const int n = static_cast<int>(it.bytes.size());
    if (meta) std::memcpy(meta, it.meta, sizeof it.meta);
    if (!out) return n;                                     /* peek */
    if (n > cap) return -2;
    std::memcpy(out, it.bytes.data(), n);
    t.q[which].pop_front();
    return n;
}

void set_ts_file(bb_de_header *dh, const char *path, int need_plp)
{
    /* bb_de_header::set_out is the slot main_window drives (main_window.cpp:318-320); called before any frame flows */
    dh->set_out(bb_de_header::out_file, 7654, QString::fromUtf8(path), need_plp);
}
void close_ts_file(bb_de_header *dh)
{
    if (dh->file != nullptr && dh->file->isOpen()) dh->file->close();           /* what ~bb_de_header does (:45-52) */
}

}  // namespace

extern "C" {

/* ---- FEC chain from the time de-interleaver down ------------------------------------------------------------------------ */
void *ref_fec_new(const char *ts_path, int need_plp)
{
    qt_once();
    ref_fec *h = new ref_fec;
    h->ti = new time_deinterleaver(h->cond, h->mutex);
    wire_fec_taps(h->ti, &h->t);
    set_ts_file(h->ti->qam->decoder->decoder->deheader, ts_path, need_plp);
    return h;
}
void ref_fec_keep(void *hv, int which, int keep) { static_cast<ref_fec *>(hv)->t.keep[which] = keep != 0; }

/* time_deinterleaver::start (dvbt2_demodulator.cpp:366): only l1_pre.l1_post_size and l1_post are read there */
void ref_fec_start(void *hv, int l1_post_size, const int *l1_post)
{
    ref_fec *h = static_cast<ref_fec *>(hv);
    h->l1.set(l1_post);
    dvbt2_parameters dv{};
    l1_presignalling pre;
    pre.l1_post_size = l1_post_size;
    h->ti->start(dv, pre, h->l1.post);
}
/* first symbol of a frame: l1_dyn_execute(l1_post, c_p2, P2 cells incl. the L1 cells in front) (:370) */
void ref_fec_frame(void *hv, const int *l1_post, int len, const float *cells)
{
    ref_fec *h = static_cast<ref_fec *>(hv);
    h->l1.set(l1_post);
    h->cells.resize(len);
    std::memcpy(h->cells.data(), cells, sizeof(complex) * len);
    h->ti->l1_dyn_execute(h->l1.post, len, h->cells.data());
}
/* data / frame-closing symbols: execute(c_data, cells) (:341, :353) */
void ref_fec_cells(void *hv, int len, const float *cells)
{
    ref_fec *h = static_cast<ref_fec *>(hv);
    h->cells.resize(len);
    std::memcpy(h->cells.data(), cells, sizeof(complex) * len);
    h->ti->execute(len, h->cells.data());
}
/* the LDPC stage's public slot by itself: ldpc_decoder::execute(idx_plp_simd, l1_post, len_in = fec_size * 32, int8 LLRs)
 * (ldpc_decoder.h:90, .cpp:157-301) on the decoder object of this chain -- its bit_bch signal then drives bch_decoder and
 * bb_de_header exactly as when the demapper is the caller (taps 2, 3, 5 and the TS file). This is how 256-QAM payload bits get a
 * reference-built pin although the reference's own demapper never produces a decodable 256-QAM batch. */
void ref_fec_ldpc_execute(void *hv, const int *l1_post, const int *idx_plp_simd, int len, const int8_t *llr)
{
    ref_fec *h = static_cast<ref_fec *>(hv);
    h->l1.set(l1_post);
    int plp[SIZEOF_SIMD];
    std::memcpy(plp, idx_plp_simd, sizeof plp);
    std::vector<int8_t> in(llr, llr + len);
    h->ti->qam->decoder->execute(plp, h->l1.post, len, in.data());
}
int ref_fec_tap(void *hv, int which, int *meta, void *out, int cap) { return tap_get(static_cast<ref_fec *>(hv)->t, which, meta, out, cap); }
void ref_fec_close(void *hv) { close_ts_file(static_cast<ref_fec *>(hv)->ti->qam->decoder->decoder->deheader); }

/* ---- bb_de_header alone: BBFRAME bits in, TS file out ----------------------------------------------------------------------- */
struct ref_bbdh {
    QMutex *mutex = new QMutex;
    QWaitCondition *cond = new QWaitCondition;
    bb_de_header *dh = nullptr;
    taps t;
    l1_holder l1;
};
void *ref_bbdh_new(const char *ts_path, int need_plp, int num_plp)
{
    qt_once();
    ref_bbdh *h = new ref_bbdh;
    h->dh = new bb_de_header(h->cond, h->mutex);
    QObject::connect(h->dh, &bb_de_header::ts_stage, h->dh,
                     [h](QString s) {
                         const QByteArray b = s.toUtf8();
                         h->t.push(5, b.size(), 0, 0, 0, b.constData(), b.size());
                     }, Qt::DirectConnection);
    set_ts_file(h->dh, ts_path, need_plp);
    std::vector<int> v(1 + 25 * num_plp + 1, 0);
    v[0] = num_plp;
    h->l1.set(v.data());
    return h;
}
void ref_bbdh_execute(void *hv, int plp_id, int len, const unsigned char *bits)
{
    ref_bbdh *h = static_cast<ref_bbdh *>(hv);
    std::vector<uint8_t> in(bits, bits + len);
    in.resize(len + 65536 + 4096);                       /* the reference reads past the frame on bad SYNCD; keep that in bounds */
    h->dh->execute(plp_id, h->l1.post, len, in.data());
}
int ref_bbdh_tap(void *hv, int which, int *meta, void *out, int cap) { return tap_get(static_cast<ref_bbdh *>(hv)->t, which, meta, out, cap); }
void ref_bbdh_close(void *hv) { close_ts_file(static_cast<ref_bbdh *>(hv)->dh); }

/* ---- the whole receiver ------------------------------------------------------------------------------------------------- */
void *ref_rx_new(int id_device, float sample_rate, const char *ts_path, int need_plp)
{
    qt_once();
    ref_rx *h = new ref_rx;
    h->dem = new dvbt2_demodulator(static_cast<id_device_t>(id_device), sample_rate);
    wire_fec_taps(h->dem->deinterleaver, &h->t);
    QObject::connect(h->dem, &dvbt2_demodulator::data, h->dem,
                     [h](int len, complex *c) { h->t.push(4, len, h->dem->idx_symbol, 0, 0, c, sizeof(complex) * len); },
                     Qt::DirectConnection);
    QObject::connect(h->dem, &dvbt2_demodulator::l1_dyn_execute, h->dem,
                     [h](l1_postsignalling, int len, complex *c) { h->t.push(4, len, 0, 1, 0, c, sizeof(complex) * len); },
                     Qt::DirectConnection);
    /* the loop trajectory: replace_null_indicator is emitted at the end of every pass through the tracking loops (:444), i.e. once per
     * P2 / data / frame-closing symbol of a tracked frame, behind the filters' update. Private members read, nothing written:
     * next_symbol_type, idx_symbol (both already advanced), chunk (the chunk that completed the symbol), phase_est_filtered,
     * frequency_est_filtered, sample_rate_est_filtered, resample, phase_nco, frequency_nco, old_sample_rate_est, and the signal's two
     * arguments (sample-rate and frequency offsets in Hz as the GUI gets them), the loop filters' integrators and gains */
    QObject::connect(h->dem, &dvbt2_demodulator::replace_null_indicator, h->dem,
                     [h](const float b1, const float b2) {
                         dvbt2_demodulator *d = h->dem;
                         const double v[TRAJ_W] = { double(d->next_symbol_type), double(d->idx_symbol), double(d->chunk), double(d->phase_est_filtered),
                                                    double(d->frequency_est_filtered), d->sample_rate_est_filtered, d->resample, double(d->phase_nco),
                                                    double(d->frequency_nco), double(d->old_sample_rate_est), double(b1), double(b2),
                                                    /* the two PI filters' integrators and proportional gains (DSP/loop_filters.hh:28-33): a symbol's raw
                                                     * phase_est follows from them as 2 (phase_est_filtered - integral) / k_p while the integrator is inside its clamp */
                                                    double(d->loop_filter_phase_offset.old_integral_error), double(d->loop_filter_phase_offset.k_p),
                                                    double(d->loop_filter_frequency_offset.old_integral_error), double(d->loop_filter_frequency_offset.k_p) };
                         h->traj.insert(h->traj.end(), v, v + TRAJ_W);
                     }, Qt::DirectConnection);
    set_ts_file(h->dem->deinterleaver->qam->decoder->decoder->deheader, ts_path, need_plp);
    return h;
}
/* the trajectory so far: returns the number of records (TRAJ_W doubles each); out may be NULL (count only) */
int ref_rx_traj(void *hv, double *out, int cap_records)
{
    ref_rx *h = static_cast<ref_rx *>(hv);
    const int n = static_cast<int>(h->traj.size() / TRAJ_W);
    if (out) std::memcpy(out, h->traj.data(), sizeof(double) * TRAJ_W * (n < cap_records ? n : cap_records));
    return n;
}
void ref_rx_keep(void *hv, int which, int keep) { static_cast<ref_rx *>(hv)->t.keep[which] = keep != 0; }

/* sig (in/out, doubles): change_frequency, coarse_freq_offset, frequency_changed, change_gain, gain_offset, gain_changed,
 * correct_resample, reset, p1_reset -- the fields of signal_estimate (dvbt2_demodulator.h:43-53) */
void ref_rx_execute(void *hv, int len, const int16_t *i_in, const int16_t *q_in, double *sig)
{
    ref_rx *h = static_cast<ref_rx *>(hv);
    signal_estimate &s = h->sig;
    s.change_frequency = sig[0] != 0; s.coarse_freq_offset = sig[1]; s.frequency_changed = sig[2] != 0; s.change_gain = sig[3] != 0;
    s.gain_offset = static_cast<int>(sig[4]); s.gain_changed = sig[5] != 0; s.correct_resample = sig[6]; s.reset = sig[7] != 0;
    s.p1_reset = sig[8] != 0;
    h->dem->execute(len, const_cast<int16_t *>(i_in), const_cast<int16_t *>(q_in), &s);
    sig[0] = s.change_frequency; sig[1] = s.coarse_freq_offset; sig[2] = s.frequency_changed; sig[3] = s.change_gain;
    sig[4] = s.gain_offset; sig[5] = s.gain_changed; sig[6] = s.correct_resample; sig[7] = s.reset; sig[8] = s.p1_reset;
}

/* private state after a call (doubles): c1, c2, level_detect, phase_nco, frequency_nco, phase_est_filtered,
 * frequency_est_filtered, sample_rate_est_filtered, resample, next_symbol_type, idx_symbol, symbol_size, idx_buffer_sym,
 * est_chunk, chunk(last), p2_init, demodulator_init, deint_start, crc32_l1_pre, guard_interval_size, fft_size, dc_re, dc_im */
void ref_rx_state(void *hv, double *o)
{
    dvbt2_demodulator *d = static_cast<ref_rx *>(hv)->dem;
    o[0] = d->c1; o[1] = d->c2; o[2] = d->level_detect; o[3] = d->phase_nco; o[4] = d->frequency_nco; o[5] = d->phase_est_filtered;
    o[6] = d->frequency_est_filtered; o[7] = d->sample_rate_est_filtered; o[8] = d->resample; o[9] = d->next_symbol_type;
    o[10] = d->idx_symbol; o[11] = d->symbol_size; o[12] = d->idx_buffer_sym; o[13] = d->est_chunk; o[14] = d->chunk;
    o[15] = d->p2_init; o[16] = d->demodulator_init; o[17] = d->deint_start; o[18] = d->crc32_l1_pre;
    o[19] = d->dvbt2.guard_interval_size; o[20] = d->dvbt2.fft_size; o[21] = d->exp_avg_dc_real.out; o[22] = d->exp_avg_dc_imag.out;
}
/* test set-up only: start a front-loop run from chosen loop values (what the tracking loops would have left there) */
void ref_rx_set_loops(void *hv, float c1, float c2, float phase_est_filtered, float frequency_est_filtered, float phase_nco,
                      float frequency_nco)
{
    dvbt2_demodulator *d = static_cast<ref_rx *>(hv)->dem;
    d->c1 = c1; d->c2 = c2; d->phase_est_filtered = phase_est_filtered; d->frequency_est_filtered = frequency_est_filtered;
    d->phase_nco = phase_nco; d->frequency_nco = frequency_nco;
}
/* buffers of the LAST chunk of the last execute(): which 0 = out_derotate_sample (chunk samples), 1 = out_decimator (n) */
int ref_rx_buffer(void *hv, int which, int n, float *out)
{
    dvbt2_demodulator *d = static_cast<ref_rx *>(hv)->dem;
    std::memcpy(out, which == 0 ? d->out_derotate_sample : d->out_decimator, sizeof(complex) * n);
    return n;
}
int ref_rx_tap(void *hv, int which, int *meta, void *out, int cap) { return tap_get(static_cast<ref_rx *>(hv)->t, which, meta, out, cap); }
void ref_rx_close(void *hv) { close_ts_file(static_cast<ref_rx *>(hv)->dem->deinterleaver->qam->decoder->decoder->deheader); }

}  // extern "C"
