/*
 * oracle/p1_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
 *
 * CPU restatement of the reference's P1 preamble detector:
 *   p1_symbol::p1_symbol, init_p1_randomize     /root/reference/src/DVB_T2/p1_symbol.cpp:23-55
 *   p1_symbol::execute  (C-A-B correlator, thresholds, part-A cut-out, carrier search)   :75-178
 *   p1_symbol::demodulate (DBPSK, descrambling, S1/S2 match)                             :180-298
 *   p1_symbol::reset_buffer                                                              :300-311
 *   delay_buffer / sum_of_buffer / save_buffer   /root/reference/src/DSP/buffers.hh:21-109
 *
 * Parity status: UNPINNED as a whole (p1_symbol is a QObject; Qt is not in this image). The delay and running-sum classes it
 * is built from are PINNED (tests/test_oracle_front.py drives the reference's buffers.hh in oracle/_ref/libref_dsp.so
 * against the restated versions here). The 1K FFT is FFTW3f in the reference (binary dependency); here a float64 DFT rounded
 * to float -- the DBPSK decisions do not depend on the last bits. The carrier list and S1/S2 patterns are the generated data
 * of csrc/tables/dsp_tables_data.h.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../sdr_receiver_dvb_t2_amd/csrc/tables/dsp_tables_data.h"

float ora_atan2_approx(float y, float x);

#define PI_F   3.14159274101257324219f
#define PI_X_2 (PI_F * 2.0f)
#define P1_C 542
#define P1_B 482
#define P1_A 1024
#define P1_LEN 2048

typedef struct { float re, im; } cf;
static cf cmul(cf a, cf b) { cf r = {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; return r; }
static cf cconj(cf a) { cf r = {a.re, -a.im}; return r; }

/* buffers.hh:61-80 delay_buffer<T,DELAY>: len = DELAY + 1 */
typedef struct { cf *buf; int len; unsigned idx; } delay_t;
static cf delay_run(delay_t *d, cf in) { d->buf[d->idx++] = in; d->idx %= (unsigned)d->len; return d->buf[d->idx]; }
/* buffers.hh:21-48 sum_of_buffer<T,LEN>: sum = sum - buffer[idx] + in, after the write and the index step */
typedef struct { cf *buf; int len; int idx; cf sum; } rsum_t;
static cf rsum_run(rsum_t *s, cf in)
{
    s->buf[s->idx++] = in; s->idx %= s->len;
    s->sum.re = s->sum.re - s->buf[s->idx].re + in.re;
    s->sum.im = s->sum.im - s->buf[s->idx].im + in.im;
    return s->sum;
}

typedef struct {
    cf fq_shift[1024]; int idx_fq_shift;                 /* :26-32, static index :81 */
    delay_t delay_b, delay_c, delay_b_x2, delay_2;       /* p1_symbol.h:53-56 */
    rsum_t average_b, average_c;                         /* :57-58 */
    cf p1_buffer[P1_LEN]; int p1_idx;                    /* save_buffer<complex, P1_LEN> :59 */
    float correlation, begin_threshold, end_threshold, max_correlation;
    int correlation_detect, idx_buffer, p1_decoded;
    cf arg_max;
    int p1_randomize[384];
} ora_p1;

static void delay_init(delay_t *d, int delay) { d->len = delay + 1; d->buf = (cf *)calloc((size_t)d->len, sizeof(cf)); d->idx = 0; }
static void rsum_init(rsum_t *s, int len) { s->len = len; s->buf = (cf *)calloc((size_t)len, sizeof(cf)); s->idx = 0; s->sum.re = s->sum.im = 0; }
static void delay_reset(delay_t *d) { d->idx = 0; memset(d->buf, 0, sizeof(cf) * (size_t)d->len); }
static void rsum_reset(rsum_t *s) { s->idx = 0; s->sum.re = s->sum.im = 0; memset(s->buf, 0, sizeof(cf) * (size_t)s->len); }

static void reset_buffer(ora_p1 *p)                      /* :300-311 */
{
    p->correlation_detect = 0; p->max_correlation = 0.0f;
    delay_reset(&p->delay_c); delay_reset(&p->delay_b); delay_reset(&p->delay_b_x2); delay_reset(&p->delay_2);
    rsum_reset(&p->average_b); rsum_reset(&p->average_c);
    p->idx_buffer = 0;
}

ora_p1 *ora_p1_create(void)
{
    ora_p1 *p = (ora_p1 *)calloc(1, sizeof(ora_p1));
    const float angle_shift = PI_X_2 / 1024.0f;
    float angle = 0.0f;
    for (int i = 0; i < 1024; i++) { p->fq_shift[i].re = sinf(angle); p->fq_shift[i].im = cosf(angle); angle += angle_shift; }
    delay_init(&p->delay_b, P1_B); delay_init(&p->delay_c, P1_C); delay_init(&p->delay_b_x2, 2 * P1_B); delay_init(&p->delay_2, 2);
    rsum_init(&p->average_b, P1_B); rsum_init(&p->average_c, P1_C);
    p->begin_threshold = 5.0e+5f; p->end_threshold = p->begin_threshold * 0.5f;          /* p1_symbol.h:63-64 */
    int sr = 0x4e46;                                                                      /* :45-55 */
    for (int i = 0; i < 384; i++) {
        const int b = (sr ^ (sr >> 1)) & 1;
        p->p1_randomize[i] = b == 0 ? 1 : -1;
        sr >>= 1;
        if (b > 0) sr |= 0x4000;
    }
    return p;
}
void ora_p1_destroy(ora_p1 *p)
{
    if (!p) return;
    free(p->delay_b.buf); free(p->delay_c.buf); free(p->delay_b_x2.buf); free(p->delay_2.buf); free(p->average_b.buf); free(p->average_c.buf);
    free(p);
}
const float *ora_p1_fq_shift(const ora_p1 *p) { return &p->fq_shift[0].re; }
void ora_p1_randomize(const ora_p1 *p, int *out384) { memcpy(out384, p->p1_randomize, sizeof p->p1_randomize); }

/* :180-298. p1: 1024 fft-shifted carriers offset by `shift`. Returns 1 and sets preamble / fft_mode / s1 / s2 when S1 is repeated
 * correctly and the fields are valid. */
static int demodulate(const ora_p1 *p, const cf *p1, int *preamble, int *fft_mode, int *s1_out, int *s2_out)
{
    cf dbpsk[384];
    for (int i = 0; i < 384; i++) { dbpsk[i].re = p1[T2_P1_ACTIVE_CARRIERS[i]].re * 0.1f; dbpsk[i].im = p1[T2_P1_ACTIVE_CARRIERS[i]].im * 0.1f; }
    int dbpsk_bit[384], old_bit = -1;
    dbpsk_bit[0] = old_bit;
    for (int i = 1; i < 384; i++) {
        const cf dif = cmul(dbpsk[i], cconj(dbpsk[i - 1]));
        const float angle = ora_atan2_approx(dif.im, dif.re);
        dbpsk_bit[i] = fabsf(angle) > (PI_F / 2.0f) ? -old_bit : old_bit;
        old_bit = dbpsk_bit[i];
        dbpsk_bit[i] *= p->p1_randomize[i];
    }
    dbpsk_bit[0] *= p->p1_randomize[0];
    old_bit = 1;
    uint8_t data[48] = {0};
    int idx_data = 0, next_bit = 0;
    for (int i = 0; i < 384; i++) {
        const int bit = dbpsk_bit[i] == old_bit ? 0 : 1;
        old_bit = dbpsk_bit[i];
        if (next_bit == 8) { next_bit = 0; ++idx_data; data[idx_data] = 0; }
        data[idx_data] = (uint8_t)((data[idx_data] << 1) + bit);
        ++next_bit;
    }
    int s1 = 0, s2 = 0;
    for (int i = 0; i < 8; i++) {
        if (data[i] != data[i + 40]) return 0;
        if (data[0] == T2_P1_S1_PATTERNS[i][0]) s1 = i;
    }
    for (int i = 0; i < 16; i++)
        if (data[8] == T2_P1_S2_PATTERNS[i][0] && data[9] == T2_P1_S2_PATTERNS[i][1]) s2 = i;
    if (s1 > 4) return 0;                                /* :233-252 (T2_SISO, T2_MISO, NON_T2, T2_LITE_SISO, T2_LITE_MISO) */
    *preamble = s1;
    *fft_mode = s2 >> 1;                                 /* :254-286: the enum values follow the S2 field 1 code */
    *s1_out = s1; *s2_out = s2;
    return 1;
}

static void dft1k_shifted(const cf *in, cf *out)         /* fast_fourier_transform::execute on 1024 points (FFTW forward + half swap) */
{
    static double tw[1024][2];
    static int ready = 0;
    if (!ready) { for (int k = 0; k < 1024; k++) { tw[k][0] = cos(-2.0 * M_PI * k / 1024.0); tw[k][1] = sin(-2.0 * M_PI * k / 1024.0); } ready = 1; }
    for (int k = 0; k < 1024; k++) {
        double sr = 0.0, si = 0.0;
        for (int n = 0; n < 1024; n++) {
            const int t = (k * n) & 1023;
            sr += in[n].re * tw[t][0] - in[n].im * tw[t][1];
            si += in[n].re * tw[t][1] + in[n].im * tw[t][0];
        }
        out[(k + 512) & 1023].re = (float)sr; out[(k + 512) & 1023].im = (float)si;
    }
}

/* p1_symbol::execute (:75-178). in: re/im pairs; *consume in/out; buffer_sym: caller's symbol buffer (>= P1_LEN + 1 cells).
 * res9 (on detection): {idx_buffer_sym, p1_decoded, preamble, fft_mode, s1, s2, shift, 0, 0}; coarse: Hz. corr_trace (optional,
 * len_in floats) receives the correlation value computed for every sample that went through the correlator (others untouched).
 * Returns p1_detect. */
int ora_p1_execute(ora_p1 *p, int gain_changed, float level_detect, int len_in, const float *in, int *consume, float *buffer_sym,
                   int *res9, double *coarse, int reset_flag, float *corr_trace, float *p1_fft_out)
{
    const cf *x = (const cf *)in;
    cf *bsym = (cf *)buffer_sym;
    int idx_in = *consume, p1_detect = 0;
    if (gain_changed) { p->begin_threshold = level_detect * 2.0e+5f; p->end_threshold = 0.5f * p->begin_threshold; }   /* :88-91 */
    while (idx_in < len_in) {
        const cf data = x[idx_in++];
        p->p1_buffer[p->p1_idx++] = data; p->p1_idx %= P1_LEN;                           /* p1_buffer.write */
        if (p->correlation_detect) {
            bsym[p->idx_buffer] = data;
            if (++p->idx_buffer > P1_LEN) reset_buffer(p);
            if (p->correlation < p->end_threshold) {
                p1_detect = 1;
                res9[0] = p->idx_buffer;
                cf lin[P1_LEN], a_part[P1_A], fft[P1_A];
                for (int k = 0; k < P1_LEN; k++) lin[k] = p->p1_buffer[(p->p1_idx + k) % P1_LEN];       /* save_buffer::read */
                memcpy(a_part, &lin[P1_C - p->idx_buffer], sizeof a_part);               /* :111-112 */
                memset(p->p1_buffer, 0, sizeof p->p1_buffer); p->p1_idx = 0;             /* p1_buffer.reset() */
                dft1k_shifted(a_part, fft);
                if (p1_fft_out) memcpy(p1_fft_out, fft, sizeof fft);
                const float hz_per_rad = ((1.0f / (1.0e-6f * 7.0f / 64.0f)) / PI_X_2) / (float)(P1_LEN << 1);
                double cfo = ora_atan2_approx(p->arg_max.im, p->arg_max.re) * (double)hz_per_rad;       /* :115 */
                res9[6] = -1;
                if (!p->p1_decoded || reset_flag) {
                    for (int shift = 76; shift < 96; shift++) {
                        if (demodulate(p, fft + shift, &res9[2], &res9[3], &res9[4], &res9[5])) {
                            p->p1_decoded = 1;
                            res9[6] = shift;
                            if (shift != 86) cfo += (double)(shift - 86) * (double)((1.0f / (1.0e-6f * 7.0f / 64.0f)) / 1024.0f);
                            break;
                        }
                    }
                }
                res9[1] = p->p1_decoded;
                *coarse = cfo;
                reset_buffer(p);
                break;
            }
        }
        const cf data_shift = cmul(data, p->fq_shift[p->idx_fq_shift++]);                /* :149-160 */
        p->idx_fq_shift &= 0x3FF;
        const cf c = delay_run(&p->delay_c, data_shift);
        const cf in_av_c = cmul(data, cconj(c));
        const cf b = delay_run(&p->delay_b, data);
        const cf in_av_b = cmul(data_shift, cconj(b));
        const cf out_av_c = rsum_run(&p->average_c, in_av_c);
        const cf out_av_b = rsum_run(&p->average_b, in_av_b);
        const cf a = delay_run(&p->delay_b_x2, out_av_c);
        const cf d = delay_run(&p->delay_2, out_av_b);
        const cf out = cmul(a, d);
        p->correlation = out.re * out.re + out.im * out.im;
        if (corr_trace) corr_trace[idx_in - 1] = p->correlation;
        if (p->correlation > p->begin_threshold) {                                       /* :163-170 */
            p->correlation_detect = 1;
            if (p->correlation > p->max_correlation) { p->max_correlation = p->correlation; p->arg_max = out; p->idx_buffer = 0; }
        }
    }
    *consume = idx_in;
    return p1_detect;
}

/* pinning helpers: the restated delay / running sum on a whole array (compared with the reference classes) */
void ora_delay_run(int delay, int n, const float *in, float *out)
{
    delay_t d; delay_init(&d, delay);
    for (int i = 0; i < n; i++) ((cf *)out)[i] = delay_run(&d, ((const cf *)in)[i]);
    free(d.buf);
}
void ora_sum_run(int len, int n, const float *in, float *out)
{
    rsum_t s; rsum_init(&s, len);
    for (int i = 0; i < n; i++) ((cf *)out)[i] = rsum_run(&s, ((const cf *)in)[i]);
    free(s.buf);
}
