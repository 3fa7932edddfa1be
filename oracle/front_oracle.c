/*
 * oracle/front_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
 *
 * CPU restatement of the reference's sample-rate front end:
 *   filter_decimator::execute            /root/reference/src/DSP/filter_decimator.h:72-131   (taps :25-38)
 *   interpolator_farrow::operator()      /root/reference/src/DSP/interpolator_farrow.hh:41-68
 *   exponential_averager, PI loop filter /root/reference/src/DSP/loop_filters.hh:20-54,56-73
 *   dvbt2_demodulator::execute front loop, est_1_bit_quantization
 *                                        /root/reference/src/DVB_T2/dvbt2_demodulator.cpp:145-254,256-265
 *   symbol_acquisition: guard-interval correlation and the three tracking updates
 *                                        /root/reference/src/DVB_T2/dvbt2_demodulator.cpp:321-330,429-439
 *
 * Parity status:
 *   decimator, Farrow resampler, exponential averager, PI loop filter -- PINNED: the reference headers are Qt-free and are
 *     compiled unmodified into oracle/_ref/libref_dsp.so (oracle/ref_dsp.cpp); tests/test_oracle_front.py compares.
 *     The decimator is bit-exact (its summation order is fixed by explicit AVX2 intrinsics); the Farrow output count and
 *     phase sequence are exact, its polynomial values agree to float rounding (the reference binary is -Ofast, so its
 *     association of the four-term sum is the compiler's choice).
 *   front loop (dc / iq-imbalance / NCO) and guard correlation -- UNPINNED: dvbt2_demodulator.cpp is a QObject and needs
 *     Qt, which this image lacks. Their building blocks (averager, LUT sin/cos, atan2_approx, loop filters) are pinned.
 *
 * Floating point: built -O2 -ffp-contract=off, every operation in source order.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../sdr_receiver_dvb_t2_amd/csrc/tables/dsp_tables_data.h"

float ora_sin_lut(float x);
float ora_cos_lut(float x);
float ora_atan2_approx(float y, float x);

#define PI_F   3.14159274101257324219f      /* M_PIf32 (glibc) */
#define PI_X_2 (PI_F * 2.0f)                /* dvbt2_definition.h:27 */

/* ---- filter_decimator (filter_decimator.h:41-131) --------------------------------------------------------------------
 * State: the 63 newest input cells (oldest first) and the decimation phase d (the reference keeps d in a function-local
 * static, :77; one decimator exists per process, so it is state of the object here). A result is produced for every second
 * input cell from the window of the 64 newest cells, tap 0 on the oldest (:83-90).
 * Summation order of the AVX2 code (:92-115): the 64 products are taken in 4 blocks of 16 cells; in a block, lane p (0..3)
 * adds ((c[p] + c[4+p]) + (c[8+p] + c[12+p])); the four block values are accumulated in order onto 0; finally
 * ((lane0 + lane1) + lane2) + lane3. Real and imaginary parts separately. */
typedef struct { float hist[63][2]; int d; } ora_decim;

ora_decim *ora_decim_create(void) { return (ora_decim *)calloc(1, sizeof(ora_decim)); }
void ora_decim_destroy(ora_decim *s) { free(s); }
void ora_decim_get_state(const ora_decim *s, float *hist63x2, int *d) { memcpy(hist63x2, s->hist, sizeof s->hist); *d = s->d; }

static float decim_sum(const float (*w)[2], int comp)
{
    float lane[4];
    for (int p = 0; p < 4; p++) {
        float acc = 0.0f;
        for (int blk = 0; blk < 4; blk++) {
            const int c = 16 * blk + p;
            float m0 = w[c][comp] * T2_DECIM_TAPS[c], m1 = w[c + 4][comp] * T2_DECIM_TAPS[c + 4];
            float m2 = w[c + 8][comp] * T2_DECIM_TAPS[c + 8], m3 = w[c + 12][comp] * T2_DECIM_TAPS[c + 12];
            float s0 = m0 + m1, s1 = m2 + m3;
            float st = s0 + s1;
            acc = acc + st;
        }
        lane[p] = acc;
    }
    return ((lane[0] + lane[1]) + lane[2]) + lane[3];
}

/* The same sums eight lanes at a time, as the reference's registers hold them (:92-115): lane 2p + comp of a 256-bit register is
 * component comp of cell p of a group of four; v0..v3 are the four groups of a 16-cell block. gcc's vector extension keeps every
 * operation in this order (no re-association without -ffast-math) and compiles to the AVX2 forms under -mavx2. decim_sum above is
 * the scalar statement of the same order and is what ora_decim_execute falls back to in the CPU tier's cross-check
 * (ora_decim_execute_scalar). */
typedef float ora_v8f __attribute__((vector_size(32)));
static inline ora_v8f ld8(const float *p) { ora_v8f v; memcpy(&v, p, sizeof v); return v; }

int ora_decim_execute_scalar(ora_decim *s, int len_in, const float *in, float *out)
{
    float w[64][2];
    int n_out = 0;
    for (int x = 0; x < len_in; x++) {
        memcpy(w, s->hist, sizeof s->hist);
        w[63][0] = in[2 * x]; w[63][1] = in[2 * x + 1];
        if (++s->d == 2) {
            s->d = 0;
            out[2 * n_out] = decim_sum(w, 0);
            out[2 * n_out + 1] = decim_sum(w, 1);
            n_out++;
        }
        memcpy(s->hist, w + 1, sizeof s->hist);
    }
    return n_out;
}

int ora_decim_execute(ora_decim *s, int len_in, const float *in, float *out)
{
    static float taps2[128] __attribute__((aligned(32)));
    static int taps_ready = 0;
    if (!taps_ready) {
        for (int c = 0; c < 64; c++) taps2[2 * c] = taps2[2 * c + 1] = T2_DECIM_TAPS[c];
        taps_ready = 1;
    }
    /* one contiguous run of cells: the 63 kept ones, then the input (the reference's ring buffer, :80-90, without the wrap) */
    float *buf = (float *)malloc(sizeof(float) * 2 * (size_t)(63 + len_in));
    if (!buf) return -1;
    memcpy(buf, s->hist, sizeof s->hist);
    memcpy(buf + 126, in, sizeof(float) * 2 * (size_t)len_in);
    int n_out = 0;
    for (int x = 0; x < len_in; x++) {
        if (++s->d != 2) continue;
        s->d = 0;
        const float *w = buf + 2 * x;                      /* window of the 64 newest cells, oldest first */
        ora_v8f sum = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 128; i += 32) {
            const ora_v8f m0 = ld8(w + i) * ld8(taps2 + i), m1 = ld8(w + i + 8) * ld8(taps2 + i + 8);
            const ora_v8f m2 = ld8(w + i + 16) * ld8(taps2 + i + 16), m3 = ld8(w + i + 24) * ld8(taps2 + i + 24);
            const ora_v8f sum0 = m0 + m1, sum1 = m2 + m3;
            const ora_v8f sumt = sum0 + sum1;
            sum = sum + sumt;
        }
        out[2 * n_out] = ((sum[0] + sum[2]) + sum[4]) + sum[6];
        out[2 * n_out + 1] = ((sum[1] + sum[3]) + sum[5]) + sum[7];
        n_out++;
    }
    memcpy(s->hist, buf + 2 * (size_t)len_in, sizeof s->hist);
    free(buf);
    return n_out;
}

/* ---- interpolator_farrow<complex,float> (interpolator_farrow.hh:41-68) ------------------------------------------------ */
typedef struct { float d1[2], d2[2], d3[2]; float x1; } ora_farrow;

ora_farrow *ora_farrow_create(void)
{
    ora_farrow *s = (ora_farrow *)calloc(1, sizeof(ora_farrow));
    s->x1 = -0.5f;                                                                 /* start, :34-38 */
    return s;
}
void ora_farrow_destroy(ora_farrow *s) { free(s); }
float ora_farrow_phase(const ora_farrow *s) { return s->x1; }

/* phases, when not NULL, receives the x1 value used for every output sample */
int ora_farrow_execute(ora_farrow *s, int len_in, const float *in, double arbitrary_resample, float *out, float *phases)
{
    const float delay_x = (float)arbitrary_resample;                               /* :45 */
    const float tr = 1.0f + -0.5f;
    int n_out = 0;
    for (int i = 0; i < len_in; i++) {
        float a0[2], a1[2], a2[2], a3[2];
        for (int c = 0; c < 2; c++) {
            const float x = in[2 * i + c];
            const float even1 = s->d3[c] + x, even2 = s->d2[c] + s->d1[c];
            const float odd1 = s->d3[c] - x, odd2 = s->d2[c] - s->d1[c];
            a0[c] = (9.0f / 16.0f) * even2 - (1.0f / 16.0f) * even1;               /* :53 */
            a1[c] = (1.0f / 8.0f) * odd1 - (11.0f / 8.0f) * odd2;
            a2[c] = (1.0f / 4.0f) * (even1 - even2);
            a3[c] = (3.0f / 2.0f) * odd2 - (1.0f / 2.0f) * odd1;
        }
        while (s->x1 < tr) {
            const float x1 = s->x1, x2 = x1 * x1, x3 = x2 * x1;
            for (int c = 0; c < 2; c++) out[2 * n_out + c] = ((a3[c] * x3 + a2[c] * x2) + a1[c] * x1) + a0[c];   /* :60 */
            if (phases) phases[n_out] = x1;
            n_out++;
            s->x1 = s->x1 + delay_x;
        }
        s->x1 = s->x1 - 1.0f;
        for (int c = 0; c < 2; c++) { s->d3[c] = s->d2[c]; s->d2[c] = s->d1[c]; s->d1[c] = in[2 * i + c]; }
    }
    return n_out;
}

/* ---- loop filters (loop_filters.hh) ---------------------------------------------------------------------------------- */
/* exponential_averager<float,float,ratio>::operator() (:63-67) */
float ora_exp_avg(float *state, float ratio, float in)
{
    *state = *state + ratio * (in - *state);
    return *state;
}

/* proportional_integral_loop_filter<float,float,damping,bw_hz,samplerate_hz> (:20-54): constants are evaluated in double
 * and stored to float members (:26-28). */
typedef struct { float k_p, k_i, old_integral; } ora_pi;
void ora_pi_init(ora_pi *s, float damping, int bw_hz, int samplerate_hz)
{
    const double dr = damping;
    const float theta = (float)(((1.0 * bw_hz) / samplerate_hz) / (dr + 1.0 / (4.0 * dr)));
    s->k_p = (float)(4.0 * dr * theta / (1.0 + 2.0 * dr * theta + (double)(theta * theta)));
    s->k_i = (float)(4.0 * theta * theta / (1.0 + 2.0 * dr * theta + (double)(theta * theta)));
    s->old_integral = 0.0f;
}
float ora_pi_step(ora_pi *s, float in_error, float max_integral)                   /* :39-47 */
{
    float integral = s->old_integral + s->k_i * in_error;
    const float out = integral + s->k_p * in_error;
    if (integral > max_integral) integral = max_integral;
    else if (integral < -max_integral) integral = -max_integral;
    s->old_integral = integral;
    return out;
}

/* ---- dvbt2_demodulator::execute, per chunk (dvbt2_demodulator.cpp:165-211) --------------------------------------------- */
typedef struct {
    float short_to_float; int stride;                 /* :31-50 */
    float dc_re, dc_im;                               /* exp_avg_dc_real / _imag, ratio 1e-6 (.h:94-96) */
    float c1, c2;                                     /* .h:98-99 */
    float phase_nco, frequency_nco;                   /* .h:105,111 */
    float level_detect;
} ora_front;

ora_front *ora_front_create(int id_device)
{
    ora_front *s = (ora_front *)calloc(1, sizeof(ora_front));
    s->stride = id_device == 1 ? 2 : 1;
    s->short_to_float = 1.0f / (float)(1 << (id_device == 0 ? 14 : id_device == 1 ? 12 : 11));
    s->c2 = 1.0f;
    s->level_detect = 3.402823466e+38f;
    return s;
}
void ora_front_destroy(ora_front *s) { free(s); }
/* test hook: overwrite the IQ-imbalance coefficients (the reference derives them from float sums of ~1e5 terms; a checker run
 * can be given the device's values so that everything else is compared at full precision) */
void ora_front_set_iq(ora_front *s, float c1, float c2) { s->c1 = c1; s->c2 = c2; }
void ora_front_get_state(const ora_front *s, float *v8)
{
    v8[0] = s->dc_re; v8[1] = s->dc_im; v8[2] = s->c1; v8[3] = s->c2; v8[4] = s->phase_nco; v8[5] = s->frequency_nco;
    v8[6] = s->level_detect; v8[7] = 0.0f;
}

static float wrap_2pi(float a)
{
    while (a > PI_X_2) a -= PI_X_2;
    while (a < -PI_X_2) a += PI_X_2;
    return a;
}

/* One chunk: samples idx_in .. idx_in+chunk-1 of the I/Q arrays; theta[3] accumulates over the whole execute() call. */
void ora_front_chunk(ora_front *s, int chunk, const int16_t *i_in, const int16_t *q_in, int idx_in,
                     float phase_est_filtered, float frequency_est_filtered, float *theta, float *out)
{
    s->phase_nco = wrap_2pi(s->phase_nco + phase_est_filtered);                                    /* :165-171 */
    for (int i = 0; i < chunk; i++) {
        const int j = (i + idx_in) * s->stride;
        float real = (float)i_in[j] * s->short_to_float, imag = (float)q_in[j] * s->short_to_float;
        real -= ora_exp_avg(&s->dc_re, 1.0e-6f, real);                                             /* :180-181 */
        imag -= ora_exp_avg(&s->dc_im, 1.0e-6f, imag);
        float sgn = real < 0 ? -1.0f : 1.0f;                                                       /* :256-265 */
        theta[0] -= imag * sgn;
        theta[1] += real * sgn;
        sgn = imag < 0 ? -1.0f : 1.0f;
        theta[2] += imag * sgn;
        real *= s->c2;                                                                             /* :184-185 */
        imag += s->c1 * real;
        s->frequency_nco = wrap_2pi(s->frequency_nco - frequency_est_filtered);                    /* :187-193 */
        const float offset_nco = wrap_2pi(s->frequency_nco - s->phase_nco);                        /* :194-200 */
        const float nco_real = ora_cos_lut(offset_nco), nco_imag = ora_sin_lut(offset_nco);
        out[2 * i] = real * nco_real - imag * nco_imag;                                            /* :203-204 */
        out[2 * i + 1] = imag * nco_real + real * nco_imag;
    }
}

/* End of execute(): IQ-imbalance coefficients for the next call and the level estimate (:228-234). */
void ora_front_finish(ora_front *s, int len_in, const float *theta)
{
    const float avg1 = theta[0] / len_in, avg2 = theta[1] / len_in, avg3 = theta[2] / len_in;
    s->c1 = avg1 / avg2;
    const float c_temp = avg3 / avg2;
    s->c2 = sqrtf(c_temp * c_temp - s->c1 * s->c1);
    s->level_detect = avg2 * avg3;
}

/* ---- symbol_acquisition pieces ---------------------------------------------------------------------------------------- */
/* Guard-interval correlation of one buffered symbol (guard + fft_size cells), :321-327. sum2 receives the complex sum. */
float ora_cp_frequency_est(const float *sym, int fft_size, int guard, float *sum2)
{
    const float *cp = sym + 2 * (size_t)fft_size;
    float sr = 0.0f, si = 0.0f;
    for (int i = 4; i < guard - 4; i++) {
        const float ar = cp[2 * i], ai = cp[2 * i + 1], br = sym[2 * i], bi = sym[2 * i + 1];
        sr += ar * br + ai * bi;
        si += ai * br - ar * bi;
    }
    if (sum2) { sum2[0] = sr; sum2[1] = si; }
    return ora_atan2_approx(si, sr) / (float)(fft_size << 1);
}

/* The tracking updates after each demodulated symbol (:328-330,429-439). */
typedef struct {
    ora_pi phase, freq;
    float phase_est_filtered, frequency_est_filtered, old_sample_rate_est;
    double sample_rate_est_filtered, resample, max_resample;
} ora_sync;

ora_sync *ora_sync_create(float sample_rate)
{
    ora_sync *s = (ora_sync *)calloc(1, sizeof(ora_sync));
    const float fs = 1.0f / (1.0e-6f * 7.0f / 64.0f);                              /* SAMPLE_RATE, dvbt2_definition.h:29-30 */
    ora_pi_init(&s->phase, 0.3f, 1000000, (int)fs);                               /* .h:106-110 */
    ora_pi_init(&s->freq, 0.7f, 4000000, (int)fs);                                /* .h:112-116 */
    s->resample = sample_rate * (1.0f / (fs * 2));     /* :54, float arithmetic stored to double; as the -Ofast reference binary evaluates the division (by a constant: times its float reciprocal) */
    s->max_resample = s->resample + s->resample * 1.0e-4;
    return s;
}
void ora_sync_destroy(ora_sync *s) { free(s); }
void ora_sync_frequency(ora_sync *s, float frequency_est, int fft_size)          /* :328-330 */
{
    s->frequency_est_filtered += ora_pi_step(&s->freq, frequency_est, 1.0f / (float)fft_size);
}
void ora_sync_symbol(ora_sync *s, float phase_est, float sample_rate_est)         /* :429-439 */
{
    s->phase_est_filtered = ora_pi_step(&s->phase, phase_est * 0.5f, PI_F * 2);
    const double step = 8.0e-9;
    if (s->old_sample_rate_est - sample_rate_est > 0.0f) {
        s->sample_rate_est_filtered -= step;
        if (s->resample - s->sample_rate_est_filtered < -s->max_resample) s->sample_rate_est_filtered += step;
    } else if (s->old_sample_rate_est - sample_rate_est < 0.0f) {
        s->sample_rate_est_filtered += step;
        if (s->resample - s->sample_rate_est_filtered > s->max_resample) s->sample_rate_est_filtered -= step;
    }
    s->old_sample_rate_est = sample_rate_est;
}
/* arbitrary_resample of the next chunk (:157-158) */
double ora_sync_resample(const ora_sync *s)
{
    double r = s->resample - s->sample_rate_est_filtered;
    if (r > s->max_resample) r = s->max_resample;
    return r;
}
void ora_sync_get(const ora_sync *s, double *v4)
{
    v4[0] = s->phase_est_filtered; v4[1] = s->frequency_est_filtered; v4[2] = s->sample_rate_est_filtered; v4[3] = s->resample;
}
