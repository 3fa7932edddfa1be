/*
 * oracle/ref_dsp.cpp -- TEST INFRASTRUCTURE ONLY. Driver around the REFERENCE's own Qt-free DSP header
 * /root/reference/src/DSP/fast_math.h (LUT sin/cos and atan2_approx), compiled where it lies (-I in oracle/Makefile) with
 * the reference's own optimisation flags (-Ofast -mavx2, sdr_receiver_dvb_t2.pro:33-39). No reference source is copied.
 * Output: oracle/_ref/libref_dsp.so (git-ignored, travels to the GPU box).
 */
#include "fast_math.h"

extern "C" void ref_lut_init(void) { table_sin_cos_instance.table_(); }
extern "C" float ref_sin_lut(float x) { return sin_lut(x); }
extern "C" float ref_cos_lut(float x) { return cos_lut(x); }
extern "C" float ref_atan2_approx(float y, float x) { return atan2_approx(y, x); }
extern "C" const float *ref_lut_table(int which) { return which ? look_up_table_cos : look_up_table_sin; }

/* ---- Qt-free front-end classes of the reference, used as they are --------------------------------------------------- */
#include "filter_decimator.h"
#include "interpolator_farrow.hh"
#include "loop_filters.hh"
#include "buffers.hh"

/* filter_decimator keeps its decimation phase in a function-local static (filter_decimator.h:77), shared by every
 * instance in the process; the wrapper tracks it so a new instance can be started at phase 0. */
static int g_decim_phase = 0;
extern "C" void *ref_decim_new(void)
{
    if (g_decim_phase) {                       /* realign the shared static with one throw-away sample */
        filter_decimator tmp; complex z(0, 0), o[2]; int n;
        tmp.execute(1, &z, n, o);
        g_decim_phase = 0;
    }
    return new filter_decimator;
}
extern "C" void ref_decim_free(void *p) { delete static_cast<filter_decimator *>(p); }
extern "C" int ref_decim_execute(void *p, int len_in, const float *in, float *out)
{
    int n = 0;
    static_cast<filter_decimator *>(p)->execute(len_in, reinterpret_cast<complex *>(const_cast<float *>(in)), n,
                                                reinterpret_cast<complex *>(out));
    g_decim_phase = (g_decim_phase + len_in) & 1;
    return n;
}

typedef interpolator_farrow<complex, float> ref_farrow_t;                         /* dvbt2_demodulator.h:129 */
extern "C" void *ref_farrow_new(void) { return new ref_farrow_t; }
extern "C" void ref_farrow_free(void *p) { delete static_cast<ref_farrow_t *>(p); }
extern "C" int ref_farrow_execute(void *p, int len_in, const float *in, double resample, float *out)
{
    int n = 0;
    (*static_cast<ref_farrow_t *>(p))(len_in, reinterpret_cast<complex *>(const_cast<float *>(in)), resample, n,
                                      reinterpret_cast<complex *>(out));
    return n;
}

/* loop filters with the template arguments dvbt2_demodulator.h:94-116 uses */
static constexpr float ref_dc_ratio = 1.0e-6f;
static constexpr float ref_damping_phase = 0.3f, ref_damping_freq = 0.7f;
static constexpr int ref_fs_hz = static_cast<int>(1.0f / (1.0e-6f * 7.0f / 64.0f));
typedef exponential_averager<float, float, ref_dc_ratio> ref_avg_t;
typedef proportional_integral_loop_filter<float, float, ref_damping_phase, 1000000, ref_fs_hz> ref_pi_phase_t;
typedef proportional_integral_loop_filter<float, float, ref_damping_freq, 4000000, ref_fs_hz> ref_pi_freq_t;
extern "C" void *ref_avg_new(void) { return new ref_avg_t; }
extern "C" void ref_avg_free(void *p) { delete static_cast<ref_avg_t *>(p); }
extern "C" void ref_avg_run(void *p, int n, const float *in, float *out)
{
    ref_avg_t &a = *static_cast<ref_avg_t *>(p);
    for (int i = 0; i < n; i++) out[i] = a(in[i]);
}
extern "C" void *ref_pi_new(int which) { return which ? static_cast<void *>(new ref_pi_freq_t) : static_cast<void *>(new ref_pi_phase_t); }
extern "C" void ref_pi_free(int which, void *p)
{
    if (which) delete static_cast<ref_pi_freq_t *>(p); else delete static_cast<ref_pi_phase_t *>(p);
}
extern "C" float ref_pi_step(int which, void *p, float err, float max_integral)
{
    return which ? (*static_cast<ref_pi_freq_t *>(p))(err, max_integral) : (*static_cast<ref_pi_phase_t *>(p))(err, max_integral);
}

/* buffers.hh with the P1 correlator's lengths (p1_symbol.h:53-60) */
typedef sum_of_buffer<complex, 482> ref_sum_b_t;
typedef sum_of_buffer<complex, 542> ref_sum_c_t;
extern "C" void ref_sum_run(int which, int n, const float *in, float *out)
{
    const complex *ci = reinterpret_cast<const complex *>(in);
    complex *co = reinterpret_cast<complex *>(out);
    if (which) { ref_sum_c_t s; for (int i = 0; i < n; i++) co[i] = s(ci[i]); }
    else { ref_sum_b_t s; for (int i = 0; i < n; i++) co[i] = s(ci[i]); }
}
extern "C" void ref_delay_run(int delay, int n, const float *in, float *out)
{
    const complex *ci = reinterpret_cast<const complex *>(in);
    complex *co = reinterpret_cast<complex *>(out);
    if (delay == 482) { delay_buffer<complex, 482> d; for (int i = 0; i < n; i++) co[i] = d(ci[i]); }
    else if (delay == 542) { delay_buffer<complex, 542> d; for (int i = 0; i < n; i++) co[i] = d(ci[i]); }
    else if (delay == 964) { delay_buffer<complex, 964> d; for (int i = 0; i < n; i++) co[i] = d(ci[i]); }
    else { delay_buffer<complex, 2> d; for (int i = 0; i < n; i++) co[i] = d(ci[i]); }
}
