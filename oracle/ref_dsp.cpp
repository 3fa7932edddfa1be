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
