// p1_kernels.h -- launchers of the P1 preamble detector kernels (p1_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

constexpr int P1_HIST = 2048;              // samples of history kept in front of every call's input

// Device-resident detector state: what p1_symbol keeps in members between execute() calls (p1_symbol.h:49-70).
struct P1State {
    float correlation;                     // value of the last correlator update (:161)
    float begin_threshold, end_threshold;  // :63-64, :88-91
    float max_correlation;
    float arg_max_re, arg_max_im;
    int32_t correlation_detect;
    int32_t idx_buffer;
    int32_t idx_fq_shift;                  // static index into fq_shift (:81)
    int32_t p1_decoded;
};

// Result of one pass over a call's samples.
struct P1Result {
    int32_t status;                        // 0 = input exhausted, 1 = P1 detected, 2 = buffer overflow reset (p1_symbol.cpp:101-104) at `consumed`
    int32_t consumed;                      // samples consumed (the reference's _consume advance)
    int32_t idx_buffer_sym;
    int32_t p1_decoded;
    int32_t preamble, fft_mode, s1, s2, shift;
    int32_t a_part_clipped;                // part A reached further back than the 2048-sample history (reference reads out of bounds there)
    float max_correlation, arg_max_re, arg_max_im;
    float pad;
    double coarse_freq_offset;
};

struct P1Params {
    float2 *xb;                            // [P1_HIST + n]: history then this call's samples
    int n;
    const float2 *fq_shift;                // 1024 entries (sin, cos) as p1_symbol.cpp:26-32 fills them
    const float2 *twiddle;                 // 1024-point FFT twiddles
    float *corr;                           // [n]
    float2 *out;                           // [n] correlator output a*d
    P1State *state;
    P1Result *result;
    float2 *p1_fft;                        // [1024] fft-shifted spectrum of part A (kept for inspection)
    int reset_flag;
    int gain_changed;                      // :88-91: thresholds follow the level estimate once the gain is settled
    float level_detect;
};

void launch_p1(const P1Params &p, hipStream_t stream);
