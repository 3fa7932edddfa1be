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

// One window of samples to search. Stream mode (one window): base = xb + P1_HIST with P1_HIST valid samples in front.
// Batch mode: windows lie in the caller's stream, each starts a fresh correlator (zeros in front of it), as the reference does
// after every detected P1 (reset_buffer, p1_symbol.cpp:133).
struct P1Window { long start; int len; int buf_off; };   // buf_off: offset of this window in corr[] / out[]

struct P1Params {
    float2 *xb;                            // stream mode: [P1_HIST + n] history then this call's samples (carry target)
    const float2 *base;                    // sample k of window w is base[win[w].start + k]
    const P1Window *win;                   // [n_windows] (device)
    int n_windows;
    int hist;                              // valid samples in front of a window's first sample (P1_HIST or 0)
    int n;                                 // longest window
    const float2 *fq_shift;                // 1024 entries (sin, cos) as p1_symbol.cpp:26-32 fills them
    const float2 *twiddle;                 // 1024-point FFT twiddles
    float *corr;                           // [sum of window lengths]
    float2 *out;                           // correlator output a*d, same layout
    P1State *state;                        // [n_windows]
    P1Result *result;                      // [n_windows]
    float2 *p1_fft;                        // [n_windows][1024] fft-shifted spectrum of part A (kept for inspection)
    int reset_flag;
    int gain_changed;                      // :88-91: thresholds follow the level estimate once the gain is settled
    int serial_detector;                   // tests: the threshold state machine sample by sample only (no closed-form stretches)
    float level_detect;
};

void launch_p1(const P1Params &p, hipStream_t stream);
