// front_plan.h -- exact closed forms for the two sequential float accumulators of the reference's sample-rate front end.
//
// The reference advances two float accumulators once per sample: the NCO phase `frequency_nco -= frequency_est_filtered`
// with 2*pi wrap loops (/root/reference/src/DVB_T2/dvbt2_demodulator.cpp:187-193) and the Farrow resampler's fractional
// position `x1 += delay_x ... x1 -= 1` (/root/reference/src/DSP/interpolator_farrow.hh:57-63). Neither depends on the
// signal, only on the loop values of the chunk, and a float accumulator that adds a constant moves by a CONSTANT number of
// ulps for as long as it stays inside one binade. The planner (host, a few microseconds per chunk) walks the accumulators
// binade by binade with real float operations and emits runs {first sample, value, per-sample step}; inside a run every GPU
// thread evaluates value + k*step in double, which is exact, so the device reproduces the reference's sequence of float
// values bit for bit without a sequential pass over the samples. Where the arithmetic is not exact (ties, roundings) the
// planner falls back to runs of length one, still exact.
#pragma once
#include <cstdint>
#include <vector>

struct FrontRun {
    int32_t i0;      // first input sample of the run (index inside the execute() call)
    int32_t o0;      // Farrow: index of the first output sample of input i0;  NCO: unused
    double base;     // accumulator value for sample i0 (a float value held in a double)
    double step;     // per-sample increment inside the run (exact)
    int32_t cnt;     // Farrow: outputs per input sample in this run;  NCO: unused
    float aux;       // NCO: phase_nco of the chunk (dvbt2_demodulator.cpp:165-171)
};
static_assert(sizeof(FrontRun) == 32, "FrontRun layout is shared with the device");

// NCO: `acc` is frequency_nco before the first sample and after the last on return. Appends runs for samples
// [i_begin, i_begin + n).
void t2_plan_nco(float &acc, int i_begin, int n, float frequency_est_filtered, float phase_nco, std::vector<FrontRun> &runs);

// Farrow: `x1` is the fractional position before the first input sample and after the last on return; `o_begin` the output
// index of the first output; returns the number of outputs the n inputs produce, or -1 if delay_x is outside (1/256, 4).
long t2_plan_farrow(float &x1, int i_begin, long o_begin, int n, float delay_x, std::vector<FrontRun> &runs);

// phase_nco += phase_est_filtered with the reference's wrap loops (dvbt2_demodulator.cpp:165-171)
float t2_wrap_2pi(float a);
