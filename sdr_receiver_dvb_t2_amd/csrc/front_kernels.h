// front_kernels.h -- launchers of the sample-rate front end kernels (front_kernels.hip): dc removal / IQ-imbalance /
// NCO de-rotation, Farrow resampler, /2 decimation FIR, guard-interval correlation. See front_kernels.hip for the mapping.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "front_plan.h"
#include "loop_device.h"
#include "ofdm_kernels.h"

constexpr int FRONT_BLOCK = 4096;          // input samples per workgroup in the dc / de-rotation kernels (256 lanes x FRONT_PER)
constexpr int FRONT_PER = FRONT_BLOCK / 256;
constexpr int FRONT_RUN_STRIDE = 256;      // one entry of the run-index arrays per this many input samples

// Device-resident state of one front end (what dvbt2_demodulator keeps in members between execute() calls).
struct FrontState {
    double dc_re, dc_im;                   // exponential averagers (dvbt2_demodulator.h:94-96), kept in double on the device
    double theta[3];                       // sign statistics of the last call (dvbt2_demodulator.cpp:256-265)
    float c1, c2;                          // IQ-imbalance coefficients used by the NEXT call (:228-234)
    float level_detect;                    // :235
    int32_t decim_phase;                   // filter_decimator's d (filter_decimator.h:77)
    int32_t error_;                        // (in the page-locked copy a commit publishes: the one-launch form's error word)
    double theta_acc[3], n_acc;            // sign statistics / samples of the chunks of one execute() still open (FRONT_STAGE_HOLD_IQ)
};

struct FrontParams {
    const int16_t *i_in, *q_in;            // device, stride `stride` elements
    int stride;
    float short_to_float;
    int n;                                 // input samples of this call
    int n_blocks;                          // ceil(n / FRONT_BLOCK)
    FrontState *state;
    double *blk;                           // [n_blocks][4] = a, b_re, b_im, -   (block aggregates) then start values
    double *theta_part;                    // [n_blocks][4]
    const FrontRun *nco_runs; const int32_t *nco_index; int n_nco_runs;
    const FrontRun *far_runs; const int32_t *far_index; int n_far_runs;
    const float *lut_sin, *lut_cos;
    float2 *derot;                         // [3 + n]: 3 carried samples (delay_data_3,2,1) then this call's
    float2 *interp;                        // [63 + n_interp]: 63 carried samples then this call's
    int stages;                            // FRONT_STAGE_* mask: which parts of the chain this call runs
    long n_interp;
    float2 *out;                           // decimated stream, n_out cells
    long n_out;
    int decim_phase;                       // phase at the start of this call (host copy)
};

// FRONT_STAGE_HOLD_IQ: this call is one chunk of an execute() that goes on -- its sign statistics are added to the open sums and
// c1 / c2 / level_detect stay as they are until launch_front_commit_iq (the reference derives them once per execute(), :227-235)
enum { FRONT_STAGE_DEROTATE = 1, FRONT_STAGE_FARROW = 2, FRONT_STAGE_DECIMATE = 4, FRONT_STAGE_HOLD_IQ = 8 };
// h_copy / h_flag (may be null): the committed state is also stored to page-locked host memory and *h_flag = seq behind it
void launch_front_commit_iq(FrontState *state, FrontState *h_copy, unsigned *h_flag, unsigned seq, const int *error, hipStream_t stream);

void launch_front(const FrontParams &p, hipStream_t stream);
// the device's loop state set / read by launches (no copy engine): h_out / h_flag page-locked, *h_flag = seq behind the state
void launch_front_loop_set(T2DevLoop *d, const T2DevLoop &v, hipStream_t stream);
void launch_front_loop_get(const T2DevLoop *d, T2DevLoop *h_out, unsigned *h_flag, unsigned seq, hipStream_t stream);
// n int16 elements of I and of Q from page-locked host memory (device-visible addresses hi / hq) to di / dq, by a kernel
void launch_front_copy_in(const int16_t *hi, const int16_t *hq, int16_t *di, int16_t *dq, size_t n, hipStream_t stream);

// One launch, one pass per workgroup for a short call (front_kernels.hip: front_one_kernel). front_one_grid: the grid such a call needs,
// or 0 when the call does not qualify (too long, too many runs, not the whole chain, a resampling ratio outside [1/2, 1]).
// flags / rec / done: device words of the front end's own, zero at creation; seq: this call's number (> 0, growing); done_target: what
// `done` has been raised to by the launches before this one (each adds its grid). pre_out: 66 cells of scratch.
constexpr int F1_B = 1024, F1_PER = 4, F1_H = 128, F1_MAX_GRID = 96, F1_WCAP = 2304, FRONT_CHAIN_RUNS = 24;
struct FrontOneArgs {
    FrontParams p;
    unsigned long long seq, done_target;
    unsigned long long *flags, *done;          // [F1_MAX_GRID], [1]
    double *rec;                               // [F1_MAX_GRID][16]
    float2 *pre_out;                           // [3 + 63]
    int *error;                                // set when a look-back wait gave up (t2gpu_front_state reports it)
    T2DevLoop *loop;                           // non-null: the NCO runs of this (one-chunk) call are planned by workgroup 0 from the device's
    FrontRun *loop_runs;                       // loop state (every workgroup for itself; long plans into loop_runs[workgroup][T2_LOOP_RUNS_CAP]) and the accumulators are left advanced there
    // cp_wgs > 0: that many extra workgroups at the END of the grid bring cp_n int16 elements of I and of Q over from page-locked host memory
    // (device-visible addresses cp_si / cp_sq) to cp_di / cp_dq -- the samples of the call's NEXT chunk, beside this chunk's work
    const int16_t *cp_si, *cp_sq; int16_t *cp_di, *cp_dq; long cp_n; int cp_wgs;
    // development (T2GPU_FRONT_STAMPS=1): page-locked [64 workgroups][16] wall-clock stamps (100 MHz) of the phases of the launch, else null
    long long *stamps;
    FrontRun runs[FRONT_CHAIN_RUNS];           // NCO runs (none with `loop`), then Farrow runs
};
// the I/Q a launch of the one-launch form brings over for the chunk behind it (t2gpu_front_loop_fft)
struct FrontCopyAhead { const int16_t *si, *sq; int16_t *di, *dq; long n; };
int front_one_grid(const FrontParams &p, const FrontRun *far_runs, size_t n_nco_runs, size_t n_far_runs);
void launch_front_one(FrontOneArgs &a, int grid, hipStream_t stream);
// ... and with the transform + synchronisation floats of the 32K symbol the chunk completes in the same launch (front_fft_one_kernel)
void launch_front_fft_one(FrontOneArgs &a, int grid, const t2gpu::FftOneArgs &f, hipStream_t stream);

// (internal, C++ linkage: t2gpu_demod.cpp) t2gpu_front_execute_loop_dev; when `fft` is given and the chunk gives exactly need_out cells, the
// launch also runs *fft (the symbol those cells complete) and *fused is set to 1
struct t2gpu_front;
// ahead (may be null): I/Q the launch brings over from page-locked memory beside its own work
long t2gpu_front_loop_fft(t2gpu_front *h, int32_t chunk, double rs, const int16_t *d_i, const int16_t *d_q, float *d_out, long out_cap_cells,
                          void *stream, long need_out, const t2gpu::FftOneArgs *fft, int *fused, const FrontCopyAhead *ahead = nullptr);

// Guard-interval correlation of symbol_acquisition (dvbt2_demodulator.cpp:321-327): one workgroup per buffered symbol.
// sym: n_symbols x symbol_size cells (guard first); out[s] = (sum.re, sum.im, frequency_est, 0).
// Symbol i starts at first + (i / per_frame) * frame_stride + (i % per_frame) * (guard + fft_size).
void launch_cp_correlate(const float2 *sym, long first, long frame_stride, int per_frame, int n_symbols, int fft_size, int guard,
                         float4 *out, hipStream_t stream);
