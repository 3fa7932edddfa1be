// ofdm_kernels.h -- launch interface of the OFDM-side kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ofdm_tables.h"
#include "loop_device.h"

namespace t2gpu {

// Where the symbols of a batch start inside the input stream (cells): symbol i is at
// first + (i / per_frame) * frame_stride + (i % per_frame) * sym_stride. The FFT reads fft_size cells from there, which is how the
// guard interval is dropped (symbol_acquisition copies buffer_sym + guard_interval_size, dvbt2_demodulator.cpp:332-333).
struct FftLayout { long first, frame_stride; int per_frame, sym_stride; };

// twiddle[m] = exp(-j*2*pi*m/N), m in [0, N); layout = nullptr: contiguous symbols
// scratch (may be null): room for scratch_symbols x fft_size cells; calls of at most that many symbols then run as two launches spread
// over many CUs (ofdm_kernels.hip: fft_stage_a_kernel / fft_stage_bc_kernel, bit-identical output) instead of one workgroup per symbol
hipError_t launch_fft(int fft_size, const float2 *in, float2 *out, const float2 *twiddle, int n_symbols, int max_blocks,
                      hipStream_t s, const FftLayout *layout = nullptr, float2 *scratch = nullptr, int scratch_symbols = 0);
constexpr int FFT_WIDE_SYMBOLS = 2;                         // calls of up to this many symbols take the two-launch form

struct EqParams {
    int fft_size, l_nulls, k_total, c_data, n_p2, max_seg;   // c_data: cells out per symbol; n_p2: frame index of table row 0
    float amp_sp, amp_cp, amp_p2;
    const uint8_t *map;        // [rows][k_total] carrier types of every data symbol of the frame
    const float *refer;        // [rows][k_total] signed pilot reference
    const int4 *segs;          // [rows][max_seg] (left pilot, right pilot, first de-interleaver index, data cells)
    const int32_t *seg_count;  // [rows]
    const int32_t *h_even, *h_odd;
    const float2 *lut_cs;      // the 65536-entry cos / sin tables of DSP/fast_math.h, interleaved: one 8-byte read per cell
    // Frame layout (per_frame > 0): symbol b of the batch is data symbol `first + b % per_frame` of frame b / per_frame; its
    // spectrum is at symbols + (frame * in_syms_per_frame + first + b % per_frame) * fft_size, its cells go to
    // out + frame * out_frame_stride + out_offset + (b % per_frame) * c_data. per_frame == 0: symbols and cells back to back,
    // positions from symbol_index[].
    int per_frame = 0, first = 0, in_syms_per_frame = 0;
    long out_frame_stride = 0, out_offset = 0;
    int row_major = 1;         // frame layout, per_frame > 1: workgroups in table-row order (all frames of a row together); 0 = symbol order
    int out_skip = 0;          // frame layout: the first out_skip cells of the symbol are not stored (the L1 cells of P2), the rest move up
    float2 *skip_out = nullptr;       // frame layout, out_skip > 0: where the skipped cells go instead (frame f at skip_out + f * out_skip) --
                                      // the L1-pre / L1-post cells of every P2 symbol, for the host's per-frame L1 parse
    // P2 and frame-closing tables: the pilot amplitude never changes inside p2_symbol::execute / fc_symbol::execute, and the
    // reference binary (-Ofast, sdr_receiver_dvb_t2.pro:33-39) evaluates sqrt(norm(cell)) / amp_pilot as a product with
    // 1 / amp_pilot there; data_symbol::execute, whose amplitude alternates, keeps the division. Pinned by tests/golden/t2sym_golden.npz.
    int recip_amp = 0;
    const uint16_t *dcar = nullptr;   // [rows][dcar_stride] carrier index (from the first active carrier) of every data cell, in cell order
    int dcar_stride = 0;
    // Output-range form (eq_split_kernel; n_splits > 0): the symbol's c_data output positions are cut into n_splits ranges
    // [split_q(s), split_q(s + 1)), one workgroup per (symbol, range) keeps its range in LDS and stores it in one contiguous run.
    const uint16_t *cellq = nullptr;  // [rows][cq_steps / 4][max_seg][4] output position (frequency de-interleaver of the row's parity) of data cells
                                      // 4 k4 .. 4 k4 + 3 of segment seg at [k4][seg] (a wavefront of segments reads one run of 8-byte words per
                                      // four steps); 0xffff past a segment's end
    int cq_steps = 0;                 // steps stored per row: the longest segment of the table, rounded up to EQS_PU
    const uint32_t *sel = nullptr;    // [rows][c_data] (carrier << 16 | output position) of the data cells, range by range, cell order inside a range
    int n_splits = 0;
    // ONE symbol (n_symbols = 1, per_frame = 0, output-range form): the cells also go to page-locked host memory by the workgroups' own stores
    // and the last workgroup raises *pub_flag = pub_seq behind them (system scope) -- what a publishing launch of its own did before
    float2 *pub_cells = nullptr;
    unsigned *pub_flag = nullptr, *pub_count = nullptr;    // pub_count: a zeroed device word of the caller's (left at zero)
    unsigned pub_seq = 0;
};
// first output position of range s (even, so that a range starts on a 16-byte boundary of the symbol's cells)
__host__ __device__ inline int eq_split_q(int c_data, int n_splits, int s) { return s >= n_splits ? c_data : (int)(((long)c_data * s / n_splits) & ~1L); }
constexpr int EQS_PU = 8;                   // eq_split_kernel reads its destinations EQS_PU steps ahead of the recurrence
hipError_t launch_eq_data(const EqParams &p, const float2 *symbols, const int32_t *symbol_index, int n_symbols, float2 *out,
                          float4 *pilot_scratch, float2 *sync, hipStream_t s);

// The results of ONE symbol to page-locked host memory in one launch (the slot-shaped path, t2gpu_demod.cpp): n_cells cells, the guard
// correlation's four floats (cp4 may be null) and the two synchronisation floats go to h_cells / h_small[0..3] / h_small[4..5] by the
// kernel's own stores, and *h_flag = seq is stored behind all of them (system scope): the host reads seq there instead of waiting for
// three copies and the stream. d_count: a zeroed device word of the caller's (left at zero).
hipError_t launch_publish_symbol(const float2 *cells, int n_cells, const float *cp4, const float *sync2, float2 *h_cells, float *h_small,
                                 unsigned *h_flag, unsigned seq, unsigned *d_count, hipStream_t s);

// The synchronisation floats of ONE symbol straight from its spectrum, with the guard correlation of the buffered symbol (ofdm_kernels.hip:
// sym_sync_kernel). symbol: the spectrum (fft_size cells); buffered (may be null): the symbol as collected, guard + fft_size cells;
// cp_out / sync (device, may be null) receive {sum.re, sum.im, frequency_est, 0} / {phase_offset, sample_rate_offset}; h_small / h_flag
// (page-locked, may be null): the same six floats at h_small[0..3] / [4..5] and *h_flag = seq behind them.
// loop (device, may be null): the tracking loops' state the launch's last lane advances with the symbol's floats (loop_device.h); its new
// phase_est_filtered / frequency input then also go to h_small[6..7].
hipError_t launch_sym_sync(const EqParams &p, const float2 *symbol, int idx_symbol, const float2 *buffered, int guard, float4 *cp_out,
                           float2 *sync, float *h_small, unsigned *h_flag, unsigned seq, hipStream_t s, T2DevLoop *loop = nullptr);

// One symbol's FFT (the two-launch form) with launch_sym_sync's work done by the last workgroup of its second launch. count: a zeroed
// device word of the caller's (left at zero). hipErrorInvalidValue when the pilot table of `p` does not fit the FFT's exchange buffer
// ((max_seg + 2) x 16 + 4096 bytes of 33 792: P2 symbols, dense pilot patterns) -- the caller then takes the two entry points one after
// the other.
hipError_t launch_fft_sym_sync(int fft_size, const float2 *in, float2 *out, const float2 *twiddle, const FftLayout &lay, float2 *scratch, unsigned *count,
                               const EqParams &p, int idx_symbol, const float2 *buffered, int guard, float4 *cp_out, float2 *sync, float *h_small,
                               unsigned *h_flag, unsigned seq, hipStream_t s, T2DevLoop *loop = nullptr, bool one_launch = true);

// what fft_one_sync_kernel takes, for the launch that runs a chunk's front end AND the symbol's transform (front_kernels.hip:
// front_fft_one_kernel; 32K symbols -- the front end's workgroups have 256 lanes, as stages B + C of a 32K transform)
struct FftOneArgs {
    const float2 *in; float2 *scratch, *out; const float2 *twiddle; unsigned *count;
    EqParams p; int idx_symbol; const float2 *buffered; int guard; float4 *cp_out; float2 *sync; float *h_small; unsigned *h_flag; unsigned seq;
    T2DevLoop *loop; int fft_size;
};
constexpr int FFT_ONE_LDS_FLOATS = 32 * 8 * 33;   // (= ofdm_device.h's FFT_BC_LDS_FLOATS)

}  // namespace t2gpu

// (internal, C++ linkage: t2gpu_demod.cpp) the arguments t2gpu_fft_sym_sync_dev would launch with, for the launch that also runs the chunk's
// front end: 0 = filled in, 1 = this symbol does not take the one-launch form (pilot tables too large for the exchange buffer, not 32K), -1 = error
struct t2gpu_ofdm;
int t2gpu_fft_one_args(t2gpu_ofdm *h, t2gpu_ofdm *tables, int kind, int idx_symbol, const float *d_buffered, int guard, int with_cp, float *d_spectrum,
                       float *h_small, unsigned *h_flag, unsigned seq, void *d_loop, t2gpu::FftOneArgs *out);

