// loop_device.h -- the tracking loops' state as the DEVICE keeps it while t2gpu_demod_execute runs a frame's data symbols without waiting
// for each symbol's results ("the loop on the device"): what t2gpu_sync (the two PI loop filters of
// /root/reference/src/DSP/loop_filters.hh:20-54, frequency_est_filtered, phase_est_filtered) and the front end's two NCO accumulators
// (dvbt2_demodulator.h: phase_nco, frequency_nco) hold on the host. sym_sync_kernel's last lane advances the filters with a symbol's
// floats exactly as t2gpu_sync_frequency / t2gpu_sync_symbol do; front_one_kernel's workgroup 0 plans the chunk's NCO runs from it
// exactly as t2_plan_nco does and leaves the accumulators advanced. The host recomputes all of it one symbol later from the same floats
// and compares when it leaves the mode.
#pragma once
#include <stdint.h>

struct T2DevLoop {
    float phase_nco, frequency_nco;                  // the NCO's accumulators
    float pe, fe;                                    // phase_est_filtered; frequency_est_filtered + the emulated tuner: inputs of the next chunk's NCO
    float frequency_est_filtered, tuner;
    float f_kp, f_ki, f_int, p_kp, p_ki, p_int;      // loop_filter_frequency_offset, loop_filter_phase_offset
    int32_t error;                                   // 2: the NCO planner ran out of room (T2_LOOP_RUNS_CAP runs)
    int32_t pad_[3];
};
constexpr int T2_LOOP_RUNS_CAP = 2048;               // NCO runs one chunk may take on the device (a 32K symbol: ~4 at 100 Hz of residual offset, ~150 at 1 kHz)
