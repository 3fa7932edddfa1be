// bch_kernels.h -- K-bch: outer-code (BCH) check and correction of LDPC-decoded FEC frames on gfx950. See bch_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace t2gpu {

struct BchDev {
    const uint16_t *exp, *log, *rem, *basis;   // device copies of BchTables
    int m, t, n_bits;                          // field, correctable errors, n_bch (= k_ldpc, a multiple of 8)
};

// bits: [n_frames][n_bits] one bit per byte, corrected in place (8-byte aligned); status[f] = corrected bits, -1 = more than t errors
hipError_t launch_bch_decode(uint8_t *bits, int n_frames, const BchDev &p, int32_t *status, hipStream_t s);

}  // namespace t2gpu
