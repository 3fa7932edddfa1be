// ldpc_kernel.h -- launch interface between the C-ABI layer and ldpc_kernel.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define T2GPU_LDPC_THREADS 384

namespace t2gpu {

struct LdpcLayerDev {
    int first_entry, cnt, lmax, n_conflict;
};

struct LdpcKernelParams {
    // code
    int n, k, q;
    const LdpcLayerDev *layers;   // [q]
    const uint32_t *entries;      // packed base | shift<<16
    const uint8_t *levels;        // [q*360]
    // job
    const int8_t *llr;            // [n_frames][n] received LLRs, transmitted order
    int n_frames;
    int group;                    // frames that stop together (reference SIMD batch: 32; 1 = independent frames)
    int max_trials;               // reference TRIALS = 25 (ldpc_decoder.h:63)
    uint8_t *bits;                // [n_frames][k] one bit per byte (may be null)
    int8_t *llr_out;              // [n_frames][n] final a-posteriori LLRs (may be null; parity tests)
    int *trials_left;             // [ceil(n_frames/group)]
    // scratch
    uint2 *state;                 // [grid][q*360] check-node records
    unsigned *sync;               // [batches][max_trials+1], zeroed before launch
    int *error;                   // zeroed before launch; 1 = batch rendezvous timed out
    long long spin_timeout_ticks; // wall_clock64 ticks (100 MHz)
    int lds_ctl_offset;           // byte offset of the control words behind the LLR array
};

hipError_t ldpc_kernel_attributes(int lds_bytes, int *blocks_per_cu);
hipError_t ldpc_kernel_launch(const LdpcKernelParams &p, int grid, int lds_bytes, hipStream_t stream);

}  // namespace t2gpu
