// ldpc_kernel.h -- launch interface between the C-ABI layer and ldpc_kernel2.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define T2GPU_LDPC_THREADS 768      // one check node of a layer on two adjacent lanes (ldpc_cn3.h): 720 of 768 lanes, 12 wavefronts

namespace t2gpu {

struct LdpcLayerDev {
    int first_entry, cnt, lmax, nc, kind, step, band, band_prefetch;   // band_prefetch bit 1: no closing barrier (ldpc_graph.h: no_close);   // band: GENERIC layers the two-frame kernel walks in bands (ldpc_graph.h)
};

struct LdpcKernelParams {
    // code
    int n, k, q;
    const LdpcLayerDev *layers;   // [q]
    const uint4 *layer_words;     // [q] the two-frame kernel's packed form: x = kind | cnt << 2 | nc << 7 | lmax << 12 | band_prefetch << 21 | no_close << 22 |
                                  // band << 23 | (layer i + 1 is GENERIC) << 29, y = step | first_entry << 16, z = entries[first_entry], w = the first_entry of layer (i + 1) mod q
    const uint32_t *entries;      // packed base | shift<<16
    const uint32_t *entries2;     // the same as pairs (base + lds_base, shift): what the check-node load reads
    int lds_base;                 // LDS address of the LLR array the pairs were made for (= the kernel's static LDS size)
    const uint32_t *cninfo;       // [q*360] level | dependent-slot mask << 8 (GENERIC layers)
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
    unsigned *ticket;             // two-frame kernel, optional: [0] = next batch to hand out, [1 + slot * ticket_rounds + r] = batch + 1
    int ticket_rounds;            //   that slot's workgroups take in their round r; zeroed before launch. Null: batches strided statically
    int *error;                   // zeroed before launch; 1 = batch rendezvous timed out
    long long spin_timeout_ticks; // wall_clock64 ticks (100 MHz)
    int lds_ctl_offset;           // byte offset of the control words behind the LLR array
    int lds_rec_offset;           // byte offset of the 360 chain-walk records (PAIR layers)
    int lds_sign_offset;          // byte offset of the packed sign words (13 dwords per 360-bit group)
    int lds_ent_offset, n_entries; // pair-lane kernel: byte offset of the LDS copy of entries2, number of entries
    long long *prof;              // optional [prof_blocks][8] cycle counters + [64] per-layer cycles of workgroup 0 (diagnostics; null in production)
    int prof_blocks;
    unsigned *resident;           // counts workgroups that have started, cumulatively over launches (t2gpu_ldpc_wait_resident)
};

// two frames per workgroup (ldpc_kernel2.hip): LLR bytes interleaved in LDS, records of `ldpc_kernel2_record_dwords` dwords per lane
// and layer ([grid][q][720][dwords], passed through LdpcKernelParams::state), group / 2 workgroups per SIMD batch
int ldpc_kernel2_record_dwords(int min_cnt, int max_cnt);
hipError_t ldpc_kernel2_attributes(int min_cnt, int max_cnt, int lds_bytes, int *blocks_per_cu, int *static_lds_bytes);
hipError_t ldpc_kernel2_launch(int min_cnt, int max_cnt, const LdpcKernelParams &p, int grid, int lds_bytes, hipStream_t stream, bool allow_cooperative = true);

// bits / verdicts / error word of a decode into page-locked host memory by a kernel of the decode's own stream (no copy engine)
hipError_t ldpc_clear(unsigned *a, size_t na, unsigned *b, size_t nb, hipStream_t stream, unsigned *c = nullptr, size_t nc = 0);
hipError_t ldpc_results_to_host(const uint8_t *d_bits, size_t bytes, const int *d_trials, int n_trials, const int *d_error, uint8_t *h_bits, int *h_trials,
                                int *h_error, hipStream_t stream);

}  // namespace t2gpu
