// fec_kernels.h -- launch interface of the FEC-side kernels other than the LDPC decoder.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace t2gpu {

// K-descramble: out[f][i] = in[f][i] ^ prbs[i], i < k_bch (bch_decoder.cpp:136-142)
hipError_t launch_bch_descramble(const uint8_t *bits, int n_frames, int k_ldpc, int k_bch, const uint8_t *prbs, uint8_t *out,
                                 hipStream_t s);

// K-descramble-pack: out[f][j] = (in[f][8j..8j+7] packed MSB first) ^ prbs_packed[j], j < k_bch / 8 (bch_decoder.cpp:136-142 +
// the byte assembly of bb_de_header.cpp:84-448)
hipError_t launch_bch_descramble_pack(const uint8_t *bits, int n_frames, int k_ldpc, int k_bch, const uint8_t *prbs_packed, uint8_t *out,
                                      hipStream_t s);

// K-rows-to-host: bytes from device memory to (page-locked, device-visible) host memory by a kernel on stream s
hipError_t launch_copy_bytes(const void *src, void *dst, long bytes, hipStream_t s);

long demap_terms_padded(int n_snr);      // term pairs of scratch launch_demap_stats* need per TI block (n_snr rounded up to whole chunks)

struct DemapParams {
    int mod;                 // 0 QPSK, 1 16-QAM, 2 64-QAM, 3 256-QAM
    int fec_size, bits_per_cell, cells_per_fec;
    int rotate;              // de-rotate by e^{-j ROT} first
    float rot_c, rot_s;      // cos(-ROT), sin(-ROT) as floats
    float d;                 // normalisation factor
    const uint16_t *address; // [fec_size] bit de-interleaver (null for QPSK)
    int saturate;            // 0 = the reference's truncating int8 cast (wraps), 1 = clamp to [-128, 127] (extension)
};
// K-snr-reduce: sums[0] = sum |s|^2, sums[1] = sum |e|^2 over the hard decisions of n_snr cells (device doubles),
// partial[] = scratch of 2*blocks doubles. Second stage folds them and writes float sums[0..2] (s, e, precision).
hipError_t launch_demap_stats(const DemapParams &p, const float2 *cells, int n_snr, double *partial, int blocks, float2 *terms, float *sums,
                              float precision_override, hipStream_t s);
hipError_t launch_demap_stats_batch(const DemapParams &p, const float2 *cells, long cells_stride, int n_snr, int n_batch, double *partial,
                                    int blocks, float2 *terms, float *sums, int sums_stride, float precision_override, hipStream_t s);
// K-demap-bdi: one workgroup per FEC frame; LLRs staged in LDS in LDPC order, written out coalesced.
// frames_per_sums / sums_stride: several TI blocks in one launch, each with its own statistics triple (0 = one triple for all)
hipError_t launch_demap_llr(const DemapParams &p, const float2 *cells, int n_frames, const float *sums, int8_t *out, hipStream_t s,
                            int frames_per_sums = 0, int sums_stride = 0);

// K-ti-cdi-qdelay: scatter cells n0 .. n0+n-1 of a TI block (time + cell de-interleave + cyclic Q delay removal)
struct TiParams {
    int cells_per_fec, rows, cols, ti_block_size;
    const int32_t *perm;
};
hipError_t launch_ti_scatter(const TiParams &p, const float2 *cells, int n0, int n, float2 *out, float *first_q, hipStream_t s);
// held-Q fix-up after the last cell of the TI block: block order / lost flags are static per geometry (see t2gpu_fec.cpp)
hipError_t launch_ti_fixup(const TiParams &p, const int32_t *order, const uint8_t *lost, int num_blocks, const float *first_q,
                           float2 *out, hipStream_t s);

// Whole TI blocks, one workgroup per FEC block with the block's cells staged in LDS (cells_per_fec * 8 bytes): reads the five
// interleaver columns of the block row by row (40-byte runs), permutes inside LDS, writes the block contiguously. frames TI
// blocks of the same geometry in one launch: block f reads cells + f * in_stride, writes out + f * out_stride (in cells).
// lost_by_block[b] = 1: the reference never stores the parked Q of FEC block b (see t2gpu_ti_begin), the last cell keeps its Q.
// Returns hipErrorInvalidValue when the FEC block does not fit LDS (QPSK with 64800-bit frames): use the scatter kernels.
// terms (optional): the demapper's per-cell statistics terms of the first n_snr de-interleaved cells of every TI block, formed while
// the cells are on their way out (two planes of demap_terms_padded(n_snr) floats per TI block, as demap_terms_kernel writes them)
// pad_key (may be null): what the zeros behind every plane's terms were last written for (geometry key); launch_ti_blocks skips the fill when it is
// unchanged -- nothing else writes there as long as the owner resets the key whenever the scratch is used otherwise
struct TiTerms { const DemapParams *dp = nullptr; float2 *terms = nullptr; int n_snr = 0; long *pad_key = nullptr; };
hipError_t launch_ti_blocks(const TiParams &p, const uint8_t *lost_by_block, int num_blocks, const float2 *cells, long in_stride,
                            float2 *out, long out_stride, int frames, hipStream_t s, const TiTerms *tt = nullptr);
// the walk + scale of launch_demap_stats_batch's exact form on terms that are already there (launch_ti_blocks with TiTerms)
bool demap_stats_exact_form();
hipError_t launch_demap_stats_from_terms(const DemapParams &p, int n_snr, int n_batch, float2 *terms, float *sums, int sums_stride,
                                         float precision_override, hipStream_t s);

}  // namespace t2gpu

