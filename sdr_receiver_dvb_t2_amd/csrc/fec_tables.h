// fec_tables.h -- host-side construction of the permutation tables the FEC-side kernels gather/scatter through.
// Own formulation of ETSI EN 302 755 6.1.3 (bit interleaver: parity part handled by the LDPC stage, column twist,
// demultiplexer) and 6.4 (cell interleaver), checked against the reference-shaped oracle in tests/.
#pragma once
#include <stdint.h>
#include <vector>

namespace t2gpu {

// Where LLR number k of an FEC frame (k = cell*bits_per_cell + position inside the cell, the order the demapper emits
// them) lands in the LDPC input frame. Replaces llr_demapper::address_generator (llr_demapper.cpp:110-130) and the
// table choice of qam16/qam64/qam256 (:288-296,:447-455,:675-686). mod: 1 = 16-QAM, 2 = 64-QAM, 3 = 256-QAM.
bool bitdeint_address(int mod, int fec_type, int code_rate, std::vector<uint16_t> &address);

// Cell de-interleaver: perm[r*ncells + L_r(w)] = r*ncells + w for FEC block r of a TI block
// (time_deinterleaver::address_cell_deinterleaving, time_deinterleaver.cpp:174-266).
void cell_deint_permutation(int num_blocks, int cells_per_fec, std::vector<int32_t> &perm);

// BB scrambler sequence 1 + x^14 + x^15, one bit per byte (bch_decoder::init_descrambler, bch_decoder.cpp:50-61)
void bb_prbs(std::vector<uint8_t> &bits, int n);

inline int fec_size_of(int fec_type) { return fec_type == 1 ? 64800 : 16200; }

}  // namespace t2gpu
