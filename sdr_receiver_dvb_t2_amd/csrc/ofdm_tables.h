// ofdm_tables.h -- host-side mode arithmetic and per-symbol tables of the OFDM side (16K / 32K, SISO):
// carrier counts, carrier-type map and signed pilot reference per symbol, frequency de-interleaver addresses.
// Replaces, as init-time table builders, dvbt2_{p2,bwt_ext,data}_parameters_init (/root/reference/src/DVB_T2/
// dvbt2_definition.cpp:20-91,93-159,161-648), pilot_generator (/root/reference/src/DVB_T2/pilot_generator.cpp:48-2166)
// and address_freq_deinterleaver (/root/reference/src/DVB_T2/address_freq_deinterleaver.cpp:28-209).
#pragma once
#include <stdint.h>
#include <vector>

namespace t2gpu {

// carrier types, numerically equal to dvbt2_carrier_type_t (dvbt2_definition.h:96-106)
enum { T2_DATA = 1, T2_P2PILOT = 2, T2_P2PAPR = 3, T2_TRPAPR = 4, T2_SCATTERED = 5, T2_CONTINUAL = 6 };

struct T2Mode {
    // inputs (reference enums): fft_mode FFTSIZE_16K = 4 / FFTSIZE_32K = 5 (+ _T2GI variants 11 / 7),
    // carrier_mode 0 normal / 1 extended, pilot_pattern PP1..PP8 = 0..7, guard_interval_mode GI_* 0..6,
    // papr_mode 0 off / 1 ACE / 2 TR / 3 both, n_data = NUM_DATA_SYMBOLS of L1-pre (data symbols incl. frame closing)
    int fft_mode = 5, carrier_mode = 1, pilot_pattern = 6, guard_interval_mode = 4, papr_mode = 0, n_data = 59;
    // derived
    int fft_size = 0, is32k = 0, k_total = 0, k_ext = 0, k_offset = 0, l_nulls = 0;
    int n_p2 = 0, c_p2 = 0, c_data = 0, n_fc = 0, c_fc = 0, l_fc = 0, len_frame = 0, guard_interval_size = 0;
    int dx = 0, dy = 0;
    float amp_sp = 0, amp_cp = 0, amp_p2 = 0;
};

bool t2_mode_init(T2Mode &m);

// carrier map + signed reference amplitude of one symbol, k_total entries each.
// idx_symbol counts OFDM symbols of the T2 frame after P1: 0 = P2 (n_p2 = 1 for 16K/32K), then data symbols,
// the last one being the frame-closing symbol when l_fc = 1.
void t2_symbol_carriers(const T2Mode &m, int idx_symbol, std::vector<uint8_t> &map, std::vector<float> &refer);

// receiver-side frequency de-interleaver: h[address in the interleaved symbol] = position in the de-interleaved
// cell sequence; kind 0 = P2 (c_p2 cells), 1 = data (c_data), 2 = frame closing (n_fc). The reference uses h_odd for
// even symbol indices and h_even for odd ones (data_symbol.cpp:148-149).
void t2_freq_deint(const T2Mode &m, int kind, std::vector<int32_t> &h_even, std::vector<int32_t> &h_odd);

}  // namespace t2gpu
