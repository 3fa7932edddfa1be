// bch_tables.h -- GF(2^m) tables for the outer code of DVB-T2 (EN 302 755 clause 6.1.1; m = 16 for 64 800-bit FEC frames, 14 for
// 16 200-bit ones). Host side, built once per (m, t). The reference has no BCH decoder (bch_decoder.cpp:136 "TODO BCH decode"):
// this is SURVEY.md 8(f)-2, an opt-in stage in front of the descrambler.
#pragma once
#include <cstdint>
#include <vector>

namespace t2gpu {

struct BchTables {
    int m = 0, t = 0, order = 0;
    std::vector<uint16_t> exp;        // [order]      alpha^i
    std::vector<uint16_t> log;        // [order + 1]  log[alpha^i] = i (log[0] unused)
    std::vector<uint32_t> minpoly;    // [t]          minimal polynomial of alpha^(2i+1), bit k = coefficient of x^k
    std::vector<uint16_t> rem;        // [t][256]     (v(x) x^m) mod minpoly[i]: byte-at-a-time remainder step
    std::vector<uint16_t> basis;      // [t][16]      alpha^((2i+1) b): a remainder c(x) is the field element sum c_b basis[b]
};

// false when (m, t) is not a DVB-T2 outer code or a minimal polynomial has a degree other than m
bool bch_build_tables(int m, int t, BchTables &out);

}  // namespace t2gpu
