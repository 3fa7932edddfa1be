// bch_tables.cpp -- see bch_tables.h
#include "bch_tables.h"

namespace t2gpu {

bool bch_build_tables(int m, int t, BchTables &o)
{
    if ((m != 14 && m != 16) || t < 1 || t > 12) return false;
    const uint32_t prim = m == 16 ? 0x1002Du : 0x402Bu;       // 1+x^2+x^3+x^5+x^16 / 1+x+x^3+x^5+x^14
    o.m = m; o.t = t; o.order = (1 << m) - 1;
    o.exp.assign(o.order, 0);
    o.log.assign(o.order + 1, 0);
    uint32_t x = 1;
    for (int i = 0; i < o.order; ++i) {
        o.exp[i] = (uint16_t)x;
        o.log[x] = (uint16_t)i;
        x <<= 1;
        if (x >> m) x ^= prim;
    }
    if (x != 1) return false;                                  // alpha must have order 2^m - 1
    auto mul = [&](uint32_t a, uint32_t b) -> uint32_t {
        return (a && b) ? o.exp[(o.log[a] + o.log[b]) % o.order] : 0u;
    };
    o.minpoly.assign(t, 0);
    o.rem.assign((size_t)t * 256, 0);
    o.basis.assign((size_t)t * 16, 0);
    const uint32_t mask = (1u << m) - 1u;
    for (int i = 0; i < t; ++i) {
        const int j = 2 * i + 1;
        // (X + beta)(X + beta^2)(X + beta^4)... over the conjugacy class of beta = alpha^j
        std::vector<uint32_t> c{1};
        int e = j % o.order;
        do {
            const uint32_t root = o.exp[e];
            std::vector<uint32_t> n(c.size() + 1, 0);
            for (size_t k = 0; k < c.size(); ++k) { n[k + 1] ^= c[k]; n[k] ^= mul(c[k], root); }
            c.swap(n);
            e = (2 * e) % o.order;
        } while (e != j % o.order);
        if ((int)c.size() != m + 1) return false;
        uint32_t bits = 0;
        for (int k = 0; k <= m; ++k) {
            if (c[k] > 1) return false;
            bits |= c[k] << k;
        }
        o.minpoly[i] = bits;
        for (int v = 0; v < 256; ++v) {                        // remainder register holds m bits, a byte enters at the top
            uint32_t reg = (uint32_t)v << (m - 8);
            for (int s = 0; s < 8; ++s) {
                const uint32_t top = reg >> (m - 1) & 1u;
                reg = (reg << 1) & mask;
                if (top) reg ^= bits & mask;
            }
            o.rem[(size_t)i * 256 + v] = (uint16_t)reg;
        }
        for (int b = 0; b < 16; ++b) o.basis[(size_t)i * 16 + b] = b < m ? o.exp[(int)((long long)j * b % o.order)] : 0;
    }
    return true;
}

}  // namespace t2gpu
