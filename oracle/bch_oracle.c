/* bch_oracle.c -- CPU checker for the outer (BCH) code of DVB-T2. TEST INFRASTRUCTURE ONLY: nothing in the product path may call it.
 *
 * PARITY UNPINNED against the reference: the reference has no BCH decoder to compare with -- bch_decoder::execute keeps the first
 * k_bch bits of every block and descrambles them ("// TODO BCH decode", /root/reference/src/DVB_T2/bch_decoder.cpp:136); its only
 * BCH-related data are the (k_bch, n_bch) pairs (:79-134). What this file restates is the published code (ETSI EN 302 755 clause
 * 6.1.1: systematic t-error-correcting narrow-sense binary BCH over GF(2^16) for 64 800-bit FEC frames, GF(2^14) for 16 200-bit
 * frames, generator = product of the first t minimal polynomials, first message bit = highest power of x), written the textbook
 * way: bitwise LFSR encoder, all 2t syndromes by Horner's rule, Berlekamp-Massey, exhaustive root search. It is pinned by known
 * answers instead (tests/test_bch_oracle.py): the generator's degree (192 / 160 / 168 = n_bch - k_bch of bch_decoder.cpp:79-134
 * for all twelve codes), the standard's table 6a/7a polynomials g_2 and g_3, an independent big-integer encoder
 * (tests/t2_tx.py), and encode -> corrupt -> decode round trips.
 *
 * Arithmetic here deliberately shares nothing with the GPU kernel (no log tables, no remainder tables, no syndrome squaring). */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static uint32_t field_poly(int m) { return m == 16 ? 0x1002Du /* 1+x^2+x^3+x^5+x^16 */ : 0x402Bu /* 1+x+x^3+x^5+x^14 */; }

static uint32_t gmul(uint32_t a, uint32_t b, int m)
{
    const uint32_t poly = field_poly(m), top = 1u << m;
    uint32_t r = 0;
    while (b) {
        if (b & 1u) r ^= a;
        b >>= 1;
        a <<= 1;
        if (a & top) a ^= poly;
    }
    return r;
}

static uint32_t gpow(uint32_t a, uint32_t e, int m)
{
    uint32_t r = 1;
    while (e) {
        if (e & 1u) r = gmul(r, a, m);
        a = gmul(a, a, m);
        e >>= 1;
    }
    return r;
}

static uint32_t ginv(uint32_t a, int m) { return gpow(a, (1u << m) - 2u, m); }

/* minimal polynomial of alpha^j over GF(2), as bits (bit k = coefficient of x^k); alpha = x */
uint32_t ora_bch_minpoly(int m, int j)
{
    uint32_t conj[16], c[18];
    int n = 0;
    uint32_t b = gpow(2u, (uint32_t)j, m), x = b;
    do { conj[n++] = x; x = gmul(x, x, m); } while (x != b && n < 16);
    memset(c, 0, sizeof c);
    c[0] = 1;                                             /* product of (X + conj[i]) with coefficients in the field */
    for (int i = 0; i < n; ++i) {
        for (int k = i + 1; k >= 1; --k) c[k] = c[k - 1] ^ gmul(c[k], conj[i], m);
        c[0] = gmul(c[0], conj[i], m);
    }
    uint32_t bits = 0;
    for (int k = 0; k <= n; ++k) {
        if (c[k] > 1u) return 0;                          /* cannot happen: the product is over a full conjugacy class */
        bits |= c[k] << k;
    }
    return bits;
}

/* generator polynomial g(x) = m_1 m_3 ... m_{2t-1}; g[k] = coefficient of x^k, k <= m*t. Returns its degree. */
int ora_bch_generator(int m, int t, uint8_t *g)
{
    int deg = 0;
    memset(g, 0, (size_t)m * t + 1);
    g[0] = 1;
    for (int i = 0; i < t; ++i) {
        const uint32_t mp = ora_bch_minpoly(m, 2 * i + 1);
        int md = 0;
        for (int k = 0; k <= 16; ++k) if (mp >> k & 1u) md = k;
        uint8_t nxt[16 * 12 + 17];
        memset(nxt, 0, sizeof nxt);
        for (int a = 0; a <= deg; ++a)
            if (g[a])
                for (int k = 0; k <= md; ++k) if (mp >> k & 1u) nxt[a + k] ^= 1;
        deg += md;
        memcpy(g, nxt, (size_t)deg + 1);
    }
    return deg;
}

/* systematic encoding: bits[0..k) message (first bit = highest power), writes bits[k..n), n - k = deg g. Returns 0, -1 on a
 * (m, t, n - k) that does not fit. */
int ora_bch_encode(int m, int t, uint8_t *bits, int k, int n)
{
    uint8_t g[16 * 12 + 1], reg[16 * 12];
    if ((m != 14 && m != 16) || t < 1 || t > 12) return -1;
    const int r = ora_bch_generator(m, t, g);
    if (n - k != r) return -1;
    memset(reg, 0, sizeof reg);                           /* reg[i] = coefficient of x^i of the running remainder */
    for (int i = 0; i < k; ++i) {
        const uint8_t fb = (uint8_t)((bits[i] & 1u) ^ reg[r - 1]);
        for (int a = r - 1; a >= 1; --a) reg[a] = (uint8_t)(reg[a - 1] ^ (fb & g[a]));
        reg[0] = (uint8_t)(fb & g[0]);
    }
    for (int i = 0; i < r; ++i) bits[k + i] = reg[r - 1 - i];
    return 0;
}

/* decode one received word of n bits in place (bit i = coefficient of x^(n-1-i)). Returns the number of corrected bits (0 =
 * already a codeword), or -1 when more than t errors are detected (the word is left as it was). */
int ora_bch_decode(int m, int t, uint8_t *bits, int n)
{
    uint32_t S[24], C[26], B[26], T[26];
    if ((m != 14 && m != 16) || t < 1 || t > 12 || n < 1 || n > (1 << m) - 1) return -2;
    int any = 0;
    for (int j = 1; j <= 2 * t; ++j) {                    /* S_j = r(alpha^j), Horner from the highest power */
        const uint32_t aj = gpow(2u, (uint32_t)j, m);
        uint32_t s = 0;
        for (int i = 0; i < n; ++i) s = gmul(s, aj, m) ^ (bits[i] & 1u);
        S[j - 1] = s;
        any |= s != 0;
    }
    if (!any) return 0;
    /* Berlekamp-Massey over the 2t syndromes */
    memset(C, 0, sizeof C); memset(B, 0, sizeof B);
    C[0] = B[0] = 1;
    int L = 0, sh = 1;
    uint32_t b = 1;
    for (int k = 0; k < 2 * t; ++k) {
        uint32_t d = S[k];
        for (int i = 1; i <= L; ++i) d ^= gmul(C[i], S[k - i], m);
        if (d == 0) { ++sh; continue; }
        const uint32_t coef = gmul(d, ginv(b, m), m);
        if (2 * L <= k) {
            memcpy(T, C, sizeof T);
            for (int i = 0; i + sh < 26; ++i) C[i + sh] ^= gmul(coef, B[i], m);
            L = k + 1 - L;
            memcpy(B, T, sizeof B);
            b = d;
            sh = 1;
        } else {
            for (int i = 0; i + sh < 26; ++i) C[i + sh] ^= gmul(coef, B[i], m);
            ++sh;
        }
    }
    if (L > t) return -1;
    /* roots: an error at bit i (power e = n-1-i) makes alpha^(-e) a root of the locator */
    int pos[12], found = 0;
    const uint32_t order = (1u << m) - 1u;
    for (int i = 0; i < n; ++i) {
        const uint32_t e = (uint32_t)(n - 1 - i), x = gpow(2u, (order - e % order) % order, m);
        uint32_t v = 0;
        for (int k = L; k >= 0; --k) v = gmul(v, x, m) ^ C[k];
        if (v == 0) {
            if (found == L) return -1;
            pos[found++] = i;
        }
    }
    if (found != L) return -1;
    for (int i = 0; i < found; ++i) bits[pos[i]] ^= 1u;
    return found;
}
