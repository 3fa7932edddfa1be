/*
 * oracle/ldpc_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
 *
 * Scalar CPU restatement of the reference's LDPC stage for DVB-T2:
 *   - frame handling of ldpc_decoder::execute      (/root/reference/src/DVB_T2/ldpc_decoder.cpp:157-301)
 *   - layered schedule of LDPCDecoder              (/root/reference/src/DVB_T2/LDPC/layered_decoder.hh:65-110,115-180)
 *   - int8 offset-min-sum check-node rule          (/root/reference/src/DVB_T2/LDPC/algorithms.hh:221-292,
 *                                                   scalar spec LDPC/generic.hh:272-341, AVX2 ops LDPC/avx2.hh)
 *   - table walk of LDPC<TABLE>                    (/root/reference/src/DVB_T2/LDPC/ldpc.hh:39-123)
 *
 * Parity status: PINNED. tests/test_oracle_ldpc.py checks this file bit-for-bit (hard bits, trials-left and
 * every final a-posteriori LLR) against oracle/_ref/libref_ldpc.so, which is the reference's own
 * LDPC headers compiled unmodified (oracle/Makefile), and against tests/golden/ldpc_*.npz made from it.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into this file.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../sdr_receiver_dvb_t2_amd/csrc/tables/ldpc_tables_data.h"

#define GROUP 360

typedef struct {
    int n, k, r, q, cnl, links_total;
    uint16_t *pos;   /* [q][360][cnl] data-bit index per check node, layered (i,j) order */
    uint8_t  *cnc;   /* [r] data links per check node, ORIGINAL check order (see note in update) */
} ora_ldpc_code;

static ora_ldpc_code g_codes[T2_LDPC_NUM_CODES];
static int g_ready[T2_LDPC_NUM_CODES];

/* layered_decoder.hh:115-167 (init) + ldpc.hh:57-121 (bit -> accumulator addresses, +q per bit mod R). */
static ora_ldpc_code *ora_code(int id)
{
    if (id < 0 || id >= T2_LDPC_NUM_CODES) return NULL;
    ora_ldpc_code *c = &g_codes[id];
    if (g_ready[id]) return c;
    const t2_ldpc_table_t *t = &T2_LDPC_TABLES[id];
    c->n = t->n; c->k = t->k; c->r = t->n - t->k; c->q = c->r / GROUP;
    /* first pass: degrees per check in original numbering */
    c->cnc = (uint8_t *)calloc(c->r, 1);
    const uint16_t *a = t->addr;
    for (int g = 0; g < t->n_groups; ++g) {
        int deg = t->group_deg[g];
        for (int m = 0; m < GROUP; ++m)
            for (int e = 0; e < deg; ++e)
                c->cnc[(a[e] + m * c->q) % c->r]++;
        a += deg;
    }
    int maxc = 0;
    for (int i = 0; i < c->r; ++i) if (c->cnc[i] > maxc) maxc = c->cnc[i];
    c->cnl = maxc;                         /* == LINKS_MAX_CN - 2 */
    uint16_t *orig = (uint16_t *)calloc((size_t)c->r * c->cnl, sizeof(uint16_t));
    uint8_t *fill = (uint8_t *)calloc(c->r, 1);
    a = t->addr;
    int bit = 0;
    for (int g = 0; g < t->n_groups; ++g) {
        int deg = t->group_deg[g];
        for (int m = 0; m < GROUP; ++m, ++bit)
            for (int e = 0; e < deg; ++e) {
                int chk = (a[e] + m * c->q) % c->r;
                orig[(size_t)c->cnl * chk + fill[chk]++] = (uint16_t)bit;
            }
        a += deg;
    }
    free(fill);
    /* layered order: node (i,j) = original check q*j+i   (layered_decoder.hh:154-162) */
    c->pos = (uint16_t *)calloc((size_t)c->r * c->cnl, sizeof(uint16_t));
    for (int i = 0; i < c->q; ++i)
        for (int j = 0; j < GROUP; ++j)
            memcpy(&c->pos[(size_t)c->cnl * (GROUP * i + j)], &orig[(size_t)c->cnl * (c->q * j + i)],
                   c->cnl * sizeof(uint16_t));
    free(orig);
    long lt = 0;
    for (int i = 0; i < c->q; ++i)
        for (int j = 0; j < GROUP; ++j)
            lt += c->cnc[i] + 2 - !(i | j);
    c->links_total = (int)lt;
    g_ready[id] = 1;
    return c;
}

/* int8 saturating helpers (avx2.hh:379-385,443-449: _mm256_adds_epi8 / _mm256_subs_epi8) */
static inline int8_t sat8(int v) { return (int8_t)(v < -128 ? -128 : v > 127 ? 127 : v); }
/* vsign == _mm256_sign_epi8 (avx2.hh:535-541) */
static inline int8_t sign8(int8_t a, int8_t b) { return b < 0 ? (int8_t)-a : b > 0 ? a : 0; }

/* algorithms.hh:250-276 (finalp): offset beta = nearbyint(0.5*FACTOR) = 1, FACTOR = 2 (ldpc_decoder.h:38). */
static void cn_finalp(int8_t *links, int cnt)
{
    int8_t mags[32];
    for (int i = 0; i < cnt; ++i) {
        int v = links[i] < -127 ? -127 : links[i];      /* vqabs: abs(max(a,-127)) (avx2.hh:491-497) */
        v = v < 0 ? -v : v;
        v -= 1; if (v < 0) v = 0;                        /* vqsub on uint8 (avx2.hh:459-465) */
        mags[i] = (int8_t)v;
    }
    int8_t m0 = mags[0] < mags[1] ? mags[0] : mags[1];
    int8_t m1 = mags[0] < mags[1] ? mags[1] : mags[0];
    for (int i = 2; i < cnt; ++i) {
        int8_t t = m0 > mags[i] ? m0 : mags[i];
        if (t < m1) m1 = t;
        if (mags[i] < m0) m0 = mags[i];
    }
    int8_t signs = links[0];
    for (int i = 1; i < cnt; ++i) signs ^= links[i];
    for (int i = 0; i < cnt; ++i) {
        int8_t other = (mags[i] == m0) ? m1 : m0;
        links[i] = sign8(other, (int8_t)((signs ^ links[i]) | 127));
    }
}

typedef struct {
    int8_t *data;   /* [k] a-posteriori LLR of information bits */
    int8_t *pty;    /* [r] parity LLRs, pty[360*i+j] */
    int8_t *bnl;    /* [links_total] check->bit messages */
} ora_state;

/* layered_decoder.hh:65-82 (bad): true if any check's sign product is not strictly positive. */
static int cw_bad(const ora_ldpc_code *c, const ora_state *s)
{
    for (int i = 0; i < c->q; ++i) {
        int cnt = c->cnc[i];   /* indexed by LAYER as in the reference (cnc itself stays in original order;
                                  original check i is node (i, j=0), and every node of a layer has that degree) */
        for (int j = 0; j < GROUP; ++j) {
            int8_t cnv = sign8(1, s->pty[GROUP * i + j]);
            if (i) cnv = sign8(cnv, s->pty[GROUP * (i - 1) + j]);
            else if (j) cnv = sign8(cnv, s->pty[j + (c->q - 1) * GROUP - 1]);
            const uint16_t *p = &c->pos[(size_t)c->cnl * (GROUP * i + j)];
            for (int k = 0; k < cnt; ++k) cnv = sign8(cnv, s->data[p[k]]);
            if (cnv <= 0) return 1;
        }
    }
    return 0;
}

/* layered_decoder.hh:83-110 (update): nodes visited strictly in (i, j) order. */
static void cw_update(const ora_ldpc_code *c, ora_state *s)
{
    int8_t *bl = s->bnl;
    for (int i = 0; i < c->q; ++i) {
        int cnt = c->cnc[i];
        for (int j = 0; j < GROUP; ++j) {
            int deg = cnt + 2 - !(i | j);
            int8_t inp[32], out[32];
            const uint16_t *p = &c->pos[(size_t)c->cnl * (GROUP * i + j)];
            for (int k = 0; k < cnt; ++k) inp[k] = out[k] = sat8(s->data[p[k]] - bl[k]);
            int8_t *p0 = &s->pty[GROUP * i + j], *p1 = NULL;
            inp[cnt] = out[cnt] = sat8(*p0 - bl[cnt]);
            if (i) p1 = &s->pty[GROUP * (i - 1) + j];
            else if (j) p1 = &s->pty[j + (c->q - 1) * GROUP - 1];
            if (p1) inp[cnt + 1] = out[cnt + 1] = sat8(*p1 - bl[cnt + 1]);
            cn_finalp(out, deg);
            for (int k = 0; k < cnt; ++k) s->data[p[k]] = sat8(inp[k] + out[k]);
            *p0 = sat8(inp[cnt] + out[cnt]);
            if (p1) *p1 = sat8(inp[cnt + 1] + out[cnt + 1]);
            for (int d = 0; d < deg; ++d) {            /* algorithms.hh:288-291: clamp stored message */
                int8_t v = out[d];
                bl[d] = v < -32 ? -32 : v > 31 ? 31 : v;
            }
            bl += deg;
        }
    }
}

/*
 * Decode `blocks` frames the way ldpc_decoder::execute + LDPCDecoder::operator() do for one SIMD batch:
 *   llr_in  [blocks][n]  received LLRs, transmitted order (k info, then parity with in[k+360*t+s];
 *                        ldpc_decoder.cpp:253-258 and layered_decoder.hh:171-173 cancel, so pty[360*t+s] = in[k+360*t+s])
 *   bits_out[blocks][k]  one bit per byte, written only when the batch converged (ldpc_decoder.cpp:264-277)
 *   llr_out [blocks][n]  optional: final a-posteriori LLRs in the same order as llr_in (always written)
 * returns trials left (>= 0) or -1 when any frame of the batch still fails after max_trials updates.
 */
int ora_ldpc_decode(int code_id, const int8_t *llr_in, int blocks, int max_trials,
                    uint8_t *bits_out, int8_t *llr_out)
{
    ora_ldpc_code *c = ora_code(code_id);
    if (!c || blocks < 1) return -2;
    ora_state *st = (ora_state *)calloc(blocks, sizeof(ora_state));
    for (int b = 0; b < blocks; ++b) {
        st[b].data = (int8_t *)malloc(c->k);
        st[b].pty = (int8_t *)malloc(c->r);
        st[b].bnl = (int8_t *)calloc(c->links_total, 1);          /* reset(): messages start at 0 */
        memcpy(st[b].data, llr_in + (size_t)b * c->n, c->k);
        memcpy(st[b].pty, llr_in + (size_t)b * c->n + c->k, c->r);
    }
    int trials = max_trials;
    for (;;) {                                                     /* while (bad() && --trials >= 0) update(); */
        int bad = 0;
        for (int b = 0; b < blocks && !bad; ++b) bad = cw_bad(c, &st[b]);
        if (!bad) break;
        if (--trials < 0) break;
        for (int b = 0; b < blocks; ++b) cw_update(c, &st[b]);
    }
    for (int b = 0; b < blocks; ++b) {
        if (llr_out) {
            memcpy(llr_out + (size_t)b * c->n, st[b].data, c->k);
            memcpy(llr_out + (size_t)b * c->n + c->k, st[b].pty, c->r);
        }
        if (trials >= 0 && bits_out)
            for (int i = 0; i < c->k; ++i) bits_out[(size_t)b * c->k + i] = st[b].data[i] < 0;
        free(st[b].data); free(st[b].pty); free(st[b].bnl);
    }
    free(st);
    return trials;
}

int ora_ldpc_params(int code_id, int *n, int *k, int *q, int *links_total)
{
    ora_ldpc_code *c = ora_code(code_id);
    if (!c) return -1;
    if (n) *n = c->n;
    if (k) *k = c->k;
    if (q) *q = c->q;
    if (links_total) *links_total = c->links_total;
    return 0;
}

/*
 * Systematic encoder (test-vector generator, not part of the reference receiver): ETSI EN 302 755 6.1.2 --
 * accumulate info bits at the table addresses, then running XOR; emitted in the transmitted order used above
 * (parity interleaver 6.1.3 part 1: u[k+360*t+s] = p[q*s+t]).
 */
int ora_ldpc_encode(int code_id, const uint8_t *info_bits, uint8_t *cw_bits)
{
    ora_ldpc_code *c = ora_code(code_id);
    if (!c) return -1;
    const t2_ldpc_table_t *t = &T2_LDPC_TABLES[code_id];
    uint8_t *p = (uint8_t *)calloc(c->r, 1);
    const uint16_t *a = t->addr;
    int bit = 0;
    for (int g = 0; g < t->n_groups; ++g) {
        int deg = t->group_deg[g];
        for (int m = 0; m < GROUP; ++m, ++bit)
            if (info_bits[bit] & 1)
                for (int e = 0; e < deg; ++e) p[(a[e] + m * c->q) % c->r] ^= 1;
        a += deg;
    }
    for (int i = 1; i < c->r; ++i) p[i] ^= p[i - 1];
    for (int i = 0; i < c->k; ++i) cw_bits[i] = info_bits[i] & 1;
    for (int tt = 0; tt < c->q; ++tt)
        for (int s = 0; s < GROUP; ++s) cw_bits[c->k + GROUP * tt + s] = p[c->q * s + tt];
    free(p);
    return 0;
}
