/*
 * oracle/l1_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU checker; never linked into or called by the product library).
 * Plain-C restatement of the reference's L1 signalling extraction, written as the reference writes it -- fixed bit positions
 * plus the idx_l1_post_*_shift offsets, not a sequential reader -- so that its quirks are kept:
 *   p2_symbol::l1_pre_info   /root/reference/src/DVB_T2/p2_symbol.cpp:301-532
 *   p2_symbol::l1_post_info  :534-718 and the field parsers :720-1089
 * Pinned against the reference's own class (oracle/_ref/libref_t2sym.so, fixtures tests/golden/t2sym_golden.npz: L1-pre and
 * L1-post of five modes, QPSK L1-post). Outputs use the reference's struct order (dvbt2_definition.h:249-343).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define L1_PRE_CELL 1840

static unsigned crc32_step(unsigned crc, unsigned bit)                       /* :308-314 */
{
    unsigned b = bit ^ ((crc >> 31) & 0x01u);
    crc <<= 1;
    if (b) crc ^= 0x04C11DB7u;
    return crc;
}

static int field(const unsigned char *bit, int *idx, int n)
{
    int v = 0;
    for (int s = n - 1; s >= 0; --s) v |= bit[(*idx)++] << s;
    return v;
}

/* cells: the de-interleaved P2 cells (re/im interleaved). out29: l1_presignalling in struct order (:249-279), crc_32 last.
 * Returns 1 when the CRC-32 matches, 0 for "CRC_32 ERROR" (then out29 holds only crc_32, as in the reference). */
int ora_l1_pre(const float *cells, int *out29)
{
    unsigned crc = 0xffffffffu;
    unsigned char bit[200];
    for (int i = 0; i < 200; ++i) bit[i] = cells[2 * i] > 0 ? 0 : 1;          /* BPSK demodulate (:309,319,337) */
    for (int i = 0; i < 168; ++i) crc = crc32_step(crc, bit[i]);
    int idx = 168;
    unsigned f = (unsigned)field(bit, &idx, 32);
    memset(out29, 0, sizeof(int) * 29);
    out29[28] = (int)f;
    if (crc != f) return 0;
    static const int width[28] = { 8, 1, 3, 3, 1, 1, 3, 4, 4, 2, 2, 18, 18, 4, 8, 16, 16, 16, 8, 12, 3, 1, 3, 3, 4, 1, 1, 4 };   /* :339-491 */
    idx = 0;
    for (int k = 0; k < 28; ++k) out29[k] = field(bit, &idx, width[k]);
    return 1;
}

/* L1-post. cells: the P2 cells (the L1-post block starts at cell 1840). pre29: ora_l1_pre's output. out: flat ints in the
 * order oracle/ref_t2sym.cpp packs the reference's struct: 16 scalars (sub_slices_per_frame, num_plp, num_aux, aux_config_rfu,
 * fef_type, fef_length, fef_interval, fef_length_msb, reserved_2, dyn.frame_idx, dyn.sub_slice_interval, dyn.type_2_start,
 * dyn.l1_change_counter, dyn.start_rf_idx, dyn.reserved_1, dyn.reserved_3), num_rf, (rf_idx, frequency) per RF, 21 + 4 ints per
 * PLP, (aux_stream_type, aux_private_config) per AUX. Returns the number of ints written, 0 for "CRC_32 ERROR", -1 when the
 * reference would index outside its buffers (it does not check). */
int ora_l1_post(const float *cells, const int *pre29, int *out, int out_cap)
{
    static const int mux16[8] = { 7, 1, 3, 5, 2, 4, 6, 0 }, mux64[12] = { 11, 8, 5, 2, 10, 7, 4, 1, 9, 6, 3, 0 }, m1[1] = { 0 };
    const int s2_field2 = pre29[4], l1_repetition_flag = pre29[5], l1_post_mod = pre29[8], l1_post_size = pre29[11],
              l1_post_info_size = pre29[12], num_rf = pre29[22], t2_version = pre29[24], l1_post_scrambled = pre29[25];
    const float *p = cells + 2 * L1_PRE_CELL;
    float amp2 = 0, amp4 = 0;
    int n_bit = 0, n_post = 0, rows = 0, colums = 0, substreams = 1;
    const int *mux = m1;
    switch (l1_post_mod) {                                                   /* :556-590 */
    case 0: n_bit = 0; n_post = l1_post_size; break;
    case 1: n_bit = 1; n_post = l1_post_size * 2; break;
    case 2: amp4 = 0.316227766f * 2; n_bit = 3; n_post = l1_post_size * 4; colums = 8; rows = n_post / colums; mux = mux16; substreams = 8; break;
    case 3: amp2 = 0.15430335f * 2; amp4 = 0.15430335f * 4; n_bit = 5; n_post = l1_post_size * 6; colums = 12; rows = n_post / colums;
            mux = mux64; substreams = 12; break;
    default: return -1;
    }
    /* the reference's buffers hold l1_post_size * l1_post_mod * 2 bytes (:405-409): BPSK gets none, 64-QAM exactly n_post */
    if (l1_post_size * l1_post_mod * 2 < n_post || l1_post_info_size + 32 > n_post) return -1;
    unsigned char *inter = (unsigned char *)calloc((size_t)n_post + 64, 1), *bit = (unsigned char *)calloc((size_t)n_post + 64, 1);
    int c_bit = n_bit, idx_mux = 0, w = 0;
    float real = p[0], imag = p[1];
    for (int i = 0; i < n_post; ++i) {                                       /* :596-646 */
        unsigned char b = 0;
        switch (n_bit - c_bit) {
        case 0: b = real > 0 ? 0 : 1; break;
        case 1: b = imag > 0 ? 0 : 1; break;
        case 2: b = fabsf(real) > amp4 ? 0 : 1; break;
        case 3: b = fabsf(imag) > amp4 ? 0 : 1; break;
        case 4: b = fabsf(fabsf(real) - amp4) > amp2 ? 0 : 1; break;
        case 5: b = fabsf(fabsf(imag) - amp4) > amp2 ? 0 : 1; break;
        }
        inter[mux[idx_mux] + w] = b;
        if (++idx_mux == substreams) { idx_mux = 0; w += substreams; }
        if (c_bit == 0) { c_bit = n_bit + 1; p += 2; real = p[0]; imag = p[1]; }
        --c_bit;
    }
    const int size_block = rows * colums;
    const int randomize = t2_version > 1 && l1_post_scrambled == 1;
    int sr = 0x4A80, step = 0, l = 0;                                        /* init_l1_randomizer (:78-87): the BB scrambler's sequence */
    unsigned char *rnd = (unsigned char *)malloc((size_t)n_post + 64);
    for (int i = 0; i < n_post; ++i) { int b = (sr ^ (sr >> 1)) & 1; rnd[i] = (unsigned char)b; sr >>= 1; if (b) sr |= 0x4000; }
    for (int i = 0; i < n_post; ++i) {                                       /* :649-660 */
        int j = l + step;
        bit[j] = inter[i];
        if (randomize) bit[j] ^= rnd[j];
        step += rows;
        if (step == size_block) { step = 0; ++l; }
    }
    unsigned crc = 0xffffffffu;
    for (int i = 0; i < l1_post_info_size; ++i) crc = crc32_step(crc, bit[i]);
    int idx = l1_post_info_size;
    unsigned f = (unsigned)field(bit, &idx, 32);
    int n_out = 0;
    if (f == crc) {
        /* offsets exactly as the reference derives them (:364,464,680-697) */
        const int fef_shift = s2_field2 * 34, rf_shift = (num_rf - 1) * 35;
        idx = 15;
        const int num_plp = field(bit, &idx, 8);
        idx = 23;
        const int num_aux = field(bit, &idx, 4);
        const int plp_shift = (num_plp - 1) * 89, aux_shift = (num_aux - 1) * 32, dyn_aux_shift = (num_aux - 1) * 48;
        const int conf_shift = rf_shift + fef_shift + plp_shift + aux_shift + 223;
        const int dyn_plp_shift = (num_plp - 1) * 48;
        (void)dyn_aux_shift; (void)l1_repetition_flag;
        const int need = 17 + 2 * num_rf + 25 * num_plp + 2 * num_aux;
        if (need > out_cap || conf_shift + 71 + 48 * num_plp > n_post) { n_out = -1; goto done; }
        int *o = out;
        idx = 0;
        o[0] = field(bit, &idx, 15);                                         /* time_frequency_slicing_info (:720-728) */
        o[1] = num_plp; o[2] = num_aux;
        idx = 27; o[3] = field(bit, &idx, 8);                                /* aux_info (:899-903) */
        o[4] = o[5] = o[6] = 0;                                              /* fef_info (:863-895): FEF_TYPE read as FIVE bits (:868) */
        if (s2_field2 == 1) { idx = rf_shift + 70; o[4] = field(bit, &idx, 5); o[5] = field(bit, &idx, 22); o[6] = field(bit, &idx, 8); }
        idx = rf_shift + fef_shift + plp_shift + 169; o[7] = field(bit, &idx, 2); o[8] = field(bit, &idx, 2);   /* reserved_2: 2 of its 30 bits (:890-893) */
        idx = conf_shift;                                                    /* dyn_info (:924-956) */
        o[9] = field(bit, &idx, 8); o[10] = field(bit, &idx, 22); o[11] = field(bit, &idx, 22); o[12] = field(bit, &idx, 8);
        o[13] = field(bit, &idx, 3); o[14] = field(bit, &idx, 8);
        idx = conf_shift + dyn_plp_shift + 119;                              /* dyn_aux_info (:990-1004): `=` not `|=`, so only the last bit survives */
        o[15] = 0;
        for (int s = 7; s >= 0; --s) o[15] = bit[idx++] << s;
        o[16] = num_rf;
        o += 17;
        idx = 35;                                                            /* rf_info (:844-860) */
        for (int i = 0; i < num_rf; ++i) { *o++ = field(bit, &idx, 3); *o++ = field(bit, &idx, 32); }
        int pidx = rf_shift + fef_shift + 70, didx = conf_shift + 71;        /* plp_info (:731-842), dyn_plp_info (:959-987) */
        static const int pw[21] = { 8, 3, 5, 1, 3, 8, 8, 3, 3, 1, 2, 10, 8, 8, 1, 1, 1, 11, 2, 1, 1 };
        for (int i = 0; i < num_plp; ++i) {
            for (int k = 0; k < 21; ++k) *o++ = field(bit, &pidx, pw[k]);
            *o++ = field(bit, &didx, 8); *o++ = field(bit, &didx, 22); *o++ = field(bit, &didx, 10); *o++ = field(bit, &didx, 8);
        }
        idx = rf_shift + fef_shift + plp_shift + 191;                        /* aux_info (:905-919) */
        for (int i = 0; i < num_aux; ++i) { *o++ = field(bit, &idx, 4); *o++ = field(bit, &idx, 28); }
        n_out = (int)(o - out);
    }
done:
    free(inter); free(bit); free(rnd);
    return n_out;
}
