/*
 * oracle/fec_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
 *
 * CPU restatement of the reference's FEC-side stages around the LDPC decoder:
 *   BB descrambler / BCH stub  bch_decoder::execute, init_descrambler  (/root/reference/src/DVB_T2/bch_decoder.cpp:50-61,63-164)
 *   bit de-interleave tables   llr_demapper ctor + address_generator     (/root/reference/src/DVB_T2/llr_demapper.cpp:29-90,110-130)
 *   LLR demapper               llr_demapper::qpsk/qam16/qam64/qam256      (llr_demapper.cpp:160-228,230-364,366-535,537-768,770-776)
 *   cell de-interleaver table  time_deinterleaver::address_cell_deinterleaving (/root/reference/src/DVB_T2/time_deinterleaver.cpp:174-266)
 *   time de-interleaver        time_deinterleaver::l1_dyn_execute/execute (time_deinterleaver.cpp:268-376)
 *
 * Parity status: UNPINNED. These reference files need Qt (QObject/QThread/... headers) to compile; Qt is not in this
 * image and the rules forbid stand-in headers, so the reference stages cannot be run here, and the reference ships no
 * tests or vectors of its own. What pins this file instead: (i) the integer tables are cross-checked against an
 * independent forward (transmitter-side) construction from ETSI EN 302 755 in tests/t2_tx.py, (ii) end-to-end: frames
 * built by that transmitter model decode to the sent transport stream through these stages.
 *
 * Floating point: the reference is built -Ofast (no FMA: -mavx2 only). This file is built -O2 -ffp-contract=off and writes
 * every product/sum in source order.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into this file.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../sdr_receiver_dvb_t2_amd/csrc/tables/bitint_tables_data.h"

/* ---------------------------------------------------------------- BB descrambler (bch_decoder.cpp:50-61) */
void ora_bb_prbs(uint8_t *out, int n)
{
    int sr = 0x4A80;
    for (int i = 0; i < n; i++) {
        uint8_t b = ((sr) ^ (sr >> 1)) & 1;
        out[i] = b;
        sr >>= 1;
        if (b) sr |= 0x4000;
    }
}

static const int K_BCH[12] = {7032, 9552, 10632, 11712, 12432, 13152, 32208, 38688, 43040, 48408, 51648, 53840};
static const int K_LDPC[12] = {7200, 9720, 10800, 11880, 12600, 13320, 32400, 38880, 43200, 48600, 51840, 54000};

/* bch_decoder::execute (bch_decoder.cpp:136-142): no BCH decoding -- keep the first k_bch bits of every n_bch = k_ldpc
 * block and XOR the PRBS. in [n_frames][k_ldpc] one bit per byte -> out [n_frames][k_bch]. Returns k_bch. */
int ora_bch_descramble(int code_id, const uint8_t *in, int n_frames, uint8_t *out)
{
    if (code_id < 0 || code_id > 11) return -1;
    const int kb = K_BCH[code_id], nb = K_LDPC[code_id];
    uint8_t *prbs = (uint8_t *)malloc(54000);
    ora_bb_prbs(prbs, 54000);
    for (int f = 0; f < n_frames; ++f)
        for (int i = 0; i < kb; ++i) out[(size_t)f * kb + i] = in[(size_t)f * nb + i] ^ prbs[i];
    free(prbs);
    return kb;
}

/* ---------------------------------------------------------------- bit de-interleaver address (llr_demapper.cpp:110-130) */
static const t2_bitint_cfg_t *bitint_cfg(int mod, int fec_type, int code_rate)
{
    /* selection exactly as qam16/qam64/qam256 do it (llr_demapper.cpp:288-296,447-455,675-686) */
    const int bpc = 2 * (mod + 1);
    int rate_sel = -1;
    if (fec_type == 1) {
        if (code_rate == 1) rate_sel = 1;                       /* C3_5 */
        else if (code_rate == 2 && mod == 3) rate_sel = 2;      /* C2_3, 256-QAM only */
    }
    for (int i = 0; i < T2_BITINT_NUM_CFG; ++i) {
        const t2_bitint_cfg_t *c = &T2_BITINT_CFG[i];
        if (c->bits_per_cell == bpc && c->fec_normal == fec_type && c->rate_sel == rate_sel) return c;
    }
    return NULL;
}

int ora_bitdeint_address(int mod, int fec_type, int code_rate, int *address_out)
{
    const t2_bitint_cfg_t *c = bitint_cfg(mod, fec_type, code_rate);
    if (!c) return -1;
    const int column = c->columns, row = c->rows, frame = column * row;
    int *address = (int *)malloc(sizeof(int) * frame);
    for (int cc = 0; cc < column; ++cc)
        for (int r = 0; r < row; ++r) address[cc * row + r] = column * r + (cc + column - c->twist[r]) % column;
    int k = 0, n = 0;
    for (int i = 0; i < frame; ++i) {
        address_out[i] = address[c->demux[n] + k];
        ++n;
        if (n == row) { n = 0; k += row; }
    }
    free(address);
    return frame;
}

/* ---------------------------------------------------------------- LLR demapper */
#define ROT_QPSK 0.506145483f
#define ROT_QAM16 0.293215314f
#define ROT_QAM64 0.150098316f
#define ROT_QAM256 0.062418810f
#define NORM_FACTOR_QPSK 0.707106781f
#define NORM_FACTOR_QAM16 0.316227766f
#define NORM_FACTOR_QAM64 0.15430335f
#define NORM_FACTOR_QAM256 0.076696499f

static float slice_axis_tree(int mod, float x, float d)
{
    /* binary trees of llr_demapper.cpp:257-276 (16), :395-436 (64, incl. the '>' typo of :407,427), :567-654 (256) */
    float x2 = d * 2.0f, x4 = d * 4.0f, x6 = d * 6.0f, x8 = d * 8.0f, x10 = d * 10.0f, x12 = d * 12.0f, x14 = d * 14.0f;
    if (mod == 1) {
        if (x > 0) return (x > x2) ? d * 3.0f : d;
        return (x < -x2) ? -(d * 3.0f) : -d;
    }
    if (mod == 2) {
        if (x > 0) {
            if (x > x4) return (x > x6) ? d * 7.0f : d * 5.0f;
            return (x > x2) ? d * 3.0f : d;
        }
        if (x < -x4) return (x > x6) ? -(d * 7.0f) : -(d * 5.0f);     /* as written: never -7d */
        return (x < -x2) ? -(d * 3.0f) : -d;
    }
    /* 256-QAM */
    if (x > 0) {
        if (x > x8) {
            if (x > x12) return (x > x14) ? d * 15.0f : d * 13.0f;
            return (x > x10) ? d * 11.0f : d * 9.0f;
        }
        if (x > x4) return (x > x6) ? d * 7.0f : d * 5.0f;
        return (x > x2) ? d * 3.0f : d;
    }
    if (x < -x8) {
        if (x < -x12) return (x < -x14) ? -(d * 15.0f) : -(d * 13.0f);
        return (x < -x10) ? -(d * 11.0f) : -(d * 9.0f);
    }
    if (x < -x4) return (x < -x6) ? -(d * 7.0f) : -(d * 5.0f);
    return (x < -x2) ? -(d * 3.0f) : -d;
}

/* The same decisions without branches (what -Ofast -mavx2 makes of the trees is compares and blends, not jumps; on noisy cells the
 * jumps of the literal form above mispredict every other cell and the restatement ran 3x slower than the code it restates): the tree
 * picks the level by counting the thresholds below |x| -- d * (2 level + 1), the same float product the tree returns -- x = 0 and NaN
 * go with the negatives, and on the negative side of 64-QAM the outermost point is never decided (the `x > x6` of :407,427).
 * ora_slice_selfcheck() holds the two forms to each other; the fixtures hold both to the reference. */
static inline float slice_axis(int mod, float x, float d)
{
    const float a = fabsf(x);
    int lvl = 0;
    if (mod >= 1) lvl += a > d * 2.0f;
    if (mod >= 2) { lvl += a > d * 4.0f; lvl += a > d * 6.0f; }
    if (mod >= 3) { lvl += a > d * 8.0f; lvl += a > d * 10.0f; lvl += a > d * 12.0f; lvl += a > d * 14.0f; }
    const int pos = x > 0;
    if (mod == 2 && !pos && lvl > 2) lvl = 2;
    const float amp = d * (float)(2 * lvl + 1);
    return pos ? amp : -amp;
}

/* number of disagreements between the two forms on a dense sweep of x for every modulation (0 expected) */
int ora_slice_selfcheck(void)
{
    static const float NORM[4] = {0.707106781f, 0.316227766f, 0.15430335f, 0.076696499f};
    int bad = 0;
    for (int mod = 1; mod <= 3; ++mod) {
        const float d = NORM[mod];
        for (int k = -40000; k <= 40000; ++k) {
            const float x = (float)k * (d / 2000.0f);
            const float t = slice_axis_tree(mod, x, d), f = slice_axis(mod, x, d);
            if (!(t == f)) ++bad;
        }
        for (int m = 1; m <= 15; ++m) {                      /* the thresholds themselves and their neighbours */
            const float th = d * (float)m;
            const float xs[6] = {th, -th, nextafterf(th, 0.0f), nextafterf(th, 100.0f), -nextafterf(th, 0.0f), -nextafterf(th, 100.0f)};
            for (int q = 0; q < 6; ++q) if (!(slice_axis_tree(mod, xs[q], d) == slice_axis(mod, xs[q], d))) ++bad;
        }
        if (!(slice_axis_tree(mod, NAN, d) == slice_axis(mod, NAN, d)) || !(slice_axis_tree(mod, 0.0f, d) == slice_axis(mod, 0.0f, d))) ++bad;
    }
    return bad;
}

static int8_t cast_i8(float v)      /* static_cast<int8_t>(float) as g++ -Ofast -mavx2 emits it: cvttss2si, low byte */
{
    return (int8_t)(uint8_t)((int32_t)v & 0xff);
}

/*
 * One TI block through llr_demapper::execute. cells: interleaved (re, im), n_cells of them, de-rotated IN PLACE when
 * rotation != 0 (llr_demapper.cpp:555-557). out: [n_cells / cells_per_fec][fec_size] int8, frame-major -- the order in
 * which the reference fills its 32-frame buffers. sums[0..1] = (sum_s, sum_e); sums[2] = precision used.
 * precision_override > 0 replaces the measured 8*norm*sum_s/sum_e (for exact-LLR tests of the GPU kernel).
 * Returns the number of FEC frames produced.
 */
int ora_demap(int mod, int fec_type, int code_rate, int rotation, float *cells, int n_cells, int8_t *out, float *sums,
              float precision_override)
{
    static const float ROT[4] = {ROT_QPSK, ROT_QAM16, ROT_QAM64, ROT_QAM256};
    static const float NORM[4] = {NORM_FACTOR_QPSK, NORM_FACTOR_QAM16, NORM_FACTOR_QAM64, NORM_FACTOR_QAM256};
    if (mod < 0 || mod > 3) return -1;
    const int fec_size = fec_type == 1 ? 64800 : 16200;
    const int bpc = 2 * (mod + 1);
    const int cells_per_fec = fec_size / bpc;
    const float d = NORM[mod];
    if (rotation) {
        const float c = (float)cos(-(double)ROT[mod]), s = (float)sin(-(double)ROT[mod]);
        for (int i = 0; i < n_cells; ++i) {
            float re = cells[2 * i], im = cells[2 * i + 1];
            cells[2 * i] = re * c - im * s;
            cells[2 * i + 1] = re * s + im * c;
        }
    }
    float sum_s = 0, sum_e = 0;
    const int n_snr = (mod == 0) ? (n_cells < 2048 ? n_cells : 2048) : n_cells;
    for (int i = 0; i < n_snr; ++i) {
        float re = cells[2 * i], im = cells[2 * i + 1], sr, si;
        if (mod == 0) { sr = re > 0 ? d : -d; si = im > 0 ? d : -d; }
        else { sr = slice_axis(mod, re, d); si = slice_axis(mod, im, d); }
        float er = re - sr, ei = im - si;
        sum_s += sr * sr + si * si;
        sum_e += er * er + ei * ei;
    }
    float precision = 8.0f * d * sum_s / sum_e;
    if (precision_override > 0) precision = precision_override;
    if (sums) { sums[0] = sum_s; sums[1] = sum_e; sums[2] = precision; }
    const int n_frames = n_cells / cells_per_fec;
    if (mod == 0) {     /* no bit interleaver for QPSK here; quantize() clamps (llr_demapper.cpp:199-204,770-776) */
        for (int i = 0; i < n_frames * cells_per_fec; ++i)
            for (int a = 0; a < 2; ++a) {
                float b = nearbyintf(cells[2 * i + a] * precision);
                b = b < -128.0f ? -128.0f : (b > 127.0f ? 127.0f : b);
                out[2 * (size_t)i + a] = (int8_t)b;
            }
        return n_frames;
    }
    int *address = (int *)malloc(sizeof(int) * fec_size);
    if (ora_bitdeint_address(mod, fec_type, code_rate, address) != fec_size) { free(address); return -1; }
    const int levels = mod + 1;                 /* LLR pairs per cell: (re, im) per level */
    for (int f = 0; f < n_frames; ++f) {
        int8_t *o = out + (size_t)f * fec_size;
        for (int cidx = 0; cidx < cells_per_fec; ++cidx) {
            const float *cell = cells + 2 * ((size_t)f * cells_per_fec + cidx);
            const int *a = address + cidx * bpc;
            float v[2] = {cell[0], cell[1]};
            float thr = d * (float)(1 << mod);      /* 2d (16), 4d (64), 8d (256): norm_*_x2 / _x4 / _x8 */
            for (int l = 0; l < levels; ++l) {
                for (int ax = 0; ax < 2; ++ax) o[a[2 * l + ax]] = cast_i8(nearbyintf(v[ax] * precision));
                for (int ax = 0; ax < 2; ++ax) v[ax] = fabsf(v[ax]) - thr;
                thr = thr * 0.5f;
            }
        }
    }
    free(address);
    return n_frames;
}

/* ---------------------------------------------------------------- cell de-interleaver permutation (time_deinterleaver.cpp:174-266) */
int ora_cell_perm(int num_fec_block_max, int cells_per_fec_block, int *permutations)
{
    int block_max = num_fec_block_max, cells_size = cells_per_fec_block;
    int pn_degree = (int)ceil(log2((double)cells_size));
    int max_states = 1 << pn_degree;
    static const int logic11[2] = {0, 3}, logic12[2] = {0, 2}, logic13[4] = {0, 1, 4, 6}, logic14[6] = {0, 1, 4, 5, 9, 11},
                     logic15[4] = {0, 1, 2, 12};
    const int *logic; int xor_size, pn_mask;
    switch (pn_degree) {
    case 11: logic = logic11; xor_size = 2; pn_mask = 0x3ff; break;
    case 12: logic = logic12; xor_size = 2; pn_mask = 0x7ff; break;
    case 13: logic = logic13; xor_size = 4; pn_mask = 0xfff; break;
    case 14: logic = logic14; xor_size = 6; pn_mask = 0x1fff; break;
    case 15: logic = logic15; xor_size = 4; pn_mask = 0x3fff; break;
    default: logic = logic14; xor_size = 6; pn_mask = 0x1fff; break;
    }
    int *first = (int *)malloc(sizeof(int) * max_states);
    int lfsr = 0, q = 0;
    for (int i = 0; i < max_states; ++i) {
        if (i == 0 || i == 1) lfsr = 0;
        else if (i == 2) lfsr = 1;
        else {
            int result = 0;
            for (int k = 0; k < xor_size; ++k) result ^= (lfsr >> logic[k]) & 1;
            lfsr &= pn_mask;
            lfsr >>= 1;
            lfsr |= result << (pn_degree - 2);
        }
        lfsr |= (i % 2) << (pn_degree - 1);
        if (lfsr < cells_size) first[q++] = lfsr;
    }
    int n = 0, index = 0, address = 0;
    for (int r = 0; r < block_max; r++) {
        int shift = cells_size;
        while (shift >= cells_size) {
            int temp = n;
            shift = 0;
            for (int p = 0; p < pn_degree; ++p) { shift |= temp & 1; shift <<= 1; temp >>= 1; }
            n++;
        }
        for (int w = 0; w < cells_size; ++w) permutations[((first[w] + shift) % cells_size) + index] = address++;
        index += cells_size;
    }
    free(first);
    return q;
}

/* ---------------------------------------------------------------- time de-interleaver (single PLP, TI type 0, one TI block per frame) */
typedef struct {
    int cells_per_fec, rows, cols, ti_block_size;
    int *perm;
    int idx_step_ti, idx_row_ti;
    float q_first; int end_cell;          /* q_first_cell_fec_block / end_cell_fec_block: persist across TI blocks */
} ora_ti;

ora_ti *ora_ti_create(int cells_per_fec, int num_blocks_max)
{
    ora_ti *t = (ora_ti *)calloc(1, sizeof(ora_ti));
    t->cells_per_fec = cells_per_fec;
    t->rows = cells_per_fec / 5;                                   /* n_split = 5 (time_deinterleaver.cpp:69-113) */
    t->perm = (int *)malloc(sizeof(int) * (size_t)num_blocks_max * cells_per_fec);
    ora_cell_perm(num_blocks_max, cells_per_fec, t->perm);
    return t;
}
void ora_ti_destroy(ora_ti *t) { if (t) { free(t->perm); free(t); } }

/* l1_dyn_execute (time_deinterleaver.cpp:268-286): geometry of the TI block for `num_blocks` FEC blocks */
void ora_ti_begin(ora_ti *t, int num_blocks)
{
    t->cols = num_blocks * 5;
    t->ti_block_size = t->cols * t->rows;
    t->idx_step_ti = 0;
    t->idx_row_ti = 0;
}

/* execute (time_deinterleaver.cpp:316-345): scatter `n` cells (re, im interleaved) into out (ti_block_size cells).
 * Returns 1 when the TI block completed with the last cell consumed (the reference then emits ti_block). */
int ora_ti_push(ora_ti *t, const float *cells, int n, float *out)
{
    int done = 0;
    for (int i = 0; i < n; ++i) {
        int d = t->idx_step_ti + t->idx_row_ti;
        int i_address = t->perm[d];
        int q_address = i_address - 1;
        if (i_address % t->cells_per_fec == 0) {
            if (i_address != 0) out[2 * t->end_cell + 1] = t->q_first;
            t->q_first = cells[2 * i + 1];
            t->end_cell = q_address + t->cells_per_fec;
        } else {
            out[2 * q_address + 1] = cells[2 * i + 1];
        }
        out[2 * i_address] = cells[2 * i];
        t->idx_step_ti += t->rows;
        if (t->idx_step_ti == t->ti_block_size) {
            out[2 * t->end_cell + 1] = t->q_first;
            t->idx_step_ti = 0;
            if (++t->idx_row_ti == t->rows) { t->idx_row_ti = 0; done = 1; }
        }
    }
    return done;
}

/* ---------------------------------------------------------------- frame de-multiplexer of the time de-interleaver (any number of PLPs)
 * The bookkeeping of time_deinterleaver::start (time_deinterleaver.cpp:38-145), l1_dyn_execute (:268-286) and execute
 * (:288-376) walked cell by cell exactly as the reference does, without moving any cell: reports for which PLP, from which
 * cell and with which size ti_block is emitted (:349). num_cells = cells of the frame behind the L1 cells (:298-300).
 * out4: rows of (plp_id, first cell, FEC blocks, cells). *plp_id_io is the member plp_id, which survives from frame to frame.
 * TIME_IL_TYPE 1 is refused (-2): the reference indexes fec_blocks_per_time_interleving[i][j] for j < time_il_length with
 * arrays of n_ti = 1 entries there (:122-129, :277-284). With one PLP the reference reads cells_per_fec_block[1] past the
 * array (:273); nothing can follow a single PLP, so PLP 0's value is used. PARITY UNPINNED (time_deinterleaver.cpp is a
 * QObject); cross-checked against the transmitter-side construction in tests/. */
typedef struct { int mod, fec_type, time_il_length, time_il_type; } ora_plp_cfg;
typedef struct { int id, start, num_blocks; } ora_plp_dyn;

int ora_ti_frame_walk(int num_plp, const ora_plp_cfg *plp, const ora_plp_dyn *dyn, int num_cells, int *plp_id_io, int *out4,
                      int max_out)
{
    enum { MAXP = 256, MAXTI = 256 };
    static int cells_per_fec_block[MAXP], num_rows[MAXP], n_ti[MAXP], p_i[MAXP], slice_end[MAXP];
    static int num_cols[MAXP][MAXTI];
    if (num_plp < 1 || num_plp > MAXP) return -1;
    for (int i = 0; i < num_plp; ++i) {                                            /* start(), :57-129 */
        int fec_len_bits = plp[i].fec_type == 0 ? 16200 : 64800;
        int bits_per_cell = 2 * (plp[i].mod + 1);
        cells_per_fec_block[i] = fec_len_bits / bits_per_cell;
        num_rows[i] = cells_per_fec_block[i] / 5;
        if (plp[i].time_il_type == 0) { n_ti[i] = plp[i].time_il_length; p_i[i] = 1; }
        else return -2;
        if (n_ti[i] < 1 || n_ti[i] > MAXTI) return -1;
    }
    for (int i = 0; i < num_plp; ++i) {                                            /* l1_dyn_execute(), :271-285 */
        slice_end[i] = dyn[i].start + dyn[i].num_blocks * cells_per_fec_block[num_plp > 1 ? 1 : 0] / p_i[i] - 1;
        int fec_blocks_per_ti_block = (int)floorf((float)dyn[i].num_blocks / (float)n_ti[i] * (float)p_i[i]);
        for (int j = 0; j < plp[i].time_il_length; ++j) {
            int f = fec_blocks_per_ti_block;
            if (j >= (n_ti[i] - dyn[i].num_blocks % n_ti[i])) f += 1;
            num_cols[i][j] = f * 5;
        }
    }
    /* execute(), first call of the frame, :296-312 */
    int plp_id = *plp_id_io;
    int idx_cell = 0;
    for (int i = 0; i < num_plp; ++i) if (dyn[i].start == 0) plp_id = i;
    if (plp_id < 0 || plp_id >= num_plp) return -1;
    int num_rows_plp = num_rows[plp_id];
    int cells_per_fec_block_plp = cells_per_fec_block[plp_id];
    int idx_time_il = 0;
    int ti_block_size = num_cols[plp_id][idx_time_il] * num_rows_plp;
    int idx_step_ti = 0, idx_row_ti = 0;
    int n_out = 0, block_first = 0;
    for (int c = 0; c < num_cells; ++c) {                                          /* :316-374 without the cell moves */
        idx_step_ti += num_rows_plp;
        if (idx_step_ti == ti_block_size) {
            idx_step_ti = 0;
            if (++idx_row_ti == num_rows_plp) {
                idx_row_ti = 0;
                if (n_out >= max_out) return -3;
                out4[4 * n_out + 0] = plp_id; out4[4 * n_out + 1] = block_first;
                out4[4 * n_out + 2] = ti_block_size / cells_per_fec_block_plp; out4[4 * n_out + 3] = ti_block_size;
                ++n_out;
                block_first = idx_cell + 1;
                if (++idx_time_il == plp[plp_id].time_il_length) {
                    idx_time_il = 0;
                    if (idx_cell == slice_end[plp_id]) {
                        for (int i = 0; i < num_plp; ++i) {
                            if (idx_cell == dyn[i].start - 1) {
                                plp_id = dyn[i].id;
                                if (plp_id < 0 || plp_id >= num_plp) return -1;
                                num_rows_plp = num_rows[plp_id];
                                ti_block_size = num_cols[plp_id][idx_time_il] * num_rows_plp;
                                cells_per_fec_block_plp = cells_per_fec_block[plp_id];
                            }
                        }
                    }
                } else {
                    ti_block_size = num_cols[plp_id][idx_time_il] * num_rows_plp;
                }
            }
        }
        ++idx_cell;
    }
    *plp_id_io = plp_id;
    return n_out;
}
