/*
 * t2gpu.h -- C ABI of the MI355X DVB-T2 demodulation/FEC back-end (libt2gpu.so).
 *
 * The reference (Oleg-Malyutin/sdr_receiver_dvb_t2) has no FFI: its stage boundary is the set of Qt slot
 * signatures between the pipeline objects in src/DVB_T2. Every entry point below names the slot it replaces and
 * keeps that slot's buffer convention (frame-major arrays, one bit per byte after the LDPC, A/B ownership stays with
 * the caller). Plain pointers and sizes only; no C++/torch types. INTEGRATION.md shows the reference-side stubs.
 *
 * Conventions
 *   - return value: 0 = ok, < 0 = error (t2gpu_last_error() gives the text). The reference's slots return void and
 *     signal failure by silently dropping data; where that applies the drop decision is reported per batch.
 *   - `*_dev` functions take DEVICE pointers and a hipStream_t (as void*); they enqueue and return without
 *     synchronising. The host-buffer forms copy in, run, copy out and synchronise: they are the drop-in shape of
 *     the reference slots (PCIe time included -- see DESIGN.md).
 *   - fec_type / code_rate are the reference enums dvbt2_fectype_t / dvbt2_code_rate_t
 *     (src/DVB_T2/dvbt2_definition.h:60-67,85-88): fec_type 0 = FECFRAME_SHORT (16200), 1 = FEC_FRAME_NORMAL (64800);
 *     code_rate 0..5 = 1/2, 3/5, 2/3, 3/4, 4/5, 5/6.
 */
#ifndef T2GPU_H
#define T2GPU_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define T2GPU_SIMD_BATCH 32 /* SIZEOF_SIMD of the reference's AVX2 build (src/DVB_T2/ldpc_decoder.h:28-32) */
#define T2GPU_LDPC_TRIALS 25 /* TRIALS (src/DVB_T2/ldpc_decoder.h:63) */

int t2gpu_version(void);
const char *t2gpu_last_error(void);
/* number of visible HIP devices (0 when there is none; never fails) */
int t2gpu_device_count(void);

/* ---------------------------------------------------------------- host-buffer hand-over between stages -----------------------------
 * The reference's slots carry HOST pointers from stage to stage (ti_block -> llr_demapper::execute -> soft_multiplexer_de_twist ->
 * ldpc_decoder::execute -> bit_bch -> ...: dvbt2_demodulator.cpp:84-95 and the main window's connect() chain). The host-buffer entry
 * points below keep that contract -- results are in the caller's buffer when a call returns -- and in addition remember where the
 * same bytes still are on the device (a "twin", found again by the buffer's address). A stage handed the previous stage's buffer
 * unmodified, which is what the signal / slot chain does, skips its copy-in -- ONCE THE CALLER HAS SAID that its buffers travel
 * unmodified: t2gpu_handoff_enable(1), process-wide, returns the previous setting. Off by default: every host-buffer entry point is then
 * a function of the bytes it is handed (each stage copies its input in), whatever was done to a buffer between two stages. The stage
 * classes of t2gpu_stages.hpp switch it on (they hand each other's buffers on untouched, as the reference's objects do). With the switch
 * on, an entry made on the way is still checked against the buffer it was made for -- its first and last 64 bytes and its length are
 * hashed when the entry is made and compared at the look-up, so a buffer that was released and whose address now holds other data
 * falls back to the copy-in -- but an edit in the middle of a handed-on buffer is the caller's to announce (switch off, or do not
 * reuse the buffer). Entries go when the handle that made them overwrites the device side or is destroyed.
 * For buffers the CALLER owns and fills from other stage outputs (llr_demapper's SIMD batch buffers, llr_demapper.cpp:742-764):
 * t2gpu_twin_attach gives the buffer a twin kept by the library (and page-locks the buffer), t2gpu_twin_copy is memcpy(dst, src, n)
 * plus the same copy between the twins, t2gpu_twin_detach releases. All return 0 or -1. */
int t2gpu_handoff_enable(int on);                 /* 0 / 1; returns the previous setting */
int t2gpu_host_pin(void *host, size_t bytes);      /* page-lock a caller's buffer: 0 done, 1 not possible (harmless), -1 bad arguments */
int t2gpu_host_unpin(void *host);
int t2gpu_twin_attach(void *host, size_t bytes, int device);
int t2gpu_twin_detach(void *host);
int t2gpu_twin_copy(void *dst, const void *src, size_t bytes, int device);

/* ---------------------------------------------------------------- LDPC stage ------------------------------------
 * Replaces  void ldpc_decoder::execute(int* idx_plp_simd, l1_postsignalling l1_post, int len_in, int8_t* in)
 *           (src/DVB_T2/ldpc_decoder.h:90, src/DVB_T2/ldpc_decoder.cpp:157-301) and the decoder it drives
 *           (src/DVB_T2/LDPC/layered_decoder.hh:168-180, LDPC/algorithms.hh:221-292).
 * Input : int8 LLRs, frame-major [n_frames][fec_size], transmitted bit order (what llr_demapper hands over).
 * Output: hard decisions of the k_ldpc information bits, frame-major [n_frames][k_ldpc], one bit per byte
 *         (what bch_decoder::execute receives, src/DVB_T2/bch_decoder.h:41), and per batch of `group` frames the
 *         value LDPCDecoder::operator() returns: trials left (>= 0) or -1 = the batch did not converge, in which
 *         case the reference drops it (ldpc_decoder.cpp:264-268). Bits of a dropped batch are still written.
 */
typedef struct t2gpu_ldpc t2gpu_ldpc;

t2gpu_ldpc *t2gpu_ldpc_create(int fec_type, int code_rate, int max_frames, int device);
void t2gpu_ldpc_destroy(t2gpu_ldpc *h);
/* group: frames that stop iterating together. 32 reproduces the reference batch (default); 1 lets every frame
 * stop on its own parity check. max_trials: default 25. */
int t2gpu_ldpc_configure(t2gpu_ldpc *h, int group, int max_trials);
/* fec_size, k_ldpc, q_ldpc, k_bch of the code this handle decodes (any pointer may be NULL) */
int t2gpu_ldpc_info(const t2gpu_ldpc *h, int *fec_size, int *k_ldpc, int *q_ldpc, int *k_bch);
/* graph statistics (host only, needs no GPU): total links, layers, dependency levels per sweep, max links per node */
int t2gpu_ldpc_graph_stats(int fec_type, int code_rate, int *links_total, int *layers, int *levels_total,
                           int *max_cnt);

int t2gpu_ldpc_execute_dev(t2gpu_ldpc *h, const int8_t *d_llr, int n_frames, uint8_t *d_bits,
                           int8_t *d_llr_out /* optional [n_frames][fec_size] final LLRs, or NULL */,
                           int *d_trials_left /* [ceil(n_frames/group)] */, void *stream);
/* For callers that run other streams beside the decoder: enqueue on `stream` a wait (no host blocking) until every workgroup of
 * every decode enqueued so far on this handle has started. The decoder's workgroups are persistent and the frames of a SIMD
 * batch meet at every sweep, so kernels of another stream that grab the CUs first stall whole batches; with this wait they
 * start once the decoder is placed and use what it leaves. */
int t2gpu_ldpc_wait_resident(t2gpu_ldpc *h, void *stream);
/* The decodes of t2gpu_ldpc_submit run on a stream whose CU mask leaves the device's first n_cus CUs to everybody else (default 32 of
 * 256; 0: no mask): short launches of other streams beside resident decodes otherwise take ~45 us longer each.
 * Before the handle's first submit. */
int t2gpu_ldpc_set_submit_cu_reserve(t2gpu_ldpc *h, int n_cus);
/* What a decode occupies: out6 = {workgroups resident per CU, wavefronts per workgroup, dynamic LDS bytes per workgroup, FEC frames per
 * workgroup, CUs of the device, SIMD batches resident at once} for the kernel variant the handle's configuration selects. */
int t2gpu_ldpc_occupancy(const t2gpu_ldpc *h, int *out6);
/* Workgroups a decode of n_frames launches (resident for its whole duration). t2gpu_ldpc_set_plain_launch(h, 1): the handle's decodes
 * become ordinary launches instead of cooperative ones, so that decodes of several handles on several streams run side by side
 * (cooperative launches of different streams run one after the other) -- for callers that keep what is in flight together within the
 * device's CUs (t2gpu_rx with t2gpu_rx_set_overlap does). */
int t2gpu_ldpc_launch_workgroups(const t2gpu_ldpc *h, int n_frames);
int t2gpu_ldpc_set_plain_launch(t2gpu_ldpc *h, int plain);
/* n >= 1: a decode of this handle keeps at most n SIMD batches resident and hands the others out by ticket as slots come free (the CUs it
 * leaves alone are where other streams' kernels run beside it: a resident decode workgroup fills its CU's LDS); 0 (default): all the device holds. */
int t2gpu_ldpc_set_max_slots(t2gpu_ldpc *h, int n);
/* reference-shaped call: len_in = fec_size * n_frames (the reference always passes 32 frames) */
int t2gpu_ldpc_execute(t2gpu_ldpc *h, const int8_t *in, int len_in, uint8_t *out,
                       int *trials_left /* [ceil(n_frames/group)] */);
/* The same slot split in two, for callers that keep the reference's call shape (one SIMD batch per ldpc_decoder::execute,
 * ldpc_decoder.h:90) and still want the device busy: t2gpu_ldpc_submit copies `in` to the handle's pinned staging, enqueues copy-in,
 * decode and copy-out on the handle's own stream and returns (the caller's buffer is free again); t2gpu_ldpc_collect waits for the
 * result (wait != 0) or polls it (wait == 0: returns 1 when the decode is still running) and hands out pointers into the staging --
 * the information bits, one per byte, and the per-batch verdicts -- valid until the next submit on this handle. One submit may be
 * pending per handle; several handles in flight run their batches side by side (t2::ldpc_decoder keeps a ring of them and emits
 * bit_bch in submission order, include/t2gpu_stages.hpp). */
int t2gpu_ldpc_submit(t2gpu_ldpc *h, const int8_t *in, int len_in);
/* t2gpu_ldpc_submit in two steps, for a caller that is handed SIMD batches one by one but wants several of them decoded by ONE launch:
 * t2gpu_ldpc_submit_add puts one more batch (len_in = fec_size * frames, whole SIMD batches except for the last addition) behind what has
 * been added to the handle's input (the caller's buffer is free again, nothing is launched), t2gpu_ldpc_submit_go launches the decode of
 * everything added; t2gpu_ldpc_collect then hands out the bits of all of them and one verdict per SIMD batch. Up to the handle's
 * max_frames. Why: with TWO OR MORE decodes resident on the device at once (launches of different streams, milliseconds each), every
 * launch of every other stream takes tens of microseconds longer to get its workgroups started; with ONE it does not (DESIGN.md section 6,
 * profiles/HISTORY.md round 5) -- t2::ldpc_decoder therefore decodes the batches a TI block gives rise to in one launch. Batches with a
 * device twin and batches without cannot share a decode (-1: launch what has been added first). */
int t2gpu_ldpc_submit_add(t2gpu_ldpc *h, const int8_t *in, int len_in);
int t2gpu_ldpc_submit_go(t2gpu_ldpc *h);
int t2gpu_ldpc_collect(t2gpu_ldpc *h, int wait, const uint8_t **out, const int **trials_left, int *n_frames);
/* diagnostics: the first call arms per-phase cycle counters inside the kernel (off by default); later calls return the
 * sums over all workgroups of the last launch: [0] parity check, [1] batch rendezvous, [2] PLAIN, [3] PAIR, [4] GENERIC
 * layers (shader clock cycles of wave 0). out8 may be NULL. */
int t2gpu_ldpc_profile(t2gpu_ldpc *h, long long *out8);
/* diagnostics: cycles workgroup 0 spent in each layer (first 64 layers) since t2gpu_ldpc_profile armed the counters */
int t2gpu_ldpc_profile_layers(t2gpu_ldpc *h, long long *out64);
/* after a synchronised execute: 0 = clean, 1 = a batch rendezvous timed out (results invalid) */
int t2gpu_ldpc_status(t2gpu_ldpc *h);

/* ---------------------------------------------------------------- LLR demapper + bit de-interleaver ---------------
 * Replaces  void llr_demapper::execute(int ti_block_size, complex* time_deint_cell, int plp_id, l1_postsignalling)
 *           (src/DVB_T2/llr_demapper.h:44-45, llr_demapper.cpp:132-158) and qpsk/qam16/qam64/qam256 (:160-768).
 * mod / fec_type / code_rate / rotation are l1_post.plp[plp_id].{plp_mod, plp_fec_type, plp_cod, plp_rotation}
 * (mod: 0 QPSK, 1 16-QAM, 2 64-QAM, 3 256-QAM). Input: one TI block of cells, complex float interleaved (re, im).
 * Output: int8 LLRs [n_frames][fec_size] in LDPC input order (n_frames = n_cells / cells per FEC block, returned), and
 * sums3 = {sum_s, sum_e, precision}: the hard-decision signal / error energies the reference derives its LLR scale
 * 8*norm*sum_s/sum_e from (SNR readout: 20*log10(sum_s/sum_e), llr_demapper.cpp:659). The reference de-rotates its input
 * buffer in place; this implementation de-rotates on the fly and leaves the input untouched.
 * The two sums are formed as the reference forms them -- SEQUENTIAL float additions in cell order (one vaddss per cell and sum in the
 * reference binary) -- reproduced bit for bit by a parallel kernel (quantised integer prefix sums inside a binade, real float adds at
 * binade changes and rounding ties; csrc/fec_kernels.hip: demap_stats_exact_kernel): the scale, hence every LLR, equals what the
 * reference's arithmetic gives on the same cells. (T2GPU_DEMAP_TREE_STATS=1 selects the tree / double sums of rounds 1-2: ~1e-4 off in
 * the scale, one LLR step on ~2 % of the positions.)
 * precision_override > 0 fixes the LLR scale instead of measuring it (used by parity tests; 0 = reference behaviour).
 * Grouping the frames into SIMD batches of 32 for the LDPC stage (static `blocks`, llr_demapper.cpp:549-552) is the
 * caller's business: frames leave in arrival order. */
typedef struct t2gpu_demap t2gpu_demap;
t2gpu_demap *t2gpu_demap_create(int mod, int fec_type, int code_rate, int rotation, int max_cells, int device);
void t2gpu_demap_destroy(t2gpu_demap *h);
/* saturate = 0 (default): the reference's int8 cast, which does not clamp for 16/64/256-QAM and wraps around once
 * |LLR| exceeds 127 (llr_demapper.cpp:722-737; for 256-QAM in AWGN this happens on the outer constellation points at any
 * SNR, because the hard-decision SNR estimate never drops below about 22 dB). saturate = 1 clamps instead -- an extension
 * beyond the reference, off unless asked for. */
int t2gpu_demap_configure(t2gpu_demap *h, int saturate);
int t2gpu_demap_execute_dev(t2gpu_demap *h, const float *d_cells, int n_cells, float precision_override, int8_t *d_llr,
                            float *d_sums3 /* 3 floats on the device, or NULL */, void *stream);
int t2gpu_demap_execute(t2gpu_demap *h, const float *cells, int n_cells, int8_t *llr, float *sums3 /* or NULL */);
/* t2gpu_demap_execute_dev as two enqueues -- the statistics pass (sum_s, sum_e, LLR scale -> d_sums3) and the LLR pass that reads
 * them -- so that a multi-stream scheduler can place them separately; calling one after the other equals execute_dev. */
int t2gpu_demap_stats_dev(t2gpu_demap *h, const float *d_cells, int n_cells, float precision_override, float *d_sums3, void *stream);
int t2gpu_demap_llr_dev(t2gpu_demap *h, const float *d_cells, int n_cells, const float *d_sums3, int8_t *d_llr, void *stream);
/* the statistics pass for n_blocks TI blocks of cells_per_block cells in one launch: block t at d_cells + 2 * t * cells_stride
 * floats, its triple at d_sums + t * sums_stride (same sums as n_blocks calls of t2gpu_demap_stats_dev, bit for bit) */
int t2gpu_demap_stats_batch_dev(t2gpu_demap *h, const float *d_cells, long cells_stride, int n_blocks, int cells_per_block,
                                float precision_override, float *d_sums, int sums_stride, void *stream);
/* the LLR pass for n_blocks TI blocks in one launch: block t = cells_per_block cells at d_cells + 2 * t * cells_per_block floats,
 * its statistics at d_sums + t * sums_stride; LLR frames back to back in d_llr. Returns the number of FEC frames. */
int t2gpu_demap_llr_batch_dev(t2gpu_demap *h, const float *d_cells, int n_blocks, int cells_per_block, const float *d_sums,
                              int sums_stride, int8_t *d_llr, void *stream);

/* ---------------------------------------------------------------- time / cell de-interleaver ----------------------
 * Replaces  void time_deinterleaver::execute(int len, complex* cells) / l1_dyn_execute(l1_post, len, cells)
 *           (src/DVB_T2/time_deinterleaver.h:33,44-45; time_deinterleaver.cpp:268-376) for one TI block of one PLP
 *           (TIME_IL_TYPE 0; which TI blocks a frame holds, for any number of PLPs: t2gpu_ti_frame_plan below):
 * time de-interleaving (N_split = 5), cell de-interleaving and removal of the cyclic Q delay in one scatter.
 * t2gpu_ti_begin(num_blocks) is the geometry step of l1_dyn_execute (PLP_NUM_BLOCKS of this frame); cells are then
 * pushed in arrival order, any number per call (the reference pushes one OFDM symbol at a time; the P2 symbol's
 * L1 cells are skipped by the caller, time_deinterleaver.cpp:296-300). out is the caller's TI buffer of
 * num_blocks * cells_per_fec complex cells (the reference's A/B buffer): a push returns 1 when the TI block is complete
 * -- the moment the reference emits ti_block -- else 0. The host form reads `out` with the first push of a block (cells the scatter
 * does not reach keep their history) and writes it when the block is complete; in between the block stays on the device. */
typedef struct t2gpu_ti t2gpu_ti;
t2gpu_ti *t2gpu_ti_create(int mod, int fec_type, int num_blocks_max, int device);
void t2gpu_ti_destroy(t2gpu_ti *h);
int t2gpu_ti_cells_per_fec(const t2gpu_ti *h);
int t2gpu_ti_begin(t2gpu_ti *h, int num_blocks);
int t2gpu_ti_push_dev(t2gpu_ti *h, const float *d_cells, int n_cells, float *d_out, void *stream);
int t2gpu_ti_push(t2gpu_ti *h, const float *cells, int n_cells, float *out);
/* t2gpu_ti_push in two steps, for a caller whose thread must not stand still while a complete block comes down (13 MB for CFG-A; the
 * reference's demodulator thread hands the block to the de-interleaver's own thread the same way, time_deinterleaver.cpp:30-36,313-353):
 * _async returns 1 with the block's copy into `out` still on its way (page-lock `out`: t2gpu_host_pin); t2gpu_ti_wait, from any thread,
 * returns when `out` holds the block -- call it before `out` is read or handed on. A push that finds the wait outstanding does it first. */
int t2gpu_ti_push_async(t2gpu_ti *h, const float *cells, int n_cells, float *out);
int t2gpu_ti_wait(t2gpu_ti *h);
/* n_blocks complete TI blocks of the geometry set by t2gpu_ti_begin in one launch (e.g. the same TI block of every T2 frame of a
 * buffer): block f reads ti_block_size cells at d_cells + 2 * f * in_stride_cells floats and writes d_out + 2 * f *
 * out_stride_cells. Same result as t2gpu_ti_begin + one whole-block t2gpu_ti_push_dev per block. Returns n_blocks. */
int t2gpu_ti_execute_blocks_dev(t2gpu_ti *h, const float *d_cells, long in_stride_cells, float *d_out, long out_stride_cells,
                                int n_blocks, void *stream);
/* t2gpu_ti_execute_blocks_dev followed by the demapper's first pass (t2gpu_demap_stats_batch_dev) on the de-interleaved blocks, as one
 * call: d_sums + b * sums_stride receives (sum_s, sum_e, precision) of TI block b. dm: the demapper of the same modulation / FEC type.
 * Returns n_blocks. */
int t2gpu_ti_execute_blocks_stats_dev(t2gpu_ti *h, t2gpu_demap *dm, const float *d_cells, long in_stride_cells, float *d_out,
                                      long out_stride_cells, int n_blocks, float precision_override, float *d_sums, int sums_stride,
                                      void *stream);

/* The same in two steps with a stage boundary between them: the de-interleaver (time_deinterleaver::execute) also forms the demapper's
 * per-cell statistics terms (the |s|^2 and |e|^2 of llr_demapper.cpp:564-676) of the cells it writes -- one pass over the cells for
 * both -- and the demapper then sums them in the reference's order. t2gpu_ti_execute_blocks_terms_dev returns 1 when the terms were
 * formed, 0 when only the de-interleaving was done (then t2gpu_demap_stats_batch_dev on d_out is the next step), -1 on error. */
int t2gpu_ti_execute_blocks_terms_dev(t2gpu_ti *h, t2gpu_demap *dm, const float *d_cells, long in_stride_cells, float *d_out,
                                      long out_stride_cells, int n_blocks, void *stream);
int t2gpu_demap_stats_terms_dev(t2gpu_demap *dm, int n_blocks, int cells_per_block, float precision_override, float *d_sums,
                                int sums_stride, void *stream);

/* ---------------------------------------------------------------- BB descrambler (the reference's BCH stage) ------
 * Replaces  void bch_decoder::execute(int* idx_plp_simd, l1_postsignalling, int len_in, uint8_t* in)
 *           (src/DVB_T2/bch_decoder.h:41, bch_decoder.cpp:63-164). The reference performs no BCH decoding (:136): it
 * keeps the first k_bch bits of every k_ldpc block and XORs the BB scrambling sequence (:50-61). in: [n_frames][k_ldpc]
 * one bit per byte; out: [n_frames][k_bch]. Returns k_bch. */
int t2gpu_bch_descramble_dev(int fec_type, int code_rate, const uint8_t *d_bits, int n_frames, uint8_t *d_out, void *stream);
int t2gpu_bch_descramble(int fec_type, int code_rate, const uint8_t *bits, int n_frames, uint8_t *out);
/* K-descramble-pack (a15 + the byte packing part of a16): the same XOR with the result packed MSB first, k_bch / 8
 * bytes per frame -- the byte assembly bb_de_header::execute does bit by bit (bb_de_header.cpp:84-448) done where the bits are.
 * in: [n_frames][k_ldpc] one bit per byte (values 0 / 1, 8-byte aligned rows); out: [n_frames][k_bch / 8]. Returns k_bch / 8. */
int t2gpu_bch_descramble_pack_dev(int fec_type, int code_rate, const uint8_t *d_bits, int n_frames, uint8_t *d_bytes, void *stream);

/* Outer code applied for real (NOT what the reference does: bch_decoder.cpp:136 is "// TODO BCH decode", so a
 * drop-in caller goes straight to t2gpu_bch_descramble). The code of EN 302 755 clause 6.1.1 for the (k_bch, n_bch) pairs of
 * bch_decoder.cpp:79-134: t = 12 over GF(2^16) (t = 10 for r = 2/3 and 5/6) on 64 800-bit frames, t = 12 over GF(2^14) on 16 200-bit ones.
 * bits: [n_frames][n_bch = k_ldpc] one bit per byte as the LDPC stage leaves them (8-byte aligned), corrected IN PLACE;
 * status[f] = number of bits corrected (0 = the frame was a codeword), -1 = more than t errors, frame left untouched.
 * Return n_frames. t2gpu_bch_info: field degree m, t, k_bch, n_bch of a code (any pointer may be NULL). */
int t2gpu_bch_info(int fec_type, int code_rate, int *m, int *t, int *k_bch, int *n_bch);
/* host only: the minimal polynomials of alpha, alpha^3, ..., alpha^(2t-1) over GF(2^m) whose product is the generator (bit k =
 * coefficient of x^k; EN 302 755 tables 6a / 6b list the same polynomials). Returns t. */
int t2gpu_table_bch_minpoly(int m, int t, uint32_t *out);
int t2gpu_bch_decode_dev(int fec_type, int code_rate, uint8_t *d_bits, int n_frames, int32_t *d_status, void *stream);
int t2gpu_bch_decode(int fec_type, int code_rate, uint8_t *bits, int n_frames, int32_t *status);

/* ---------------------------------------------------------------- L1 signalling from the P2 cells (host) ------------
 * Replaces  bool p2_symbol::l1_pre_info(dvbt2_parameters&)  (src/DVB_T2/p2_symbol.cpp:301-532) and
 * bool p2_symbol::l1_post_info()  with its field parsers (:534-1089). Input: the equalised, frequency-de-interleaved cells
 * of the P2 symbol (what t2gpu_eq_p2_execute_dev returns): the first 1840 carry L1-pre (200 systematic BPSK bits are read),
 * the next l1_post_size carry L1-post. As the reference: hard decisions on the systematic bits + CRC-32, no L1 FEC decoding.
 * Field names are those of the reference's l1_presignalling / l1_postsignalling structs (dvbt2_definition.h:249-344).
 * Return 1 = CRC-32 ok (structs filled), 0 = CRC-32 error, -1 = bad arguments. */
typedef struct {
    int32_t type, bwt_ext, s1, s2_field1, s2_field2, l1_repetition_flag, guard_interval, papr, l1_post_mod, l1_cod, l1_fec_type,
        l1_post_size, l1_post_info_size, pilot_pattern, tx_id_availability, cell_id, network_id, t2_system_id, num_t2_frames,
        num_data_symbols, regen_flag, l1_post_extension, num_rf, current_rf_index, t2_version, l1_post_scrambled, t2_base_lite,
        reserved;
    uint32_t crc_32;
} t2gpu_l1_pre;
typedef struct {
    int32_t id, plp_type, plp_payload_type, ff_flag, first_rf_idx, first_frame_idx, plp_group_id, plp_cod, plp_mod, plp_rotation,
        plp_fec_type, plp_num_blocks_max, frame_interval, time_il_length, time_il_type, in_band_a_flag, in_band_b_flag, reserved_1,
        plp_mode, static_flag, static_padding_flag;
} t2gpu_l1_plp;
typedef struct { int32_t id, start, num_blocks, reserved_2; } t2gpu_l1_dyn_plp;
typedef struct {
    int32_t sub_slices_per_frame, num_plp, num_aux, aux_config_rfu, rf_idx[8];
    uint32_t frequency[8];
    int32_t fef_type, fef_length, fef_interval, fef_length_msb, reserved_2;
    int32_t frame_idx, sub_slice_interval, type_2_start, l1_change_counter, start_rf_idx, dyn_reserved_1, dyn_reserved_3;
} t2gpu_l1_post;
int t2gpu_l1_pre_parse(const float *p2_cells, t2gpu_l1_pre *out);
int t2gpu_l1_post_parse(const float *l1_post_cells /* = p2_cells + 2*1840 */, const t2gpu_l1_pre *pre, t2gpu_l1_post *post,
                        t2gpu_l1_plp *plp /* [max_plp] */, t2gpu_l1_dyn_plp *dyn /* [max_plp] */, int max_plp);

/* ---------------------------------------------------------------- frame de-multiplexer of the time de-interleaver (host) --
 * The PLP / TI-block bookkeeping of  time_deinterleaver::start, l1_dyn_execute, execute  (src/DVB_T2/time_deinterleaver.cpp
 * :38-145, :268-286, :296-312, :337-371) for any number of PLPs: which TI blocks the cell stream of one T2 frame holds, in the
 * order the reference emits ti_block for them. The cell stream is what follows the L1 cells of P2 (frame_cells = its length:
 * PLP cells of P2, of every data symbol and of the frame-closing symbol). Block k covers cells [offset, offset + size) and
 * belongs to PLP index `plp` (index into plp[] / dyn[], which the reference also uses PLP_ID as, :360); it is de-interleaved by
 * t2gpu_ti_begin(handle of that PLP, num_blocks) + t2gpu_ti_push[_dev]. Reproduced as the reference behaves:
 *  - the frame starts with the PLP whose PLP_START is 0 (else the PLP the previous frame ended in: *plp_state, in/out);
 *  - TIME_IL_TYPE 0 only: TIME_IL_LENGTH TI blocks per T2 frame, the later blocks take the remainder of PLP_NUM_BLOCKS;
 *  - a PLP is followed by the PLP whose PLP_START is the next cell, provided the PLP ends where the reference computes its end
 *    -- with the cell count per FEC block of PLP index 1 (:273-274), i.e. PLPs of one modulation and FEC type;
 *  - otherwise the same PLP's geometry is applied to the following cells again (what comes out is noise for the LDPC to drop);
 *  - a TI block that does not complete inside the frame is discarded at the next frame start.
 * Returns the number of blocks written to out (<= max_out), -1 bad arguments, -2 TIME_IL_TYPE 1 (the reference writes past its
 * per-PLP arrays for it and never collects cells over several frames), -3 out too small. */
typedef struct { int32_t plp, offset, num_blocks, size; } t2gpu_ti_block;
int t2gpu_ti_frame_plan(int num_plp, const t2gpu_l1_plp *plp, const t2gpu_l1_dyn_plp *dyn, int frame_cells, int *plp_state,
                        t2gpu_ti_block *out, int max_out);

/* ---------------------------------------------------------------- BBFRAME de-framing -> transport stream (host) -----
 * Replaces  void bb_de_header::execute(int plp_id, l1_postsignalling, int len_in, uint8_t* in)  (src/DVB_T2/bb_de_header.h:59,
 * bb_de_header.cpp:84-448) and set_out's need_plp (:500-525). in: the k_bch descrambled bits of one BBFRAME, one per byte.
 * Sequential host code, as in the reference (TS packets and the normal-mode CRC-8 straddle BBFRAMEs); the reference writes
 * the bytes to UDP 127.0.0.1:7654 or a file, here they are returned: the return value is the number of TS bytes placed in
 * out (out_cap >= len_in/8 + 376), or -1 BBHEADER CRC-8 error (frame dropped, :108-113), -2 frame skipped (other PLP,
 * :139-142, or SYNCD = 65535, :160-163), -3 bad arguments. *ts_errors counts packets flagged with TEI (normal mode).
 * t2gpu_bbdh_mode: 0 normal mode, 1 high-efficiency mode of the last accepted frame. t2gpu_bbdh_resync_count: how many times
 * the last call raised the reference's "Baseband header resynchronizing." (:218,235,369,384).
 * Bounds (the reference has none: it trusts SYNCD / DFL, and in normal mode consumes 8 bits per packet more than it takes off
 * DFL, :290-321): bits past len_in read as 0, never as memory; a frame whose SYNCD / DFL would write more than out_cap bytes is
 * refused with -3 and the packet state is reset. Byte-identical to the reference's class on tests/golden/t2fec_golden.npz
 * ("bbdh/...": HEM, NM as written, lost frames, other PLP, broken header, SYNCD 0xFFFF). */
typedef struct t2gpu_bbdh t2gpu_bbdh;
t2gpu_bbdh *t2gpu_bbdh_create(int need_plp);
void t2gpu_bbdh_destroy(t2gpu_bbdh *h);
int t2gpu_bbdh_execute(t2gpu_bbdh *h, int plp_id, int len_in, const uint8_t *bits, uint8_t *out, int out_cap, int *ts_errors);
/* The same on a PACKED BBFRAME: len_in bits in (len_in + 7) / 8 bytes, MSB first -- the byte assembly of bb_de_header.cpp:84-448
 * (`temp |= *in++ << n`, n = 7..0) already done on the device by t2gpu_bch_descramble_pack_dev / t2gpu_rx. Same results, same
 * state: calls of the two forms may be mixed on one handle. */
int t2gpu_bbdh_execute_packed(t2gpu_bbdh *h, int plp_id, int len_in, const uint8_t *bytes, uint8_t *out, int out_cap, int *ts_errors);
/* Many packed BBFRAMEs in stream order through the one de-framer: bb_de_header::execute (bb_de_header.cpp:84-448) called row after row
 * as bch_decoder emits them (bch_decoder.cpp:140-160), with the LDPC stage's drop rule in front (ldpc_decoder.cpp:264-268: trials[i /
 * group] < 0 -> the rows of that SIMD batch are never emitted; trials may be NULL). rows + i * row_stride = row i (k_bch / 8 bytes);
 * TS bytes are appended to out (out_cap >= the bytes the rows carry + k_bch / 8 + 376). counts (6 longs, may be NULL): rows de-framed,
 * rows dropped by the LDPC rule, BBHEADER CRC errors, frames skipped, TS packets flagged, resynchronisations. Returns TS bytes or -3.
 * The rank-0 end of the frame-sharded receiver (sdr_receiver_dvb_t2_amd/shard.py) is this one call per gathered share. */
long t2gpu_bbdh_execute_packed_rows(t2gpu_bbdh *h, int plp_id, int k_bch, const uint8_t *rows, long n_rows, long row_stride,
                                    const int32_t *trials, int group, uint8_t *out, long out_cap, long *counts);
int t2gpu_bbdh_mode(const t2gpu_bbdh *h);
int t2gpu_bbdh_resync_count(const t2gpu_bbdh *h);
/* The packet state of a freshly constructed bb_de_header (bb_de_header.cpp:29-54: no split packet pending, idx_packet = 0, crc = 0):
 * a new stream on the same handle. */
int t2gpu_bbdh_reset(t2gpu_bbdh *h);

/* ---------------------------------------------------------------- OFDM side: FFT and data-symbol equaliser ----------
 * Mode arguments are the reference's dvbt2_parameters fields (src/DVB_T2/dvbt2_definition.h:215-248): fft_mode
 * (dvbt2_fft_mode_t: FFTSIZE_16K = 4, FFTSIZE_32K = 5, and their _T2GI twins 11 / 7), carrier_mode (0 normal, 1 extended),
 * pilot_pattern (PP1..PP8 = 0..7), guard_interval_mode (dvbt2_guardinterval_t), papr_mode (dvbt2_papr_t), n_data
 * (NUM_DATA_SYMBOLS of L1-pre). Supported: 16K and 32K, SISO -- the reference's own scope (README.md:17-23).
 *
 * t2gpu_fft_execute replaces  complex* fast_fourier_transform::execute()  (src/DSP/fast_fourier_transform.h:62-70): forward
 * DFT of fft_size complex samples per symbol (guard interval already removed), output fft-shifted (DC at fft_size/2),
 * batched over n_symbols symbols laid out back to back (complex float, re/im interleaved).
 *
 * t2gpu_eq_data_execute replaces  complex* data_symbol::execute(int idx_symbol, complex* ofdm_cell, float& sample_rate_offset,
 * float& phase_offset)  (src/DVB_T2/data_symbol.h:31-32, data_symbol.cpp:108-335): pilot-based channel estimate, per-carrier
 * equalisation and frequency de-interleaving of one data symbol: fft_size shifted FFT bins in, c_data cells out (returned
 * count), plus the two synchronisation feedback values. idx_symbol counts the symbols of the T2 frame after P1 (P2 = 0).
 * The _dev form takes a batch: symbols [n][fft_size], d_symbol_index[n], cells [n][c_data], sync [n][2] = (phase_offset,
 * sample_rate_offset) per symbol (may be NULL). */
typedef struct t2gpu_ofdm t2gpu_ofdm;
t2gpu_ofdm *t2gpu_ofdm_create(int fft_mode, int carrier_mode, int pilot_pattern, int guard_interval_mode, int papr_mode,
                              int n_data, int max_symbols, int device);
void t2gpu_ofdm_destroy(t2gpu_ofdm *h);
int t2gpu_fft_execute_dev(t2gpu_ofdm *h, const float *d_in, float *d_out, int n_symbols, void *stream);
int t2gpu_fft_execute(t2gpu_ofdm *h, const float *in, float *out, int n_symbols);
/* The same transform reading its symbols straight out of the decimated sample stream: symbol i of the batch starts at cell
 * first + (i / per_frame) * frame_stride + (i % per_frame) * sym_stride. With first = position of the first useful sample,
 * sym_stride = guard_interval_size + fft_size this is symbol_acquisition's guard removal + fft->execute()
 * (src/DVB_T2/dvbt2_demodulator.cpp:332-334) for whole frames, without the intermediate copy. */
int t2gpu_fft_execute_strided_dev(t2gpu_ofdm *h, const float *d_stream, long first, long frame_stride, int per_frame, int sym_stride,
                                  float *d_out, int n_symbols, void *stream);
int t2gpu_eq_data_execute_dev(t2gpu_ofdm *h, const float *d_symbols, const int32_t *d_symbol_index, int n_symbols,
                              float *d_cells, float *d_sync, void *stream);
/* t2gpu_eq_data_execute_dev for ONE symbol whose cells the equaliser's own workgroups also store to page-locked host memory (h_cells: c_data
 * cells), *h_flag = seq behind them (system scope): the `data` signal's buffer without a copy or a launch of its own. d_count: a zeroed
 * device word of the caller's (left at zero). */
int t2gpu_eq_data_publish_dev(t2gpu_ofdm *h, const float *d_symbol, const int32_t *d_symbol_index, float *d_cells, float *h_cells,
                              unsigned *h_flag, unsigned seq, unsigned *d_count, void *stream);
/* The same for the data symbols of whole frames, read in place from the frames' spectra (n_frames x syms_per_frame x fft_size
 * cells, what t2gpu_fft_execute_strided_dev wrote) and written in place into the frames' cell streams: data symbol first_symbol + l
 * of frame f is read at d_spectrum + 2 * (f * syms_per_frame + first_symbol + l) * fft_size floats and its c_data cells go to
 * d_cells + 2 * (f * cells_frame_stride + cells_offset + l * c_data) floats. d_sync (optional): n_frames * n_data_symbols pairs. */
int t2gpu_eq_data_frames_dev(t2gpu_ofdm *h, const float *d_spectrum, int n_frames, int syms_per_frame, int first_symbol,
                             int n_data_symbols, float *d_cells, long cells_frame_stride, long cells_offset, float *d_sync,
                             void *stream);
int t2gpu_eq_data_execute(t2gpu_ofdm *h, int idx_symbol, const float *ofdm_cell, float *cells, float *sample_rate_offset,
                          float *phase_offset);
/* Equaliser + frequency de-interleaver part of  complex* p2_symbol::execute(...)  (src/DVB_T2/p2_symbol.h:42-44,
 * p2_symbol.cpp:89-262) for the P2 symbols of n_symbols frames: [n][fft_size] in, [n][c_p2] cells out (returned count; the
 * first 1840 + l1_post_size cells of each are L1 signalling, time_deinterleaver.cpp:296-300). L1 parsing is not done here. */
int t2gpu_eq_p2_execute_dev(t2gpu_ofdm *h, const float *d_symbols, int n_symbols, float *d_cells, float *d_sync, void *stream);
/* Replaces  complex* fc_symbol::execute(complex* ofdm_cell, float& sample_rate_offset, float& phase_offset)
 * (src/DVB_T2/fc_symbol.h:31, fc_symbol.cpp:82-271) for the frame-closing symbols of n_symbols frames: [n][fft_size] in,
 * [n][n_fc] cells out (returned count). Fails when the mode has no frame-closing symbol (l_fc = 0). */
int t2gpu_eq_fc_execute_dev(t2gpu_ofdm *h, const float *d_symbols, int n_symbols, float *d_cells, float *d_sync, void *stream);
/* phase_offset / sample_rate_offset of ONE symbol -- the two by-reference outputs of data_symbol::execute (src/DVB_T2/data_symbol.cpp:108-109,
 * 319-324), p2_symbol::execute (p2_symbol.cpp:89-262) and fc_symbol::execute (fc_symbol.cpp:82-271) -- from the symbol's pilots alone,
 * without the equaliser's cell work (the reference's loop hands them to the tracking filters before the next chunk,
 * dvbt2_demodulator.cpp:429-439), together with the guard-interval correlation of the symbol as buffered (dvbt2_demodulator.cpp:321-327).
 * The same floats, bit for bit, as t2gpu_eq_*_execute_dev's d_sync and t2gpu_cp_correlate_dev's d_out4.
 * kind: 0 data symbol idx_symbol, 1 P2, 2 frame closing. d_spectrum: fft_size cells (what the FFT wrote). d_buffered (may be null):
 * guard + fft_size cells, guard first. d_cp4 (4 floats) / d_sync (2 floats): device, may be null. h_small (8 floats) / h_flag:
 * page-locked host memory of the caller, may be null -- the kernel itself stores {sum.re, sum.im, frequency_est, 0} at h_small[0..3]
 * (only with d_buffered) and {phase_offset, sample_rate_offset} at h_small[4..5], then seq at *h_flag (system scope, behind them).
 * d_loop (may be null): the device's tracking-loop state (t2gpu_front_loop_dev): the launch's last lane then also advances the two loop
 * filters with these floats as t2gpu_sync_frequency / t2gpu_sync_symbol do, and h_small[6..7] receive the new phase_est_filtered and
 * frequency_est_filtered + tuner. */
int t2gpu_sym_sync_dev(t2gpu_ofdm *h, int kind, int idx_symbol, const float *d_spectrum, const float *d_buffered, int guard,
                       float *d_cp4, float *d_sync, float *h_small, unsigned *h_flag, unsigned seq, void *d_loop, void *stream);
/* symbol_acquisition's guard removal + fft->execute() (dvbt2_demodulator.cpp:332-334) for ONE buffered symbol (d_buffered: guard +
 * fft_size cells, guard first; spectrum to d_spectrum) with t2gpu_sym_sync_dev's outputs formed inside the FFT's last launch: what
 * t2gpu_fft_execute_strided_dev + t2gpu_sym_sync_dev give, bit for bit, in one launch (per handle: t2gpu_ofdm_set_one_launch(h, 0) = the two launches of before, same values) instead of three. h: the handle whose FFT
 * runs; tables: the handle whose pilot tables apply (the same FFT size; kind / idx_symbol as in t2gpu_sym_sync_dev). with_cp = 0: no
 * guard correlation (h_small[0..3] untouched). Pilot tables that do not fit the FFT's exchange buffer (P2, dense pilot patterns) take
 * the separate launches inside. */
int t2gpu_fft_sym_sync_dev(t2gpu_ofdm *h, t2gpu_ofdm *tables, int kind, int idx_symbol, const float *d_buffered, int guard, int with_cp,
                           float *d_spectrum, float *d_cp4, float *d_sync, float *h_small, unsigned *h_flag, unsigned seq, void *d_loop, void *stream);
/* on = 1 (default): t2gpu_fft_sym_sync_dev is ONE launch (stage A in its first four workgroups, stages B + C and the synchronisation
 * floats in the other four, which wait for them); 0: the two launches. Process-wide; same values bit for bit. */
int t2gpu_ofdm_set_one_launch(t2gpu_ofdm *h, int on);
/* The same two for whole frames, in place like t2gpu_eq_data_frames_dev: the P2 (frame-closing) symbol of frame f is read at
 * d_spectrum + 2 * f * syms_per_frame * fft_size floats (+ the symbol's position in the frame); P2: the cells behind the first
 * skip_cells (the L1 cells, time_deinterleaver.cpp:296-300) go to d_cells + 2 * f * cells_frame_stride floats; frame closing: the
 * n_fc cells go to d_cells + 2 * (f * cells_frame_stride + cells_offset). Return the cells stored per frame. */
int t2gpu_eq_p2_frames_dev(t2gpu_ofdm *h, const float *d_spectrum, int n_frames, int syms_per_frame, float *d_cells,
                           long cells_frame_stride, int skip_cells, float *d_sync, void *stream);
int t2gpu_eq_fc_frames_dev(t2gpu_ofdm *h, const float *d_spectrum, int n_frames, int syms_per_frame, float *d_cells,
                           long cells_frame_stride, long cells_offset, float *d_sync, void *stream);
/* t2gpu_eq_p2_frames_dev that also keeps the skipped cells: the L1-pre / L1-post cells of frame f at d_l1_cells + 2 * f * skip_cells
 * floats (what t2gpu_l1_pre_parse / t2gpu_l1_post_parse read; p2_symbol.cpp:264-299 hands the same cells to l1_pre_info). */
int t2gpu_eq_p2_frames_l1_dev(t2gpu_ofdm *h, const float *d_spectrum, int n_frames, int syms_per_frame, float *d_cells,
                              long cells_frame_stride, int skip_cells, float *d_l1_cells, float *d_sync, void *stream);
/* host only: {fft_size, k_total, k_ext, k_offset, l_nulls, c_p2, c_data, n_fc, c_fc, l_fc, len_frame, guard_interval_size}
 * (dvbt2_{p2,bwt_ext,data}_parameters_init, src/DVB_T2/dvbt2_definition.cpp:20-648) */
int t2gpu_ofdm_mode_info(int fft_mode, int carrier_mode, int pilot_pattern, int guard_interval_mode, int papr_mode, int n_data,
                         int *out12);
/* host only: carrier-type map (dvbt2_carrier_type_t values) and signed pilot reference of symbol idx_symbol, k_total entries
 * (pilot_generator, src/DVB_T2/pilot_generator.cpp:69-132,2093-2166); receiver frequency de-interleaver tables, kind 0 = P2,
 * 1 = data, 2 = frame closing (address_freq_deinterleaver.cpp:136-209). Return the number of entries. */
int t2gpu_table_symbol_carriers(int fft_mode, int carrier_mode, int pilot_pattern, int guard_interval_mode, int papr_mode,
                                int n_data, int idx_symbol, uint8_t *map, float *refer);
int t2gpu_table_freq_deint(int fft_mode, int carrier_mode, int pilot_pattern, int guard_interval_mode, int papr_mode, int n_data,
                           int kind, int32_t *h_even, int32_t *h_odd);

/* ---------------------------------------------------------------- sample-rate front end ----------------------------
 * Replaces the per-sample loop of
 *     void dvbt2_demodulator::execute(int len_in, int16_t* i_in, int16_t* q_in, signal_estimate* signal)
 *     (src/DVB_T2/dvbt2_demodulator.h:64, src/DVB_T2/dvbt2_demodulator.cpp:145-254)
 * up to the hand-over to symbol_acquisition: int16 -> float, dc removal (exponential_averager, DSP/loop_filters.hh:56-73),
 * IQ-imbalance correction from the previous call's sign statistics (:184-185,228-234,256-265), NCO de-rotation (:187-205),
 * cubic Farrow resampling (DSP/interpolator_farrow.hh:41-68) and the 64-tap /2 decimator (DSP/filter_decimator.h:72-131).
 * The handle owns what the reference keeps in dvbt2_demodulator members between calls (averagers, NCO phases, c1/c2, Farrow
 * delay line and position, decimator history and phase).
 *
 * The reference closes its tracking loops once per OFDM symbol: inside one execute() the input is cut into chunks (one
 * per symbol, :151-163) and each chunk runs with constant loop values. Here one call takes all chunks of an execute() with
 * the loop values of each: phase_est_filtered (added to phase_nco at the chunk start, :165), frequency_est_filtered
 * (subtracted from frequency_nco per sample, :187) and arbitrary_resample (:157-158). A closed loop calls it with one chunk
 * at a time; an open-loop replay (or the benchmark: zeros and the nominal resample) passes a whole buffer.
 * id_device: 0 = SDRplay (scale 2^-14), 1 = AirSpy (2^-12, I/Q interleaved: stride 2), 2 = PlutoSDR (2^-11)  (:31-50).
 * out: decimated complex samples (re,im float pairs) of all chunks back to back; chunk_out_len[c] = cells chunk c produced
 * (what symbol_acquisition receives as len_in). Returns the total number of output cells, < 0 on error. */
typedef struct t2gpu_front t2gpu_front;
t2gpu_front *t2gpu_front_create(int id_device, float sample_rate, int max_samples, int device);
void t2gpu_front_destroy(t2gpu_front *h);
int t2gpu_front_reset(t2gpu_front *h); /* dvbt2_demodulator::reset, :111-127 */
/* dvbt2_demodulator::reset (:111-127) as far as it concerns this handle: dc averagers and both NCO accumulators to zero;
 * c1 / c2 / level_detect and the filter delay lines live on (t2gpu_front_reset is the constructor state). */
int t2gpu_front_reset_loops(t2gpu_front *h);
/* frequency_nco = value (set_guard_interval_by_brute_force zeroes it, :486-487) */
int t2gpu_front_set_frequency_nco(t2gpu_front *h, float frequency_nco);
int t2gpu_front_set_iq(t2gpu_front *h, float c1, float c2); /* c1 / c2 as a previous execute() left them (:228-234); test set-up */
/* One execute() of the reference = several calls here when the loops are closed (one per chunk). hold = 1: the calls only add
 * to the sign statistics; t2gpu_front_commit_iq at the end of the execute() derives c1 / c2 / level_detect from all of them
 * (:227-235) for the NEXT execute(), as the reference does. hold = 0 (default): every call is an execute() of its own. */
int t2gpu_front_hold_iq(t2gpu_front *h, int hold);
/* a call of up to 98 304 samples (about three 32K symbols) whose resampling ratio gives one or two cells per sample runs as ONE launch
 * with one pass per workgroup (default); 0 keeps the five launches of longer calls. Same NCO phases, Farrow positions and output counts
 * bit for bit; the cells agree to the last bits of the dc averager's double-precision scan (1e-6; both within 2e-6 of the reference). */
int t2gpu_front_set_chain(t2gpu_front *h, int on);
int t2gpu_front_commit_iq(t2gpu_front *h, void *stream);
/* nominal resample = sample_rate / (SAMPLE_RATE * 2) and its limit (+100 ppm), as :54-55 computes them */
int t2gpu_front_resample(const t2gpu_front *h, double *resample, double *max_resample);
long t2gpu_front_execute_dev(t2gpu_front *h, int n_chunks, const int32_t *chunk_len, const float *phase_est_filtered,
                             const float *frequency_est_filtered, const double *arbitrary_resample, const int16_t *d_i,
                             const int16_t *d_q, float *d_out, long out_cap_cells, int32_t *chunk_out_len /* host, or NULL */,
                             void *stream);
long t2gpu_front_execute(t2gpu_front *h, int n_chunks, const int32_t *chunk_len, const float *phase_est_filtered,
                         const float *frequency_est_filtered, const double *arbitrary_resample, const int16_t *i_in,
                         const int16_t *q_in, float *out, long out_cap_cells, int32_t *chunk_out_len);
/* synchronises; out8 = {dc_real, dc_imag, c1, c2, phase_nco, frequency_nco, level_detect, farrow position x1} */
int t2gpu_front_state(t2gpu_front *h, float *out8);
/* the same eight as the LAST t2gpu_front_commit_iq left them, whatever has been launched on the handle since (waits for that commit's launch only) */
int t2gpu_front_committed_state(t2gpu_front *h, float *out8);
/* ---- the loop on the device. The reference hands a symbol's two synchronisation floats and its guard correlation to the tracking filters
 * and the filters' outputs to the next chunk's NCO (dvbt2_demodulator.cpp:165-171, 187-193, 328-330, 429-439): one host round trip per
 * symbol for a caller that keeps the filters on the host. These entry points keep them on the device for as long as the caller likes (a
 * T2 frame's data symbols in t2gpu_demod_execute): t2gpu_front_loop_begin sends the host's state up; t2gpu_sym_sync_dev /
 * t2gpu_fft_sym_sync_dev with d_loop = t2gpu_front_loop_dev(h) advance the filters behind every symbol; t2gpu_front_execute_loop_dev runs a
 * chunk whose NCO runs workgroup 0 plans from that state (the Farrow stage is planned by the host as always: its resampling value changes
 * by -8e-9, 0 or +8e-9 per symbol, so the host knows it one symbol ahead whenever the three agree as floats). Nothing waits for the host;
 * the host follows one symbol behind with the floats the launches publish (t2gpu_sync_*; t2gpu_front_loop_follow for the NCO accumulators,
 * chunk by chunk, oldest first) and t2gpu_front_loop_read lets it check that it arrived where the device did. Returns: 0 / cells, -1 on
 * an error; t2gpu_front_execute_loop_dev: -2 when the chunk does not qualify for the one-launch form (nothing has happened then). */
void *t2gpu_front_loop_dev(t2gpu_front *h);
int t2gpu_front_loop_begin(t2gpu_front *h, const float *state10, void *stream);   /* state10: t2gpu_sync_export's, [2] = the tuner (rad / sample) */
long t2gpu_front_execute_loop_dev(t2gpu_front *h, int32_t chunk, double arbitrary_resample, const int16_t *d_i, const int16_t *d_q, float *d_out,
                                  long out_cap_cells, void *stream);
int t2gpu_front_loop_follow(t2gpu_front *h, float phase_est_filtered, float frequency_est_filtered_plus_tuner);
int t2gpu_front_loop_pending(const t2gpu_front *h);
int t2gpu_front_nco(const t2gpu_front *h, float *out2);   /* the host's {phase_nco, frequency_nco}, no device access */
int t2gpu_front_loop_read(t2gpu_front *h, float *out8, void *stream);   /* {phase_nco, frequency_nco, pe, fe, frequency_est_filtered, f_int, p_int, error} */
/* intermediate streams of the last call, for tests: which 0 = de-rotated samples (n_in cells), 1 = resampled (before the
 * decimator). Synchronises. Returns the number of cells copied. Since t2gpu_front_execute runs the Farrow stage and the decimator
 * as one kernel, stream 1 holds real data in its last 63 cells only (what the next call starts from); its length is right, the
 * cells before are whatever the buffer held. t2gpu_farrow_execute is the way to look at resampled cells. Stream 0 exists behind a call
 * of the five launches only (t2gpu_front_set_chain(h, 0), or a call too long for the one-launch form): in the one-launch form the
 * de-rotated samples never leave the workgroups. */
long t2gpu_front_debug_stream(t2gpu_front *h, int which, float *out, long cap_cells);

/* Stand-alone stages with the call shape of the reference classes (host buffers; state kept in the handle):
 *   filter_decimator::execute(len_in, in, len_out, out)          DSP/filter_decimator.h:72
 *   interpolator_farrow::operator()(len_in, in, resample, len_out, out)   DSP/interpolator_farrow.hh:41 */
int t2gpu_decim_execute(t2gpu_front *h, int len_in, const float *in, float *out);
int t2gpu_farrow_execute(t2gpu_front *h, int len_in, const float *in, double arbitrary_resample, float *out, int out_cap_cells);

/* Guard-interval correlation of symbol_acquisition (dvbt2_demodulator.cpp:321-327) for a batch of buffered symbols
 * (guard + fft_size cells each, guard first): d_out[s] = {sum.re, sum.im, frequency_est, 0}. */
int t2gpu_cp_correlate_dev(const float *d_symbols, int n_symbols, int fft_size, int guard, float *d_out4, void *stream);
/* the same on symbols lying in the sample stream: symbol i starts (guard first) at first + (i / per_frame) * frame_stride +
 * (i % per_frame) * (guard + fft_size) */
int t2gpu_cp_correlate_stream_dev(const float *d_stream, long first, long frame_stride, int per_frame, int n_symbols, int fft_size,
                                  int guard, float *d_out4, void *stream);

/* Tracking loops of symbol_acquisition, host scalar state exactly as the reference keeps it (:328-330,429-439;
 * proportional_integral_loop_filter, DSP/loop_filters.hh:20-54). */
typedef struct t2gpu_sync t2gpu_sync;
t2gpu_sync *t2gpu_sync_create(float sample_rate);
void t2gpu_sync_destroy(t2gpu_sync *h);
void t2gpu_sync_frequency(t2gpu_sync *h, float frequency_est, int fft_size);
void t2gpu_sync_symbol(t2gpu_sync *h, float phase_est, float sample_rate_est);
/* out4 = {phase_est_filtered, frequency_est_filtered, sample_rate_est_filtered, arbitrary_resample of the next chunk} */
void t2gpu_sync_get(const t2gpu_sync *h, double *out4);
/* {phase_est_filtered, frequency_est_filtered, 0, f_kp, f_ki, f_int, p_kp, p_ki, p_int, old_sample_rate_est}: the filters' state (t2gpu_front_loop_begin) */
void t2gpu_sync_export(const t2gpu_sync *h, float *out10);
/* the three values t2gpu_sync_get's arbitrary_resample can take after the next t2gpu_sync_symbol (the tracker steps by -8e-9, 0 or +8e-9, :430-439) */
void t2gpu_sync_candidates(const t2gpu_sync *h, double *out3);
/* dvbt2_demodulator::reset (:111-127); frequency_est_filtered = 0 (brute-force guard search, :486); resample -= correct_resample *
 * resample after a re-tune (:291) */
void t2gpu_sync_reset(t2gpu_sync *h, float sample_rate);
void t2gpu_sync_clear_frequency(t2gpu_sync *h);
void t2gpu_sync_correct_resample(t2gpu_sync *h, double correct_resample);

/* host only (no GPU): the exact run-table expansion of the two float accumulators (csrc/front_plan.h), for tests.
 * nco: values[i] = frequency_nco used for sample i; farrow: counts[i] = outputs of input i, positions[i] = x1 at input i. */
int t2gpu_plan_nco(float *frequency_nco /* in/out */, int n, float frequency_est_filtered, float *values, int *n_runs);
long t2gpu_plan_farrow(float *x1 /* in/out */, int n, double arbitrary_resample, int32_t *counts, float *positions, int *n_runs);

/* ---------------------------------------------------------------- P1 preamble ----------------------------------------
 * Replaces  bool p1_symbol::execute(bool gain_changed, float level_detect, const int len_in, complex* in, int& consume,
 *                                   complex* buffer_sym, int& idx_buffer_sym, dvbt2_parameters& dvbt2,
 *                                   double& coarse_freq_offset, bool& p1_decoded, bool& reset)
 *           (src/DVB_T2/p1_symbol.h:33-37, src/DVB_T2/p1_symbol.cpp:75-178) and p1_symbol::demodulate (:180-298).
 * in: the decimated complex stream symbol_acquisition hands over; *consume: index of the first sample to look at on entry,
 * index behind the last consumed sample on return (the reference's by-reference argument). Returns 1 when a P1 symbol ended
 * inside the samples (the reference's return value), 0 when the input was used up, < 0 on error. The handle keeps the
 * correlator history, the thresholds and the decoded flag between calls exactly as the object does; like the reference it
 * starts a fresh correlator after every detection.
 * On detection, in[consume - idx_buffer_sym .. consume) are the samples the reference has already copied into buffer_sym
 * (the start of the P2 symbol): a caller that keeps buffer_sym copies them itself.
 * preamble / fft_mode use the reference enums (dvbt2_definition.h:115-133); -1 when this call did not decode (already decoded
 * and reset_flag = 0, or no carrier shift in 76..95 gave a consistent S1). */
typedef struct t2gpu_p1 t2gpu_p1;
typedef struct {
    int32_t detected, idx_buffer_sym, p1_decoded, preamble, fft_mode, s1, s2, shift, a_part_clipped;
    float max_correlation, arg_max[2];
    double coarse_freq_offset; /* Hz */
} t2gpu_p1_result;
t2gpu_p1 *t2gpu_p1_create(int max_samples, int device);
void t2gpu_p1_destroy(t2gpu_p1 *h);
int t2gpu_p1_reset(t2gpu_p1 *h);
/* tests: on = run the threshold / arg-max state machine (p1_symbol.cpp:93-109,162-170) sample by sample only, as the reference
 * writes it, instead of taking whole stretches in closed form; decisions are the same either way (tests/test_p1_gpu.py) */
int t2gpu_p1_set_serial_detector(t2gpu_p1 *h, int on);
int t2gpu_p1_execute_dev(t2gpu_p1 *h, int gain_changed, float level_detect, int len_in, const float *d_in, int *consume,
                         int reset_flag, t2gpu_p1_result *res, void *stream);
int t2gpu_p1_execute(t2gpu_p1 *h, int gain_changed, float level_detect, int len_in, const float *in, int *consume, int reset_flag,
                     t2gpu_p1_result *res);
/* Batch form for whole buffers of frames whose P1 positions are known to a few hundred samples (steady state: one frame length
 * after the previous P1): window w = d_stream[win_start[w] .. + win_len[w]) is searched by a FRESH correlator -- what the reference
 * has at that point too, since it clears its correlator after every detected P1 (reset_buffer, p1_symbol.cpp:133,300-311) -- with
 * the handle's thresholds and decoded flag. All windows run in one launch sequence. res[w] / consumed[w] per window (consumed
 * relative to the window start); returns the number of windows with a detection. Differences to n_windows stream calls: the
 * frequency-shift table index restarts at 0 in every window (a constant phase that cancels in the correlator output), and when
 * nothing was decoded before the call every window attempts the S1/S2 decode. */
int t2gpu_p1_execute_batch_dev(t2gpu_p1 *h, int gain_changed, float level_detect, const float *d_stream, int n_windows,
                               const long *win_start, const int *win_len, int reset_flag, t2gpu_p1_result *res, int *consumed,
                               void *stream);
/* for tests: correlation trace of the last pass (returns the number of values) and the fft-shifted 1K spectrum of part A */
int t2gpu_p1_debug(t2gpu_p1 *h, float *corr, int n_corr, float *p1_fft1024);

/* ---------------------------------------------------------------- the demodulator object ------------------------------
 * Replaces the class behind the boundary slot
 *     void dvbt2_demodulator::execute(int len_in, int16_t* i_in, int16_t* q_in, signal_estimate* signal_)
 *     (src/DVB_T2/dvbt2_demodulator.h:56-78, src/DVB_T2/dvbt2_demodulator.cpp:145-254)
 * with everything it drives up to the time de-interleaver: symbol_acquisition (:267-448: P1 search, symbol buffering, guard
 * correlation, FFT, P2 / data / frame-closing equalisers, L1-pre / L1-post, tracking loops), init_dvbt2 (:129-143), reset
 * (:111-127), set_guard_interval (:450-480) and set_guard_interval_by_brute_force (:482-548). Acquisition is the reference's:
 * the mode comes from P1 (FFT size, SISO) and L1-pre (carrier mode, guard interval, pilot pattern, PAPR, NUM_DATA_SYMBOLS);
 * like the reference the first P2 is demodulated with extended carriers and a guessed guard interval, and the P2 tables are
 * never rebuilt (a normal-carrier signal therefore never passes L1-pre, there as here).
 * i_in / q_in: host buffers as the SDR thread hands them over (AirSpy: interleaved, stride 2, dvbt2_demodulator.cpp:36-42).
 * signal_: the struct shared with the SDR thread (dvbt2_demodulator.h:42-52), same fields and meaning, bool as int32_t.
 * Signals of the class (dvbt2_demodulator.h:70-75) are C callbacks; cell pointers are host memory valid during the call:
 *   start            time_deinterleaver::start(dvbt2, l1_pre, l1_post) -- the direct call at :386
 *   l1_dyn_execute   emit l1_dyn_execute(l1_post, c_p2, cells)          :391
 *   data             emit data(c_data | n_fc, cells)                    :346,361  (from the handle's own thread while the loop is on the device, below)
 *   amount_plp       emit amount_plp(num_plp)                           :388
 *   replace_null_indicator(sample_rate_offset_hz, frequency_offset_hz)  :444
 * t2gpu_demod_execute returns 0, or -1 when a stage failed (t2gpu_last_error; e.g. P1 announces an FFT size outside 16K / 32K).
 * t2gpu_demod_set_tuner: there is no tuner behind a recorded buffer. The reference asks the SDR thread to move the local
 * oscillator by signal_->coarse_freq_offset (change_frequency, :301; rx_sdrplay.cpp:158-176); a caller replaying samples moves
 * it here instead: offset_hz is the sum of the requested moves, applied as one more NCO term. Extension, inert when never called. */
typedef struct {
    int32_t change_frequency;
    double coarse_freq_offset;
    int32_t frequency_changed; /* initial value 1 */
    int32_t change_gain, gain_offset;
    int32_t gain_changed;      /* initial value 1 */
    double correct_resample;
    int32_t reset, p1_reset;
} t2gpu_signal_estimate;
typedef struct {
    void *user;
    void (*start)(void *user, const t2gpu_l1_pre *l1_pre, const t2gpu_l1_post *l1_post, const t2gpu_l1_plp *plp, const t2gpu_l1_dyn_plp *dyn);
    void (*l1_dyn_execute)(void *user, const t2gpu_l1_post *l1_post, const t2gpu_l1_plp *plp, const t2gpu_l1_dyn_plp *dyn, int len_in,
                           const float *cells);
    void (*data)(void *user, int len_in, const float *cells);
    void (*amount_plp)(void *user, int num_plp);
    void (*replace_null_indicator)(void *user, float sample_rate_offset_hz, float frequency_offset_hz);
} t2gpu_demod_signals;
typedef struct {
    int32_t next_symbol_type /* 0 P1, 1 P2, 2 DATA, 3 FC */, p2_init, demodulator_init, deint_start, crc32_l1_pre, fft_mode, fft_size,
        guard_interval_size, symbol_size, carrier_mode, pilot_pattern, n_data, idx_symbol;
    int64_t symbols, frames, resets; /* OFDM symbols demodulated, T2 frames completed, reset() calls */
    float level_detect;
    double phase_est_filtered, frequency_est_filtered, sample_rate_est_filtered, arbitrary_resample;
    int64_t loop_resyncs;            /* the host's copies of the loops re-set from the device's state (0 by construction; T2GPU_DEMOD_STRICT_LOOPS=1: an error instead) */
} t2gpu_demod_info;
typedef struct t2gpu_demod t2gpu_demod;
t2gpu_demod *t2gpu_demod_create(int id_device, float sample_rate, int device);
void t2gpu_demod_destroy(t2gpu_demod *h);
int t2gpu_demod_connect(t2gpu_demod *h, const t2gpu_demod_signals *callbacks);   /* (not `signals`: a macro wherever Qt headers are included) */
int t2gpu_demod_execute(t2gpu_demod *h, int len_in, const int16_t *i_in, const int16_t *q_in, t2gpu_signal_estimate *signal_);
int t2gpu_demod_set_tuner(t2gpu_demod *h, double offset_hz);
/* The tracking loops of a frame's data symbols on the device (on = 1, the DEFAULT: a symbol's launches are followed by the next chunk's
 * without waiting for its results, the host reads them one symbol behind, and a thread of the handle's own makes the equaliser's launches
 * and emits the `data` signal -- "the loop on the device" above; DESIGN.md section 6: the base of 270 -> 478 Msamples/s on the slot-shaped
 * path) or on the host for every symbol (0: one round trip per symbol, every signal on the caller's thread). Same cells, same loop
 * values, same TS either way (tests/test_ref_pins_gpu.py holds both to the reference's own trajectory).
 * THREADS: with the loop on the device the `data` callback is called from the handle's own thread, concurrently with the caller's
 * thread (which calls `start`, `l1_dyn_execute`, `amount_plp`, `replace_null_indicator` and the frame-closing symbol's `data`); the
 * order of the calls is the reference's (a P2 / frame-closing symbol waits until the data symbols before it have been handed on). What
 * the callback throws (C++ callers) is caught on that thread and comes back as -1 from the next t2gpu_demod_execute / _flush / _status.
 * t2gpu_demod_execute may return with up to three symbols still to be handed on; they follow within a symbol's time, or at once with
 * t2gpu_demod_flush: everything launched so far read and handed on -- to be called at the end of a stream BEFORE the stages behind the
 * demodulator are flushed or destroyed. */
int t2gpu_demod_set_device_loop(t2gpu_demod *h, int on);
int t2gpu_demod_flush(t2gpu_demod *h);
/* With the loop on the device: the chunk that completes a data symbol and the symbol's transform + synchronisation floats as ONE launch
 * (on = 1, the default: front_fft_one_kernel -- the front end's workgroups, then the eight of the transform, which wait for them) or as two
 * (0). Same cells, same floats, same TS. Symbols whose pilot tables do not fit the transform's exchange buffer (dense patterns) take the two. */
int t2gpu_demod_set_chain_one(t2gpu_demod *h, int on);
/* Page-locked I/Q of an execute() (t2gpu_host_pin) comes over chunk by chunk (on = 1, the default): the call's first chunk's samples by a
 * launch in front of it, every later chunk's inside the launch of the chunk before it -- instead of the whole buffer by one launch in front
 * of the call's first chunk (0). Either way the caller's buffers are free when the call returns. Same samples, same TS. */
int t2gpu_demod_set_copy_ahead(t2gpu_demod *h, int on);
/* The loop trajectory (telemetry; what the reference's private members hold when it emits replace_null_indicator, dvbt2_demodulator.cpp:
 * 429-444): one record of T2GPU_DEMOD_TRACE_W doubles per P2 / data / frame-closing symbol that reaches the tracking loops, written on the
 * caller's thread when the loops have taken the symbol's floats -- {ordinal of the symbol (t2gpu_demod_info::symbols), next_symbol_type and
 * idx_symbol behind it, input samples of the chunk that completed it, phase_est_filtered, frequency_est_filtered,
 * sample_rate_est_filtered, arbitrary_resample, and the symbol's raw estimates: phase_est, sample_rate_est (the equaliser's two floats) and
 * frequency_est (the guard correlation's angle / (2 fft_size); 0 before the first good L1-pre)}. records: caller's memory for cap_records records (NULL / 0: off);
 * t2gpu_demod_trace_count: records written so far (counts on beyond cap_records without writing). */
#define T2GPU_DEMOD_TRACE_W 11
int t2gpu_demod_set_trace(t2gpu_demod *h, double *records, long cap_records);
long t2gpu_demod_trace_count(const t2gpu_demod *h);
int t2gpu_demod_status(const t2gpu_demod *h, t2gpu_demod_info *out);

/* ---------------------------------------------------------------- batch receiver: buffers of whole T2 frames --------------
 * The receive path of dvbt2_demodulator::execute (src/DVB_T2/dvbt2_demodulator.cpp:145-448) and of the stages behind it, for
 * buffers that hold n_frames whole T2 frames starting at a P1 symbol, in one object: int16 I/Q in device memory ->
 * descrambled BBFRAME bits in device memory. Front end (nominal resample, loops open) -> P1 search at every frame start ->
 * guard-interval correlation of every symbol -> FFT (guard dropped by addressing) -> P2 / data / frame-closing equalisers ->
 * time de-interleaver -> demapper -> LDPC -> BB descrambler + bit packing (K-descramble-pack) -> [host worker: per-frame L1 parse,
 * BBFRAME de-framing -> TS]. Every frame is independent given the mode, so all frames of the buffer share each launch.
 * SIMD batches are formed as the reference forms them (llr_demapper.cpp:742-764, `static int blocks`): the 32-frame LLR buffer fills
 * across TI blocks, T2 frames and CALLS, and only complete batches of ldpc_group frames go to the LDPC stage; the frames of an
 * incomplete batch wait in the handle (t2gpu_rx_carry) and lead the next call's batches, so a stream cut into calls anywhere gives the
 * batches -- hence the stop / drop decisions -- of a sequential reference. t2gpu_rx_flush_dev (end of stream; an addition, the
 * reference never decodes the tail) decodes the waiting frames as one short batch behind the rows of the last back half.
 * t2gpu_rx_reset drops the waiting frames and restarts the frame counters (a new stream on the same handle).
 * The mode is the caller's (what L1-pre / L1-post signal: fields as in dvbt2_parameters / l1_postsignalling_plp); one PLP
 * starting at cell 0 with one TI block per frame -- the reference's tested configuration; other layouts go through the
 * stage entry points (t2gpu_ti_frame_plan).
 * t2gpu_rx_front_dev / t2gpu_rx_back_dev are the two halves of t2gpu_rx_execute_dev for callers that software-pipeline buffers
 * (front half of buffer k, back half of buffer k-1); the back half works on the spectra the last front half left.
 * Returns: front 0 / back and execute the number of FEC frames decoded BY THIS CALL (whole batches: carried + new frames);
 * -1 stage error, -2 P1 missing or frames not equally spaced.
 * d_bytes_out: [fec frames][k_bch / 8] descrambled BBFRAMEs, packed MSB first (t2gpu_bbdh_execute_packed reads them);
 * d_trials_out: per SIMD batch, trials left or -1 = dropped by the reference (ldpc_decoder.cpp:264-268). Both point into the handle
 * and stay valid until the next back half.
 * t2gpu_rx_results (synchronises): P1 decisions, first P2 sample of every frame, guard correlations [frames * n_sym][4], the
 * level estimate of the last buffer (dvbt2_demodulator.cpp:235), duration of the last LDPC launch in ms. Any pointer may be NULL.
 * t2gpu_rx_fetch_packed / t2gpu_rx_fetch (synchronise): rows / trials of the last back half (and flush) into host memory, packed
 * (k_bch / 8 bytes per frame over the bus either way) or one bit per byte (unpacked on the host: the reference's bit_descramble form).
 * The host end: t2gpu_rx_ts_enable starts a worker thread in the handle; from then on every back half / flush queues, behind its
 * kernels, the copies of its packed BBFRAMEs, batch verdicts and the L1 cells of its P2 symbols to pinned memory, and the worker --
 * while the device runs later calls -- does what the reference does per frame on the host: L1-pre / L1-post parse with CRC-32
 * (p2_symbol.cpp:301-718; l1_check != 0: BBFRAMEs of a T2 frame whose CRC fails or whose signalling -- carrier mode, guard
 * interval, PAPR, pilot pattern, data symbols, L1-post size, the first PLP's modulation / code / FEC type / rotation / TI mode and
 * its dynamic PLP_NUM_BLOCKS -- differs from the handle's configuration are withheld and counted: the reference resets on an
 * L1-pre failure and pushes misplaced cells on the others, dvbt2_demodulator.cpp:373-425), the batch drop rule, and
 * bb_de_header::execute (bb_de_header.cpp:84-448). t2gpu_rx_ts_read hands out the TS bytes in order. */
typedef struct {
    int32_t id_device;                /* id_device_t: 0 SDRplay, 1 AirSpy, 2 PlutoSDR (scale and stride of the int16 input) */
    float sample_rate;                /* 0 = 64e6 / 7 */
    int32_t fft_mode, carrier_mode, pilot_pattern, guard_interval_mode, papr_mode, n_data, l1_post_size;
    int32_t plp_mod, plp_fec_type, plp_cod, plp_rotation, plp_num_blocks;
    int32_t max_frames, ldpc_group /* 0 = 32 */, ldpc_trials /* 0 = 25 */, saturate_llr /* t2gpu_demap_configure */;
} t2gpu_rx_config;
typedef struct { int32_t frame_len, n_sym, fft_size, guard_interval_size, frame_cells, fec_frames_per_t2_frame, k_bch, k_ldpc; } t2gpu_rx_geometry;
typedef struct t2gpu_rx t2gpu_rx;
t2gpu_rx *t2gpu_rx_create(const t2gpu_rx_config *cfg, int device);
void t2gpu_rx_destroy(t2gpu_rx *h);
int t2gpu_rx_info(const t2gpu_rx *h, t2gpu_rx_geometry *out);
int t2gpu_rx_front_dev(t2gpu_rx *h, const int16_t *d_i, const int16_t *d_q, int n_frames, float level_detect /* <= 0: the handle's own */,
                       int first_call, void *stream);
int t2gpu_rx_back_dev(t2gpu_rx *h, int n_frames, uint8_t **d_bytes_out, int32_t **d_trials_out, void *stream);
int t2gpu_rx_execute_dev(t2gpu_rx *h, const int16_t *d_i, const int16_t *d_q, int n_frames, float level_detect, int first_call,
                         uint8_t **d_bytes_out, int32_t **d_trials_out, void *stream);
int t2gpu_rx_flush_dev(t2gpu_rx *h, void *stream);
int t2gpu_rx_carry(const t2gpu_rx *h);
/* The decode of a call (LDPC, outer code, descrambler + packing) on a stream of the handle's own, so that the NEXT call's front end ..
 * demapper run beside it where the decoder leaves CUs free: calls of one or two T2 frames are 6 - 13 SIMD batches = 96 - 208 of the 256
 * workgroups the device keeps resident. Decodes of at most half the device (one-frame calls) also run beside EACH OTHER, two at a time:
 * three LLR buffers rotate, two decode sets (decoder state, stream, output rows) alternate; larger ones follow one another on one set.
 * Calls of fewer than 2 x 14 SIMD batches (up to four CFG-A frames) do not get a decode each: their LLR frames collect in the handle until 14
 * batches are there, and every decode is then a whole number of rounds of 14 resident batches (t2gpu_ldpc_set_max_slots) -- ONE decode
 * resident at a time, all of its slots busy for all of its time, and 32 CUs left to the front halves of the calls that run beside it
 * (two or more dispatches of other queues resident at once slow every launch of a latency-bound chain down). A call
 * then returns the FEC frames whose decode IT launched -- 0 for a call that only collected (as for a call that completes no SIMD batch).
 * Larger calls decode whole rounds of ALL the resident slots (16 for a 64800-bit code); the batches of a last, mostly empty round wait for
 * the next call's (8-frame calls: 50.5 batches = three rounds and 2.5 batches in 16 slots otherwise; 1622 -> 1850 Msamples/s). The end of
 * a stream: t2gpu_rx_flush_dev, which decodes what waits -- complete batches as the reference forms them, then the incomplete one.
 * Same results, same batch formation, same TS. With it on, a call's stream no longer covers the call (see STREAMS):
 * d_bytes_out / d_trials_out (the rows of THIS call's decode) are complete after t2gpu_rx_wait (or any fetch / results / stage_ms /
 * TS read, which wait themselves). To be switched on a drained handle with no frames waiting for a batch.
 * STREAMS: with it on, a call's work runs on FOUR STREAMS OF THE HANDLE'S OWN -- the call's chain (front end .. demapper), the side
 * stream of the P2 / frame-closing equalisers, the two decode sets -- each with a hardware queue of its own, made one after the other so
 * that they sit on four different pipes of the command processor (a chain of short launches whose queue shares a hardware queue, or only
 * a pipe, with a queue that holds a decode of milliseconds is 20 % and more slower; which queues a process's streams get depends on the
 * order it made them in: csrc/t2gpu_rx.cpp, StreamBundle). The chain waits for `stream` at entry (the inputs); NOTHING is enqueued on
 * `stream` behind the call: d_i / d_q have been read when the call returns, and everything the call produces is complete after
 * t2gpu_rx_wait (or any fetch / results / stage_ms / TS read). Whatever `stream` is -- the legacy NULL stream included -- the rate is
 * the same (one-frame calls 1.45 - 1.5 Gsamples/s with the host end on; 0.9 - 1.2 before, depending on the caller's stream). */
int t2gpu_rx_set_overlap(t2gpu_rx *h, int enable);
int t2gpu_rx_wait(t2gpu_rx *h);
int t2gpu_rx_ldpc_occupancy(const t2gpu_rx *h, int *out6);          /* t2gpu_ldpc_occupancy of the handle's decoder */
int t2gpu_rx_reset(t2gpu_rx *h);
int t2gpu_rx_results(t2gpu_rx *h, int n_frames, t2gpu_p1_result *p1, long *p2_start, float *cp4, float *level_detect, float *ldpc_ms);
int t2gpu_rx_fetch(t2gpu_rx *h, int n_fec_frames, uint8_t *bits, int32_t *trials);
int t2gpu_rx_fetch_packed(t2gpu_rx *h, int n_fec_frames, uint8_t *bytes, int32_t *trials);
typedef struct {
    int64_t t2_frames, l1_pre_crc_errors, l1_post_crc_errors, l1_mismatches;   /* per T2 frame (l1_check) */
    int64_t fec_frames, fec_frames_dropped_ldpc, fec_frames_dropped_l1;        /* per FEC frame */
    int64_t bbheader_crc_errors, ts_packet_errors, resync;                     /* de-framer messages (bb_de_header.cpp:108,218,235,...) */
    int64_t ts_bytes, ts_bytes_pending, device_errors;
} t2gpu_rx_ts_counters;
int t2gpu_rx_ts_enable(t2gpu_rx *h, int need_plp, int l1_check);
long t2gpu_rx_ts_read(t2gpu_rx *h, uint8_t *out, long cap, int wait_all);
int t2gpu_rx_ts_counters_get(t2gpu_rx *h, int wait_all, t2gpu_rx_ts_counters *out);
/* Per-stage durations of the last call, from HIP events recorded on the call's stream between the stages (waits for the call):
 * ms[T2GPU_RX_STAGES] in the order front end, P1 (incl. the call's one host round trip), guard correlation, FFT, equalisers +
 * frequency de-interleave, time / cell de-interleave, demapper, LDPC, BCH stub / descrambler + bit packing; -1 = stage not run by
 * that call. One set of events per handle: valid for a call that has drained before the next one records (the function waits for
 * both ends of every interval; with calls overlapped on several streams the intervals mix calls). */
#define T2GPU_RX_STAGES 9
int t2gpu_rx_stage_ms(t2gpu_rx *h, float *ms);
/* BASELINE.json config 2 as a call of its own: FFT (guard dropped by addressing) + P2 / data / FC equalisers with frequency
 * de-interleave + time / cell de-interleave + demapper over the n_frames frames whose stream and frame positions the last
 * t2gpu_rx_front_dev / t2gpu_rx_execute_dev left in the handle. Enqueue only. */
int t2gpu_rx_fft_eq_demap_dev(t2gpu_rx *h, int n_frames, void *stream);
/* The synchronisation sums the reference forms in every symbol (p2_symbol.cpp:253-258, data_symbol.cpp:319-324, fc_symbol.cpp:
 * 257-262) as the last call left them: (phase_offset, sample_rate_offset) per symbol, [P2 of every frame][data symbols, frame
 * major][FC of every frame]. Synchronises. */
int t2gpu_rx_sync_sums(t2gpu_rx *h, int n_frames, float *sync2);
/* opt-in (off by default = the reference's behaviour, bch_decoder.cpp:136): run t2gpu_bch_decode_dev on the LDPC output of every
 * back half before the descrambler; t2gpu_rx_outer_code_status (synchronises) copies the per-FEC-frame status of the last back
 * half (bits corrected, -1 = more than t errors; frames of batches the LDPC stage dropped carry whatever the decoder left). */
int t2gpu_rx_set_outer_code(t2gpu_rx *h, int enable);
int t2gpu_rx_outer_code_status(t2gpu_rx *h, int n_fec_frames, int32_t *status);

/* ---------------------------------------------------------------- several devices, one transport stream ---------------------------
 * The reference is one process (rx_sdrplay.cpp:199-261 -> dvbt2_demodulator::execute): what it binds on a node with several GPUs is one
 * object. A pool owns one t2gpu_rx per listed device (a device may be listed more than once) and one thread per member. t2gpu_rx_pool_execute
 * takes HOST int16 I/Q of n_frames whole T2 frames (frame after frame, each starting at its P1 symbol: what t2gpu_rx_execute_dev takes in
 * device memory; page-lock it with t2gpu_host_pin for the link's rate) and gives member k the contiguous frames t2gpu_rx_pool_share reports.
 * Share boundaries are multiples of t2gpu_rx_pool_frame_alignment -- the smallest number of T2 frames that is a whole number of the
 * reference's SIMD batches (llr_demapper.cpp:742-764) -- and n_frames must be a multiple of it (-3 otherwise): every batch of 32 FEC frames is
 * then formed, decoded or dropped (ldpc_decoder.cpp:264-268) exactly as ONE sequential receiver would, whichever device it lands on. Nothing
 * crosses between devices (no collective, no RCCL); the packed BBFRAMEs of the shares go, in member order = frame order,
 * through ONE bb_de_header state machine (a TS packet straddles BBFRAMEs, bb_de_header.cpp:166-322), and t2gpu_rx_pool_ts_read hands out
 * the transport stream: byte for byte what a single t2gpu_rx with t2gpu_rx_ts_enable(need_plp, 0) writes for the same frames.
 * cfg->max_frames bounds a member's share per call. Returns: execute the FEC frames decoded by the call; -1 a stage failed on a member
 * (t2gpu_last_error names the device), -2 as t2gpu_rx_execute_dev, -3 n_frames not aligned. counters: rows de-framed, rows dropped by the
 * LDPC rule, BBHEADER CRC errors, frames skipped, TS packets flagged, resynchronisations; the seconds the last call's de-framing took. */
typedef struct t2gpu_rx_pool t2gpu_rx_pool;
t2gpu_rx_pool *t2gpu_rx_pool_create(const t2gpu_rx_config *cfg, const int *devices, int n_devices, int need_plp);
void t2gpu_rx_pool_destroy(t2gpu_rx_pool *p);
int t2gpu_rx_pool_frame_alignment(const t2gpu_rx_pool *p);
int t2gpu_rx_pool_info(const t2gpu_rx_pool *p, t2gpu_rx_geometry *out);
int t2gpu_rx_pool_share(const t2gpu_rx_pool *p, int n_frames, int k, int *lo, int *hi);
long t2gpu_rx_pool_execute(t2gpu_rx_pool *p, const int16_t *i_in, const int16_t *q_in, int n_frames);
long t2gpu_rx_pool_ts_read(t2gpu_rx_pool *p, uint8_t *out, long cap);
int t2gpu_rx_pool_counters(const t2gpu_rx_pool *p, int64_t *out6, double *merge_seconds);

/* ---------------------------------------------------------------- mode tables (host only, no GPU needed) -----------
 * The permutations the kernels gather/scatter through, as this library builds them (for inspection and for tests):
 * bit de-interleaver address per LLR of an FEC frame (llr_demapper::address_generator, llr_demapper.cpp:110-130),
 * cell de-interleaver permutation (time_deinterleaver::address_cell_deinterleaving, time_deinterleaver.cpp:174-266),
 * BB scrambler sequence (bch_decoder::init_descrambler, bch_decoder.cpp:50-61). Return the number of entries written. */
int t2gpu_table_bitdeint(int mod, int fec_type, int code_rate, uint16_t *out /* [fec_size] */);
int t2gpu_table_cell_deint(int num_blocks, int cells_per_fec, int32_t *out /* [num_blocks * cells_per_fec] */);
int t2gpu_table_bb_prbs(uint8_t *out, int n);

#ifdef __cplusplus
}
#endif
#endif /* T2GPU_H */
