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
/* reference-shaped call: len_in = fec_size * n_frames (the reference always passes 32 frames) */
int t2gpu_ldpc_execute(t2gpu_ldpc *h, const int8_t *in, int len_in, uint8_t *out,
                       int *trials_left /* [ceil(n_frames/group)] */);
/* diagnostics: the first call arms per-phase cycle counters inside the kernel (off by default); later calls return the
 * sums over all workgroups of the last launch: [0] parity check, [1] batch rendezvous, [2] PLAIN, [3] PAIR, [4] GENERIC
 * layers (shader clock cycles of wave 0). out8 may be NULL. */
int t2gpu_ldpc_profile(t2gpu_ldpc *h, long long *out8);
/* after a synchronised execute: 0 = clean, 1 = a batch rendezvous timed out (results invalid) */
int t2gpu_ldpc_status(t2gpu_ldpc *h);

#ifdef __cplusplus
}
#endif
#endif /* T2GPU_H */
