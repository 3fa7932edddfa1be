// t2gpu_rx.cpp -- the batch form of the receive path as one C-ABI object: whole buffers of T2 frames, int16 I/Q in HBM ->
// descrambled BBFRAME bits in HBM, every stage a kernel of this library, no host language above it.
//
// Call sequence per buffer (reference: /root/reference/src/DVB_T2/dvbt2_demodulator.cpp): front end (:145-226) -> P1 detection at
// every frame start (p1_symbol::execute via symbol_acquisition :279-310) -> guard-interval correlation of every symbol
// (:321-330) -> FFT with the guard dropped by addressing (:332-334) -> P2 / data / frame-closing equalisers -> time
// de-interleaver -> demapper -> LDPC -> BCH stub. The tracking loops run open: a synchronous source needs none, and what the
// loops would consume (P1 position and offset, guard correlation) is handed back. This is what bench.py times; the closed-loop,
// symbol-by-symbol form is t2gpu_demod.cpp.
#include "../../include/t2gpu.h"
#include "t2gpu_common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

using namespace t2gpu;

namespace {
constexpr int P1_LEN = 2048, L1_PRE_CELL = 1840;
}

struct t2gpu_rx {
    t2gpu_rx_config cfg{};
    int device = 0;
    t2gpu_front *front = nullptr;
    t2gpu_p1 *p1 = nullptr;
    t2gpu_ofdm *ofdm = nullptr;
    t2gpu_ti *ti = nullptr;
    t2gpu_demap *demap = nullptr;
    t2gpu_ldpc *ldpc = nullptr;
    // geometry
    int fft_size = 0, guard = 0, sym_size = 0, n_sym = 0, n_dat = 0, c_p2 = 0, c_data = 0, n_fc = 0, l_fc = 0, frame_len = 0;
    int p2_skip = 0, frame_cells = 0, cells_per_fec = 0, fec_size = 0, k_ldpc = 0, k_bch = 0, n_ti = 0, search = 0, timing_slack = 0;
    // device buffers
    float *d_stream = nullptr, *d_spec = nullptr, *d_p2_in = nullptr, *d_p2_cells = nullptr, *d_fc_cells = nullptr, *d_cells = nullptr,
          *d_ti_out = nullptr, *d_sums = nullptr, *d_cp = nullptr;
    int8_t *d_llr = nullptr;
    uint8_t *d_bits = nullptr, *d_out = nullptr;
    int32_t *d_trials = nullptr, *d_outer = nullptr;
    bool outer_code = false;
    hipEvent_t ev_ldpc0 = nullptr, ev_ldpc1 = nullptr;
    bool timed = false;
    // stage boundaries of the last call, recorded on the stream the kernels run on (t2gpu_rx_stage_ms)
    hipEvent_t ev[T2GPU_RX_STAGES + 1] = {};
    bool ev_set[T2GPU_RX_STAGES + 1] = {};
    float *d_sync = nullptr;                  // sample_rate_offset / phase_offset of every symbol (data_symbol.cpp:319-324): always computed
    std::vector<t2gpu_p1_result> p1_res;
    std::vector<long> p2_start;
};

namespace {

void free_all(t2gpu_rx *h)
{
    if (h->front) t2gpu_front_destroy(h->front);
    if (h->p1) t2gpu_p1_destroy(h->p1);
    if (h->ofdm) t2gpu_ofdm_destroy(h->ofdm);
    if (h->ti) t2gpu_ti_destroy(h->ti);
    if (h->demap) t2gpu_demap_destroy(h->demap);
    if (h->ldpc) t2gpu_ldpc_destroy(h->ldpc);
    hipFree(h->d_stream); hipFree(h->d_spec); hipFree(h->d_p2_in); hipFree(h->d_p2_cells); hipFree(h->d_fc_cells); hipFree(h->d_cells);
    hipFree(h->d_ti_out); hipFree(h->d_sums); hipFree(h->d_cp); hipFree(h->d_llr); hipFree(h->d_bits); hipFree(h->d_out); hipFree(h->d_trials); hipFree(h->d_outer);
    if (h->ev_ldpc0) hipEventDestroy(h->ev_ldpc0);
    if (h->ev_ldpc1) hipEventDestroy(h->ev_ldpc1);
    for (hipEvent_t e : h->ev) if (e) hipEventDestroy(e);
    hipFree(h->d_sync);
}

// stage k ends here: event k + 1 (event 0 = start of the call)
bool mark(t2gpu_rx *h, int k, hipStream_t s)
{
    if (hipEventRecord(h->ev[k], s) != hipSuccess) return false;
    h->ev_set[k] = true;
    return true;
}

template <class T> bool dev_alloc(T *&p, size_t count) { return hipMalloc(&p, count * sizeof(T)) == hipSuccess; }

}  // namespace

extern "C" t2gpu_rx *t2gpu_rx_create(const t2gpu_rx_config *c, int device)
{
    if (!c || c->max_frames < 1 || c->plp_num_blocks < 1) { set_error("t2gpu_rx_create: bad arguments"); return nullptr; }
    int info[12];
    if (t2gpu_ofdm_mode_info(c->fft_mode, c->carrier_mode, c->pilot_pattern, c->guard_interval_mode, c->papr_mode, c->n_data, info) != 0)
        return nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev || hipSetDevice(device) != hipSuccess) {
        set_error("t2gpu_rx_create: no usable HIP device (this library has no CPU path)");
        return nullptr;
    }
    t2gpu_rx *h = new t2gpu_rx();
    h->cfg = *c; h->device = device;
    h->fft_size = info[0]; h->c_p2 = info[5]; h->c_data = info[6]; h->n_fc = info[7]; h->l_fc = info[9]; h->guard = info[11];
    const int F = c->max_frames;
    h->n_sym = info[10];                                         // len_frame: P2 (one symbol for 16K / 32K), data symbols, frame-closing symbol
    h->n_dat = c->n_data - h->l_fc;
    h->sym_size = h->fft_size + h->guard;
    h->frame_len = P1_LEN + h->n_sym * h->sym_size;
    h->p2_skip = L1_PRE_CELL + c->l1_post_size;                  // time_deinterleaver.cpp:46,296-300
    h->frame_cells = (h->c_p2 - h->p2_skip) + h->n_dat * h->c_data + h->l_fc * h->n_fc;
    h->fec_size = c->plp_fec_type == 1 ? 64800 : 16200;
    h->cells_per_fec = h->fec_size / (2 * (c->plp_mod + 1));
    h->n_ti = c->plp_num_blocks * h->cells_per_fec;              // one PLP from cell 0, one TI block per frame
    h->search = P1_LEN + 1024;
    h->timing_slack = std::min(16, std::max(2, h->guard / 8));
    if (h->p2_skip > h->c_p2 || h->n_ti > h->frame_cells) { set_error("t2gpu_rx_create: L1 / PLP cells do not fit the frame"); delete h; return nullptr; }
    const long n_max = (long)F * h->frame_len + 4096;
    const int nb = F * c->plp_num_blocks;
    h->front = t2gpu_front_create(c->id_device, c->sample_rate > 0.0f ? c->sample_rate : 64.0e6f / 7.0f, (int)n_max, device);
    h->p1 = t2gpu_p1_create(std::max(F * 4096, 2 * h->sym_size + 8192), device);
    h->ofdm = t2gpu_ofdm_create(c->fft_mode, c->carrier_mode, c->pilot_pattern, c->guard_interval_mode, c->papr_mode, c->n_data, F * h->n_sym, device);
    h->ti = t2gpu_ti_create(c->plp_mod, c->plp_fec_type, c->plp_num_blocks, device);
    h->demap = t2gpu_demap_create(c->plp_mod, c->plp_fec_type, c->plp_cod, c->plp_rotation, h->n_ti, device);
    h->ldpc = t2gpu_ldpc_create(c->plp_fec_type, c->plp_cod, nb + 64, device);
    bool ok = h->front && h->p1 && h->ofdm && h->ti && h->demap && h->ldpc;
    if (ok) {
        ok = t2gpu_ldpc_configure(h->ldpc, c->ldpc_group > 0 ? c->ldpc_group : T2GPU_SIMD_BATCH, c->ldpc_trials > 0 ? c->ldpc_trials : 25) == 0 &&
             t2gpu_demap_configure(h->demap, c->saturate_llr) == 0 && t2gpu_ldpc_info(h->ldpc, nullptr, &h->k_ldpc, nullptr, &h->k_bch) == 0 &&
             t2gpu_ti_begin(h->ti, c->plp_num_blocks) == 0;
    }
    const int group = c->ldpc_group > 0 ? c->ldpc_group : T2GPU_SIMD_BATCH;
    ok = ok && dev_alloc(h->d_stream, 2 * (size_t)(n_max + 64)) && dev_alloc(h->d_spec, 2 * (size_t)F * h->n_sym * h->fft_size) &&
         dev_alloc(h->d_p2_in, 2 * (size_t)F * h->fft_size) && dev_alloc(h->d_p2_cells, 2 * (size_t)F * h->c_p2) &&
         dev_alloc(h->d_fc_cells, 2 * (size_t)F * std::max(h->n_fc, 1)) && dev_alloc(h->d_cells, 2 * (size_t)F * h->frame_cells) &&
         dev_alloc(h->d_ti_out, 2 * (size_t)F * h->n_ti) && dev_alloc(h->d_sums, (size_t)F * 4) && dev_alloc(h->d_cp, (size_t)F * h->n_sym * 4) &&
         dev_alloc(h->d_llr, (size_t)(nb + 64) * h->fec_size) && dev_alloc(h->d_bits, (size_t)(nb + 64) * h->k_ldpc) &&
         dev_alloc(h->d_out, (size_t)(nb + 64) * h->k_bch) && dev_alloc(h->d_trials, (size_t)(nb + 64) / group + 2) &&
         hipMemset(h->d_stream, 0, 2 * (size_t)(n_max + 64) * 4) == hipSuccess && hipMemset(h->d_cells, 0, 2 * (size_t)F * h->frame_cells * 4) == hipSuccess &&
         hipMemset(h->d_ti_out, 0, 2 * (size_t)F * h->n_ti * 4) == hipSuccess && hipMemset(h->d_sums, 0, (size_t)F * 16) == hipSuccess &&
         hipEventCreate(&h->ev_ldpc0) == hipSuccess && hipEventCreate(&h->ev_ldpc1) == hipSuccess &&
         dev_alloc(h->d_sync, 2 * (size_t)F * h->n_sym);
    for (int k = 0; ok && k <= T2GPU_RX_STAGES; ++k) ok = hipEventCreate(&h->ev[k]) == hipSuccess;
    if (!ok) {
        if (h->front && h->p1 && h->ofdm && h->ti && h->demap && h->ldpc) set_error("t2gpu_rx_create: device allocation failed");
        free_all(h);
        delete h;
        return nullptr;
    }
    h->p1_res.resize(F);
    h->p2_start.resize(F);
    return h;
}

extern "C" void t2gpu_rx_destroy(t2gpu_rx *h)
{
    if (!h) return;
    hipSetDevice(h->device);
    hipDeviceSynchronize();
    free_all(h);
    delete h;
}

extern "C" int t2gpu_rx_info(const t2gpu_rx *h, t2gpu_rx_geometry *g)
{
    if (!h || !g) { set_error("t2gpu_rx_info: bad arguments"); return -1; }
    g->frame_len = h->frame_len; g->n_sym = h->n_sym; g->fft_size = h->fft_size; g->guard_interval_size = h->guard;
    g->frame_cells = h->frame_cells; g->fec_frames_per_t2_frame = h->cfg.plp_num_blocks; g->k_bch = h->k_bch; g->k_ldpc = h->k_ldpc;
    return 0;
}

// front half: front end, P1 windows, guard correlation, FFT of n_frames frames. Synchronises once (the P1 decisions are host data).
extern "C" int t2gpu_rx_front_dev(t2gpu_rx *h, const int16_t *d_i, const int16_t *d_q, int n_frames, float level_detect, int first_call,
                                  void *stream)
{
    if (!h || !d_i || !d_q || n_frames < 1 || n_frames > h->cfg.max_frames) { set_error("t2gpu_rx_front_dev: bad arguments"); return -1; }
    T2_HIP(hipSetDevice(h->device));
    const int32_t n_in = (int32_t)((long)n_frames * h->frame_len);
    for (bool &b : h->ev_set) b = false;
    if (!mark(h, 0, (hipStream_t)stream)) return -1;
    const long cells = t2gpu_front_execute_dev(h->front, 1, &n_in, nullptr, nullptr, nullptr, d_i, d_q, h->d_stream,
                                               (long)h->cfg.max_frames * h->frame_len + 4096, nullptr, stream);
    if (cells < 0) return -1;
    if (cells != n_in) { set_error("t2gpu_rx_front_dev: the nominal resample did not give one cell per sample"); return -1; }
    if (!(level_detect > 0.0f)) {                         // what execute() hands to p1_symbol (:235,283): the previous buffer's estimate
        float st[8];
        if (t2gpu_front_state(h->front, st) != 0) return -1;
        level_detect = st[6];
    }
    if (!mark(h, 1, (hipStream_t)stream)) return -1;                                    // T2GPU_RX_STAGE_FRONT done
    std::vector<long> starts(n_frames);
    std::vector<int> lens(n_frames), cons(n_frames);
    for (int f = 0; f < n_frames; ++f) {
        starts[f] = (long)f * h->frame_len;
        lens[f] = (int)std::min<long>(h->search, n_in - starts[f]);
    }
    if (t2gpu_p1_execute_batch_dev(h->p1, first_call, level_detect, h->d_stream, n_frames, starts.data(), lens.data(), 0, h->p1_res.data(),
                                   cons.data(), stream) < 0) return -1;
    for (int f = 0; f < n_frames; ++f) {
        if (!h->p1_res[f].detected) { set_error("t2gpu_rx_front_dev: P1 not found in a frame"); return -2; }
        h->p2_start[f] = starts[f] + cons[f] - h->p1_res[f].idx_buffer_sym;
        if (std::labs(h->p2_start[f] - (h->p2_start[0] + starts[f])) > h->timing_slack) {
            set_error("t2gpu_rx_front_dev: frames are not equally spaced");
            return -2;
        }
    }
    if (!mark(h, 2, (hipStream_t)stream)) return -1;                                    // P1
    const long first = h->p2_start[0];
    if (t2gpu_cp_correlate_stream_dev(h->d_stream, first, h->frame_len, h->n_sym, n_frames * h->n_sym, h->fft_size, h->guard, h->d_cp, stream) != 0) return -1;
    if (!mark(h, 3, (hipStream_t)stream)) return -1;                                    // guard correlation
    if (t2gpu_fft_execute_strided_dev(h->ofdm, h->d_stream, first + h->guard, h->frame_len, h->n_sym, h->sym_size, h->d_spec, n_frames * h->n_sym,
                                      stream) != 0) return -1;
    if (!mark(h, 4, (hipStream_t)stream)) return -1;                                    // FFT
    return 0;
}

namespace {
// equalisers -> time de-interleaver -> demapper on the spectra in d_spec (stages 4..6)
int rx_eq_ti_demap(t2gpu_rx *h, int F, hipStream_t s)
{
    // P2, data symbols, frame-closing symbol: read in place from the spectra, written in place into the cell streams (P2 without
    // its L1 cells, time_deinterleaver.cpp:296-300). The per-symbol synchronisation sums the reference always forms
    // (p2_symbol.cpp:253-258, data_symbol.cpp:319-324, fc_symbol.cpp:257-262) are formed too; with the loops open nobody reads them.
    const int a = h->c_p2 - h->p2_skip;
    float *sy = h->d_sync;
    if (t2gpu_eq_p2_frames_dev(h->ofdm, h->d_spec, F, h->n_sym, h->d_cells, h->frame_cells, h->p2_skip, sy, s) < 0) return -1;
    if (h->n_dat > 0 && t2gpu_eq_data_frames_dev(h->ofdm, h->d_spec, F, h->n_sym, 1, h->n_dat, h->d_cells, h->frame_cells, a, sy + 2 * (size_t)F, s) < 0) return -1;
    if (h->l_fc && t2gpu_eq_fc_frames_dev(h->ofdm, h->d_spec, F, h->n_sym, h->d_cells, h->frame_cells, a + (long)h->n_dat * h->c_data,
                                          sy + 2 * (size_t)F * (1 + h->n_dat), s) < 0)
        return -1;
    if (!mark(h, 5, s)) return -1;                                                      // equalisers
    // TI block of every frame in one launch, statistics in one launch, LLRs in one launch
    if (t2gpu_ti_execute_blocks_dev(h->ti, h->d_cells, h->frame_cells, h->d_ti_out, h->n_ti, F, s) < 0) return -1;
    if (!mark(h, 6, s)) return -1;                                                      // time / cell de-interleaver
    if (t2gpu_demap_stats_batch_dev(h->demap, h->d_ti_out, h->n_ti, F, h->n_ti, 0.0f, h->d_sums, 4, s) != 0) return -1;
    if (t2gpu_demap_llr_batch_dev(h->demap, h->d_ti_out, F, h->n_ti, h->d_sums, 4, h->d_llr, s) < 0) return -1;
    if (!mark(h, 7, s)) return -1;                                                      // demapper
    return 0;
}
}  // namespace

// BASELINE config 2: FFT (guard dropped) + equalisers / frequency de-interleave + time de-interleave + demap of the frames the last
// front half left in the handle (stream, frame positions). Enqueue only.
extern "C" int t2gpu_rx_fft_eq_demap_dev(t2gpu_rx *h, int n_frames, void *stream)
{
    if (!h || n_frames < 1 || n_frames > h->cfg.max_frames) { set_error("t2gpu_rx_fft_eq_demap_dev: bad arguments"); return -1; }
    T2_HIP(hipSetDevice(h->device));
    for (bool &b : h->ev_set) b = false;
    if (!mark(h, 3, (hipStream_t)stream)) return -1;
    if (t2gpu_fft_execute_strided_dev(h->ofdm, h->d_stream, h->p2_start[0] + h->guard, h->frame_len, h->n_sym, h->sym_size, h->d_spec,
                                      n_frames * h->n_sym, stream) != 0) return -1;
    if (!mark(h, 4, (hipStream_t)stream)) return -1;
    return rx_eq_ti_demap(h, n_frames, (hipStream_t)stream);
}

// back half: equalisers, time de-interleaver, demapper, LDPC (every FEC frame, the tail batch short), descrambler
extern "C" int t2gpu_rx_back_dev(t2gpu_rx *h, int n_frames, uint8_t **d_bits_out, int32_t **d_trials_out, void *stream)
{
    if (!h || n_frames < 1 || n_frames > h->cfg.max_frames) { set_error("t2gpu_rx_back_dev: bad arguments"); return -1; }
    T2_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const int F = n_frames, nb = h->cfg.plp_num_blocks;
    if (rx_eq_ti_demap(h, F, s) != 0) return -1;
    const int count = F * nb;
    T2_HIP(hipEventRecord(h->ev_ldpc0, s));
    if (t2gpu_ldpc_execute_dev(h->ldpc, h->d_llr, count, h->d_bits, nullptr, h->d_trials, s) != 0) return -1;
    T2_HIP(hipEventRecord(h->ev_ldpc1, s));
    h->timed = true;
    if (!mark(h, 8, s)) return -1;                                                      // LDPC
    if (h->outer_code && t2gpu_bch_decode_dev(h->cfg.plp_fec_type, h->cfg.plp_cod, h->d_bits, count, h->d_outer, s) < 0) return -1;
    if (t2gpu_bch_descramble_dev(h->cfg.plp_fec_type, h->cfg.plp_cod, h->d_bits, count, h->d_out, s) < 0) return -1;
    if (!mark(h, 9, s)) return -1;                                                      // BCH stub / descrambler
    if (d_bits_out) *d_bits_out = h->d_out;
    if (d_trials_out) *d_trials_out = h->d_trials;
    return count;
}

extern "C" int t2gpu_rx_execute_dev(t2gpu_rx *h, const int16_t *d_i, const int16_t *d_q, int n_frames, float level_detect, int first_call,
                                    uint8_t **d_bits_out, int32_t **d_trials_out, void *stream)
{
    const int rc = t2gpu_rx_front_dev(h, d_i, d_q, n_frames, level_detect, first_call, stream);
    if (rc != 0) return rc;
    return t2gpu_rx_back_dev(h, n_frames, d_bits_out, d_trials_out, stream);
}

extern "C" int t2gpu_rx_results(t2gpu_rx *h, int n_frames, t2gpu_p1_result *p1, long *p2_start, float *cp4, float *level_detect,
                                float *ldpc_ms)
{
    if (!h || n_frames < 0 || n_frames > h->cfg.max_frames) { set_error("t2gpu_rx_results: bad arguments"); return -1; }
    T2_HIP(hipSetDevice(h->device));
    for (int f = 0; f < n_frames; ++f) {
        if (p1) p1[f] = h->p1_res[f];
        if (p2_start) p2_start[f] = h->p2_start[f];
    }
    if (cp4) T2_HIP(hipMemcpy(cp4, h->d_cp, (size_t)n_frames * h->n_sym * 16, hipMemcpyDeviceToHost));
    if (level_detect) {
        float st[8];
        if (t2gpu_front_state(h->front, st) != 0) return -1;
        *level_detect = st[6];
    }
    if (ldpc_ms) {
        *ldpc_ms = -1.0f;
        if (h->timed) {
            T2_HIP(hipEventSynchronize(h->ev_ldpc1));
            T2_HIP(hipEventElapsedTime(ldpc_ms, h->ev_ldpc0, h->ev_ldpc1));
        }
    }
    if (n_frames == 0) return 0;                           // a pure timing query: no read-back of the decoder's status word
    return t2gpu_ldpc_status(h->ldpc) == 0 ? 0 : -1;
}

// Durations of the stages of the last call from the events recorded between them (milliseconds; -1 for a stage the call did not
// run). Waits for the call. Stage k = time between event k and event k + 1; the P1 stage contains the one host round trip of a
// call (its decisions are host data).
extern "C" int t2gpu_rx_stage_ms(t2gpu_rx *h, float *ms)
{
    if (!h || !ms) { set_error("t2gpu_rx_stage_ms: bad arguments"); return -1; }
    T2_HIP(hipSetDevice(h->device));
    for (int k = 0; k < T2GPU_RX_STAGES; ++k) {
        ms[k] = -1.0f;
        if (!h->ev_set[k] || !h->ev_set[k + 1]) continue;
        T2_HIP(hipEventSynchronize(h->ev[k + 1]));
        T2_HIP(hipEventElapsedTime(&ms[k], h->ev[k], h->ev[k + 1]));
    }
    return 0;
}

extern "C" int t2gpu_rx_sync_sums(t2gpu_rx *h, int n_frames, float *sync2)
{
    if (!h || !sync2 || n_frames < 1 || n_frames > h->cfg.max_frames) { set_error("t2gpu_rx_sync_sums: bad arguments"); return -1; }
    T2_HIP(hipSetDevice(h->device));
    T2_HIP(hipDeviceSynchronize());
    T2_HIP(hipMemcpy(sync2, h->d_sync, (size_t)n_frames * h->n_sym * 8, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int t2gpu_rx_set_outer_code(t2gpu_rx *h, int enable)
{
    if (!h) { set_error("t2gpu_rx_set_outer_code: null handle"); return -1; }
    T2_HIP(hipSetDevice(h->device));
    if (enable && !h->d_outer) T2_HIP(hipMalloc(&h->d_outer, ((size_t)h->cfg.max_frames * h->cfg.plp_num_blocks + 64) * sizeof(int32_t)));
    h->outer_code = enable != 0;
    return 0;
}

extern "C" int t2gpu_rx_outer_code_status(t2gpu_rx *h, int n_fec_frames, int32_t *status)
{
    if (!h || !status || n_fec_frames < 0 || n_fec_frames > h->cfg.max_frames * h->cfg.plp_num_blocks || !h->outer_code) {
        set_error("t2gpu_rx_outer_code_status: bad arguments, or the outer code is not enabled on this handle");
        return -1;
    }
    T2_HIP(hipSetDevice(h->device));
    T2_HIP(hipDeviceSynchronize());
    if (n_fec_frames) T2_HIP(hipMemcpy(status, h->d_outer, (size_t)n_fec_frames * sizeof(int32_t), hipMemcpyDeviceToHost));
    return n_fec_frames;
}

extern "C" int t2gpu_rx_fetch(t2gpu_rx *h, int n_fec_frames, uint8_t *bits, int32_t *trials)
{
    if (!h || n_fec_frames < 0 || n_fec_frames > h->cfg.max_frames * h->cfg.plp_num_blocks) { set_error("t2gpu_rx_fetch: bad arguments"); return -1; }
    T2_HIP(hipSetDevice(h->device));
    T2_HIP(hipDeviceSynchronize());
    const int group = h->cfg.ldpc_group > 0 ? h->cfg.ldpc_group : T2GPU_SIMD_BATCH;
    if (bits && n_fec_frames) T2_HIP(hipMemcpy(bits, h->d_out, (size_t)n_fec_frames * h->k_bch, hipMemcpyDeviceToHost));
    if (trials && n_fec_frames) T2_HIP(hipMemcpy(trials, h->d_trials, (size_t)((n_fec_frames + group - 1) / group) * 4, hipMemcpyDeviceToHost));
    return t2gpu_ldpc_status(h->ldpc) == 0 ? 0 : -1;
}
