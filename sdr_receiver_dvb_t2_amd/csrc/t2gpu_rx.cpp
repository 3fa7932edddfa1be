// t2gpu_rx.cpp -- the batch form of the receive path as one C-ABI object: whole buffers of T2 frames, int16 I/Q in HBM ->
// descrambled BBFRAME bits in HBM, every stage a kernel of this library, no host language above it.
//
// Call sequence per buffer (reference: /root/reference/src/DVB_T2/dvbt2_demodulator.cpp): front end (:145-226) -> P1 detection at
// every frame start (p1_symbol::execute via symbol_acquisition :279-310) -> guard-interval correlation of every symbol
// (:321-330) -> FFT with the guard dropped by addressing (:332-334) -> P2 / data / frame-closing equalisers -> time
// de-interleaver -> demapper -> LDPC -> BCH stub + bit packing -> (host worker) L1 parse + BBFRAME de-framing -> TS. The tracking
// loops run open: a synchronous source needs none, and what the loops would consume (P1 position and offset, guard correlation) is
// handed back. This is what bench.py times; the closed-loop, symbol-by-symbol form is t2gpu_demod.cpp.
//
// SIMD batches are formed as the reference forms them (llr_demapper.cpp:742-764: `static int blocks`, the 32-frame LLR buffer
// fills across TI blocks and T2 frames and only a full one is handed to the LDPC stage): the LLR frames of an incomplete batch stay
// at the head of the LLR buffer in the handle and the next call's frames are demapped behind them. t2gpu_rx_flush_dev (end of
// stream; the reference has no such thing and simply never decodes the tail) decodes the remainder as one short batch.
#include "../../include/t2gpu.h"
#include "t2gpu_common.h"
#include "fec_kernels.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

using namespace t2gpu;

namespace {
// The overlap mode's four streams -- the call's chain (front end .. demapper), the side stream of the P2 / frame-closing equalisers, the two
// decode sets -- each with a hardware queue of its own, made ONE AFTER THE OTHER. What that buys (profiles/HISTORY.md, round 6, measured with
// idle queues made ahead of the caller's / the decode's to shift them): the runtime deals plain streams round-robin onto a pool of FOUR
// hardware queues per device (a fifth stream shares a queue with the first, and a queue is served in order: a decode set's stream in the
// caller's queue put every call BEHIND the decode it was meant to run beside), and the hardware queues themselves go round-robin onto
// four pipes of the command processor in the order they were made: a latency-bound chain of short launches whose queue shares a PIPE with a
// queue that holds a decode of milliseconds (or the launches that wait for it) is ~20 % slower (one-frame calls 1.2 against 1.45
// Gsamples/s; the slot-shaped path 390 - 490 Msamples/s over the four positions of its decode queue). Streams made with a CU mask (here:
// all CUs) get queues outside the pool, one each, and four made in a row sit on four different pipes whatever the process made before.
// They are blocking streams in the legacy NULL stream's sense: a caller that works on the NULL stream serialises with the decodes
// (t2gpu.h). A destroyed handle's four are kept (per device) for the next one: making and destroying hardware queues again and again
// left later handles of a process ~10 % slower than its first.
struct StreamBundle { hipStream_t s[4] = {nullptr, nullptr, nullptr, nullptr}; };    // chain, side, decode set 0, decode set 1
std::mutex g_bundle_m;
std::vector<std::pair<int, StreamBundle>> g_bundle_free;
bool stream_bundle_acquire(StreamBundle *b, int device, int num_cu)
{
    {
        std::lock_guard<std::mutex> lk(g_bundle_m);
        for (size_t k = 0; k < g_bundle_free.size(); ++k)
            if (g_bundle_free[k].first == device) { *b = g_bundle_free[k].second; g_bundle_free.erase(g_bundle_free.begin() + (long)k); return true; }
    }
    uint32_t mask[32] = {0};
    const bool masked = num_cu > 0 && num_cu <= 1024;
    for (int c = 0; masked && c < num_cu; ++c) mask[c >> 5] |= 1u << (c & 31);
    for (int k = 0; k < 4; ++k) {
        if (masked && hipExtStreamCreateWithCUMask(&b->s[k], (uint32_t)((num_cu + 31) / 32), mask) == hipSuccess) continue;
        (void)hipGetLastError();
        if (hipStreamCreateWithFlags(&b->s[k], hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            for (int q = 0; q < k; ++q) hipStreamDestroy(b->s[q]);
            *b = StreamBundle();
            return false;
        }
    }
    return true;
}
void stream_bundle_release(StreamBundle *b, int device)
{
    if (!b->s[0]) return;
    for (hipStream_t st : b->s) (void)hipStreamSynchronize(st);
    std::lock_guard<std::mutex> lk(g_bundle_m);
    g_bundle_free.emplace_back(device, *b);
    *b = StreamBundle();
}
constexpr int P1_LEN = 2048, L1_PRE_CELL = 1840;
constexpr int ACC_BATCHES = 14;             // overlap mode, small calls: see t2gpu_rx_back_dev (14 of the 16 batch slots a 64800-bit decode can keep resident)
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
    int8_t *d_llr = nullptr;                  // the LLR rows the next back half fills (= d_llr_ab[cur])
    // t2gpu_rx_set_overlap: the decode of a call (LDPC, descrambler + packing) runs on a stream of the handle's own, so that the next
    // call's front end .. demapper run beside it where the decoder leaves CUs free (calls of one or two T2 frames: 6 - 13 SIMD batches
    // = 96 - 208 of 256 workgroups), and -- while two decodes fit the device together (one-frame calls) -- beside the decode of the call
    // before. Three LLR buffers rotate (the demapper fills one, up to two decodes read the others; the frames behind a call's last
    // complete batch are copied to the head of the next buffer), two decode sets alternate (decoder handle with its state, stream,
    // output rows): decode k runs on set k % 2 from buffer k % 3.
    bool overlap = false, overlap_ready = false;   // overlap_ready: the second decode set, the streams and the events of the mode all exist
    int8_t *d_llr_ab[3] = {nullptr, nullptr, nullptr};
    int cur = 0;                              // LLR buffer the next back half fills
    int set = 0, last_set = 0;                // decode set of the next decode (when it is a small one) / of the last decode
    t2gpu_ldpc *ldpc_s[2] = {nullptr, nullptr};
    StreamBundle bundle;                      // overlap mode's streams (see StreamBundle): [0] the call's chain, [1] the side stream, [2], [3] = dec_s
    hipStream_t dec_s[2] = {nullptr, nullptr};
    hipEvent_t ev_enter = nullptr;                              // the caller's stream -> the chain (ChainScope)
    uint8_t *d_bits_s[2] = {nullptr, nullptr}, *d_pack_s[2] = {nullptr, nullptr};
    int32_t *d_trials_s[2] = {nullptr, nullptr};
    hipEvent_t ev_demap = nullptr, ev_ti = nullptr;
    bool demap_pending = false;               // overlap mode: ev_demap was recorded on the side stream (the demapper runs there)
    hipEvent_t ev_llr_read[3] = {nullptr, nullptr, nullptr};    // the decode that read buffer b is through
    hipEvent_t ev_carry[3] = {nullptr, nullptr, nullptr};       // the waiting frames have been copied to the head of buffer b
    hipEvent_t ev_dec_done[2] = {nullptr, nullptr};             // the last decode of set s is through
    bool llr_read_set[3] = {false, false, false}, carry_set[3] = {false, false, false}, dec_done_set[2] = {false, false};
    int in_flight_wg[2] = {0, 0};             // workgroups of the last decode of each set (what may still be resident)
    int all_slots = 0;                        // overlap mode: SIMD batches a decode of this handle's code keeps resident on the whole device
    int acc_batches = ACC_BATCHES;            // overlap mode, small calls: SIMD batches that collect before a decode is launched = its resident slots (0: off)
    bool pair_allowed = true;                 // T2GPU_RX_PAIR=0 (read by t2gpu_rx_set_overlap): every decode on set 0, one after the other
    int num_cu = 0;
    uint8_t *d_bits = nullptr, *d_out = nullptr;
    int32_t *d_trials = nullptr, *d_outer = nullptr;
    bool outer_code = false;
    hipEvent_t ev_ldpc0 = nullptr, ev_ldpc1 = nullptr;
    hipStream_t side = nullptr;               // the P2 / frame-closing equalisers (one small launch per call each) run beside the data symbols'
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool timed = false;
    // stage boundaries of the last call, recorded on the stream the kernels run on (t2gpu_rx_stage_ms)
    hipEvent_t ev[T2GPU_RX_STAGES + 1] = {};
    bool ev_set[T2GPU_RX_STAGES + 1] = {};
    float *d_sync = nullptr;                  // sample_rate_offset / phase_offset of every symbol (data_symbol.cpp:319-324): always computed
    std::vector<t2gpu_p1_result> p1_res;
    std::vector<long> p2_start;
    // SIMD-batch formation across calls
    int group = T2GPU_SIMD_BATCH;
    int row_pad = 64;                       // rows beyond max_frames * blocks in every per-FEC-frame buffer: the frames carried between calls
                                            // (fewer than one group) sit in front of a call's own rows, a flush appends them behind
    int carry = 0;                            // LLR frames of the incomplete batch at the head of d_llr
    int last_ready = 0;                       // FEC frames the last back half (+ flush) decoded: rows of d_pack / entries of d_trials
    long fec_seq = 0;                         // running FEC-frame number of the first frame of the next decode
    long t2_seq = 0;                          // running T2-frame number of the first frame of the next back half
    uint8_t *d_pack = nullptr;                // descrambled BBFRAMEs, packed MSB first: [frames][k_bch / 8]
    float *d_l1 = nullptr;                    // equalised L1-pre + L1-post cells of every P2 symbol: [F][p2_skip][2]
    struct TsEnd *ts = nullptr;               // host end of the path (t2gpu_rx_ts_enable)
};

// ---- the host end: per-frame L1 parse (CRC-32 gating, dynamic signalling against the configured mode) and BBFRAME de-framing on a
// worker thread, overlapped with the device work of later calls. Replaces, for the batch form, p2_symbol::l1_pre_info / l1_post_info
// as called per P2 symbol (p2_symbol.cpp:301-718, dvbt2_demodulator.cpp:373-425) and bb_de_header::execute per BBFRAME
// (bb_de_header.cpp:84-448; the drop of undecodable SIMD batches is ldpc_decoder.cpp:264-268).
struct TsJob {
    int slot = 0, fec_frames = 0, t2_frames = 0;
    long fec_first = 0, t2_first = 0;
    bool l1_event = false;                    // the slot's L1 cells have an event of their own (they left on another stream than the rows)
};
struct TsSlot {
    uint8_t *pack = nullptr;                  // pinned: [frames][k_bch / 8]
    int32_t *trials = nullptr;                // pinned: [batches]
    float *l1 = nullptr;                      // pinned: [F][p2_skip][2]
    hipEvent_t ready = nullptr, l1_ready = nullptr;
    bool busy = false;
};
struct TsEnd {
    static constexpr int SLOTS = 12;          // jobs between a call and the worker. Large calls use 4 (two decodes in flight, one being de-framed, one
                                              // being filled); small calls collect for a decode that ends milliseconds -- several calls -- later: all 12
    int n_slots = 4;
    t2gpu_rx *rx = nullptr;
    t2gpu_bbdh *bbdh = nullptr;
    int need_plp = 0, l1_check = 0;
    TsSlot slot[SLOTS];
    std::thread worker;
    std::mutex m;
    std::condition_variable cv_job, cv_done;
    std::deque<TsJob> jobs;
    int next_slot = 0, in_flight = 0;
    bool stop = false;
    hipStream_t copy_stream = nullptr;        // plain mode: the device -> host copies run beside the next call's kernels (overlap mode: ts_submit)
    hipEvent_t decoded = nullptr;             // recorded on the call's stream behind K-descramble-pack
    hipEvent_t last_copy = nullptr;           // `ready` of the newest job: the next back half waits for it before it overwrites the rows
    hipEvent_t last_copy_s[2] = {nullptr, nullptr};   // overlap mode: the same per decode set (a decode overwrites its own set's rows only)
    struct Chunk { std::unique_ptr<uint8_t[]> p; size_t cap = 0, size = 0; };
    std::deque<Chunk> ts;                     // TS bytes not yet read: one chunk per job, oldest first
    std::vector<Chunk> pool;                  // chunks handed out completely, reused by the worker (no fresh pages per job)
    size_t ts_head = 0, ts_pending = 0;       // bytes of the first chunk already handed out / bytes waiting in all chunks
    std::deque<std::pair<long, int>> l1_status;   // (running T2-frame number, status bits) of frames whose FEC frames may still come
    t2gpu_rx_ts_counters n{};
};

namespace {

void free_all(t2gpu_rx *h)
{
    if (h->d_llr_ab[0]) {                                        // overlap mode was set up: the plain names go back to the objects they own
        h->d_llr = h->d_llr_ab[0]; h->ldpc = h->ldpc_s[0]; h->d_bits = h->d_bits_s[0]; h->d_pack = h->d_pack_s[0]; h->d_trials = h->d_trials_s[0];
    }
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
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    if (h->ev_join) hipEventDestroy(h->ev_join);
    if (h->side) hipStreamDestroy(h->side);
    for (hipEvent_t e : h->ev) if (e) hipEventDestroy(e);
    hipFree(h->d_sync); hipFree(h->d_pack); hipFree(h->d_l1);
    // overlap mode's second / third copies (index 0 of each is the plain mode's own object, freed with it)
    for (int b = 1; b < 3; ++b) hipFree(h->d_llr_ab[b]);
    if (h->ldpc_s[1]) t2gpu_ldpc_destroy(h->ldpc_s[1]);
    hipFree(h->d_bits_s[1]); hipFree(h->d_pack_s[1]); hipFree(h->d_trials_s[1]);
    stream_bundle_release(&h->bundle, h->device);
    if (h->ev_enter) hipEventDestroy(h->ev_enter);
    for (hipEvent_t e : {h->ev_demap, h->ev_ti, h->ev_llr_read[0], h->ev_llr_read[1], h->ev_llr_read[2], h->ev_carry[0], h->ev_carry[1], h->ev_carry[2],
                         h->ev_dec_done[0], h->ev_dec_done[1]}) if (e) hipEventDestroy(e);
}

void ts_stop(t2gpu_rx *h)
{
    TsEnd *t = h->ts;
    if (!t) return;
    {
        std::lock_guard<std::mutex> lk(t->m);
        t->stop = true;
    }
    t->cv_job.notify_all();
    if (t->worker.joinable()) t->worker.join();
    for (TsSlot &sl : t->slot) {
        if (sl.pack) hipHostFree(sl.pack);
        if (sl.trials) hipHostFree(sl.trials);
        if (sl.l1) hipHostFree(sl.l1);
        if (sl.ready) hipEventDestroy(sl.ready);
        if (sl.l1_ready) hipEventDestroy(sl.l1_ready);
    }
    if (t->bbdh) t2gpu_bbdh_destroy(t->bbdh);
    if (t->copy_stream) hipStreamDestroy(t->copy_stream);
    if (t->decoded) hipEventDestroy(t->decoded);
    delete t;
    h->ts = nullptr;
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
    const int group = c->ldpc_group > 0 ? c->ldpc_group : T2GPU_SIMD_BATCH;
    // rows beyond a call's own: the frames carried between calls. Plain mode: fewer than one group. Overlap mode with small calls: up to
    // ACC_BATCHES batches collect before a decode is launched, a decode takes at most two rounds of them, and a flush appends what is left
    // behind the last decode's rows: 3 x ACC_BATCHES x group + one group.
    const int pad = std::max(64, group) + 3 * ACC_BATCHES * group;
    h->group = group;
    h->row_pad = pad;
    h->ldpc = t2gpu_ldpc_create(c->plp_fec_type, c->plp_cod, nb + pad, device);
    bool ok = h->front && h->p1 && h->ofdm && h->ti && h->demap && h->ldpc;
    if (ok) {
        ok = t2gpu_ldpc_configure(h->ldpc, c->ldpc_group > 0 ? c->ldpc_group : T2GPU_SIMD_BATCH, c->ldpc_trials > 0 ? c->ldpc_trials : 25) == 0 &&
             t2gpu_demap_configure(h->demap, c->saturate_llr) == 0 && t2gpu_ldpc_info(h->ldpc, nullptr, &h->k_ldpc, nullptr, &h->k_bch) == 0 &&
             t2gpu_ti_begin(h->ti, c->plp_num_blocks) == 0;
    }
    ok = ok && dev_alloc(h->d_pack, (size_t)(nb + pad) * (h->k_bch / 8)) && dev_alloc(h->d_l1, 2 * (size_t)F * std::max(h->p2_skip, 1));
    ok = ok && dev_alloc(h->d_stream, 2 * (size_t)(n_max + 64)) && dev_alloc(h->d_spec, 2 * (size_t)F * h->n_sym * h->fft_size) &&
         dev_alloc(h->d_p2_in, 2 * (size_t)F * h->fft_size) && dev_alloc(h->d_p2_cells, 2 * (size_t)F * h->c_p2) &&
         dev_alloc(h->d_fc_cells, 2 * (size_t)F * std::max(h->n_fc, 1)) && dev_alloc(h->d_cells, 2 * (size_t)F * h->frame_cells) &&
         dev_alloc(h->d_ti_out, 2 * (size_t)F * h->n_ti) && dev_alloc(h->d_sums, (size_t)F * 4) && dev_alloc(h->d_cp, (size_t)F * h->n_sym * 4) &&
         dev_alloc(h->d_llr, (size_t)(nb + pad) * h->fec_size) && dev_alloc(h->d_bits, (size_t)(nb + pad) * h->k_ldpc) &&
         dev_alloc(h->d_out, (size_t)(nb + pad) * h->k_bch) && dev_alloc(h->d_trials, (size_t)(nb + pad) / group + 2) &&
         hipMemset(h->d_stream, 0, 2 * (size_t)(n_max + 64) * 4) == hipSuccess && hipMemset(h->d_cells, 0, 2 * (size_t)F * h->frame_cells * 4) == hipSuccess &&
         hipMemset(h->d_ti_out, 0, 2 * (size_t)F * h->n_ti * 4) == hipSuccess && hipMemset(h->d_sums, 0, (size_t)F * 16) == hipSuccess &&
         hipEventCreate(&h->ev_ldpc0) == hipSuccess && hipEventCreate(&h->ev_ldpc1) == hipSuccess &&
         hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) == hipSuccess &&
         hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking) == hipSuccess &&
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
    ts_stop(h);
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

namespace {
// Overlap mode: a call's work goes on the handle's own chain stream (StreamBundle), which waits for the caller's stream at entry (the
// inputs). Nothing is put on the caller's stream behind the call: a wait there for the call's end is a queue with a blocked packet in it
// for most of every call, and on the wrong pipe that alone cost 12 % (1278 against 1450 Msamples/s). The inputs have been read when the
// call returns (the front end runs ahead of the call's one host round trip); everything else is complete after t2gpu_rx_wait (or any
// fetch / results / stage_ms / TS read), as the decode's rows have been since round 5.
struct ChainScope {
    t2gpu_rx *h;
    hipStream_t caller, s;
    bool on = false;
    ChainScope(t2gpu_rx *h_, void *stream) : h(h_), caller((hipStream_t)stream), s((hipStream_t)stream)
    {
        if (!h || !h->overlap || !h->bundle.s[0]) return;
        // (the legacy NULL stream: the chain, a blocking stream in its sense, is ordered behind it by the runtime already -- and an event
        // recorded THERE would wait for every decode in flight)
        if (caller == nullptr || (hipEventRecord(h->ev_enter, caller) == hipSuccess && hipStreamWaitEvent(h->bundle.s[0], h->ev_enter, 0) == hipSuccess)) { s = h->bundle.s[0]; on = true; }
        else (void)hipGetLastError();
    }
    ChainScope(const ChainScope &) = delete;
    ChainScope &operator=(const ChainScope &) = delete;
};
int rx_front(t2gpu_rx *h, const int16_t *d_i, const int16_t *d_q, int n_frames, float level_detect, int first_call, void *stream);
int rx_back(t2gpu_rx *h, int n_frames, uint8_t **d_bytes_out, int32_t **d_trials_out, void *stream);
}  // namespace

// front half: front end, P1 windows, guard correlation, FFT of n_frames frames. Synchronises once (the P1 decisions are host data).
extern "C" int t2gpu_rx_front_dev(t2gpu_rx *h, const int16_t *d_i, const int16_t *d_q, int n_frames, float level_detect, int first_call,
                                  void *stream)
{
    if (!h || !d_i || !d_q || n_frames < 1 || n_frames > h->cfg.max_frames) { set_error("t2gpu_rx_front_dev: bad arguments"); return -1; }
    T2_HIP(hipSetDevice(h->device));
    ChainScope sc(h, stream);
    return rx_front(h, d_i, d_q, n_frames, level_detect, first_call, sc.s);
}
namespace {
int rx_front(t2gpu_rx *h, const int16_t *d_i, const int16_t *d_q, int n_frames, float level_detect, int first_call, void *stream)
{
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
}  // namespace

namespace {
// equalisers -> time de-interleaver -> demapper on the spectra in d_spec (stages 4..6); the LLR frames land behind the `llr_at`
// frames already waiting in d_llr
// demap_s (overlap mode): the demapper -- the statistics' two sequential walks (one workgroup per TI block and sum: 260 us of latency
// whatever the call's size) and the LLR pass -- on that stream instead of s, behind the de-interleaver: the NEXT call's front end and
// equalisers do not depend on them and no longer stand behind them. The caller records ev_demap on demap_s; the next de-interleaver
// launch (which overwrites the cells and terms the demapper reads) waits for it.
int rx_eq_ti_demap(t2gpu_rx *h, int F, int llr_at, hipStream_t s, hipStream_t demap_s = nullptr)
{
    // P2, data symbols, frame-closing symbol: read in place from the spectra, written in place into the cell streams (P2 without
    // its L1 cells, time_deinterleaver.cpp:296-300; those go to d_l1 for the host's per-frame L1 parse). The per-symbol
    // synchronisation sums the reference always forms (p2_symbol.cpp:253-258, data_symbol.cpp:319-324, fc_symbol.cpp:257-262) are
    // formed too; with the loops open nobody reads them.
    const int a = h->c_p2 - h->p2_skip;
    float *sy = h->d_sync;
    // The P2 launch (one symbol per frame: fewer workgroups than CUs, its synchronisation sums a chain of 4640 pilots per symbol) and
    // the frame-closing one run on a side stream beside the data symbols' launch: disjoint symbols, cells and scratch.
    hipStream_t side = h->overlap && h->bundle.s[1] ? h->bundle.s[1] : h->side;
    T2_HIP(hipEventRecord(h->ev_fork, s));
    T2_HIP(hipStreamWaitEvent(side, h->ev_fork, 0));
    if (t2gpu_eq_p2_frames_l1_dev(h->ofdm, h->d_spec, F, h->n_sym, h->d_cells, h->frame_cells, h->p2_skip, h->d_l1, sy, side) < 0) return -1;
    if (h->l_fc && t2gpu_eq_fc_frames_dev(h->ofdm, h->d_spec, F, h->n_sym, h->d_cells, h->frame_cells, a + (long)h->n_dat * h->c_data,
                                          sy + 2 * (size_t)F * (1 + h->n_dat), side) < 0)
        return -1;
    T2_HIP(hipEventRecord(h->ev_join, side));
    if (h->n_dat > 0 && t2gpu_eq_data_frames_dev(h->ofdm, h->d_spec, F, h->n_sym, 1, h->n_dat, h->d_cells, h->frame_cells, a, sy + 2 * (size_t)F, s) < 0) return -1;
    T2_HIP(hipStreamWaitEvent(s, h->ev_join, 0));
    if (!mark(h, 5, s)) return -1;                                                      // equalisers
    // TI block of every frame in one launch, statistics (exact sequential sums, one workgroup per TI block) in one launch, LLRs in one launch
    // the de-interleaver forms the statistics terms of the cells it writes (one pass over the cells for both)
    if (demap_s && h->demap_pending) T2_HIP(hipStreamWaitEvent(s, h->ev_demap, 0));     // the previous call's demapper still reads d_ti_out / the terms
    const int with_terms = t2gpu_ti_execute_blocks_terms_dev(h->ti, h->demap, h->d_cells, h->frame_cells, h->d_ti_out, h->n_ti, F, s);
    if (with_terms < 0) return -1;
    if (!mark(h, 6, s)) return -1;                                                      // time / cell de-interleaver
    if (demap_s && demap_s != s) {
        T2_HIP(hipEventRecord(h->ev_ti, s));
        T2_HIP(hipStreamWaitEvent(demap_s, h->ev_ti, 0));
        s = demap_s;
        h->demap_pending = true;
    }
    if (with_terms ? t2gpu_demap_stats_terms_dev(h->demap, F, h->n_ti, 0.0f, h->d_sums, 4, s) != 0
                   : t2gpu_demap_stats_batch_dev(h->demap, h->d_ti_out, h->n_ti, F, h->n_ti, 0.0f, h->d_sums, 4, s) != 0)
        return -1;
    if (t2gpu_demap_llr_batch_dev(h->demap, h->d_ti_out, F, h->n_ti, h->d_sums, 4, h->d_llr + (size_t)llr_at * h->fec_size, s) < 0) return -1;
    if (!mark(h, 7, s)) return -1;                                                      // demapper
    return 0;
}

// LDPC + (opt-in outer code) + K-descramble-pack of d_llr[0 .. count) into rows `at`.. of d_bits / d_pack, trials from batch `at / group`
int rx_decode(t2gpu_rx *h, int count, int at, hipStream_t s, const int8_t *llr = nullptr)
{
    // (overlap mode: h->ldpc / d_bits / d_pack / d_trials name the decode set this decode runs on)
    uint8_t *bits = h->d_bits + (size_t)at * h->k_ldpc;
    T2_HIP(hipEventRecord(h->ev_ldpc0, s));
    if (t2gpu_ldpc_execute_dev(h->ldpc, llr ? llr : h->d_llr, count, bits, nullptr, h->d_trials + at / h->group, s) != 0) return -1;
    T2_HIP(hipEventRecord(h->ev_ldpc1, s));
    h->timed = true;
    if (!mark(h, 8, s)) return -1;                                                      // LDPC
    if (h->outer_code && t2gpu_bch_decode_dev(h->cfg.plp_fec_type, h->cfg.plp_cod, bits, count, h->d_outer + at, s) < 0) return -1;
    if (t2gpu_bch_descramble_pack_dev(h->cfg.plp_fec_type, h->cfg.plp_cod, bits, count, h->d_pack + (size_t)at * (h->k_bch / 8), s) < 0) return -1;
    if (!mark(h, 9, s)) return -1;                                                      // BCH stub / descrambler + bit packing
    return 0;
}

int ts_submit(t2gpu_rx *h, int fec_frames, int at, int t2_frames, hipStream_t s, hipStream_t call_s = nullptr, int set = -1);
}  // namespace

// BASELINE config 2: FFT (guard dropped) + equalisers / frequency de-interleave + time de-interleave + demap of the frames the last
// front half left in the handle (stream, frame positions). Enqueue only. The LLRs overwrite the head of the LLR buffer: not to be
// mixed with back halves on a handle that carries an incomplete batch.
extern "C" int t2gpu_rx_fft_eq_demap_dev(t2gpu_rx *h, int n_frames, void *stream)
{
    if (!h || n_frames < 1 || n_frames > h->cfg.max_frames) { set_error("t2gpu_rx_fft_eq_demap_dev: bad arguments"); return -1; }
    T2_HIP(hipSetDevice(h->device));
    ChainScope sc(h, stream);
    stream = sc.s;
    for (bool &b : h->ev_set) b = false;
    if (!mark(h, 3, (hipStream_t)stream)) return -1;
    if (t2gpu_fft_execute_strided_dev(h->ofdm, h->d_stream, h->p2_start[0] + h->guard, h->frame_len, h->n_sym, h->sym_size, h->d_spec,
                                      n_frames * h->n_sym, stream) != 0) return -1;
    if (!mark(h, 4, (hipStream_t)stream)) return -1;
    h->carry = 0;
    return rx_eq_ti_demap(h, n_frames, 0, (hipStream_t)stream);
}

// back half: equalisers, time de-interleaver, demapper; then LDPC + descrambler + bit packing of every COMPLETE SIMD batch (the
// frames waiting from earlier calls first), the rest waits in the handle. Returns the number of FEC frames decoded by this call.
extern "C" int t2gpu_rx_back_dev(t2gpu_rx *h, int n_frames, uint8_t **d_bytes_out, int32_t **d_trials_out, void *stream)
{
    if (!h || n_frames < 1 || n_frames > h->cfg.max_frames) { set_error("t2gpu_rx_back_dev: bad arguments"); return -1; }
    T2_HIP(hipSetDevice(h->device));
    ChainScope sc(h, stream);
    return rx_back(h, n_frames, d_bytes_out, d_trials_out, sc.s);
}
namespace {
int rx_back(t2gpu_rx *h, int n_frames, uint8_t **d_bytes_out, int32_t **d_trials_out, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    const int F = n_frames, nb = h->cfg.plp_num_blocks;
    const int total = h->carry + F * nb;
    int ready = (total / h->group) * h->group, rest = total - ready;
    // Overlap mode, calls of fewer than 2 K batches (K = acc_batches): the frames collect until K batches are there, and a decode is a whole
    // number of rounds of K resident batches -- one dispatch at a time with every slot busy, 16 - K slots' CUs left to the front halves
    // that run beside it. (Larger calls: as before, every complete batch at once on all the device holds.)
    const bool collect = h->overlap && h->acc_batches > 0 && !h->outer_code && F * nb < 2 * h->acc_batches * h->group;
    if (collect) {
        const int k_rows = h->acc_batches * h->group;
        ready = (total / k_rows) * k_rows;
        rest = total - ready;
    } else if (h->overlap && h->acc_batches > 0 && !h->outer_code && h->all_slots > 0 && h->all_slots <= 3 * ACC_BATCHES) {
        // larger calls: whole rounds of ALL the slots -- the batches of a round that would run with most slots empty wait for the next call's
        // (8-frame calls end on a round of 2.5 batches in 16 slots otherwise)
        const int k_rows = h->all_slots * h->group;
        if (total >= k_rows) { ready = (total / k_rows) * k_rows; rest = total - ready; }
    }
    if (h->overlap) {
        // ---- the decode on a stream of the handle's own (t2gpu_rx_set_overlap)
        const int b = h->cur, nb_ = (h->cur + 1) % 3;
        // a decode that could not run beside another one like it (more than half the device) stays on set 0 and is launched
        // cooperatively, as the one-stream schedule of round 4's first form did; smaller ones alternate between the sets, plain launches
        for (t2gpu_ldpc *l : h->ldpc_s) if (l) t2gpu_ldpc_set_max_slots(l, collect ? h->acc_batches : 0);
        const int wg_est = ready > 0 ? (h->outer_code ? h->num_cu : t2gpu_ldpc_launch_workgroups(h->ldpc_s[0], ready)) : 0;
        const bool pair_ok = ready > 0 && !collect && h->pair_allowed && 2 * wg_est <= h->num_cu;
        const int set = pair_ok ? h->set : 0;
        int8_t *llr = h->d_llr_ab[b];
        // this buffer was last read by the decode of three calls ago (d_l1: copied out on this stream by the previous call's ts_submit)
        if (h->llr_read_set[b]) T2_HIP(hipStreamWaitEvent(s, h->ev_llr_read[b], 0));
        hipStream_t dm = h->bundle.s[1] ? h->bundle.s[1] : s;                           // the demapper on the side stream
        if (rx_eq_ti_demap(h, F, h->carry, s, dm) != 0) return -1;
        T2_HIP(hipEventRecord(h->ev_demap, dm));
        if (ready > 0) {
            hipStream_t d = h->dec_s[set];
            h->ldpc = h->ldpc_s[set]; h->d_bits = h->d_bits_s[set]; h->d_pack = h->d_pack_s[set]; h->d_trials = h->d_trials_s[set];
            T2_HIP(hipStreamWaitEvent(d, h->ev_demap, 0));
            // the frames waiting at the head of this buffer were put there by the previous decode's stream
            if (h->carry > 0 && h->carry_set[b]) T2_HIP(hipStreamWaitEvent(d, h->ev_carry[b], 0));
            // the host end's copies of this set's previous decode read the rows this decode overwrites
            if (h->ts && h->ts->last_copy_s[set]) T2_HIP(hipStreamWaitEvent(d, h->ts->last_copy_s[set], 0));
            // two decodes side by side only while both fit the device (the frames of a batch meet at every sweep: partly resident grids
            // of two launches would wait for each other); otherwise behind the other set's decode, as one stream would order them
            // (the opt-in outer code keeps its per-frame status in one buffer for both sets: its decodes go one after the other)
            const int wg = wg_est;
            if (wg < 0) return -1;
            t2gpu_ldpc_set_plain_launch(h->ldpc, 1);
            if (h->dec_done_set[1 - set] && wg + h->in_flight_wg[1 - set] > h->num_cu) T2_HIP(hipStreamWaitEvent(d, h->ev_dec_done[1 - set], 0));
            if (rest > 0) {
                // the frames behind the last complete batch go to the head of the next buffer (its last reader: the decode of two calls ago)
                if (h->llr_read_set[nb_]) T2_HIP(hipStreamWaitEvent(d, h->ev_llr_read[nb_], 0));
                T2_HIP(hipMemcpyAsync(h->d_llr_ab[nb_], llr + (size_t)ready * h->fec_size, (size_t)rest * h->fec_size, hipMemcpyDeviceToDevice, d));
                T2_HIP(hipEventRecord(h->ev_carry[nb_], d));
                h->carry_set[nb_] = true;
            }
            if (rx_decode(h, ready, 0, d, llr) != 0) return -1;
            T2_HIP(hipEventRecord(h->ev_llr_read[b], d));
            T2_HIP(hipEventRecord(h->ev_dec_done[set], d));
            h->llr_read_set[b] = true; h->dec_done_set[set] = true; h->in_flight_wg[set] = wg;
            h->cur = nb_;
            h->d_llr = h->d_llr_ab[h->cur];
            h->set = pair_ok ? 1 - set : 0;
            h->last_set = set;
            h->carry = rest;
            h->last_ready = ready;
            if (h->ts && ts_submit(h, ready, 0, F, d, s, set) != 0) return -1;
        } else {
            // nothing to decode yet: the frames stay where they are; the host end still gets the call's L1 cells
            h->carry = rest;
            h->last_ready = 0;
            if (h->ts && ts_submit(h, 0, 0, F, s, s, -1) != 0) return -1;
        }
        h->fec_seq += ready;
        h->t2_seq += F;
        if (d_bytes_out) *d_bytes_out = h->d_pack;
        if (d_trials_out) *d_trials_out = h->d_trials;
        return ready;
    }
    // the host end's copies of the previous decode run on their own stream: they must be through before this call overwrites the rows
    if (h->ts && h->ts->last_copy) T2_HIP(hipStreamWaitEvent(s, h->ts->last_copy, 0));
    if (rx_eq_ti_demap(h, F, h->carry, s) != 0) return -1;
    if (ready > 0 && rx_decode(h, ready, 0, s) != 0) return -1;
    if (rest > 0 && ready > 0)                                   // ready >= group > rest: source and destination do not overlap
        T2_HIP(hipMemcpyAsync(h->d_llr, h->d_llr + (size_t)ready * h->fec_size, (size_t)rest * h->fec_size, hipMemcpyDeviceToDevice, s));
    h->carry = rest;
    h->last_ready = ready;
    if (h->ts && ts_submit(h, ready, 0, F, s) != 0) return -1;
    h->fec_seq += ready;
    h->t2_seq += F;
    if (d_bytes_out) *d_bytes_out = h->d_pack;
    if (d_trials_out) *d_trials_out = h->d_trials;
    return ready;
}

}  // namespace

// End of stream: the frames still waiting for a full batch are decoded as one short batch (an addition: the reference never decodes
// them). Their rows follow those of the last back half in the output buffers. Returns the number of FEC frames flushed.
extern "C" int t2gpu_rx_flush_dev(t2gpu_rx *h, void *stream)
{
    if (!h) { set_error("t2gpu_rx_flush_dev: null handle"); return -1; }
    T2_HIP(hipSetDevice(h->device));
    const int n = h->carry;
    if (n == 0) return 0;
    ChainScope sc(h, stream);
    stream = sc.s;
    if (h->overlap) {
        // the waiting frames sit at the head of buffer `cur` (copied there on a decode stream). They are decoded on the set of the last
        // decode, on its stream and so behind it, and their rows follow that decode's rows, as in the plain schedule
        const int b = h->cur, set = h->last_set;
        const int at = h->dec_done_set[set] ? h->last_ready : 0;
        hipStream_t d = h->dec_s[set];
        h->ldpc = h->ldpc_s[set]; h->d_bits = h->d_bits_s[set]; h->d_pack = h->d_pack_s[set]; h->d_trials = h->d_trials_s[set];
        if (h->carry_set[b]) T2_HIP(hipStreamWaitEvent(d, h->ev_carry[b], 0));
        // the last call's demapper may have put frames there itself (calls without a full batch): it ran on the side stream, ev_demap is its end
        if (!h->demap_pending) T2_HIP(hipEventRecord(h->ev_demap, (hipStream_t)stream));
        T2_HIP(hipStreamWaitEvent(d, h->ev_demap, 0));
        if (h->dec_done_set[1 - set]) T2_HIP(hipStreamWaitEvent(d, h->ev_dec_done[1 - set], 0));   // the end of a stream: nothing runs beside it
        // the set's output rows may still be on their way to the host from an earlier decode (a last call that formed no batch puts the
        // flushed rows at 0, over them): behind that copy, as the plain flush and the overlapped back half do
        if (h->ts && h->ts->last_copy_s[set]) T2_HIP(hipStreamWaitEvent(d, h->ts->last_copy_s[set], 0));
        if (rx_decode(h, n, at, d, h->d_llr_ab[b]) != 0) return -1;
        T2_HIP(hipEventRecord(h->ev_llr_read[b], d));
        T2_HIP(hipEventRecord(h->ev_dec_done[set], d));
        h->llr_read_set[b] = true; h->dec_done_set[set] = true; h->in_flight_wg[set] = h->num_cu;
        h->carry = 0;
        h->last_ready = at + n;
        if (h->ts && ts_submit(h, n, at, 0, d, d, set) != 0) return -1;
        h->fec_seq += n;
        return n;
    }
    hipStream_t s = (hipStream_t)stream;
    const int at = h->last_ready;                                // a multiple of the group: trials of the short batch at at / group
    if (h->ts && h->ts->last_copy) T2_HIP(hipStreamWaitEvent(s, h->ts->last_copy, 0));
    if (rx_decode(h, n, at, s) != 0) return -1;
    h->carry = 0;
    h->last_ready = at + n;
    if (h->ts && ts_submit(h, n, at, 0, s) != 0) return -1;
    h->fec_seq += n;
    return n;
}

extern "C" int t2gpu_rx_carry(const t2gpu_rx *h) { return h ? h->carry : -1; }

// Decode of a call beside the next call's front half (see the members). To be set on a drained handle with nothing waiting.
extern "C" int t2gpu_rx_set_overlap(t2gpu_rx *h, int enable)
{
    if (!h) { set_error("t2gpu_rx_set_overlap: null handle"); return -1; }
    T2_HIP(hipSetDevice(h->device));
    T2_HIP(hipDeviceSynchronize());
    if (h->carry != 0) { set_error("t2gpu_rx_set_overlap: frames are waiting for a batch (flush or reset first)"); return -1; }
    if (enable && !h->overlap_ready) {
        // (overlap_ready only once everything below exists: a call that failed half way starts over and frees nothing twice)
        const t2gpu_rx_config &c = h->cfg;
        const size_t rows = (size_t)c.max_frames * c.plp_num_blocks + h->row_pad;
        hipDeviceProp_t prop;
        T2_HIP(hipGetDeviceProperties(&prop, h->device));
        h->num_cu = prop.multiProcessorCount;
        h->d_llr_ab[0] = h->d_llr; h->ldpc_s[0] = h->ldpc; h->d_bits_s[0] = h->d_bits; h->d_pack_s[0] = h->d_pack; h->d_trials_s[0] = h->d_trials;
        bool ok = true;
        for (int b = 1; b < 3; ++b) ok = ok && (h->d_llr_ab[b] || hipMalloc(&h->d_llr_ab[b], rows * h->fec_size) == hipSuccess);
        ok = ok && (h->d_bits_s[1] || hipMalloc(&h->d_bits_s[1], rows * h->k_ldpc) == hipSuccess) &&
             (h->d_pack_s[1] || hipMalloc(&h->d_pack_s[1], rows * (h->k_bch / 8)) == hipSuccess) &&
             (h->d_trials_s[1] || hipMalloc(&h->d_trials_s[1], (rows / h->group + 2) * sizeof(int32_t)) == hipSuccess);
        if (ok && !h->ldpc_s[1]) {
            h->ldpc_s[1] = t2gpu_ldpc_create(c.plp_fec_type, c.plp_cod, (int)rows, h->device);
            ok = h->ldpc_s[1] && t2gpu_ldpc_configure(h->ldpc_s[1], h->group, c.ldpc_trials > 0 ? c.ldpc_trials : 25) == 0;
        }
        // (default priority: at the lowest one the next call's front half, issued later, was dispatched ahead of the decode it is meant
        // to run beside -- 2- to 16-frame calls lost 2 - 7 %)
        ok = ok && (h->bundle.s[0] || stream_bundle_acquire(&h->bundle, h->device, h->num_cu));
        if (ok) { h->dec_s[0] = h->bundle.s[2]; h->dec_s[1] = h->bundle.s[3]; }
        ok = ok && (h->ev_enter || hipEventCreateWithFlags(&h->ev_enter, hipEventDisableTiming) == hipSuccess);
        for (hipEvent_t *e : {&h->ev_demap, &h->ev_ti, &h->ev_llr_read[0], &h->ev_llr_read[1], &h->ev_llr_read[2], &h->ev_carry[0], &h->ev_carry[1],
                              &h->ev_carry[2], &h->ev_dec_done[0], &h->ev_dec_done[1]})
            ok = ok && (*e || hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess);
        if (!ok) { (void)hipGetLastError(); set_error("t2gpu_rx_set_overlap: allocation failed"); return -1; }
        h->overlap_ready = true;
    }
    if (h->overlap_ready) {
        // from a known state either way: buffer 0, set 0, nothing in flight (the device is idle here)
        h->cur = 0; h->set = 0; h->last_set = 0;
        h->d_llr = h->d_llr_ab[0]; h->ldpc = h->ldpc_s[0]; h->d_bits = h->d_bits_s[0]; h->d_pack = h->d_pack_s[0]; h->d_trials = h->d_trials_s[0];
        for (bool &v : h->llr_read_set) v = false;
        for (bool &v : h->carry_set) v = false;
        h->dec_done_set[0] = h->dec_done_set[1] = false; h->demap_pending = false;
        h->in_flight_wg[0] = h->in_flight_wg[1] = 0;
        if (h->ts) h->ts->last_copy_s[0] = h->ts->last_copy_s[1] = nullptr;
        for (t2gpu_ldpc *l : h->ldpc_s) if (l) t2gpu_ldpc_set_plain_launch(l, 0);    // (decided per decode in overlap mode)
    }
    h->overlap = enable != 0;
    if (const char *e = std::getenv("T2GPU_RX_PAIR")) h->pair_allowed = std::atoi(e) != 0;
    {
        // the slots a collected decode keeps resident: what the device holds less 32 CUs' worth, at most ACC_BATCHES (the rows were sized for that)
        int occ[6] = {0, 0, 0, 0, 0, 0};
        if (t2gpu_ldpc_occupancy(h->ldpc_s[0] ? h->ldpc_s[0] : h->ldpc, occ) == 0 && occ[4] > 32 && occ[5] > 0)
            h->acc_batches = std::max(1, std::min(ACC_BATCHES, occ[5] * (occ[4] - 32) / occ[4]));
        h->all_slots = occ[5];
        if (const char *e = std::getenv("T2GPU_RX_COLLECT")) { const int v = std::atoi(e); if (v >= 0 && v <= ACC_BATCHES) h->acc_batches = v; }   // 0: a decode per call, as round 5
    }
    return 0;
}
extern "C" int t2gpu_rx_wait(t2gpu_rx *h)
{
    if (!h) { set_error("t2gpu_rx_wait: null handle"); return -1; }
    T2_HIP(hipSetDevice(h->device));
    if (h->bundle.s[0]) { T2_HIP(hipStreamSynchronize(h->bundle.s[0])); T2_HIP(hipStreamSynchronize(h->bundle.s[1])); }
    for (hipStream_t st : h->dec_s) if (st) T2_HIP(hipStreamSynchronize(st));
    return 0;
}
extern "C" int t2gpu_rx_ldpc_occupancy(const t2gpu_rx *h, int *out6) { return h ? t2gpu_ldpc_occupancy(h->ldpc, out6) : -1; }

extern "C" int t2gpu_rx_reset(t2gpu_rx *h)
{
    if (!h) { set_error("t2gpu_rx_reset: null handle"); return -1; }
    T2_HIP(hipSetDevice(h->device));
    T2_HIP(hipDeviceSynchronize());
    if (h->ts) {
        std::unique_lock<std::mutex> lk(h->ts->m);
        h->ts->cv_done.wait(lk, [&] { return h->ts->in_flight == 0; });
        h->ts->l1_status.clear();
        // a new stream: no half packet of the old one may lead its first BBFRAME; TS bytes already de-framed stay readable
        t2gpu_bbdh_reset(h->ts->bbdh);
    }
    h->carry = 0; h->last_ready = 0; h->fec_seq = 0; h->t2_seq = 0;
    h->set = 0; h->last_set = 0;
    for (bool &v : h->llr_read_set) v = false;
    for (bool &v : h->carry_set) v = false;
    h->dec_done_set[0] = h->dec_done_set[1] = false; h->demap_pending = false;
    h->in_flight_wg[0] = h->in_flight_wg[1] = 0;
    return 0;
}

extern "C" int t2gpu_rx_execute_dev(t2gpu_rx *h, const int16_t *d_i, const int16_t *d_q, int n_frames, float level_detect, int first_call,
                                    uint8_t **d_bytes_out, int32_t **d_trials_out, void *stream)
{
    if (!h || !d_i || !d_q || n_frames < 1 || n_frames > h->cfg.max_frames) { set_error("t2gpu_rx_execute_dev: bad arguments"); return -1; }
    T2_HIP(hipSetDevice(h->device));
    ChainScope sc(h, stream);
    const int rc = rx_front(h, d_i, d_q, n_frames, level_detect, first_call, sc.s);
    if (rc != 0) return rc;
    return rx_back(h, n_frames, d_bytes_out, d_trials_out, sc.s);
}

// ---------------------------------------------------------------------------------------------------------------- host end
namespace {
// status bits of a T2 frame's L1 signalling
constexpr int L1_PRE_OK = 1, L1_POST_OK = 2, L1_MATCH = 4;

int l1_frame_status(const t2gpu_rx *h, const float *cells)
{
    const t2gpu_rx_config &c = h->cfg;
    t2gpu_l1_pre pre;
    if (t2gpu_l1_pre_parse(cells, &pre) != 1) return 0;
    int st = L1_PRE_OK;
    bool match = pre.bwt_ext == c.carrier_mode && pre.guard_interval == c.guard_interval_mode && pre.papr == c.papr_mode &&
                 pre.pilot_pattern == c.pilot_pattern && pre.num_data_symbols == c.n_data && pre.l1_post_size == c.l1_post_size;
    if (!match) return st;                                       // L1-post cannot be located with another L1_POST_SIZE
    t2gpu_l1_post post;
    t2gpu_l1_plp plp[16];
    t2gpu_l1_dyn_plp dyn[16];
    if (t2gpu_l1_post_parse(cells + 2 * L1_PRE_CELL, &pre, &post, plp, dyn, 16) != 1) return st;
    st |= L1_POST_OK;
    int first = -1;                                              // the PLP that starts the frame (time_deinterleaver.cpp:302-304)
    for (int i = 0; i < post.num_plp && i < 16; ++i) if (dyn[i].start == 0) first = i;
    match = first >= 0 && plp[first].plp_mod == c.plp_mod && plp[first].plp_cod == c.plp_cod && plp[first].plp_fec_type == c.plp_fec_type &&
            (plp[first].plp_rotation != 0) == (c.plp_rotation != 0) && plp[first].time_il_type == 0 && plp[first].time_il_length == 1 &&
            dyn[first].num_blocks == c.plp_num_blocks;
    return match ? st | L1_MATCH : st;
}

void ts_worker(TsEnd *t)
{
    t2gpu_rx *h = t->rx;
    hipSetDevice(h->device);
    const int row = h->k_bch / 8, nb = h->cfg.plp_num_blocks;
    for (;;) {
        TsJob j;
        {
            std::unique_lock<std::mutex> lk(t->m);
            t->cv_job.wait(lk, [&] { return t->stop || !t->jobs.empty(); });
            if (t->jobs.empty()) return;                         // stop, nothing queued
            j = t->jobs.front();
            t->jobs.pop_front();
        }
        TsSlot &sl = t->slot[j.slot];
        const bool ok = hipEventSynchronize(sl.ready) == hipSuccess && (!j.l1_event || hipEventSynchronize(sl.l1_ready) == hipSuccess);
        t2gpu_rx_ts_counters d{};
        const size_t per = (size_t)row + 2 * 188 + 64;           // the de-framer's out_cap contract: len_in / 8 + 376
        const size_t need = (size_t)j.fec_frames * per + per;
        TsEnd::Chunk local;                                      // BBFRAMEs are de-framed straight into the chunk the reader gets
        {
            std::lock_guard<std::mutex> lk(t->m);
            for (size_t k = 0; k < t->pool.size(); ++k)
                if (t->pool[k].cap >= need) { local = std::move(t->pool[k]); t->pool.erase(t->pool.begin() + (long)k); break; }
        }
        if (local.cap < need) { local.p.reset(new uint8_t[need]); local.cap = need; }
        size_t used = 0;
        if (ok) {
            std::vector<std::pair<long, int>> st;
            if (t->l1_check)
                for (int f = 0; f < j.t2_frames; ++f) {
                    const int v = l1_frame_status(h, sl.l1 + 2 * (size_t)f * h->p2_skip);
                    st.emplace_back(j.t2_first + f, v);
                    ++d.t2_frames;
                    if (!(v & L1_PRE_OK)) ++d.l1_pre_crc_errors;
                    else if (!(v & L1_POST_OK)) ++d.l1_post_crc_errors;
                    else if (!(v & L1_MATCH)) ++d.l1_mismatches;
                }
            else d.t2_frames = j.t2_frames;
            std::deque<std::pair<long, int>> known;
            {
                std::lock_guard<std::mutex> lk(t->m);
                for (auto &e : st) t->l1_status.push_back(e);
                const long oldest = j.fec_first / nb;            // T2 frame of the first FEC frame of this job
                while (!t->l1_status.empty() && t->l1_status.front().first < oldest) t->l1_status.pop_front();
                known = t->l1_status;
            }
            long tf_known = -1;                                  // FEC frames come in order: one look-up per T2 frame
            int v_known = 0;
            for (int i = 0; i < j.fec_frames; ++i) {
                ++d.fec_frames;
                if (sl.trials[i / h->group] < 0) { ++d.fec_frames_dropped_ldpc; continue; }        // ldpc_decoder.cpp:264-268
                if (t->l1_check) {
                    const long tf = (j.fec_first + i) / nb;
                    if (tf != tf_known) {
                        tf_known = tf; v_known = 0;
                        for (auto &e : known) if (e.first == tf) v_known = e.second;
                    }
                    if (v_known != (L1_PRE_OK | L1_POST_OK | L1_MATCH)) { ++d.fec_frames_dropped_l1; continue; }
                }
                int err = 0;
                const int n = t2gpu_bbdh_execute_packed(t->bbdh, t->need_plp, h->k_bch, sl.pack + (size_t)i * row, local.p.get() + used, (int)per, &err);
                if (n == -1) ++d.bbheader_crc_errors;
                d.ts_packet_errors += err;
                d.resync += t2gpu_bbdh_resync_count(t->bbdh);
                if (n > 0) { used += (size_t)n; d.ts_bytes += n; }
            }
        }
        local.size = ok ? used : 0;
        {
            std::lock_guard<std::mutex> lk(t->m);
            if (local.size) { t->ts_pending += local.size; t->ts.emplace_back(std::move(local)); }
            else if (local.cap) t->pool.emplace_back(std::move(local));
            t->n.t2_frames += d.t2_frames; t->n.l1_pre_crc_errors += d.l1_pre_crc_errors; t->n.l1_post_crc_errors += d.l1_post_crc_errors;
            t->n.l1_mismatches += d.l1_mismatches; t->n.fec_frames += d.fec_frames; t->n.fec_frames_dropped_ldpc += d.fec_frames_dropped_ldpc;
            t->n.fec_frames_dropped_l1 += d.fec_frames_dropped_l1; t->n.bbheader_crc_errors += d.bbheader_crc_errors;
            t->n.ts_packet_errors += d.ts_packet_errors; t->n.resync += d.resync; t->n.ts_bytes += d.ts_bytes;
            if (!ok) ++t->n.device_errors;
            sl.busy = false;
            --t->in_flight;
        }
        t->cv_done.notify_all();
    }
}

// queue the device -> host copies of one decode (packed BBFRAME rows at.., their trials, the L1 cells of the call's T2 frames) behind it
// and hand the job to the worker. Blocks only when all slots are still in use (the host end is SLOTS calls behind).
// Plain mode (call_s == nullptr): copy-engine copies on the host end's own stream behind the decode, so that they run beside the next
// call's front half. Overlap mode (call_s = the call's stream, s = the decode's): NO stream of the host end's own -- the rows and trials
// leave by a kernel on the decode's stream, the L1 cells by a kernel on the call's stream (complete there already; the next call's P2
// equaliser, which overwrites them, follows in stream order). A copy stream spends its life waiting for decodes; the runtime deals
// streams onto a handful of hardware queues, a queue is served in order, and whatever shared the copy stream's queue waited with it:
// the L1 copy of the NEXT call behind the rows of a decode still running, and the call after that behind the L1 copy
// (profiles/HISTORY.md, round 6).
int ts_submit(t2gpu_rx *h, int fec_frames, int at, int t2_frames, hipStream_t s, hipStream_t call_s, int set)
{
    TsEnd *t = h->ts;
    if (fec_frames == 0 && t2_frames == 0) return 0;
    int k;
    {
        std::unique_lock<std::mutex> lk(t->m);
        k = t->next_slot;
        t->cv_done.wait(lk, [&] { return !t->slot[k].busy; });
        t->slot[k].busy = true;
        t->next_slot = (k + 1) % t->n_slots;
        ++t->in_flight;
    }
    TsSlot &sl = t->slot[k];
    const int row = h->k_bch / 8;
    bool ok = true;
    const bool l1_now = t2_frames > 0 && t->l1_check;
    bool l1_event = false;
    if (call_s) {
        if (l1_now) {
            ok = launch_copy_bytes(h->d_l1, sl.l1, (long)t2_frames * h->p2_skip * 8, call_s) == hipSuccess &&
                 hipEventRecord(sl.l1_ready, call_s) == hipSuccess;
            l1_event = ok && call_s != s;                            // (the same stream: `ready` below covers it)
        }
        if (ok && fec_frames > 0)
            ok = launch_copy_bytes(h->d_trials + at / h->group, sl.trials, (long)((fec_frames + h->group - 1) / h->group) * 4, s) == hipSuccess &&
                 launch_copy_bytes(h->d_pack + (size_t)at * row, sl.pack, (long)fec_frames * row, s) == hipSuccess;
        ok = ok && hipEventRecord(sl.ready, fec_frames > 0 ? s : call_s) == hipSuccess;
    } else {
        hipStream_t cs = t->copy_stream;
        ok = hipEventRecord(t->decoded, s) == hipSuccess && hipStreamWaitEvent(cs, t->decoded, 0) == hipSuccess;
        if (ok && fec_frames > 0) {
            ok = hipMemcpyAsync(sl.trials, h->d_trials + at / h->group, (size_t)((fec_frames + h->group - 1) / h->group) * 4, hipMemcpyDeviceToHost, cs) == hipSuccess &&
                 hipMemcpyAsync(sl.pack, h->d_pack + (size_t)at * row, (size_t)fec_frames * row, hipMemcpyDeviceToHost, cs) == hipSuccess;
        }
        if (ok && l1_now)
            ok = hipMemcpyAsync(sl.l1, h->d_l1, (size_t)t2_frames * h->p2_skip * 8, hipMemcpyDeviceToHost, cs) == hipSuccess;
        ok = ok && hipEventRecord(sl.ready, cs) == hipSuccess;
    }
    t->last_copy = ok ? sl.ready : nullptr;
    if (set >= 0) t->last_copy_s[set] = t->last_copy;
    TsJob j;
    j.slot = k; j.fec_frames = ok ? fec_frames : 0; j.t2_frames = ok ? t2_frames : 0; j.fec_first = h->fec_seq; j.t2_first = h->t2_seq;
    j.l1_event = l1_event;
    {
        std::lock_guard<std::mutex> lk(t->m);
        t->jobs.push_back(j);
    }
    t->cv_job.notify_one();
    if (!ok) { (void)hipGetLastError(); set_error("t2gpu_rx: device -> host copy of the decoded frames failed"); return -1; }
    return 0;
}
}  // namespace

// From now on every back half / flush also copies its packed BBFRAMEs (k_bch / 8 bytes per FEC frame), the batch verdicts and the
// L1 cells of its T2 frames to pinned host memory behind the decode, and a worker thread of the handle turns them into TS bytes
// while the device runs the next calls: per T2 frame L1-pre / L1-post parse with CRC-32 (l1_check != 0: BBFRAMEs of a T2 frame whose
// CRCs fail or whose signalling differs from the handle's configuration are withheld and counted), per FEC frame the reference's
// batch drop rule and bb_de_header. need_plp: bb_de_header::set_out's PLP (bb_de_header.cpp:500-525).
extern "C" int t2gpu_rx_ts_enable(t2gpu_rx *h, int need_plp, int l1_check)
{
    if (!h) { set_error("t2gpu_rx_ts_enable: null handle"); return -1; }
    if (h->ts) return 0;                                         // already running (need_plp / l1_check of the first call stay)
    T2_HIP(hipSetDevice(h->device));
    TsEnd *t = new TsEnd();
    t->rx = h; t->need_plp = need_plp; t->l1_check = l1_check;
    t->bbdh = t2gpu_bbdh_create(need_plp);
    const size_t frames = (size_t)h->cfg.max_frames * h->cfg.plp_num_blocks + h->row_pad;
    bool ok = t->bbdh != nullptr;
    t->n_slots = h->cfg.max_frames * h->cfg.plp_num_blocks < 2 * ACC_BATCHES * h->group ? TsEnd::SLOTS : 4;
    for (int q = 0; q < t->n_slots; ++q) {
        TsSlot &sl = t->slot[q];
        ok = ok && hipHostMalloc(reinterpret_cast<void **>(&sl.pack), frames * (h->k_bch / 8), hipHostMallocDefault) == hipSuccess &&
             hipHostMalloc(reinterpret_cast<void **>(&sl.trials), (frames / h->group + 2) * 4, hipHostMallocDefault) == hipSuccess &&
             hipHostMalloc(reinterpret_cast<void **>(&sl.l1), (size_t)h->cfg.max_frames * std::max(h->p2_skip, 1) * 8, hipHostMallocDefault) == hipSuccess &&
             hipEventCreateWithFlags(&sl.ready, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&sl.l1_ready, hipEventDisableTiming) == hipSuccess;
    }
    ok = ok && hipStreamCreateWithFlags(&t->copy_stream, hipStreamNonBlocking) == hipSuccess &&
         hipEventCreateWithFlags(&t->decoded, hipEventDisableTiming) == hipSuccess;
    h->ts = t;
    if (!ok) { ts_stop(h); set_error("t2gpu_rx_ts_enable: pinned host memory allocation failed"); return -1; }
    t->worker = std::thread(ts_worker, t);
    return 0;
}

// TS bytes the worker has produced so far, oldest first; wait_all != 0 first waits until every call enqueued so far has been de-framed.
// Returns the number of bytes copied (<= cap); the rest stays queued.
extern "C" long t2gpu_rx_ts_read(t2gpu_rx *h, uint8_t *out, long cap, int wait_all)
{
    if (!h || !h->ts || (!out && cap > 0) || cap < 0) { set_error("t2gpu_rx_ts_read: bad arguments, or the TS end is not enabled"); return -1; }
    TsEnd *t = h->ts;
    std::unique_lock<std::mutex> lk(t->m);
    if (wait_all) t->cv_done.wait(lk, [&] { return t->in_flight == 0; });
    long n = 0;
    while (n < cap && !t->ts.empty()) {
        TsEnd::Chunk &c = t->ts.front();
        const size_t take = std::min<size_t>((size_t)(cap - n), c.size - t->ts_head);
        if (t->ts_head + take == c.size) {
            // the rest of the chunk: taken off the queue and copied WITHOUT the lock (a call's worth of TS is tens of megabytes; the
            // worker and the next call's ts_submit need the lock meanwhile)
            TsEnd::Chunk mine = std::move(c);
            const size_t from = t->ts_head;
            t->ts.pop_front(); t->ts_head = 0; t->ts_pending -= take;
            lk.unlock();
            std::memcpy(out + n, mine.p.get() + from, take);
            lk.lock();
            if (t->pool.size() < 4) t->pool.emplace_back(std::move(mine));
        } else {
            std::memcpy(out + n, c.p.get() + t->ts_head, take);
            t->ts_head += take; t->ts_pending -= take;
        }
        n += (long)take;
    }
    return n;
}

extern "C" int t2gpu_rx_ts_counters_get(t2gpu_rx *h, int wait_all, t2gpu_rx_ts_counters *out)
{
    if (!h || !h->ts || !out) { set_error("t2gpu_rx_ts_counters_get: bad arguments, or the TS end is not enabled"); return -1; }
    TsEnd *t = h->ts;
    std::unique_lock<std::mutex> lk(t->m);
    if (wait_all) t->cv_done.wait(lk, [&] { return t->in_flight == 0; });
    *out = t->n;
    out->ts_bytes_pending = (long)t->ts_pending;
    return 0;
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
    if (cp4) {
        T2_HIP(hipDeviceSynchronize());                     // (the call may have been made on a non-blocking stream: a blocking copy alone does not wait for it)
        T2_HIP(hipMemcpy(cp4, h->d_cp, (size_t)n_frames * h->n_sym * 16, hipMemcpyDeviceToHost));
    }
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
        T2_HIP(hipEventSynchronize(h->ev[k]));                   // both ends: a later call may have re-recorded event k already
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
    if (enable && !h->d_outer) T2_HIP(hipMalloc(&h->d_outer, ((size_t)h->cfg.max_frames * h->cfg.plp_num_blocks + h->row_pad) * sizeof(int32_t)));
    h->outer_code = enable != 0;
    return 0;
}

extern "C" int t2gpu_rx_outer_code_status(t2gpu_rx *h, int n_fec_frames, int32_t *status)
{
    if (!h || !status || n_fec_frames < 0 || n_fec_frames > h->last_ready || !h->outer_code) {
        set_error("t2gpu_rx_outer_code_status: bad arguments, or the outer code is not enabled on this handle");
        return -1;
    }
    T2_HIP(hipSetDevice(h->device));
    T2_HIP(hipDeviceSynchronize());
    if (n_fec_frames) T2_HIP(hipMemcpy(status, h->d_outer, (size_t)n_fec_frames * sizeof(int32_t), hipMemcpyDeviceToHost));
    return n_fec_frames;
}

// rows of the last back half (+ flush): packed BBFRAMEs (k_bch / 8 bytes each, MSB first) and the verdict of every SIMD batch
extern "C" int t2gpu_rx_fetch_packed(t2gpu_rx *h, int n_fec_frames, uint8_t *bytes, int32_t *trials)
{
    if (!h || n_fec_frames < 0 || n_fec_frames > h->last_ready) {
        set_error("t2gpu_rx_fetch_packed: bad arguments (more frames than the last back half and flush decoded)");
        return -1;
    }
    T2_HIP(hipSetDevice(h->device));
    T2_HIP(hipDeviceSynchronize());
    if (bytes && n_fec_frames) T2_HIP(hipMemcpy(bytes, h->d_pack, (size_t)n_fec_frames * (h->k_bch / 8), hipMemcpyDeviceToHost));
    if (trials && n_fec_frames) T2_HIP(hipMemcpy(trials, h->d_trials, (size_t)((n_fec_frames + h->group - 1) / h->group) * 4, hipMemcpyDeviceToHost));
    return t2gpu_ldpc_status(h->ldpc) == 0 ? 0 : -1;
}

// the same with one bit per byte (the reference's bit_descramble form, bch_decoder.h): the packed rows cross the bus, the host unpacks
extern "C" int t2gpu_rx_fetch(t2gpu_rx *h, int n_fec_frames, uint8_t *bits, int32_t *trials)
{
    if (!h) { set_error("t2gpu_rx_fetch: null handle"); return -1; }
    const int row = h->k_bch / 8;
    std::vector<uint8_t> packed(bits ? (size_t)std::max(n_fec_frames, 0) * row : 0);
    const int rc = t2gpu_rx_fetch_packed(h, n_fec_frames, bits ? packed.data() : nullptr, trials);
    if (rc != 0 || !bits) return rc;
    for (size_t i = 0; i < packed.size(); ++i)
        for (int n = 0; n < 8; ++n) bits[8 * i + n] = (uint8_t)((packed[i] >> (7 - n)) & 1);
    return 0;
}
