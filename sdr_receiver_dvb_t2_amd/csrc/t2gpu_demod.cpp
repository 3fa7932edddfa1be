// t2gpu_demod.cpp -- the reference's top-level slot as one C-ABI object:
//     void dvbt2_demodulator::execute(int len_in, int16_t* i_in, int16_t* q_in, signal_estimate* signal_)
//     (/root/reference/src/DVB_T2/dvbt2_demodulator.h:78, dvbt2_demodulator.cpp:145-254) with symbol_acquisition (:267-448),
//     init_dvbt2 (:129-143), reset (:111-127), set_guard_interval (:450-480) and set_guard_interval_by_brute_force (:482-548).
// Everything that touches samples runs on the GPU through the stage entry points of this library (front end, P1, guard
// correlation, FFT, equalisers); what remains here is the reference's own host logic: chunk sizing, the symbol state machine,
// acquisition from P1 / L1-pre / L1-post, the tracking-loop scalars. Buffers stay in HBM between the stages; what crosses to
// the host per symbol is what the reference's signals carry (the equalised cells of a symbol for `data` / `l1_dyn_execute`),
// two feedback floats and, for P2, the L1 cells.
#include "../../include/t2gpu.h"
#include "t2gpu_common.h"
#include "ofdm_kernels.h"
#include "front_kernels.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <vector>

using namespace t2gpu;

namespace {
enum { SYMBOL_TYPE_P1 = 0, SYMBOL_TYPE_P2, SYMBOL_TYPE_DATA, SYMBOL_TYPE_FC };   // dvbt2_demodulator.h:146-151
constexpr int P1_LEN = 2048;                                                        // dvbt2_definition.h:52
constexpr int MAX_PLP = 32;
constexpr float SAMPLE_RATE_HZ = 1.0f / (1.0e-6f * 7.0f / 64.0f);                  // SAMPLE_RATE, dvbt2_definition.h:36-38
}

// T2GPU_DEMOD_PROF=1: host wall time of execute() by part, printed when the handle is destroyed (where a closed-loop call spends its time)
enum { PF_COPY_IN = 0, PF_FRONT, PF_P1, PF_BUFFER, PF_CP, PF_FFT_EQ, PF_SV, PF_CELLS, PF_SIGNAL, PF_L1, PF_TAIL, PF_N };
static const char *const PF_NAME[PF_N] = {"copy in (H2D)", "front end (plan + launches)", "P1", "symbol buffer (D2D)", "guard corr + sync", "FFT + equaliser launch",
                                          "sync floats (D2H)", "cells (D2H)", "signals (data / l1_dyn -> de-interleaver ...)", "L1 parse", "commit iq + state"};
struct DemodProf {
    bool on = false;
    double t[PF_N] = {};
    long n[PF_N] = {};
    std::chrono::steady_clock::time_point mark;
    void start() { if (on) mark = std::chrono::steady_clock::now(); }
    void stop(int k) { if (on) { const auto now = std::chrono::steady_clock::now(); t[k] += std::chrono::duration<double>(now - mark).count(); ++n[k]; mark = now; } }
};

struct t2gpu_demod {
    DemodProf prof;
    int device = 0, id_device = 0, stride = 1;
    float sample_rate = 0.0f, level_min = 0.02f, level_max = 0.04f;                 // :31-50
    t2gpu_front *front = nullptr;
    t2gpu_p1 *p1 = nullptr;
    t2gpu_sync *sync = nullptr;
    t2gpu_ofdm *p2_ofdm = nullptr, *data_ofdm = nullptr;
    t2gpu_demod_signals sig{};
    // members of dvbt2_demodulator (.h:84-176)
    int next_symbol_type = SYMBOL_TYPE_P1;
    bool p2_init = false, demodulator_init = false, deint_start = false, crc32_l1_pre = false, frame_closing_symbol = false;
    int est_chunk = 0, symbol_size = P1_LEN, idx_buffer_sym = 0, idx_symbol = 0, end_data_symbol = 0;
    float level_detect = 3.402823466e+38f;
    // dvbt2_parameters as far as this path reads them
    int preamble = -1, fft_mode = -1, fft_size = 0, guard_interval_mode = 0, guard_interval_size = 0, carrier_mode = 1,
        pilot_pattern = 0, papr_mode = 0, n_data = 0, c_p2 = 0, c_data = 0, n_fc = 0, l_fc = 0, len_frame = 0;
    int bf_idx = 0, bf_c = 0;                                                       // the statics of the brute-force search (:484-485)
    t2gpu_l1_pre l1_pre{};
    t2gpu_l1_post l1_post{};
    t2gpu_l1_plp plp[MAX_PLP]{};
    t2gpu_l1_dyn_plp dyn[MAX_PLP]{};
    double tuner = 0.0;                                                             // rad / sample: the emulated local oscillator
    long symbols = 0, frames = 0, resets = 0;
    // device buffers
    int16_t *d_i = nullptr, *d_q = nullptr;
    size_t in_cap = 0;
    // Spectrum, cells and the cells' page-locked copy exist twice and alternate symbol by symbol (the reference's demodulators alternate
    // deinterleaved_buffer_a / _b the same way, data_symbol.cpp:140-147): a symbol's equaliser runs on eq_stream beside the NEXT chunk's
    // front end -- what the next chunk waits for is the symbol's two synchronisation floats and its guard correlation, and those come
    // from the pilots alone by a launch of their own right behind the FFT (t2gpu_sym_sync_dev).
    static constexpr int NSETS = 4;
    float *d_out = nullptr, *d_buffer_sym = nullptr, *d_spec[NSETS] = {}, *d_cells[NSETS] = {};
    int32_t *d_symidx = nullptr;
    long out_cap = 0;
    float *h_cells[NSETS] = {};               // pinned: the cells of a symbol, as the `data` / `l1_dyn_execute` signals carry them
    float *h_small = nullptr;          // pinned: guard correlation (4 floats) + the two synchronisation floats of a symbol
    int cur = 0;                       // which of the two buffer sets the symbol in hand uses
    // Nothing of this object runs on the null stream. stream: the per-symbol chain (I/Q copies, front end, P1, FFT, synchronisation
    // floats); eq_stream: a symbol's equaliser and the publishing of its cells, and -- as the home stream of the cells' twins -- what a
    // consumer (t2gpu_ti_push) goes on doing with them. Both at the highest priority: the runtime keeps the hardware queues of a priority
    // to itself, so no launch of theirs ever waits in a queue behind a decode of milliseconds (t2gpu_ldpc_submit: default priority).
    hipStream_t stream = nullptr, eq_stream = nullptr;
    hipEvent_t ev_fft = nullptr, ev_eq[NSETS] = {};
    bool eq_busy[NSETS] = {};           // ev_eq[k] has been recorded: buffer set k's last equaliser / publishing launches may still run
    // results by the device's own stores and sequence words the host reads: h_flag[0] behind the floats (seq), h_flag[16] behind the cells (seq_b)
    unsigned seq = 0, seq_b = 0;
    unsigned *h_flag = nullptr, *d_count = nullptr;
    // (the front end writes a symbol's chunk straight into the symbol buffer; P1 searches read the chunk from d_out)
    float *d_bounce = nullptr;
    // A data symbol's `data` signal is emitted once the NEXT chunk's front-end launch is on its way (or at the end of the call): what the
    // consumer does in it (the de-interleaver's push: ~9 us of host time and a launch) then runs beside the front-end kernel instead of
    // in front of it. The cells stay where they are until the next equaliser launch, which comes later still.
    // The loop on the device (include/t2gpu.h): between a frame's P2 and its last data symbol the tracking filters and the NCO's accumulators
    // live on the device, a data symbol's launches are followed by the next chunk's without waiting for its results, and the host reads
    // them one symbol behind (pend), recomputing the same floats for its own copies. Checked when the mode is left, once per frame.
    bool dev_loop = true;              // t2gpu_demod_set_device_loop
    // the chunk that completes a data symbol and the symbol's transform + floats as ONE launch (t2gpu_demod_set_chain_one): what the symbol's
    // launches need -- its buffer set (with the waits for the set's last users), its sequence word -- is then settled ahead of the chunk
    bool chain_one = true;
    bool copy_ahead = true;            // page-locked I/Q comes over chunk by chunk, a chunk's samples inside the launch of the chunk before (t2gpu_demod_set_copy_ahead)
    struct { bool valid = false, have_cp = false; int k = 0; unsigned seq_a = 0; } prep;
    bool fft_fused = false;            // the chunk just launched took the symbol's transform with it
    long fused_symbols = 0;
    bool dev_mode = false;
    struct { bool valid = false, have_cp = false, carry = false; unsigned seq_a = 0, seq_cells = 0; int k = 0, chunk = 0, next_type = 0, idx_symbol = 0; long ordinal = 0; } pend;
    long dev_symbols = 0, dev_speculated = 0, dev_waited = 0;
    // ... and in that mode a data symbol's equaliser, the publishing of its cells and its `data` signal are the business of a thread of
    // the object's own (cells_run): the caller's thread then makes three launches per symbol and reads six floats, which is what lets the
    // device's chain, not the host's runtime calls, set the pace. Jobs go through a ring; `done` counts the symbols whose signal has
    // been emitted (their buffer set is free again); the caller's thread drains the ring before it emits a signal itself (P2, FC).
    struct CellsJob { int k = 0, idx_symbol = 0, n_cells = 0; bool carry = false; };
    std::thread cells_thread;
    std::mutex cells_m;
    std::condition_variable cells_cv;
    CellsJob cells_q[4];
    unsigned cells_head = 0, cells_tail = 0;
    bool cells_stop = false, cells_drain = false;
    std::atomic<unsigned> cells_done{0};
    unsigned cells_submitted = 0;
    std::atomic<int> cells_failed{0};
    std::string cells_error;
    double cells_t[4] = {0, 0, 0, 0};     // T2GPU_DEMOD_PROF: the thread's time in launches / waiting for cells / the signal / idle
    long cells_n = 0;
    hipEvent_t ev_fft2[NSETS] = {};
    int pending_data = 0;              // cells of the symbol whose `data` signal is still to be emitted (0: none)
    int pending_buf = 0;               // ... the buffer set they are in
    unsigned pending_seq = 0;          // ... and the value h_flag[16] takes when they have arrived
    // The IQ / level estimates of a buffer (c1, c2, level_detect: once per execute(), :227-235) are committed by a launch at the end of the
    // call; nothing before the NEXT call reads them unless the caller asked for the gain decision (signal->gain_changed). So the call
    // does not wait for that launch (59 us with the device idle, 14 times per 32K frame): the next call picks the state up first thing.
    bool state_pending = false;        // a commit is on its way whose state has not been read back yet
    bool saw_results = false;          // this call has read a symbol's results: everything enqueued before them -- the I/Q copies -- is through
    // t2gpu_demod_set_trace: one record per symbol that reaches the tracking loops, written on the caller's thread when the loops have its floats
    double *trace = nullptr;
    long trace_cap = 0, trace_n = 0;
    int last_chunk = 0;                // input samples of the chunk launched last (dvbt2_demodulator.h:86 `chunk`)
    long loop_resyncs = 0;             // the host's copies of the loops set from the device's after a disagreement (never, by construction)
    bool strict_loops = false;         // T2GPU_DEMOD_STRICT_LOOPS=1 (the tests): such a disagreement is an error instead
};

// internal to the library (t2gpu_front.cpp), not part of the ABI: the host's copies of the loops take the device's values
extern "C" void t2gpu_sync_adopt(t2gpu_sync *s, float phase_est_filtered, float frequency_est_filtered, float f_int, float p_int);
extern "C" int t2gpu_front_adopt_nco(t2gpu_front *h, float phase_nco, float frequency_nco);

namespace {

constexpr int MAX_SYMBOL = 32768 + 32768 / 4 + P1_LEN;                              // max_len_symbol, :52
constexpr int SYM_BUF_CELLS = MAX_SYMBOL + 4096;                                    // the symbol buffer: a symbol and what a chunk may overshoot it by
constexpr int CHUNK_MAX = 2 * (MAX_SYMBOL + P1_LEN) + 4096;                         // input samples of one chunk, resample <= ~1

void free_all(t2gpu_demod *h)
{
    if (h->cells_thread.joinable()) {
        { std::lock_guard<std::mutex> lk(h->cells_m); h->cells_stop = true; }
        h->cells_cv.notify_one();
        h->cells_thread.join();
        t2_exit_mark("t2gpu_demod_destroy: cells' thread joined");
    }
    for (int k = 0; k < t2gpu_demod::NSETS; ++k) if (h->ev_fft2[k]) hipEventDestroy(h->ev_fft2[k]);
    if (h->front) t2gpu_front_destroy(h->front);
    if (h->p1) t2gpu_p1_destroy(h->p1);
    if (h->sync) t2gpu_sync_destroy(h->sync);
    if (h->p2_ofdm) t2gpu_ofdm_destroy(h->p2_ofdm);
    if (h->data_ofdm) t2gpu_ofdm_destroy(h->data_ofdm);
    hipFree(h->d_i); hipFree(h->d_q); hipFree(h->d_out); hipFree(h->d_buffer_sym); hipFree(h->d_symidx);
    for (int k = 0; k < t2gpu_demod::NSETS; ++k) {
        hipFree(h->d_spec[k]); hipFree(h->d_cells[k]);
        if (h->h_cells[k]) { twin_retire(h->h_cells[k]); hipHostFree(h->h_cells[k]); }
        if (h->ev_eq[k]) hipEventDestroy(h->ev_eq[k]);
    }
    if (h->ev_fft) hipEventDestroy(h->ev_fft);
    t2_exit_mark("t2gpu_demod_destroy: stage handles and buffers gone, streams next");
    if (h->eq_stream) hipStreamDestroy(h->eq_stream);
    if (h->stream) hipStreamDestroy(h->stream);
    t2_exit_mark("t2gpu_demod_destroy: streams gone");
    hipHostFree(h->h_small);
    if (h->h_flag) hipHostFree(h->h_flag);
    hipFree(h->d_count);
    hipFree(h->d_bounce);
}

// dvbt2_demodulator::reset (:111-127)
int reset(t2gpu_demod *h)
{
    if (t2gpu_front_reset_loops(h->front) != 0) return -1;
    t2gpu_sync_reset(h->sync, h->sample_rate);
    h->p2_init = false;
    h->demodulator_init = false;
    h->next_symbol_type = SYMBOL_TYPE_P1;
    h->dev_mode = false; h->pend.valid = false;
    ++h->resets;
    return 0;
}

// init_dvbt2 (:129-143): mode from P1 alone -- extended carriers assumed (dvbt2_definition.cpp:89), guard interval 1/4 "for start"
int init_dvbt2(t2gpu_demod *h)
{
    if (h->preamble != 0 && h->preamble != 3) {     // T2_SISO / T2_LITE_SISO (dvbt2_definition.h:115-121)
        set_error("t2gpu_demod: MISO preamble -- outside the supported set (README.md:17-23)");
        return -1;
    }
    int info[12];
    // P2 tables do not depend on pilot pattern, guard interval, PAPR or NUM_DATA_SYMBOLS: any valid combination serves
    const int n_any = (h->fft_mode == 5 || h->fft_mode == 7) ? 59 : 40;
    if (t2gpu_ofdm_mode_info(h->fft_mode, 1, 6, 4, 0, n_any, info) != 0) {
        set_error("t2gpu_demod: FFT size signalled by P1 is outside the supported set (16K / 32K)");
        return -1;
    }
    if (h->p2_ofdm) t2gpu_ofdm_destroy(h->p2_ofdm);
    if (!(h->p2_ofdm = t2gpu_ofdm_create(h->fft_mode, 1, 6, 4, 0, n_any, 1, h->device))) return -1;
    h->carrier_mode = 1;
    h->fft_size = info[0];
    h->c_p2 = info[5];
    h->guard_interval_size = h->fft_size / 4;
    h->symbol_size = h->fft_size + h->guard_interval_size;
    h->est_chunk = h->symbol_size;
    h->p2_init = true;
    return 0;
}

void set_guard_interval(t2gpu_demod *h)                                             // :450-480
{
    const int f = h->fft_size;
    switch (h->guard_interval_mode) {                                               // dvbt2_guardinterval_t
    case 0: h->guard_interval_size = f / 32; break;
    case 1: h->guard_interval_size = f / 16; break;
    case 2: h->guard_interval_size = f / 8; break;
    case 3: h->guard_interval_size = f / 4; break;
    case 4: h->guard_interval_size = f / 128; break;
    case 5: h->guard_interval_size = (f / 128) * 19; break;
    case 6: h->guard_interval_size = (f / 256) * 19; break;
    default: break;
    }
    h->symbol_size = f + h->guard_interval_size;
    h->est_chunk = h->symbol_size;
}

void set_guard_interval_by_brute_force(t2gpu_demod *h)                              // :482-548
{
    const int num = 5, f = h->fft_size;
    if (h->bf_c == 0) {
        t2gpu_sync_clear_frequency(h->sync);
        t2gpu_front_set_frequency_nco(h->front, 0.0f);
    }
    static const int div_mul[7][2] = {{32, 1}, {16, 1}, {8, 1}, {4, 1}, {128, 1}, {128, 19}, {256, 19}};
    h->guard_interval_size = (f / div_mul[h->bf_idx][0]) * div_mul[h->bf_idx][1];
    if (h->bf_c++ > num) {
        h->bf_idx = h->bf_idx == 6 ? 0 : h->bf_idx + 1;
        h->bf_c = 0;
    }
    h->symbol_size = f + h->guard_interval_size;
    h->est_chunk = h->symbol_size;
}

// the data / frame-closing demodulators of the mode L1-pre signals (data_demodulator->init, fc_demod->init, :396-404)
int init_data(t2gpu_demod *h)
{
    int info[12];
    if (t2gpu_ofdm_mode_info(h->fft_mode, h->carrier_mode, h->pilot_pattern, h->guard_interval_mode, h->papr_mode, h->n_data, info) != 0) {
        set_error("t2gpu_demod: the mode L1-pre signals is outside the supported set");
        return -1;
    }
    if (h->data_ofdm) t2gpu_ofdm_destroy(h->data_ofdm);
    if (!(h->data_ofdm = t2gpu_ofdm_create(h->fft_mode, h->carrier_mode, h->pilot_pattern, h->guard_interval_mode, h->papr_mode,
                                           h->n_data, 1, h->device))) return -1;
    h->c_data = info[6]; h->n_fc = info[7]; h->l_fc = info[9]; h->len_frame = info[10];
    h->end_data_symbol = h->len_frame - h->l_fc;
    h->frame_closing_symbol = h->l_fc != 0;
    return 0;
}

// A sequence word the device raises behind its own stores into page-locked memory (publish_symbol_kernel, sym_sync_kernel): the host
// reads it instead of waiting for copies and the stream (~40 us per symbol). Values only grow; false on an error.
bool wait_word(t2gpu_demod *h, volatile unsigned *flag, unsigned seq, hipStream_t stream, bool callers_thread = true)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0; (int)(*flag - seq) < 0; ++spins) {
        t2_cpu_relax();
        if ((spins & 0xfffff) == 0xfffff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
            std::fprintf(stderr, "t2gpu_demod: no results for 5 s (word %u, have %u, %s stream, symbols %ld, resets %ld, device loop %d): synchronising\n", seq, *flag,
                         stream == h->stream ? "chain" : "cells'", h->symbols, h->resets, (int)h->dev_mode);
            if (!hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize")) return false;            // a failed launch shows here
            if ((int)(*flag - seq) < 0) { set_error("t2gpu_demod: the symbol's results did not arrive"); return false; }
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (callers_thread) h->saw_results = true;
    return true;
}

// the symbol's two synchronisation floats and (cp != null) its guard correlation, stored by sym_sync_kernel; seq = the launch's word
bool sync_results(t2gpu_demod *h, unsigned seq, int k, float *cp, float *sv)
{
    if (!wait_word(h, h->h_flag, seq, h->stream)) return false;
    if (cp) std::memcpy(cp, h->h_small + 8 * k, 16);
    std::memcpy(sv, h->h_small + 8 * k + 4, 8);
    return true;
}

// n_cells cells of buffer set k to their page-locked copy by a launch on `stream`, h_flag[16] raised behind them; returns the word's value
// (0 on an error: the words start at 1)
unsigned publish_cells(t2gpu_demod *h, int k, int n_cells, hipStream_t stream)
{
    const unsigned seq = ++h->seq_b;
    if (!hip_ok(t2gpu::launch_publish_symbol(reinterpret_cast<const float2 *>(h->d_cells[k]), n_cells, nullptr, nullptr,
                                             reinterpret_cast<float2 *>(h->h_cells[k]), h->h_small, h->h_flag + 16, seq, h->d_count, stream),
                "publish_symbol_kernel")) return 0;
    return seq;
}

// the state the last commit left (level_detect for the P1 detector and the gain decision)
int finish_state(t2gpu_demod *h)
{
    if (!h->state_pending) return 0;
    float st[8];
    if (t2gpu_front_committed_state(h->front, st) != 0) return -1;
    h->level_detect = st[6];
    h->state_pending = false;
    return 0;
}

int flush_data_signal(t2gpu_demod *h)
{
    if (!h->pending_data) return 0;
    const int n = h->pending_data;
    h->pending_data = 0;
    h->prof.start();
    if (!wait_word(h, h->h_flag + 16, h->pending_seq, h->eq_stream)) return -1;
    h->prof.stop(PF_CELLS);
    h->sig.data(h->sig.user, n, h->h_cells[h->pending_buf]);
    h->prof.stop(PF_SIGNAL);
    return 0;
}

// symbol_acquisition (:267-448). Returns 0, or -1 on an error of a stage.
// src: the front end's output of this chunk -- h->d_out, or the symbol buffer itself from idx_buffer_sym on (t2gpu_demod_execute lets the
// front end write there when the chunk continues a symbol: the copy below is then no copy at all).
// D2D copy that may be asked to move cells inside one buffer: bounced through d_out's tail when the ranges overlap
int move_cells(t2gpu_demod *h, float *dst, const float *src, int n)
{
    if (n <= 0 || dst == src) return 0;
    const bool overlap = src < dst + 2 * (size_t)n && dst < src + 2 * (size_t)n;
    if (!overlap) { T2_HIP(hipMemcpyAsync(dst, src, (size_t)n * 8, hipMemcpyDeviceToDevice, h->stream)); return 0; }
    float *tmp = h->d_bounce;
    T2_HIP(hipMemcpyAsync(tmp, src, (size_t)n * 8, hipMemcpyDeviceToDevice, h->stream));
    T2_HIP(hipMemcpyAsync(dst, tmp, (size_t)n * 8, hipMemcpyDeviceToDevice, h->stream));
    return 0;
}

// ---- the cells' thread (the loop on the device): equaliser + publishing of symbol j, then the `data` signal of symbol j - 1
void cells_run(t2gpu_demod *h)
{
    hipSetDevice(h->device);
    // launched, not yet handed on: at most two -- a symbol's cells are on the host a good symbol's time after its equaliser was launched (the
    // launch waits for the FFT, which waits for the chunk's front end), so the thread hands on the symbol before last, never waiting
    t2gpu_demod::CellsJob held[2];
    unsigned held_seq[2] = {0, 0};
    int n_held = 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    auto emit_oldest = [&]() {
        const t2gpu_demod::CellsJob prev = held[0];
        const unsigned prev_seq = held_seq[0];
        held[0] = held[1]; held_seq[0] = held_seq[1]; --n_held;
        if (prev.carry) {
            const auto t0 = now();
            if (!wait_word(h, h->h_flag + 16, prev_seq, h->eq_stream, false)) { h->cells_error = last_error(); h->cells_failed.store(1); }
            else {
                const auto t1 = now();
                // the consumer's callback runs on THIS thread (include/t2gpu.h says so); what it throws -- the stage classes' fail() does --
                // is kept for the caller's thread, whose next execute() / status() / flush() returns -1 with the message
                try { h->sig.data(h->sig.user, prev.n_cells, h->h_cells[prev.k]); }
                catch (const std::exception &e) { h->cells_error = std::string("the `data` callback threw: ") + e.what(); h->cells_failed.store(1); }
                catch (...) { h->cells_error = "the `data` callback threw"; h->cells_failed.store(1); }
                if (h->prof.on) { h->cells_t[1] += secs(t0, t1); h->cells_t[2] += secs(t1, now()); }
            }
        } else if (hipStreamSynchronize(h->eq_stream) != hipSuccess) { h->cells_error = "hipStreamSynchronize (cells' stream)"; h->cells_failed.store(1); }
        h->cells_done.fetch_add(1, std::memory_order_release);
    };
    for (;;) {
        const auto t_idle = now();
        std::unique_lock<std::mutex> lk(h->cells_m);
        h->cells_cv.wait(lk, [&] { return h->cells_stop || h->cells_head != h->cells_tail || (h->cells_drain && n_held > 0); });
        if (h->cells_stop) return;
        if (h->prof.on) h->cells_t[3] += secs(t_idle, now());
        if (h->cells_head != h->cells_tail) {
            const t2gpu_demod::CellsJob job = h->cells_q[h->cells_head % 4];
            ++h->cells_head;
            lk.unlock();
            const auto t_l = now();
            unsigned seq = 0;
            // (with a consumer the equaliser's workgroups store the cells to the host themselves: no publishing launch)
            bool ok = hipStreamWaitEvent(h->eq_stream, h->ev_fft2[job.k], 0) == hipSuccess;
            if (ok && job.carry) {
                seq = ++h->seq_b;
                ok = t2gpu_eq_data_publish_dev(h->data_ofdm, h->d_spec[job.k], h->d_symidx + job.idx_symbol, h->d_cells[job.k], h->h_cells[job.k], h->h_flag + 16, seq,
                                               h->d_count, h->eq_stream) >= 0;
            } else if (ok) {
                ok = t2gpu_eq_data_execute_dev(h->data_ofdm, h->d_spec[job.k], h->d_symidx + job.idx_symbol, 1, h->d_cells[job.k], nullptr, h->eq_stream) >= 0;
            }
            if (!ok) { h->cells_error = last_error(); h->cells_failed.store(1); }
            if (h->prof.on) { h->cells_t[0] += secs(t_l, now()); ++h->cells_n; }
            if (n_held == 2) emit_oldest();
            held[n_held] = job; held_seq[n_held] = seq; ++n_held;
        } else {
            lk.unlock();
            while (n_held > 0) emit_oldest();
        }
    }
}

// the caller's thread: hands a symbol to the cells' thread / waits until the symbols up to number `upto` have been handed on / all of them
void cells_submit(t2gpu_demod *h, const t2gpu_demod::CellsJob &job)
{
    { std::lock_guard<std::mutex> lk(h->cells_m); h->cells_q[h->cells_tail % 4] = job; ++h->cells_tail; }
    ++h->cells_submitted;
    h->cells_cv.notify_one();
}
int cells_wait(t2gpu_demod *h, unsigned upto)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0; (int)(h->cells_done.load(std::memory_order_acquire) - upto) < 0; ++spins) {
        t2_cpu_relax();
        if (h->cells_failed.load()) break;
        if ((spins & 0xfffff) == 0xfffff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10)) { set_error("t2gpu_demod: the cells' thread does not answer"); return -1; }
    }
    if (h->cells_failed.load()) { set_error("t2gpu_demod (cells' thread): " + h->cells_error); return -1; }
    return 0;
}
int cells_drain(t2gpu_demod *h)
{
    if (h->cells_done.load(std::memory_order_acquire) == h->cells_submitted) return h->cells_failed.load() ? cells_wait(h, h->cells_submitted) : 0;
    { std::lock_guard<std::mutex> lk(h->cells_m); h->cells_drain = true; }
    h->cells_cv.notify_one();
    const int rc = cells_wait(h, h->cells_submitted);
    { std::lock_guard<std::mutex> lk(h->cells_m); h->cells_drain = false; }
    return rc;
}

// t2gpu_demod_set_trace: the loops as they stand behind a symbol's update (what dvbt2_demodulator holds at :444)
void trace_symbol(t2gpu_demod *h, long ordinal, int next_type, int idx_symbol, int chunk, const float *sv, bool have_cp, const float *cp)
{
    if (!h->trace || h->trace_n >= h->trace_cap) { if (h->trace) ++h->trace_n; return; }
    double g[4];
    t2gpu_sync_get(h->sync, g);
    double *r = h->trace + T2GPU_DEMOD_TRACE_W * h->trace_n++;
    r[0] = (double)ordinal; r[1] = next_type; r[2] = idx_symbol; r[3] = chunk;
    r[4] = g[0]; r[5] = g[1]; r[6] = g[2]; r[7] = g[3];
    r[8] = sv[0]; r[9] = sv[1]; r[10] = have_cp ? cp[2] : 0.0;
}

// the readout the GUI gets (:441-444)
void emit_null_indicator(t2gpu_demod *h)
{
    if (!h->sig.replace_null_indicator) return;
    double g[4];
    t2gpu_sync_get(h->sync, g);
    const float PI_X_2 = 3.14159274101257324219f * 2.0f;
    h->sig.replace_null_indicator(h->sig.user, (float)(g[2] * SAMPLE_RATE_HZ) / PI_X_2, ((float)g[1] * SAMPLE_RATE_HZ) / PI_X_2);
}

// the tracking loops with a symbol's floats (:328-330, 429-439)
void loops_after_symbol(t2gpu_demod *h, bool have_cp, const float *cp, const float *sv)
{
    if (have_cp) t2gpu_sync_frequency(h->sync, cp[2], h->fft_size);
    t2gpu_sync_symbol(h->sync, sv[0], sv[1]);
}

// the chunks the device has run on its own loop values and the host has not followed yet: with the host's values as they stand
int follow_chunks(t2gpu_demod *h)
{
    double g[4];
    t2gpu_sync_get(h->sync, g);
    const float pe = (float)g[0], fe = (float)g[1] + (float)h->tuner;
    while (t2gpu_front_loop_pending(h->front) > 0)
        if (t2gpu_front_loop_follow(h->front, pe, fe) != 0) return -1;
    return 0;
}

// the data symbol whose launches went out without waiting: its floats, the loops, the chunks launched behind it, its `data` signal in turn
int consume_pending(t2gpu_demod *h)
{
    if (!h->pend.valid) return 0;
    h->pend.valid = false;
    float cp[4] = {0.0f, 0.0f, 0.0f, 0.0f}, sv[2] = {0.0f, 0.0f};
    h->prof.start();
    if (!sync_results(h, h->pend.seq_a, h->pend.k, h->pend.have_cp ? cp : nullptr, sv)) return -1;
    h->prof.stop(PF_SV);
    loops_after_symbol(h, h->pend.have_cp, cp, sv);
    // what the device's filters made of the same floats travels with them: the host's copies must be the same numbers
    {
        double g[4];
        t2gpu_sync_get(h->sync, g);
        const float pe = (float)g[0], fe = (float)g[1] + (float)h->tuner;
        if (std::memcmp(&pe, h->h_small + 8 * h->pend.k + 6, 4) != 0 || std::memcmp(&fe, h->h_small + 8 * h->pend.k + 7, 4) != 0) {
            // bit-equal by construction (same operations, -ffp-contract=off on both sides). Should a build ever break that, reception
            // must not: T2GPU_DEMOD_STRICT_LOOPS=1 (the tests) makes it an error, otherwise this frame's remaining symbols run with the
            // loops on the host, starting from the host's values, and the event is counted (t2gpu_demod_info::loop_resyncs)
            if (h->strict_loops) { set_error("t2gpu_demod: the tracking loops on the device and on the host disagree"); return -1; }
            ++h->loop_resyncs;                                 // (leave_dev_mode adopts the device's state: the device's values are what the samples saw)
        }
    }
    trace_symbol(h, h->pend.ordinal, h->pend.next_type, h->pend.idx_symbol, h->pend.chunk, sv, h->pend.have_cp, cp);
    emit_null_indicator(h);
    return follow_chunks(h);                                   // (its cells are the cells' thread's business)
}

int enter_dev_mode(t2gpu_demod *h)
{
    if (flush_data_signal(h) != 0) return -1;                  // (a symbol read on the host hands its cells on before the cells' thread's first)
    float st[10];
    t2gpu_sync_export(h->sync, st);
    st[2] = (float)h->tuner;
    if (t2gpu_front_loop_begin(h->front, st, h->stream) != 0) return -1;
    if (!h->cells_thread.joinable()) {
        for (int k = 0; k < t2gpu_demod::NSETS; ++k) if (!h->ev_fft2[k]) T2_HIP(hipEventCreateWithFlags(&h->ev_fft2[k], hipEventDisableTiming));
        h->cells_thread = std::thread(cells_run, h);
    }
    h->dev_mode = true;
    return 0;
}

// back to the loops on the host: everything outstanding is read, and the device's state must be where the host's copies are
int leave_dev_mode(t2gpu_demod *h)
{
    if (!h->dev_mode) return 0;
    if (consume_pending(h) != 0 || cells_drain(h) != 0) { h->dev_mode = false; h->pend.valid = false; return -1; }
    h->dev_mode = false;
    float d[8], st[8] = {};
    if (t2gpu_front_loop_read(h->front, d, h->stream) != 0 || t2gpu_front_nco(h->front, st + 4) != 0) return -1;
    double g[4];
    t2gpu_sync_get(h->sync, g);
    const float pe = (float)g[0], fe = (float)g[1] + (float)h->tuner;
    if (d[7] != 0.0f) { set_error("t2gpu_demod: the NCO planner on the device ran out of room"); return -1; }
    if (std::memcmp(&d[0], &st[4], 4) != 0 || std::memcmp(&d[1], &st[5], 4) != 0 || std::memcmp(&d[2], &pe, 4) != 0 || std::memcmp(&d[3], &fe, 4) != 0) {
        if (!h->strict_loops) {
            // never, by construction; if a build breaks the construction, reception goes on from what the device did to the samples
            ++h->loop_resyncs;
            t2gpu_sync_adopt(h->sync, d[2], d[4], d[5], d[6]);
            return t2gpu_front_adopt_nco(h->front, d[0], d[1]);
        }
        char msg[320];
        std::snprintf(msg, sizeof msg, "t2gpu_demod: the device's loop state is not where the host's copy is (phase_nco %.9g / %.9g, frequency_nco %.9g / %.9g, "
                      "phase_est_filtered %.9g / %.9g, frequency input %.9g / %.9g)", d[0], st[4], d[1], st[5], d[2], pe, d[3], fe);
        set_error(msg);
        return -1;
    }
    return 0;
}

// the buffer set and the sequence word of the symbol being collected (symbol_acquisition's first steps at a complete symbol, :321-330): settled
// once per symbol, either here ahead of the chunk that will complete it (the one-launch chain) or when the symbol is complete
int prepare_symbol(t2gpu_demod *h)
{
    if (h->prep.valid) return 0;
    h->prep.have_cp = h->crc32_l1_pre;                                              // :321-330; its result is read with the symbol's other results
    // this symbol's buffer set; the launches that used it two symbols ago (equaliser, publishing: eq_stream) are through before the
    // FFT writes into it -- long since, as a rule
    const int k = h->cur = (h->cur + 1) % t2gpu_demod::NSETS;
    if (h->eq_busy[k]) { T2_HIP(hipStreamWaitEvent(h->stream, h->ev_eq[k], 0)); h->eq_busy[k] = false; }
    // (the loop on the device: set k was the symbol's four back, which the cells' thread has handed on by now -- as a rule)
    if (h->dev_mode && h->cells_submitted >= (unsigned)t2gpu_demod::NSETS && cells_wait(h, h->cells_submitted - (t2gpu_demod::NSETS - 1)) != 0) return -1;
    h->prep.k = k;
    h->prep.seq_a = ++h->seq;
    h->prep.valid = true;
    return 0;
}

int symbol_acquisition(t2gpu_demod *h, int len_in, t2gpu_signal_estimate *signal_, const float *src)
{
    int consume = 0;
    while (consume < len_in) {
        if (h->next_symbol_type != SYMBOL_TYPE_DATA && leave_dev_mode(h) != 0) return -1;   // (the loop on the device is for a frame's data symbols)
        if (h->next_symbol_type == SYMBOL_TYPE_P1) {
            if (flush_data_signal(h) != 0) return -1;                               // (the frame's last symbol completes its TI block: not held back)
            if (finish_state(h) != 0) return -1;                                    // level_detect as the execute() before left it (:227-235)
            t2gpu_p1_result r;
            h->prof.start();
            const int det = t2gpu_p1_execute_dev(h->p1, signal_->gain_changed, h->level_detect, len_in, src, &consume,
                                                 signal_->p1_reset, &r, h->stream);
            h->prof.stop(PF_P1);
            if (det < 0) return -1;
            if (det == 1) {
                const int k = r.idx_buffer_sym;                                     // p1_symbol.cpp:97: already in buffer_sym
                if (move_cells(h, h->d_buffer_sym, src + 2 * (size_t)(consume - k), k) != 0) return -1;
                h->idx_buffer_sym = k;
                if (r.fft_mode >= 0) { h->fft_mode = r.fft_mode; h->preamble = r.preamble; }
                signal_->coarse_freq_offset = r.coarse_freq_offset;
                if (h->p2_init) {
                    h->next_symbol_type = SYMBOL_TYPE_P2;
                } else if (signal_->frequency_changed) {
                    t2gpu_sync_correct_resample(h->sync, signal_->correct_resample);
                    if (std::fabs((float)signal_->coarse_freq_offset) < 10.0f) {
                        if (r.p1_decoded) {
                            if (!signal_->p1_reset) {
                                if (init_dvbt2(h) != 0) return -1;
                            } else {
                                h->p2_init = true;
                                h->demodulator_init = true;
                                signal_->p1_reset = 0;
                            }
                        }
                    } else {
                        signal_->change_frequency = 1;
                    }
                }
            }
            continue;
        }
        // ---- buffer one symbol, guard correlation, FFT (:312-341)
        const int len_in_sym = len_in - consume, len_out_sym = h->symbol_size - h->idx_buffer_sym;
        const int len_cpy_sym = len_out_sym > len_in_sym ? len_in_sym : len_out_sym;
        h->prof.start();
        if (move_cells(h, h->d_buffer_sym + 2 * (size_t)h->idx_buffer_sym, src + 2 * (size_t)consume, len_cpy_sym) != 0) return -1;
        h->prof.stop(PF_BUFFER);
        consume += len_cpy_sym;
        h->idx_buffer_sym += len_cpy_sym;
        if (h->idx_buffer_sym != h->symbol_size) {
            h->est_chunk = h->symbol_size - h->idx_buffer_sym;
            continue;
        }
        h->idx_buffer_sym = 0;
        h->prof.start();
        float cp[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (prepare_symbol(h) != 0) return -1;
        const bool have_cp = h->prep.have_cp;
        const int k = h->prep.k;
        const unsigned seq_a = h->prep.seq_a;
        h->prep.valid = false;
        // FFT (:332-334), and in its last launch the guard correlation (:321-327) and the symbol's two synchronisation floats, from the
        // pilots alone, stored to the host with the sequence word behind them -- unless the chunk's launch has run them already
        const int kind = h->next_symbol_type == SYMBOL_TYPE_DATA ? 0 : h->next_symbol_type == SYMBOL_TYPE_P2 ? 1 : 2;
        if (h->fft_fused) { h->fft_fused = false; ++h->fused_symbols; }
        else if (t2gpu_fft_sym_sync_dev(h->p2_ofdm, kind == 1 ? h->p2_ofdm : h->data_ofdm, kind, h->idx_symbol, h->d_buffer_sym, h->guard_interval_size,
                                        have_cp ? 1 : 0, h->d_spec[k], nullptr, nullptr, h->h_small + 8 * k, h->h_flag, seq_a,
                                        h->dev_mode ? t2gpu_front_loop_dev(h->front) : nullptr, h->stream) != 0) return -1;
        if (!h->dev_mode) T2_HIP(hipEventRecord(h->ev_fft, h->stream));
        h->prof.stop(PF_CP);
        h->est_chunk = 0;
        ++h->symbols;
        float sv[2] = {0.0f, 0.0f};                                                 // phase_est, sample_rate_est of this symbol
        // ---- the symbol demodulators (:343-427): the equaliser on eq_stream, behind the FFT and beside whatever the chain does next (the
        // next chunk's front end); the cells to the host by a launch behind it
        if (!h->dev_mode) T2_HIP(hipStreamWaitEvent(h->eq_stream, h->ev_fft, 0));
        if (h->next_symbol_type == SYMBOL_TYPE_DATA) {
            if (h->dev_mode) {
                // the loop on the device: this symbol's filters run behind its floats in the launch just made and nobody waits for them here. Its
                // equaliser, its cells and its `data` signal go to the cells' thread; the symbol BEFORE it is read now (its floats, the loops'
                // host copies, its chunks' NCO).
                T2_HIP(hipEventRecord(h->ev_fft2[k], h->stream));
                t2gpu_demod::CellsJob job;
                job.k = k; job.idx_symbol = h->idx_symbol; job.n_cells = h->c_data; job.carry = h->deint_start && h->sig.data;
                cells_submit(h, job);
                h->prof.stop(PF_FFT_EQ);
                if (consume_pending(h) != 0) return -1;
                h->pend.valid = true; h->pend.seq_a = seq_a; h->pend.have_cp = have_cp; h->pend.k = k;
                ++h->dev_symbols;
                ++h->idx_symbol;
                if (h->idx_symbol == h->end_data_symbol) {
                    h->next_symbol_type = h->frame_closing_symbol ? SYMBOL_TYPE_FC : SYMBOL_TYPE_P1;
                    if (!h->frame_closing_symbol) ++h->frames;
                }
                h->pend.ordinal = h->symbols; h->pend.chunk = h->last_chunk; h->pend.next_type = h->next_symbol_type; h->pend.idx_symbol = h->idx_symbol;
                continue;
            }
            if (t2gpu_eq_data_execute_dev(h->data_ofdm, h->d_spec[k], h->d_symidx + h->idx_symbol, 1, h->d_cells[k], nullptr, h->eq_stream) < 0) return -1;
            const bool carry = h->deint_start && h->sig.data;
            unsigned seq_cells = 0;
            if (carry && !(seq_cells = publish_cells(h, k, h->c_data, h->eq_stream))) return -1;
            // (with a consumer the host waits for these cells -- published behind the equaliser -- before the symbol after next is launched:
            // buffer set k is free by then without an event)
            if (!carry) { T2_HIP(hipEventRecord(h->ev_eq[k], h->eq_stream)); h->eq_busy[k] = true; }
            h->prof.stop(PF_FFT_EQ);
            ++h->idx_symbol;
            if (h->idx_symbol == h->end_data_symbol) {
                h->next_symbol_type = h->frame_closing_symbol ? SYMBOL_TYPE_FC : SYMBOL_TYPE_P1;
                if (!h->frame_closing_symbol) ++h->frames;
            }
            // the symbol BEFORE this one hands its cells on now: everything of this symbol is on its way, the consumer's work (the
            // de-interleaver's push: ~9 us of host time and a launch) runs beside it instead of in front of it
            if (flush_data_signal(h) != 0) return -1;
            h->prof.start();
            if (!sync_results(h, seq_a, k, have_cp ? cp : nullptr, sv)) return -1;
            if (have_cp) t2gpu_sync_frequency(h->sync, cp[2], h->fft_size);
            h->prof.stop(PF_SV);
            if (carry) { h->pending_data = h->c_data; h->pending_buf = k; h->pending_seq = seq_cells; }
        } else if (h->next_symbol_type == SYMBOL_TYPE_FC) {
            if (flush_data_signal(h) != 0) return -1;
            if (t2gpu_eq_fc_execute_dev(h->data_ofdm, h->d_spec[k], 1, h->d_cells[k], nullptr, h->eq_stream) < 0) return -1;
            const bool carry = h->deint_start && h->sig.data;
            const unsigned seq_cells = carry ? publish_cells(h, k, h->n_fc, h->eq_stream) : 0;
            if (carry && !seq_cells) return -1;
            T2_HIP(hipEventRecord(h->ev_eq[k], h->eq_stream));
            h->eq_busy[k] = true;
            if (!sync_results(h, seq_a, k, have_cp ? cp : nullptr, sv)) return -1;
            if (carry && !wait_word(h, h->h_flag + 16, seq_cells, h->eq_stream)) return -1;
            if (have_cp) t2gpu_sync_frequency(h->sync, cp[2], h->fft_size);
            if (carry) h->sig.data(h->sig.user, h->n_fc, h->h_cells[k]);
            h->next_symbol_type = SYMBOL_TYPE_P1;
            ++h->frames;
        } else {                                                                    // SYMBOL_TYPE_P2
            if (flush_data_signal(h) != 0) return -1;
            h->idx_symbol = 0;
            if (t2gpu_eq_p2_execute_dev(h->p2_ofdm, h->d_spec[k], 1, h->d_cells[k], nullptr, h->eq_stream) < 0) return -1;
            const unsigned seq_cells = publish_cells(h, k, h->c_p2, h->eq_stream);
            if (!seq_cells) return -1;
            T2_HIP(hipEventRecord(h->ev_eq[k], h->eq_stream));
            h->eq_busy[k] = true;
            if (!sync_results(h, seq_a, k, have_cp ? cp : nullptr, sv) || !wait_word(h, h->h_flag + 16, seq_cells, h->eq_stream)) return -1;
            const float *c = h->h_cells[k];
            if (have_cp) t2gpu_sync_frequency(h->sync, cp[2], h->fft_size);
            h->prof.start();
            // p2_symbol::execute's tail (p2_symbol.cpp:281-296): L1-pre, then L1-post
            bool crc32_l1_post = false;
            t2gpu_l1_pre pre;
            h->crc32_l1_pre = t2gpu_l1_pre_parse(c, &pre) == 1;
            if (h->crc32_l1_pre) {
                h->l1_pre = pre;
                h->carrier_mode = pre.bwt_ext;                                      // p2_symbol.cpp:493-500
                h->guard_interval_mode = pre.guard_interval;
                h->papr_mode = pre.papr;
                h->pilot_pattern = pre.pilot_pattern;
                h->n_data = pre.num_data_symbols;
                if (1840 + pre.l1_post_size <= h->c_p2)
                    crc32_l1_post = t2gpu_l1_post_parse(c + 2 * 1840, &pre, &h->l1_post, h->plp, h->dyn, MAX_PLP) == 1;
            }
            h->prof.stop(PF_L1);
            if (h->crc32_l1_pre) {
                if (h->demodulator_init) {
                    if (crc32_l1_post) {
                        if (!h->deint_start) {
                            if (h->sig.start) h->sig.start(h->sig.user, &h->l1_pre, &h->l1_post, h->plp, h->dyn);
                            h->deint_start = true;
                            if (h->sig.amount_plp) h->sig.amount_plp(h->sig.user, h->l1_post.num_plp);
                        }
                        if (h->sig.l1_dyn_execute) h->sig.l1_dyn_execute(h->sig.user, &h->l1_post, h->plp, h->dyn, h->c_p2, c);
                        h->prof.stop(PF_SIGNAL);
                    }
                    ++h->idx_symbol;
                    h->next_symbol_type = SYMBOL_TYPE_DATA;
                } else {
                    set_guard_interval(h);
                    if (init_data(h) != 0) return -1;
                    h->demodulator_init = true;
                    h->next_symbol_type = SYMBOL_TYPE_P1;
                    continue;
                }
            } else {
                if (!h->demodulator_init) {
                    set_guard_interval_by_brute_force(h);
                    h->next_symbol_type = SYMBOL_TYPE_P1;
                    continue;
                } else {
                    signal_->reset = 1;
                    signal_->p1_reset = 1;
                    return reset(h);
                }
            }
        }
        // ---- tracking loops (:429-439) and the readout the GUI gets (:441-444)
        t2gpu_sync_symbol(h->sync, sv[0], sv[1]);
        trace_symbol(h, h->symbols, h->next_symbol_type, h->idx_symbol, h->last_chunk, sv, have_cp, cp);
        emit_null_indicator(h);
        // behind a frame's P2 symbol the loops go to the device for its data symbols
        if (h->dev_loop && !h->dev_mode && h->next_symbol_type == SYMBOL_TYPE_DATA && h->crc32_l1_pre && h->demodulator_init && h->data_ofdm &&
            enter_dev_mode(h) != 0) return -1;
    }
    return 0;
}

}  // namespace

extern "C" t2gpu_demod *t2gpu_demod_create(int id_device, float sample_rate, int device)
{
    if (id_device < 0 || id_device > 2 || !(sample_rate > 0.0f)) { set_error("t2gpu_demod_create: bad arguments"); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev || hipSetDevice(device) != hipSuccess) {
        set_error("t2gpu_demod_create: no usable HIP device (this library has no CPU path)");
        return nullptr;
    }
    t2gpu_demod *h = new t2gpu_demod();
    if (const char *e = std::getenv("T2GPU_DEMOD_PROF")) h->prof.on = std::atoi(e) != 0;
    if (const char *e = std::getenv("T2GPU_DEMOD_STRICT_LOOPS")) h->strict_loops = std::atoi(e) != 0;
    h->device = device; h->id_device = id_device; h->sample_rate = sample_rate;
    h->stride = id_device == 1 ? 2 : 1;                                             // convert_input, :31-50
    h->front = t2gpu_front_create(id_device, sample_rate, CHUNK_MAX, device);
    h->out_cap = CHUNK_MAX + 4096;
    h->p1 = t2gpu_p1_create((int)h->out_cap, device);
    h->sync = t2gpu_sync_create(sample_rate);
    bool ok = h->front && h->p1 && h->sync;
    ok = ok && hipMalloc(&h->d_out, (size_t)h->out_cap * 8) == hipSuccess;
    ok = ok && hipMalloc(&h->d_buffer_sym, (size_t)SYM_BUF_CELLS * 8) == hipSuccess;
    ok = ok && hipMalloc(&h->d_bounce, (size_t)SYM_BUF_CELLS * 8) == hipSuccess;
    for (int k = 0; k < t2gpu_demod::NSETS; ++k) {
        ok = ok && hipMalloc(&h->d_spec[k], (size_t)32768 * 8) == hipSuccess && hipMalloc(&h->d_cells[k], (size_t)32768 * 8) == hipSuccess;
        // coherent (fine-grained) whatever HIP_HOST_COHERENT says: the device stores into these while its kernels run and the host reads them then
        ok = ok && hipHostMalloc(reinterpret_cast<void **>(&h->h_cells[k]), (size_t)32768 * 8, hipHostMallocCoherent) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&h->ev_eq[k], hipEventDisableTiming) == hipSuccess;
    }
    ok = ok && hipEventCreateWithFlags(&h->ev_fft, hipEventDisableTiming) == hipSuccess;
    int prio_least = 0, prio_greatest = 0;
    ok = ok && hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) == hipSuccess;
    ok = ok && hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, prio_greatest) == hipSuccess;
    ok = ok && hipStreamCreateWithPriority(&h->eq_stream, hipStreamNonBlocking, prio_greatest) == hipSuccess;
    ok = ok && hipMalloc(&h->d_symidx, 4096 * 4) == hipSuccess;
    ok = ok && hipHostMalloc(reinterpret_cast<void **>(&h->h_small), 32 * t2gpu_demod::NSETS, hipHostMallocCoherent) == hipSuccess;   // a slot of 8 floats per buffer set: a symbol's floats are read while the next symbol's launch may already store its own
    ok = ok && hipHostMalloc(reinterpret_cast<void **>(&h->h_flag), 128, hipHostMallocCoherent) == hipSuccess;
    ok = ok && hipMalloc(&h->d_count, 4) == hipSuccess && hipMemset(h->d_count, 0, 4) == hipSuccess;
    if (ok) h->h_flag[0] = h->h_flag[16] = 0;
    for (int k = 0; ok && k < t2gpu_demod::NSETS; ++k) twin_publish(h->h_cells[k], h->d_cells[k], (size_t)32768 * 8, device, false, h->eq_stream);   // what the signals hand on is still on the device
    if (ok) {
        std::vector<int32_t> idx(4096);
        for (int i = 0; i < 4096; ++i) idx[i] = i;
        ok = hipMemcpy(h->d_symidx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemset(h->d_buffer_sym, 0, (size_t)SYM_BUF_CELLS * 8) == hipSuccess;
    }
    if (!ok) {
        if (h->front && h->p1 && h->sync) set_error("t2gpu_demod_create: device allocation failed");
        free_all(h);
        delete h;
        return nullptr;
    }
    t2gpu_front_hold_iq(h->front, 1);                                               // c1 / c2 / level once per execute(), :227-235
    return h;
}

extern "C" void t2gpu_demod_destroy(t2gpu_demod *h)
{
    if (!h) return;
    hipSetDevice(h->device);
    t2_exit_mark("t2gpu_demod_destroy");
    if (h->dev_mode) (void)cells_drain(h);                     // (the last symbols' cells are handed on before the thread goes)
    t2_exit_mark("t2gpu_demod_destroy: cells drained");
    hipDeviceSynchronize();
    t2_exit_mark("t2gpu_demod_destroy: device idle");
    if (h->prof.on) {
        double tot = 0;
        for (int k = 0; k < PF_N; ++k) tot += h->prof.t[k];
        std::fprintf(stderr, "t2gpu_demod profile (host wall time inside execute(), %ld symbols):\n", h->symbols);
        if (h->cells_n) std::fprintf(stderr, "  cells' thread, per symbol: launches %.1f us, waiting for the cells %.1f us, the signal %.1f us, idle %.1f us\n", h->cells_t[0] * 1e6 / h->cells_n,
                                     h->cells_t[1] * 1e6 / h->cells_n, h->cells_t[2] * 1e6 / h->cells_n, h->cells_t[3] * 1e6 / h->cells_n);
        std::fprintf(stderr, "  loop on the device: %ld data symbols (%ld with their transform in the chunk's launch); chunks launched ahead of a symbol's results %ld, chunks that waited for them %ld\n",
                     h->dev_symbols, h->fused_symbols, h->dev_speculated, h->dev_waited);
        for (int k = 0; k < PF_N; ++k)
            std::fprintf(stderr, "  %-46s %9.3f ms  %5.1f %%  %7ld calls  %8.1f us each\n", PF_NAME[k], h->prof.t[k] * 1e3, 100.0 * h->prof.t[k] / (tot > 0 ? tot : 1),
                         h->prof.n[k], h->prof.n[k] ? h->prof.t[k] * 1e6 / h->prof.n[k] : 0.0);
    }
    free_all(h);
    delete h;
}

extern "C" int t2gpu_demod_connect(t2gpu_demod *h, const t2gpu_demod_signals *signals)
{
    if (!h) { set_error("t2gpu_demod_connect: bad arguments"); return -1; }
    h->sig = signals ? *signals : t2gpu_demod_signals{};
    return 0;
}

extern "C" int t2gpu_demod_set_tuner(t2gpu_demod *h, double offset_hz)
{
    if (!h) { set_error("t2gpu_demod_set_tuner: bad arguments"); return -1; }
    if (h->dev_mode) {                                          // (the device's copy of the loops holds the tuner it was sent up with)
        T2_HIP(hipSetDevice(h->device));
        if (leave_dev_mode(h) != 0) return -1;
    }
    h->tuner = 2.0 * 3.14159265358979323846 * offset_hz / (double)SAMPLE_RATE_HZ;
    return 0;
}

// on = 1 (default): with the loop on the device, the chunk that completes a 32K data symbol and the symbol's transform + synchronisation floats are
// ONE launch (front_fft_one_kernel); 0: two. Same values.
extern "C" int t2gpu_demod_set_chain_one(t2gpu_demod *h, int on)
{
    if (!h) { set_error("t2gpu_demod_set_chain_one: bad arguments"); return -1; }
    h->chain_one = on != 0;
    return 0;
}

// on = 1 (default): page-locked I/Q of an execute() comes over chunk by chunk -- the first chunk's samples by a launch in front of it, every
// later chunk's inside the launch of the chunk before it (extra workgroups of front_one_kernel / front_fft_one_kernel) -- instead of the whole
// buffer by one launch in front of the call's first chunk; 0: the whole buffer at once. Same samples either way.
extern "C" int t2gpu_demod_set_copy_ahead(t2gpu_demod *h, int on)
{
    if (!h) { set_error("t2gpu_demod_set_copy_ahead: bad arguments"); return -1; }
    h->copy_ahead = on != 0;
    return 0;
}

// on = 1 (default): the tracking loops on the device for a frame's data symbols; 0: on the host for every symbol (one round trip per symbol)
extern "C" int t2gpu_demod_set_device_loop(t2gpu_demod *h, int on)
{
    if (!h) { set_error("t2gpu_demod_set_device_loop: bad arguments"); return -1; }
    if (!on && h->dev_mode) {
        T2_HIP(hipSetDevice(h->device));
        if (leave_dev_mode(h) != 0) return -1;
    }
    h->dev_loop = on != 0;
    return 0;
}

extern "C" int t2gpu_demod_execute(t2gpu_demod *h, int len_in, const int16_t *i_in, const int16_t *q_in, t2gpu_signal_estimate *signal_)
{
    if (!h || len_in < 0 || !i_in || !q_in || !signal_) { set_error("t2gpu_demod_execute: bad arguments"); return -1; }
    T2_HIP(hipSetDevice(h->device));
    if (len_in == 0) return 0;
    h->saw_results = false;
    const size_t el = (size_t)len_in * h->stride;
    if (el > h->in_cap) {
        T2_HIP(hipDeviceSynchronize());
        hipFree(h->d_i); hipFree(h->d_q);
        h->d_i = h->d_q = nullptr; h->in_cap = 0;
        T2_HIP(hipMalloc(&h->d_i, el * 2));
        T2_HIP(hipMalloc(&h->d_q, el * 2));
        h->in_cap = el;
    }
    const int16_t *src_i = nullptr, *src_q = nullptr;          // page-locked I/Q coming over chunk by chunk (copy_ahead)
    size_t copied = 0;                                         // ... elements of each that are on the device or on their way
    // [copied, upto) by a launch of its own (the call's first chunk; a chunk larger than the one before it had announced)
    auto ensure_copied = [&](size_t upto) -> int {
        if (!src_i || upto <= copied) return 0;
        const size_t from = copied & ~size_t(7);              // (16-byte steps; the few elements copied twice are the same)
        launch_front_copy_in(src_i + from, src_q + from, h->d_i + from, h->d_q + from, upto - from, h->stream);
        T2_HIP(hipGetLastError());
        copied = upto;
        return 0;
    };
    h->prof.start();
    // in stream order ahead of the kernels that read them; the caller's buffers are free when this call returns (it ends with
    // t2gpu_front_state, which waits for everything launched here). From page-locked buffers (t2gpu_host_pin) the copies do not block.
    struct drain_on_error { hipStream_t s; bool armed = true; ~drain_on_error() { if (armed) hipStreamSynchronize(s); } } drain{h->stream};   // an early return leaves no copy in flight
    {
        // page-locked buffers (t2gpu_host_pin) come over by a kernel of the chain's own stream; anything else through the copy engine
        void *vi = nullptr, *vq = nullptr;
        if (hipHostGetDevicePointer(&vi, const_cast<int16_t *>(i_in), 0) == hipSuccess && hipHostGetDevicePointer(&vq, const_cast<int16_t *>(q_in), 0) == hipSuccess) {
            if (h->copy_ahead) {
                // (t2gpu_demod_set_copy_ahead) nothing comes over here: a chunk's samples are brought over by the launch of the chunk before it, the
                // call's first chunk's by a launch of their own in front of it (ensure_copied)
                src_i = static_cast<const int16_t *>(vi); src_q = static_cast<const int16_t *>(vq);
            } else {
                launch_front_copy_in(static_cast<const int16_t *>(vi), static_cast<const int16_t *>(vq), h->d_i, h->d_q, el, h->stream);
                T2_HIP(hipGetLastError());
            }
        } else {
            (void)hipGetLastError();
            T2_HIP(hipMemcpyAsync(h->d_i, i_in, el * 2, hipMemcpyHostToDevice, h->stream));
            T2_HIP(hipMemcpyAsync(h->d_q, q_in, el * 2, hipMemcpyHostToDevice, h->stream));
        }
    }
    h->prof.stop(PF_COPY_IN);
    int idx_in = 0;
    while (idx_in < len_in) {
        if (h->est_chunk == 0) {                                                    // :151-155
            if (h->next_symbol_type == SYMBOL_TYPE_P1) h->est_chunk += P1_LEN;
            h->est_chunk += h->symbol_size;
        }
        double g[4];
        t2gpu_sync_get(h->sync, g);
        double arbitrary_resample = g[3];                                           // :157-158
        auto chunk_for = [&](double r) {
            int32_t c = (int32_t)std::nearbyint(h->est_chunk * r * 2.0);            // :160
            return c > len_in - idx_in ? (int32_t)(len_in - idx_in) : c;
        };
        if (h->dev_mode && h->pend.valid) {
            // the loop on the device: the symbol whose results are still out moves the resampling value by -8e-9, 0 or +8e-9 (:430-439). When
            // all three mean the same float for the Farrow stage and the same chunk, this chunk does not wait for it.
            double c3[3];
            t2gpu_sync_candidates(h->sync, c3);
            const bool same = (float)c3[0] == (float)c3[1] && (float)c3[2] == (float)c3[1] && chunk_for(c3[0]) == chunk_for(c3[1]) && chunk_for(c3[2]) == chunk_for(c3[1]);
            if (same) ++h->dev_speculated;
            else {
                ++h->dev_waited;
                if (consume_pending(h) != 0) return -1;
                t2gpu_sync_get(h->sync, g);
                arbitrary_resample = g[3];
            }
        }
        int32_t chunk = chunk_for(arbitrary_resample);
        if (chunk > CHUNK_MAX) { set_error("t2gpu_demod_execute: chunk larger than the work buffers"); return -1; }
        h->last_chunk = chunk;
        h->prof.start();
        // a chunk that continues (or starts) an OFDM symbol is written where the symbol is collected; its few cells beyond the symbol's
        // end, if any, are moved by symbol_acquisition. P1 searches read the chunk from d_out.
        float *dst = h->d_out;
        long cap = h->out_cap;
        if (h->next_symbol_type != SYMBOL_TYPE_P1) {
            dst = h->d_buffer_sym + 2 * (size_t)h->idx_buffer_sym;
            cap = SYM_BUF_CELLS - h->idx_buffer_sym;
        }
        long n_out = -2;
        if (ensure_copied(((size_t)idx_in + (size_t)chunk) * h->stride) != 0) return -1;
        FrontCopyAhead ahead{nullptr, nullptr, nullptr, nullptr, 0};
        if (src_i && copied < el) {
            // the chunk behind this one is about as long: its samples (and a margin) come over inside this chunk's launch
            const size_t from = copied & ~size_t(7), upto = std::min(el, copied + ((size_t)chunk + 4096) * h->stride);
            ahead = FrontCopyAhead{src_i + from, src_q + from, h->d_i + from, h->d_q + from, (long)(upto - from)};
        }
        if (h->dev_mode) {
            // the chunk that completes a 32K data symbol takes the symbol's transform and floats with it (one launch; t2gpu_demod_set_chain_one)
            t2gpu::FftOneArgs fa;
            const t2gpu::FftOneArgs *fft = nullptr;
            const long need = (long)h->symbol_size - h->idx_buffer_sym;
            if (h->chain_one && h->next_symbol_type == SYMBOL_TYPE_DATA && (h->fft_size == 32768 || h->fft_size == 16384) && h->data_ofdm) {
                if (prepare_symbol(h) != 0) return -1;
                const int rc = t2gpu_fft_one_args(h->p2_ofdm, h->data_ofdm, 0, h->idx_symbol, h->d_buffer_sym, h->guard_interval_size, h->prep.have_cp ? 1 : 0,
                                                  h->d_spec[h->prep.k], h->h_small + 8 * h->prep.k, h->h_flag, h->prep.seq_a, t2gpu_front_loop_dev(h->front), &fa);
                if (rc < 0) return -1;
                if (rc == 0) fft = &fa;
            }
            int fused = 0;
            n_out = t2gpu_front_loop_fft(h->front, chunk, arbitrary_resample, h->d_i + (size_t)idx_in * h->stride, h->d_q + (size_t)idx_in * h->stride,
                                         dst, cap, h->stream, need, fft, &fused, ahead.n > 0 ? &ahead : nullptr);
            h->fft_fused = fused != 0;
            if (n_out >= 0 && ahead.n > 0) copied = std::min(el, copied + ((size_t)chunk + 4096) * h->stride);
            if (n_out == -1) return -1;
            if (n_out == -2) { if (leave_dev_mode(h) != 0) return -1; }             // (a chunk the one-launch form does not take: this symbol goes on with the loops on the host)
            else if (!h->pend.valid && follow_chunks(h) != 0) return -1;            // nothing out: the device ran on the values the host holds
        }
        if (n_out == -2) {
            t2gpu_sync_get(h->sync, g);
            const float pe = (float)g[0], fe = (float)g[1] + (float)h->tuner;
            n_out = t2gpu_front_execute_dev(h->front, 1, &chunk, &pe, &fe, &arbitrary_resample, h->d_i + (size_t)idx_in * h->stride,
                                            h->d_q + (size_t)idx_in * h->stride, dst, cap, nullptr, h->stream);
        }
        h->prof.stop(PF_FRONT);
        if (n_out < 0) return -1;
        idx_in += chunk;
        if (symbol_acquisition(h, (int)n_out, signal_, dst) != 0) return -1;
    }
    // (the loop on the device: a symbol whose results are still out stays out across the call's end -- the next call's first chunk goes ahead of
    // them as any other -- unless nothing of this call has been read yet, which is what says that its I/Q has come over)
    if (!(h->dev_mode && h->saw_results) && consume_pending(h) != 0) return -1;
    if (flush_data_signal(h) != 0) return -1;
    // ---- IQ-imbalance and level estimates of this buffer (:227-235), gain request (:236-249)
    h->prof.start();
    if (t2gpu_front_commit_iq(h->front, h->stream) != 0) return -1;
    h->state_pending = true;
    // wait for the commit only when something of this call still needs it: the gain decision below, or the caller's I/Q buffers (their
    // copies are in stream order ahead of every symbol's kernels: a call that has read a symbol's results knows they are through)
    // ... which holds for I/Q that came over at the head of the call. Page-locked I/Q that came over chunk by chunk (copy_ahead) is read by
    // the call's later launches too -- the last chunks' own copy-in, the copy-ahead workgroups inside chunk launches, whose sequence word
    // does not cover them -- and the symbol results read so far say nothing about those: the commit, launched behind all of them, does.
    if (signal_->gain_changed || !h->saw_results || src_i) {
        if (finish_state(h) != 0) return -1;
    }
    h->prof.stop(PF_TAIL);
    drain.armed = false;                                                            // the I/Q copies are through either way
    if (signal_->gain_changed) {
        if (h->level_detect < h->level_min) { signal_->gain_offset = 1; signal_->change_gain = 1; }
        else if (h->level_detect > h->level_max) { signal_->gain_offset = -1; signal_->change_gain = 1; }
        else { signal_->gain_offset = 0; signal_->change_gain = 0; }
    }
    return 0;
}

// Everything of the symbols launched so far is read and handed on: the symbol whose results are still out (the loop on the device), the
// symbols the cells' thread still holds, a `data` signal held back for the next launch. To be called before the stages behind the
// demodulator are flushed or destroyed at the end of a stream.
extern "C" int t2gpu_demod_flush(t2gpu_demod *h)
{
    if (!h) { set_error("t2gpu_demod_flush: bad arguments"); return -1; }
    T2_HIP(hipSetDevice(h->device));
    if (h->dev_mode && (consume_pending(h) != 0 || cells_drain(h) != 0)) return -1;
    if (h->cells_failed.load()) { set_error("t2gpu_demod (cells' thread): " + h->cells_error); return -1; }
    return flush_data_signal(h);
}

extern "C" int t2gpu_demod_set_trace(t2gpu_demod *h, double *records, long cap_records)
{
    if (!h || cap_records < 0 || (cap_records > 0 && !records)) { set_error("t2gpu_demod_set_trace: bad arguments"); return -1; }
    h->trace = cap_records > 0 ? records : nullptr;
    h->trace_cap = cap_records;
    h->trace_n = 0;
    return 0;
}
extern "C" long t2gpu_demod_trace_count(const t2gpu_demod *h) { return h ? h->trace_n : -1; }

extern "C" int t2gpu_demod_status(const t2gpu_demod *h, t2gpu_demod_info *out)
{
    if (!h || !out) { set_error("t2gpu_demod_status: bad arguments"); return -1; }
    // (the loop on the device: every symbol launched has been read and handed on when this returns -- the values below are then the device's)
    if (h->dev_mode && (consume_pending(const_cast<t2gpu_demod *>(h)) != 0 || cells_drain(const_cast<t2gpu_demod *>(h)) != 0)) return -1;
    if (finish_state(const_cast<t2gpu_demod *>(h)) != 0) return -1;                // level_detect of the last call's commit
    double g[4];
    t2gpu_sync_get(h->sync, g);
    out->next_symbol_type = h->next_symbol_type;
    out->p2_init = h->p2_init; out->demodulator_init = h->demodulator_init; out->deint_start = h->deint_start;
    out->crc32_l1_pre = h->crc32_l1_pre;
    out->fft_mode = h->fft_mode; out->fft_size = h->fft_size; out->guard_interval_size = h->guard_interval_size;
    out->symbol_size = h->symbol_size; out->carrier_mode = h->carrier_mode; out->pilot_pattern = h->pilot_pattern;
    out->n_data = h->n_data; out->idx_symbol = h->idx_symbol;
    out->symbols = h->symbols; out->frames = h->frames; out->resets = h->resets;
    out->level_detect = h->level_detect;
    out->phase_est_filtered = g[0]; out->frequency_est_filtered = g[1]; out->sample_rate_est_filtered = g[2];
    out->arbitrary_resample = g[3];
    out->loop_resyncs = h->loop_resyncs;
    return 0;
}
