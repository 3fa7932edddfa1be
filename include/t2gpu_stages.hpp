// t2gpu_stages.hpp -- host side above the C ABI, in the reference's own language: Qt-free C++ classes with the NAMES, slot
// signatures and drop behaviour of the reference's pipeline objects for the accelerated path, so that code written against
// src/DVB_T2/*.h reads the same here. A Qt `signal` becomes a std::function member of the same name (connect = assign); a
// slot's body is one call into libt2gpu.so (include/t2gpu.h). Nothing here computes: without the library / a GPU every
// constructor throws. Header-only; link with -lt2gpu.
//
//   reference object (file)                               here
//   ldpc_decoder       src/DVB_T2/ldpc_decoder.h:66-93    t2::ldpc_decoder
//   bch_decoder        src/DVB_T2/bch_decoder.h:26-58     t2::bch_decoder
//   bb_de_header       src/DVB_T2/bb_de_header.h:37-110   t2::bb_de_header
//   llr_demapper       src/DVB_T2/llr_demapper.h:29-48    t2::llr_demapper
//   time_deinterleaver src/DVB_T2/time_deinterleaver.h    t2::time_deinterleaver  (any number of PLPs, TI type 0)
//   filter_decimator   src/DSP/filter_decimator.h:14-131  t2::filter_decimator
//   interpolator_farrow src/DSP/interpolator_farrow.hh    t2::interpolator_farrow
//   p1_symbol          src/DVB_T2/p1_symbol.h:26-48       t2::p1_symbol
//   fast_fourier_transform + data_symbol / p2_symbol / fc_symbol (src/DSP/fast_fourier_transform.h, src/DVB_T2/data_symbol.h)
//                                                         t2::ofdm_demodulator (one handle owns the mode tables of all three)
//   dvbt2_demodulator  src/DVB_T2/dvbt2_demodulator.h:56  t2::dvbt2_demodulator   (the boundary slot execute(len, i, q, signal))
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <complex>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "t2gpu.h"

namespace t2 {

typedef std::complex<float> complex;                       // dvbt2_definition.h:23
constexpr int SIZEOF_SIMD = T2GPU_SIMD_BATCH;              // ldpc_decoder.h:28-32 (AVX2 build)

// l1_postsignalling as the stages receive it by value (dvbt2_definition.h): configurable part, PLP loop, dynamic part
struct l1_postsignalling {
    t2gpu_l1_post post{};
    std::vector<t2gpu_l1_plp> plp;
    std::vector<t2gpu_l1_dyn_plp> dyn_plp;
};

inline void fail(const char *what) { throw std::runtime_error(std::string(what) + ": " + t2gpu_last_error()); }
// T2GPU_EXIT_TRACE=1: marks on stderr inside the destructors (where a process that will not end is standing)
inline void exit_mark(const char *what)
{
    static const bool on = std::getenv("T2GPU_EXIT_TRACE") != nullptr;
    if (on) { std::fprintf(stderr, "exit trace: %s\n", what); std::fflush(stderr); }
}

// The stage classes hand each other's output buffers on untouched, as the reference's objects do through its signal / slot chain, and
// say so to the library once (t2gpu.h, host-buffer hand-over: off at the plain C ABI). A program that edits a stage's buffer between two
// stages calls t2::handoff(false) after constructing its stages.
inline void handoff(bool on) { t2gpu_handoff_enable(on ? 1 : 0); }
inline void handoff_default() { static const int once = (t2gpu_handoff_enable(1), 0); (void)once; }

// T2GPU_RX_PROF=1: host wall time inside the library calls of the stage classes, by call site (exclusive of the signals a slot emits while
// it runs: a nested scope's time is taken off its parent's). prof_report() prints the table; nothing is measured without the variable.
struct prof_table {
    enum { TI_PUSH = 0, DEMAP, DEMAP_BATCH_COPY, LDPC_SUBMIT, LDPC_WAIT, LDPC_POLL, BCH, DEHEADER, N };
    bool on = false;
    double t[N] = {};
    long n[N] = {};
    prof_table() { const char *e = std::getenv("T2GPU_RX_PROF"); on = e && std::atoi(e) != 0; }
};
inline prof_table &prof() { static prof_table p; return p; }
inline double &prof_child() { static thread_local double c = 0.0; return c; }   // time of the scopes nested in the open one, per thread
class prof_scope {
public:
    explicit prof_scope(int k) : k_(k), on_(prof().on)
    {
        if (!on_) return;
        saved_child_ = prof_child();
        prof_child() = 0.0;
        t0_ = std::chrono::steady_clock::now();
    }
    ~prof_scope()
    {
        if (!on_) return;
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count();
        prof().t[k_] += dt - prof_child();
        ++prof().n[k_];
        prof_child() = saved_child_ + dt;
    }
private:
    int k_;
    bool on_;
    double saved_child_ = 0.0;
    std::chrono::steady_clock::time_point t0_;
};
inline void prof_report(std::FILE *to)
{
    static const char *const name[prof_table::N] = {"time_deinterleaver: t2gpu_ti_push", "llr_demapper: t2gpu_demap_execute", "llr_demapper: batch copies (t2gpu_twin_copy)",
        "ldpc_decoder: t2gpu_ldpc_submit", "ldpc_decoder: t2gpu_ldpc_collect (waits)", "ldpc_decoder: t2gpu_ldpc_collect (polls)", "bch_decoder: t2gpu_bch_descramble", "bb_de_header: t2gpu_bbdh_execute"};
    if (!prof().on) return;
    for (int k = 0; k < prof_table::N; ++k)
        std::fprintf(to, "  %-46s %9.3f ms  %7ld calls  %8.1f us each\n", name[k], prof().t[k] * 1e3, prof().n[k], prof().n[k] ? prof().t[k] * 1e6 / prof().n[k] : 0.0);
}

// ---------------------------------------------------------------------------------------------------------------- LDPC
class ldpc_decoder {
public:
    // in_flight: SIMD batches the stage keeps on the device at once. The reference's stage runs on a thread of its own and its slot is
    // fed through a queued connection (dvbt2_demodulator.cpp:84-95, main_window wiring): the caller does not wait for the decode. Here
    // execute() submits the batch (t2gpu_ldpc_submit: copy-in, decode and copy-out on a stream of the handle's own) and returns;
    // bit_bch is emitted in submission order as results arrive. One batch occupies 16 of the device's 256 CUs, so a caller with the
    // reference's call shape needs several in flight to use the device. in_flight = 1 is the synchronous form of rounds 2-3.
    // own_thread = false: bit_bch is emitted on the caller's thread, on later execute() calls and in flush().
    // own_thread = true: the stage has a thread of its own as the reference's has (ldpc_decoder.cpp:129-133): it waits for the batches in
    // submission order and emits bit_bch -- and with it whatever the consumer wires behind (bch_decoder, bb_de_header, the sink) -- there,
    // beside the caller. flush() returns when everything submitted has been emitted; an exception thrown on that thread is rethrown by
    // the next execute() / flush().
    // merge (own_thread only; default on): the batches handed over in a burst -- the 6 - 7 a TI block gives rise to come within a few
    // hundred microseconds -- are decoded by ONE launch (t2gpu_ldpc_submit_add / _go), and the next launch waits for the one in flight:
    // with two or more decodes resident at once (launches of different streams) every launch of every other stream of the process -- the
    // demodulator's per-symbol chain first of all -- takes tens of microseconds longer to get its workgroups started, with one it does not
    // (DESIGN.md section 6). A batch waits at most `merge_linger_us` for the burst to end while nothing is in flight. Same bits, same
    // verdicts, bit_bch in the order of execute() as before.
    explicit ldpc_decoder(int device = 0, int in_flight = 8, bool own_thread = false, bool merge = true, int merge_linger_us = 250)
        : device_(device), depth_(in_flight < 1 ? 1 : in_flight), threaded_(own_thread && in_flight > 1), merged_(own_thread && in_flight > 1 && merge),
          linger_(merge_linger_us < 0 ? 0 : merge_linger_us)
    {
        handoff_default();
        if (merged_) groups_.resize(3);
        if (threaded_) worker_ = std::thread([this] { if (merged_) run_merged(); else run(); });
    }
    ~ldpc_decoder()
    {
        exit_mark("~ldpc_decoder");
        try { flush(); } catch (...) {}
        exit_mark("~ldpc_decoder: flushed");
        if (threaded_) {
            { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
            cv_work_.notify_all();
            worker_.join();
        }
        exit_mark("~ldpc_decoder: thread joined");
        for (auto &ring : gpu_) for (auto &code : ring) for (slot &s : code) if (s.h) t2gpu_ldpc_destroy(s.h);
        for (group &g : groups_) for (auto &ft : g.h) for (t2gpu_ldpc *h : ft) if (h) t2gpu_ldpc_destroy(h);
    }
    ldpc_decoder(const ldpc_decoder &) = delete;
    ldpc_decoder &operator=(const ldpc_decoder &) = delete;
    // signals (ldpc_decoder.h:83-87)
    std::function<void(int *idx_plp_simd, const l1_postsignalling &l1_post, int len_out, uint8_t *out)> bit_bch;
    // slot (ldpc_decoder.h:90, ldpc_decoder.cpp:157-301): 32 frames of int8 LLRs in, information bits (one per byte) out through
    // bit_bch; a batch the decoder cannot recover is reported on stderr and DROPPED, exactly as the reference does (:264-268)
    void execute(int *idx_plp_simd, const l1_postsignalling &l1_post, int len_in, int8_t *in)
    {
        const t2gpu_l1_plp &p = l1_post.plp.at((size_t)idx_plp_simd[0]);
        if (p.plp_cod < 0 || p.plp_cod > 5 || p.plp_fec_type < 0 || p.plp_fec_type > 1)    // T2-Lite codes 6, 7: not in the reference's switch (:173-246)
            fail("ldpc_decoder: PLP_COD / PLP_FEC_TYPE outside the reference's twelve codes");
        if (merged_) { execute_merged(idx_plp_simd, l1_post, p, len_in, in); return; }
        // a free handle of this code's ring; when all are busy, the oldest batch in flight is awaited (and emitted) first
        slot *s = nullptr;
        {
            std::unique_lock<std::mutex> lk(m_);
            rethrow_locked();
            std::vector<slot> &ring = gpu_[p.plp_fec_type][p.plp_cod];
            if (ring.empty()) ring.resize((size_t)depth_);
            for (;;) {
                for (slot &c : ring) if (!c.busy) { s = &c; break; }
                if (s) break;
                if (threaded_) { cv_free_.wait(lk); rethrow_locked(); }
                else { lk.unlock(); emit_front(true); lk.lock(); }
            }
            s->busy = true;                                      // reserved: nobody else touches it until it is in the queue
        }
        if (!s->h && !(s->h = t2gpu_ldpc_create(p.plp_fec_type, p.plp_cod, SIZEOF_SIMD, device_))) { release(s); fail("t2gpu_ldpc_create"); }
        t2gpu_ldpc_info(s->h, nullptr, &s->k_ldpc, nullptr, nullptr);
        {
            prof_scope ps(prof_table::LDPC_SUBMIT);
            if (t2gpu_ldpc_submit(s->h, in, len_in) != 0) { release(s); fail("t2gpu_ldpc_submit"); }
        }
        std::copy(idx_plp_simd, idx_plp_simd + SIZEOF_SIMD, s->idx);
        s->l1 = l1_post;
        { std::lock_guard<std::mutex> lk(m_); fifo_.push_back(s); }
        if (threaded_) { cv_work_.notify_one(); return; }
        if (depth_ == 1) { emit_front(true); return; }
        while (emit_front(false)) {}
    }
    // everything still inside the stage comes out (end of stream; a caller that needs the synchronous behaviour calls it after execute)
    void flush()
    {
        if (merged_) {
            std::unique_lock<std::mutex> lk(m_);
            flush_req_ = true;
            cv_work_.notify_all();
            cv_free_.wait(lk, [this] { return (gfifo_.empty() && !forming_) || error_; });
            flush_req_ = false;
            rethrow_locked();
        } else if (threaded_) {
            std::unique_lock<std::mutex> lk(m_);
            cv_free_.wait(lk, [this] { return fifo_.empty() || error_; });
            rethrow_locked();
        } else {
            while (emit_front(true)) {}
        }
    }
    int in_flight() const
    {
        std::lock_guard<std::mutex> lk(m_);
        int n = (int)fifo_.size() + (forming_ ? forming_->count : 0);
        for (const group *g : gfifo_) n += g->count;
        return n;
    }
private:
    struct slot {
        t2gpu_ldpc *h = nullptr;
        bool busy = false;
        int k_ldpc = 0;
        int idx[SIZEOF_SIMD] = {};
        l1_postsignalling l1;
    };
    void release(slot *s) { std::lock_guard<std::mutex> lk(m_); s->busy = false; cv_free_.notify_all(); }
    void rethrow_locked() { if (error_) { std::exception_ptr e = error_; error_ = nullptr; std::rethrow_exception(e); } }
    // the oldest batch in flight: waits for it (or polls); emits bit_bch or the reference's message. false: nothing there / still decoding.
    bool emit_front(bool wait)
    {
        slot *s = nullptr;
        { std::lock_guard<std::mutex> lk(m_); if (fifo_.empty()) return false; s = fifo_.front(); }
        const uint8_t *out = nullptr;
        const int *trials = nullptr;
        int rc;
        {
            prof_scope ps(wait ? prof_table::LDPC_WAIT : prof_table::LDPC_POLL);
            rc = t2gpu_ldpc_collect(s->h, wait ? 1 : 0, &out, &trials, nullptr);
        }
        if (rc == 1) return false;
        if (rc != 0) fail("t2gpu_ldpc_collect");
        // the result stays in the handle's staging until the handle is submitted to again: it is marked free AFTER the consumer returned
        // (also when the consumer throws: the batch is then lost, the stage stays usable)
        struct done_guard {
            ldpc_decoder *d; slot *s;
            ~done_guard()
            {
                { std::lock_guard<std::mutex> lk(d->m_); d->fifo_.erase(d->fifo_.begin()); s->busy = false; }
                d->cv_free_.notify_all();
            }
        } guard{this, s};
        const int trials_left = trials[0];
        if (trials_left < 0) std::fprintf(stderr, "LDPC decoder could not recover the codeword! %d\n", trials_left);
        else if (bit_bch) bit_bch(s->idx, s->l1, s->k_ldpc * SIZEOF_SIMD, const_cast<uint8_t *>(out));
        return true;
    }
    void run()                                                   // the stage's own thread
    {
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_work_.wait(lk, [this] { return stop_ || (!fifo_.empty() && !error_); });
                if (stop_) return;
            }
            try {
                emit_front(true);
            } catch (...) {
                std::lock_guard<std::mutex> lk(m_);
                error_ = std::current_exception();
                for (slot *q : fifo_) {                          // what was in flight is lost with the error; its handles are made reusable
                    const uint8_t *o = nullptr; const int *t = nullptr;
                    (void)t2gpu_ldpc_collect(q->h, 1, &o, &t, nullptr);
                    q->busy = false;
                }
                fifo_.clear();
                cv_free_.notify_all();
            }
        }
    }
    // ---- the merged form: a group = one handle of 8 SIMD batches' capacity per code, the batches added to it and their headers
    static constexpr int MERGE_MAX = 8;
    struct group {
        t2gpu_ldpc *h[2][6] = {};
        int fec_type = -1, cod = -1, k_ldpc = 0, count = 0;
        enum { FREE, FORMING, FLYING } state = FREE;
        std::chrono::steady_clock::time_point last_add;
        int idx[MERGE_MAX][SIZEOF_SIMD] = {};
        l1_postsignalling l1[MERGE_MAX];
    };
    // (m_ held) the forming group becomes the newest decode in flight
    void launch_locked(group *g)
    {
        g->state = group::FLYING;
        forming_ = nullptr;
        int rc;
        {
            prof_scope ps(prof_table::LDPC_SUBMIT);
            rc = t2gpu_ldpc_submit_go(g->h[g->fec_type][g->cod]);
        }
        if (rc != 0) { g->state = group::FREE; g->count = 0; cv_free_.notify_all(); fail("t2gpu_ldpc_submit_go"); }
        gfifo_.push_back(g);
        cv_work_.notify_all();
    }
    void execute_merged(int *idx_plp_simd, const l1_postsignalling &l1_post, const t2gpu_l1_plp &p, int len_in, int8_t *in)
    {
        std::unique_lock<std::mutex> lk(m_);
        rethrow_locked();
        group *g = forming_;
        if (g && (g->fec_type != p.plp_fec_type || g->cod != p.plp_cod)) { launch_locked(g); g = nullptr; }   // another code: what is forming goes first (order kept)
        if (!g) {
            for (;;) {
                for (group &c : groups_) if (c.state == group::FREE) { g = &c; break; }
                if (g) break;
                cv_free_.wait(lk);
                rethrow_locked();
            }
            g->fec_type = p.plp_fec_type; g->cod = p.plp_cod; g->count = 0;
            t2gpu_ldpc *&h = g->h[g->fec_type][g->cod];
            if (!h && !(h = t2gpu_ldpc_create(g->fec_type, g->cod, SIZEOF_SIMD * MERGE_MAX, device_))) fail("t2gpu_ldpc_create");
            t2gpu_ldpc_info(h, nullptr, &g->k_ldpc, nullptr, nullptr);
            g->state = group::FORMING;
            forming_ = g;
        }
        {
            prof_scope ps(prof_table::LDPC_SUBMIT);
            if (t2gpu_ldpc_submit_add(g->h[g->fec_type][g->cod], in, len_in) != 0) {
                if (g->count == 0) { g->state = group::FREE; forming_ = nullptr; }
                fail("t2gpu_ldpc_submit_add");
            }
        }
        std::copy(idx_plp_simd, idx_plp_simd + SIZEOF_SIMD, g->idx[g->count]);
        g->l1[g->count] = l1_post;
        ++g->count;
        g->last_add = std::chrono::steady_clock::now();
        if (g->count == MERGE_MAX) launch_locked(g);
        cv_work_.notify_all();
    }
    // the oldest decode in flight: waits for it, launches what has formed meanwhile (the device is free again), emits its batches in order
    void emit_group_front()
    {
        group *g = nullptr;
        { std::lock_guard<std::mutex> lk(m_); if (gfifo_.empty()) return; g = gfifo_.front(); }
        const uint8_t *out = nullptr;
        const int *trials = nullptr;
        int rc;
        {
            prof_scope ps(prof_table::LDPC_WAIT);
            rc = t2gpu_ldpc_collect(g->h[g->fec_type][g->cod], 1, &out, &trials, nullptr);
        }
        if (rc != 0) fail("t2gpu_ldpc_collect");
        struct done_guard {
            ldpc_decoder *d; group *g;
            ~done_guard()
            {
                { std::lock_guard<std::mutex> lk(d->m_); d->gfifo_.erase(d->gfifo_.begin()); g->state = group::FREE; g->count = 0; }
                d->cv_free_.notify_all();
            }
        } guard{this, g};
        {
            std::lock_guard<std::mutex> lk(m_);
            if (gfifo_.size() == 1 && forming_ && (flush_req_ || std::chrono::steady_clock::now() - forming_->last_add >= linger_)) launch_locked(forming_);
        }
        for (int b = 0; b < g->count; ++b) {
            if (trials[b] < 0) std::fprintf(stderr, "LDPC decoder could not recover the codeword! %d\n", trials[b]);
            else if (bit_bch) bit_bch(g->idx[b], g->l1[b], g->k_ldpc * SIZEOF_SIMD, const_cast<uint8_t *>(out) + (size_t)b * g->k_ldpc * SIZEOF_SIMD);
        }
    }
    void run_merged()                                            // the stage's own thread, merged form
    {
        for (;;) {
            bool hand_on = false;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_work_.wait(lk, [this] { return stop_ || ((!gfifo_.empty() || forming_) && !error_); });
                if (stop_) return;
                if (!gfifo_.empty()) hand_on = true;
                else {
                    // nothing in flight and a group forming: launched when the burst is over (no addition for linger_), or at once in a flush
                    const auto ripe_at = forming_->last_add + linger_;
                    if (flush_req_ || std::chrono::steady_clock::now() >= ripe_at) {
                        try { launch_locked(forming_); } catch (...) { error_ = std::current_exception(); cv_free_.notify_all(); }
                    } else cv_work_.wait_until(lk, ripe_at);
                    continue;
                }
            }
            if (!hand_on) continue;
            try {
                emit_group_front();
            } catch (...) {
                std::lock_guard<std::mutex> lk(m_);
                error_ = std::current_exception();
                for (group *q : gfifo_) {                        // what was in flight is lost with the error; its handles are made reusable
                    const uint8_t *o = nullptr; const int *t = nullptr;
                    (void)t2gpu_ldpc_collect(q->h[q->fec_type][q->cod], 1, &o, &t, nullptr);
                    q->state = group::FREE; q->count = 0;
                }
                gfifo_.clear();
                cv_free_.notify_all();
            }
        }
    }
    int device_, depth_;
    bool threaded_, merged_ = false, flush_req_ = false, stop_ = false;
    std::chrono::microseconds linger_{250};
    std::vector<group> groups_;
    group *forming_ = nullptr;
    std::vector<group *> gfifo_;
    mutable std::mutex m_;
    std::condition_variable cv_work_, cv_free_;
    std::exception_ptr error_;
    std::thread worker_;
    std::vector<slot> gpu_[2][6];
    std::vector<slot *> fifo_;
};

// ---------------------------------------------------------------------------------------------------------------- BCH stub
class bch_decoder {
public:
    std::function<void(int plp_id, const l1_postsignalling &l1_post, int len_out, uint8_t *out)> bit_descramble;   // bch_decoder.h:35
    // slot (bch_decoder.h:41, bch_decoder.cpp:63-164): strips the BCH parity (the reference corrects nothing) and descrambles;
    // one bit_descramble per FEC frame
    // not in the reference (its decoder is a TODO, bch_decoder.cpp:136): when set, execute applies the outer code to `in` first and
    // outer_code_status holds, per FEC frame of the last call, the bits corrected or -1 (more than t errors, frame untouched)
    bool outer_code = false;
    std::vector<int32_t> outer_code_status;
    bch_decoder() { handoff_default(); }
    void execute(int *idx_plp_simd, const l1_postsignalling &l1_post, int len_in, uint8_t *in)
    {
        const t2gpu_l1_plp &p = l1_post.plp.at((size_t)idx_plp_simd[0]);
        const int k_ldpc = ldpc_k(p.plp_fec_type, p.plp_cod);
        if (k_ldpc < 0) fail("bch_decoder: PLP_COD outside the reference's twelve codes");
        const int frames = len_in / k_ldpc;
        out_.resize((size_t)len_in);
        if (outer_code) {
            outer_code_status.assign((size_t)frames, 0);
            if (t2gpu_bch_decode(p.plp_fec_type, p.plp_cod, in, frames, outer_code_status.data()) != frames) fail("t2gpu_bch_decode");
        }
        int k_bch;
        {
            prof_scope ps(prof_table::BCH);
            k_bch = t2gpu_bch_descramble(p.plp_fec_type, p.plp_cod, in, frames, out_.data());
        }
        if (k_bch < 0) fail("t2gpu_bch_descramble");
        for (int n = 0; n < frames; ++n)
            if (bit_descramble) bit_descramble(idx_plp_simd[n], l1_post, k_bch, out_.data() + (size_t)n * k_bch);
    }
private:
    static int ldpc_k(int fec_type, int cod)
    {
        static const int kn[6] = {32400, 38880, 43200, 48600, 51840, 54000}, ks[6] = {7200, 9720, 10800, 11880, 12600, 13320};
        if (cod < 0 || cod > 5) return -1;
        return fec_type ? kn[cod] : ks[cod];                          // k_ldpc, ldpc_decoder.cpp:177-246
    }
    std::vector<uint8_t> out_;
};

// ---------------------------------------------------------------------------------------------------------------- BB de-header
class bb_de_header {
public:
    explicit bb_de_header(int need_plp = 0) : h_(t2gpu_bbdh_create(need_plp)) { if (!h_) fail("t2gpu_bbdh_create"); }
    ~bb_de_header() { t2gpu_bbdh_destroy(h_); }
    std::function<void(const std::string &info)> ts_stage;                       // bb_de_header.h:56
    std::function<void(const uint8_t *ts, int len)> write_out;                   // the UDP datagram / file write of :433-443
    // slot (bb_de_header.h:59, bb_de_header.cpp:84-448)
    void execute(int plp_id, const l1_postsignalling &, int len_in, uint8_t *in)
    {
        buf_.resize((size_t)len_in / 8 + 400);
        int errors = 0;
        int n;
        {
            prof_scope ps(prof_table::DEHEADER);
            n = t2gpu_bbdh_execute(h_, plp_id, len_in, in, buf_.data(), (int)buf_.size(), &errors);
        }
        if (n > 0 && write_out) write_out(buf_.data(), n);
        else if (n == -1 && ts_stage) ts_stage("Baseband header CRC8 error.");
    }
private:
    t2gpu_bbdh *h_;
    std::vector<uint8_t> buf_;
};

// ---------------------------------------------------------------------------------------------------------------- LLR demapper
class llr_demapper {
public:
    explicit llr_demapper(int device = 0) : device_(device) { handoff_default(); }
    ~llr_demapper() { release(); }
    std::function<void(float snr)> signal_noise_ratio;                                                             // llr_demapper.h:38
    std::function<void(int *idx_plp_simd, const l1_postsignalling &, int len_out, int8_t *out)> soft_multiplexer_de_twist;   // :39
    // NOT in the reference (t2gpu_demap_configure): clamp the LLRs to int8 instead of the reference's wrapping cast (llr_demapper.cpp:722-737).
    // With the cast, 256-QAM loses every SIMD batch in AWGN (the outer points wrap at any SNR); off unless the caller sets it.
    bool saturate_llr = false;
    // slot (llr_demapper.h:44-45, llr_demapper.cpp:132-158): one TI block of cells in; LLR frames are collected into batches of
    // SIZEOF_SIMD and handed on whenever one is full (:742-764), the remainder waits for the next TI block.
    // The two batch buffers (the reference's A / B) are page-locked and have a twin on the device that is filled as they are
    // (t2gpu_twin_copy): ldpc_decoder, handed such a buffer, finds the batch there (t2gpu.h, host-buffer hand-over).
    void execute(int ti_block_size, complex *time_deint_cell, int plp_id, const l1_postsignalling &l1_post)
    {
        const t2gpu_l1_plp &p = l1_post.plp.at((size_t)plp_id);
        const int fec_size = p.plp_fec_type ? 64800 : 16200;
        const int cpf = fec_size / (2 * (p.plp_mod + 1));
        if (!h_ || key_ != key(p) || ti_block_size > cells_max_) {
            if (h_) { t2gpu_demap_destroy(h_); h_ = nullptr; }
            cells_max_ = std::max((p.plp_num_blocks_max > 0 ? p.plp_num_blocks_max : 1) * cpf, ti_block_size);
            if (!(h_ = t2gpu_demap_create(p.plp_mod, p.plp_fec_type, p.plp_cod, p.plp_rotation, cells_max_, device_))) fail("t2gpu_demap_create");
            key_ = key(p);
            saturated_ = false;
        }
        if (saturated_ != saturate_llr) {
            if (t2gpu_demap_configure(h_, saturate_llr ? 1 : 0) != 0) fail("t2gpu_demap_configure");
            saturated_ = saturate_llr;
        }
        // buffers grow, never shrink, and a batch in the making survives a change of PLP (frames of several PLPs share a SIMD batch,
        // llr_demapper.cpp:742-764: idx_plp_simd says whose each one is)
        const size_t need_frames = (size_t)(cells_max_ / cpf + 1) * fec_size, need_batch = (size_t)fec_size * SIZEOF_SIMD;
        if (frames_.size() < need_frames) {
            if (!frames_.empty()) t2gpu_host_unpin(frames_.data());
            frames_.assign(need_frames, 0);
            t2gpu_host_pin(frames_.data(), frames_.size());
        }
        for (std::vector<int8_t> *b : {&buffer_a, &buffer_b}) {
            if (b->size() >= need_batch) continue;
            const std::vector<int8_t> keep(*b);
            if (!b->empty()) t2gpu_twin_detach(b->data());
            b->assign(need_batch, 0);
            if (t2gpu_twin_attach(b->data(), b->size(), device_) != 0) fail("t2gpu_twin_attach");
            if (!keep.empty() && t2gpu_twin_copy(b->data(), keep.data(), keep.size(), device_) != 0) fail("t2gpu_twin_copy");
        }
        float sums[3] = {0, 0, 0};
        int frames;
        {
            prof_scope ps(prof_table::DEMAP);
            frames = t2gpu_demap_execute(h_, reinterpret_cast<const float *>(time_deint_cell), ti_block_size, frames_.data(), sums);
        }
        if (frames < 0) fail("t2gpu_demap_execute");
        if (signal_noise_ratio) signal_noise_ratio(20.0f * std::log10(sums[0] / sums[1]));                       // :659
        for (int f = 0; f < frames;) {
            std::vector<int8_t> &out = swap_buffer ? buffer_a : buffer_b;
            const int run = std::min(frames - f, SIZEOF_SIMD - blocks);      // consecutive frames go to consecutive places of one batch: one copy
            {
                prof_scope ps(prof_table::DEMAP_BATCH_COPY);
                if (t2gpu_twin_copy(out.data() + (size_t)blocks * fec_size, frames_.data() + (size_t)f * fec_size, (size_t)run * fec_size, device_) != 0)
                    fail("t2gpu_twin_copy");
            }
            for (int k = 0; k < run; ++k) idx_plp_simd[blocks + k] = plp_id;
            blocks += run; f += run;
            if (blocks == SIZEOF_SIMD) {
                blocks = 0;
                swap_buffer = !swap_buffer;
                if (soft_multiplexer_de_twist) soft_multiplexer_de_twist(idx_plp_simd, l1_post, fec_size * SIZEOF_SIMD, out.data());
            }
        }
    }
private:
    static int key(const t2gpu_l1_plp &p) { return p.plp_mod | (p.plp_fec_type << 4) | (p.plp_cod << 8) | (p.plp_rotation << 12) | (p.plp_num_blocks_max << 13); }
    void release()
    {
        if (!buffer_a.empty()) t2gpu_twin_detach(buffer_a.data());
        if (!buffer_b.empty()) t2gpu_twin_detach(buffer_b.data());
        if (!frames_.empty()) t2gpu_host_unpin(frames_.data());
        if (h_) { t2gpu_demap_destroy(h_); h_ = nullptr; }
    }
    int device_, key_ = -1, blocks = 0, cells_max_ = 0;
    t2gpu_demap *h_ = nullptr;
    int idx_plp_simd[SIZEOF_SIMD] = {};
    std::vector<int8_t> frames_, buffer_a, buffer_b;
    bool swap_buffer = true, saturated_ = false;
};

// ---------------------------------------------------------------------------------------------------------------- time de-interleaver
class time_deinterleaver {
public:
    // own_thread = true: ti_block -- and with it the demapper and whatever is wired behind -- is emitted on a thread of the stage's own
    // (the reference's stages each live on a QThread, time_deinterleaver.cpp:30-36), the caller goes on with the next symbols. The
    // blocks alternate between two buffers as the reference's do (buffer_a / buffer_b, :313-353); a block is handed over when the
    // thread is through with the one before, so the buffer being filled is never the one being read. flush() waits for that thread.
    explicit time_deinterleaver(int device = 0, bool own_thread = false) : device_(device), threaded_(own_thread)
    {
        handoff_default();
        if (threaded_) worker_ = std::thread([this] { run(); });
    }
    ~time_deinterleaver()
    {
        exit_mark("~time_deinterleaver");
        if (threaded_) {
            { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
            cv_.notify_all();
            worker_.join();
        }
        exit_mark("~time_deinterleaver: thread joined");
        release();
        exit_mark("~time_deinterleaver: released");
    }
    time_deinterleaver(const time_deinterleaver &) = delete;
    time_deinterleaver &operator=(const time_deinterleaver &) = delete;
    std::function<void(int ti_block_size, complex *time_deint_cell, int plp_id, const l1_postsignalling &)> ti_block;   // :38
    // time_deinterleaver.h:33, cpp:38-145: one de-interleaver per PLP (the reference keeps one permutation per PLP)
    void start(const t2gpu_l1_pre &l1_pre, const l1_postsignalling &l1_post)
    {
        flush();
        release();
        for (const t2gpu_l1_plp &p : l1_post.plp) {
            lane l;
            for (int k = 0; k < 2; ++k) {
                if (!(l.h[k] = t2gpu_ti_create(p.plp_mod, p.plp_fec_type, p.plp_num_blocks_max, device_))) fail("t2gpu_ti_create");
                l.out[k].assign((size_t)p.plp_num_blocks_max * t2gpu_ti_cells_per_fec(l.h[k]), complex());
                t2gpu_host_pin(l.out[k].data(), l.out[k].size() * sizeof(complex));      // the TI block comes down at the link's rate
            }
            lanes_.push_back(std::move(l));
        }
        p2_start_idx_cell = 1840 + l1_pre.l1_post_size;
        plp_state_ = 0;
    }
    // :43, cpp:268-288. The PLP / TI-block sequence of the frame follows from the dynamic signalling alone
    // (t2gpu_ti_frame_plan); cells are then routed to the PLP's de-interleaver as they arrive.
    void l1_dyn_execute(const l1_postsignalling &l1_post, int len_in, complex *ofdm_cell)
    {
        l1_post_ = l1_post;
        plan_.resize(4096);
        const int n = t2gpu_ti_frame_plan((int)l1_post.plp.size(), l1_post.plp.data(), l1_post.dyn_plp.data(), 1 << 22, &plp_state_,
                                          plan_.data(), (int)plan_.size());
        if (n < 0) fail("t2gpu_ti_frame_plan");
        plan_.resize(n);
        k_ = 0; pos_ = 0;
        push(len_in - p2_start_idx_cell, ofdm_cell + p2_start_idx_cell);
    }
    void execute(int len_in, complex *ofdm_cell) { push(len_in, ofdm_cell); }                  // :45, cpp:290-376
    // the stage's thread has emitted everything handed to it (a no-op without own_thread); rethrows what that thread threw
    void flush()
    {
        if (!threaded_) return;
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [this] { return !job_.pending; });
        rethrow_locked();
    }
private:
    struct lane {                                                // one PLP: two de-interleavers, two output buffers, used in turn
        t2gpu_ti *h[2] = {nullptr, nullptr};
        std::vector<complex> out[2];
        int cur = 0;
    };
    struct job { bool pending = false; int size = 0, plp = 0; complex *cells = nullptr; t2gpu_ti *h = nullptr; l1_postsignalling l1; };
    void push(int n, complex *cells)
    {
        while (n > 0 && k_ < plan_.size()) {
            const t2gpu_ti_block &b = plan_[k_];
            lane &l = lanes_.at((size_t)b.plp);
            t2gpu_ti *h = l.h[l.cur];
            if (pos_ == b.offset && t2gpu_ti_begin(h, b.num_blocks) != 0) fail("t2gpu_ti_begin");
            const int take = std::min(n, b.offset + b.size - pos_);
            int done;
            {
                prof_scope ps(prof_table::TI_PUSH);
                // with a thread of its own the complete block's copy down is waited for there (t2gpu_ti_wait), not here
                done = threaded_ ? t2gpu_ti_push_async(h, reinterpret_cast<const float *>(cells), take, reinterpret_cast<float *>(l.out[l.cur].data()))
                                 : t2gpu_ti_push(h, reinterpret_cast<const float *>(cells), take, reinterpret_cast<float *>(l.out[l.cur].data()));
            }
            if (done < 0) fail("t2gpu_ti_push");
            cells += take; n -= take; pos_ += take;
            if (done == 1) {
                hand_over(b.size, l.out[l.cur].data(), b.plp, h);
                l.cur ^= 1;
                ++k_;
            }
        }
    }
    void hand_over(int size, complex *cells, int plp, t2gpu_ti *h)
    {
        if (!threaded_) { if (ti_block) ti_block(size, cells, plp, l1_post_); return; }
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [this] { return !job_.pending; });          // the block before this one has been dealt with: its buffer is free again
        rethrow_locked();
        job_.pending = true; job_.size = size; job_.plp = plp; job_.cells = cells; job_.h = h; job_.l1 = l1_post_;
        cv_.notify_all();
    }
    void run()
    {
        for (;;) {
            std::unique_lock<std::mutex> lk(m_);
            cv_.wait(lk, [this] { return stop_ || job_.pending; });
            if (stop_) return;
            lk.unlock();
            try {
                if (t2gpu_ti_wait(job_.h) != 0) fail("t2gpu_ti_wait");                 // the block has come down
                if (ti_block) ti_block(job_.size, job_.cells, job_.plp, job_.l1);
            } catch (...) {
                std::lock_guard<std::mutex> g(m_);
                error_ = std::current_exception();
            }
            lk.lock();
            job_.pending = false;
            cv_.notify_all();
        }
    }
    void rethrow_locked() { if (error_) { std::exception_ptr e = error_; error_ = nullptr; std::rethrow_exception(e); } }
    void release()
    {
        for (lane &l : lanes_)
            for (int k = 0; k < 2; ++k) {
                if (l.h[k]) t2gpu_ti_destroy(l.h[k]);
                if (!l.out[k].empty()) t2gpu_host_unpin(l.out[k].data());
            }
        lanes_.clear();
    }
    int device_, p2_start_idx_cell = 0, plp_state_ = 0, pos_ = 0;
    bool threaded_, stop_ = false;
    size_t k_ = 0;
    std::vector<lane> lanes_;
    std::vector<t2gpu_ti_block> plan_;
    l1_postsignalling l1_post_;
    std::mutex m_;
    std::condition_variable cv_;
    std::exception_ptr error_;
    job job_;
    std::thread worker_;
};

// ---------------------------------------------------------------------------------------------------------------- DSP front end
// filter_decimator / interpolator_farrow keep their state in a front-end handle each (they are members of dvbt2_demodulator in
// the reference, dvbt2_demodulator.h:129-130)
class filter_decimator {
public:
    explicit filter_decimator(int max_len = 1 << 20, int device = 0) : h_(t2gpu_front_create(0, 64.0e6f / 7.0f, max_len, device)) { if (!h_) fail("t2gpu_front_create"); }
    ~filter_decimator() { t2gpu_front_destroy(h_); }
    void execute(int len_in, complex *in, int &len_out, complex *out)                      // filter_decimator.h:72
    {
        len_out = t2gpu_decim_execute(h_, len_in, reinterpret_cast<const float *>(in), reinterpret_cast<float *>(out));
        if (len_out < 0) fail("t2gpu_decim_execute");
    }
private:
    t2gpu_front *h_;
};

class interpolator_farrow {
public:
    explicit interpolator_farrow(int max_len = 1 << 20, int device = 0) : cap_(max_len), h_(t2gpu_front_create(0, 64.0e6f / 7.0f, max_len, device)) { if (!h_) fail("t2gpu_front_create"); }
    ~interpolator_farrow() { t2gpu_front_destroy(h_); }
    void operator()(int len_in, complex *in, double &arbitrary_resample, int &len_out, complex *out)   // interpolator_farrow.hh:41
    {
        len_out = t2gpu_farrow_execute(h_, len_in, reinterpret_cast<const float *>(in), arbitrary_resample, reinterpret_cast<float *>(out), 4 * cap_ + 64);
        if (len_out < 0) fail("t2gpu_farrow_execute");
    }
private:
    int cap_;
    t2gpu_front *h_;
};

// p1_symbol (p1_symbol.h:33-37): the by-reference outputs of the reference signature, dvbt2_parameters reduced to the two
// fields P1 sets
class p1_symbol {
public:
    explicit p1_symbol(int max_len = 1 << 20, int device = 0) : h_(t2gpu_p1_create(max_len, device)) { if (!h_) fail("t2gpu_p1_create"); }
    ~p1_symbol() { t2gpu_p1_destroy(h_); }
    bool execute(bool gain_changed, float level_detect, const int len_in, complex *in, int &consume, complex *buffer_sym,
                 int &idx_buffer_sym, int &preamble, int &fft_mode, double &coarse_freq_offset, bool &p1_decoded, bool &reset)
    {
        t2gpu_p1_result r;
        const int rc = t2gpu_p1_execute(h_, gain_changed, level_detect, len_in, reinterpret_cast<const float *>(in), &consume, reset, &r);
        if (rc < 0) fail("t2gpu_p1_execute");
        if (rc == 0) return false;
        idx_buffer_sym = r.idx_buffer_sym;
        for (int i = 0; i < r.idx_buffer_sym; ++i) buffer_sym[i] = in[consume - r.idx_buffer_sym + i];   // p1_symbol.cpp:97
        if (r.shift >= 0) { preamble = r.preamble; fft_mode = r.fft_mode; }
        coarse_freq_offset = r.coarse_freq_offset;
        p1_decoded = r.p1_decoded != 0;
        return true;
    }
private:
    t2gpu_p1 *h_;
};

// fast_fourier_transform::execute + data_symbol::execute for one symbol at a time (the reference call shape; whole frames go
// through the *_dev entry points instead)
class ofdm_demodulator {
public:
    ofdm_demodulator(int fft_mode, int carrier_mode, int pilot_pattern, int guard_interval_mode, int papr_mode, int n_data, int device = 0)
        : h_(t2gpu_ofdm_create(fft_mode, carrier_mode, pilot_pattern, guard_interval_mode, papr_mode, n_data, 1, device))
    {
        if (!h_) fail("t2gpu_ofdm_create");
        int info[12];
        t2gpu_ofdm_mode_info(fft_mode, carrier_mode, pilot_pattern, guard_interval_mode, papr_mode, n_data, info);
        fft_size = info[0]; c_data = info[6];
        spectrum_.resize((size_t)fft_size); cells_.resize((size_t)fft_size);
    }
    ~ofdm_demodulator() { t2gpu_ofdm_destroy(h_); }
    complex *fft_execute(complex *in)                                                          // fast_fourier_transform.h:62-70
    {
        if (t2gpu_fft_execute(h_, reinterpret_cast<const float *>(in), reinterpret_cast<float *>(spectrum_.data()), 1) != 0) fail("t2gpu_fft_execute");
        return spectrum_.data();
    }
    complex *data_execute(int idx_symbol, complex *ofdm_cell, float &sample_rate_offset, float &phase_offset)   // data_symbol.h:33
    {
        if (t2gpu_eq_data_execute(h_, idx_symbol, reinterpret_cast<const float *>(ofdm_cell), reinterpret_cast<float *>(cells_.data()),
                                  &sample_rate_offset, &phase_offset) < 0) fail("t2gpu_eq_data_execute");
        return cells_.data();
    }
    int fft_size = 0, c_data = 0;
private:
    t2gpu_ofdm *h_;
    std::vector<complex> spectrum_, cells_;
};

// ---------------------------------------------------------------------------------------------------------------- demodulator
enum id_device_t { id_sdrplay = 0, id_airspy, id_plutosdr };                                  // dvbt2_demodulator.h:36-40
struct signal_estimate {                                                                      // dvbt2_demodulator.h:42-52
    bool change_frequency = false;
    double coarse_freq_offset = 0.0;
    bool frequency_changed = true;
    bool change_gain = false;
    int gain_offset = 0;
    bool gain_changed = true;
    double correct_resample = 0.0;
    bool reset = false;
    bool p1_reset = false;
};

// dvbt2_demodulator (dvbt2_demodulator.h:56-176): owns the time de-interleaver and is wired to it as the reference's constructor
// wires it (dvbt2_demodulator.cpp:84-95: data -> execute, l1_dyn_execute -> l1_dyn_execute; start() is called directly, :386).
// The signals stay re-assignable std::function members.
class dvbt2_demodulator {
public:
    // fec_thread: the time de-interleaver emits on a thread of its own (time_deinterleaver's own_thread)
    dvbt2_demodulator(id_device_t id_device, float sample_rate, int device = 0, bool fec_thread = false)
        : deinterleaver(new time_deinterleaver(device, fec_thread)), h_(t2gpu_demod_create((int)id_device, sample_rate, device))
    {
        if (!h_) { delete deinterleaver; fail("t2gpu_demod_create"); }
        l1_dyn_execute = [this](const l1_postsignalling &p, int len, complex *c) { deinterleaver->l1_dyn_execute(p, len, c); };
        data = [this](int len, complex *c) { deinterleaver->execute(len, c); };
        t2gpu_demod_signals s{};
        s.user = this;
        s.start = [](void *u, const t2gpu_l1_pre *pre, const t2gpu_l1_post *post, const t2gpu_l1_plp *plp, const t2gpu_l1_dyn_plp *dyn) {
            static_cast<dvbt2_demodulator *>(u)->deinterleaver->start(*pre, pack(post, plp, dyn));
        };
        s.l1_dyn_execute = [](void *u, const t2gpu_l1_post *post, const t2gpu_l1_plp *plp, const t2gpu_l1_dyn_plp *dyn, int len, const float *cells) {
            auto *d = static_cast<dvbt2_demodulator *>(u);
            if (d->l1_dyn_execute) d->l1_dyn_execute(pack(post, plp, dyn), len, reinterpret_cast<complex *>(const_cast<float *>(cells)));
        };
        s.data = [](void *u, int len, const float *cells) {
            auto *d = static_cast<dvbt2_demodulator *>(u);
            if (d->data) d->data(len, reinterpret_cast<complex *>(const_cast<float *>(cells)));
        };
        s.amount_plp = [](void *u, int n) { auto *d = static_cast<dvbt2_demodulator *>(u); if (d->amount_plp) d->amount_plp(n); };
        s.replace_null_indicator = [](void *u, float b1, float b2) {
            auto *d = static_cast<dvbt2_demodulator *>(u);
            if (d->replace_null_indicator) d->replace_null_indicator(b1, b2);
        };
        t2gpu_demod_connect(h_, &s);
    }
    ~dvbt2_demodulator() { exit_mark("~dvbt2_demodulator"); t2gpu_demod_destroy(h_); exit_mark("~dvbt2_demodulator: handle gone"); delete deinterleaver; exit_mark("~dvbt2_demodulator: de-interleaver gone"); }
    dvbt2_demodulator(const dvbt2_demodulator &) = delete;
    dvbt2_demodulator &operator=(const dvbt2_demodulator &) = delete;

    time_deinterleaver *deinterleaver;                                                        // .h:68
    // signals (.h:70-75)
    std::function<void(float b1, float b2)> replace_null_indicator;
    std::function<void(const l1_postsignalling &l1_post, int len_in, complex *in)> l1_dyn_execute;
    std::function<void(int num_plp)> amount_plp;
    std::function<void(int len_in, complex *in)> data;
    // slot (.h:78)
    void execute(int len_in, int16_t *i_in, int16_t *q_in, signal_estimate *signal_)
    {
        t2gpu_signal_estimate s{};
        s.change_frequency = signal_->change_frequency; s.coarse_freq_offset = signal_->coarse_freq_offset;
        s.frequency_changed = signal_->frequency_changed; s.change_gain = signal_->change_gain; s.gain_offset = signal_->gain_offset;
        s.gain_changed = signal_->gain_changed; s.correct_resample = signal_->correct_resample; s.reset = signal_->reset;
        s.p1_reset = signal_->p1_reset;
        const int rc = t2gpu_demod_execute(h_, len_in, i_in, q_in, &s);
        signal_->change_frequency = s.change_frequency != 0; signal_->coarse_freq_offset = s.coarse_freq_offset;
        signal_->frequency_changed = s.frequency_changed != 0; signal_->change_gain = s.change_gain != 0; signal_->gain_offset = s.gain_offset;
        signal_->gain_changed = s.gain_changed != 0; signal_->correct_resample = s.correct_resample; signal_->reset = s.reset != 0;
        signal_->p1_reset = s.p1_reset != 0;
        if (rc != 0) fail("t2gpu_demod_execute");
    }
    // a recorded buffer has no tuner to move: see t2gpu_demod_set_tuner
    void set_tuner(double offset_hz) { if (t2gpu_demod_set_tuner(h_, offset_hz) != 0) fail("t2gpu_demod_set_tuner"); }
    // the tracking loops of a frame's data symbols on the device (t2gpu_demod_set_device_loop: the default) or on the host
    void set_device_loop(bool on) { if (t2gpu_demod_set_device_loop(h_, on ? 1 : 0) != 0) fail("t2gpu_demod_set_device_loop"); }
    // the chunk that completes a 32K data symbol and the symbol's transform as one launch (default) or two (t2gpu_demod_set_chain_one)
    void set_chain_one(bool on) { if (t2gpu_demod_set_chain_one(h_, on ? 1 : 0) != 0) fail("t2gpu_demod_set_chain_one"); }
    void set_copy_ahead(bool on) { if (t2gpu_demod_set_copy_ahead(h_, on ? 1 : 0) != 0) fail("t2gpu_demod_set_copy_ahead"); }
    // end of a stream: every symbol launched so far is read and handed on (t2gpu_demod_flush) -- before the stages behind are flushed
    void flush() { if (t2gpu_demod_flush(h_) != 0) fail("t2gpu_demod_flush"); }
    // the loop trajectory, one record of T2GPU_DEMOD_TRACE_W doubles per tracked symbol (t2gpu_demod_set_trace)
    void set_trace(double *records, long cap_records) { if (t2gpu_demod_set_trace(h_, records, cap_records) != 0) fail("t2gpu_demod_set_trace"); }
    long trace_count() const { return t2gpu_demod_trace_count(h_); }
    t2gpu_demod_info status() const { t2gpu_demod_info i{}; t2gpu_demod_status(h_, &i); return i; }
private:
    static l1_postsignalling pack(const t2gpu_l1_post *post, const t2gpu_l1_plp *plp, const t2gpu_l1_dyn_plp *dyn)
    {
        l1_postsignalling p;
        p.post = *post;
        p.plp.assign(plp, plp + post->num_plp);
        p.dyn_plp.assign(dyn, dyn + post->num_plp);
        return p;
    }
    t2gpu_demod *h_;
};

}  // namespace t2
