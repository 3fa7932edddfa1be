// t2gpu_rx_pool.cpp -- a stream of T2 frames over SEVERAL DEVICES of one process, one transport stream out.
// The reference is one C++ process (rx_sdrplay.cpp:199-261 -> dvbt2_demodulator::execute); what a maintainer with an 8-GPU node binds is
// therefore a C object, not a launcher: t2gpu_rx_pool takes a buffer of whole T2 frames, gives every device a contiguous range of them
// whose boundaries are multiples of the frame alignment (t2gpu_rx_pool_frame_alignment: the smallest number of T2 frames that is a whole
// number of the reference's SIMD batches, llr_demapper.cpp:742-764 -- so every batch of 32 FEC frames is formed, decoded or dropped
// exactly as a single sequential receiver forms, decodes or drops it, whichever device it lands on), runs the shares on one thread per
// device -- t2gpu_rx_execute_dev on each: NO data crosses between devices, no collective, no RCCL -- and passes the packed BBFRAMEs of all
// shares, in device order = frame order, with the LDPC stage's drop rule (ldpc_decoder.cpp:264-268) through ONE bb_de_header state machine
// (a TS packet straddles BBFRAMEs, bb_de_header.cpp:166-322, so the de-framer must see one stream).
#include "../../include/t2gpu.h"
#include "t2gpu_common.h"

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

using namespace t2gpu;

namespace {
struct Member {
    int device = 0;
    t2gpu_rx *rx = nullptr;
    int16_t *d_i = nullptr, *d_q = nullptr;
    uint8_t *h_rows = nullptr;             // page-locked: [fec frames of a share][k_bch / 8]
    int32_t *h_trials = nullptr;           // page-locked: one verdict per SIMD batch of the share
    hipStream_t stream = nullptr;
    std::thread worker;
    // the job in hand (guarded by the pool's mutex)
    bool has_job = false, done = true, first = true;
    const int16_t *src_i = nullptr, *src_q = nullptr;
    int n_frames = 0, n_fec = 0, rc = 0;
    std::string error;
};
}  // namespace

struct t2gpu_rx_pool {
    t2gpu_rx_config cfg{};
    t2gpu_rx_geometry geo{};
    int group = 32, align = 1, max_frames = 0, need_plp = 0;
    std::vector<Member> m;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    bool stop = false;
    t2gpu_bbdh *bbdh = nullptr;
    std::vector<uint8_t> ts;               // TS bytes not read yet
    size_t ts_read = 0;
    int64_t counts[6] = {0, 0, 0, 0, 0, 0};
    double last_merge_seconds = 0.0;
};

namespace {
int gcd_int(int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; }

void member_run(t2gpu_rx_pool *p, int k)
{
    Member &me = p->m[(size_t)k];
    hipSetDevice(me.device);
    for (;;) {
        std::unique_lock<std::mutex> lk(p->mu);
        p->cv_job.wait(lk, [&] { return p->stop || me.has_job; });
        if (p->stop) return;
        me.has_job = false;
        const int16_t *si = me.src_i, *sq = me.src_q;
        const int n = me.n_frames;
        const bool first = me.first;
        lk.unlock();
        int rc = 0, n_fec = 0;
        std::string err;
        if (n > 0) {
            const size_t samples = (size_t)n * (size_t)p->geo.frame_len;
            // the share's int16 I/Q to this member's device (from page-locked caller memory at the link's rate), then the whole receive
            // path on it; the rows and the batch verdicts come down behind the decode
            if (hipMemcpyAsync(me.d_i, si, samples * 2, hipMemcpyHostToDevice, me.stream) != hipSuccess ||
                hipMemcpyAsync(me.d_q, sq, samples * 2, hipMemcpyHostToDevice, me.stream) != hipSuccess) { rc = -1; err = "H2D copy of a share"; }
            if (rc == 0) {
                n_fec = t2gpu_rx_execute_dev(me.rx, me.d_i, me.d_q, n, 0.0f, first ? 1 : 0, nullptr, nullptr, me.stream);
                if (n_fec < 0) { rc = n_fec; err = last_error(); }
            }
            if (rc == 0 && n_fec > 0 && t2gpu_rx_fetch_packed(me.rx, n_fec, me.h_rows, me.h_trials) != 0) { rc = -1; err = last_error(); }
        }
        lk.lock();
        me.rc = rc; me.n_fec = n_fec; me.error = err; me.first = false;
        me.done = true;
        p->cv_done.notify_all();
    }
}

void pool_free(t2gpu_rx_pool *p)
{
    { std::lock_guard<std::mutex> lk(p->mu); p->stop = true; }
    p->cv_job.notify_all();
    for (Member &me : p->m) if (me.worker.joinable()) me.worker.join();
    for (Member &me : p->m) {
        hipSetDevice(me.device);
        if (me.rx) t2gpu_rx_destroy(me.rx);
        hipFree(me.d_i); hipFree(me.d_q);
        if (me.h_rows) hipHostFree(me.h_rows);
        if (me.h_trials) hipHostFree(me.h_trials);
        if (me.stream) hipStreamDestroy(me.stream);
    }
    if (p->bbdh) t2gpu_bbdh_destroy(p->bbdh);
    delete p;
}
}  // namespace

extern "C" t2gpu_rx_pool *t2gpu_rx_pool_create(const t2gpu_rx_config *cfg, const int *devices, int n_devices, int need_plp)
{
    if (!cfg || !devices || n_devices < 1 || n_devices > 64 || cfg->max_frames < 1) { set_error("t2gpu_rx_pool_create: bad arguments"); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { set_error("t2gpu_rx_pool_create: no usable HIP device (this library has no CPU path)"); return nullptr; }
    for (int k = 0; k < n_devices; ++k)
        if (devices[k] < 0 || devices[k] >= ndev) { set_error("t2gpu_rx_pool_create: device index out of range"); return nullptr; }
    t2gpu_rx_pool *p = new t2gpu_rx_pool();
    p->cfg = *cfg;
    p->need_plp = need_plp;
    p->group = cfg->ldpc_group > 0 ? cfg->ldpc_group : 32;
    p->max_frames = cfg->max_frames;                     // per member and call
    p->m.resize((size_t)n_devices);
    bool ok = true;
    for (int k = 0; ok && k < n_devices; ++k) {
        Member &me = p->m[(size_t)k];
        me.device = devices[k];
        ok = hipSetDevice(me.device) == hipSuccess;
        ok = ok && (me.rx = t2gpu_rx_create(cfg, me.device)) != nullptr;
        if (ok && k == 0) ok = t2gpu_rx_info(me.rx, &p->geo) == 0;
        if (!ok) break;
        const size_t samples = (size_t)p->max_frames * (size_t)p->geo.frame_len;
        const size_t rows = (size_t)p->max_frames * (size_t)p->geo.fec_frames_per_t2_frame + (size_t)p->group;
        ok = hipMalloc(&me.d_i, samples * 2) == hipSuccess && hipMalloc(&me.d_q, samples * 2) == hipSuccess &&
             hipHostMalloc(reinterpret_cast<void **>(&me.h_rows), rows * (size_t)(p->geo.k_bch / 8), hipHostMallocDefault) == hipSuccess &&
             hipHostMalloc(reinterpret_cast<void **>(&me.h_trials), (rows / (size_t)p->group + 2) * 4, hipHostMallocDefault) == hipSuccess &&
             hipStreamCreateWithFlags(&me.stream, hipStreamNonBlocking) == hipSuccess;
        if (!ok) set_error("t2gpu_rx_pool_create: device allocation failed");
    }
    if (ok) {
        p->align = p->group / gcd_int(p->group, p->geo.fec_frames_per_t2_frame);
        ok = (p->bbdh = t2gpu_bbdh_create(need_plp)) != nullptr;
    }
    if (!ok) { pool_free(p); return nullptr; }
    for (int k = 0; k < n_devices; ++k) p->m[(size_t)k].worker = std::thread(member_run, p, k);
    return p;
}

extern "C" void t2gpu_rx_pool_destroy(t2gpu_rx_pool *p) { if (p) pool_free(p); }

extern "C" int t2gpu_rx_pool_frame_alignment(const t2gpu_rx_pool *p) { return p ? p->align : -1; }
extern "C" int t2gpu_rx_pool_info(const t2gpu_rx_pool *p, t2gpu_rx_geometry *out)
{
    if (!p || !out) { set_error("t2gpu_rx_pool_info: bad arguments"); return -1; }
    *out = p->geo;
    return 0;
}

// the share of member k of n_frames frames over n members: contiguous, boundaries at multiples of the alignment (shard.py: shard_frames)
extern "C" int t2gpu_rx_pool_share(const t2gpu_rx_pool *p, int n_frames, int k, int *lo, int *hi)
{
    if (!p || k < 0 || k >= (int)p->m.size() || n_frames < 0 || !lo || !hi) { set_error("t2gpu_rx_pool_share: bad arguments"); return -1; }
    const int world = (int)p->m.size(), units = (n_frames + p->align - 1) / p->align;
    const int base = units / world, rem = units % world;
    const int lo_u = k * base + (k < rem ? k : rem), hi_u = lo_u + base + (k < rem ? 1 : 0);
    *lo = lo_u * p->align < n_frames ? lo_u * p->align : n_frames;
    *hi = hi_u * p->align < n_frames ? hi_u * p->align : n_frames;
    return 0;
}

extern "C" long t2gpu_rx_pool_execute(t2gpu_rx_pool *p, const int16_t *i_in, const int16_t *q_in, int n_frames)
{
    if (!p || !i_in || !q_in || n_frames < 0) { set_error("t2gpu_rx_pool_execute: bad arguments"); return -1; }
    if (n_frames % p->align != 0) {
        set_error("t2gpu_rx_pool_execute: n_frames must be a multiple of t2gpu_rx_pool_frame_alignment (whole SIMD batches per share)");
        return -3;
    }
    const int world = (int)p->m.size();
    {
        std::lock_guard<std::mutex> lk(p->mu);
        for (int k = 0; k < world; ++k) {
            int lo = 0, hi = 0;
            t2gpu_rx_pool_share(p, n_frames, k, &lo, &hi);
            if (hi - lo > p->max_frames) { set_error("t2gpu_rx_pool_execute: a share exceeds cfg.max_frames"); return -1; }
            Member &me = p->m[(size_t)k];
            me.src_i = i_in + (size_t)lo * (size_t)p->geo.frame_len; me.src_q = q_in + (size_t)lo * (size_t)p->geo.frame_len;
            me.n_frames = hi - lo; me.done = false; me.has_job = true;
        }
    }
    p->cv_job.notify_all();
    {
        std::unique_lock<std::mutex> lk(p->mu);
        p->cv_done.wait(lk, [&] { for (const Member &me : p->m) if (!me.done) return false; return true; });
    }
    long total = 0;
    for (const Member &me : p->m) {
        if (me.rc < 0) { set_error("t2gpu_rx_pool_execute (device " + std::to_string(me.device) + "): " + me.error); return me.rc; }
        total += me.n_fec;
    }
    // ONE de-framer, the shares in frame order
    const auto t0 = std::chrono::steady_clock::now();
    const long row_bytes = p->geo.k_bch / 8;
    for (const Member &me : p->m) {
        if (me.n_fec == 0) continue;
        const size_t need = (size_t)me.n_fec * (size_t)row_bytes + (size_t)row_bytes + 376u * ((size_t)me.n_fec + 1);
        const size_t at = p->ts.size();
        p->ts.resize(at + need);
        int64_t c[6] = {0, 0, 0, 0, 0, 0};
        const long n = t2gpu_bbdh_execute_packed_rows(p->bbdh, p->need_plp, p->geo.k_bch, me.h_rows, me.n_fec, row_bytes, me.h_trials, p->group,
                                                      p->ts.data() + at, (long)need, reinterpret_cast<long *>(c));
        if (n < 0) { p->ts.resize(at); return -1; }
        p->ts.resize(at + (size_t)n);
        for (int j = 0; j < 6; ++j) p->counts[j] += c[j];
    }
    p->last_merge_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return total;
}

extern "C" long t2gpu_rx_pool_ts_read(t2gpu_rx_pool *p, uint8_t *out, long cap)
{
    if (!p || (cap > 0 && !out) || cap < 0) { set_error("t2gpu_rx_pool_ts_read: bad arguments"); return -1; }
    const size_t have = p->ts.size() - p->ts_read;
    const size_t n = have < (size_t)cap ? have : (size_t)cap;
    if (n) std::memcpy(out, p->ts.data() + p->ts_read, n);
    p->ts_read += n;
    if (p->ts_read == p->ts.size()) { p->ts.clear(); p->ts_read = 0; }
    return (long)n;
}

extern "C" int t2gpu_rx_pool_counters(const t2gpu_rx_pool *p, int64_t *out6, double *merge_seconds)
{
    if (!p) { set_error("t2gpu_rx_pool_counters: bad arguments"); return -1; }
    if (out6) std::memcpy(out6, p->counts, sizeof p->counts);
    if (merge_seconds) *merge_seconds = p->last_merge_seconds;
    return 0;
}
