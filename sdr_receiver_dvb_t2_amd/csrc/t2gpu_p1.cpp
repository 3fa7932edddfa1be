// t2gpu_p1.cpp -- C ABI of the P1 preamble detector (include/t2gpu.h, "P1 preamble").
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/t2gpu.h"
#include "p1_kernels.h"
#include "t2gpu_common.h"

using t2gpu::set_error;

struct t2gpu_p1 {
    int device = 0, max_samples = 0;
    float2 *d_xb = nullptr, *d_fq = nullptr, *d_tw = nullptr, *d_out = nullptr, *d_fft = nullptr;
    float *d_corr = nullptr;
    P1State *d_state = nullptr;         // [1 + MAX_WINDOWS]: stream state, then the batch windows' states
    P1Result *d_result = nullptr, *h_result = nullptr;   // [MAX_WINDOWS]
    P1Window *d_win = nullptr, *h_win = nullptr;
    P1State *h_state = nullptr;         // pinned staging for the batch states
    float2 *d_stage = nullptr;          // host-call staging
    int last_n = 0;
    // host mirror of the scalars the batch form seeds every window with
    float begin_threshold = 5.0e+5f;
    int p1_decoded = 0;
    int serial_detector = 0;
};
static constexpr int MAX_WINDOWS = 1024;

extern "C" t2gpu_p1 *t2gpu_p1_create(int max_samples, int device)
{
    if (max_samples < 1) { set_error("t2gpu_p1_create: bad arguments"); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev || hipSetDevice(device) != hipSuccess) {
        set_error("t2gpu_p1_create: no usable HIP device (this library has no CPU path)");
        return nullptr;
    }
    t2gpu_p1 *h = new t2gpu_p1();
    h->device = device; h->max_samples = max_samples;
    std::vector<float2> fq(1024), tw(1024);
    {   // p1_symbol::p1_symbol (p1_symbol.cpp:23-32): angle accumulated in float, real = sin, imag = cos
        const float angle_shift = (3.14159274101257324219f * 2.0f) / 1024.0f;
        float angle = 0.0f;
        for (int i = 0; i < 1024; ++i) { fq[i] = make_float2(sinf(angle), cosf(angle)); angle += angle_shift; }
        for (int i = 0; i < 1024; ++i) { const double a = -2.0 * M_PI * i / 1024.0; tw[i] = make_float2((float)std::cos(a), (float)std::sin(a)); }
    }
    const size_t n = (size_t)max_samples;
    bool ok = hipMalloc(&h->d_xb, (n + P1_HIST) * sizeof(float2)) == hipSuccess && hipMalloc(&h->d_fq, 1024 * sizeof(float2)) == hipSuccess &&
              hipMalloc(&h->d_tw, 1024 * sizeof(float2)) == hipSuccess && hipMalloc(&h->d_out, n * sizeof(float2)) == hipSuccess &&
              hipMalloc(&h->d_corr, n * sizeof(float)) == hipSuccess &&
              hipMalloc(&h->d_state, (1 + MAX_WINDOWS) * sizeof(P1State)) == hipSuccess &&
              hipMalloc(&h->d_result, MAX_WINDOWS * sizeof(P1Result)) == hipSuccess &&
              hipHostMalloc(&h->h_result, MAX_WINDOWS * sizeof(P1Result)) == hipSuccess &&
              hipMalloc(&h->d_win, MAX_WINDOWS * sizeof(P1Window)) == hipSuccess && hipHostMalloc(&h->h_win, MAX_WINDOWS * sizeof(P1Window)) == hipSuccess &&
              hipHostMalloc(&h->h_state, MAX_WINDOWS * sizeof(P1State)) == hipSuccess &&
              hipMalloc(&h->d_fft, MAX_WINDOWS * 1024 * sizeof(float2)) == hipSuccess &&
              hipMemcpy(h->d_fq, fq.data(), 1024 * sizeof(float2), hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(h->d_tw, tw.data(), 1024 * sizeof(float2), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok || t2gpu_p1_reset(h) != 0) {
        set_error("t2gpu_p1_create: device allocation failed");
        t2gpu_p1_destroy(h);
        return nullptr;
    }
    return h;
}

extern "C" void t2gpu_p1_destroy(t2gpu_p1 *h)
{
    if (!h) return;
    hipSetDevice(h->device);
    hipDeviceSynchronize();
    hipFree(h->d_xb); hipFree(h->d_fq); hipFree(h->d_tw); hipFree(h->d_out); hipFree(h->d_fft); hipFree(h->d_corr);
    hipFree(h->d_state); hipFree(h->d_result); hipFree(h->d_stage); hipFree(h->d_win);
    if (h->h_result) hipHostFree(h->h_result);
    if (h->h_win) hipHostFree(h->h_win);
    if (h->h_state) hipHostFree(h->h_state);
    delete h;
}

extern "C" int t2gpu_p1_reset(t2gpu_p1 *h)
{
    if (!h) return -1;
    T2_HIP(hipSetDevice(h->device));
    T2_HIP(hipDeviceSynchronize());
    P1State s{};
    s.begin_threshold = 5.0e+5f; s.end_threshold = s.begin_threshold * 0.5f;              // p1_symbol.h:63-64
    T2_HIP(hipMemcpy(h->d_state, &s, sizeof s, hipMemcpyHostToDevice));
    T2_HIP(hipMemset(h->d_xb, 0, P1_HIST * sizeof(float2)));
    h->begin_threshold = s.begin_threshold; h->p1_decoded = 0;
    return 0;
}

extern "C" int t2gpu_p1_set_serial_detector(t2gpu_p1 *h, int on)
{
    if (!h) { set_error("t2gpu_p1_set_serial_detector: null handle"); return -1; }
    h->serial_detector = on != 0;
    return 0;
}

extern "C" int t2gpu_p1_execute_dev(t2gpu_p1 *h, int gain_changed, float level_detect, int len_in, const float *d_in, int *consume,
                                    int reset_flag, t2gpu_p1_result *res, void *stream_)
{
    if (!h || !d_in || !consume || !res || len_in < 0 || *consume < 0 || *consume > len_in) { set_error("t2gpu_p1_execute: bad arguments"); return -1; }
    if (len_in - *consume > h->max_samples) { set_error("t2gpu_p1_execute: more samples than max_samples"); return -1; }
    hipStream_t stream = (hipStream_t)stream_;
    T2_HIP(hipSetDevice(h->device));
    std::memset(res, 0, sizeof *res);
    int pos = *consume;
    while (pos < len_in) {
        const int n = len_in - pos;
        T2_HIP(hipMemcpyAsync(h->d_xb + P1_HIST, reinterpret_cast<const float2 *>(d_in) + pos, (size_t)n * sizeof(float2), hipMemcpyDeviceToDevice, stream));
        h->h_win[0] = P1Window{0, n, 0};
        T2_HIP(hipMemcpyAsync(h->d_win, h->h_win, sizeof(P1Window), hipMemcpyHostToDevice, stream));
        P1Params p{};
        p.xb = h->d_xb; p.base = h->d_xb + P1_HIST; p.win = h->d_win; p.n_windows = 1; p.hist = P1_HIST; p.n = n; p.fq_shift = h->d_fq; p.twiddle = h->d_tw; p.corr = h->d_corr; p.out = h->d_out;
        p.state = h->d_state; p.result = h->d_result; p.p1_fft = h->d_fft; p.reset_flag = reset_flag;
        p.gain_changed = gain_changed; p.level_detect = level_detect; p.serial_detector = h->serial_detector;
        launch_p1(p, stream);
        T2_HIP(hipGetLastError());
        T2_HIP(hipMemcpyAsync(h->h_result, h->d_result, sizeof(P1Result), hipMemcpyDeviceToHost, stream));
        T2_HIP(hipStreamSynchronize(stream));
        const P1Result &r = *h->h_result;
        h->last_n = n;
        if (gain_changed) h->begin_threshold = level_detect * 2.0e+5f;
        pos += r.consumed;
        if (r.status == 1) {
            res->detected = 1; res->idx_buffer_sym = r.idx_buffer_sym; res->p1_decoded = r.p1_decoded; res->preamble = r.preamble;
            res->fft_mode = r.fft_mode; res->s1 = r.s1; res->s2 = r.s2; res->shift = r.shift; res->a_part_clipped = r.a_part_clipped;
            res->max_correlation = r.max_correlation; res->arg_max[0] = r.arg_max_re; res->arg_max[1] = r.arg_max_im;
            res->coarse_freq_offset = r.coarse_freq_offset;
            h->p1_decoded = r.p1_decoded;
            break;
        }
        if (r.status == 0) break;
        // status 2: the reference cleared its correlator at this sample (p1_symbol.cpp:101-104) and goes on; so do we
    }
    *consume = pos;
    return res->detected;
}

// Batch form: every window starts a fresh correlator (as the reference does after each detected P1) with the handle's thresholds and
// decoded flag; all windows run in one launch sequence.
extern "C" int t2gpu_p1_execute_batch_dev(t2gpu_p1 *h, int gain_changed, float level_detect, const float *d_stream, int n_windows,
                                          const long *win_start, const int *win_len, int reset_flag, t2gpu_p1_result *res,
                                          int *consumed, void *stream_)
{
    if (!h || !d_stream || !win_start || !win_len || !res || n_windows < 1 || n_windows > MAX_WINDOWS) { set_error("t2gpu_p1_execute_batch: bad arguments"); return -1; }
    hipStream_t stream = (hipStream_t)stream_;
    T2_HIP(hipSetDevice(h->device));
    if (gain_changed) h->begin_threshold = level_detect * 2.0e+5f;                          // p1_symbol.cpp:88-91
    long total = 0;
    int longest = 0;
    for (int w = 0; w < n_windows; ++w) {
        if (win_len[w] < 0 || win_start[w] < 0) { set_error("t2gpu_p1_execute_batch: bad window"); return -1; }
        h->h_win[w] = P1Window{win_start[w], win_len[w], (int)total};
        total += win_len[w];
        if (win_len[w] > longest) longest = win_len[w];
        P1State s{};
        s.begin_threshold = h->begin_threshold; s.end_threshold = 0.5f * h->begin_threshold; s.p1_decoded = h->p1_decoded;
        h->h_state[w] = s;
    }
    if (total > h->max_samples) { set_error("t2gpu_p1_execute_batch: windows exceed max_samples"); return -1; }
    T2_HIP(hipMemcpyAsync(h->d_win, h->h_win, n_windows * sizeof(P1Window), hipMemcpyHostToDevice, stream));
    T2_HIP(hipMemcpyAsync(h->d_state + 1, h->h_state, n_windows * sizeof(P1State), hipMemcpyHostToDevice, stream));
    P1Params p{};
    p.xb = nullptr; p.base = reinterpret_cast<const float2 *>(d_stream); p.win = h->d_win; p.n_windows = n_windows; p.hist = 0; p.n = longest;
    p.fq_shift = h->d_fq; p.twiddle = h->d_tw; p.corr = h->d_corr; p.out = h->d_out; p.state = h->d_state + 1; p.result = h->d_result;
    p.p1_fft = h->d_fft; p.reset_flag = reset_flag; p.gain_changed = 0; p.level_detect = 0.0f; p.serial_detector = h->serial_detector;
    launch_p1(p, stream);
    T2_HIP(hipGetLastError());
    T2_HIP(hipMemcpyAsync(h->h_result, h->d_result, n_windows * sizeof(P1Result), hipMemcpyDeviceToHost, stream));
    T2_HIP(hipStreamSynchronize(stream));
    int found = 0;
    for (int w = 0; w < n_windows; ++w) {
        const P1Result &r = h->h_result[w];
        t2gpu_p1_result &o = res[w];
        std::memset(&o, 0, sizeof o);
        if (consumed) consumed[w] = r.consumed;
        if (r.status != 1) continue;
        ++found;
        o.detected = 1; o.idx_buffer_sym = r.idx_buffer_sym; o.p1_decoded = r.p1_decoded; o.preamble = r.preamble; o.fft_mode = r.fft_mode;
        o.s1 = r.s1; o.s2 = r.s2; o.shift = r.shift; o.a_part_clipped = r.a_part_clipped; o.max_correlation = r.max_correlation;
        o.arg_max[0] = r.arg_max_re; o.arg_max[1] = r.arg_max_im; o.coarse_freq_offset = r.coarse_freq_offset;
        if (r.p1_decoded) h->p1_decoded = 1;
    }
    h->last_n = (int)total;
    return found;
}

extern "C" int t2gpu_p1_execute(t2gpu_p1 *h, int gain_changed, float level_detect, int len_in, const float *in, int *consume,
                                int reset_flag, t2gpu_p1_result *res)
{
    if (!h || !in || len_in < 0 || len_in > h->max_samples) { set_error("t2gpu_p1_execute: bad arguments"); return -1; }
    T2_HIP(hipSetDevice(h->device));
    if (!h->d_stage && hipMalloc(&h->d_stage, (size_t)h->max_samples * sizeof(float2)) != hipSuccess) { set_error("t2gpu_p1_execute: staging allocation failed"); return -1; }
    if (len_in) T2_HIP(hipMemcpy(h->d_stage, in, (size_t)len_in * sizeof(float2), hipMemcpyHostToDevice));
    return t2gpu_p1_execute_dev(h, gain_changed, level_detect, len_in, reinterpret_cast<const float *>(h->d_stage), consume, reset_flag, res, nullptr);
}

extern "C" int t2gpu_p1_debug(t2gpu_p1 *h, float *corr, int n_corr, float *p1_fft1024)
{
    if (!h) return -1;
    T2_HIP(hipSetDevice(h->device));
    T2_HIP(hipDeviceSynchronize());
    const int n = n_corr < h->last_n ? n_corr : h->last_n;
    if (corr && n > 0) T2_HIP(hipMemcpy(corr, h->d_corr, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    if (p1_fft1024) T2_HIP(hipMemcpy(p1_fft1024, h->d_fft, 1024 * sizeof(float2), hipMemcpyDeviceToHost));
    return n;
}
