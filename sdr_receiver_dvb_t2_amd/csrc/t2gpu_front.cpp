// t2gpu_front.cpp -- C ABI of the sample-rate front end (include/t2gpu.h, "sample-rate front end"): host planner for the two
// float accumulators (front_plan.h), device buffers that carry the reference's per-object state between calls, and the
// scalar tracking loops of symbol_acquisition.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/t2gpu.h"
#include "front_kernels.h"
#include "front_plan.h"
#include "t2gpu_common.h"

using t2gpu::set_error;

struct t2gpu_front {
    int device = 0, id_device = 0, stride = 1, max_samples = 0;
    float short_to_float = 1.0f, sample_rate = 0.0f;
    double resample = 0.5, max_resample = 0.5;
    long interp_cap = 0;
    // the two accumulators that do not depend on the signal live on the host (front_plan.h)
    float phase_nco = 0.0f, frequency_nco = 0.0f, x1 = -0.5f;
    int decim_phase = 0;
    bool hold_iq = false;
    // device state and work buffers
    FrontState *d_state = nullptr;
    double *d_blk = nullptr, *d_theta = nullptr;
    float *d_lut = nullptr;
    float2 *d_derot = nullptr, *d_interp = nullptr;
    FrontRun *d_runs = nullptr, *h_runs = nullptr;            // [nco runs | farrow runs], pinned staging + device copy
    int32_t *d_index = nullptr, *h_index = nullptr;           // [nco index | farrow index]
    size_t run_cap = 0, index_cap = 0;
    hipEvent_t staged = nullptr;
    bool staged_pending = false;
    hipStream_t last_stream = nullptr;
    long last_n = 0, last_n_interp = 0;
    // host-call staging
    int16_t *d_i = nullptr, *d_q = nullptr;
    float2 *d_out = nullptr;
    std::vector<FrontRun> nco_runs, far_runs;
    std::vector<FrontRun> dev_runs;            // what d_runs holds (nco runs, then Farrow runs)
    size_t dev_nn = 0;
    bool dev_runs_valid = false;
    // short calls in one launch (front_kernels.hip: front_one_kernel); t2gpu_front_set_chain(h, 0) keeps the five launches
    long long *h_stamps = nullptr;     // T2GPU_FRONT_STAMPS=1: phase stamps of the one-launch form's last launch (printed when the handle goes)
    long stamps_fused = 0;
    char *d_one = nullptr;             // its flags [F1_MAX_GRID], done word, records [F1_MAX_GRID][16], scratch prefix [66]
    unsigned long long one_seq = 0, one_done = 0;
    int *d_chain_error = nullptr;      // raised by a look-back wait of that kernel that gave up
    bool chain_on = true;
    // the tracking loops on the device (loop_device.h; t2gpu_front_loop_*): state, the NCO runs workgroup 0 plans, a page-locked staging
    // slot for the state going up, and the chunks launched in that mode whose NCO the host has not followed yet
    T2DevLoop *d_loop = nullptr, *h_loop = nullptr, *h_loop_out = nullptr;
    unsigned *h_loop_flag = nullptr, loop_read_seq = 0;
    FrontRun *d_loop_runs = nullptr;
    std::vector<int32_t> loop_pending;
    // the state a commit leaves, stored to page-locked memory by the commit's own launch (t2gpu_front_state then reads it there)
    FrontState *h_state = nullptr;
    unsigned *h_flag = nullptr, state_seq = 0;
    bool state_published = false;      // nothing has touched the device state since the last commit
};

namespace {

bool ensure_runs(t2gpu_front *h, size_t need)
{
    if (need <= h->run_cap) return true;
    size_t cap = std::max<size_t>(need + need / 2, 1 << 16);
    if (h->staged_pending) { hipEventSynchronize(h->staged); h->staged_pending = false; }
    hipDeviceSynchronize();
    if (h->d_runs) hipFree(h->d_runs);
    if (h->h_runs) hipHostFree(h->h_runs);
    h->d_runs = nullptr; h->h_runs = nullptr; h->run_cap = 0; h->dev_runs_valid = false;
    if (hipMalloc(&h->d_runs, cap * sizeof(FrontRun)) != hipSuccess || hipHostMalloc(&h->h_runs, cap * sizeof(FrontRun)) != hipSuccess) {
        set_error("t2gpu_front: cannot allocate the run tables");
        return false;
    }
    h->run_cap = cap;
    return true;
}

// plan one execute(): fills h->nco_runs / far_runs, advances the host accumulators, returns the interpolated sample count
long plan_call(t2gpu_front *h, int n_chunks, const int32_t *chunk_len, const float *pe, const float *fe, const double *rs,
               bool nco, bool farrow, int32_t *chunk_out_len, long *n_out_total)
{
    h->nco_runs.clear(); h->far_runs.clear();
    long pos = 0, o = 0, outs = 0;
    int phase = h->decim_phase;
    for (int c = 0; c < n_chunks; ++c) {
        const int len = chunk_len[c];
        if (nco) {
            h->phase_nco = t2_wrap_2pi(h->phase_nco + (pe ? pe[c] : 0.0f));                      // dvbt2_demodulator.cpp:165-171
            t2_plan_nco(h->frequency_nco, (int)pos, len, fe ? fe[c] : 0.0f, h->phase_nco, h->nco_runs);
        }
        long made = 0;
        if (farrow) {
            const float d = (float)(rs ? rs[c] : h->resample);                                   // interpolator_farrow.hh:45
            const size_t first = h->far_runs.size();
            made = t2_plan_farrow(h->x1, (int)pos, o, len, d, h->far_runs);
            if (made < 0) { set_error("t2gpu_front: arbitrary_resample outside (1/256, 4)"); return -1; }
            for (size_t r = first; r < h->far_runs.size(); ++r) h->far_runs[r].aux = d;
        }
        const long before = (phase + o) / 2, after = (phase + o + made) / 2;                     // filter_decimator.h:88-90
        if (chunk_out_len) chunk_out_len[c] = (int32_t)(after - before);
        outs += after - before;
        o += made;
        pos += len;
    }
    *n_out_total = outs;
    return o;
}

int stage_tables(t2gpu_front *h, int n, hipStream_t stream, FrontParams &p)
{
    const size_t nn = h->nco_runs.size(), nf = h->far_runs.size();
    const size_t groups = (size_t)(n + FRONT_RUN_STRIDE - 1) / FRONT_RUN_STRIDE;
    if (!ensure_runs(h, nn + nf + 2)) return -1;
    if (2 * groups + 2 > h->index_cap) { set_error("t2gpu_front: internal index capacity"); return -1; }
    // A source that stands still (loops open, nominal resample) plans the same runs call after call: the table on the device is
    // then already the right one and nothing is shipped. (The kernels find a sample's run by binary search: no per-block index.)
    const bool same = h->dev_runs_valid && h->dev_nn == nn && h->dev_runs.size() == nn + nf &&
                      (!nn || !std::memcmp(h->dev_runs.data(), h->nco_runs.data(), nn * sizeof(FrontRun))) &&
                      (!nf || !std::memcmp(h->dev_runs.data() + nn, h->far_runs.data(), nf * sizeof(FrontRun)));
    if (!same) {
        if (h->staged_pending) { T2_HIP(hipEventSynchronize(h->staged)); h->staged_pending = false; }
        if (nn) std::memcpy(h->h_runs, h->nco_runs.data(), nn * sizeof(FrontRun));
        if (nf) std::memcpy(h->h_runs + nn, h->far_runs.data(), nf * sizeof(FrontRun));
        if (nn + nf) T2_HIP(hipMemcpyAsync(h->d_runs, h->h_runs, (nn + nf) * sizeof(FrontRun), hipMemcpyHostToDevice, stream));
        T2_HIP(hipEventRecord(h->staged, stream));
        h->staged_pending = true;
        h->dev_runs.assign(h->h_runs, h->h_runs + nn + nf);
        h->dev_nn = nn; h->dev_runs_valid = true;
    }
    p.nco_runs = h->d_runs; p.n_nco_runs = (int)nn; p.nco_index = nullptr;
    p.far_runs = h->d_runs + nn; p.n_far_runs = (int)nf; p.far_index = nullptr;
    (void)groups;
    return 0;
}

FrontParams base_params(t2gpu_front *h)
{
    FrontParams p{};
    p.stride = h->stride; p.short_to_float = h->short_to_float;
    p.state = h->d_state; p.blk = h->d_blk; p.theta_part = h->d_theta;
    p.lut_sin = h->d_lut; p.lut_cos = h->d_lut + 65536;
    p.derot = h->d_derot; p.interp = h->d_interp;
    p.decim_phase = h->decim_phase;
    return p;
}

}  // namespace

extern "C" t2gpu_front *t2gpu_front_create(int id_device, float sample_rate, int max_samples, int device)
{
    if (id_device < 0 || id_device > 2 || max_samples < 1 || !(sample_rate > 0.0f)) {
        set_error("t2gpu_front_create: bad arguments");
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev || hipSetDevice(device) != hipSuccess) {
        set_error("t2gpu_front_create: no usable HIP device (this library has no CPU path)");
        return nullptr;
    }
    t2gpu_front *h = new t2gpu_front();
    h->device = device; h->id_device = id_device; h->max_samples = max_samples; h->sample_rate = sample_rate;
    h->stride = id_device == 1 ? 2 : 1;                                                    // dvbt2_demodulator.cpp:31-50
    h->short_to_float = 1.0f / (float)(1 << (id_device == 0 ? 14 : id_device == 1 ? 12 : 11));
    const float fs = 1.0f / (1.0e-6f * 7.0f / 64.0f);                                      // SAMPLE_RATE
    // :54 `sample_rate / (SAMPLE_RATE * upsample)` in float arithmetic as the reference's own build (-Ofast, sdr_receiver_dvb_t2.pro:33-39)
    // evaluates it: a division by a constant becomes a multiplication by the constant's float reciprocal -- one float ulp (6e-8) apart
    // from the true quotient for some sample rates (9 142 875 Hz: tests/test_ref_pins_gpu.py, rxoff/rx32k)
    h->resample = sample_rate * (1.0f / (fs * 2));
    h->max_resample = h->resample + h->resample * 1.0e-4;                                  // :55
    h->interp_cap = (long)((double)max_samples / std::min(h->resample, 1.0) * 1.001) + 64;
    const size_t nb = (size_t)(max_samples + FRONT_BLOCK - 1) / FRONT_BLOCK;
    h->index_cap = 2 * ((size_t)max_samples / FRONT_RUN_STRIDE + 2) + 2;
    std::vector<float> lut(2 * 65536, 0.0f);
    {   // DSP/fast_math.h:31-42 as the reference binary evaluates it (see t2gpu_ofdm.cpp)
        const float k_table = 32767.0f / (2.0f * 3.14159274101257324219f);
        const float rk = 1.0f / k_table;
        for (int i = -32767; i < 32768; ++i) sincosf((float)i * rk, &lut[i + 32767], &lut[65536 + i + 32767]);
    }
    if (const char *e = std::getenv("T2GPU_FRONT_STAMPS"))
        if (std::atoi(e) != 0 && hipHostMalloc(reinterpret_cast<void **>(&h->h_stamps), 64 * 16 * sizeof(long long), hipHostMallocCoherent) == hipSuccess)
            std::memset(h->h_stamps, 0, 64 * 16 * sizeof(long long));
    if (hipHostMalloc(reinterpret_cast<void **>(&h->h_state), sizeof(FrontState) + 64, hipHostMallocCoherent) == hipSuccess) {
        h->h_flag = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(h->h_state) + ((sizeof(FrontState) + 15) & ~size_t(15)));
        *h->h_flag = 0;
    }
    constexpr size_t ONE_BYTES = 8 * (size_t)F1_MAX_GRID + 64 + 128 * (size_t)F1_MAX_GRID + 66 * sizeof(float2);
    bool ok = hipMalloc(&h->d_loop, sizeof(T2DevLoop)) == hipSuccess && hipMemset(h->d_loop, 0, sizeof(T2DevLoop)) == hipSuccess &&
              hipMalloc(&h->d_loop_runs, (size_t)F1_MAX_GRID * T2_LOOP_RUNS_CAP * sizeof(FrontRun)) == hipSuccess &&
              hipHostMalloc(reinterpret_cast<void **>(&h->h_loop), sizeof(T2DevLoop), hipHostMallocDefault) == hipSuccess &&
              hipMalloc(&h->d_one, ONE_BYTES) == hipSuccess && hipMemset(h->d_one, 0, ONE_BYTES) == hipSuccess &&
              hipMalloc(&h->d_chain_error, 4) == hipSuccess && hipMemset(h->d_chain_error, 0, 4) == hipSuccess &&
              hipMalloc(&h->d_state, sizeof(FrontState)) == hipSuccess && hipMalloc(&h->d_blk, ((nb + 1023) / 1024 * 1024) * 4 * sizeof(double)) == hipSuccess &&      // lane-interleaved slots (front_kernels.hip: dc_slot)
              hipMalloc(&h->d_theta, nb * 4 * sizeof(double)) == hipSuccess && hipMalloc(&h->d_lut, lut.size() * 4) == hipSuccess &&
              hipMalloc(&h->d_derot, ((size_t)max_samples + 3) * sizeof(float2)) == hipSuccess &&
              hipMalloc(&h->d_interp, ((size_t)h->interp_cap + 63) * sizeof(float2)) == hipSuccess &&
              hipMalloc(&h->d_index, h->index_cap * sizeof(int32_t)) == hipSuccess &&
              hipHostMalloc(&h->h_index, h->index_cap * sizeof(int32_t)) == hipSuccess &&
              hipEventCreateWithFlags(&h->staged, hipEventDisableTiming) == hipSuccess &&
              hipMemcpy(h->d_lut, lut.data(), lut.size() * 4, hipMemcpyHostToDevice) == hipSuccess && ensure_runs(h, 1 << 16);
    if (!ok || t2gpu_front_reset(h) != 0) {
        set_error("t2gpu_front_create: device allocation failed");
        t2gpu_front_destroy(h);
        return nullptr;
    }
    return h;
}

extern "C" void t2gpu_front_destroy(t2gpu_front *h)
{
    if (!h) return;
    hipSetDevice(h->device);
    hipDeviceSynchronize();
    if (h->h_stamps) {
        // development: the phases of the LAST one-launch form launched (front_kernels.hip T2_STAMP), microseconds from the launch's first stamp
        hipDeviceSynchronize();
        long long t0 = 0;
        for (int i = 0; i < 64 * 16; ++i) if (h->h_stamps[i] && (!t0 || h->h_stamps[i] < t0)) t0 = h->h_stamps[i];
        std::fprintf(stderr, "t2gpu_front stamps (us from the first; rows: workgroup -- 0.. front end, 48..51 FFT stage A, 52..55 stages B + C):\n");
        for (int w = 0; w < 64; ++w) {
            bool any = false;
            for (int k = 0; k < 16; ++k) any = any || h->h_stamps[16 * w + k];
            if (!any) continue;
            std::fprintf(stderr, "  wg %2d:", w);
            for (int k = 0; k < 10; ++k) { if (h->h_stamps[16 * w + k]) std::fprintf(stderr, " %6.2f", (h->h_stamps[16 * w + k] - t0) * 0.01); else std::fprintf(stderr, "      -"); }
            std::fprintf(stderr, "\n");
        }
        hipHostFree(h->h_stamps);
    }
    hipFree(h->d_one); hipFree(h->d_chain_error); hipFree(h->d_loop); hipFree(h->d_loop_runs);
    if (h->h_loop) hipHostFree(h->h_loop);
    if (h->h_loop_out) hipHostFree(h->h_loop_out);
    if (h->h_state) hipHostFree(h->h_state);
    hipFree(h->d_state); hipFree(h->d_blk); hipFree(h->d_theta); hipFree(h->d_lut); hipFree(h->d_derot); hipFree(h->d_interp);
    hipFree(h->d_runs); hipFree(h->d_index); hipFree(h->d_i); hipFree(h->d_q); hipFree(h->d_out);
    if (h->h_runs) hipHostFree(h->h_runs);
    if (h->h_index) hipHostFree(h->h_index);
    if (h->staged) hipEventDestroy(h->staged);
    delete h;
}

extern "C" int t2gpu_front_reset(t2gpu_front *h)
{
    if (!h) return -1;
    T2_HIP(hipSetDevice(h->device));
    T2_HIP(hipDeviceSynchronize());
    FrontState s{};
    s.c1 = 0.0f; s.c2 = 1.0f;                                                              // dvbt2_demodulator.h:98-99
    s.level_detect = 3.402823466e+38f;                                                     // .h:161
    T2_HIP(hipMemcpy(h->d_state, &s, sizeof s, hipMemcpyHostToDevice));
    h->state_published = false;
    T2_HIP(hipMemset(h->d_derot, 0, 3 * sizeof(float2)));
    T2_HIP(hipMemset(h->d_interp, 0, 63 * sizeof(float2)));
    h->phase_nco = 0.0f; h->frequency_nco = 0.0f; h->x1 = -0.5f; h->decim_phase = 0;
    h->last_n = 0; h->last_n_interp = 0;
    return 0;
}

// dvbt2_demodulator::reset (:111-127) touches the dc averagers and the two NCOs only: c1 / c2 / level_detect and the filter
// delay lines live on
extern "C" int t2gpu_front_reset_loops(t2gpu_front *h)
{
    if (!h) return -1;
    T2_HIP(hipSetDevice(h->device));
    T2_HIP(hipStreamSynchronize(h->last_stream));
    FrontState s;
    T2_HIP(hipMemcpy(&s, h->d_state, sizeof s, hipMemcpyDeviceToHost));
    s.dc_re = 0.0; s.dc_im = 0.0;
    T2_HIP(hipMemcpy(h->d_state, &s, sizeof s, hipMemcpyHostToDevice));
    h->state_published = false;
    h->phase_nco = 0.0f; h->frequency_nco = 0.0f;
    return 0;
}

extern "C" int t2gpu_front_set_frequency_nco(t2gpu_front *h, float frequency_nco)
{
    if (!h) return -1;
    h->frequency_nco = frequency_nco;
    return 0;
}

// c1 / c2 as a previous execute() would have left them (dvbt2_demodulator.cpp:228-234): lets a test start the sample loop from the
// state the reference's object was in when a fixture was recorded
extern "C" int t2gpu_front_set_iq(t2gpu_front *h, float c1, float c2)
{
    if (!h) return -1;
    T2_HIP(hipSetDevice(h->device));
    T2_HIP(hipStreamSynchronize(h->last_stream));
    FrontState s;
    T2_HIP(hipMemcpy(&s, h->d_state, sizeof s, hipMemcpyDeviceToHost));
    s.c1 = c1; s.c2 = c2;
    T2_HIP(hipMemcpy(h->d_state, &s, sizeof s, hipMemcpyHostToDevice));
    h->state_published = false;
    return 0;
}

// short calls (a symbol's worth of samples) as one launch (default) or as the five launches every longer call takes: same values
extern "C" int t2gpu_front_set_chain(t2gpu_front *h, int on)
{
    if (!h) return -1;
    h->chain_on = on != 0;
    return 0;
}

extern "C" int t2gpu_front_hold_iq(t2gpu_front *h, int hold)
{
    if (!h) return -1;
    h->hold_iq = hold != 0;
    return 0;
}

extern "C" int t2gpu_front_commit_iq(t2gpu_front *h, void *stream)
{
    if (!h) return -1;
    T2_HIP(hipSetDevice(h->device));
    const unsigned seq = ++h->state_seq;
    launch_front_commit_iq(h->d_state, h->h_state, h->h_state ? h->h_flag : nullptr, seq, h->d_chain_error, (hipStream_t)stream);
    T2_HIP(hipGetLastError());
    h->last_stream = (hipStream_t)stream;
    h->state_published = h->h_state != nullptr;
    return 0;
}

extern "C" int t2gpu_front_resample(const t2gpu_front *h, double *resample, double *max_resample)
{
    if (!h) return -1;
    if (resample) *resample = h->resample;
    if (max_resample) *max_resample = h->max_resample;
    return 0;
}

extern "C" long t2gpu_front_execute_dev(t2gpu_front *h, int n_chunks, const int32_t *chunk_len, const float *pe, const float *fe,
                                        const double *rs, const int16_t *d_i, const int16_t *d_q, float *d_out, long out_cap_cells,
                                        int32_t *chunk_out_len, void *stream_)
{
    if (!h || n_chunks < 0 || (n_chunks && !chunk_len) || !d_i || !d_q || !d_out) { set_error("t2gpu_front_execute: bad arguments"); return -1; }
    hipStream_t stream = (hipStream_t)stream_;
    long n = 0;
    for (int c = 0; c < n_chunks; ++c) { if (chunk_len[c] < 0) { set_error("t2gpu_front_execute: negative chunk"); return -1; } n += chunk_len[c]; }
    if (n > h->max_samples) { set_error("t2gpu_front_execute: more samples than max_samples"); return -1; }
    if (n == 0) return 0;
    T2_HIP(hipSetDevice(h->device));
    // plan on copies so that a capacity error leaves the state untouched
    const float s_phase = h->phase_nco, s_freq = h->frequency_nco, s_x1 = h->x1;
    long n_out = 0;
    const long n_interp = plan_call(h, n_chunks, chunk_len, pe, fe, rs, true, true, chunk_out_len, &n_out);
    if (n_interp < 0 || n_interp > h->interp_cap || n_out > out_cap_cells) {
        if (n_interp >= 0) set_error("t2gpu_front_execute: output does not fit (interp_cap / out_cap_cells)");
        h->phase_nco = s_phase; h->frequency_nco = s_freq; h->x1 = s_x1;
        return -1;
    }
    FrontParams p = base_params(h);
    p.i_in = d_i; p.q_in = d_q; p.n = (int)n; p.n_blocks = (int)((n + FRONT_BLOCK - 1) / FRONT_BLOCK);
    p.n_interp = n_interp; p.out = reinterpret_cast<float2 *>(d_out); p.n_out = n_out;
    p.stages = FRONT_STAGE_DEROTATE | FRONT_STAGE_FARROW | FRONT_STAGE_DECIMATE | (h->hold_iq ? FRONT_STAGE_HOLD_IQ : 0);
    h->state_published = false;
    const size_t nn = h->nco_runs.size(), nf = h->far_runs.size();
    const int one_grid = h->chain_on ? front_one_grid(p, h->far_runs.data(), nn, nf) : 0;
    if (one_grid) {
        // a symbol's worth of samples: one launch, one pass per workgroup, the run tables in its arguments (no table copy, nothing to wait for)
        FrontOneArgs a;
        a.p = p;
        a.p.n_nco_runs = (int)nn; a.p.n_far_runs = (int)nf;
        a.p.nco_runs = nullptr; a.p.far_runs = nullptr; a.p.nco_index = nullptr; a.p.far_index = nullptr;
        if (nn) std::memcpy(a.runs, h->nco_runs.data(), nn * sizeof(FrontRun));
        if (nf) std::memcpy(a.runs + nn, h->far_runs.data(), nf * sizeof(FrontRun));
        a.flags = reinterpret_cast<unsigned long long *>(h->d_one);
        a.done = reinterpret_cast<unsigned long long *>(h->d_one + 8 * (size_t)F1_MAX_GRID);
        a.rec = reinterpret_cast<double *>(h->d_one + 8 * (size_t)F1_MAX_GRID + 64);
        a.pre_out = reinterpret_cast<float2 *>(h->d_one + 8 * (size_t)F1_MAX_GRID + 64 + 128 * (size_t)F1_MAX_GRID);
        a.seq = h->one_seq + 1; a.done_target = h->one_done; a.error = h->d_chain_error;
        a.loop = nullptr; a.loop_runs = nullptr;
        a.cp_si = a.cp_sq = nullptr; a.cp_di = a.cp_dq = nullptr; a.cp_n = 0; a.cp_wgs = 0;
        a.stamps = nullptr;
        launch_front_one(a, one_grid, stream);
        // the device's words move only if the launch was accepted: the host's copies follow it, not the attempt
        T2_HIP(hipGetLastError());
        h->one_seq += 1;
        h->one_done += (unsigned long long)one_grid;
    } else {
        if (stage_tables(h, (int)n, stream, p) != 0) return -1;
        launch_front(p, stream);
    }
    T2_HIP(hipGetLastError());
    h->decim_phase = (int)((h->decim_phase + n_interp) & 1);
    h->last_stream = stream; h->last_n = n; h->last_n_interp = n_interp;
    return n_out;
}

// ---- the tracking loops on the device (include/t2gpu.h, "the loop on the device")
extern "C" void *t2gpu_front_loop_dev(t2gpu_front *h) { return h ? h->d_loop : nullptr; }

// state10 = {phase_est_filtered, frequency_est_filtered, tuner, f_kp, f_ki, f_int, p_kp, p_ki, p_int, -}: with the handle's own two NCO
// accumulators, the device's loop state from here on (stream order)
extern "C" int t2gpu_front_loop_begin(t2gpu_front *h, const float *state10, void *stream)
{
    if (!h || !state10) { set_error("t2gpu_front_loop_begin: bad arguments"); return -1; }
    if (!h->loop_pending.empty()) { set_error("t2gpu_front_loop_begin: chunks of the mode before are still to be followed"); return -1; }
    T2_HIP(hipSetDevice(h->device));
    T2DevLoop l{};                                             // (travels in the launch's arguments: nothing to stage, nothing to wait for)
    l.phase_nco = h->phase_nco; l.frequency_nco = h->frequency_nco;
    l.pe = state10[0]; l.frequency_est_filtered = state10[1]; l.tuner = state10[2]; l.fe = state10[1] + state10[2];
    l.f_kp = state10[3]; l.f_ki = state10[4]; l.f_int = state10[5]; l.p_kp = state10[6]; l.p_ki = state10[7]; l.p_int = state10[8];
    launch_front_loop_set(h->d_loop, l, (hipStream_t)stream);
    T2_HIP(hipGetLastError());
    return 0;
}

// One chunk whose NCO runs the device plans itself from its loop state (front_one_kernel with `loop`): the host plans the Farrow stage
// only. Returns the decimated cells, -1 on an error, -2 when the call does not qualify for the one-launch form (nothing has happened
// then: the caller follows the loops on the host for this chunk).
extern "C" long t2gpu_front_execute_loop_dev(t2gpu_front *h, int32_t chunk, double rs, const int16_t *d_i, const int16_t *d_q, float *d_out,
                                             long out_cap_cells, void *stream_)
{
    return t2gpu_front_loop_fft(h, chunk, rs, d_i, d_q, d_out, out_cap_cells, stream_, -1, nullptr, nullptr, nullptr);
}

long t2gpu_front_loop_fft(t2gpu_front *h, int32_t chunk, double rs, const int16_t *d_i, const int16_t *d_q, float *d_out, long out_cap_cells,
                          void *stream_, long need_out, const t2gpu::FftOneArgs *fft, int *fused, const FrontCopyAhead *ahead)
{
    if (fused) *fused = 0;
    if (!h || chunk < 1 || !d_i || !d_q || !d_out) { set_error("t2gpu_front_execute_loop_dev: bad arguments"); return -1; }
    if (chunk > h->max_samples) { set_error("t2gpu_front_execute: more samples than max_samples"); return -1; }
    hipStream_t stream = (hipStream_t)stream_;
    T2_HIP(hipSetDevice(h->device));
    const float s_x1 = h->x1;
    long n_out = 0;
    const long n_interp = plan_call(h, 1, &chunk, nullptr, nullptr, &rs, false, true, nullptr, &n_out);
    FrontParams p = base_params(h);
    p.i_in = d_i; p.q_in = d_q; p.n = chunk; p.n_blocks = (chunk + FRONT_BLOCK - 1) / FRONT_BLOCK;
    p.n_interp = n_interp; p.out = reinterpret_cast<float2 *>(d_out); p.n_out = n_out;
    p.stages = FRONT_STAGE_DEROTATE | FRONT_STAGE_FARROW | FRONT_STAGE_DECIMATE | (h->hold_iq ? FRONT_STAGE_HOLD_IQ : 0);
    const size_t nf = h->far_runs.size();
    const int one_grid = (n_interp >= 0 && n_interp <= h->interp_cap && n_out <= out_cap_cells && h->chain_on) ? front_one_grid(p, h->far_runs.data(), 0, nf) : 0;
    if (!one_grid) { h->x1 = s_x1; return n_interp < 0 ? -1 : -2; }
    h->state_published = false;
    FrontOneArgs a;
    a.p = p;
    a.p.n_nco_runs = 0; a.p.n_far_runs = (int)nf;
    a.p.nco_runs = nullptr; a.p.far_runs = nullptr; a.p.nco_index = nullptr; a.p.far_index = nullptr;
    std::memcpy(a.runs, h->far_runs.data(), nf * sizeof(FrontRun));
    a.flags = reinterpret_cast<unsigned long long *>(h->d_one);
    a.done = reinterpret_cast<unsigned long long *>(h->d_one + 8 * (size_t)F1_MAX_GRID);
    a.rec = reinterpret_cast<double *>(h->d_one + 8 * (size_t)F1_MAX_GRID + 64);
    a.pre_out = reinterpret_cast<float2 *>(h->d_one + 8 * (size_t)F1_MAX_GRID + 64 + 128 * (size_t)F1_MAX_GRID);
    a.seq = h->one_seq + 1; a.done_target = h->one_done; a.error = h->d_chain_error;
    a.loop = h->d_loop; a.loop_runs = h->d_loop_runs;
    a.cp_si = a.cp_sq = nullptr; a.cp_di = a.cp_dq = nullptr; a.cp_n = 0; a.cp_wgs = 0;
    a.stamps = h->h_stamps;
    if (ahead && ahead->n > 0) {                                  // the next chunk's I/Q comes over beside this chunk's work: 8 KB per workgroup and pass
        a.cp_si = ahead->si; a.cp_sq = ahead->sq; a.cp_di = ahead->di; a.cp_dq = ahead->dq; a.cp_n = ahead->n;
        a.cp_wgs = (int)std::min<long>(16, (ahead->n / 8 + 255) / 256);
        if (a.cp_wgs < 1) a.cp_wgs = 1;
    }
    if (fft && fused && n_out == need_out && (fft->fft_size == 32768 || fft->fft_size == 16384)) { launch_front_fft_one(a, one_grid, *fft, stream); *fused = 1; }
    else launch_front_one(a, one_grid, stream);
    T2_HIP(hipGetLastError());
    h->one_seq += 1;
    h->one_done += (unsigned long long)one_grid;
    h->decim_phase = (int)((h->decim_phase + n_interp) & 1);
    h->last_stream = stream; h->last_n = chunk; h->last_n_interp = n_interp;
    h->loop_pending.push_back(chunk);
    return n_out;
}

// the host follows the NCO of the OLDEST chunk launched by t2gpu_front_execute_loop_dev with the loop values that chunk ran on
// (dvbt2_demodulator.cpp:165-171, 187-193 through t2_plan_nco): its accumulators are then where the device's are behind that chunk
extern "C" int t2gpu_front_loop_follow(t2gpu_front *h, float pe, float fe)
{
    if (!h || h->loop_pending.empty()) { set_error("t2gpu_front_loop_follow: no chunk to follow"); return -1; }
    const int32_t len = h->loop_pending.front();
    h->loop_pending.erase(h->loop_pending.begin());
    h->phase_nco = t2_wrap_2pi(h->phase_nco + pe);
    h->nco_runs.clear();
    t2_plan_nco(h->frequency_nco, 0, len, fe, h->phase_nco, h->nco_runs);
    return 0;
}
// the host's two NCO accumulators as they stand (no device access): {phase_nco, frequency_nco}
extern "C" int t2gpu_front_nco(const t2gpu_front *h, float *out2)
{
    if (!h || !out2) return -1;
    out2[0] = h->phase_nco; out2[1] = h->frequency_nco;
    return 0;
}
extern "C" int t2gpu_front_loop_pending(const t2gpu_front *h) { return h ? (int)h->loop_pending.size() : -1; }

// waits for `stream` and brings the device's loop state down: out8 = {phase_nco, frequency_nco, pe, fe, frequency_est_filtered, f_int,
// p_int, error} -- what the host's own copies must equal when every chunk has been followed
extern "C" int t2gpu_front_loop_read(t2gpu_front *h, float *out8, void *stream)
{
    if (!h || !out8) { set_error("t2gpu_front_loop_read: bad arguments"); return -1; }
    T2_HIP(hipSetDevice(h->device));
    // by a launch of `stream` into page-locked memory, the sequence word behind it (a blocking 64-byte copy stood in the copy engine's queue
    // behind the TI block coming down: ~250 us per T2 frame on the slot-shaped path)
    if (!h->h_loop_out) {
        T2_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->h_loop_out), sizeof(T2DevLoop) + 64, hipHostMallocCoherent));
        h->h_loop_flag = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(h->h_loop_out) + sizeof(T2DevLoop));
        *h->h_loop_flag = 0;
    }
    const unsigned seq = ++h->loop_read_seq;
    launch_front_loop_get(h->d_loop, h->h_loop_out, h->h_loop_flag, seq, (hipStream_t)stream);
    T2_HIP(hipGetLastError());
    {
        volatile unsigned *flag = h->h_loop_flag;
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0; *flag != seq; ++spins) {
            t2_cpu_relax();
            if ((spins & 0xfffff) == 0xfffff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
                T2_HIP(hipStreamSynchronize((hipStream_t)stream));                 // a failed launch shows here
                if (*flag != seq) { set_error("t2gpu_front_loop_read: the loop state did not arrive"); return -1; }
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    const T2DevLoop l = *h->h_loop_out;
    out8[0] = l.phase_nco; out8[1] = l.frequency_nco; out8[2] = l.pe; out8[3] = l.fe; out8[4] = l.frequency_est_filtered; out8[5] = l.f_int;
    out8[6] = l.p_int; out8[7] = (float)l.error;
    return 0;
}

extern "C" long t2gpu_front_execute(t2gpu_front *h, int n_chunks, const int32_t *chunk_len, const float *pe, const float *fe,
                                    const double *rs, const int16_t *i_in, const int16_t *q_in, float *out, long out_cap_cells,
                                    int32_t *chunk_out_len)
{
    if (!h || !i_in || !q_in || !out) { set_error("t2gpu_front_execute: bad arguments"); return -1; }
    T2_HIP(hipSetDevice(h->device));
    long n = 0;
    for (int c = 0; c < n_chunks; ++c) n += chunk_len[c];
    if (n > h->max_samples || n < 0) { set_error("t2gpu_front_execute: more samples than max_samples"); return -1; }
    if (!h->d_i) {
        const size_t el = (size_t)h->max_samples * h->stride;
        if (hipMalloc(&h->d_i, el * 2) != hipSuccess || hipMalloc(&h->d_q, el * 2) != hipSuccess ||
            hipMalloc(&h->d_out, ((size_t)h->interp_cap / 2 + 2) * sizeof(float2)) != hipSuccess) { set_error("t2gpu_front_execute: staging allocation failed"); return -1; }
    }
    if (n == 0) return 0;
    T2_HIP(hipMemcpy(h->d_i, i_in, (size_t)n * h->stride * 2, hipMemcpyHostToDevice));
    T2_HIP(hipMemcpy(h->d_q, q_in, (size_t)n * h->stride * 2, hipMemcpyHostToDevice));
    const long cap = std::min<long>(out_cap_cells, h->interp_cap / 2 + 2);
    const long n_out = t2gpu_front_execute_dev(h, n_chunks, chunk_len, pe, fe, rs, h->d_i, h->d_q, reinterpret_cast<float *>(h->d_out), cap,
                                               chunk_out_len, nullptr);
    if (n_out < 0) return -1;
    T2_HIP(hipStreamSynchronize(nullptr));
    if (n_out) T2_HIP(hipMemcpy(out, h->d_out, (size_t)n_out * sizeof(float2), hipMemcpyDeviceToHost));
    return n_out;
}

extern "C" int t2gpu_front_state(t2gpu_front *h, float *out8)
{
    if (!h || !out8) return -1;
    T2_HIP(hipSetDevice(h->device));
    FrontState s;
    bool have = false;
    if (h->state_published) {
        // the last thing launched on this front end was a commit: its launch stores the state to page-locked memory and raises the
        // sequence word (a stream wait and a blocking copy took 95 us per execute() of the slot-shaped path)
        volatile unsigned *flag = h->h_flag;
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0; *flag != h->state_seq; ++spins) {
            t2_cpu_relax();
            if ((spins & 0xfffff) == 0xfffff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) break;
        }
        if (*flag == h->state_seq) {
            std::atomic_thread_fence(std::memory_order_acquire);
            s = *h->h_state; have = true;
            if (s.error_) { set_error("t2gpu_front: a look-back wait of the one-launch form timed out"); return -1; }
        }
    }
    if (!have) {
        T2_HIP(hipStreamSynchronize(h->last_stream));
        T2_HIP(hipMemcpy(&s, h->d_state, sizeof s, hipMemcpyDeviceToHost));
        int err = 0;
        T2_HIP(hipMemcpy(&err, h->d_chain_error, 4, hipMemcpyDeviceToHost));
        if (err) { set_error("t2gpu_front: a look-back wait of the one-launch form timed out"); return -1; }
    }
    out8[0] = (float)s.dc_re; out8[1] = (float)s.dc_im; out8[2] = s.c1; out8[3] = s.c2;
    out8[4] = h->phase_nco; out8[5] = h->frequency_nco; out8[6] = s.level_detect; out8[7] = h->x1;
    return 0;
}

// the state the LAST t2gpu_front_commit_iq left (c1 / c2 / level_detect as the reference derives them at the end of an execute(), :227-235),
// whatever has been launched on the front end since: the commit's launch stored it to page-locked memory. Waits for that launch only.
extern "C" int t2gpu_front_committed_state(t2gpu_front *h, float *out8)
{
    if (!h || !out8) return -1;
    if (!h->h_state || h->state_seq == 0) return t2gpu_front_state(h, out8);
    volatile unsigned *flag = h->h_flag;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0; (int)(*flag - h->state_seq) < 0; ++spins) {
        t2_cpu_relax();
        if ((spins & 0xfffff) == 0xfffff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) return t2gpu_front_state(h, out8);
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    const FrontState s = *h->h_state;
    if (s.error_) { set_error("t2gpu_front: a look-back wait of the one-launch form timed out"); return -1; }
    out8[0] = (float)s.dc_re; out8[1] = (float)s.dc_im; out8[2] = s.c1; out8[3] = s.c2;
    out8[4] = h->phase_nco; out8[5] = h->frequency_nco; out8[6] = s.level_detect; out8[7] = h->x1;
    return 0;
}

extern "C" long t2gpu_front_debug_stream(t2gpu_front *h, int which, float *out, long cap_cells)
{
    if (!h || !out || which < 0 || which > 1) return -1;
    T2_HIP(hipSetDevice(h->device));
    T2_HIP(hipStreamSynchronize(h->last_stream));
    // after the call the buffers' prefixes already hold the carried tail; the call's own samples start behind them
    const long n = std::min(cap_cells, which == 0 ? h->last_n : h->last_n_interp);
    // the first 3 (63) cells of the stream were overwritten by the carry when the call was shorter than the prefix; only
    // the part behind the prefix is reported intact, which is all of it for n >= 3 (63) -- callers use long calls
    const float2 *src = which == 0 ? h->d_derot + 3 : h->d_interp + 63;
    if (n > 0) T2_HIP(hipMemcpy(out, src, (size_t)n * sizeof(float2), hipMemcpyDeviceToHost));
    return n;
}

extern "C" int t2gpu_decim_execute(t2gpu_front *h, int len_in, const float *in, float *out)
{
    if (!h || len_in < 0 || !in || !out || len_in > h->interp_cap) { set_error("t2gpu_decim_execute: bad arguments"); return -1; }
    T2_HIP(hipSetDevice(h->device));
    if (len_in == 0) return 0;
    if (!h->d_out && hipMalloc(&h->d_out, ((size_t)h->interp_cap / 2 + 2) * sizeof(float2)) != hipSuccess) { set_error("t2gpu_decim_execute: staging allocation failed"); return -1; }
    T2_HIP(hipMemcpy(h->d_interp + 63, in, (size_t)len_in * sizeof(float2), hipMemcpyHostToDevice));
    FrontParams p = base_params(h);
    p.n = 0; p.n_blocks = 0; p.n_interp = len_in; p.out = h->d_out;
    p.n_out = (h->decim_phase + len_in) / 2;
    p.stages = FRONT_STAGE_DECIMATE;
    h->state_published = false;
    launch_front(p, nullptr);
    T2_HIP(hipGetLastError());
    T2_HIP(hipStreamSynchronize(nullptr));
    h->decim_phase = (h->decim_phase + len_in) & 1;
    h->last_n_interp = len_in;
    if (p.n_out) T2_HIP(hipMemcpy(out, h->d_out, (size_t)p.n_out * sizeof(float2), hipMemcpyDeviceToHost));
    return (int)p.n_out;
}

extern "C" int t2gpu_farrow_execute(t2gpu_front *h, int len_in, const float *in, double arbitrary_resample, float *out, int out_cap_cells)
{
    if (!h || len_in < 0 || !in || !out || len_in > h->max_samples) { set_error("t2gpu_farrow_execute: bad arguments"); return -1; }
    T2_HIP(hipSetDevice(h->device));
    if (len_in == 0) return 0;
    const float s_x1 = h->x1;
    const int32_t chunk = len_in;
    long n_out_unused = 0;
    const long n_interp = plan_call(h, 1, &chunk, nullptr, nullptr, &arbitrary_resample, false, true, nullptr, &n_out_unused);
    if (n_interp < 0 || n_interp > h->interp_cap || n_interp > out_cap_cells) {
        if (n_interp >= 0) set_error("t2gpu_farrow_execute: output does not fit");
        h->x1 = s_x1;
        return -1;
    }
    T2_HIP(hipMemcpy(h->d_derot + 3, in, (size_t)len_in * sizeof(float2), hipMemcpyHostToDevice));
    FrontParams p = base_params(h);
    p.n = len_in; p.n_blocks = (len_in + FRONT_BLOCK - 1) / FRONT_BLOCK; p.n_interp = n_interp; p.n_out = 0;
    p.stages = FRONT_STAGE_FARROW;
    if (stage_tables(h, len_in, nullptr, p) != 0) return -1;
    h->state_published = false;
    launch_front(p, nullptr);
    T2_HIP(hipGetLastError());
    T2_HIP(hipStreamSynchronize(nullptr));
    h->last_n = len_in; h->last_n_interp = n_interp;
    if (n_interp) T2_HIP(hipMemcpy(out, h->d_interp + 63, (size_t)n_interp * sizeof(float2), hipMemcpyDeviceToHost));
    return (int)n_interp;
}

extern "C" int t2gpu_cp_correlate_dev(const float *d_symbols, int n_symbols, int fft_size, int guard, float *d_out4, void *stream)
{
    if (!d_symbols || !d_out4 || n_symbols < 0 || fft_size < 1 || guard < 9) { set_error("t2gpu_cp_correlate_dev: bad arguments"); return -1; }
    launch_cp_correlate(reinterpret_cast<const float2 *>(d_symbols), 0, 0, n_symbols > 0 ? n_symbols : 1, n_symbols, fft_size, guard,
                        reinterpret_cast<float4 *>(d_out4), (hipStream_t)stream);
    T2_HIP(hipGetLastError());
    return 0;
}

extern "C" int t2gpu_cp_correlate_stream_dev(const float *d_stream, long first, long frame_stride, int per_frame, int n_symbols, int fft_size,
                                             int guard, float *d_out4, void *stream)
{
    if (!d_stream || !d_out4 || n_symbols < 0 || fft_size < 1 || guard < 9 || per_frame < 1 || first < 0 || frame_stride < 0) {
        set_error("t2gpu_cp_correlate_stream_dev: bad arguments");
        return -1;
    }
    launch_cp_correlate(reinterpret_cast<const float2 *>(d_stream), first, frame_stride, per_frame, n_symbols, fft_size, guard,
                        reinterpret_cast<float4 *>(d_out4), (hipStream_t)stream);
    T2_HIP(hipGetLastError());
    return 0;
}

// ---- tracking loops (host scalars) ------------------------------------------------------------------------------------
namespace {
struct PiFilter {                                                                          // DSP/loop_filters.hh:20-54
    float k_p = 0.0f, k_i = 0.0f, old_integral = 0.0f;
    void init(float damping, int bw_hz, int samplerate_hz)
    {
        const double dr = damping;
        const float theta = (float)(((1.0 * bw_hz) / samplerate_hz) / (dr + 1.0 / (4.0 * dr)));
        k_p = (float)(4.0 * dr * theta / (1.0 + 2.0 * dr * theta + (double)(theta * theta)));
        k_i = (float)(4.0 * theta * theta / (1.0 + 2.0 * dr * theta + (double)(theta * theta)));
    }
    float step(float err, float max_integral)
    {
        float integral = old_integral + k_i * err;
        const float out = integral + k_p * err;
        if (integral > max_integral) integral = max_integral;
        else if (integral < -max_integral) integral = -max_integral;
        old_integral = integral;
        return out;
    }
    void reset() { old_integral = 0.0f; }
};
}  // namespace

struct t2gpu_sync {
    PiFilter phase, freq;
    float phase_est_filtered = 0.0f, frequency_est_filtered = 0.0f, old_sample_rate_est = 0.0f;
    double sample_rate_est_filtered = 0.0, resample = 0.5, max_resample = 0.5;
};

extern "C" t2gpu_sync *t2gpu_sync_create(float sample_rate)
{
    t2gpu_sync *s = new t2gpu_sync();
    const float fs = 1.0f / (1.0e-6f * 7.0f / 64.0f);
    s->phase.init(0.3f, 1000000, (int)fs);                                                 // dvbt2_demodulator.h:106-110
    s->freq.init(0.7f, 4000000, (int)fs);                                                  // .h:112-116
    s->resample = sample_rate * (1.0f / (fs * 2));                          // (the reference binary's form of `/`: t2gpu_front_create)
    s->max_resample = s->resample + s->resample * 1.0e-4;
    return s;
}
extern "C" void t2gpu_sync_destroy(t2gpu_sync *h) { delete h; }
extern "C" void t2gpu_sync_frequency(t2gpu_sync *s, float frequency_est, int fft_size)
{
    if (s) s->frequency_est_filtered += s->freq.step(frequency_est, 1.0f / (float)fft_size);   // dvbt2_demodulator.cpp:328-330
}
extern "C" void t2gpu_sync_symbol(t2gpu_sync *s, float phase_est, float sample_rate_est)       // :429-439
{
    if (!s) return;
    s->phase_est_filtered = s->phase.step(phase_est * 0.5f, 3.14159274101257324219f * 2);
    const double step = 8.0e-9;
    if (s->old_sample_rate_est - sample_rate_est > 0.0f) {
        s->sample_rate_est_filtered -= step;
        if (s->resample - s->sample_rate_est_filtered < -s->max_resample) s->sample_rate_est_filtered += step;
    } else if (s->old_sample_rate_est - sample_rate_est < 0.0f) {
        s->sample_rate_est_filtered += step;
        if (s->resample - s->sample_rate_est_filtered > s->max_resample) s->sample_rate_est_filtered -= step;
    }
    s->old_sample_rate_est = sample_rate_est;
}
// dvbt2_demodulator::reset (:111-127): loop filters, both estimates, the sample-rate tracker and the nominal resample
extern "C" void t2gpu_sync_reset(t2gpu_sync *s, float sample_rate)
{
    if (!s) return;
    s->phase.reset(); s->freq.reset();
    s->frequency_est_filtered = 0.0f; s->old_sample_rate_est = 0.0f;            // phase_est_filtered is not touched there
    s->sample_rate_est_filtered = 0.0;
    const float fs = 1.0f / (1.0e-6f * 7.0f / 64.0f);
    s->resample = sample_rate * (1.0f / (fs * 2));                          // (the reference binary's form of `/`: t2gpu_front_create)
}
// set_guard_interval_by_brute_force, first attempt of a guard length (:484-487): the frequency estimate starts from zero
extern "C" void t2gpu_sync_clear_frequency(t2gpu_sync *s) { if (s) s->frequency_est_filtered = 0.0f; }
// symbol_acquisition after a re-tune (:291): resample -= correct_resample * resample
extern "C" void t2gpu_sync_correct_resample(t2gpu_sync *s, double correct_resample) { if (s) s->resample -= correct_resample * s->resample; }

// {phase_est_filtered, frequency_est_filtered, -, f_kp, f_ki, f_int, p_kp, p_ki, p_int, old_sample_rate_est}: t2gpu_front_loop_begin's state10 (the
// caller fills in [2], its tuner)
extern "C" void t2gpu_sync_export(const t2gpu_sync *s, float *out10)
{
    if (!s || !out10) return;
    out10[0] = s->phase_est_filtered; out10[1] = s->frequency_est_filtered; out10[2] = 0.0f;
    out10[3] = s->freq.k_p; out10[4] = s->freq.k_i; out10[5] = s->freq.old_integral;
    out10[6] = s->phase.k_p; out10[7] = s->phase.k_i; out10[8] = s->phase.old_integral; out10[9] = s->old_sample_rate_est;
}

extern "C" void t2gpu_sync_get(const t2gpu_sync *s, double *out4)
{
    if (!s || !out4) return;
    double r = s->resample - s->sample_rate_est_filtered;                                  // :157-158
    if (r > s->max_resample) r = s->max_resample;
    out4[0] = s->phase_est_filtered; out4[1] = s->frequency_est_filtered; out4[2] = s->sample_rate_est_filtered; out4[3] = r;
}

// internal (t2gpu_demod.cpp, after a disagreement between the host's copies and the device's loop state): the device's values win
extern "C" void t2gpu_sync_adopt(t2gpu_sync *s, float phase_est_filtered, float frequency_est_filtered, float f_int, float p_int)
{
    if (!s) return;
    s->phase_est_filtered = phase_est_filtered; s->frequency_est_filtered = frequency_est_filtered;
    s->freq.old_integral = f_int; s->phase.old_integral = p_int;
}
extern "C" int t2gpu_front_adopt_nco(t2gpu_front *h, float phase_nco, float frequency_nco)
{
    if (!h) return -1;
    h->phase_nco = phase_nco; h->frequency_nco = frequency_nco;
    return 0;
}

// the three values arbitrary_resample (t2gpu_sync_get's [3]) can take after the NEXT t2gpu_sync_symbol: the tracker steps by -8e-9, 0 or
// +8e-9 on the sign of a difference of two floats (dvbt2_demodulator.cpp:430-439), whatever they are
extern "C" void t2gpu_sync_candidates(const t2gpu_sync *s, double *out3)
{
    if (!s || !out3) return;
    const double step = 8.0e-9;
    for (int d = -1; d <= 1; ++d) {
        double f = s->sample_rate_est_filtered;
        if (d < 0) { f -= step; if (s->resample - f < -s->max_resample) f += step; }
        else if (d > 0) { f += step; if (s->resample - f > s->max_resample) f -= step; }
        double r = s->resample - f;
        if (r > s->max_resample) r = s->max_resample;
        out3[d + 1] = r;
    }
}

// ---- planner expansion for tests (host only) -----------------------------------------------------------------------------
extern "C" int t2gpu_plan_nco(float *frequency_nco, int n, float fe, float *values, int *n_runs)
{
    if (!frequency_nco || n < 0 || !values) return -1;
    std::vector<FrontRun> runs;
    t2_plan_nco(*frequency_nco, 0, n, fe, 0.0f, runs);
    for (size_t r = 0; r < runs.size(); ++r) {
        const int end = r + 1 < runs.size() ? runs[r + 1].i0 : n;
        for (int i = runs[r].i0; i < end; ++i) values[i] = (float)(runs[r].base + (double)(i - runs[r].i0) * runs[r].step);
    }
    if (n_runs) *n_runs = (int)runs.size();
    return 0;
}

extern "C" long t2gpu_plan_farrow(float *x1, int n, double arbitrary_resample, int32_t *counts, float *positions, int *n_runs)
{
    if (!x1 || n < 0 || !counts || !positions) return -1;
    std::vector<FrontRun> runs;
    const long total = t2_plan_farrow(*x1, 0, 0, n, (float)arbitrary_resample, runs);
    if (total < 0) return -1;
    for (size_t r = 0; r < runs.size(); ++r) {
        const int end = r + 1 < runs.size() ? runs[r + 1].i0 : n;
        for (int i = runs[r].i0; i < end; ++i) {
            counts[i] = runs[r].cnt;
            positions[i] = (float)(runs[r].base + (double)(i - runs[r].i0) * runs[r].step);
        }
    }
    if (n_runs) *n_runs = (int)runs.size();
    return total;
}
