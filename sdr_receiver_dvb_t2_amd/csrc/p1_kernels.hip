// p1_kernels.hip -- P1 preamble detection on gfx950.
//
// Replaces p1_symbol::execute / demodulate (/root/reference/src/DVB_T2/p1_symbol.cpp:75-178,180-298). The reference runs a per-sample
// recursive correlator (delay lines and running sums of /root/reference/src/DSP/buffers.hh). Written out, its output for sample n is
//     out[n] = AC[n-964] * AB[n-2],   AC[m] = sum_{k=m-540..m} x[k] conj(s[k-542]),   AB[m] = sum_{k=m-480..m} s[k] conj(x[k-482]),
//     s[k] = x[k] * fq_shift[k mod 1024]
// (sum_of_buffer<LEN> holds LEN-1 terms: it subtracts the slot it is about to overwrite, buffers.hh:33-39), with x = 0 before the
// start of the search because reset_buffer() clears every delay line. So the correlator is a pure function of the last 2047
// samples and every sample is computed independently here: a workgroup stages the two product sequences of its 256 outputs in
// LDS and each lane adds its two windows in double. The reference's float running sums drift by rounding (sum - old + new,
// millions of times); parity of the correlation value is therefore by tolerance (tests/test_p1_gpu.py), the detection logic
// that consumes it is restated exactly.
// The threshold / arg-max state machine is sequential over samples but trivial; one workgroup runs it from LDS-staged
// correlation values, skipping in parallel over stretches below the begin threshold. Part A's 1K FFT and the 20 carrier-shift
// DBPSK decodes run in the same launch sequence; only the 64-byte result goes back to the host.
#include "p1_kernels.h"
#include "tables/dsp_tables_data.h"

#pragma clang fp contract(off)

namespace {

constexpr float PI_F = 3.14159274101257324219f;
constexpr float PI_X_2 = PI_F * 2.0f;

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) { return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }   // a * conj(b)

__device__ __forceinline__ float atan2_approx_dev(float y, float x)              // DSP/fast_math.h:61-81
{
    const float PI_2 = 1.57079637050628662109f;
    if (x == 0.0f) return y > 0.0f ? PI_2 : -PI_2;
    if (y == 0.0f) return x > 0.0f ? 0.0f : -PI_F;
    const float abs_x = fabsf(x), abs_y = fabsf(y);
    const bool min_x = abs_x < abs_y;
    const float a = min_x ? abs_x / abs_y : abs_y / abs_x;
    const float s = a * a;
    float r = ((-4.6496475e-2f * s + 1.5931422e-1f) * s - 3.2762276e-1f) * s * a + a;
    if (min_x) r = PI_2 - r;
    if (x < 0.0f) r = PI_F - r;
    if (y < 0.0f) r = -r;
    return r;
}

// ---- correlator: 256 outputs per workgroup
constexpr int P1_TILE = 128;                 // outputs per workgroup: (128+540 + 128+480) * 8 B = 10.2 KB of LDS, small enough to
constexpr int NC = P1_TILE + 540, NB = P1_TILE + 480;   // run beside another stream's LDS-heavy kernel

__global__ __launch_bounds__(P1_TILE) void p1_correlate_kernel(P1Params p)
{
    __shared__ float2 pc[NC], pb[NB];
    const int n0 = blockIdx.x * P1_TILE, tid = threadIdx.x;
    const P1Window w = p.win[blockIdx.y];
    if (n0 >= w.len) return;
    const float2 *xw = p.base + w.start;
    const int lo = -p.hist, len = w.len;
    auto x = [&](int k) { return k >= lo ? xw[k] : make_float2(0.f, 0.f); };       // zeros in front of the search
    const int f0 = p.state[blockIdx.y].idx_fq_shift;
    for (int t = tid; t < NC; t += P1_TILE) {               // pc[k] = x[k] * conj(s[k - 542]), k = n0 - 964 - 540 + t
        const int k = n0 - 1504 + t;
        float2 v = make_float2(0.f, 0.f);
        if (k < len && k >= lo) v = cmulc(x(k), cmul(x(k - 542), p.fq_shift[(f0 + k - 542) & 1023]));
        pc[t] = v;
    }
    for (int t = tid; t < NB; t += P1_TILE) {               // pb[k] = s[k] * conj(x[k - 482]), k = n0 - 2 - 480 + t
        const int k = n0 - 482 + t;
        float2 v = make_float2(0.f, 0.f);
        if (k < len && k >= lo) v = cmulc(cmul(x(k), p.fq_shift[(f0 + k) & 1023]), x(k - 482));
        pb[t] = v;
    }
    __syncthreads();
    const int n = n0 + tid;
    if (n >= len) return;
    double acr = 0.0, aci = 0.0, abr = 0.0, abi = 0.0;
    for (int j = 0; j < 541; ++j) { const float2 v = pc[tid + j]; acr += (double)v.x; aci += (double)v.y; }
    for (int j = 0; j < 481; ++j) { const float2 v = pb[tid + j]; abr += (double)v.x; abi += (double)v.y; }
    const float2 a = make_float2((float)acr, (float)aci), d = make_float2((float)abr, (float)abi);
    const float2 o = cmul(a, d);                            // :159
    p.out[w.buf_off + n] = o;
    p.corr[w.buf_off + n] = o.x * o.x + o.y * o.y;          // norm(out), :160
}

// ---- threshold / arg-max state machine (p1_symbol.cpp:93-109,162-170), one workgroup
__global__ __launch_bounds__(256) void p1_detect_kernel(P1Params p)
{
    __shared__ float sc[2048];
    __shared__ int s_first, s_stop, s_end;
    __shared__ unsigned long long s_best;
    __shared__ P1State st;                  // the detector state lives in LDS / registers during the pass, global memory only at both ends
    const int tid = threadIdx.x;
    P1Result &res = p.result[blockIdx.x];
    const P1Window w = p.win[blockIdx.x];
    const float *corr = p.corr + w.buf_off;
    const float2 *outv = p.out + w.buf_off;
    const int N = w.len;
    int n = 0;
    if (tid == 0) {
        st = p.state[blockIdx.x];
        res.status = 0; res.consumed = N; res.a_part_clipped = 0; s_stop = 0;
        if (p.gain_changed) { st.begin_threshold = p.level_detect * 2.0e+5f; st.end_threshold = 0.5f * st.begin_threshold; }   // :88-91
    }
    __syncthreads();
    while (n < N) {
        if (!st.correlation_detect) {
            // nothing but `correlation` changes while the value stays at or below the begin threshold: skip ahead in parallel
            if (tid == 0) s_first = N;
            __syncthreads();
            for (int base = n; base < N && s_first == N; base += 4096) {
                // sixteen independent loads per lane, then the first crossing among them (a loop that stops at the first crossing
                // would wait for one global load after the other)
                const int lim = min(N, base + 4096);
                const float thr = st.begin_threshold;
                float vals[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) { const int k = base + tid + 256 * u; vals[u] = k < lim ? corr[k] : thr; }
                int mine = N;
#pragma unroll
                for (int u = 15; u >= 0; --u) if (vals[u] > thr) mine = base + tid + 256 * u;
                if (mine < N) atomicMin(&s_first, mine);
                __syncthreads();
            }
            const int first = s_first;
            __syncthreads();
            if (first >= N) { if (tid == 0 && N > n) st.correlation = corr[N - 1]; n = N; break; }
            if (tid == 0 && first > n) st.correlation = corr[first - 1];
            n = first;
            __syncthreads();
        }
        // sequential stretch of up to 2048 samples from LDS
        const int cnt = min(2048, N - n);
        for (int k = tid; k < cnt; k += 256) sc[k] = corr[n + k];
        if (tid == 0) { s_end = cnt; s_best = 0ull; }
        __syncthreads();
#ifndef T2_P1_SERIAL
        // The common stretch in closed form, all lanes: the search is (or becomes, with sample 0) active and its sample counter
        // cannot reach 2048 inside the stretch. Then the loop below does nothing but (i) stop at the first k whose PREVIOUS
        // correlation is under the end threshold and (ii) keep the first strict maximum among the samples above the begin
        // threshold that come before that k; the counter is the distance from that maximum (or its old value plus the samples
        // seen). Everything else -- a stretch entered below the threshold, a counter that may overflow (:101-104) -- goes through
        // the sequential form, which is the reference's loop as written.
        const bool d0 = st.correlation_detect != 0;
        const bool fast = !p.serial_detector && (d0 || sc[0] > st.begin_threshold) && st.idx_buffer + cnt <= 2048;
        if (fast) {
            const int kd = d0 ? 0 : 1;                         // first sample whose iteration starts with the search active
            const float c_prev = st.correlation, thr_b = st.begin_threshold, thr_e = st.end_threshold;
            for (int k = kd + tid; k < cnt; k += 256)
                if ((k ? sc[k - 1] : c_prev) < thr_e) { atomicMin(&s_end, k); break; }
            __syncthreads();
            const int k_end = s_end;                           // cnt: the stretch ends with the search still active
            unsigned long long best = 0ull;                    // (value, first index): correlations are norms, their bit patterns order like the values
            for (int k = tid; k < k_end; k += 256)
                if (sc[k] > thr_b) {
                    const unsigned long long key = ((unsigned long long)__float_as_uint(sc[k]) << 32) | (0xffffffffu - (unsigned)k);
                    best = key > best ? key : best;
                }
            if (best) atomicMax(&s_best, best);
            __syncthreads();
            if (tid == 0) {
                P1State l = st;
                const bool stop = k_end < cnt;
                const int last = stop ? k_end : cnt - 1;       // last iteration that starts with the search active
                const unsigned long long b = s_best;
                const float m = __uint_as_float((unsigned)(b >> 32));
                l.correlation_detect = 1;
                if (b && m > l.max_correlation) {
                    const int k_max = (int)(0xffffffffu - (unsigned)(b & 0xffffffffu));
                    const float2 o = outv[n + k_max];
                    l.max_correlation = m; l.arg_max_re = o.x; l.arg_max_im = o.y;
                    l.idx_buffer = last - k_max;
                } else {
                    l.idx_buffer += last - kd + 1;
                }
                l.correlation = stop ? (k_end ? sc[k_end - 1] : c_prev) : sc[cnt - 1];
                if (stop) { res.status = 1; res.consumed = n + k_end + 1; res.idx_buffer_sym = l.idx_buffer; s_stop = 1; }
                st = l;
                s_first = stop ? k_end : cnt;
            }
        } else
#endif
#ifndef T2_P1_OLD
        // The stretch is one state machine, but a lone lane reading LDS once per sample spends its time on LDS latency: the first
        // wavefront runs it in step (every lane the same values), fetching 64 samples with one LDS read and handing them out of
        // a register with v_readlane; only lane 0's copies are stored.
        if (tid < 64) {
            P1State l = st;
            // every lane holds the same state: say so, and the branches below become scalar branches instead of EXEC masking
            auto uf = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
            l.correlation = uf(l.correlation); l.begin_threshold = uf(l.begin_threshold); l.end_threshold = uf(l.end_threshold);
            l.max_correlation = uf(l.max_correlation);
            l.correlation_detect = __builtin_amdgcn_readfirstlane(l.correlation_detect);
            l.idx_buffer = __builtin_amdgcn_readfirstlane(l.idx_buffer);
            const int cnt_u = __builtin_amdgcn_readfirstlane(cnt);
            int k = 0;
            float cv = 0.0f;
            int k_max = -1;                 // where the maximum moved to in this stretch: its correlator output is fetched once, at the end
            for (; k < cnt_u; ++k) {
                if ((k & 63) == 0) cv = k + tid < cnt_u ? sc[k + tid] : 0.0f;
                if (l.correlation_detect) {
                    if (++l.idx_buffer > 2048) {           // :101-104: reset_buffer() clears the correlator
                        l.correlation_detect = 0; l.max_correlation = 0.0f; l.idx_buffer = 0;
                        if (!(l.correlation < l.end_threshold)) {     // otherwise the reference falls into :106 with idx_buffer = 0
                            if (tid == 0) { res.status = 2; res.consumed = n + k; s_stop = 1; }   // the caller restarts the search at this sample
                            break;
                        }
                    }
                    if (l.correlation < l.end_threshold) {   // :106
                        if (tid == 0) { res.status = 1; res.consumed = n + k + 1; res.idx_buffer_sym = l.idx_buffer; s_stop = 1; }
                        break;
                    }
                }
                const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cv), k & 63));
                l.correlation = c;
                if (c > l.begin_threshold) {
                    l.correlation_detect = 1;
                    if (c > l.max_correlation) {
                        l.max_correlation = c;
                        k_max = k;          // on the rising flank of the peak every sample is a new maximum: no global load per sample
                        l.idx_buffer = 0;
                    }
                } else if (!l.correlation_detect) { ++k; break; }   // back to the parallel skip
            }
            if (k_max >= 0) { const float2 o = outv[n + k_max]; l.arg_max_re = o.x; l.arg_max_im = o.y; }
            if (tid == 0) { st = l; s_first = k; }
        }
#else
        if (tid == 0) {
            P1State l = st;                 // registers for the sequential stretch
            int k = 0;
            for (; k < cnt; ++k) {
                if (l.correlation_detect) {
                    if (++l.idx_buffer > 2048) {           // :101-104: reset_buffer() clears the correlator
                        l.correlation_detect = 0; l.max_correlation = 0.0f; l.idx_buffer = 0;
                        if (!(l.correlation < l.end_threshold)) {     // otherwise the reference falls into :106 with idx_buffer = 0
                            res.status = 2; res.consumed = n + k; s_stop = 1;   // the caller restarts the search at this sample
                            break;
                        }
                    }
                    if (l.correlation < l.end_threshold) {   // :106
                        res.status = 1; res.consumed = n + k + 1; res.idx_buffer_sym = l.idx_buffer; s_stop = 1;
                        break;
                    }
                }
                const float c = sc[k];
                l.correlation = c;
                if (c > l.begin_threshold) {
                    l.correlation_detect = 1;
                    if (c > l.max_correlation) {
                        l.max_correlation = c;
                        const float2 o = outv[n + k];
                        l.arg_max_re = o.x; l.arg_max_im = o.y;
                        l.idx_buffer = 0;
                    }
                } else if (!l.correlation_detect) { ++k; break; }   // back to the parallel skip
            }
            st = l;
            s_first = k;
        }
#endif
        __syncthreads();
        if (s_stop) break;
        n += s_first;
        __syncthreads();
    }
    // the correlator consumed every sample except the one that ended the search (:106-147 breaks before the update)
    if (tid == 0) {
        const int fed = res.status == 1 ? res.consumed - 1 : res.consumed;
        st.idx_fq_shift = (st.idx_fq_shift + fed) & 1023;
        res.max_correlation = st.max_correlation; res.arg_max_re = st.arg_max_re; res.arg_max_im = st.arg_max_im;
        p.state[blockIdx.x] = st;
    }
}

// ---- part A: 1K FFT (fft-shifted) + carrier search + DBPSK / S1 / S2 decode (:111-131,180-298), one workgroup
__global__ __launch_bounds__(256) void p1_decode_kernel(P1Params p)
{
    __shared__ float2 buf[1024];              // 8 KB: in-place transform (runs beside other streams' LDS-heavy kernels)
    __shared__ int ok[20], r_pre[20], r_fft[20], r_s1[20], r_s2[20];
    const int tid = threadIdx.x;
    P1State &st = p.state[blockIdx.x];
    P1Result &res = p.result[blockIdx.x];
    if (res.status != 1) return;
    // save_buffer::read()[P1_C_PART - idx_buffer + j]: the newest sample (the one that ended the search) is element 2047
    const int newest = res.consumed - 1, start = newest - 2047 + 542 - res.idx_buffer_sym;
    const float2 *x = p.base + p.win[blockIdx.x].start;
    for (int j = tid; j < 1024; j += 256) {
        const int k = start + j;
        const bool in = k >= -p.hist && k <= newest;
        buf[j] = in ? x[k] : make_float2(0.f, 0.f);
        if (!in && tid == 0) res.a_part_clipped = 1;
    }
    __syncthreads();
    // radix-2 decimation in frequency, in place: after the 10 stages bin k sits at the bit-reversed index of k
    for (int half = 512; half >= 1; half >>= 1) {
        for (int b = tid; b < 512; b += 256) {
            const int grp = b / half, k = b - grp * half, i = grp * 2 * half + k;
            const float2 a = buf[i], c = buf[i + half];
            const float2 w = p.twiddle[k * (512 / half)];               // exp(-2 pi i k / (2 half))
            buf[i] = make_float2(a.x + c.x, a.y + c.y);
            buf[i + half] = cmul(make_float2(a.x - c.x, a.y - c.y), w);
        }
        __syncthreads();
    }
    // fftshift (fast_fourier_transform::execute): out[(k + 512) mod 1024] = X[k]; X[k] = buf[bitrev10(k)]
    float2 mine[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int j = tid + 256 * u;                                     // shifted position
        const int k = (j + 512) & 1023;
        mine[u] = buf[__brev((unsigned)k) >> 22];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) { buf[tid + 256 * u] = mine[u]; p.p1_fft[1024 * blockIdx.x + tid + 256 * u] = mine[u]; }
    if (tid < 20) ok[tid] = 0;
    __syncthreads();
    const bool try_decode = !st.p1_decoded || p.reset_flag;          // :116
    if (try_decode && tid < 20) {
        const float2 *p1 = buf + 76 + tid;                           // shift = 76 + tid
        int randomize_sr = 0x4e46;                                   // init_p1_randomize, :45-55 (generated on the fly)
        int old_bit = -1, prev_desc = 0, dec_old = 1;
        uint8_t data[48];
        int idx_data = 0, next_bit = 0, cur_byte = 0;
        float2 prev = make_float2(0.f, 0.f);
        for (int i = 0; i < 384; ++i) {
            const int b = (randomize_sr ^ (randomize_sr >> 1)) & 1;
            const int rnd = b == 0 ? 1 : -1;
            randomize_sr >>= 1;
            if (b > 0) randomize_sr |= 0x4000;
            const float2 c = p1[T2_P1_ACTIVE_CARRIERS[i]];
            const float2 v = make_float2(c.x * 0.1f, c.y * 0.1f);
            int bit_i;
            if (i == 0) bit_i = old_bit;
            else {
                const float2 dif = cmulc(v, prev);
                const float angle = atan2_approx_dev(dif.y, dif.x);
                bit_i = fabsf(angle) > (PI_F / 2.0f) ? -old_bit : old_bit;
                old_bit = bit_i;
            }
            prev = v;
            prev_desc = bit_i * rnd;                                 // dbpsk_bit[i] *= p1_randomize[i]
            const int bit = prev_desc == dec_old ? 0 : 1;
            dec_old = prev_desc;
            if (next_bit == 8) { data[idx_data++] = (uint8_t)cur_byte; cur_byte = 0; next_bit = 0; }
            cur_byte = ((cur_byte << 1) + bit) & 0xFF;
            ++next_bit;
        }
        data[idx_data] = (uint8_t)cur_byte;
        bool good = true;
        int s1 = 0, s2 = 0;
        for (int i = 0; i < 8; ++i) {
            if (data[i] != data[i + 40]) { good = false; break; }
            if (data[0] == T2_P1_S1_PATTERNS[i][0]) s1 = i;
        }
        if (good) {
            for (int i = 0; i < 16; ++i)
                if (data[8] == T2_P1_S2_PATTERNS[i][0] && data[9] == T2_P1_S2_PATTERNS[i][1]) s2 = i;
            if (s1 > 4) good = false;                                // :233-252
        }
        ok[tid] = good; r_pre[tid] = s1; r_fft[tid] = s2 >> 1; r_s1[tid] = s1; r_s2[tid] = s2;
    }
    __syncthreads();
    if (tid == 0) {
        const float fs = 1.0f / (1.0e-6f * 7.0f / 64.0f);
        const float hz_per_rad = (fs / PI_X_2) / (float)(2048 << 1);                      // P1_HERTZ_PER_RADIAN
        double cfo = (double)atan2_approx_dev(st.arg_max_im, st.arg_max_re) * (double)hz_per_rad;   // :115
        res.shift = -1; res.preamble = -1; res.fft_mode = -1; res.s1 = -1; res.s2 = -1;
        if (try_decode) {
            for (int t = 0; t < 20; ++t) {
                if (ok[t]) {                                         // first shift that decodes (:117-126)
                    st.p1_decoded = 1;
                    res.shift = 76 + t; res.preamble = r_pre[t]; res.fft_mode = r_fft[t]; res.s1 = r_s1[t]; res.s2 = r_s2[t];
                    if (res.shift != 86) cfo += (double)(res.shift - 86) * (double)(fs / 1024.0f);
                    break;
                }
            }
        }
        res.p1_decoded = st.p1_decoded;
        res.coarse_freq_offset = cfo;
        // reset_buffer() (:133,300-311); the correlator history is cleared by the carry kernel
        st.correlation_detect = 0; st.max_correlation = 0.0f; st.idx_buffer = 0;
    }
}

// ---- carry the history: the last 2048 samples, or zeros after a detection / overflow reset (delay lines cleared)
__global__ __launch_bounds__(256) void p1_carry_kernel(P1Params p)
{
    __shared__ float2 keep[P1_HIST];
    const int status = p.result->status;
    for (int j = threadIdx.x; j < P1_HIST; j += 256) {
        // not finished: history = samples [n - 2048, n). finished: zeros (the caller restarts behind `consumed`)
        keep[j] = status == 0 ? p.xb[p.n + j] : make_float2(0.f, 0.f);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < P1_HIST; j += 256) p.xb[j] = keep[j];
}

}  // namespace

void launch_p1(const P1Params &p, hipStream_t stream)
{
    if (p.n <= 0 || p.n_windows <= 0) return;
    hipLaunchKernelGGL(p1_correlate_kernel, dim3((p.n + P1_TILE - 1) / P1_TILE, p.n_windows), dim3(P1_TILE), 0, stream, p);
    hipLaunchKernelGGL(p1_detect_kernel, dim3(p.n_windows), dim3(256), 0, stream, p);
    hipLaunchKernelGGL(p1_decode_kernel, dim3(p.n_windows), dim3(256), 0, stream, p);
    if (p.xb) hipLaunchKernelGGL(p1_carry_kernel, dim3(1), dim3(256), 0, stream, p);
}
