// t2gpu_ofdm.cpp -- C-ABI of the OFDM-side stages (include/t2gpu.h): batched FFT + fftshift, data-symbol equaliser +
// frequency de-interleaver, and the host-only mode tables behind them. Compute is in ofdm_kernels.hip; no CPU fallback.
#define _GNU_SOURCE
#include "../../include/t2gpu.h"
#include "ofdm_kernels.h"
#include "ofdm_tables.h"
#include "t2gpu_common.h"
#include <cmath>
#include <cstdlib>
#include <vector>

using namespace t2gpu;

static bool mode_from_args(T2Mode &m, int fft_mode, int carrier_mode, int pilot_pattern, int guard_interval_mode, int papr_mode,
                           int n_data)
{
    m = T2Mode();
    m.fft_mode = fft_mode; m.carrier_mode = carrier_mode; m.pilot_pattern = pilot_pattern;
    m.guard_interval_mode = guard_interval_mode; m.papr_mode = papr_mode; m.n_data = n_data;
    if (n_data < 1 || n_data > 2098) return false;
    return t2_mode_init(m);
}

extern "C" int t2gpu_ofdm_mode_info(int fft_mode, int carrier_mode, int pilot_pattern, int guard_interval_mode, int papr_mode,
                                    int n_data, int *out12)
{
    T2Mode m;
    if (!out12 || !mode_from_args(m, fft_mode, carrier_mode, pilot_pattern, guard_interval_mode, papr_mode, n_data)) {
        set_error("t2gpu_ofdm_mode_info: unsupported mode (16K / 32K SISO only)");
        return -1;
    }
    const int v[12] = {m.fft_size, m.k_total, m.k_ext, m.k_offset, m.l_nulls, m.c_p2, m.c_data, m.n_fc, m.c_fc, m.l_fc,
                       m.len_frame, m.guard_interval_size};
    for (int i = 0; i < 12; ++i) out12[i] = v[i];
    return 0;
}

extern "C" int t2gpu_table_symbol_carriers(int fft_mode, int carrier_mode, int pilot_pattern, int guard_interval_mode,
                                           int papr_mode, int n_data, int idx_symbol, uint8_t *map, float *refer)
{
    T2Mode m;
    if (!map || !refer || !mode_from_args(m, fft_mode, carrier_mode, pilot_pattern, guard_interval_mode, papr_mode, n_data) ||
        idx_symbol < 0 || idx_symbol >= m.len_frame) {
        set_error("t2gpu_table_symbol_carriers: bad arguments");
        return -1;
    }
    std::vector<uint8_t> mp; std::vector<float> rf;
    t2_symbol_carriers(m, idx_symbol, mp, rf);
    std::copy(mp.begin(), mp.end(), map);
    std::copy(rf.begin(), rf.end(), refer);
    return m.k_total;
}

extern "C" int t2gpu_table_freq_deint(int fft_mode, int carrier_mode, int pilot_pattern, int guard_interval_mode, int papr_mode,
                                      int n_data, int kind, int32_t *h_even, int32_t *h_odd)
{
    T2Mode m;
    if (!h_even || !h_odd || kind < 0 || kind > 2 ||
        !mode_from_args(m, fft_mode, carrier_mode, pilot_pattern, guard_interval_mode, papr_mode, n_data)) {
        set_error("t2gpu_table_freq_deint: bad arguments");
        return -1;
    }
    std::vector<int32_t> he, ho;
    t2_freq_deint(m, kind, he, ho);
    std::copy(he.begin(), he.end(), h_even);
    std::copy(ho.begin(), ho.end(), h_odd);
    return (int)he.size();
}

// pilot-to-pilot segments of one symbol: (left pilot, right pilot, first de-interleaver index, data cells between).
// The centre carrier is never a segment boundary (data_symbol.cpp:224-256, p2_symbol.cpp:195-204).
static int build_segments(const std::vector<uint8_t> &mp, int K, std::vector<int4> &segs)
{
    int left = 0, d = 0, n = 0;
    for (int k = 1; k < K; ++k) {
        const bool pilot = (mp[k] == T2_SCATTERED || mp[k] == T2_CONTINUAL || mp[k] == T2_P2PILOT) && k != K / 2;
        if (mp[k] == T2_DATA) ++n;            // a data cell on the centre carrier is buffered like any other (data_symbol.cpp:226-229)
        if (pilot) {
            segs.push_back(make_int4(left, k, d, n));
            d += n; n = 0; left = k;
        }
    }
    return d;
}

struct t2gpu_ofdm {
    T2Mode m;
    int device = 0, max_symbols = 0, rows = 0, num_cu = 256;
    EqParams eq{}, eq_p2{}, eq_fc{};
    float2 *d_twiddle = nullptr;
    float2 *d_fft_scratch = nullptr;   // the first exchange of the two-launch FFT of a one- or two-symbol call (ofdm_kernels.h)
    unsigned *d_fft_count = nullptr;   // launch_fft_sym_sync's workgroup counter
    bool one_launch = true;            // t2gpu_ofdm_set_one_launch: a symbol's transform + floats as one launch or two
    uint8_t *d_map = nullptr;
    uint16_t *d_dcar = nullptr, *d_dcar_p2 = nullptr, *d_dcar_fc = nullptr;   // carrier of every data cell, per table row
    float *d_refer = nullptr;
    float2 *d_lut = nullptr;
    int4 *d_segs = nullptr, *d_segs_p2 = nullptr;
    int32_t *d_seg_count = nullptr, *d_h_even = nullptr, *d_h_odd = nullptr;
    uint8_t *d_map_p2 = nullptr;
    float *d_refer_p2 = nullptr;
    int32_t *d_seg_count_p2 = nullptr, *d_h_even_p2 = nullptr, *d_h_odd_p2 = nullptr;
    uint8_t *d_map_fc = nullptr;
    float *d_refer_fc = nullptr;
    int4 *d_segs_fc = nullptr;
    int32_t *d_seg_count_fc = nullptr, *d_h_even_fc = nullptr, *d_h_odd_fc = nullptr, *d_index_fc = nullptr;
    float4 *d_pilot_scratch = nullptr, *d_pilot_scratch_p2 = nullptr, *d_pilot_scratch_fc = nullptr;   // per table: the three launches of a frame batch may run side by side
    uint16_t *d_cellq = nullptr, *d_cellq_p2 = nullptr, *d_cellq_fc = nullptr;   // output-range form of the equaliser (EqParams::cellq, sel)
    uint32_t *d_sel = nullptr, *d_sel_p2 = nullptr, *d_sel_fc = nullptr;
    // host-call staging
    float2 *d_in = nullptr, *d_out = nullptr, *d_sync = nullptr;
    int32_t *d_index = nullptr, *d_index_zero = nullptr;   // d_index_zero: max_symbols zeros (every P2 symbol of a batch is frame symbol 0)
};

extern "C" t2gpu_ofdm *t2gpu_ofdm_create(int fft_mode, int carrier_mode, int pilot_pattern, int guard_interval_mode, int papr_mode,
                                         int n_data, int max_symbols, int device)
{
    T2Mode m;
    if (max_symbols < 1 || !mode_from_args(m, fft_mode, carrier_mode, pilot_pattern, guard_interval_mode, papr_mode, n_data)) {
        set_error("t2gpu_ofdm_create: unsupported mode (16K / 32K SISO only) or bad arguments");
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev || hipSetDevice(device) != hipSuccess) {
        set_error("t2gpu_ofdm_create: no usable HIP device (this library has no CPU path)");
        return nullptr;
    }
    t2gpu_ofdm *h = new t2gpu_ofdm();
    h->m = m; h->device = device; h->max_symbols = max_symbols;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) h->num_cu = prop.multiProcessorCount;
    const int N = m.fft_size, K = m.k_total;
    // FFT twiddles W_N^m in double, stored as float
    // then, for the kernel's stage twiddles (powers of W^t, ofdm_kernels.hip: cpow_table): W^(t 2^b) for the T = N / 32 lanes, five
    // planes of T (a wavefront reads 64 consecutive entries), and W^(32 t1 2^b) for t1 < T / 32, five planes
    const int T = N / 32, T2 = T / 32;
    std::vector<float2> tw(N + 5 * T + 5 * T2);
    auto wn = [&](long m) {
        const double a = -2.0 * M_PI * (double)(m % N) / (double)N;
        return make_float2((float)std::cos(a), (float)std::sin(a));
    };
    for (int i = 0; i < N; ++i) tw[i] = wn(i);
    for (int b = 0; b < 5; ++b) {
        for (int t = 0; t < T; ++t) tw[N + b * T + t] = wn((long)t << b);
        for (int t1 = 0; t1 < T2; ++t1) tw[N + 5 * T + b * T2 + t1] = wn((long)(32 * t1) << b);
    }
    // LUT of DSP/fast_math.h:27-42 as the reference binary fills it (-Ofast: i * (1/k) in float, one sincosf)
    std::vector<float2> lut(65536, make_float2(0.0f, 0.0f));                   // (cos, sin)
    {
        const float k_table = 32767.0f / (2.0f * 3.14159274101257324219f);
        const float rk = 1.0f / k_table;
        for (int i = -32767; i < 32768; ++i) sincosf((float)i * rk, &lut[i + 32767].y, &lut[i + 32767].x);
    }
    // per data symbol: carrier map, pilot reference, pilot-to-pilot segments
    const int rows = m.n_data - m.l_fc;          // frame-closing symbol is handled by its own stage
    h->rows = rows;
    std::vector<uint8_t> map((size_t)rows * K);
    std::vector<float> refer((size_t)rows * K);
    std::vector<std::vector<int4>> segs(rows);
    std::vector<uint16_t> dcar;
    int max_seg = 0;
    for (int r = 0; r < rows; ++r) {
        std::vector<uint8_t> mp; std::vector<float> rf;
        t2_symbol_carriers(m, m.n_p2 + r, mp, rf);
        std::copy(mp.begin(), mp.end(), map.begin() + (size_t)r * K);
        std::copy(rf.begin(), rf.end(), refer.begin() + (size_t)r * K);
        for (int i = 0; i < K; ++i)
            if (mp[i] == T2_DATA) dcar.push_back((uint16_t)i);
        int d = build_segments(mp, K, segs[r]);
        if (d != m.c_data || dcar.size() != (size_t)(r + 1) * m.c_data) { set_error("carrier map does not hold c_data data cells"); delete h; return nullptr; }
        max_seg = std::max(max_seg, (int)segs[r].size());
    }
    std::vector<int4> segflat((size_t)rows * max_seg, make_int4(0, 0, 0, 0));
    std::vector<int32_t> segcount(rows);
    for (int r = 0; r < rows; ++r) {
        segcount[r] = (int32_t)segs[r].size();
        std::copy(segs[r].begin(), segs[r].end(), segflat.begin() + (size_t)r * max_seg);
    }
    std::vector<int32_t> he, ho;
    t2_freq_deint(m, 1, he, ho);
    bool ok = true;
    auto up = [&](auto **dst, const void *src, size_t bytes) {
        ok = ok && hip_ok(hipMalloc((void **)dst, bytes), "hipMalloc") && hip_ok(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice), "hipMemcpy");
    };
    up(&h->d_twiddle, tw.data(), tw.size() * sizeof(float2));
    ok = ok && hip_ok(hipMalloc((void **)&h->d_fft_scratch, (size_t)FFT_WIDE_SYMBOLS * m.fft_size * sizeof(float2)), "hipMalloc");
    ok = ok && hip_ok(hipMalloc((void **)&h->d_fft_count, 64), "hipMalloc") && hip_ok(hipMemset(h->d_fft_count, 0, 64), "hipMemset");
    up(&h->d_lut, lut.data(), lut.size() * sizeof(float2));
    up(&h->d_map, map.data(), map.size());
    up(&h->d_dcar, dcar.data(), dcar.size() * 2);
    up(&h->d_refer, refer.data(), refer.size() * 4);
    up(&h->d_segs, segflat.data(), segflat.size() * sizeof(int4));
    up(&h->d_seg_count, segcount.data(), segcount.size() * 4);
    up(&h->d_h_even, he.data(), he.size() * 4);
    up(&h->d_h_odd, ho.data(), ho.size() * 4);
    // the P2 symbol: one table row of its own (pilots every 3rd / 6th carrier, c_p2 cells, P2 frequency de-interleaver) -- built
    // for EXTENDED carriers whatever the mode: the reference's p2_symbol keeps the tables of init_dvbt2 (dvbt2_p2_parameters_init
    // forces CARRIERS_EXTENDED, dvbt2_definition.cpp:88-90) and never rebuilds them after L1-pre, so a normal-carrier signal's P2 is
    // read with k_ext noise carriers on either side taken for pilots (c_p2 is the same in both modes, the data cells line up)
    T2Mode mP2;
    mode_from_args(mP2, fft_mode, 1, pilot_pattern, guard_interval_mode, papr_mode, n_data);
    const int K2 = mP2.k_total;
    std::vector<uint8_t> mp2; std::vector<float> rf2; std::vector<int4> seg2;
    t2_symbol_carriers(mP2, 0, mp2, rf2);
    if (build_segments(mp2, K2, seg2) != m.c_p2) { set_error("P2 carrier map does not hold c_p2 data cells"); t2gpu_ofdm_destroy(h); return nullptr; }
    const int32_t nseg2 = (int32_t)seg2.size();
    std::vector<int32_t> he2, ho2;
    t2_freq_deint(mP2, 0, he2, ho2);
    std::vector<uint16_t> dcar2;
    for (int i = 0; i < K2; ++i)
        if (mp2[i] == T2_DATA) dcar2.push_back((uint16_t)i);
    up(&h->d_dcar_p2, dcar2.data(), dcar2.size() * 2);
    up(&h->d_map_p2, mp2.data(), mp2.size());
    up(&h->d_refer_p2, rf2.data(), rf2.size() * 4);
    up(&h->d_segs_p2, seg2.data(), seg2.size() * sizeof(int4));
    up(&h->d_seg_count_p2, &nseg2, 4);
    up(&h->d_h_even_p2, he2.data(), he2.size() * 4);
    up(&h->d_h_odd_p2, ho2.data(), ho2.size() * 4);
    // the frame-closing symbol (when the mode has one): pilots every dx carriers, n_fc cells, its own de-interleaver
    int nseg3 = 0;
    std::vector<int4> seg3_keep;
    std::vector<int32_t> he3_keep, ho3_keep;
    std::vector<uint16_t> dcar3_keep;
    if (m.l_fc) {
        std::vector<uint8_t> mp3; std::vector<float> rf3; std::vector<int4> seg3;
        t2_symbol_carriers(m, m.len_frame - 1, mp3, rf3);
        if (build_segments(mp3, K, seg3) != m.n_fc) { set_error("frame-closing carrier map does not hold n_fc data cells"); t2gpu_ofdm_destroy(h); return nullptr; }
        nseg3 = (int)seg3.size();
        seg3_keep = seg3;
        const int32_t n3 = nseg3, fcidx = m.len_frame - 1;
        std::vector<int32_t> he3, ho3;
        t2_freq_deint(m, 2, he3, ho3);
        std::vector<int32_t> idxfill(max_symbols, fcidx);
        std::vector<uint16_t> dcar3;
        for (int i = 0; i < K; ++i)
            if (mp3[i] == T2_DATA) dcar3.push_back((uint16_t)i);
        up(&h->d_dcar_fc, dcar3.data(), dcar3.size() * 2);
        he3_keep = he3; ho3_keep = ho3; dcar3_keep = dcar3;
        up(&h->d_map_fc, mp3.data(), mp3.size());
        up(&h->d_refer_fc, rf3.data(), rf3.size() * 4);
        up(&h->d_segs_fc, seg3.data(), seg3.size() * sizeof(int4));
        up(&h->d_seg_count_fc, &n3, 4);
        up(&h->d_h_even_fc, he3.data(), he3.size() * 4);
        up(&h->d_h_odd_fc, ho3.data(), ho3.size() * 4);
        up(&h->d_index_fc, idxfill.data(), idxfill.size() * 4);
    }
    const int max_all = std::max(std::max(max_seg, (int)nseg2), nseg3);
    (void)max_all;
    ok = ok && hip_ok(hipMalloc(&h->d_pilot_scratch, (size_t)max_symbols * (max_seg + 1) * sizeof(float4)), "hipMalloc");
    ok = ok && hip_ok(hipMalloc(&h->d_pilot_scratch_p2, (size_t)max_symbols * (nseg2 + 1) * sizeof(float4)), "hipMalloc");
    if (m.l_fc) ok = ok && hip_ok(hipMalloc(&h->d_pilot_scratch_fc, (size_t)max_symbols * (nseg3 + 1) * sizeof(float4)), "hipMalloc");
    if (!ok) { t2gpu_ofdm_destroy(h); return nullptr; }
    h->eq = EqParams{N, m.l_nulls, K, m.c_data, m.n_p2, max_seg, m.amp_sp, m.amp_cp, m.amp_p2, h->d_map, h->d_refer, h->d_segs,
                     h->d_seg_count, h->d_h_even, h->d_h_odd, h->d_lut};
    if (m.l_fc)
        h->eq_fc = EqParams{N, m.l_nulls, K, m.n_fc, m.len_frame - 1, nseg3, m.amp_sp, m.amp_cp, m.amp_p2, h->d_map_fc, h->d_refer_fc,
                            h->d_segs_fc, h->d_seg_count_fc, h->d_h_even_fc, h->d_h_odd_fc, h->d_lut};
    h->eq_p2 = EqParams{N, mP2.l_nulls, K2, m.c_p2, 0, (int)nseg2, m.amp_sp, m.amp_cp, m.amp_p2, h->d_map_p2, h->d_refer_p2, h->d_segs_p2,
                        h->d_seg_count_p2, h->d_h_even_p2, h->d_h_odd_p2, h->d_lut};
    h->eq.dcar = h->d_dcar; h->eq.dcar_stride = m.c_data;
    h->eq_p2.dcar = h->d_dcar_p2; h->eq_p2.dcar_stride = m.c_p2;
    h->eq_fc.dcar = h->d_dcar_fc; h->eq_fc.dcar_stride = m.n_fc;
    h->eq_p2.recip_amp = 1;
    h->eq_fc.recip_amp = 1;
    // Output-range form (eq_split_kernel): per table row the output position of every data cell (the de-interleaver of the row's
    // parity, data_symbol.cpp:148-149) and, range by range, the (carrier, output position) list of the cells that land in it.
    // Ranges: the fewest that fit 75 KB of LDS each -- two workgroups per CU (T2GPU_EQ_SPLITS=n forces n >= 1: tests).
    {
        int forced = -1;
        if (const char *e = getenv("T2GPU_EQ_SPLITS")) forced = atoi(e) >= 1 ? atoi(e) : -1;
        auto build = [&](EqParams &q, int n_rows, int first_idx, const std::vector<int32_t> &hev, const std::vector<int32_t> &hod,
                         const std::vector<uint16_t> &dc, const std::vector<int4> *sg_rows, uint16_t **d_cq, uint32_t **d_sl) {
            const int C = q.c_data;
            int ns = forced >= 0 ? forced : (C * 8 + 75 * 1024 - 1) / (75 * 1024);
            while (C / ns + 2 > 14 * 1024) ++ns;                                // a range is at most 28 cells per lane of 512
            if (C >= 65535) { q.n_splits = 0; return; }                         // (16-bit output positions; no supported mode comes near)
            int steps = 0;
            for (int r = 0; r < n_rows; ++r)
                for (const int4 &g : sg_rows[r]) steps = std::max(steps, g.w);
            steps = (steps + EQS_PU - 1) / EQS_PU * EQS_PU;
            std::vector<uint16_t> cq((size_t)n_rows * steps * q.max_seg, (uint16_t)0xffff);
            std::vector<uint32_t> sl((size_t)n_rows * C);
            std::vector<int> fill(ns);
            for (int r = 0; r < n_rows; ++r) {
                const std::vector<int32_t> &hh = ((first_idx + r) & 1) ? hev : hod;
                for (size_t g = 0; g < sg_rows[r].size(); ++g)
                    for (int k = 0; k < sg_rows[r][g].w; ++k)
                        cq[(((size_t)r * (steps / 4) + k / 4) * q.max_seg + g) * 4 + (k & 3)] = (uint16_t)hh[sg_rows[r][g].z + k];
                for (int sidx = 0; sidx < ns; ++sidx) fill[sidx] = eq_split_q(C, ns, sidx);
                for (int d = 0; d < C; ++d) {
                    const int qq = hh[d];
                    int sidx = (int)((long)qq * ns / C);                         // the range qq falls in (boundaries are rounded down to even)
                    while (sidx + 1 < ns && qq >= eq_split_q(C, ns, sidx + 1)) ++sidx;
                    while (sidx > 0 && qq < eq_split_q(C, ns, sidx)) --sidx;
                    sl[(size_t)r * C + fill[sidx]++] = ((uint32_t)dc[(size_t)r * C + d] << 16) | (uint32_t)qq;
                }
            }
            up(d_cq, cq.data(), cq.size() * 2);
            up(d_sl, sl.data(), sl.size() * 4);
            q.cellq = *d_cq; q.cq_steps = steps; q.sel = *d_sl; q.n_splits = ns;
        };
        build(h->eq, rows, m.n_p2, he, ho, dcar, segs.data(), &h->d_cellq, &h->d_sel);
        build(h->eq_p2, 1, 0, he2, ho2, dcar2, &seg2, &h->d_cellq_p2, &h->d_sel_p2);
        if (m.l_fc) build(h->eq_fc, 1, m.len_frame - 1, he3_keep, ho3_keep, dcar3_keep, &seg3_keep, &h->d_cellq_fc, &h->d_sel_fc);
        if (!ok) { t2gpu_ofdm_destroy(h); return nullptr; }
    }
    return h;
}

extern "C" void t2gpu_ofdm_destroy(t2gpu_ofdm *h)
{
    if (!h) return;
    hipFree(h->d_twiddle); hipFree(h->d_fft_scratch); hipFree(h->d_fft_count); hipFree(h->d_lut); hipFree(h->d_map); hipFree(h->d_refer); hipFree(h->d_segs); hipFree(h->d_seg_count);
    hipFree(h->d_h_even); hipFree(h->d_h_odd); hipFree(h->d_pilot_scratch); hipFree(h->d_pilot_scratch_p2); hipFree(h->d_pilot_scratch_fc); hipFree(h->d_in); hipFree(h->d_out);
    hipFree(h->d_dcar); hipFree(h->d_dcar_p2); hipFree(h->d_dcar_fc);
    hipFree(h->d_cellq); hipFree(h->d_cellq_p2); hipFree(h->d_cellq_fc); hipFree(h->d_sel); hipFree(h->d_sel_p2); hipFree(h->d_sel_fc);
    hipFree(h->d_sync); hipFree(h->d_index); hipFree(h->d_index_zero); hipFree(h->d_map_p2); hipFree(h->d_refer_p2); hipFree(h->d_segs_p2);
    hipFree(h->d_seg_count_p2); hipFree(h->d_h_even_p2); hipFree(h->d_h_odd_p2);
    hipFree(h->d_map_fc); hipFree(h->d_refer_fc); hipFree(h->d_segs_fc); hipFree(h->d_seg_count_fc); hipFree(h->d_h_even_fc);
    hipFree(h->d_h_odd_fc); hipFree(h->d_index_fc);
    delete h;
}

extern "C" int t2gpu_fft_execute_dev(t2gpu_ofdm *h, const float *d_in, float *d_out, int n_symbols, void *stream)
{
    if (!h || !d_in || !d_out || n_symbols < 1) { set_error("t2gpu_fft_execute_dev: bad arguments"); return -1; }
    T2_HIP(launch_fft(h->m.fft_size, reinterpret_cast<const float2 *>(d_in), reinterpret_cast<float2 *>(d_out), h->d_twiddle, n_symbols,
                      h->num_cu, (hipStream_t)stream, nullptr, h->d_fft_scratch, FFT_WIDE_SYMBOLS));
    return 0;
}

extern "C" int t2gpu_fft_execute_strided_dev(t2gpu_ofdm *h, const float *d_stream, long first, long frame_stride, int per_frame,
                                             int sym_stride, float *d_out, int n_symbols, void *stream)
{
    if (!h || !d_stream || !d_out || n_symbols < 1 || per_frame < 1 || first < 0 || sym_stride < 0 || frame_stride < 0) {
        set_error("t2gpu_fft_execute_strided_dev: bad arguments");
        return -1;
    }
    const FftLayout lay{first, frame_stride, per_frame, sym_stride};
    T2_HIP(launch_fft(h->m.fft_size, reinterpret_cast<const float2 *>(d_stream), reinterpret_cast<float2 *>(d_out), h->d_twiddle, n_symbols,
                      h->num_cu, (hipStream_t)stream, &lay, h->d_fft_scratch, FFT_WIDE_SYMBOLS));
    return 0;
}

static int ensure_staging(t2gpu_ofdm *h)
{
    if (h->d_in) return 0;
    T2_HIP(hipSetDevice(h->device));
    T2_HIP(hipMalloc(&h->d_in, (size_t)h->max_symbols * h->m.fft_size * sizeof(float2)));
    T2_HIP(hipMalloc(&h->d_out, (size_t)h->max_symbols * h->m.fft_size * sizeof(float2)));
    T2_HIP(hipMalloc(&h->d_sync, (size_t)h->max_symbols * sizeof(float2)));
    T2_HIP(hipMalloc(&h->d_index, (size_t)h->max_symbols * 4));
    T2_HIP(hipMalloc(&h->d_index_zero, (size_t)h->max_symbols * 4));
    T2_HIP(hipMemset(h->d_index_zero, 0, (size_t)h->max_symbols * 4));
    return 0;
}

extern "C" int t2gpu_fft_execute(t2gpu_ofdm *h, const float *in, float *out, int n_symbols)
{
    if (!h || !in || !out || n_symbols < 1 || n_symbols > h->max_symbols) { set_error("t2gpu_fft_execute: bad arguments"); return -1; }
    if (ensure_staging(h)) return -1;
    const size_t bytes = (size_t)n_symbols * h->m.fft_size * sizeof(float2);
    T2_HIP(hipMemcpy(h->d_in, in, bytes, hipMemcpyHostToDevice));
    if (t2gpu_fft_execute_dev(h, (const float *)h->d_in, (float *)h->d_out, n_symbols, nullptr)) return -1;
    T2_HIP(hipDeviceSynchronize());
    T2_HIP(hipMemcpy(out, h->d_out, bytes, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int t2gpu_eq_data_execute_dev(t2gpu_ofdm *h, const float *d_symbols, const int32_t *d_symbol_index, int n_symbols,
                                         float *d_cells, float *d_sync, void *stream)
{
    if (!h || !d_symbols || !d_symbol_index || !d_cells || n_symbols < 1 || n_symbols > h->max_symbols) {
        set_error("t2gpu_eq_data_execute_dev: bad arguments");
        return -1;
    }
    T2_HIP(launch_eq_data(h->eq, reinterpret_cast<const float2 *>(d_symbols), d_symbol_index, n_symbols, reinterpret_cast<float2 *>(d_cells),
                          h->d_pilot_scratch, reinterpret_cast<float2 *>(d_sync), (hipStream_t)stream));
    return h->m.c_data;
}

// ONE data symbol with its cells also stored to page-locked host memory by the equaliser's own workgroups, *h_flag = seq behind them
// (the slot-shaped path: no publishing launch). d_count: a zeroed device word of the caller's. Needs the output-range form (default).
extern "C" int t2gpu_eq_data_publish_dev(t2gpu_ofdm *h, const float *d_symbol, const int32_t *d_symbol_index, float *d_cells, float *h_cells,
                                         unsigned *h_flag, unsigned seq, unsigned *d_count, void *stream)
{
    if (!h || !d_symbol || !d_symbol_index || !d_cells || !h_cells || !h_flag || !d_count) { set_error("t2gpu_eq_data_publish_dev: bad arguments"); return -1; }
    if (h->eq.n_splits < 1) { set_error("t2gpu_eq_data_publish_dev: needs the output-range form of the equaliser"); return -1; }
    EqParams p = h->eq;
    p.pub_cells = reinterpret_cast<float2 *>(h_cells); p.pub_flag = h_flag; p.pub_seq = seq; p.pub_count = d_count;
    T2_HIP(launch_eq_data(p, reinterpret_cast<const float2 *>(d_symbol), d_symbol_index, 1, reinterpret_cast<float2 *>(d_cells), h->d_pilot_scratch,
                          nullptr, (hipStream_t)stream));
    return h->m.c_data;
}

// Data symbols of whole frames straight from the frames' spectra into the frames' cell streams (no gather / scatter copies)
extern "C" int t2gpu_eq_data_frames_dev(t2gpu_ofdm *h, const float *d_spectrum, int n_frames, int syms_per_frame, int first_symbol,
                                        int n_data_symbols, float *d_cells, long cells_frame_stride, long cells_offset, float *d_sync,
                                        void *stream)
{
    const long n = (long)n_frames * n_data_symbols;
    if (!h || !d_spectrum || !d_cells || n_frames < 1 || n_data_symbols < 1 || n > h->max_symbols || first_symbol < h->m.n_p2 ||
        first_symbol + n_data_symbols > syms_per_frame || cells_frame_stride < (long)n_data_symbols * h->m.c_data || cells_offset < 0) {
        set_error("t2gpu_eq_data_frames_dev: bad arguments");
        return -1;
    }
    EqParams p = h->eq;
    p.per_frame = n_data_symbols; p.first = first_symbol; p.in_syms_per_frame = syms_per_frame;
    p.out_frame_stride = cells_frame_stride; p.out_offset = cells_offset;
    T2_HIP(launch_eq_data(p, reinterpret_cast<const float2 *>(d_spectrum), nullptr, (int)n, reinterpret_cast<float2 *>(d_cells),
                          h->d_pilot_scratch, reinterpret_cast<float2 *>(d_sync), (hipStream_t)stream));
    return h->m.c_data;
}

// P2 symbols of a batch of frames (idx_symbol = 0 for all of them: n_p2 = 1 for 16K / 32K)
extern "C" int t2gpu_eq_p2_execute_dev(t2gpu_ofdm *h, const float *d_symbols, int n_symbols, float *d_cells, float *d_sync, void *stream)
{
    if (!h || !d_symbols || !d_cells || n_symbols < 1 || n_symbols > h->max_symbols) { set_error("t2gpu_eq_p2_execute_dev: bad arguments"); return -1; }
    hipStream_t s = (hipStream_t)stream;
    if (ensure_staging(h)) return -1;
    T2_HIP(launch_eq_data(h->eq_p2, reinterpret_cast<const float2 *>(d_symbols), h->d_index_zero, n_symbols, reinterpret_cast<float2 *>(d_cells),
                          h->d_pilot_scratch_p2, reinterpret_cast<float2 *>(d_sync), s));
    return h->m.c_p2;
}

// The P2 / frame-closing symbol of every frame read in place from the frames' spectra and written in place into the frames' cell
// streams (as t2gpu_eq_data_frames_dev does for the data symbols): no gather of the symbols, no copy of the cells.
namespace {
int eq_p2_frames(t2gpu_ofdm *h, const float *d_spectrum, int n_frames, int syms_per_frame, float *d_cells, long cells_frame_stride,
                 int skip_cells, float *d_l1_cells, float *d_sync, void *stream);
}
extern "C" int t2gpu_eq_p2_frames_dev(t2gpu_ofdm *h, const float *d_spectrum, int n_frames, int syms_per_frame, float *d_cells,
                                      long cells_frame_stride, int skip_cells, float *d_sync, void *stream)
{
    return eq_p2_frames(h, d_spectrum, n_frames, syms_per_frame, d_cells, cells_frame_stride, skip_cells, nullptr, d_sync, stream);
}
// the same, with the skipped cells (L1-pre + L1-post) of frame f stored at d_l1_cells + 2 * f * skip_cells
extern "C" int t2gpu_eq_p2_frames_l1_dev(t2gpu_ofdm *h, const float *d_spectrum, int n_frames, int syms_per_frame, float *d_cells,
                                         long cells_frame_stride, int skip_cells, float *d_l1_cells, float *d_sync, void *stream)
{
    return eq_p2_frames(h, d_spectrum, n_frames, syms_per_frame, d_cells, cells_frame_stride, skip_cells, d_l1_cells, d_sync, stream);
}
namespace {
int eq_p2_frames(t2gpu_ofdm *h, const float *d_spectrum, int n_frames, int syms_per_frame, float *d_cells, long cells_frame_stride,
                 int skip_cells, float *d_l1_cells, float *d_sync, void *stream)
{
    if (!h || !d_spectrum || !d_cells || n_frames < 1 || n_frames > h->max_symbols || syms_per_frame < 1 || skip_cells < 0 ||
        skip_cells > h->m.c_p2 || cells_frame_stride < h->m.c_p2 - skip_cells) {
        set_error("t2gpu_eq_p2_frames_dev: bad arguments");
        return -1;
    }
    EqParams p = h->eq_p2;
    p.per_frame = 1; p.first = 0; p.in_syms_per_frame = syms_per_frame;
    p.out_frame_stride = cells_frame_stride; p.out_offset = 0; p.out_skip = skip_cells;
    p.skip_out = reinterpret_cast<float2 *>(d_l1_cells);
    T2_HIP(launch_eq_data(p, reinterpret_cast<const float2 *>(d_spectrum), nullptr, n_frames, reinterpret_cast<float2 *>(d_cells),
                          h->d_pilot_scratch_p2, reinterpret_cast<float2 *>(d_sync), (hipStream_t)stream));
    return h->m.c_p2 - skip_cells;
}
}  // namespace

extern "C" int t2gpu_eq_fc_frames_dev(t2gpu_ofdm *h, const float *d_spectrum, int n_frames, int syms_per_frame, float *d_cells,
                                      long cells_frame_stride, long cells_offset, float *d_sync, void *stream)
{
    if (!h || !d_spectrum || !d_cells || n_frames < 1 || n_frames > h->max_symbols || syms_per_frame < h->m.len_frame || cells_offset < 0 ||
        cells_frame_stride < cells_offset + h->m.n_fc) {
        set_error("t2gpu_eq_fc_frames_dev: bad arguments");
        return -1;
    }
    if (!h->m.l_fc) { set_error("t2gpu_eq_fc_frames_dev: this mode has no frame-closing symbol"); return -1; }
    EqParams p = h->eq_fc;
    p.per_frame = 1; p.first = h->m.len_frame - 1; p.in_syms_per_frame = syms_per_frame;
    p.out_frame_stride = cells_frame_stride; p.out_offset = cells_offset;
    T2_HIP(launch_eq_data(p, reinterpret_cast<const float2 *>(d_spectrum), nullptr, n_frames, reinterpret_cast<float2 *>(d_cells),
                          h->d_pilot_scratch_fc, reinterpret_cast<float2 *>(d_sync), (hipStream_t)stream));
    return h->m.n_fc;
}

// frame-closing symbols of a batch of frames (idx_symbol = len_frame - 1 for all of them)
extern "C" int t2gpu_eq_fc_execute_dev(t2gpu_ofdm *h, const float *d_symbols, int n_symbols, float *d_cells, float *d_sync, void *stream)
{
    if (!h || !d_symbols || !d_cells || n_symbols < 1 || n_symbols > h->max_symbols) { set_error("t2gpu_eq_fc_execute_dev: bad arguments"); return -1; }
    if (!h->m.l_fc) { set_error("t2gpu_eq_fc_execute_dev: this mode has no frame-closing symbol"); return -1; }
    T2_HIP(launch_eq_data(h->eq_fc, reinterpret_cast<const float2 *>(d_symbols), h->d_index_fc, n_symbols, reinterpret_cast<float2 *>(d_cells),
                          h->d_pilot_scratch_fc, reinterpret_cast<float2 *>(d_sync), (hipStream_t)stream));
    return h->m.n_fc;
}

// phase_offset / sample_rate_offset of one symbol from its pilots alone, and the guard correlation of the buffered symbol, in one launch
// (ofdm_kernels.hip: sym_sync_kernel). kind: 0 data symbol idx_symbol, 1 P2, 2 frame closing.
extern "C" int t2gpu_sym_sync_dev(t2gpu_ofdm *h, int kind, int idx_symbol, const float *d_spectrum, const float *d_buffered, int guard,
                                  float *d_cp4, float *d_sync, float *h_small, unsigned *h_flag, unsigned seq, void *d_loop, void *stream)
{
    if (!h || !d_spectrum || kind < 0 || kind > 2 || (h_small && !h_flag) || (d_buffered && guard < 0)) { set_error("t2gpu_sym_sync_dev: bad arguments"); return -1; }
    const EqParams &p = kind == 0 ? h->eq : kind == 1 ? h->eq_p2 : h->eq_fc;
    if (kind == 1) idx_symbol = 0;
    else if (kind == 2) {
        if (!h->m.l_fc) { set_error("t2gpu_sym_sync_dev: this mode has no frame-closing symbol"); return -1; }
        idx_symbol = h->m.len_frame - 1;
    } else if (idx_symbol < h->m.n_p2 || idx_symbol >= h->m.n_p2 + h->rows) { set_error("t2gpu_sym_sync_dev: symbol index outside the frame's data symbols"); return -1; }
    T2_HIP(launch_sym_sync(p, reinterpret_cast<const float2 *>(d_spectrum), idx_symbol, reinterpret_cast<const float2 *>(d_buffered), guard,
                           reinterpret_cast<float4 *>(d_cp4), reinterpret_cast<float2 *>(d_sync), h_small, h_flag, seq, (hipStream_t)stream,
                           static_cast<T2DevLoop *>(d_loop)));
    return 0;
}

// symbol_acquisition's guard removal + fft->execute() (dvbt2_demodulator.cpp:332-334) for ONE buffered symbol with t2gpu_sym_sync_dev's work
// behind it: two launches where t2gpu_fft_execute_strided_dev + t2gpu_sym_sync_dev are three (the synchronisation floats are formed by the
// last workgroup of the FFT's second launch); pilot tables too large for that (P2, dense patterns) take the three. Same values either way.
extern "C" int t2gpu_fft_sym_sync_dev(t2gpu_ofdm *h, t2gpu_ofdm *tables, int kind, int idx_symbol, const float *d_buffered, int guard, int with_cp,
                                      float *d_spectrum, float *d_cp4, float *d_sync, float *h_small, unsigned *h_flag, unsigned seq, void *d_loop, void *stream)
{
    if (!h || !tables || !d_buffered || !d_spectrum || kind < 0 || kind > 2 || guard < 0 || (h_small && !h_flag) || tables->m.fft_size != h->m.fft_size) {
        set_error("t2gpu_fft_sym_sync_dev: bad arguments");
        return -1;
    }
    const EqParams &p = kind == 0 ? tables->eq : kind == 1 ? tables->eq_p2 : tables->eq_fc;
    int idx = idx_symbol;
    if (kind == 1) idx = 0;
    else if (kind == 2) {
        if (!tables->m.l_fc) { set_error("t2gpu_fft_sym_sync_dev: this mode has no frame-closing symbol"); return -1; }
        idx = tables->m.len_frame - 1;
    } else if (idx < tables->m.n_p2 || idx >= tables->m.n_p2 + tables->rows) { set_error("t2gpu_fft_sym_sync_dev: symbol index outside the frame's data symbols"); return -1; }
    const FftLayout lay{guard, 0, 1, h->m.fft_size + guard};
    const float2 *buffered = with_cp ? reinterpret_cast<const float2 *>(d_buffered) : nullptr;
    const hipError_t e = launch_fft_sym_sync(h->m.fft_size, reinterpret_cast<const float2 *>(d_buffered), reinterpret_cast<float2 *>(d_spectrum), h->d_twiddle, lay,
                                             h->d_fft_scratch, h->d_fft_count, p, idx, buffered, guard, reinterpret_cast<float4 *>(d_cp4),
                                             reinterpret_cast<float2 *>(d_sync), h_small, h_flag, seq, (hipStream_t)stream, static_cast<T2DevLoop *>(d_loop), h->one_launch);
    if (e == hipSuccess) return 0;
    if (e != hipErrorInvalidValue) { T2_HIP(e); }
    (void)hipGetLastError();
    if (t2gpu_fft_execute_strided_dev(h, d_buffered, guard, 0, 1, h->m.fft_size + guard, d_spectrum, 1, stream) != 0) return -1;
    return t2gpu_sym_sync_dev(tables, kind, idx_symbol, d_spectrum, with_cp ? d_buffered : nullptr, guard, d_cp4, d_sync, h_small, h_flag, seq, d_loop, stream);
}

int t2gpu_fft_one_args(t2gpu_ofdm *h, t2gpu_ofdm *tables, int kind, int idx_symbol, const float *d_buffered, int guard, int with_cp, float *d_spectrum,
                       float *h_small, unsigned *h_flag, unsigned seq, void *d_loop, t2gpu::FftOneArgs *out)
{
    if (!h || !tables || !d_buffered || !d_spectrum || !out || kind != 0 || guard < 0 || (h_small && !h_flag) || tables->m.fft_size != h->m.fft_size) {
        set_error("t2gpu_fft_one_args: bad arguments");
        return -1;
    }
    if (idx_symbol < tables->m.n_p2 || idx_symbol >= tables->m.n_p2 + tables->rows) { set_error("t2gpu_fft_one_args: symbol index outside the frame's data symbols"); return -1; }
    const EqParams &p = tables->eq;
    if ((h->m.fft_size != 32768 && h->m.fft_size != 16384) || (p.max_seg + 2) * 16 + 2 * 256 * 8 > FFT_ONE_LDS_FLOATS * 4 || !h->d_fft_scratch || !h->d_fft_count) return 1;
    out->in = reinterpret_cast<const float2 *>(d_buffered) + guard;
    out->scratch = h->d_fft_scratch; out->out = reinterpret_cast<float2 *>(d_spectrum); out->twiddle = h->d_twiddle; out->count = h->d_fft_count;
    out->p = p; out->idx_symbol = idx_symbol; out->buffered = with_cp ? reinterpret_cast<const float2 *>(d_buffered) : nullptr; out->guard = guard;
    out->cp_out = nullptr; out->sync = nullptr; out->h_small = h_small; out->h_flag = h_flag; out->seq = seq; out->loop = static_cast<T2DevLoop *>(d_loop);
    out->fft_size = h->m.fft_size;
    return 0;
}

extern "C" int t2gpu_ofdm_set_one_launch(t2gpu_ofdm *h, int on)
{
    if (!h) { set_error("t2gpu_ofdm_set_one_launch: bad arguments"); return -1; }
    h->one_launch = on != 0;
    return 0;
}

extern "C" int t2gpu_eq_data_execute(t2gpu_ofdm *h, int idx_symbol, const float *ofdm_cell, float *cells, float *sample_rate_offset,
                                     float *phase_offset)
{
    if (!h || !ofdm_cell || !cells || idx_symbol < h->m.n_p2 || idx_symbol >= h->m.n_p2 + h->rows) {
        set_error("t2gpu_eq_data_execute: bad arguments");
        return -1;
    }
    if (ensure_staging(h)) return -1;
    T2_HIP(hipMemcpy(h->d_in, ofdm_cell, (size_t)h->m.fft_size * sizeof(float2), hipMemcpyHostToDevice));
    T2_HIP(hipMemcpy(h->d_index, &idx_symbol, 4, hipMemcpyHostToDevice));
    if (t2gpu_eq_data_execute_dev(h, (const float *)h->d_in, h->d_index, 1, (float *)h->d_out, (float *)h->d_sync, nullptr) < 0) return -1;
    T2_HIP(hipDeviceSynchronize());
    T2_HIP(hipMemcpy(cells, h->d_out, (size_t)h->m.c_data * sizeof(float2), hipMemcpyDeviceToHost));
    float s[2];
    T2_HIP(hipMemcpy(s, h->d_sync, 8, hipMemcpyDeviceToHost));
    if (phase_offset) *phase_offset = s[0];
    if (sample_rate_offset) *sample_rate_offset = s[1];
    return h->m.c_data;
}
