// t2gpu_fec.cpp -- C-ABI of the FEC-side stages around the LDPC decoder (include/t2gpu.h): LLR demapper + bit
// de-interleaver, time / cell de-interleaver, BB descrambler (the reference's BCH stub). Host side only; the compute is
// in fec_kernels.hip and this library has no CPU fallback.
#include "../../include/t2gpu.h"
#include "fec_kernels.h"
#include "fec_tables.h"
#include "ldpc_graph.h"
#include "t2gpu_common.h"
#include <algorithm>
#include <cmath>
#include <map>
#include <mutex>
#include <vector>

using namespace t2gpu;

static bool have_device(int device, const char *who)
{
    int ndev = 0;
    hipError_t de = hipGetDeviceCount(&ndev);
    if (de != hipSuccess || device < 0 || device >= ndev) {
        set_error(std::string(who) + ": no usable HIP device (this library has no CPU path)");
        return false;
    }
    return hipSetDevice(device) == hipSuccess;
}

// ------------------------------------------------------------------------------------------------ tables (host only)
extern "C" int t2gpu_table_bitdeint(int mod, int fec_type, int code_rate, uint16_t *out)
{
    std::vector<uint16_t> a;
    if (!out || !bitdeint_address(mod, fec_type, code_rate, a)) { set_error("t2gpu_table_bitdeint: no such mode"); return -1; }
    std::copy(a.begin(), a.end(), out);
    return (int)a.size();
}
extern "C" int t2gpu_table_cell_deint(int num_blocks, int cells_per_fec, int32_t *out)
{
    if (!out || num_blocks < 1 || cells_per_fec < 1024 || cells_per_fec > 32768) { set_error("t2gpu_table_cell_deint: bad arguments"); return -1; }
    std::vector<int32_t> p;
    cell_deint_permutation(num_blocks, cells_per_fec, p);
    std::copy(p.begin(), p.end(), out);
    return (int)p.size();
}
extern "C" int t2gpu_table_bb_prbs(uint8_t *out, int n)
{
    if (!out || n < 1) return -1;
    std::vector<uint8_t> b;
    bb_prbs(b, n);
    std::copy(b.begin(), b.end(), out);
    return n;
}

// ------------------------------------------------------------------------------------------------ demapper
struct t2gpu_demap {
    DemapParams p{};
    int device = 0, max_cells = 0, stats_blocks = 0, partial_batches = 1;
    uint16_t *d_address = nullptr;
    double *d_partial = nullptr;
    float2 *d_terms = nullptr;         // the statistics' per-cell terms (|s|^2, |e|^2): scratch of the exact sequential sums
    long terms_pad_key = 0;            // TiTerms::pad_key: 0 whenever anything but launch_ti_blocks has written the scratch
    size_t terms_cells = 0;
    float *d_sums = nullptr;
    float *d_cells = nullptr;      // host-call staging
    int8_t *d_llr = nullptr;
};

extern "C" t2gpu_demap *t2gpu_demap_create(int mod, int fec_type, int code_rate, int rotation, int max_cells, int device)
{
    static const float ROT[4] = {0.506145483f, 0.293215314f, 0.150098316f, 0.062418810f};     // dvbt2_definition.h:45-48
    static const float NORM[4] = {0.707106781f, 0.316227766f, 0.15430335f, 0.076696499f};     // dvbt2_definition.h:49-52
    if (mod < 0 || mod > 3 || fec_type < 0 || fec_type > 1 || code_rate < 0 || code_rate > 5 || max_cells < 1) {
        set_error("t2gpu_demap_create: bad arguments");
        return nullptr;
    }
    if (!have_device(device, "t2gpu_demap_create")) return nullptr;
    t2gpu_demap *h = new t2gpu_demap();
    h->device = device; h->max_cells = max_cells;
    h->p.mod = mod;
    h->p.fec_size = fec_size_of(fec_type);
    h->p.bits_per_cell = 2 * (mod + 1);
    h->p.cells_per_fec = h->p.fec_size / h->p.bits_per_cell;
    h->p.rotate = rotation ? 1 : 0;
    h->p.rot_c = (float)std::cos(-(double)ROT[mod]);
    h->p.rot_s = (float)std::sin(-(double)ROT[mod]);
    h->p.d = NORM[mod];
    h->stats_blocks = 512;
    bool ok = true;
    if (mod > 0) {
        std::vector<uint16_t> addr;
        if (!bitdeint_address(mod, fec_type, code_rate, addr)) { delete h; set_error("no bit interleaver for this mode"); return nullptr; }
        ok = ok && hip_ok(hipMalloc(&h->d_address, addr.size() * 2), "hipMalloc");
        ok = ok && hip_ok(hipMemcpy(h->d_address, addr.data(), addr.size() * 2, hipMemcpyHostToDevice), "hipMemcpy");
        h->p.address = h->d_address;
    }
    ok = ok && hip_ok(hipMalloc(&h->d_partial, sizeof(double) * 2 * h->stats_blocks), "hipMalloc");
    ok = ok && hip_ok(hipMalloc(&h->d_sums, sizeof(float) * 4), "hipMalloc");
    if (!ok) { t2gpu_demap_destroy(h); return nullptr; }
    return h;
}

extern "C" int t2gpu_demap_configure(t2gpu_demap *h, int saturate)
{
    if (!h) return -1;
    h->p.saturate = saturate ? 1 : 0;
    return 0;
}

extern "C" void t2gpu_demap_destroy(t2gpu_demap *h)
{
    if (!h) return;
    t2_exit_mark("t2gpu_demap_destroy");
    if (h->d_llr) twin_retire_dev(h->d_llr, (size_t)h->max_cells * h->p.bits_per_cell);
    hipFree(h->d_address); hipFree(h->d_partial); hipFree(h->d_terms); hipFree(h->d_sums); hipFree(h->d_cells); hipFree(h->d_llr);
    delete h;
}

// scratch for `cells` term pairs (grown on demand; a growth synchronises the stream once)
static bool ensure_terms(t2gpu_demap *h, size_t cells, hipStream_t s)
{
    if (cells <= h->terms_cells) return true;
    if (!hip_ok(hipStreamSynchronize(s), "hipStreamSynchronize")) return false;
    float2 *p = nullptr;
    if (!hip_ok(hipMalloc(&p, cells * sizeof(float2)), "hipMalloc")) return false;
    if (!hip_ok(hipDeviceSynchronize(), "hipDeviceSynchronize")) { hipFree(p); return false; }
    hipFree(h->d_terms);
    h->d_terms = p; h->terms_cells = cells; h->terms_pad_key = 0;
    return true;
}

extern "C" int t2gpu_demap_execute_dev(t2gpu_demap *h, const float *d_cells, int n_cells, float precision_override,
                                       int8_t *d_llr, float *d_sums3, void *stream)
{
    if (!h || !d_cells || !d_llr || n_cells < 1 || n_cells > h->max_cells) { set_error("t2gpu_demap_execute_dev: bad arguments"); return -1; }
    hipStream_t s = (hipStream_t)stream;
    const float2 *cells = reinterpret_cast<const float2 *>(d_cells);
    float *sums = d_sums3 ? d_sums3 : h->d_sums;
    const int n_snr = h->p.mod == 0 ? std::min(n_cells, 2048) : n_cells;       // llr_demapper.cpp:184 (QPSK: first 2048 cells)
    int blocks = std::min(h->stats_blocks, (n_snr + 255) / 256);
    if (!ensure_terms(h, (size_t)demap_terms_padded(n_snr), s)) return -1;
    h->terms_pad_key = 0;
    T2_HIP(launch_demap_stats(h->p, cells, n_snr, h->d_partial, blocks, h->d_terms, sums, precision_override, s));
    const int n_frames = n_cells / h->p.cells_per_fec;
    if (n_frames > 0) T2_HIP(launch_demap_llr(h->p, cells, n_frames, sums, d_llr, s));
    return n_frames;
}

// The two passes of the demapper as separate enqueues (same kernels, same results): a scheduler can then put the statistics pass
// (no LDS to speak of) beside another stream's work and the LLR pass (one FEC frame staged in LDS) where a CU is free.
extern "C" int t2gpu_demap_stats_dev(t2gpu_demap *h, const float *d_cells, int n_cells, float precision_override, float *d_sums3,
                                     void *stream)
{
    if (!h || !d_cells || !d_sums3 || n_cells < 1 || n_cells > h->max_cells) { set_error("t2gpu_demap_stats_dev: bad arguments"); return -1; }
    const int n_snr = h->p.mod == 0 ? std::min(n_cells, 2048) : n_cells;
    const int blocks = std::min(h->stats_blocks, (n_snr + 255) / 256);
    if (!ensure_terms(h, (size_t)demap_terms_padded(n_snr), (hipStream_t)stream)) return -1;
    h->terms_pad_key = 0;
    T2_HIP(launch_demap_stats(h->p, reinterpret_cast<const float2 *>(d_cells), n_snr, h->d_partial, blocks, h->d_terms, d_sums3, precision_override,
                              (hipStream_t)stream));
    return 0;
}

// the statistics pass for n_blocks TI blocks of cells_per_block cells in one launch: block t at d_cells + 2 * t * cells_stride
// floats, its triple at d_sums + t * sums_stride
extern "C" int t2gpu_demap_stats_batch_dev(t2gpu_demap *h, const float *d_cells, long cells_stride, int n_blocks, int cells_per_block,
                                           float precision_override, float *d_sums, int sums_stride, void *stream)
{
    if (!h || !d_cells || !d_sums || n_blocks < 1 || cells_per_block < 1 || cells_per_block > h->max_cells || sums_stride < 3) {
        set_error("t2gpu_demap_stats_batch_dev: bad arguments");
        return -1;
    }
    T2_HIP(hipSetDevice(h->device));
    if (n_blocks > h->partial_batches) {
        T2_HIP(hipStreamSynchronize((hipStream_t)stream));
        double *p = nullptr;
        T2_HIP(hipMalloc(&p, sizeof(double) * 2 * h->stats_blocks * (size_t)n_blocks));
        T2_HIP(hipDeviceSynchronize());
        hipFree(h->d_partial);
        h->d_partial = p; h->partial_batches = n_blocks;
    }
    const int n_snr = h->p.mod == 0 ? std::min(cells_per_block, 2048) : cells_per_block;
    const int blocks = std::min(h->stats_blocks, (n_snr + 255) / 256);
    if (!ensure_terms(h, (size_t)demap_terms_padded(n_snr) * n_blocks, (hipStream_t)stream)) return -1;
    h->terms_pad_key = 0;
    T2_HIP(launch_demap_stats_batch(h->p, reinterpret_cast<const float2 *>(d_cells), cells_stride, n_snr, n_blocks, h->d_partial, blocks,
                                    h->d_terms, d_sums, sums_stride, precision_override, (hipStream_t)stream));
    return 0;
}

extern "C" int t2gpu_demap_llr_dev(t2gpu_demap *h, const float *d_cells, int n_cells, const float *d_sums3, int8_t *d_llr, void *stream)
{
    if (!h || !d_cells || !d_sums3 || !d_llr || n_cells < 1 || n_cells > h->max_cells) { set_error("t2gpu_demap_llr_dev: bad arguments"); return -1; }
    const int n_frames = n_cells / h->p.cells_per_fec;
    if (n_frames > 0) T2_HIP(launch_demap_llr(h->p, reinterpret_cast<const float2 *>(d_cells), n_frames, d_sums3, d_llr, (hipStream_t)stream));
    return n_frames;
}

// n_blocks TI blocks of cells_per_block cells each, back to back in d_cells, statistics of block t at d_sums + t * sums_stride
extern "C" int t2gpu_demap_llr_batch_dev(t2gpu_demap *h, const float *d_cells, int n_blocks, int cells_per_block, const float *d_sums,
                                         int sums_stride, int8_t *d_llr, void *stream)
{
    if (!h || !d_cells || !d_sums || !d_llr || n_blocks < 1 || cells_per_block < 1 || cells_per_block > h->max_cells ||
        cells_per_block % h->p.cells_per_fec || sums_stride < 3) {
        set_error("t2gpu_demap_llr_batch_dev: bad arguments");
        return -1;
    }
    const int per = cells_per_block / h->p.cells_per_fec;
    T2_HIP(launch_demap_llr(h->p, reinterpret_cast<const float2 *>(d_cells), n_blocks * per, d_sums, d_llr, (hipStream_t)stream, per, sums_stride));
    return n_blocks * per;
}

extern "C" int t2gpu_demap_execute(t2gpu_demap *h, const float *cells, int n_cells, int8_t *llr, float *sums3)
{
    if (!h || !cells || !llr || n_cells < 1 || n_cells > h->max_cells) { set_error("t2gpu_demap_execute: bad arguments"); return -1; }
    T2_HIP(hipSetDevice(h->device));
    if (!h->d_cells) {
        T2_HIP(hipMalloc(&h->d_cells, (size_t)h->max_cells * 8));
        T2_HIP(hipMalloc(&h->d_llr, (size_t)h->max_cells * h->p.bits_per_cell));
    }
    // a TI block straight from t2gpu_ti_push is still on the device (include/t2gpu.h, host-buffer hand-over). The demapper de-rotates
    // its input in place (llr_demapper.cpp:555-557): it works on its own copy, the twin stays what the host buffer holds.
    const float *twin = static_cast<const float *>(twin_lookup(cells, (size_t)n_cells * 8, h->device));
    twin_retire_dev(h->d_llr, (size_t)h->max_cells * h->p.bits_per_cell);       // the LLRs of the previous call are about to be overwritten
    // on the device's side stream (t2gpu_common.h): a TI block's worth of passes and 13 MB of LLRs coming down do not hold up whatever
    // the caller's other threads have on the null stream
    hipStream_t side = side_stream(h->device);
    if (!side) return -1;
    if (twin) T2_HIP(hipMemcpyAsync(h->d_cells, twin, (size_t)n_cells * 8, hipMemcpyDeviceToDevice, side));
    else T2_HIP(hipMemcpyAsync(h->d_cells, cells, (size_t)n_cells * 8, hipMemcpyHostToDevice, side));
    int nf = t2gpu_demap_execute_dev(h, h->d_cells, n_cells, 0.0f, h->d_llr, nullptr, side);
    if (nf < 0) return -1;
    T2_HIP(hipMemcpyAsync(llr, h->d_llr, (size_t)nf * h->p.fec_size, hipMemcpyDeviceToHost, side));
    if (sums3) T2_HIP(hipMemcpyAsync(sums3, h->d_sums, 12, hipMemcpyDeviceToHost, side));
    T2_HIP(hipStreamSynchronize(side));
    twin_publish(llr, h->d_llr, (size_t)nf * h->p.fec_size, h->device);
    return nf;
}

// ------------------------------------------------------------------------------------------------ time de-interleaver
struct t2gpu_ti {
    int device = 0, cells_per_fec = 0, num_blocks_max = 0, num_blocks = 0, pos = 0;
    TiParams p{};
    std::vector<int32_t> perm;         // host copy, for the per-geometry event order
    int32_t *d_perm = nullptr, *d_order = nullptr;
    uint8_t *d_lost = nullptr, *d_lost_blk = nullptr;
    float *d_first_q = nullptr;
    float *d_in = nullptr, *d_out = nullptr;   // host-call staging
    // t2gpu_ti_push_async: the complete block's copy down is on its way (side stream, behind `fixed`); t2gpu_ti_wait ends it
    hipEvent_t fixed = nullptr, down = nullptr;
    // d_out holds the image of this host buffer plus whatever an abandoned block (t2gpu_ti_begin in the middle of one: a block that does
    // not complete inside its T2 frame) scattered over it -- which is what the reference's A / B buffer holds then
    const float *res_host = nullptr;
    bool abandoned = false;
    hipStream_t blk_stream = nullptr;          // where the pushes of the block in the making put their launches (the cells' twin's home stream)
    float *down_out = nullptr;
    size_t down_bytes = 0;
    std::map<int, std::pair<std::vector<int32_t>, std::vector<uint8_t>>> geom;   // num_blocks -> (order, lost)
};

extern "C" t2gpu_ti *t2gpu_ti_create(int mod, int fec_type, int num_blocks_max, int device)
{
    if (mod < 0 || mod > 3 || fec_type < 0 || fec_type > 1 || num_blocks_max < 1 || num_blocks_max > 1023) {
        set_error("t2gpu_ti_create: bad arguments");
        return nullptr;
    }
    if (!have_device(device, "t2gpu_ti_create")) return nullptr;
    t2gpu_ti *h = new t2gpu_ti();
    h->device = device; h->num_blocks_max = num_blocks_max;
    h->cells_per_fec = fec_size_of(fec_type) / (2 * (mod + 1));
    cell_deint_permutation(num_blocks_max, h->cells_per_fec, h->perm);
    bool ok = hip_ok(hipMalloc(&h->d_perm, h->perm.size() * 4), "hipMalloc");
    ok = ok && hip_ok(hipMemcpy(h->d_perm, h->perm.data(), h->perm.size() * 4, hipMemcpyHostToDevice), "hipMemcpy");
    ok = ok && hip_ok(hipMalloc(&h->d_order, num_blocks_max * 4), "hipMalloc");
    ok = ok && hip_ok(hipMalloc(&h->d_lost, num_blocks_max), "hipMalloc");
    ok = ok && hip_ok(hipMalloc(&h->d_lost_blk, num_blocks_max), "hipMalloc");
    ok = ok && hip_ok(hipMalloc(&h->d_first_q, num_blocks_max * 4), "hipMalloc");
    if (!ok) { t2gpu_ti_destroy(h); return nullptr; }
    h->p.cells_per_fec = h->cells_per_fec;
    h->p.rows = h->cells_per_fec / 5;                 // N_split = 5 columns per FEC block (time_deinterleaver.cpp:69-113)
    h->p.perm = h->d_perm;
    return h;
}

extern "C" void t2gpu_ti_destroy(t2gpu_ti *h)
{
    if (!h) return;
    t2_exit_mark("t2gpu_ti_destroy");
    if (h->down_out) { hipEventSynchronize(h->down); h->down_out = nullptr; }  // a block's copy down still on its way: `out` is the caller's
    if (h->fixed) hipEventDestroy(h->fixed);
    if (h->down) hipEventDestroy(h->down);
    if (h->d_out) twin_retire_dev(h->d_out, (size_t)h->num_blocks_max * h->cells_per_fec * 8);
    hipFree(h->d_perm); hipFree(h->d_order); hipFree(h->d_lost); hipFree(h->d_lost_blk); hipFree(h->d_first_q); hipFree(h->d_in); hipFree(h->d_out);
    delete h;
}

extern "C" int t2gpu_ti_cells_per_fec(const t2gpu_ti *h) { return h ? h->cells_per_fec : -1; }

// l1_dyn_execute (time_deinterleaver.cpp:268-286): the TI block of this T2 frame holds num_blocks FEC blocks
extern "C" int t2gpu_ti_begin(t2gpu_ti *h, int num_blocks)
{
    if (!h || num_blocks < 1 || num_blocks > h->num_blocks_max) { set_error("t2gpu_ti_begin: bad arguments"); return -1; }
    T2_HIP(hipSetDevice(h->device));
    const bool loaded = h->num_blocks == num_blocks;       // the device tables of this geometry are already in place
    if (h->pos != 0) h->abandoned = true;
    h->num_blocks = num_blocks; h->pos = 0;
    h->p.cols = 5 * num_blocks;
    h->p.ti_block_size = h->p.cols * h->p.rows;
    if (loaded) return 0;
    auto it = h->geom.find(num_blocks);
    if (it == h->geom.end()) {
        // Order in which the first cells of the FEC blocks arrive, and which of their parked Q values the reference
        // loses: a parked value is stored when the next block-start arrives (unless that is block 0) or at the end of
        // the interleaver row; it is overwritten unsaved when block 0's start follows in the same row (:321-336).
        const int n = h->cells_per_fec, rows = h->p.rows, cols = h->p.cols;
        std::vector<int32_t> inv((size_t)num_blocks * n);
        for (int d = 0; d < num_blocks * n; ++d) inv[h->perm[d]] = d;
        std::vector<std::pair<int64_t, int>> ev;
        for (int b = 0; b < num_blocks; ++b) {
            const int d = inv[(size_t)b * n];
            const int col = d / rows, row = d % rows;
            ev.push_back({(int64_t)row * cols + col, b});
        }
        std::sort(ev.begin(), ev.end());
        std::vector<int32_t> order(num_blocks);
        std::vector<uint8_t> lost(num_blocks, 0);
        for (int k = 0; k < num_blocks; ++k) {
            order[k] = ev[k].second;
            if (k + 1 < num_blocks && ev[k + 1].second == 0 && ev[k].first / cols == ev[k + 1].first / cols) lost[k] = 1;
        }
        it = h->geom.emplace(num_blocks, std::make_pair(order, lost)).first;
    }
    T2_HIP(hipMemcpy(h->d_order, it->second.first.data(), num_blocks * 4, hipMemcpyHostToDevice));
    T2_HIP(hipMemcpy(h->d_lost, it->second.second.data(), num_blocks, hipMemcpyHostToDevice));
    std::vector<uint8_t> lost_by_block(num_blocks, 0);
    for (int k = 0; k < num_blocks; ++k) lost_by_block[it->second.first[k]] = it->second.second[k];
    T2_HIP(hipMemcpy(h->d_lost_blk, lost_by_block.data(), num_blocks, hipMemcpyHostToDevice));
    return 0;
}

// Complete TI blocks of the current geometry (t2gpu_ti_begin), n_blocks of them in one launch -- e.g. the same TI block of every
// T2 frame of a buffer. Equivalent to t2gpu_ti_begin + one t2gpu_ti_push_dev of ti_block_size cells per block.
extern "C" int t2gpu_ti_execute_blocks_dev(t2gpu_ti *h, const float *d_cells, long in_stride_cells, float *d_out, long out_stride_cells,
                                           int n_blocks, void *stream)
{
    if (!h || !d_cells || !d_out || n_blocks < 0 || !h->num_blocks || h->pos != 0) {
        set_error("t2gpu_ti_execute_blocks_dev: bad arguments (t2gpu_ti_begin first; no TI block may be half pushed)");
        return -1;
    }
    if (n_blocks == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = launch_ti_blocks(h->p, h->d_lost_blk, h->num_blocks, reinterpret_cast<const float2 *>(d_cells), in_stride_cells,
                                    reinterpret_cast<float2 *>(d_out), out_stride_cells, n_blocks, s);
    if (e == hipErrorInvalidValue) {            // FEC block larger than LDS: block by block through the scatter kernels
        for (int f = 0; f < n_blocks; ++f) {
            const int rc = t2gpu_ti_push_dev(h, d_cells + 2 * f * in_stride_cells, h->p.ti_block_size, d_out + 2 * f * out_stride_cells, stream);
            if (rc != 1) return -1;
        }
        return n_blocks;
    }
    T2_HIP(e);
    return n_blocks;
}

// One pass over the cells for the de-interleaver AND the demapper's statistics terms: the de-interleaver forms (|s|^2, |e|^2) of the
// cells it writes into dm's term planes (whole TI blocks in LDS-sized FEC blocks, exact sums); t2gpu_demap_stats_terms_dev then walks
// them (4 B per cell and sum). Returns 1 when the terms were formed, 0 when only the de-interleaving was done (tree sums selected, FEC
// block larger than LDS: t2gpu_demap_stats_batch_dev on d_out is the caller's next step then), -1 on error.
extern "C" int t2gpu_ti_execute_blocks_terms_dev(t2gpu_ti *h, t2gpu_demap *dm, const float *d_cells, long in_stride_cells, float *d_out,
                                                 long out_stride_cells, int n_blocks, void *stream)
{
    if (!h || !dm || dm->p.cells_per_fec != h->cells_per_fec) {
        set_error("t2gpu_ti_execute_blocks_terms_dev: bad arguments (de-interleaver and demapper of one modulation / FEC type)");
        return -1;
    }
    const int n_snr = dm->p.mod == 0 ? std::min(h->p.ti_block_size, 2048) : h->p.ti_block_size;
    const bool fused = demap_stats_exact_form() && d_cells && d_out && n_blocks >= 1 && h->num_blocks && h->pos == 0 &&
                       (size_t)h->p.cells_per_fec * 8 <= 150 * 1024 && h->p.ti_block_size <= dm->max_cells;
    if (!fused) return t2gpu_ti_execute_blocks_dev(h, d_cells, in_stride_cells, d_out, out_stride_cells, n_blocks, stream) < 0 ? -1 : 0;
    hipStream_t s = (hipStream_t)stream;
    if (!ensure_terms(dm, (size_t)demap_terms_padded(n_snr) * n_blocks, s)) return -1;
    const TiTerms tt{&dm->p, dm->d_terms, n_snr, &dm->terms_pad_key};
    T2_HIP(launch_ti_blocks(h->p, h->d_lost_blk, h->num_blocks, reinterpret_cast<const float2 *>(d_cells), in_stride_cells,
                            reinterpret_cast<float2 *>(d_out), out_stride_cells, n_blocks, s, &tt));
    return 1;
}

// the exact sums of n_blocks TI blocks from the terms t2gpu_ti_execute_blocks_terms_dev left in dm (same n_blocks, same TI block size)
extern "C" int t2gpu_demap_stats_terms_dev(t2gpu_demap *dm, int n_blocks, int cells_per_block, float precision_override, float *d_sums,
                                           int sums_stride, void *stream)
{
    if (!dm || !d_sums || n_blocks < 1 || cells_per_block < 1 || sums_stride < 3) { set_error("t2gpu_demap_stats_terms_dev: bad arguments"); return -1; }
    const int n_snr = dm->p.mod == 0 ? std::min(cells_per_block, 2048) : cells_per_block;
    if (!dm->d_terms || (size_t)demap_terms_padded(n_snr) * n_blocks > dm->terms_cells) {
        set_error("t2gpu_demap_stats_terms_dev: no terms of that size (t2gpu_ti_execute_blocks_terms_dev first)");
        return -1;
    }
    T2_HIP(launch_demap_stats_from_terms(dm->p, n_snr, n_blocks, dm->d_terms, d_sums, sums_stride, precision_override, (hipStream_t)stream));
    return 0;
}

// t2gpu_ti_execute_blocks_dev followed by t2gpu_demap_stats_batch_dev on the de-interleaved blocks, as one call (one pass over the
// cells where t2gpu_ti_execute_blocks_terms_dev applies)
extern "C" int t2gpu_ti_execute_blocks_stats_dev(t2gpu_ti *h, t2gpu_demap *dm, const float *d_cells, long in_stride_cells, float *d_out,
                                                 long out_stride_cells, int n_blocks, float precision_override, float *d_sums, int sums_stride,
                                                 void *stream)
{
    if (!h || !dm || dm->p.cells_per_fec != h->cells_per_fec) {
        set_error("t2gpu_ti_execute_blocks_stats_dev: bad arguments (de-interleaver and demapper of one modulation / FEC type)");
        return -1;
    }
    const int rc = t2gpu_ti_execute_blocks_terms_dev(h, dm, d_cells, in_stride_cells, d_out, out_stride_cells, n_blocks, stream);
    if (rc < 0) return rc;
    if (rc == 1) return t2gpu_demap_stats_terms_dev(dm, n_blocks, h->p.ti_block_size, precision_override, d_sums, sums_stride, stream) == 0 ? n_blocks : -1;
    return t2gpu_demap_stats_batch_dev(dm, d_out, out_stride_cells, n_blocks, h->p.ti_block_size, precision_override, d_sums, sums_stride, stream) == 0
               ? n_blocks : -1;
}

extern "C" int t2gpu_ti_push_dev(t2gpu_ti *h, const float *d_cells, int n_cells, float *d_out, void *stream)
{
    if (!h || !d_cells || !d_out || n_cells < 0 || !h->num_blocks || h->pos + n_cells > h->p.ti_block_size) {
        set_error("t2gpu_ti_push_dev: bad arguments (t2gpu_ti_begin first; a push may not cross the TI block end)");
        return -1;
    }
    hipStream_t s = (hipStream_t)stream;
    if (n_cells)
        T2_HIP(launch_ti_scatter(h->p, reinterpret_cast<const float2 *>(d_cells), h->pos, n_cells, reinterpret_cast<float2 *>(d_out),
                                 h->d_first_q, s));
    h->pos += n_cells;
    if (h->pos == h->p.ti_block_size) {
        T2_HIP(launch_ti_fixup(h->p, h->d_order, h->d_lost, h->num_blocks, h->d_first_q, reinterpret_cast<float2 *>(d_out), s));
        h->pos = 0;
        return 1;
    }
    return 0;
}

// ends the copy down a t2gpu_ti_push_async left on its way: `out` then holds the block and has its twin
extern "C" int t2gpu_ti_wait(t2gpu_ti *h)
{
    if (!h) { set_error("t2gpu_ti_wait: bad arguments"); return -1; }
    if (!h->down_out) return 0;
    T2_HIP(hipSetDevice(h->device));
    float *out = h->down_out;
    h->down_out = nullptr;
    T2_HIP(hipEventSynchronize(h->down));
    twin_publish(out, h->d_out, h->down_bytes, h->device);                      // the complete TI block: what llr_demapper is handed next
    return 0;
}

extern "C" int t2gpu_ti_push_async(t2gpu_ti *h, const float *cells, int n_cells, float *out)
{
    if (!h || !cells || !out || !h->num_blocks) { set_error("t2gpu_ti_push: bad arguments"); return -1; }
    if (n_cells < 0 || (long)h->pos + n_cells > (long)h->p.ti_block_size) {      // before anything is copied into the staging buffer
        set_error("t2gpu_ti_push: n_cells runs past the TI block");
        return -1;
    }
    if (n_cells == 0) return 0;                                                 // (nothing arrives, nothing starts: a block begins with its first cell)
    T2_HIP(hipSetDevice(h->device));
    if (h->down_out && t2gpu_ti_wait(h) != 0) return -1;                        // (a caller that pushes on without having waited)
    const size_t cap = (size_t)h->num_blocks_max * h->cells_per_fec * 8;
    if (!h->d_in) { T2_HIP(hipMalloc(&h->d_in, cap)); T2_HIP(hipMalloc(&h->d_out, cap)); }
    const size_t blk = (size_t)h->p.ti_block_size * 8;
    // The TI block lives in the handle between the pushes of a block: the caller's buffer (the reference's A/B buffer, whose cells the
    // scatter does not reach keep their history) goes up once, with the block's first push, and comes back once, when the block is
    // complete -- not with every OFDM symbol (60 round trips of 13 MB per 32K frame in rounds 1-3: 0.9 ms per symbol, three quarters of
    // the slot-shaped path's time). Between those two moments `out` is not touched.
    // (`out` handed back unmodified from the previous block of this handle: its twin IS d_out, nothing to bring up)
    // cells straight from an equaliser's output (t2gpu_demod / t2gpu_eq_*_execute): they are still on the device. The scatter reads them
    // there after this call has returned: it goes to the stream their producer writes them on (the twin's home stream), so that the
    // producer's next pass over that buffer comes behind it. A block whose pushes change stream half way waits for what it has launched.
    hipStream_t s = nullptr;
    const float *d_cells = static_cast<const float *>(twin_lookup(cells, (size_t)n_cells * 8, h->device, &s));
    if (!d_cells) s = nullptr;
    if (h->pos != 0 && s != h->blk_stream) T2_HIP(hipStreamSynchronize(h->blk_stream));
    h->blk_stream = s;
    if (h->pos == 0) {
        // (a block abandoned half way -- CFG-A's 878 cells behind the frame's TI block start one that the next frame's start discards --
        // leaves d_out as the reference's buffer is left: the old block with those cells scattered over it. It is not brought up again.)
        const bool resident = twin_lookup(out, blk, h->device) == h->d_out || (h->abandoned && out == h->res_host && handoff_on());
        if (!resident) T2_HIP(hipMemcpyAsync(h->d_out, out, blk, hipMemcpyHostToDevice, s));
        twin_retire_dev(h->d_out, cap);                                         // from here on d_out is a block in the making, nobody's twin
        h->res_host = out; h->abandoned = false;
    }
    if (!d_cells) {
        T2_HIP(hipMemcpyAsync(h->d_in, cells, (size_t)n_cells * 8, hipMemcpyHostToDevice, s));
        d_cells = h->d_in;
    }
    int done = t2gpu_ti_push_dev(h, d_cells, n_cells, h->d_out, s);
    if (done < 0) return -1;
    if (done == 1) {
        // the block comes down on the device's side stream, behind the scatter / fix-up launches: 13 MB at the link's rate (235 us)
        // that the pushes' stream -- the caller's next symbols -- does not wait for
        hipStream_t side = side_stream(h->device);
        if (!side) return -1;
        if (!h->fixed) T2_HIP(hipEventCreateWithFlags(&h->fixed, hipEventDisableTiming));
        if (!h->down) T2_HIP(hipEventCreateWithFlags(&h->down, hipEventDisableTiming));
        T2_HIP(hipEventRecord(h->fixed, s));
        T2_HIP(hipStreamWaitEvent(side, h->fixed, 0));
        T2_HIP(hipMemcpyAsync(out, h->d_out, blk, hipMemcpyDeviceToHost, side));
        T2_HIP(hipEventRecord(h->down, side));
        h->down_out = out; h->down_bytes = blk;
    }
    if (d_cells == h->d_in) T2_HIP(hipStreamSynchronize(s));                    // `cells` is the caller's again
    return done;
}

extern "C" int t2gpu_ti_push(t2gpu_ti *h, const float *cells, int n_cells, float *out)
{
    const int done = t2gpu_ti_push_async(h, cells, n_cells, out);
    if (done == 1 && t2gpu_ti_wait(h) != 0) return -1;
    return done;
}

// ------------------------------------------------------------------------------------------------ BB descrambler / BCH stub
static std::mutex g_prbs_mutex;
static std::map<int, uint8_t *> g_prbs;      // per device

// bit-per-byte sequence [54000] followed by the same sequence packed MSB first [6750]
static const uint8_t *prbs_on_device(int device)
{
    std::lock_guard<std::mutex> lk(g_prbs_mutex);
    auto it = g_prbs.find(device);
    if (it != g_prbs.end()) return it->second;
    std::vector<uint8_t> bits;
    bb_prbs(bits, 54000);
    bits.resize(54000 + 6750, 0);
    for (int j = 0; j < 6750; ++j)
        for (int n = 0; n < 8; ++n) bits[54000 + j] |= (uint8_t)((bits[8 * j + n] & 1) << (7 - n));
    uint8_t *d = nullptr;
    if (!hip_ok(hipMalloc(&d, bits.size()), "hipMalloc") || !hip_ok(hipMemcpy(d, bits.data(), bits.size(), hipMemcpyHostToDevice), "hipMemcpy"))
        return nullptr;
    g_prbs[device] = d;
    return d;
}

extern "C" int t2gpu_bch_descramble_dev(int fec_type, int code_rate, const uint8_t *d_bits, int n_frames, uint8_t *d_out,
                                        void *stream)
{
    const int id = ldpc_code_id(fec_type, code_rate);
    if (id < 0 || !d_bits || !d_out || n_frames < 1) { set_error("t2gpu_bch_descramble_dev: bad arguments"); return -1; }
    int device = 0;
    T2_HIP(hipGetDevice(&device));
    const uint8_t *prbs = prbs_on_device(device);
    if (!prbs) return -1;
    static const int k_ldpc[12] = {7200, 9720, 10800, 11880, 12600, 13320, 32400, 38880, 43200, 48600, 51840, 54000};
    T2_HIP(launch_bch_descramble(d_bits, n_frames, k_ldpc[id], ldpc_k_bch(id), prbs, d_out, (hipStream_t)stream));
    return ldpc_k_bch(id);
}

extern "C" int t2gpu_bch_descramble_pack_dev(int fec_type, int code_rate, const uint8_t *d_bits, int n_frames, uint8_t *d_bytes,
                                             void *stream)
{
    const int id = ldpc_code_id(fec_type, code_rate);
    if (id < 0 || !d_bits || !d_bytes || n_frames < 1) { set_error("t2gpu_bch_descramble_pack_dev: bad arguments"); return -1; }
    int device = 0;
    T2_HIP(hipGetDevice(&device));
    const uint8_t *prbs = prbs_on_device(device);
    if (!prbs) return -1;
    static const int k_ldpc[12] = {7200, 9720, 10800, 11880, 12600, 13320, 32400, 38880, 43200, 48600, 51840, 54000};
    T2_HIP(launch_bch_descramble_pack(d_bits, n_frames, k_ldpc[id], ldpc_k_bch(id), prbs + 54000, d_bytes, (hipStream_t)stream));
    return ldpc_k_bch(id) / 8;
}

extern "C" int t2gpu_bch_descramble(int fec_type, int code_rate, const uint8_t *bits, int n_frames, uint8_t *out)
{
    const int id = ldpc_code_id(fec_type, code_rate);
    if (id < 0 || !bits || !out || n_frames < 1) { set_error("t2gpu_bch_descramble: bad arguments"); return -1; }
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess || !have_device(device, "t2gpu_bch_descramble")) return -1;
    static const int k_ldpc[12] = {7200, 9720, 10800, 11880, 12600, 13320, 32400, 38880, 43200, 48600, 51840, 54000};
    const size_t in_b = (size_t)n_frames * k_ldpc[id], out_b = (size_t)n_frames * ldpc_k_bch(id);
    // staging kept per device and grown on demand (rounds 1-3 allocated and freed it in every call)
    struct Stage { uint8_t *in = nullptr, *out = nullptr; size_t in_cap = 0, out_cap = 0; };
    static std::mutex m;
    static std::map<int, Stage> stages;
    std::lock_guard<std::mutex> lk(m);
    Stage &st = stages[device];
    if (st.in_cap < in_b) { hipFree(st.in); st.in = nullptr; st.in_cap = 0; T2_HIP(hipMalloc(&st.in, in_b)); st.in_cap = in_b; }
    if (st.out_cap < out_b) { hipFree(st.out); st.out = nullptr; st.out_cap = 0; T2_HIP(hipMalloc(&st.out, out_b)); st.out_cap = out_b; }
    // bits straight from t2gpu_ldpc_collect / t2gpu_ldpc_execute are still on the device
    hipStream_t side = side_stream(device);
    if (!side) return -1;
    const uint8_t *d_in = static_cast<const uint8_t *>(twin_lookup(bits, in_b, device));
    if (!d_in) { T2_HIP(hipMemcpyAsync(st.in, bits, in_b, hipMemcpyHostToDevice, side)); d_in = st.in; }
    const int kb = t2gpu_bch_descramble_dev(fec_type, code_rate, d_in, n_frames, st.out, side);
    if (kb < 0) return -1;
    T2_HIP(hipMemcpyAsync(out, st.out, out_b, hipMemcpyDeviceToHost, side));
    T2_HIP(hipStreamSynchronize(side));
    return kb;
}
