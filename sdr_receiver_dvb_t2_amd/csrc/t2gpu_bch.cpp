// t2gpu_bch.cpp -- C ABI of the outer-code (BCH) stage: t2gpu_bch_decode[_dev]. Opt-in: the reference's bch_decoder::execute
// does not decode (bch_decoder.cpp:136), so the drop-in path keeps calling t2gpu_bch_descramble on the LDPC output as it is;
// a caller that wants the outer code applied calls this on the same buffer first.
#include "../../include/t2gpu.h"
#include "bch_kernels.h"
#include "bch_tables.h"
#include "ldpc_graph.h"
#include "t2gpu_common.h"

#include <map>
#include <mutex>
#include <tuple>
#include <vector>

using namespace t2gpu;

namespace {

std::mutex g_mutex;
std::map<std::tuple<int, int, int>, BchDev> g_tables;      // (device, m, t) -> device tables, kept for the life of the process

template <class V>
bool upload(const V &v, const uint16_t **out)
{
    void *d = nullptr;
    const size_t bytes = v.size() * sizeof(v[0]);
    if (!hip_ok(hipMalloc(&d, bytes), "hipMalloc") || !hip_ok(hipMemcpy(d, v.data(), bytes, hipMemcpyHostToDevice), "hipMemcpy"))
        return false;
    *out = static_cast<const uint16_t *>(d);
    return true;
}

const BchDev *tables_on_device(int device, int m, int t)
{
    std::lock_guard<std::mutex> lk(g_mutex);
    const auto key = std::make_tuple(device, m, t);
    auto it = g_tables.find(key);
    if (it != g_tables.end()) return &it->second;
    BchTables tb;
    if (!bch_build_tables(m, t, tb)) { set_error("t2gpu_bch_decode: field tables failed to build"); return nullptr; }
    BchDev d{};
    d.m = m; d.t = t;
    if (!upload(tb.exp, &d.exp) || !upload(tb.log, &d.log) || !upload(tb.rem, &d.rem) || !upload(tb.basis, &d.basis)) return nullptr;
    return &(g_tables[key] = d);
}

const int kLdpc[12] = {7200, 9720, 10800, 11880, 12600, 13320, 32400, 38880, 43200, 48600, 51840, 54000};

}  // namespace

extern "C" int t2gpu_bch_info(int fec_type, int code_rate, int *m, int *t, int *k_bch, int *n_bch)
{
    const int id = ldpc_code_id(fec_type, code_rate);
    if (id < 0) { set_error("t2gpu_bch_info: no such code"); return -1; }
    const int mm = id < 6 ? 14 : 16;
    if (m) *m = mm;
    if (t) *t = (kLdpc[id] - ldpc_k_bch(id)) / mm;
    if (k_bch) *k_bch = ldpc_k_bch(id);
    if (n_bch) *n_bch = kLdpc[id];
    return 0;
}

extern "C" int t2gpu_table_bch_minpoly(int m, int t, uint32_t *out)
{
    BchTables tb;
    if (!out || !bch_build_tables(m, t, tb)) { set_error("t2gpu_table_bch_minpoly: bad arguments"); return -1; }
    for (int i = 0; i < t; ++i) out[i] = tb.minpoly[i];
    return t;
}

extern "C" int t2gpu_bch_decode_dev(int fec_type, int code_rate, uint8_t *d_bits, int n_frames, int32_t *d_status, void *stream)
{
    int m = 0, t = 0, n = 0;
    if (t2gpu_bch_info(fec_type, code_rate, &m, &t, nullptr, &n) != 0 || !d_bits || !d_status || n_frames < 1 ||
        (reinterpret_cast<uintptr_t>(d_bits) & 7u)) {
        set_error("t2gpu_bch_decode_dev: bad arguments (the bit buffer must be 8-byte aligned)");
        return -1;
    }
    int device = 0;
    T2_HIP(hipGetDevice(&device));
    const BchDev *tb = tables_on_device(device, m, t);
    if (!tb) return -1;
    BchDev p = *tb;
    p.n_bits = n;
    T2_HIP(launch_bch_decode(d_bits, n_frames, p, d_status, (hipStream_t)stream));
    return n_frames;
}

extern "C" int t2gpu_bch_decode(int fec_type, int code_rate, uint8_t *bits, int n_frames, int32_t *status)
{
    int n = 0;
    if (t2gpu_bch_info(fec_type, code_rate, nullptr, nullptr, nullptr, &n) != 0 || !bits || !status || n_frames < 1) {
        set_error("t2gpu_bch_decode: bad arguments");
        return -1;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        set_error("t2gpu_bch_decode: no usable HIP device (this library has no CPU path)");
        return -1;
    }
    uint8_t *d_bits = nullptr;
    int32_t *d_status = nullptr;
    const size_t bytes = (size_t)n_frames * n;
    int device = 0;
    T2_HIP(hipGetDevice(&device));
    if (void *twin = const_cast<void *>(t2gpu::twin_lookup(bits, bytes, device))) {
        // bits straight from the LDPC stage are still on the device (t2gpu.h, host-buffer hand-over): corrected there in place, so that
        // the twin goes on holding what the host buffer holds, and copied back
        T2_HIP(hipMalloc(&d_status, (size_t)n_frames * sizeof(int32_t)));
        int rc = -1;
        if (t2gpu_bch_decode_dev(fec_type, code_rate, static_cast<uint8_t *>(twin), n_frames, d_status, nullptr) == n_frames &&
            t2gpu::hip_ok(hipMemcpy(bits, twin, bytes, hipMemcpyDeviceToHost), "hipMemcpy") &&
            t2gpu::hip_ok(hipMemcpy(status, d_status, (size_t)n_frames * sizeof(int32_t), hipMemcpyDeviceToHost), "hipMemcpy"))
            rc = n_frames;
        hipFree(d_status);
        return rc;
    }
    T2_HIP(hipMalloc(&d_bits, bytes));
    if (!hip_ok(hipMalloc(&d_status, (size_t)n_frames * sizeof(int32_t)), "hipMalloc")) { hipFree(d_bits); return -1; }
    int rc = -1;
    if (hip_ok(hipMemcpy(d_bits, bits, bytes, hipMemcpyHostToDevice), "hipMemcpy") &&
        t2gpu_bch_decode_dev(fec_type, code_rate, d_bits, n_frames, d_status, nullptr) == n_frames &&
        hip_ok(hipDeviceSynchronize(), "hipDeviceSynchronize") &&
        hip_ok(hipMemcpy(bits, d_bits, bytes, hipMemcpyDeviceToHost), "hipMemcpy") &&
        hip_ok(hipMemcpy(status, d_status, (size_t)n_frames * sizeof(int32_t), hipMemcpyDeviceToHost), "hipMemcpy"))
        rc = n_frames;
    hipFree(d_bits); hipFree(d_status);
    return rc;
}
