#include "t2gpu_common.h"
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>
#include <vector>
#include "../../include/t2gpu.h"

namespace t2gpu {
static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
bool hip_ok(hipError_t e, const char *what)
{
    if (e == hipSuccess) return true;
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return false;
}
const std::string &last_error() { return g_err; }

hipError_t ensure_dynamic_lds(const void *fn, int bytes)
{
    static std::mutex m;
    static std::map<std::pair<int, const void *>, int> have;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(m);
    int &cur = have[std::make_pair(dev, fn)];
    if (bytes <= cur) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) cur = bytes;
    return e;
}

namespace {
struct Twin { const char *host; const char *dev; size_t bytes; int device; bool owned; bool pinned; };
std::mutex g_twin_m;
std::vector<Twin> g_twins;
bool twins_on()
{
    static const bool on = [] { const char *e = std::getenv("T2GPU_HANDOFF"); return !(e && std::atoi(e) == 0); }();
    return on;
}
}  // namespace

void twin_publish(const void *host, const void *dev, size_t bytes, int device)
{
    if (!host || !dev || !bytes) return;
    std::lock_guard<std::mutex> lk(g_twin_m);
    for (Twin &t : g_twins)
        if (t.host == host && !t.owned) { t.dev = static_cast<const char *>(dev); t.bytes = bytes; t.device = device; return; }
    g_twins.push_back(Twin{static_cast<const char *>(host), static_cast<const char *>(dev), bytes, device, false, false});
}
void twin_retire(const void *host)
{
    std::lock_guard<std::mutex> lk(g_twin_m);
    for (size_t i = 0; i < g_twins.size(); ++i)
        if (g_twins[i].host == host && !g_twins[i].owned) { g_twins.erase(g_twins.begin() + (long)i); return; }
}
void twin_retire_dev(const void *dev_lo, size_t bytes)
{
    const char *lo = static_cast<const char *>(dev_lo);
    std::lock_guard<std::mutex> lk(g_twin_m);
    for (size_t i = 0; i < g_twins.size();) {
        if (!g_twins[i].owned && g_twins[i].dev >= lo && g_twins[i].dev < lo + bytes) g_twins.erase(g_twins.begin() + (long)i);
        else ++i;
    }
}
const void *twin_lookup(const void *host, size_t bytes, int device)
{
    if (!host || !twins_on()) return nullptr;
    const char *h = static_cast<const char *>(host);
    std::lock_guard<std::mutex> lk(g_twin_m);
    for (const Twin &t : g_twins)
        if (t.device == device && h >= t.host && h + bytes <= t.host + t.bytes) return t.dev + (h - t.host);
    return nullptr;
}
}  // namespace t2gpu

// ---- caller-owned host buffers with a twin the library keeps for them (include/t2gpu.h)
extern "C" int t2gpu_twin_attach(void *host, size_t bytes, int device)
{
    using namespace t2gpu;
    if (!host || !bytes) { set_error("t2gpu_twin_attach: bad arguments"); return -1; }
    T2_HIP(hipSetDevice(device));
    void *dev = nullptr;
    T2_HIP(hipMalloc(&dev, bytes));
    const bool pinned = hipHostRegister(host, bytes, hipHostRegisterDefault) == hipSuccess;   // faster copies; not essential
    if (!pinned) (void)hipGetLastError();
    std::lock_guard<std::mutex> lk(g_twin_m);
    g_twins.push_back(Twin{static_cast<const char *>(host), static_cast<const char *>(dev), bytes, device, true, pinned});
    return 0;
}
extern "C" int t2gpu_twin_detach(void *host)
{
    using namespace t2gpu;
    Twin t{};
    bool found = false;
    {
        std::lock_guard<std::mutex> lk(g_twin_m);
        for (size_t i = 0; i < g_twins.size(); ++i)
            if (g_twins[i].host == host && g_twins[i].owned) { t = g_twins[i]; g_twins.erase(g_twins.begin() + (long)i); found = true; break; }
    }
    if (!found) return 0;
    hipSetDevice(t.device);
    hipDeviceSynchronize();
    if (t.pinned) hipHostUnregister(host);
    hipFree(const_cast<char *>(t.dev));
    return 0;
}
// page-lock a caller's buffer (copies to and from it then run at the link's rate and asynchronously); not an error when it cannot be
extern "C" int t2gpu_host_pin(void *host, size_t bytes)
{
    if (!host || !bytes) return -1;
    if (hipHostRegister(host, bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return 1; }
    return 0;
}
extern "C" int t2gpu_host_unpin(void *host)
{
    if (!host) return -1;
    if (hipHostUnregister(host) != hipSuccess) (void)hipGetLastError();
    return 0;
}

// memcpy(dst, src, bytes) on the host and the same bytes between the twins: dst (inside an attached buffer) then holds on the device
// what it holds on the host -- from src's twin when it has one, else from the host bytes.
extern "C" int t2gpu_twin_copy(void *dst, const void *src, size_t bytes, int device)
{
    using namespace t2gpu;
    if (!dst || !src) { set_error("t2gpu_twin_copy: bad arguments"); return -1; }
    std::memcpy(dst, src, bytes);
    const void *d_dst = twin_lookup(dst, bytes, device);
    if (!d_dst) return 0;
    T2_HIP(hipSetDevice(device));
    const void *d_src = twin_lookup(src, bytes, device);
    if (d_src) T2_HIP(hipMemcpyAsync(const_cast<void *>(d_dst), d_src, bytes, hipMemcpyDeviceToDevice, nullptr));
    else T2_HIP(hipMemcpyAsync(const_cast<void *>(d_dst), dst, bytes, hipMemcpyHostToDevice, nullptr));
    return 0;
}

extern "C" int t2gpu_version(void) { return 100; }
extern "C" const char *t2gpu_last_error(void)
{
    return t2gpu::last_error().c_str();
}
extern "C" int t2gpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
