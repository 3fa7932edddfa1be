#include "t2gpu_common.h"
#include <atomic>
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

hipStream_t side_stream(int device)
{
    static std::mutex m;
    static hipStream_t streams[64] = {};
    if (device < 0 || device >= 64) { set_error("t2gpu: device index outside the side-stream table"); return nullptr; }
    std::lock_guard<std::mutex> lk(m);
    if (!streams[device]) {
        int cur = 0;
        if (!hip_ok(hipGetDevice(&cur), "hipGetDevice")) return nullptr;
        if (cur != device && !hip_ok(hipSetDevice(device), "hipSetDevice")) return nullptr;
        int least = 0, greatest = 0;
        const bool ok = hip_ok(hipDeviceGetStreamPriorityRange(&least, &greatest), "hipDeviceGetStreamPriorityRange") &&
                        hip_ok(hipStreamCreateWithPriority(&streams[device], hipStreamNonBlocking, greatest), "hipStreamCreateWithPriority");
        if (cur != device) hipSetDevice(cur);
        if (!ok) { streams[device] = nullptr; return nullptr; }
    }
    return streams[device];
}

namespace {
// guard: what the first and the last 64 bytes of the host range and its length hashed to when the entry was made (0 = not guarded:
// page-locked buffers of the library's own that the DEVICE fills behind a sequence word)
struct Twin { const char *host; const char *dev; size_t bytes; int device; bool owned; bool pinned; uint64_t guard; hipStream_t home; };
std::mutex g_twin_m;
std::vector<Twin> g_twins;
std::atomic<int> g_handoff{0};           // t2gpu_handoff_enable: implicit entries are consulted only when the caller has said its buffers travel unmodified

uint64_t guard_of(const char *host, size_t bytes)
{
    uint64_t hsh = 1469598103934665603ull ^ (uint64_t)bytes;
    auto mix = [&](const char *q, size_t n) { for (size_t i = 0; i < n; ++i) { hsh ^= (unsigned char)q[i]; hsh *= 1099511628211ull; } };
    const size_t head = bytes < 64 ? bytes : 64;
    mix(host, head);
    if (bytes > 64) { const size_t tail = bytes - 64 < 64 ? bytes - 64 : 64; mix(host + bytes - tail, tail); }
    return hsh | 1ull;                   // never 0
}
}  // namespace

bool handoff_on() { return g_handoff.load(std::memory_order_relaxed) != 0; }

void twin_publish(const void *host, const void *dev, size_t bytes, int device, bool guarded, hipStream_t home)
{
    if (!host || !dev || !bytes) return;
    const uint64_t g = guarded ? guard_of(static_cast<const char *>(host), bytes) : 0;
    std::lock_guard<std::mutex> lk(g_twin_m);
    for (Twin &t : g_twins)
        if (t.host == host && !t.owned) { t.dev = static_cast<const char *>(dev); t.bytes = bytes; t.device = device; t.guard = g; t.home = home; return; }
    g_twins.push_back(Twin{static_cast<const char *>(host), static_cast<const char *>(dev), bytes, device, false, false, g, home});
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
const void *twin_lookup(const void *host, size_t bytes, int device, hipStream_t *home)
{
    if (home) *home = nullptr;
    if (!host) return nullptr;
    const bool implicit_ok = g_handoff.load(std::memory_order_relaxed) != 0;
    const char *h = static_cast<const char *>(host);
    std::lock_guard<std::mutex> lk(g_twin_m);
    for (size_t i = 0; i < g_twins.size(); ++i) {
        const Twin &t = g_twins[i];
        if (t.device != device || h < t.host || h + bytes > t.host + t.bytes) continue;
        if (home) *home = t.home;
        if (t.owned) return t.dev + (h - t.host);                 // attached by the caller, buffer by buffer: always honoured
        if (!implicit_ok) return nullptr;
        // an entry made on the way (a stage's output buffer): the address alone proves nothing -- the caller may have released the
        // buffer and a new one may sit there. What was hashed when the entry was made must still be there, else the host bytes win
        // and the entry is dropped.
        if (t.guard && guard_of(t.host, t.bytes) != t.guard) { g_twins.erase(g_twins.begin() + (long)i); return nullptr; }
        return t.dev + (h - t.host);
    }
    return nullptr;
}
}  // namespace t2gpu

// ---- the switch of the implicit hand-over (include/t2gpu.h): off unless the caller says its buffers travel unmodified
extern "C" int t2gpu_handoff_enable(int on)
{
    return t2gpu::g_handoff.exchange(on ? 1 : 0);
}

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
    g_twins.push_back(Twin{static_cast<const char *>(host), static_cast<const char *>(dev), bytes, device, true, pinned, 0, nullptr});
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
    hipStream_t side = side_stream(device);                  // in order with the demapper's passes before it and the batch copy behind it
    if (!side) return -1;
    const void *d_src = twin_lookup(src, bytes, device);
    if (d_src) T2_HIP(hipMemcpyAsync(const_cast<void *>(d_dst), d_src, bytes, hipMemcpyDeviceToDevice, side));
    else T2_HIP(hipMemcpyAsync(const_cast<void *>(d_dst), dst, bytes, hipMemcpyHostToDevice, side));
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
