#include "t2gpu_common.h"
#include <map>
#include <mutex>
#include <utility>
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
}  // namespace t2gpu

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
