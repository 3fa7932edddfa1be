#include "t2gpu_common.h"
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
