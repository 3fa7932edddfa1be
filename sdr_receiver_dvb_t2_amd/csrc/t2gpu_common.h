// t2gpu_common.h -- error plumbing shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <string>

namespace t2gpu {
void set_error(const std::string &msg);
bool hip_ok(hipError_t e, const char *what);
const std::string &last_error();
// hipFuncSetAttribute(fn, MaxDynamicSharedMemorySize, bytes), remembered per (current device, function): the attribute belongs to
// the device's copy of the kernel, a process may drive several devices and several host threads. Thread-safe; a call
// that asks for no more than what is already set for this device returns at once.
hipError_t ensure_dynamic_lds(const void *fn, int bytes);
}  // namespace t2gpu

#define T2_HIP(call)                                   \
    do {                                               \
        if (!t2gpu::hip_ok((call), #call)) return -1;  \
    } while (0)
