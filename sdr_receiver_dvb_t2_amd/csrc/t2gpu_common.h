// t2gpu_common.h -- error plumbing shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <string>

namespace t2gpu {
void set_error(const std::string &msg);
bool hip_ok(hipError_t e, const char *what);
const std::string &last_error();
}  // namespace t2gpu

#define T2_HIP(call)                                   \
    do {                                               \
        if (!t2gpu::hip_ok((call), #call)) return -1;  \
    } while (0)
