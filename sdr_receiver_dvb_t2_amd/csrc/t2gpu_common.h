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

// ---- device twins of host buffers (include/t2gpu.h, "host-buffer hand-over"). The reference's slots carry host pointers from stage to
// stage; every host-buffer entry point of this library leaves its result in the caller's buffer AND remembers where the same bytes
// still are on the device. The next stage, handed that buffer unmodified (which is what the reference's signal / slot chain does),
// finds the twin by address and skips its copy-in. All twins are written and read on the null stream (or ordered against it by events).
// T2GPU_HANDOFF=0 disables every look-up (each stage then copies in, as in rounds 1-3).
void twin_publish(const void *host, const void *dev, size_t bytes, int device);   // replaces an entry with the same host base
void twin_retire(const void *host);                                               // the entry whose base is host, if any
void twin_retire_dev(const void *dev_lo, size_t bytes);                           // every entry whose device range lies in [dev_lo, +bytes)
const void *twin_lookup(const void *host, size_t bytes, int device);              // device address of [host, host + bytes) or nullptr
}  // namespace t2gpu

#define T2_HIP(call)                                   \
    do {                                               \
        if (!t2gpu::hip_ok((call), #call)) return -1;  \
    } while (0)
