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
// finds the twin by address and skips its copy-in -- but only once the caller has said so (t2gpu_handoff_enable; the stage classes of
// t2gpu_stages.hpp do): by default a host-buffer entry point is a function of the bytes it is handed and nothing else. An entry made
// on the way carries a hash of the buffer's first and last 64 bytes and its length; a look-up that finds them changed (a recycled
// address) drops the entry. Buffers attached by the caller (t2gpu_twin_attach) are honoured always.
// Twins are written and read on the null stream or on the device's side stream (below), one after the other by stream order, by an
// event, or because the host has waited for the writer before it calls the reader.
// home: the stream the twin is written on -- a reader that goes on using the twin after its call has returned (t2gpu_ti_push's scatter)
// puts that work there, so that the writer's next pass over the buffer comes behind it by stream order
void twin_publish(const void *host, const void *dev, size_t bytes, int device, bool guarded = true, hipStream_t home = nullptr);   // replaces an entry with the same host base
void twin_retire(const void *host);                                               // the entry whose base is host, if any
void twin_retire_dev(const void *dev_lo, size_t bytes);                           // every entry whose device range lies in [dev_lo, +bytes)
bool handoff_on();                                                                // t2gpu_handoff_enable's state
const void *twin_lookup(const void *host, size_t bytes, int device, hipStream_t *home = nullptr);   // device address of [host, host + bytes) or nullptr
// The device's SIDE STREAM (non-blocking, highest priority -- a hardware queue no decode of t2gpu_ldpc_submit ever sits in --, created on
// first use, lives as long as the process): where the host-buffer entry points of
// the FEC side put their block-sized work -- a TI block's copy down, the demapper's passes, the SIMD batch copies, the descrambler --
// so that a caller's per-symbol launches on the null stream (t2gpu_demod_execute) never queue behind a 13 MB copy or a statistics
// walk. nullptr (and last_error set) when it cannot be created.
hipStream_t side_stream(int device);
}  // namespace t2gpu

// one step of a host spin loop (the page-locked sequence words the device raises): the CPU's own hint where there is one
static inline void t2_cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    asm volatile("" ::: "memory");
#endif
}

#define T2_HIP(call)                                   \
    do {                                               \
        if (!t2gpu::hip_ok((call), #call)) return -1;  \
    } while (0)

// T2GPU_EXIT_TRACE=1: marks on stderr inside the destroy calls (where a process that will not end is standing: tools/hang_hunt.py)
#include <cstdio>
#include <cstdlib>
static inline void t2_exit_mark(const char *what)
{
    static const bool on = std::getenv("T2GPU_EXIT_TRACE") != nullptr;
    if (on) { std::fprintf(stderr, "exit trace: %s\n", what); std::fflush(stderr); }
}
