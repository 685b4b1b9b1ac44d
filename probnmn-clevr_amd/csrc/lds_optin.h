// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: a process-wide "configured" flag
// would launch the kernel on a second GPU without the opt-in.  One bit per (kernel, device), thread-safe;
// the fast path is a thread-local hipGetDevice and one atomic load.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

namespace pnmn {

inline int opt_in_lds(const void* kernel, size_t bytes, std::atomic<uint64_t>& done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return 0;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
    done.fetch_or(bit, std::memory_order_release);
    return 0;
}

}  // namespace pnmn
