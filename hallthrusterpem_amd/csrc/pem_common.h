// pem_common.h -- host-side helpers shared by the translation units of libpem_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "pem_hip.h"

namespace pem {

char* error_buffer();          // thread-local, 512 bytes (defined in pem_kernels.hip)

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

inline int check_device() {
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    if (e != hipSuccess || cnt == 0) {
        (void)hipGetLastError();
        return fail(PEM_ERR_NO_DEVICE, "no HIP device available (%s); libpem_hip has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    }
    return PEM_OK;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel; a function-local `static` would set it
// for the first device a process launches on only, and every later launch on another GPU of the process that asks for more
// than 64 KB of dynamic LDS would fail.  One of these per kernel instantiation: a bit per device, set after the first
// successful call on it.  The call is idempotent, so two threads racing through it do no harm (relaxed atomics only keep
// the flag word itself well defined).
struct LdsAttrOnce {
    std::atomic<uint64_t> done{0};
    hipError_t ensure(const void* kernel, int bytes = 160 * 1024) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        const uint64_t bit = 1ull << (dev & 63);
        if (done.load(std::memory_order_relaxed) & bit) return hipSuccess;
        e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess) done.fetch_or(bit, std::memory_order_relaxed);
        return e;
    }
};

// compute units of the calling thread's current device (cached per device, safe from any thread)
inline hipError_t device_cus(int* out) {
    static std::atomic<int> cus[64];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    int v = cus[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        e = hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return e;
        cus[dev].store(v, std::memory_order_relaxed);
    }
    *out = v;
    return hipSuccess;
}

// XCD-aware block order.  Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one): with tile = workgroup index
// every XCD touches every eighth piece of a streamed array; through this bijective remap (cdna_hip_programming.md: "XCD swizzle
// must be bijective") the workgroups that share an XCD walk one contiguous eighth instead, and each XCD's L2 hands the memory
// system one sequential stream.  A speed matter only: any assignment of tiles to workgroups is correct.  -DPEM_XCD_REMAP=0 turns
// it off (A/B builds).  Used by plume_r1_kernel ONLY: measured on the other streaming kernels of the library (r03z) it costs --
// SVD compress / reconstruct without a norm 191 -> 217 / 177 -> 199 us, the staged radii kernel 389 -> 500 us at 64 radii, the
// likelihood kernel 172 -> 176 us: their workgroups walk several larger tiles each, and the contiguous eighths then put the 8
// XCDs' frontiers 1/8 of the array apart instead of next to each other (profiles/grid_modes_r03.txt).
#ifndef PEM_XCD_REMAP
#define PEM_XCD_REMAP 1
#endif
__device__ __forceinline__ unsigned xcd_contiguous_block() {
#if PEM_XCD_REMAP
    const unsigned q = gridDim.x >> 3, r = gridDim.x & 7, x = blockIdx.x & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (blockIdx.x >> 3);
#else
    return blockIdx.x;
#endif
}

}  // namespace pem

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return pem::fail(e_ == hipErrorNoDevice ? PEM_ERR_NO_DEVICE : PEM_ERR_HIP, "%s: %s", #expr, \
                             hipGetErrorString(e_));                                                    \
    } while (0)
