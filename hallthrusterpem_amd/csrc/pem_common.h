// pem_common.h -- host-side helpers shared by the translation units of libpem_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
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

}  // namespace pem

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return pem::fail(e_ == hipErrorNoDevice ? PEM_ERR_NO_DEVICE : PEM_ERR_HIP, "%s: %s", #expr, \
                             hipGetErrorString(e_));                                                    \
    } while (0)
