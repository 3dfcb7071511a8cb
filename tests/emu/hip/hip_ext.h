// TEST INFRASTRUCTURE: host stand-in for <hip/hip_ext.h> (see hip_runtime.h in this directory).
#pragma once
#include "hip_runtime.h"

template <typename... KArgs, typename... Args>
void hipExtLaunchKernelGGL(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds_bytes, hipStream_t s,
                           hipEvent_t start, hipEvent_t stop, int /*flags*/, Args... args) {
    if (start) hipEventRecord(start, s);
    hipLaunchKernelGGL(kernel, grid, block, lds_bytes, s, args...);
    if (stop) hipEventRecord(stop, s);
}
