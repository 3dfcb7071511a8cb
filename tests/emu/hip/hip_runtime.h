// TEST INFRASTRUCTURE — a host-only stand-in for <hip/hip_runtime.h>.  NOT PRODUCT CODE.
//
// Lets g++ compile the UNCHANGED product sources (robotic-warehouse_amd/csrc/rware_capi.hip +
// rware_kernels.h) into tests/emu/librware_emu.so, where each workgroup runs as `blockDim.x`
// OS threads with a pthread barrier for __syncthreads() and __atomic builtins for the LDS
// atomics.  Purpose: exercise the kernel's control flow, the C-ABI host code and the Python
// host layer against the oracle in the GPU-less build container.  It proves nothing about
// gfx950 codegen or performance; the `-m gpu` tests do that on a real MI355X.  Only tests/
// load this library, always by explicit path; the product loader never looks for it.
#pragma once
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#if defined(__SANITIZE_ADDRESS__)
#include <sanitizer/asan_interface.h>
#endif

#include <algorithm>
#include <chrono>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__
#define __align__(x) alignas(x)
#define __launch_bounds__(...)

using std::max;
using std::min;

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(8) int2 { int x, y; };
struct alignas(16) float4 { float x, y, z, w; };

extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
extern thread_local pthread_barrier_t *emu_barrier, *emu_wave_barrier;
namespace rw { extern int32_t smem[]; }

inline unsigned long long wall_clock64() {
    return (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() / 10;
}
inline int __mul24(int a, int b) { return a * b; }  // (v_mul_i32_i24 on the GPU: both factors fit 24 bits where it is used)
inline void __syncthreads() { pthread_barrier_wait(emu_barrier); }
inline int atomicOr(int *p, int v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
inline unsigned atomicOr(unsigned *p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
inline unsigned long long atomicOr(unsigned long long *p, unsigned long long v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
inline int atomicAdd(int *p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline int atomicMax(int *p, int v) {
    int old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return old;
}

typedef int hipError_t;
enum { hipSuccess = 0 };
inline const char *hipGetErrorString(hipError_t) { return "emu"; }
typedef struct emu_stream *hipStream_t;
struct emu_event { std::chrono::steady_clock::time_point t; };
typedef emu_event *hipEvent_t;
struct hipDeviceProp_t { char name[256]; char gcnArchName[256]; int multiProcessorCount; };
enum { hipStreamNonBlocking = 1 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };

inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
    memset(p, 0, sizeof *p);
    strcpy(p->name, "host-thread emulation (tests only)");
    strcpy(p->gcnArchName, "emu");
    p->multiProcessorCount = 1;
    return hipSuccess;
}
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, int) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1 };
inline hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus *s) { *s = hipStreamCaptureStatusNone; return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new emu_event(); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
inline hipError_t hipMalloc(void **p, size_t n) { return posix_memalign(p, 256, n ? n : 16) ? 1 : hipSuccess; }
template <typename T> inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
inline hipError_t hipHostMalloc(void **p, size_t n) { return hipMalloc(p, n); }
inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
// (module API: the run-time specialised builds do not exist in this build — RW_NO_JIT — the names only have to compile)
typedef struct emu_module *hipModule_t;
typedef struct emu_function *hipFunction_t;
inline hipError_t hipModuleUnload(hipModule_t) { return hipSuccess; }
inline hipError_t hipModuleLaunchKernel(hipFunction_t, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, hipStream_t, void **, void **) { return 1; }
inline hipError_t hipExtModuleLaunchKernel(hipFunction_t, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, size_t, hipStream_t, void **, void **,
                                           hipEvent_t, hipEvent_t, unsigned) { return 1; }
inline hipError_t hipFuncSetAttribute(const void *, int, int) { return hipSuccess; }
// (two persistent workgroups on the emulation's one "CU": the pipelined builds then walk several chunks per workgroup)
inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, const void *, int, size_t) { *n = 2; return hipSuccess; }

#include <mutex>
extern std::mutex emu_launch_mutex;  // one "device": launches from several host threads (rw_multi) run one after the other
template <typename... KArgs, typename... Args>
void hipLaunchKernelGGL(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds_bytes, hipStream_t, Args... args) {
    (void)lds_bytes;
    // RWARE_EMU_LAUNCH_COST_US=n (tests of the host-side fan-out): a "null device" — the launch holds the CALLING thread for n
    // microseconds (asleep, not spinning: the measurement must not depend on how many cores the test box has free) and runs nothing;
    // outside the one-device mutex, so launches from several host threads overlap the way enqueues on several real devices would
    if (const char *lc = getenv("RWARE_EMU_LAUNCH_COST_US")) {
        std::this_thread::sleep_for(std::chrono::microseconds(atoi(lc)));
        return;
    }
    std::lock_guard<std::mutex> emu_launch_lock(emu_launch_mutex);
#if defined(__SANITIZE_ADDRESS__)
    // AddressSanitizer build (oracle/sanitize.sh): the "LDS" behind what this launch asked for is poisoned, so a kernel that indexes past
    // its dynamic shared memory faults here the way it would corrupt a neighbour workgroup's LDS on the GPU
    const size_t emu_lds_all = 160 * 1024, emu_lds_used = std::min(emu_lds_all, (lds_bytes + 7) & ~(size_t)7);
    __asan_poison_memory_region((char *)rw::smem + emu_lds_used, emu_lds_all - emu_lds_used);
    struct EmuUnpoison { ~EmuUnpoison() { __asan_unpoison_memory_region((char *)rw::smem, 160 * 1024); } } emu_unpoison;
#endif
    for (unsigned b = 0; b < grid.x; ++b) {
        pthread_barrier_t bar;
        pthread_barrier_init(&bar, nullptr, block.x);
        const unsigned n_waves = (block.x + 63) / 64;
        std::vector<pthread_barrier_t> wbar(n_waves);
        for (unsigned w = 0; w < n_waves; ++w) pthread_barrier_init(&wbar[w], nullptr, std::min(64u, block.x - w * 64));
        pthread_barrier_t *wbars = wbar.data();
        std::vector<std::thread> th;
        th.reserve(block.x);
        for (unsigned t = 0; t < block.x; ++t)
            th.emplace_back([=, &bar]() {
                emu_wave_barrier = wbars + t / 64;
                threadIdx = dim3(t);
                blockIdx = dim3(b);
                blockDim = block;
                gridDim = grid;
                emu_barrier = &bar;
                kernel(args...);
            });
        for (auto &x : th) x.join();
        pthread_barrier_destroy(&bar);
        for (auto &wb : wbar) pthread_barrier_destroy(&wb);
    }
}
