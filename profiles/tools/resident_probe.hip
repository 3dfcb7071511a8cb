// MEASUREMENT TOOL (not product code): what does a device-side mailbox cost per step?  (a resident step kernel
// that waits on a step counter the policy's last kernel bumps, instead of being launched once per step.)
//
// A persistent kernel of WG workgroups (the headline launch: 1024 x 256 threads, 4 per CU) loops over steps: wavefront 0's first
// lane polls `post` (agent-scope acquire) until step t has been posted, the workgroup then writes BYTES bytes of "observations" with
// the step kernel's store instruction (16 bytes per lane, optionally non-temporal), waits for its stores, and signals completion:
// release fence + atomic count; the last workgroup to arrive publishes `done = t`.  On a second stream a one-wavefront kernel per
// step posts t and spins until done == t — the stand-in for "the policy's last kernel" and "the policy's first kernel of the next
// round".  Every spin is bounded (2 s of wall clock -> error flag, exit): a broken protocol cannot hang the box.
// Reported per step: the resident round trip with no stores, with the headline's store volume, and — the baseline — the same
// stores from one ordinary launch per step on one stream.
//   hipcc --offload-arch=gfx950 -O3 -o profiles/tools/resident_probe profiles/tools/resident_probe.hip && ./profiles/tools/resident_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <vector>

struct Mailbox { uint32_t post, done, count, error; uint32_t pad[28]; uint32_t group[32 * 32]; };  // (group counters: one 128-byte line each)

__device__ __forceinline__ uint64_t now_ticks() { return wall_clock64(); }  // 100 MHz

template <bool kNT>
__device__ __forceinline__ void store16(float4 *dst, float4 v) {
    typedef float f4_t __attribute__((ext_vector_type(4)));
    f4_t vv = {v.x, v.y, v.z, v.w};
    if (kNT) __builtin_nontemporal_store(vv, reinterpret_cast<f4_t *>(dst)); else *reinterpret_cast<f4_t *>(dst) = vv;
}

template <bool kNT>
__device__ __forceinline__ void write_chunk(float *out, int bytes_per_wg, int t) {
    float4 *o = reinterpret_cast<float4 *>(out + (size_t)blockIdx.x * (bytes_per_wg / 4));
    const float f = (float)(t & 1);
    for (int i = threadIdx.x; i < bytes_per_wg / 16; i += blockDim.x) store16<kNT>(o + i, float4{f, 0.f, 1.f, f});
}

template <bool kNT>
__global__ void __launch_bounds__(256) resident_kernel(Mailbox *mb, float *out, int bytes_per_wg, int n_steps, int fence_mode) {
    __shared__ int go;
    const uint64_t t0 = now_ticks();
    for (int t = 1; t <= n_steps; ++t) {
        if (threadIdx.x == 0) {
            // (the poll itself is a RELAXED agent-scope load — no cache invalidate per iteration — with one acquire fence behind the loop;
            //  the first version polled with acquire loads and read the wall clock every iteration: 62 us per empty round)
            int ok = 1;
            uint32_t spins = 0;
            while (__hip_atomic_load(&mb->post, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)t) {
                __builtin_amdgcn_s_sleep(2);
                if ((++spins & 0xfff) == 0 && now_ticks() - t0 > 200000000ull) { ok = 0; __hip_atomic_store(&mb->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            go = ok;
        }
        __syncthreads();
        if (!go) return;
        if (bytes_per_wg) write_chunk<kNT>(out, bytes_per_wg, t);
        __syncthreads();  // (hipcc: waits for every wavefront's stores — vmcnt(0) — before the barrier)
        if (threadIdx.x == 0) {
            // fence_mode 0: agent-scope release on the counting atomic (L2 write-back on every workgroup); 1: workgroup-scope only
            // (NOT a valid protocol across XCDs — a lower bound for what the fence costs)
            // 2: release, counted through a two-level tree (32 groups of 32 workgroups, one cache line per group counter): 64 serialised
            // same-address atomics per step instead of 1024
            if (fence_mode == 2) {
                const uint32_t g = blockIdx.x >> 5, per = 32u;
                const uint32_t o1 = __hip_atomic_fetch_add(&mb->group[g * 32], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                if (o1 + 1 == (uint32_t)t * per) {
                    const uint32_t o2 = __hip_atomic_fetch_add(&mb->count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                    if (o2 + 1 == (uint32_t)t * (gridDim.x / per)) __hip_atomic_store(&mb->done, (uint32_t)t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
            const uint32_t old = fence_mode == 0 ? __hip_atomic_fetch_add(&mb->count, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT)
                                                 : __hip_atomic_fetch_add(&mb->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == (uint32_t)t * gridDim.x) __hip_atomic_store(&mb->done, (uint32_t)t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

__global__ void post_and_wait_kernel(Mailbox *mb, uint32_t t) {
    if (threadIdx.x != 0) return;
    const uint64_t t0 = now_ticks();
    __hip_atomic_store(&mb->post, t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t spins = 0;
    while (__hip_atomic_load(&mb->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < t) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 0xfff) == 0 && now_ticks() - t0 > 200000000ull) { __hip_atomic_store(&mb->error, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
}

template <bool kNT>
__global__ void __launch_bounds__(256) launched_kernel(float *out, int bytes_per_wg, int t) { write_chunk<kNT>(out, bytes_per_wg, t); }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const int WG = 1024, T = 2000;
    Mailbox *mb;
    float *out;
    CK(hipMalloc(&mb, sizeof(Mailbox)));
    CK(hipMalloc(&out, (size_t)WG * 65536));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    auto wall = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const int sizes[3] = {0, 18176, 46848};  // bytes of observations per workgroup: none, small-4ag (16 envs x 4 x 71 floats), config 5 (4 x 16 x 183)
    for (int nt = 1; nt >= 0; --nt)
        for (int si = 0; si < 3; ++si) {
            const int bytes = sizes[si];
            if (bytes == 0 && nt == 0) continue;
            // ---- baseline: one launch per step, one stream
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipDeviceSynchronize());
                const double t0 = wall();
                for (int t = 1; t <= T; ++t) {
                    if (nt) hipLaunchKernelGGL(launched_kernel<true>, dim3(WG), dim3(256), 0, sa, out, bytes, t);
                    else hipLaunchKernelGGL(launched_kernel<false>, dim3(WG), dim3(256), 0, sa, out, bytes, t);
                }
                CK(hipDeviceSynchronize());
                if (rep) printf("%-10s %6d B/wg  launched, one stream:            %7.3f us/step\n", nt ? "nt stores" : "cached", bytes, (wall() - t0) / T * 1e6);
            }
            // ---- resident kernel + post/wait kernels on a second stream
            for (int fence = 0; fence < 3; ++fence) {
                double best = 1e9;
                uint32_t err = 0;
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipMemset(mb, 0, sizeof(Mailbox)));
                    CK(hipDeviceSynchronize());
                    const double t0 = wall();
                    if (nt) hipLaunchKernelGGL(resident_kernel<true>, dim3(WG), dim3(256), 0, sa, mb, out, bytes, T, fence);
                    else hipLaunchKernelGGL(resident_kernel<false>, dim3(WG), dim3(256), 0, sa, mb, out, bytes, T, fence);
                    for (int t = 1; t <= T; ++t) hipLaunchKernelGGL(post_and_wait_kernel, dim3(1), dim3(64), 0, sb, mb, (uint32_t)t);
                    CK(hipDeviceSynchronize());
                    const double dt = wall() - t0;
                    if (dt < best) best = dt;
                    Mailbox h;
                    CK(hipMemcpy(&h, mb, sizeof h, hipMemcpyDeviceToHost));
                    err |= h.error | (h.done != (uint32_t)T ? 4u : 0u);
                }
                printf("%-10s %6d B/wg  resident, %s: %7.3f us/step%s\n", nt ? "nt stores" : "cached", bytes,
                       fence == 0 ? "agent-scope release per wg" : fence == 1 ? "no release (lower bound)  " : "release, two-level count  ", best / T * 1e6, err ? "  [PROTOCOL ERROR / TIMEOUT]" : "");
            }
        }
    // ---- the post/wait kernels alone (back-to-back dependent one-wavefront launches on one stream): the floor of the policy side
    {
        CK(hipMemset(mb, 0xff, sizeof(Mailbox)));  // done = 0xffffffff: nobody waits
        CK(hipDeviceSynchronize());
        const double t0 = wall();
        for (int t = 1; t <= T; ++t) hipLaunchKernelGGL(post_and_wait_kernel, dim3(1), dim3(64), 0, sb, mb, (uint32_t)t);
        CK(hipDeviceSynchronize());
        printf("one-wavefront kernels back to back on one stream (no waiting): %7.3f us each\n", (wall() - t0) / T * 1e6);
    }
    return 0;
}
