// TEST INFRASTRUCTURE — storage for the HIP stand-in (see hip/hip_runtime.h).  NOT PRODUCT CODE.
#include <hip/hip_runtime.h>
thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
thread_local pthread_barrier_t *emu_barrier = nullptr, *emu_wave_barrier = nullptr;
namespace rw { alignas(16) int32_t smem[160 * 1024 / 4]; }
namespace rw { int emu_wave_any_flag[16] = {0}; }
namespace rw { int emu_xlane[16][64] = {{0}}; }
std::mutex emu_launch_mutex;
