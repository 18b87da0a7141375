// Developer probe (not product code): what grid-wide bookkeeping costs inside a launch on this part.
//   1. ARRIVAL COUNTERS: every block of a 1 024-block launch adds 1 to a counter with a device-scope atomic at its end -- all to
//      ONE word, or to word (block % R) of R words in 128-byte lines of their own.  Atomics of different XCDs on one word are
//      served one after the other.
//   2. FENCES: every block stores 24 KB (the fused kernels' output rate) and then executes __threadfence() -- an agent-scope
//      release is an L2 write-back on a part whose XCDs have L2s of their own -- against the same launch without the fence.
// Found while building the in-launch policy update of the cache tier (docs/r04_inline_update.patch).
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/atomic_probe tools/atomic_probe.hip && tools/_build/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

// mode 0: nothing; 1: atomicAdd (no return) on word (block % R) * 32; 2: the same with the returned value used
__global__ void __launch_bounds__(256) arrive_kernel(unsigned *ctr, int R, int mode, float *out, int work) {
    // a little work per block so that the launch is not empty (each thread stores `work` float4s)
    f32x4 v = {1.f, (float)threadIdx.x, 2.f, 3.f};
    f32x4 *o = reinterpret_cast<f32x4 *>(out) + ((size_t)blockIdx.x * 256 + threadIdx.x) * work;
    for (int i = 0; i < work; i++) o[i] = v;
    __syncthreads();
    if (threadIdx.x == 0 && mode) {
        unsigned *p = ctr + (size_t)(blockIdx.x % R) * 32;
        if (mode == 1) atomicAdd(p, 1u);
        else if (atomicAdd(p, 1u) == 0xffffffffu) out[0] = 0.f;
    }
}

// every block stores `work` float4s per thread, then (fence = 1) thread 0 executes __threadfence()
__global__ void __launch_bounds__(256) fence_kernel(float *out, int work, int fence) {
    f32x4 v = {1.f, (float)threadIdx.x, 2.f, 3.f};
    f32x4 *o = reinterpret_cast<f32x4 *>(out) + ((size_t)blockIdx.x * 256 + threadIdx.x) * work;
    for (int i = 0; i < work; i++) o[i] = v;
    __syncthreads();
    if (threadIdx.x == 0 && fence) __threadfence();
}

template <typename F> static float timed(F launch, int iters = 200) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; i++) launch();
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; i++) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters * 1e3f;
}

int main() {
    const int blocks = 1024, work = 6;                       // 6 x 16 B x 256 threads = 24 KB per block, 25 MB per launch
    float *out; CK(hipMalloc(&out, (size_t)blocks * 256 * work * 16));
    unsigned *ctr; CK(hipMalloc(&ctr, 1024 * 128)); CK(hipMemset(ctr, 0, 1024 * 128));
    const float base = timed([&] { hipLaunchKernelGGL(arrive_kernel, dim3(blocks), dim3(256), 0, 0, ctr, 1, 0, out, work); });
    printf("1024 blocks x 24 KB of stores, no counter: %.1f us per launch\n", base);
    for (int R : {1, 8, 32, 128, 1024}) {
        const float t1 = timed([&] { hipLaunchKernelGGL(arrive_kernel, dim3(blocks), dim3(256), 0, 0, ctr, R, 1, out, work); });
        const float t2 = timed([&] { hipLaunchKernelGGL(arrive_kernel, dim3(blocks), dim3(256), 0, 0, ctr, R, 2, out, work); });
        printf("  + one arrival per block on %4d word(s): %.1f us (fire-and-forget), %.1f us (value used) -> %.0f ns per arrival on one word\n",
               R, t1, t2, (t1 - base) * 1e3f / (blocks / (float)R));
    }
    const float f0 = timed([&] { hipLaunchKernelGGL(fence_kernel, dim3(blocks), dim3(256), 0, 0, out, work, 0); });
    const float f1 = timed([&] { hipLaunchKernelGGL(fence_kernel, dim3(blocks), dim3(256), 0, 0, out, work, 1); });
    printf("1024 blocks x 24 KB of stores: %.1f us; with one __threadfence() per block behind them: %.1f us\n", f0, f1);
    return 0;
}
