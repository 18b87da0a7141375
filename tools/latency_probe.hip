// Developer probe (not product code): what ONE dependent memory access costs a lone wave on this part -- the unit the exact
// batch-1 engine (csrc/evs_cache.hip, one wave per cache) is made of.  A pointer chase over a buffer of a given size (random
// cycle, one 64-byte line per hop), timed with the 100 MHz wall clock and the shader clock (s_memtime) side by side: the ratio
// is the clock the part runs at while it does nothing else.  Optionally a second stream keeps the other CUs busy with ALU work
// ("heater") to see whether the lone wave's latency is a clock-management artefact.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/latency_probe tools/latency_probe.hip && tools/_build/latency_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <random>
#include <numeric>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(64) chase_kernel(const unsigned *next, int hops, unsigned start, long long *res, int scope) {
    unsigned i = start;
    const long long w0 = (long long)wall_clock64(), c0 = (long long)clock64();
    if (scope == 0) for (int h = 0; h < hops; h++) i = next[(size_t)i * 16];
    else if (scope == 1) for (int h = 0; h < hops; h++) i = __hip_atomic_load(next + (size_t)i * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else for (int h = 0; h < hops; h++) i = __hip_atomic_load(next + (size_t)i * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const long long w1 = (long long)wall_clock64(), c1 = (long long)clock64();
    if (threadIdx.x == 0) { res[0] = w1 - w0; res[1] = c1 - c0; res[2] = i; }
}
// W lanes of the wave chase chains of their own (lane l starts l * 997 hops into the cycle): what a stage of the engine is --
// 26 lanes, each after a different line
__global__ void __launch_bounds__(64) chase_wide_kernel(const unsigned *next, int hops, const unsigned *starts, int W, long long *res) {
    unsigned i = starts[threadIdx.x];
    const bool on = (int)threadIdx.x < W;
    const long long w0 = (long long)wall_clock64();
    for (int h = 0; h < hops; h++) if (on) i = next[(size_t)i * 16];
    const long long w1 = (long long)wall_clock64();
    if (i == 0xffffffffu) res[3] = 1;
    if (threadIdx.x == 0) res[0] = w1 - w0;
}
// what reading the 100 MHz clock itself costs (the stage probes read it ten times per request / batch)
__global__ void __launch_bounds__(64) clock_cost_kernel(long long *res) {
    const long long c0 = (long long)clock64();
    long long acc = 0;
    const long long w0 = (long long)wall_clock64();
    for (int i = 0; i < 1000; i++) acc += (long long)wall_clock64();
    const long long w1 = (long long)wall_clock64();
    if (threadIdx.x == 0) { res[0] = w1 - w0; res[1] = (long long)clock64() - c0; res[2] = acc; }
}
__global__ void __launch_bounds__(256) heater_kernel(float *sink, const int *stop, long long max_ticks) {
    float a = (float)threadIdx.x, b = 1.0001f;
    const long long t0 = (long long)wall_clock64();
    for (;;) {
#pragma unroll
        for (int k = 0; k < 256; k++) a = a * b + 0.5f;
        if (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) || (long long)wall_clock64() - t0 > max_ticks) break;
    }
    if (a == 12345.f) sink[0] = a;
}

int main() {
    long long *res; CK(hipHostMalloc((void **)&res, 64, hipHostMallocMapped));
    int *stop; CK(hipHostMalloc((void **)&stop, 64, hipHostMallocMapped));
    float *sink; CK(hipMalloc((void **)&sink, 64));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const char *scopes[] = {"plain load", "agent-scope load", "system-scope load"};
    hipLaunchKernelGGL(clock_cost_kernel, dim3(1), dim3(64), 0, s1, res);
    CK(hipStreamSynchronize(s1));
    printf("one read of wall_clock64() (s_memrealtime + the wait for it): %.0f ns\n", res[0] * 10.0 / 1000);
    for (int heat = 0; heat < 2; heat++) {
        *stop = 0;
        if (heat) { hipLaunchKernelGGL(heater_kernel, dim3(255 * 4), dim3(256), 0, s2, sink, stop, 100000000ll * 20); CK(hipGetLastError()); }
        for (size_t mb : {1ul, 64ul, 4096ul}) {
            const size_t lines = mb * (1ul << 20) / 64;
            std::vector<unsigned> perm(lines); std::iota(perm.begin(), perm.end(), 0u);
            std::mt19937_64 g(7); std::shuffle(perm.begin(), perm.end(), g);
            std::vector<unsigned> buf(lines * 16, 0u);
            for (size_t k = 0; k < lines; k++) buf[(size_t)perm[k] * 16] = perm[(k + 1) % lines];
            unsigned *d; CK(hipMalloc((void **)&d, lines * 64));
            CK(hipMemcpy(d, buf.data(), lines * 64, hipMemcpyHostToDevice));
            for (int scope = 0; scope < 3; scope++) {
                const int hops = 20000;
                hipLaunchKernelGGL(chase_kernel, dim3(1), dim3(64), 0, s1, d, 2000, perm[0], res, scope);   // (warm)
                hipLaunchKernelGGL(chase_kernel, dim3(1), dim3(64), 0, s1, d, hops, perm[0], res, scope);
                CK(hipStreamSynchronize(s1));
                printf("%s  %5zu MB  %-18s  %.0f ns per dependent access  (shader clock %.2f GHz)\n", heat ? "others busy" : "lone wave  ", mb, scopes[scope],
                       res[0] * 10.0 / hops, (double)res[1] / (res[0] * 10.0));
            }
            if (!heat) {
                std::vector<unsigned> st(64);
                for (int l = 0; l < 64; l++) st[l] = perm[((size_t)l * 99991) % lines];
                unsigned *ds; CK(hipMalloc((void **)&ds, 256)); CK(hipMemcpy(ds, st.data(), 256, hipMemcpyHostToDevice));
                for (int W : {1, 8, 26, 64}) {
                    const int hops = 5000;
                    hipLaunchKernelGGL(chase_wide_kernel, dim3(1), dim3(64), 0, s1, d, hops, ds, W, res);
                    CK(hipStreamSynchronize(s1));
                    printf("lone wave    %5zu MB  %2d lanes, a chain each  %.0f ns per step\n", mb, W, res[0] * 10.0 / hops);
                }
                CK(hipFree(ds));
            }
            CK(hipFree(d));
        }
        if (heat) { *stop = 1; CK(hipStreamSynchronize(s2)); }
    }
    return 0;
}
