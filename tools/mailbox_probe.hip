// Developer probe (not product code): where should the mailbox of a resident kernel live?  One wave stays on the device and
// polls a 4-byte sequence word; the host writes k, the wave answers k into a word in pinned host memory, the host spins on
// that.  The request word is placed (a) in pinned host memory (the kernel's poll is a read over the bus: what
// evs_cache_serve_* and evs_emb_interact_serve_* do today), (b) in DEVICE memory the host writes through the PCIe aperture
// (fine-grained allocation; only if the host can address it at all -- probed under a SIGSEGV handler, a fault is an answer).
// Prints the round trip of each placement (p50 / p95 over N posts).
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/mailbox_probe tools/mailbox_probe.hip && tools/_build/mailbox_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <csetjmp>
#include <csignal>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(64) echo_kernel(const unsigned *req, unsigned *ans, unsigned last, long long max_ticks) {
    unsigned want = 1;
    const long long t0 = (long long)wall_clock64();
    while (want <= last) {
        unsigned v;
        for (;;) {
            v = __hip_atomic_load(req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (v >= want) break;
            if ((long long)wall_clock64() - t0 > max_ticks) return;    // (a lost host: leave)
        }
        if (threadIdx.x == 0) __hip_atomic_store(ans, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        want = v + 1;
    }
}

static double run(const char *name, volatile unsigned *req_host_view, const unsigned *req_dev_view, int n) {
    unsigned *ans = nullptr;
    CK(hipHostMalloc((void **)&ans, 64, hipHostMallocMapped));
    *ans = 0; *req_host_view = 0;
    __sync_synchronize();
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipLaunchKernelGGL(echo_kernel, dim3(1), dim3(64), 0, s, req_dev_view, ans, (unsigned)n, 100000000ll * 20);   // (100 MHz clock: 20 s)
    CK(hipGetLastError());
    std::vector<double> us((size_t)n);
    volatile unsigned *va = ans;
    for (int k = 1; k <= n; k++) {
        const auto t0 = std::chrono::steady_clock::now();
        *req_host_view = (unsigned)k;
        __sync_synchronize();
        while (*va != (unsigned)k) { }
        us[(size_t)k - 1] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
    CK(hipStreamSynchronize(s));
    CK(hipStreamDestroy(s));
    CK(hipHostFree(ans));
    std::sort(us.begin(), us.end());
    printf("%-44s round trip p50 %.2f us  p95 %.2f us  min %.2f us\n", name, us[us.size() / 2], us[us.size() * 95 / 100], us[0]);
    fflush(stdout);
    return us[us.size() / 2];
}

static sigjmp_buf g_jmp;
static void on_fault(int) { siglongjmp(g_jmp, 1); }
static bool host_can_write(void *p) {     // a fault is an answer (the driver's mappings are not inherited by a child: probed in place)
    struct sigaction sa, old_segv, old_bus;
    memset(&sa, 0, sizeof sa); sa.sa_handler = on_fault; sigemptyset(&sa.sa_mask);
    sigaction(SIGSEGV, &sa, &old_segv); sigaction(SIGBUS, &sa, &old_bus);
    bool ok = false;
    if (sigsetjmp(g_jmp, 1) == 0) { *(volatile unsigned *)p = 0u; ok = *(volatile unsigned *)p == 0u; }
    sigaction(SIGSEGV, &old_segv, nullptr); sigaction(SIGBUS, &old_bus, nullptr);
    return ok;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 20000;
    unsigned *h = nullptr, *hd = nullptr;
    CK(hipHostMalloc((void **)&h, 64, hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void **)&hd, h, 0));
    run("request word in pinned host memory", h, hd, n);
    run("request word in pinned host memory (again)", h, hd, n);
    struct { const char *name; unsigned flags; int plain; } kinds[] = {
        {"device memory, fine-grained", hipDeviceMallocFinegrained, 0},
        {"device memory, uncached", hipDeviceMallocUncached, 0},
        {"device memory, hipMalloc", 0, 1},
    };
    for (auto &k : kinds) {
        void *d = nullptr;
        hipError_t e = k.plain ? hipMalloc(&d, 4096) : hipExtMallocWithFlags(&d, 4096, k.flags);
        if (e != hipSuccess) { printf("%-44s allocation refused: %s\n", k.name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        CK(hipMemset(d, 0, 4096)); CK(hipDeviceSynchronize());
        if (!host_can_write(d)) { printf("%-44s the host cannot address it (fault in the probing child)\n", k.name); CK(hipFree(d)); continue; }
        run(k.name, (volatile unsigned *)d, (const unsigned *)d, n);
        CK(hipFree(d));
    }
    CK(hipHostFree(h));
    return 0;
}
