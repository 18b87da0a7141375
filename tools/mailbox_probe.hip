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

// the dispatcher's case: a post = one 64-byte descriptor written into R lines of device memory (a replica line per group of
// polling blocks) through the aperture; R blocks poll a line each and the LAST of them to see number k answers (an arrival counter)
__global__ void __launch_bounds__(64) echo_lines_kernel(const unsigned *lines, unsigned *ctr, unsigned *ans, unsigned last, long long max_ticks) {
    const unsigned *my = lines + (size_t)blockIdx.x * 32;     // 128-byte stride
    const long long t0 = (long long)wall_clock64();
    for (unsigned want = 1; want <= last; want++) {
        unsigned v;
        for (;;) {
            v = __hip_atomic_load(my + (threadIdx.x & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((__ballot(v == want) & 0x8080ull) == 0x8080ull) break;         // both guards (words 7 and 15)
            if ((long long)wall_clock64() - t0 > max_ticks) return;
        }
        if (threadIdx.x == 0) {
            const unsigned before = atomicAdd(ctr + (want & 63u) * 32, 1u);
            if (before + 1u == gridDim.x) { __hip_atomic_store(ctr + (want & 63u) * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   /* (a plain store stays in this XCD's L2) */ __hip_atomic_store(ans, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
        }
    }
}
static void run_lines(int R, int n) {
    unsigned *lines = nullptr, *ctr = nullptr, *ans = nullptr;
    // (fine-grained: a poll of plain hipMalloc memory is served by the polling XCD's L2 once the line is in it -- eight blocks on
    //  eight XCDs never saw the second post)
    CK(hipExtMallocWithFlags((void **)&lines, (size_t)R * 128, hipDeviceMallocFinegrained)); CK(hipMemset(lines, 0, (size_t)R * 128));
    CK(hipMalloc((void **)&ctr, 64 * 128)); CK(hipMemset(ctr, 0, 64 * 128));
    CK(hipHostMalloc((void **)&ans, 64, hipHostMallocMapped)); *ans = 0;
    CK(hipDeviceSynchronize());
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipLaunchKernelGGL(echo_lines_kernel, dim3(R), dim3(64), 0, s, lines, ctr, ans, (unsigned)n, 100000000ll * 3);
    CK(hipGetLastError());
    std::vector<double> us((size_t)n), wr((size_t)n);
    volatile unsigned *va = ans;
    for (int k = 1; k <= n; k++) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < R; r++) {
            volatile unsigned *l = lines + (size_t)r * 32;
            for (int w = 0; w < 16; w++) if (w != 7 && w != 15) l[w] = 0x1000u + (unsigned)w;
            l[7] = (unsigned)k; l[15] = (unsigned)k;
        }
        __builtin_ia32_sfence();
        const auto t1 = std::chrono::steady_clock::now();
        bool lost = false;
        while (*va != (unsigned)k) { if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 1.0) { lost = true; break; } }
        if (lost) { printf("%3d lines: post %d was never answered (the pollers do not see it)\n", R, k); break; }
        us[(size_t)k - 1] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        wr[(size_t)k - 1] = std::chrono::duration<double, std::micro>(t1 - t0).count();
    }
    CK(hipStreamSynchronize(s)); CK(hipStreamDestroy(s));
    std::sort(us.begin(), us.end()); std::sort(wr.begin(), wr.end());
    printf("%3d lines of 64 bytes written through the aperture per post: the writes p50 %.2f us; post -> all %d pollers arrived -> answer p50 %.2f us (p95 %.2f)\n",
           R, wr[wr.size() / 2], R, us[us.size() / 2], us[us.size() * 95 / 100]);
    fflush(stdout);
    CK(hipFree(lines)); CK(hipFree(ctr)); CK(hipHostFree(ans));
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
    {   void *d = nullptr; CK(hipMalloc(&d, 4096));
        if (host_can_write(d)) for (int R : {1, 8, 32, 33}) run_lines(R, n / 8);
        CK(hipFree(d)); }
    CK(hipHostFree(h));
    return 0;
}
