// ThreadSanitizer stress of the file tier's reader pool (csrc/evs_filetier.hip compiled host-only): two caller threads share one
// tier and fetch overlapping key lists through the persistent pool; every fetched row is compared with the file bytes.
// Built and run by tests/test_hostcache_asan.py.
#include "evstore_hip.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

extern "C" int evs_filetier_fetch(evs_filetier *ft, int64_t n, const uint64_t *keys, void *dst, uint32_t skip_mask);

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const int T = 5, rb = 144;
    const long long rows[T] = {4000, 3, 25000, 1, 900};
    std::vector<std::vector<unsigned char>> tabs(T);
    std::vector<std::string> paths(T);
    std::vector<const char *> cp(T);
    std::mt19937_64 rng(3);
    for (int k = 0; k < T; k++) {
        tabs[k].resize((size_t)rows[k] * rb);
        for (auto &b : tabs[k]) b = (unsigned char)rng();
        paths[k] = std::string(argv[1]) + "/ev-table-" + std::to_string(k + 1) + ".bin";
        FILE *f = fopen(paths[k].c_str(), "wb");
        if (!f || fwrite(tabs[k].data(), 1, tabs[k].size(), f) != tabs[k].size()) return 3;
        fclose(f);
        cp[k] = paths[k].c_str();
    }
    evs_filetier *ft = nullptr;
    if (evs_filetier_open(&ft, T, cp.data(), rb, 0)) { printf("open failed: %s\n", evs_last_error()); return 1; }
    int bad = 0;
    auto caller = [&](unsigned seed) {
        std::mt19937_64 r(seed);
        for (int rep = 0; rep < 12; rep++) {
            const int64_t n = 3000 + (int64_t)(r() % 9000);
            std::vector<uint64_t> keys(n);
            for (auto &k : keys) { const int t = (int)(r() % T); k = ((uint64_t)(t + 1) << 32) | (uint64_t)(r() % rows[t]); }
            std::vector<unsigned char> out((size_t)n * rb, 0);
            if (evs_filetier_fetch(ft, n, keys.data(), out.data(), 0)) { __atomic_add_fetch(&bad, 1, __ATOMIC_RELAXED); return; }
            for (int64_t i = 0; i < n; i++) {
                const int t = (int)(keys[i] >> 32) - 1;
                if (memcmp(&out[(size_t)i * rb], &tabs[t][(size_t)(keys[i] & 0xffffffffull) * rb], rb) != 0) { __atomic_add_fetch(&bad, 1, __ATOMIC_RELAXED); break; }
            }
        }
    };
    std::thread a(caller, 11u), b(caller, 12u);
    a.join(); b.join();
    evs_filetier_close(ft);
    if (bad) { printf("mismatches: %d\n", bad); return 1; }
    printf("reader pool sanitizer stress ok\n");
    return 0;
}
