// AddressSanitizer / UBSan stress of the host cache engine (csrc/evs_hostcache.hip compiled host-only): random tables, all four
// codecs, EvLFU / LRU / LFU incl. the approximate mode, the two- and three-tier request.  Built and run by tests/test_hostcache_asan.py.
#include "evstore_hip.h"
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <random>
int main() {
    std::mt19937_64 rng(7);
    const int codecs[] = {32, 16, 8, 4};
    long total = 0;
    for (int iter = 0; iter < 60; iter++) {
        const int T = (int)(rng() % 26) + 1, d = (rng() & 1) ? 36 : 16, codec = codecs[rng() % 4];
        const int policy = (int)(rng() % 3);
        const int64_t cap = (int64_t)(rng() % 300) + 1;
        std::vector<int64_t> n_rows(T);
        std::vector<std::vector<unsigned char>> tabs(T);
        std::vector<const void *> ptrs(T);
        const int row_bytes = d * codec / 8;
        for (int k = 0; k < T; k++) {
            n_rows[k] = (int64_t)(rng() % 500) + 1;
            tabs[k].resize((size_t)n_rows[k] * row_bytes);
            for (auto &b : tabs[k]) b = (unsigned char)(rng() % 200);
            ptrs[k] = tabs[k].data();
        }
        evs_hostcache *c = nullptr;
        if (evs_hostcache_create(&c, policy, cap, T, d, codec, 0.3, 0.95, iter & 1, iter % 3)) { printf("create failed: %s\n", evs_last_error()); return 1; }
        evs_hostcache_set_backing(c, ptrs.data(), n_rows.data());
        const int B = 64;
        std::vector<int32_t> rows((size_t)B * T);
        std::vector<float> out((size_t)B * T * d);
        std::vector<uint8_t> hit((size_t)B * T);
        for (int rep = 0; rep < 40; rep++) {
            for (int b = 0; b < B; b++) for (int k = 0; k < T; k++) rows[(size_t)b * T + k] = (int32_t)((rng() % 3 ? rng() % 40 : rng()) % n_rows[k]);
            const int rc = evs_hostcache_request(c, B, rows.data(), out.data(), hit.data(), (rep % 5 == 0 && policy == 0) ? (int)(rng() % (T + 1)) : -1);
            if (rc) { printf("request rc=%d %s\n", rc, evs_last_error()); return 1; }
            total += B;
        }
        int64_t st[8]; evs_hostcache_stats(c, st);
        std::vector<int64_t> tri((size_t)cap * 3 + 3);
        evs_hostcache_dump(c, tri.data(), cap);
        // two / three tiers (EvLFU only)
        if (policy == 0) {
            evs_hostcache *a = nullptr, *b2 = nullptr; evs_hostaprx *x = nullptr;
            evs_hostcache_create(&a, 0, cap, T, d, codec, 0.3, 0.95, 0, 2);
            evs_hostcache_create(&b2, 0, cap * 2, T, d, codec, 0.3, 0.95, 0, 2);
            evs_hostcache_set_backing(a, ptrs.data(), n_rows.data()); evs_hostcache_set_backing(b2, ptrs.data(), n_rows.data());
            std::vector<std::vector<uint32_t>> alt(T); std::vector<const uint32_t *> ap(T);
            for (int k = 0; k < T; k++) { alt[k].resize(n_rows[k]); for (int64_t r = 0; r < n_rows[k]; r++) alt[k][r] = (uint32_t)((rng() % n_rows[(k + 1) % T]) * 100 + ((k + 1) % T + 1)); ap[k] = alt[k].data(); }
            const bool with3 = rng() & 1;
            if (with3) { if (evs_hostaprx_create(&x, 50 + (int64_t)(rng() % 100), T)) { printf("aprx create: %s\n", evs_last_error()); return 1; } evs_hostaprx_set_altkeys(x, ap.data(), n_rows.data()); }
            for (int rep = 0; rep < 30; rep++) {
                for (int b = 0; b < B; b++) for (int k = 0; k < T; k++) rows[(size_t)b * T + k] = (int32_t)((rng() % 3 ? rng() % 40 : rng()) % n_rows[k]);
                const int rc = evs_hostcache_request_c1c2c3(a, b2, x, B, rows.data(), out.data(), hit.data(), 23);
                if (rc) { printf("c1c2c3 rc=%d %s\n", rc, evs_last_error()); return 1; }
                total += B;
            }
            evs_hostcache_destroy(a); evs_hostcache_destroy(b2); if (x) evs_hostaprx_destroy(x);
        }
        evs_hostcache_destroy(c);
    }
    printf("host engine sanitizer stress ok: %ld requests\n", total);
    return 0;
}
