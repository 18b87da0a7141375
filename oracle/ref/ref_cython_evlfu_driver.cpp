// TEST INFRASTRUCTURE (oracle/): drives the reference's C++ EvLFU of the Cython build
// (cache_algo/EvLFU_C1_Cython/EvLFU.cpp:70-232, flush_rate 0.4 / perfect_item_cap 1.0, :12-13) over a request file and
// records its hit flags, the rows it returned and its final priority lists.  The reference source is compiled where it
// lies (oracle/Makefile: -Dmain=ref_cython_unused_main drops its own workload-replay main); nothing of it is copied here.
//   usage: ref_cython_evlfu <requests.bin> <out.bin>
//   requests.bin: int32 n, int32 capacity, then n x 26 int32 row ids
//   out.bin: n x 26 u8 hit flags | n x 26 x 36 f32 rows | int64 state[3] = {min_C1, n_perfect_item_C1, size}
//            | int64 m | m x 3 int64 (bucket, table_1based, row) in list order, buckets 0..26
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include "evlfu.hpp"

extern int min_C1, n_perfect_item_C1;
extern unordered_map<string, Cache_data> vals_C1;
extern unordered_map<int, list<string>> lists_C1;

int main(int argc, char **argv) {
    if (argc != 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 3;
    int32_t n = 0, cap = 0;
    if (fread(&n, 4, 1, f) != 1 || fread(&cap, 4, 1, f) != 1) return 4;
    std::vector<int32_t> req((size_t)n * 26);
    if (fread(req.data(), 4, req.size(), f) != req.size()) return 4;
    fclose(f);
    load_ev_tables();
    init(cap);
    std::vector<uint8_t> hits((size_t)n * 26);
    std::vector<float> rows((size_t)n * 26 * 36);
    for (int i = 0; i < n; i++) {
        vector<int> keys(req.begin() + (size_t)i * 26, req.begin() + (size_t)(i + 1) * 26);
        vector<bool> rec(26);
        vector<vector<float>> emb(26);
        request_to_ev_lfu(keys, rec, emb, false);
        for (int k = 0; k < 26; k++) {
            hits[(size_t)i * 26 + k] = rec[k];
            for (int j = 0; j < 36; j++) rows[((size_t)i * 26 + k) * 36 + j] = emb[k].size() == 36 ? emb[k][j] : 0.f;
        }
    }
    FILE *o = fopen(argv[2], "wb");
    if (!o) return 5;
    fwrite(hits.data(), 1, hits.size(), o);
    fwrite(rows.data(), 4, rows.size(), o);
    int64_t st[3] = {min_C1, n_perfect_item_C1, (int64_t)vals_C1.size()};
    fwrite(st, 8, 3, o);
    std::vector<int64_t> tri;
    for (int b = 0; b <= 26; b++)
        for (const string &key : lists_C1[b]) {
            const size_t dash = key.find('-');
            tri.push_back(b);
            tri.push_back(atoll(key.substr(0, dash).c_str()));
            tri.push_back(atoll(key.substr(dash + 1).c_str()));
        }
    int64_t m = (int64_t)tri.size() / 3;
    fwrite(&m, 8, 1, o);
    fwrite(tri.data(), 8, tri.size(), o);
    fclose(o);
    return 0;
}
