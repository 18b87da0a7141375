// TEST INFRASTRUCTURE (oracle/): drives the reference's alt-key tier APRX_EV (mixed_precs_caching/aprx_embedding.cpp)
// SINGLE-THREADED through its public methods -- insert_altkey (:278-288), get_altkey_str (:341-350),
// set_recency_flag_c3 (:402-411), evict_one_key (:390-400, recency_aware_eviction :360-388), get_from_file_as_uint
// (:218-251, the big-endian alt-key files) and print_all_keys_in_c3 (:430-434, the FIFO in order).  The batch path
// (add_key_to_batched_io -> 5 worker threads -> insert_altkey_batched_obj, whose loops read uninitialised counters,
// :293,:298,:314,:319) is never triggered, so the run is deterministic; the constructor's worker threads stay parked
// on their semaphores and the process leaves with _exit.  The reference source is compiled where it lies (oracle/Makefile).
//   usage: ref_aprx_driver <ops.bin> <out.bin>        (EVS_REF_ROOT redirects the alt-key directory)
//   ops.bin: int32 n, int32 capacity, then n x (int32 op, int32 table_1based, int32 row)
//            op 0 insert_altkey(key, alt from file) | 1 get_altkey_str | 2 set_recency_flag_c3 | 3 evict_one_key
//   out.bin: n x uint32 (op 0: the alt key read from the file; op 1: alt_row*100+alt_table or 0xffffffff on a miss; else 0)
//   stdout : the queue dump of print_all_keys_in_c3 (after a line of '=')
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#include "aprx_embedding.hpp"

int main(int argc, char **argv) {
    if (argc != 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 3;
    int32_t n = 0, cap = 0;
    if (fread(&n, 4, 1, f) != 1 || fread(&cap, 4, 1, f) != 1) return 4;
    std::vector<int32_t> ops((size_t)n * 3);
    if (fread(ops.data(), 4, ops.size(), f) != ops.size()) return 4;
    fclose(f);
    APRX_EV *c3 = new APRX_EV(cap);
    std::vector<uint32_t> res((size_t)n, 0);
    for (int i = 0; i < n; i++) {
        const int op = ops[3 * i], t = ops[3 * i + 1], r = ops[3 * i + 2];
        const string key = to_string(t) + "-" + to_string(r);
        if (op == 0) {
            const uint32_t alt = c3->get_from_file_as_uint(t, r);
            c3->insert_altkey(key, alt);
            res[i] = alt;
        } else if (op == 1) {
            const string s = c3->get_altkey_str(key);
            if (s.empty()) res[i] = 0xffffffffu;
            else {
                const size_t dash = s.find('-');
                res[i] = (uint32_t)(atoll(s.substr(dash + 1).c_str()) * 100 + atoll(s.substr(0, dash).c_str()));
            }
        } else if (op == 2) {
            c3->set_recency_flag_c3(key);
        } else if (op == 3) {
            c3->evict_one_key();
        }
    }
    FILE *o = fopen(argv[2], "wb");
    if (!o) return 5;
    fwrite(res.data(), 4, res.size(), o);
    fclose(o);
    c3->print_all_keys_in_c3();
    fflush(stdout);
    _exit(0);
}
