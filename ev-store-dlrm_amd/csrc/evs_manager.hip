// Cache manager: the reference's C ABI (mixed_precs_caching/cache_manager.cpp) on top of the
// GPU cache tier.  What ctypes binds in cache_algo/cpp_socket_client.py:69-83 keeps working:
//   float* ev_lookup(int*)                 cache_manager.cpp:231-237
//   float* get_ev_values(int*)             :257-259
//   void   print_perfect_hit()             :262-290
//   int    ev_lookup_based_on_list_keys()  :239-253 (dead in the reference: prints and exit(-1))
//   void   test_arr(int*)                  :154-168
//   init_global_vars / start_server_threads :410-445 (epoll socket server: out of scope)
// The reference is configured by editing #defines and recompiling (cache_manager.cpp:13-20);
// here the same five knobs are runtime arguments of evs_manager_configure() or environment
// variables read on the first ev_lookup (so a zero-argument dlopen + ev_lookup still works).
//
// Engines.  ev_lookup is ONE request per call by construction, and the policies behind it are sequential (request n sees
// the inserts of request n-1): that is the host engine's job (evs_hostcache.hip: a few microseconds per request on one
// core, tables = read-only mappings of the .bin files, nothing touches the GPU) and it is the DEFAULT (backing 2,
// EVS_BACKING=host).  backing 0 / 1 (EVS_BACKING=hbm / pinned) run the same requests through the GPU tier's exact
// kernel, one launch + one synchronise per call (~50 us) -- for a deployment that shares ONE cache between this batch-1
// surface and the batched GPU lookups.  Same results either way (tests/_ev_lookup_child.py runs all three).
#include "evs_common.h"

#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

namespace evs {

constexpr int kEvDim = 36;     // EV_DIMENSION (cache_manager.hpp:30)
constexpr int kEvTables = 26;  // N_EV_TABLE   (cache_manager.hpp:31)

struct Manager {
    bool ready = false;
    int n_layer = 1, main_prec = 32, secondary_prec = 4, backing_kind = 2;   // 0 HBM, 1 pinned (GPU engine); 2 host engine
    long long total_size = 75425;
    long long cap1 = 0, cap2 = 0, cap3 = 0;   // entries per tier, as configured
    std::string proportion = "", root = "";
    evs_cache *c1 = nullptr, *c2 = nullptr;
    evs_aprx *c3 = nullptr;
    evs_hostcache *h1 = nullptr, *h2 = nullptr;   // host engine
    evs_hostaprx *h3 = nullptr;
    std::vector<unsigned> host_alt[kEvTables];
    std::string altkey_dir = "";
    void *alt_tables[kEvTables] = {nullptr};
    long long alt_rows[kEvTables] = {0};
    void *tables[kEvTables] = {nullptr}, *tables2[kEvTables] = {nullptr};
    long long rows[kEvTables] = {0}, rows2[kEvTables] = {0};
    int *d_rows = nullptr, *h_rows = nullptr;
    float *d_out = nullptr, *h_out = nullptr;
    unsigned char *d_hit = nullptr, *h_hit = nullptr;
    hipStream_t stream = nullptr;
};
static Manager g_mgr;
static float g_emb_weights_in_1d_floats[kEvTables * kEvDim];  // cache_manager.hpp:51 (library-owned, static)

static const char *precision_dir(int bits) {
    // evlfu_32.hpp:61, evlfu_16.hpp:64, evlfu_8.hpp:58, evlfu_4.hpp:61
    switch (bits) {
    case 32: return "ev-table/binary/";
    case 16: return "ev-table-16/binary/";
    case 8: return "ev-table-8/binary/";
    default: return "ev-table-4/binary/";
    }
}

static int load_tables(Manager &m, int prec, void **tables, long long *rows) {
    const long long rb = (long long)kEvDim * prec / 8;
    for (int k = 0; k < kEvTables; k++) {
        const std::string path = m.root + "/" + precision_dir(prec) + "ev-table-" + std::to_string(k + 1) + ".bin";
        FILE *fp = fopen(path.c_str(), "rb");
        if (!fp) {  // evlfu_32.cpp:45-48 exits; the 8/4-bit tiers silently keep NULL and crash later
            set_error("ERROR: Failed to load_ev_tables() when opening %s", path.c_str());
            return EVS_EIO;
        }
        fseek(fp, 0, SEEK_END);
        const long long bytes = ftell(fp);
        fseek(fp, 0, SEEK_SET);
        if (bytes % rb) { fclose(fp); set_error("%s: not a whole number of %lld-byte rows", path.c_str(), rb); return EVS_EIO; }
        void *host = nullptr;
        if (hipHostMalloc(&host, bytes > 0 ? bytes : 16, hipHostMallocDefault) != hipSuccess) { fclose(fp); return EVS_ENOMEM; }
        const size_t got = bytes ? fread(host, 1, bytes, fp) : 0;
        fclose(fp);
        if ((long long)got != bytes) { (void)hipHostFree(host); set_error("%s: short read", path.c_str()); return EVS_EIO; }
        if (m.backing_kind == 0) {  // HBM-resident tables
            void *dev = nullptr;
            if (hipMalloc(&dev, bytes > 0 ? bytes : 16) != hipSuccess) { (void)hipHostFree(host); return EVS_ENOMEM; }
            if (hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice) != hipSuccess) return EVS_EHIP;
            (void)hipHostFree(host);
            tables[k] = dev;
        } else {                    // pinned host memory, read by the GPU on a miss
            tables[k] = host;
        }
        rows[k] = bytes / rb;
    }
    return EVS_OK;
}

// host engine: the table files mapped read-only -- a miss reads one row through the page cache, exactly what the
// reference's fseek + fread does (evlfu_8.cpp:380-414) without the two system calls
static int map_tables(Manager &m, int prec, void **tables, long long *rows) {
    const long long rb = (long long)kEvDim * prec / 8;
    for (int k = 0; k < kEvTables; k++) {
        const std::string path = m.root + "/" + precision_dir(prec) + "ev-table-" + std::to_string(k + 1) + ".bin";
        const int fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) {
            set_error("ERROR: Failed to load_ev_tables() when opening %s", path.c_str());
            return EVS_EIO;
        }
        struct stat sb;
        if (fstat(fd, &sb) != 0) { close(fd); set_error("%s: fstat failed", path.c_str()); return EVS_EIO; }
        const long long bytes = (long long)sb.st_size;
        if (bytes % rb) { close(fd); set_error("%s: not a whole number of %lld-byte rows", path.c_str(), rb); return EVS_EIO; }
        void *p = nullptr;
        if (bytes > 0) {
            p = mmap(nullptr, (size_t)bytes, PROT_READ, MAP_PRIVATE, fd, 0);
            if (p == MAP_FAILED) { close(fd); set_error("%s: mmap failed", path.c_str()); return EVS_EIO; }
            (void)madvise(p, (size_t)bytes, MADV_RANDOM);
        }
        close(fd);
        tables[k] = p;
        rows[k] = bytes / rb;
    }
    return EVS_OK;
}

static int read_altkeys(Manager &m, int k, std::vector<unsigned> &native) {
    // alt-key files: 4-byte BIG-endian words, alt_row*100 + alt_table (script/convert_altkeys_to_binary.py:27-57)
    const std::string path = m.altkey_dir + "/ev-table-" + std::to_string(k + 1) + ".bin";
    FILE *fp = fopen(path.c_str(), "rb");
    if (!fp) { set_error("cannot open alt-key file %s", path.c_str()); return EVS_EIO; }
    fseek(fp, 0, SEEK_END);
    const long long bytes = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    std::vector<unsigned char> buf(bytes > 0 ? bytes : 4);
    const size_t got = bytes ? fread(buf.data(), 1, bytes, fp) : 0;
    fclose(fp);
    if ((long long)got != bytes || bytes % 4) { set_error("%s: bad size", path.c_str()); return EVS_EIO; }
    native.assign(bytes / 4 + 1, 0u);
    for (long long r = 0; r < bytes / 4; r++)
        native[r] = ((unsigned)buf[4 * r] << 24) | ((unsigned)buf[4 * r + 1] << 16) | ((unsigned)buf[4 * r + 2] << 8) | buf[4 * r + 3];
    m.alt_rows[k] = bytes / 4;
    return EVS_OK;
}

static int ensure_ready() {
    Manager &m = g_mgr;
    if (m.ready) return EVS_OK;
    if (m.root.empty()) {
        const char *e;
        if ((e = getenv("EVS_N_CACHING_LAYER"))) m.n_layer = atoi(e);
        if ((e = getenv("EVS_MAIN_PRECISION"))) m.main_prec = atoi(e);
        if ((e = getenv("EVS_SECONDARY_PRECISION"))) m.secondary_prec = atoi(e);
        if ((e = getenv("EVS_TOTAL_SIZE"))) m.total_size = atoll(e);
        if ((e = getenv("EVS_SIZE_PROPORTION"))) m.proportion = e;
        if ((e = getenv("EVS_BACKING"))) m.backing_kind = strcmp(e, "pinned") == 0 ? 1 : strcmp(e, "hbm") == 0 ? 0 : strcmp(e, "host") == 0 ? 2 : -1;
        if ((e = getenv("EVS_EV_TABLE_ROOT"))) m.root = e;
        if ((e = getenv("EVS_ALTKEY_DIR"))) m.altkey_dir = e;
        if (m.root.empty()) {
            set_error("cache manager is not configured: call evs_manager_configure() or set EVS_EV_TABLE_ROOT");
            return EVS_ESTATE;
        }
    }
    if (m.n_layer < 1 || m.n_layer > 3) {
        set_error("ERROR: cache_manager.cpp N_CACHING_LAYER is NOT recognized!!");  // cache_manager.cpp:224-225
        return EVS_ESTATE;
    }
    if (m.backing_kind < 0 || m.backing_kind > 2) {
        set_error("backing / EVS_BACKING must be host (2, default), hbm (0) or pinned (1)");
        return EVS_EINVAL;
    }
    if (m.n_layer == 3 && m.altkey_dir.empty()) {
        set_error("N_CACHING_LAYER=3 needs the alt-key directory (evs_manager_set_altkey_dir or EVS_ALTKEY_DIR)");
        return EVS_ESTATE;
    }
    if (m.n_layer >= 2 && !(m.secondary_prec == 16 || m.secondary_prec == 8 || m.secondary_prec == 4) ) {
        set_error("ERROR: Secondary precision (%d) is NOT recognized!", m.secondary_prec);  // evlfu_8.cpp:147
        return EVS_EINVAL;
    }
    if (!(m.main_prec == 32 || m.main_prec == 16 || m.main_prec == 8 || m.main_prec == 4)) {
        set_error("ERROR: MAIN_PRECISION %d", m.main_prec);
        return EVS_EINVAL;
    }
    const bool host_engine = m.backing_kind == 2;
    int rc = host_engine ? map_tables(m, m.main_prec, m.tables, m.rows) : load_tables(m, m.main_prec, m.tables, m.rows);
    if (rc) return rc;
    // Entries per tier exactly as the reference's constructors compute them; sizes are in fp32-row equivalents.
    //   one tier  (cache_manager.cpp:40-53): TOTAL_SIZE x 32/main
    //   two tiers (cache_manager.cpp:36-38 cacheSize = TOTAL_SIZE/2; evlfu_8.cpp:86-88): C1 = share x 32/main;
    //     C2 = share x 32/secondary -- except an 8-bit SECONDARY tier under a 32- or 16-bit main tier: the reference
    //     builds it as EVLFU_8BIT(cap_C1*4 | cap_C1*2, ...) (evlfu_32.cpp:102, evlfu_16.cpp:95) and that constructor
    //     multiplies by 4 again (evlfu_8.cpp:93), i.e. share x 16 entries.  Reproduced: hit rates must match the build.
    //   three tiers (8-bit main only, evlfu_8.cpp:63-85): "a-b-c" proportions x4 / x8 / x36, else equal thirds
    long long share1 = m.total_size, share2 = 0, share3 = 0;
    if (m.n_layer == 2) { share1 = share2 = m.total_size / 2; }
    if (m.n_layer == 3) {
        int p1 = 0, p2 = 0, p3 = 0;
        if (!m.proportion.empty() && sscanf(m.proportion.c_str(), "%d-%d-%d", &p1, &p2, &p3) == 3) {
            if (p1 + p2 + p3 != 100) { set_error("SIZE_PROPORTION must add up to 100"); return EVS_EINVAL; }  // :75 assert
            share1 = p1 * m.total_size / 100; share2 = p2 * m.total_size / 100; share3 = p3 * m.total_size / 100;
        } else {
            share1 = share2 = share3 = m.total_size / 3;
        }
    }
    const long long cap = share1 * (32 / m.main_prec);
    long long cap2 = share2 * (32 / m.secondary_prec);
    if (m.n_layer >= 2 && m.secondary_prec == 8 && m.main_prec > 8) cap2 = share2 * 16;
    m.cap1 = cap; m.cap2 = m.n_layer >= 2 ? cap2 : 0; m.cap3 = (m.n_layer == 3 && share3 * 36 >= 50) ? share3 * 36 : 0;
    // mixed_precs_caching constants: 0.3 / 0.95, n keys flushed, n_perfect -= n (evlfu_8.hpp:50-51, evlfu_8.cpp:256-270)
    long long rows64[kEvTables];
    if (host_engine) {
        rc = evs_hostcache_create(&m.h1, 0, cap, kEvTables, kEvDim, m.main_prec, 0.3, 0.95, 0, 2);
        if (rc) return rc;
        for (int k = 0; k < kEvTables; k++) rows64[k] = m.rows[k];
        rc = evs_hostcache_set_backing(m.h1, m.tables, (const int64_t *)rows64);
        if (rc) return rc;
        if (m.n_layer >= 2) {
            rc = map_tables(m, m.secondary_prec, m.tables2, m.rows2);
            if (rc) return rc;
            rc = evs_hostcache_create(&m.h2, 0, cap2, kEvTables, kEvDim, m.secondary_prec, 0.3, 0.95, 0, 2);
            if (rc) return rc;
            for (int k = 0; k < kEvTables; k++) rows64[k] = m.rows2[k];
            rc = evs_hostcache_set_backing(m.h2, m.tables2, (const int64_t *)rows64);
            if (rc) return rc;
        }
        if (m.cap3) {   // capacity_c3 = share * 36 (one fp32 row = 36 alt keys, evlfu_8.cpp:80)
            const uint32_t *alt[kEvTables];
            for (int k = 0; k < kEvTables; k++) {
                rc = read_altkeys(m, k, m.host_alt[k]);
                if (rc) return rc;
                alt[k] = m.host_alt[k].data();
                rows64[k] = m.alt_rows[k];
            }
            rc = evs_hostaprx_create(&m.h3, m.cap3, kEvTables);
            if (rc) return rc;
            rc = evs_hostaprx_set_altkeys(m.h3, alt, (const int64_t *)rows64);
            if (rc) return rc;
        }
        m.ready = true;
        return EVS_OK;
    }
    rc = evs_cache_create(&m.c1, 0, cap, kEvTables, kEvDim, m.main_prec, 0.3, 0.95, 0, 2);
    if (rc) return rc;
    for (int k = 0; k < kEvTables; k++) rows64[k] = m.rows[k];
    rc = evs_cache_set_backing(m.c1, m.tables, (const int64_t *)rows64);
    if (rc) return rc;
    if (m.n_layer >= 2) {
        rc = load_tables(m, m.secondary_prec, m.tables2, m.rows2);
        if (rc) return rc;
        rc = evs_cache_create(&m.c2, 0, cap2, kEvTables, kEvDim, m.secondary_prec, 0.3, 0.95, 0, 2);
        if (rc) return rc;
        for (int k = 0; k < kEvTables; k++) rows64[k] = m.rows2[k];
        rc = evs_cache_set_backing(m.c2, m.tables2, (const int64_t *)rows64);
        if (rc) return rc;
    }
    if (m.n_layer == 3 && share3 * 36 >= 50) {  // capacity_c3 = share * 36 (one fp32 row = 36 alt keys, evlfu_8.cpp:80)
        // alt-key files: 4-byte BIG-endian words, alt_row*100 + alt_table (script/convert_altkeys_to_binary.py:27-57)
        for (int k = 0; k < kEvTables; k++) {
            std::vector<unsigned> native;
            rc = read_altkeys(m, k, native);
            if (rc) return rc;
            const long long bytes = m.alt_rows[k] * 4;
            void *dev = nullptr;
            EVS_HIP_CHECK(hipMalloc(&dev, bytes > 0 ? bytes : 4));
            EVS_HIP_CHECK(hipMemcpy(dev, native.data(), bytes, hipMemcpyHostToDevice));
            m.alt_tables[k] = dev;
        }
        rc = evs_aprx_create(&m.c3, share3 * 36, kEvTables);
        if (rc) return rc;
        long long ar[kEvTables];
        for (int k = 0; k < kEvTables; k++) ar[k] = m.alt_rows[k];
        rc = evs_aprx_set_altkeys(m.c3, (const uint32_t *const *)m.alt_tables, (const int64_t *)ar);
        if (rc) return rc;
    }
    EVS_HIP_CHECK(hipStreamCreateWithFlags(&m.stream, hipStreamNonBlocking));
    // the 26 ids and the 936 floats of a request cross the bus inside the kernel (mapped pinned buffers):
    // one launch + one sync per ev_lookup, no copy commands
    EVS_HIP_CHECK(hipHostMalloc(&m.h_rows, kEvTables * 4, hipHostMallocMapped));
    EVS_HIP_CHECK(hipHostMalloc(&m.h_out, kEvTables * kEvDim * 4, hipHostMallocMapped));
    EVS_HIP_CHECK(hipHostMalloc(&m.h_hit, kEvTables, hipHostMallocMapped));
    EVS_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void **>(&m.d_rows), m.h_rows, 0));
    EVS_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void **>(&m.d_out), m.h_out, 0));
    EVS_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void **>(&m.d_hit), m.h_hit, 0));
    m.ready = true;
    return EVS_OK;
}

}  // namespace evs

extern "C" int evs_manager_set_altkey_dir(const char *dir) {
    using namespace evs;
    EVS_REQUIRE(!g_mgr.ready && dir, "evs_manager_set_altkey_dir: call before the first lookup");
    g_mgr.altkey_dir = dir;
    return EVS_OK;
}

extern "C" long long evs_manager_tier_capacity(int tier) {  // entries of tier 1 / 2 / 3 (0: absent or not yet configured)
    using namespace evs;
    if (!g_mgr.ready) return 0;
    return tier == 1 ? g_mgr.cap1 : tier == 2 ? g_mgr.cap2 : tier == 3 ? g_mgr.cap3 : 0;
}

extern "C" long long evs_manager_aprx_hit() {  // evlfu_8bit->aprx_ev_hit (cache_manager.cpp:279)
    using namespace evs;
    if (!g_mgr.ready || !(g_mgr.c3 || g_mgr.h3)) return 0;
    int64_t s4[4] = {0};
    if (g_mgr.h3) (void)evs_hostaprx_stats(g_mgr.h3, s4);
    else (void)evs_aprx_stats(g_mgr.c3, s4, g_mgr.stream);
    return s4[1];
}

extern "C" int evs_manager_configure(int n_caching_layer, int main_precision, int secondary_precision,
                                     int64_t total_size, const char *size_proportion, const char *ev_table_root,
                                     int backing) {
    using namespace evs;
    EVS_REQUIRE(!g_mgr.ready, "evs_manager_configure: the cache manager is already initialised");
    EVS_REQUIRE(ev_table_root && *ev_table_root, "evs_manager_configure: ev_table_root is empty");
    EVS_REQUIRE(total_size > 0, "evs_manager_configure: TOTAL_SIZE %lld", (long long)total_size);
    g_mgr.n_layer = n_caching_layer; g_mgr.main_prec = main_precision; g_mgr.secondary_prec = secondary_precision;
    g_mgr.total_size = total_size; g_mgr.proportion = size_proportion ? size_proportion : "";
    g_mgr.root = ev_table_root; g_mgr.backing_kind = backing;
    return ensure_ready();
}

// cache_manager.cpp:231-237.  Reads N_EV_TABLE int32 row ids (table = position), returns the
// library-owned static float[26*36] (valid until the next call; single caller thread).
// On a configuration error the reference prints and exit(-1)s; this returns NULL after printing.
extern "C" float *ev_lookup(int *arr) {
    using namespace evs;
    if (ensure_ready() != EVS_OK) {
        printf("%s\n", evs_last_error());
        return nullptr;
    }
    Manager &m = g_mgr;
    if (m.h1) {   // host engine: straight into the static buffer the caller reads
        unsigned char tier[kEvTables];
        const int hrc = m.h2 ? evs_hostcache_request_c1c2c3(m.h1, m.h2, m.h3, 1, arr, g_emb_weights_in_1d_floats, tier, 23 /* evlfu_8.hpp:70 */)
                             : evs_hostcache_request(m.h1, 1, arr, g_emb_weights_in_1d_floats, tier, -1);
        if (hrc != EVS_OK) {
            printf("%s\n", evs_last_error());
            return nullptr;
        }
        return g_emb_weights_in_1d_floats;
    }
    memcpy(m.h_rows, arr, kEvTables * sizeof(int));
    const int rc = m.c2 ? evs_cache_request_c1c2c3(m.c1, m.c2, m.c3, 1, m.d_rows, m.d_out, m.d_hit, 23 /* evlfu_8.hpp:70 */, m.stream)
                        : evs_cache_request(m.c1, 1, m.d_rows, m.d_out, m.d_hit, -1, m.stream);
    if (rc != EVS_OK) {
        printf("%s\n", evs_last_error());
        return nullptr;
    }
    if (hipStreamSynchronize(m.stream) != hipSuccess) return nullptr;
    memcpy(g_emb_weights_in_1d_floats, m.h_out, sizeof g_emb_weights_in_1d_floats);
    // perfectHit += request_to_*(...) (cache_manager.cpp:179-207): counted on the device (n_perfect_hits)
    return g_emb_weights_in_1d_floats;
}

extern "C" float *get_ev_values(int *) { return evs::g_emb_weights_in_1d_floats; }  // cache_manager.cpp:257-259

extern "C" long long evs_manager_perfect_hit() {
    using namespace evs;
    if (!g_mgr.ready) return 0;
    int64_t s8[8] = {0};
    if ((g_mgr.h1 ? evs_hostcache_stats(g_mgr.h1, s8) : evs_cache_stats(g_mgr.c1, s8, g_mgr.stream)) != EVS_OK) printf("%s\n", evs_last_error());
    return s8[6];
}

extern "C" void print_perfect_hit() {  // cache_manager.cpp:262-290: prints, then resets the counter
    using namespace evs;
    printf("\n[epoll worker] C1_PRECISION    = %d\n", g_mgr.main_prec);
    if (g_mgr.n_layer >= 2) printf("[epoll worker] C2_PRECISION    = %d\n", g_mgr.secondary_prec);
    if (g_mgr.n_layer == 3) {
        printf("[epoll worker] C3 APRX_EV      = ACTIVE\n");
        printf("[epoll worker] SIZE_PROPORTION = %s\n", g_mgr.proportion.c_str());
        printf("[epoll worker] C3 Indiv-Hit    = %lld\n", evs_manager_aprx_hit());
    }
    printf("[epoll worker] TOTAL_SIZE      = %lld\n", g_mgr.total_size);
    printf("[epoll worker] Perfect hit     = %lld\n", evs_manager_perfect_hit());
    fflush(stdout);
    if (g_mgr.ready && g_mgr.h1) (void)evs_hostcache_reset_counters(g_mgr.h1);
    else if (g_mgr.ready) (void)evs_cache_reset_counters(g_mgr.c1, g_mgr.stream);
}

extern "C" int ev_lookup_based_on_list_keys(int *) {  // cache_manager.cpp:239-243: dead in the reference
    printf("ERROR: This ev_lookup_based_on_list_keys() is outdated, better to use ev_lookup() instead!\n");
    return -1;  // the reference exit(-1)s here; the symbol is kept, not the behaviour
}

extern "C" void test_arr(int *arr) {  // cache_manager.cpp:154-168
    for (int i = 0; i < 5; i++) printf("key %d, ", arr[i]);
    printf("\n");
    for (int i = 0; i < 5; i++) printf("vec %d, ", arr[i]);
    printf("\n");
}

// cache_manager.cpp:410-445 start an epoll TCP server on :8080.  Networking is out of scope of
// this build; the symbols exist so a binding that resolves them still loads.
extern "C" void init_global_vars() { printf("INFO: the socket transport is not part of libevstore_hip (use ev_lookup)\n"); }
extern "C" void start_server_threads() { printf("INFO: the socket transport is not part of libevstore_hip (use ev_lookup)\n"); }
