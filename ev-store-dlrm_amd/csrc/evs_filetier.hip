// File-backed miss tier: the reference's mmap miss path (emb_storage/mmap_file_read.py:32-40: ev-table-N.bin mapped
// read-only, row r at byte row_bytes * r; C++ reader pool mixed_precs_caching/evlfu_8.cpp:191-250) for the GPU cache.
//
// Every table file is mmap'ed (PROT_READ, MAP_SHARED).  Tables are then REGISTERED with the GPU (hipHostRegister,
// mapped: the kernels read their rows over the bus, zero-copy, exactly like the PINNED tier) smallest first while the
// running total fits `pinned_budget_bytes` -- registering pins the pages in RAM.  Tables that do not fit stay plain
// mappings backed by the page cache / the disk: their missing rows are STAGED -- the batched lookup hands the host the
// de-duplicated list of new keys, a pool of reader threads copies those rows out of the mappings into a pinned staging
// buffer, and the fill kernel takes them from there (evs_cache.hip: cache_batch_impl, file mode).
#include "evs_common.h"

#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

// The reader pool (round 3): persistent threads, woken per fetch, that take chunks of the key list from a shared counter.
// A missing row is a cold random line of a page-cache mapping: a thread that copies row after row waits ~200 ns for each
// (57 k rows on 16 freshly spawned threads: 0.8 ms of a 0.94 ms batch).  So (a) the threads stay alive between batches,
// (b) work is dealt in chunks from a shared counter, and (c) every thread PREFETCHES the rows kPrefetch keys ahead
// of the one it copies, so that a dozen misses are in flight per thread instead of one.
namespace {
struct ReaderPool {
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    long long generation = 0;
    int running = 0;
    bool stop = false;
    // the job of the current generation
    const evs_filetier *ft = nullptr;
    const uint64_t *keys = nullptr;
    char *dst = nullptr;
    int64_t n = 0;
    uint32_t skip_mask = 0;
    std::atomic<int64_t> next{0};
    static constexpr int64_t kChunk = 256;
    static constexpr int kPrefetch = 12;

    void work();
    void loop() {
        long long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || generation != seen; });
                if (stop) return;
                seen = generation;
            }
            work();
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--running == 0) cv_done.notify_all();
            }
        }
    }
    void start(int n_threads) {
        for (int i = 0; i < n_threads; i++) threads.emplace_back([this] { loop(); });
    }
    std::mutex run_mu;   // one fetch at a time (two caches may share a tier)
    void run(const evs_filetier *ft_, const uint64_t *keys_, char *dst_, int64_t n_, uint32_t skip) {
        std::lock_guard<std::mutex> one(run_mu);
        {
            std::lock_guard<std::mutex> lk(mu);
            ft = ft_; keys = keys_; dst = dst_; n = n_; skip_mask = skip;
            next.store(0, std::memory_order_relaxed);
            running = (int)threads.size();
            generation++;
        }
        cv_work.notify_all();
        work();   // the caller is a reader too
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return running == 0; });
    }
    ~ReaderPool() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv_work.notify_all();
        for (auto &t : threads) t.join();
    }
};
}  // namespace

struct evs_filetier {
    int n_tables = 0;
    long long row_bytes = 0;
    std::vector<int> fd;
    std::vector<void *> map;            // host mapping
    std::vector<long long> bytes, rows;
    std::vector<void *> dev;            // device-visible address of a registered table, else nullptr
    std::vector<char> registered;
    long long pinned_bytes = 0;
    int n_threads = 8;
    ReaderPool *pool = nullptr;         // created by the first fetch that is worth it
    std::mutex pool_mu;                 // ... under this lock (two caches of two threads may share a tier)
};

namespace {
inline const char *row_of(const evs_filetier *ft, uint64_t key, uint32_t skip_mask) {
    const int t = (int)(key >> 32) - 1;
    const long long r = (long long)(key & 0xffffffffull);
    if (t < 0 || t >= ft->n_tables || ((skip_mask >> t) & 1u) || r >= ft->rows[t]) return nullptr;
    return static_cast<const char *>(ft->map[t]) + r * ft->row_bytes;
}
void ReaderPool::work() {
    const long long rb = ft->row_bytes;
    for (;;) {
        const int64_t a = next.fetch_add(kChunk, std::memory_order_relaxed);
        if (a >= n) return;
        const int64_t b = std::min<int64_t>(n, a + kChunk);
        for (int64_t i = a; i < b; i++) {
            if (i + kPrefetch < b) {
                const char *pf = row_of(ft, keys[i + kPrefetch], skip_mask);
                if (pf) { __builtin_prefetch(pf, 0, 0); __builtin_prefetch(pf + 64, 0, 0); if (rb > 128) __builtin_prefetch(pf + rb - 1, 0, 0); }
            }
            const char *src = row_of(ft, keys[i], skip_mask);
            if (src) memcpy(dst + i * rb, src, (size_t)rb);
        }
    }
}
}  // namespace

extern "C" int evs_filetier_close(evs_filetier *ft) {
    if (!ft) return EVS_OK;
    delete ft->pool;
    ft->pool = nullptr;
    for (int k = 0; k < ft->n_tables; k++) {
        if (ft->registered[k] && ft->map[k]) (void)hipHostUnregister(ft->map[k]);
        if (ft->map[k] && ft->bytes[k] > 0) munmap(ft->map[k], (size_t)ft->bytes[k]);
        if (ft->fd[k] >= 0) close(ft->fd[k]);
    }
    delete ft;
    return EVS_OK;
}

extern "C" int evs_filetier_open(evs_filetier **out, int n_tables, const char *const *paths, int64_t row_bytes,
                                 int64_t pinned_budget_bytes) {
    using namespace evs;
    EVS_REQUIRE(out && paths && n_tables >= 1 && n_tables <= 32 && row_bytes > 0, "evs_filetier_open: bad argument");
    evs_filetier *ft = new evs_filetier();
    ft->n_tables = n_tables; ft->row_bytes = row_bytes;
    ft->fd.assign(n_tables, -1); ft->map.assign(n_tables, nullptr); ft->bytes.assign(n_tables, 0);
    ft->rows.assign(n_tables, 0); ft->dev.assign(n_tables, nullptr); ft->registered.assign(n_tables, 0);
    for (int k = 0; k < n_tables; k++) {
        const int fd = open(paths[k], O_RDONLY);
        if (fd < 0) { set_error("evs_filetier_open: cannot open %s", paths[k]); evs_filetier_close(ft); return EVS_EIO; }
        ft->fd[k] = fd;
        struct stat sb;
        if (fstat(fd, &sb) != 0) { set_error("evs_filetier_open: fstat(%s) failed", paths[k]); evs_filetier_close(ft); return EVS_EIO; }
        if (sb.st_size % row_bytes) {
            set_error("evs_filetier_open: %s is not a whole number of %lld-byte rows", paths[k], (long long)row_bytes);
            evs_filetier_close(ft);
            return EVS_EIO;
        }
        ft->bytes[k] = sb.st_size; ft->rows[k] = sb.st_size / row_bytes;
        if (sb.st_size > 0) {
            void *p = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_SHARED, fd, 0);
            if (p == MAP_FAILED) { set_error("evs_filetier_open: mmap(%s) failed", paths[k]); evs_filetier_close(ft); return EVS_EIO; }
            ft->map[k] = p;
        }
    }
    // smallest tables first while they fit the budget (the small tables take most of the lookups of a skewed stream)
    std::vector<int> order(n_tables);
    for (int k = 0; k < n_tables; k++) order[k] = k;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return ft->bytes[a] != ft->bytes[b] ? ft->bytes[a] < ft->bytes[b] : a < b; });
    for (int k : order) {
        if (ft->bytes[k] == 0 || ft->pinned_bytes + ft->bytes[k] > pinned_budget_bytes) continue;
        if (hipHostRegister(ft->map[k], (size_t)ft->bytes[k], hipHostRegisterMapped) != hipSuccess) { (void)hipGetLastError(); continue; }
        void *d = nullptr;
        if (hipHostGetDevicePointer(&d, ft->map[k], 0) != hipSuccess || !d) { (void)hipGetLastError(); (void)hipHostUnregister(ft->map[k]); continue; }
        ft->dev[k] = d; ft->registered[k] = 1; ft->pinned_bytes += ft->bytes[k];
    }
    const unsigned hc = std::thread::hardware_concurrency();
    // readers incl. the calling thread (EVS_FILETIER_THREADS overrides).  Measured on the Kaggle workload, every table staged
    // (tools/file_tier_bench.py, ~57 k new rows per batch, per-batch time): 4 threads 627 us, 8 515, 12 495, 16 476, 24 478,
    // 32 482, 64 561 (waking them costs more than they return) -- against 936 us for 16 threads spawned per fetch without prefetch
    ft->n_threads = hc >= 16 ? 16 : (hc >= 2 ? (int)hc : 2);
    if (const char *e = getenv("EVS_FILETIER_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 256) ft->n_threads = v; }
    *out = ft;
    return EVS_OK;
}

// per table: rows, device-visible address (NULL = staged), registered flag; returns the pinned total in *pinned_bytes
extern "C" int evs_filetier_info(evs_filetier *ft, int64_t *n_rows, const void **dev_ptrs, int *registered, int64_t *pinned_bytes) {
    using namespace evs;
    EVS_REQUIRE(ft, "evs_filetier_info: NULL tier");
    for (int k = 0; k < ft->n_tables; k++) {
        if (n_rows) n_rows[k] = ft->rows[k];
        if (dev_ptrs) dev_ptrs[k] = ft->dev[k];
        if (registered) registered[k] = ft->registered[k];
    }
    if (pinned_bytes) *pinned_bytes = ft->pinned_bytes;
    return EVS_OK;
}

// Reader pool: rows of keys[i] = (table_1based << 32 | row) -> dst + i * row_bytes (host memory).  Keys of tables in
// `skip_mask` (bit k = table k is registered: the GPU reads it itself) and invalid keys are left untouched.
extern "C" int evs_filetier_fetch(evs_filetier *ft, int64_t n, const uint64_t *keys, void *dst, uint32_t skip_mask) {
    using namespace evs;
    EVS_REQUIRE(ft && (n == 0 || (keys && dst)) && n >= 0, "evs_filetier_fetch: bad argument");
    const long long rb = ft->row_bytes;
    if (n < 2048 || ft->n_threads <= 1) {   // not worth waking anybody
        for (int64_t i = 0; i < n; i++) {
            const char *src = row_of(ft, keys[i], skip_mask);
            if (src) memcpy(static_cast<char *>(dst) + i * rb, src, (size_t)rb);
        }
        return EVS_OK;
    }
    ReaderPool *pool;
    {
        std::lock_guard<std::mutex> lk(ft->pool_mu);
        if (!ft->pool) {
            ft->pool = new ReaderPool();
            ft->pool->start(ft->n_threads - 1);
        }
        pool = ft->pool;
    }
    pool->run(ft, keys, static_cast<char *>(dst), n, skip_mask);
    return EVS_OK;
}

// internal accessors for evs_cache.hip
namespace evs {
int filetier_tables(const evs_filetier *ft) { return ft->n_tables; }
long long filetier_row_bytes(const evs_filetier *ft) { return ft->row_bytes; }
long long filetier_rows(const evs_filetier *ft, int k) { return ft->rows[k]; }
const void *filetier_dev(const evs_filetier *ft, int k) { return ft->dev[k]; }
}  // namespace evs
