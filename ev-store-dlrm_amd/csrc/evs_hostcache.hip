// Host engine of the EXACT, one-request-at-a-time cache policies (EvLFU / LRU / LFU, C1 + C2, C1 + C2 + alt keys).
//
// Why this exists (docs/HISTORY.md 3.3): the reference's EVStore loop is batch 1 (--test-mini-batch-size=1) and its policies are
// defined sequentially -- request n sees the inserts of request n-1, and inside a request key i sees the evictions keys
// 0..i-1 caused.  That is a chain of dependent pointer updates over a few megabytes: one wavefront replays it at ~1 us
// per dependent access (cache_exact_kernel: 28 us per request + launch + synchronise = 51 us), one host core walks it
// out of its caches in a few microseconds.  So the per-request form of the cache tier runs HERE, on the host, inside
// libevstore_hip.so, and the GPU keeps what it is good at (the batched snapshot lookups, the gather, the interaction).
// Rows live where the host can read them: the tier's arena in host memory, the miss tier any host-readable mapping of the
// ev-table-N.bin files (mmap / pinned / plain memory) -- the reference's own arrangement (fseek + fread per missing row,
// evlfu_8.cpp:380-414).
//
// This is product code: it shares nothing with oracle/ (the checker) -- own hash map (slots of {key, entry}, linear
// probing, backward-shift delete), own intrusive lists, keys prefetched T at a time so that the T probes, the T entry
// records and the T rows of a request are T overlapping cache misses instead of T dependent ones.
//
// Semantics (every rule cites the reference; the same golden traces the GPU exact kernel is held to run through this
// engine bit for bit -- tests/test_hostcache.py):
//   EvLFU   cache_algo/EvLFU_C1.py:21-166; constants per variant as for evs_cache_create (include/evstore_hip.h)
//   LRU     cache_algo/LRU.py:14-64          LFU  cache_algo/LFU.py:12-95
//   C1 + C2 mixed_precs_caching/evlfu_8.cpp:669-796 (+ evlfu_4.cpp:374-425 as the C2 half)
//   + alt keys  evlfu_8.cpp:474-490,492-667 with aprx_embedding.cpp as the deterministic re-specification of
//           include/evstore_hip.h (evs_aprx_*): evicted keys become visible 50 at a time, at the start of a request.
#include "evs_common.h"

#include <algorithm>
#include <math.h>
#include <new>
#include <stdlib.h>
#include <string.h>
#include <unordered_map>
#include <vector>

namespace evs {
namespace host {

constexpr int kMaxT = 64;
constexpr int kMaxDim = 256;

static inline uint64_t mix(uint64_t x) {   // splitmix64 finaliser
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}

// ---- codecs on the host: the reference's C++ expressions, compiled with -ffp-contract=off -----------------------------
// 8-bit  evlfu_8.cpp:370-378: ((float)u / 254) * 2 - 1 in fp32
// 16-bit evlfu_16.cpp:332-356: float x double products, rounded to float on the store
// 4-bit  evlfu_4.cpp:319-341 + evlfu_4.hpp:46: high nibble first through the 15-entry table (entry 15: NaN, never emitted)
struct Decoders {
    float u8[256];
    float u4[16];
    Decoders() {
        for (int v = 0; v < 256; v++) { float f = (float)v; f = f / 254; f = f * 2; u8[v] = f - 1; }
        const float t[16] = {1.0f, 0.8f, 0.6f, 0.4f, 0.0625f, 0.00390625f, 0.0000153f, 0.0f, -0.0000153f, -0.00390625f,
                             -0.0625f, -0.4f, -0.6f, -0.8f, -1.0f, NAN};
        memcpy(u4, t, sizeof t);
    }
};
static const Decoders &decoders() { static const Decoders d; return d; }

static inline float dec16(uint16_t v) {
    if (v > 65000) {
        const float diff = ((float)(v - 65000)) / 100;
        return (v % 2 == 1) ? (float)(-1 * (0.65 + diff)) : (float)(0.65 + diff);
    }
    return (float)((((float)v) * 0.00002) - 0.65);
}

static void decode_row(const void *raw, int codec, int dim, float *out) {
    switch (codec) {
    case 32: memcpy(out, raw, sizeof(float) * (size_t)dim); break;
    case 16: { const uint16_t *p = static_cast<const uint16_t *>(raw); for (int c = 0; c < dim; c++) out[c] = dec16(p[c]); break; }
    case 8: { const uint8_t *p = static_cast<const uint8_t *>(raw); const float *t = decoders().u8; for (int c = 0; c < dim; c++) out[c] = t[p[c]]; break; }
    default: { const uint8_t *p = static_cast<const uint8_t *>(raw); const float *t = decoders().u4;
               for (int c = 0; c < dim / 2; c++) { out[2 * c] = t[p[c] >> 4]; out[2 * c + 1] = t[p[c] & 15]; } break; }
    }
}

// ---- key -> entry map ---------------------------------------------------------------------------------------------------
struct Slot { uint64_t key; int32_t entry; int32_t pad; };   // key 0 = empty (table ids are 1-based: no key is 0)

struct FlatMap {
    Slot *s = nullptr;
    uint64_t mask = 0;
    int64_t count = 0;
    bool init(int64_t cap) {
        uint64_t n = 64;
        while ((int64_t)n < 2 * cap + 16) n <<= 1;   // load <= 0.5
        s = static_cast<Slot *>(calloc(n, sizeof(Slot)));
        mask = n - 1;
        return s != nullptr;
    }
    void destroy() { free(s); s = nullptr; }
    inline uint64_t home(uint64_t key) const { return mix(key) & mask; }
    inline void prefetch(uint64_t key) const { __builtin_prefetch(&s[home(key)], 0, 3); }
    inline int32_t find(uint64_t key) const {
        for (uint64_t i = home(key);; i = (i + 1) & mask) {
            if (s[i].key == key) return s[i].entry;
            if (!s[i].key) return -1;
        }
    }
    inline void insert(uint64_t key, int32_t e) {   // the key is known to be absent
        uint64_t i = home(key);
        while (s[i].key) i = (i + 1) & mask;
        s[i].key = key; s[i].entry = e;
        count++;
    }
    void erase(uint64_t key) {
        uint64_t i = home(key);
        while (s[i].key != key) {
            if (!s[i].key) return;
            i = (i + 1) & mask;
        }
        count--;
        // close the gap: pull back every later member of the cluster whose home slot is not inside (i, j]
        for (uint64_t j = (i + 1) & mask; s[j].key; j = (j + 1) & mask) {
            const uint64_t h = home(s[j].key);
            const bool stays = ((j - h) & mask) < ((j - i) & mask);   // distance from its home is shorter than to the hole
            if (!stays) { s[i] = s[j]; i = j; }
        }
        s[i].key = 0;
    }
};

struct List { int32_t head = -1, tail = -1; int64_t len = 0; };

// ---- one tier -------------------------------------------------------------------------------------------------------------
struct Entry { uint64_t key; int32_t prev, next; int64_t prio; };   // prio: EvLFU bucket (agg_hit) | LFU frequency | unused

struct Tier {
    int policy = 0;   // 0 EvLFU, 1 LRU, 2 LFU
    int64_t cap = 0;
    int T = 0, dim = 0, codec = 32, row_bytes = 0;
    // EvLFU state (EvLFU_C1.py:21-31)
    int min_c1 = 0;
    int64_t n_perfect = 0, max_perfect = 0;
    double flush_rate = 0.3;
    int flush_extra = 1, perfect_mode = 0;
    int64_t least_freq = 1;   // LFU.py:16
    // counters (what evs_cache_stats reports for the GPU tier)
    int64_t n_flush = 0, n_evict = 0, n_requests = 0, n_perfect_hits = 0, n_hits = 0;
    int error = 0;
    uint64_t last_evicted = 0;   // key evicted (not flushed) by the latest insert: the alt-key tier's feed
    FlatMap map;
    Entry *ent = nullptr;
    uint8_t *arena = nullptr;   // cap rows in the tier's codec
    int32_t *free_stack = nullptr;
    int64_t n_free = 0;
    List buckets[kMaxT + 1];                      // EvLFU priority FIFOs 0..T; LRU: buckets[0] is the recency order
    std::unordered_map<int64_t, List> freq;       // LFU: FIFO per frequency
    const uint8_t *tables[kMaxT] = {nullptr};
    int64_t n_rows[kMaxT] = {0};
    bool bound = false;

    ~Tier() { map.destroy(); free(ent); free(arena); free(free_stack); }

    inline uint8_t *row_of(int32_t e) { return arena + (int64_t)e * row_bytes; }
    inline const uint8_t *backing_row(int table0, int64_t row) const { return tables[table0] + row * row_bytes; }

    inline void push_back(List &l, int32_t e) {
        ent[e].prev = l.tail; ent[e].next = -1;
        if (l.tail >= 0) ent[l.tail].next = e; else l.head = e;
        l.tail = e; l.len++;
    }
    inline void unlink(List &l, int32_t e) {
        const int32_t p = ent[e].prev, n = ent[e].next;
        if (p >= 0) ent[p].next = n; else l.head = n;
        if (n >= 0) ent[n].prev = p; else l.tail = p;
        l.len--;
    }
    inline void release(int32_t e) {   // the key leaves the tier
        map.erase(ent[e].key);
        ent[e].key = 0;
        free_stack[n_free++] = e;
    }
    inline int32_t take(uint64_t key, const uint8_t *raw) {
        const int32_t e = free_stack[--n_free];
        ent[e].key = key;
        memcpy(row_of(e), raw, (size_t)row_bytes);
        map.insert(key, e);
        return e;
    }

    // update_agg_hit (EvLFU_C1.py:65-78, evlfu_8.cpp:303-321): monotone max, a move goes to the TAIL of the new bucket
    inline void touch(int32_t e, int agg) {
        if (ent[e].prio < agg) {
            unlink(buckets[ent[e].prio], e);
            push_back(buckets[agg], e);
            ent[e].prio = agg;
        }
    }

    // set() of EvLFU_C1.py:32-63 (evlfu_8.cpp:252-300): flush | evict, then insert into bucket agg
    int evlfu_insert(uint64_t key, const uint8_t *raw, int agg) {
        last_evicted = 0;
        if (n_perfect >= max_perfect) {   // :36-44 the top bucket has grown to its cap: its oldest keys go
            const int64_t n = (int64_t)(flush_rate * cap) + flush_extra;
            List &top = buckets[T];
            for (int64_t i = 0; i < n; i++) {
                const int32_t e = top.head;
                if (e < 0) {
                    if (perfect_mode == 0) return -1;   // Python: pop(0) on an empty list raises
                    break;                               // C++ (evlfu_8.cpp:256-270): stops at the end of the bucket
                }
                unlink(top, e);
                release(e);
            }
            n_perfect = perfect_mode == 0 ? top.len : n_perfect - (int64_t)(flush_rate * cap);
            n_flush++;
        } else if (map.count >= cap) {    // :47-56 FIFO-oldest key of the lowest non-empty bucket, the scan wraps T -> 1
            while (buckets[min_c1].len == 0) {
                min_c1++;
                if (min_c1 > T) min_c1 = 1;
            }
            const int32_t e = buckets[min_c1].head;
            unlink(buckets[min_c1], e);
            last_evicted = ent[e].key;
            release(e);
            n_evict++;
        }
        if (n_free <= 0) return -2;
        const int32_t e = take(key, raw);
        ent[e].prio = agg;
        push_back(buckets[agg], e);
        if (agg < min_c1) min_c1 = agg;   // :62-63
        return 0;
    }
};

static inline uint64_t make_key(int table0, int32_t row) { return ((uint64_t)(table0 + 1) << 32) | (uint32_t)row; }

// One EvLFU request (EvLFU_C1.py:97-166).  Three passes over the T keys: hash + prefetch the home slots; probe + prefetch
// what each key will touch (entry record and arena row of a hit, backing row of a miss); then the policy in table order.
static int evlfu_request(Tier &c, const int32_t *rows, uint8_t *hit_out, float *out, int approx_thres) {
    const int T = c.T, dim = c.dim;
    uint64_t key[kMaxT];
    int32_t e0[kMaxT];
    uint8_t hit[kMaxT];
    for (int i = 0; i < T; i++) {
        if (rows[i] < 0 || rows[i] >= c.n_rows[i]) { set_error("evs_hostcache_request: row %d of table %d is out of range", rows[i], i + 1); return EVS_EINDEX; }
        key[i] = make_key(i, rows[i]);
        c.map.prefetch(key[i]);
    }
    int agg = 0;
    for (int i = 0; i < T; i++) {   // :105-120 agg_hit = keys present when the request starts
        e0[i] = c.map.find(key[i]);
        hit[i] = e0[i] >= 0;
        agg += hit[i];
        if (hit[i]) { __builtin_prefetch(&c.ent[e0[i]], 1, 3); __builtin_prefetch(c.row_of(e0[i]), 0, 3); }
    }
    const bool approx = approx_thres > 0 && agg >= approx_thres;   // :122-125 misses are not fetched
    if (!approx)
        for (int i = 0; i < T; i++)
            if (!hit[i]) {
                const uint8_t *p = c.backing_row(i, rows[i]);
                __builtin_prefetch(p, 0, 3);
                __builtin_prefetch(p + c.row_bytes - 1, 0, 3);
            }
    const float *last_val = nullptr;   // :141 the vector of the latest hit
    for (int i = 0; i < T; i++) {      // :135-161
        float *o = out + (int64_t)i * dim;
        if (hit[i]) {
            // the entry found by the probe is still this key's unless an insert of THIS request evicted it
            int32_t e = (c.ent[e0[i]].key == key[i]) ? e0[i] : c.map.find(key[i]);
            if (e >= 0) {
                c.touch(e, agg);
                decode_row(c.row_of(e), c.codec, dim, o);
            } else {   // :90-94 kicked out while a previous key of the request was inserted: fetch and set again
                const uint8_t *raw = c.backing_row(i, rows[i]);
                const int rc = c.evlfu_insert(key[i], raw, agg);
                if (rc) return rc - 10;
                decode_row(raw, c.codec, dim, o);
            }
            last_val = o;
        } else if (approx) {   // :142-152 served from the previous hit and REPORTED as a hit
            if (last_val) memcpy(o, last_val, sizeof(float) * (size_t)dim);
            else memset(o, 0, sizeof(float) * (size_t)dim);   // (the reference draws 36 random numbers here)
            hit[i] = 1;
        } else {
            const int32_t e = c.map.find(key[i]);   // (present only if the same key came earlier in this request)
            if (e >= 0) {
                c.touch(e, agg);
                decode_row(c.row_of(e), c.codec, dim, o);
            } else {
                const uint8_t *raw = c.backing_row(i, rows[i]);
                const int rc = c.evlfu_insert(key[i], raw, agg);
                if (rc) return rc - 20;
                decode_row(raw, c.codec, dim, o);
            }
        }
    }
    if (agg == T) c.n_perfect = c.buckets[T].len;   // :163-165
    int reported = 0;
    for (int i = 0; i < T; i++) reported += hit[i];
    c.n_requests++; c.n_hits += reported; c.n_perfect_hits += reported == T;
    if (hit_out) memcpy(hit_out, hit, (size_t)T);
    return 0;
}

// LRU.py:14-64: per key in table order; a hit moves to the MRU end, a miss evicts the LRU head when full
static int lru_request(Tier &c, const int32_t *rows, uint8_t *hit_out, float *out) {
    const int T = c.T, dim = c.dim;
    uint64_t key[kMaxT];
    for (int i = 0; i < T; i++) {
        if (rows[i] < 0 || rows[i] >= c.n_rows[i]) { set_error("evs_hostcache_request: row %d of table %d is out of range", rows[i], i + 1); return EVS_EINDEX; }
        key[i] = make_key(i, rows[i]);
        c.map.prefetch(key[i]);
    }
    List &order = c.buckets[0];
    int agg = 0;
    for (int i = 0; i < T; i++) {
        float *o = out + (int64_t)i * dim;
        int32_t e = c.map.find(key[i]);
        if (e >= 0) {   // :24-28
            c.unlink(order, e);
            c.push_back(order, e);
            decode_row(c.row_of(e), c.codec, dim, o);
            hit_out[i] = 1; agg++;
        } else {        // :30-34 -> set() :14-20
            const uint8_t *raw = c.backing_row(i, rows[i]);
            if (c.map.count >= c.cap) {
                const int32_t v = order.head;
                c.unlink(order, v);
                c.release(v);
                c.n_evict++;
            }
            e = c.take(key[i], raw);
            c.push_back(order, e);
            decode_row(raw, c.codec, dim, o);
            hit_out[i] = 0;
        }
    }
    c.n_requests++; c.n_hits += agg; c.n_perfect_hits += agg == T;
    return 0;
}

// LFU.py:12-95: FIFO within a frequency; the reference's least_freq bookkeeping is kept as written (it only ever steps by
// one after a hit, :27-28, and resets to 1 on an insert, :50)
static int lfu_request(Tier &c, const int32_t *rows, uint8_t *hit_out, float *out) {
    const int T = c.T, dim = c.dim;
    uint64_t key[kMaxT];
    for (int i = 0; i < T; i++) {
        if (rows[i] < 0 || rows[i] >= c.n_rows[i]) { set_error("evs_hostcache_request: row %d of table %d is out of range", rows[i], i + 1); return EVS_EINDEX; }
        key[i] = make_key(i, rows[i]);
        c.map.prefetch(key[i]);
    }
    int agg = 0;
    for (int i = 0; i < T; i++) {
        float *o = out + (int64_t)i * dim;
        int32_t e = c.map.find(key[i]);
        if (e >= 0) {   // :53-60 -> _update :19-34
            const int64_t f = c.ent[e].prio;
            auto it = c.freq.find(f);
            c.unlink(it->second, e);
            if (it->second.len == 0) c.freq.erase(it);   // (an absent list is an empty one)
            if (c.freq.find(c.least_freq) == c.freq.end()) c.least_freq += 1;   // :25-28 steps by ONE, whatever lies above
            c.ent[e].prio = f + 1;
            c.push_back(c.freq[f + 1], e);
            decode_row(c.row_of(e), c.codec, dim, o);
            hit_out[i] = 1; agg++;
        } else {        // :61-65 -> set() :36-51
            const uint8_t *raw = c.backing_row(i, rows[i]);
            if (c.map.count >= c.cap) {
                auto it = c.freq.find(c.least_freq);
                if (it == c.freq.end() || it->second.head < 0) return -1;   // Python: popitem on an empty OrderedDict raises
                const int32_t v = it->second.head;
                c.unlink(it->second, v);
                if (it->second.len == 0) c.freq.erase(it);
                c.release(v);
                c.n_evict++;
            }
            e = c.take(key[i], raw);
            c.ent[e].prio = 1;
            c.push_back(c.freq[1], e);
            c.least_freq = 1;
            decode_row(raw, c.codec, dim, o);
            hit_out[i] = 0;
        }
    }
    c.n_requests++; c.n_hits += agg; c.n_perfect_hits += agg == T;
    return 0;
}

// ---- the alt-key tier (aprx_embedding.cpp; semantics: include/evstore_hip.h "a12") -------------------------------------
constexpr int kAprxBatch = 50;   // IO_JOB_Q_SIZE (aprx_embedding.hpp)

struct AltSlot { uint32_t alt; uint8_t recent; };

struct AltKeys {
    int64_t cap = 0, n_hit = 0;
    int T = 0;
    int error = 0;
    std::unordered_map<uint64_t, AltSlot> members;   // key -> {alt key, second-chance flag}
    std::vector<uint64_t> fifo;                       // ring buffer; may hold stale duplicates of a key (as the reference's queue does)
    int64_t qhead = 0, qtail = 0;
    uint64_t pending[kAprxBatch];
    int n_pending = 0;
    bool batch_ready = false;
    const uint32_t *alt_tables[kMaxT] = {nullptr};
    int64_t alt_rows[kMaxT] = {0};

    inline uint32_t alt_of(uint64_t key) const { return alt_tables[(int)(key >> 32) - 1][(uint32_t)key]; }
    inline bool push(uint64_t key) {
        if (qtail - qhead >= (int64_t)fifo.size()) {
            // full: the reference's std::queue is unbounded, and no entry may be dropped -- a duplicate of a key, even of one
            // that is not a member right now, is that key's turn again if it comes back before the entry is popped -- so the
            // ring doubles, order kept
            std::vector<uint64_t> grown(fifo.size() * 2);
            const int64_t n = qtail - qhead;
            for (int64_t i = 0; i < n; i++) grown[i] = fifo[(qhead + i) % fifo.size()];
            fifo.swap(grown);
            qhead = 0; qtail = n;
        }
        fifo[qtail++ % fifo.size()] = key;
        return true;
    }
    // recency_aware_eviction (aprx_embedding.cpp:360-388): a flagged front key loses its flag and goes to the back
    void evict_one() {
        while (qhead < qtail) {
            const uint64_t key = fifo[qhead++ % fifo.size()];
            auto it = members.find(key);
            if (it == members.end()) continue;   // stale duplicate
            if (it->second.recent) { it->second.recent = 0; push(key); }
            else { members.erase(it); return; }
        }
    }
    void admit(uint64_t key, uint32_t alt) {   // push on the FIFO (duplicates allowed), map[key] = {alt, false}
        if (!push(key)) return;
        members[key] = AltSlot{alt, 0};
    }
    // insert_altkey_batched_obj (:308-324) as re-specified: room for the whole batch first, then the 50 keys
    void insert_batch() {
        const int64_t n_erase = (int64_t)members.size() + kAprxBatch - cap;
        for (int64_t i = 0; i < n_erase; i++) evict_one();
        for (int i = 0; i < kAprxBatch && !error; i++) {
            if (members.find(pending[i]) == members.end() && (int64_t)members.size() >= cap) { error = 2; break; }
            admit(pending[i], alt_of(pending[i]));
        }
        n_pending = 0; batch_ready = false;
    }
    void queue_evicted(uint64_t key) {   // evlfu_8.cpp:284-287,617-620,654-658: only EVICTED keys, flushed ones are not queued
        if (!key) return;
        if (n_pending < kAprxBatch) pending[n_pending++] = key;
        if (n_pending == kAprxBatch) batch_ready = true;
    }
};

// request_to_c1_c2 / request_to_c1_c2_c3 (evlfu_8.cpp:669-796 / :492-667); c3 == nullptr: two tiers
static int tiers_request(Tier &c1, Tier &c2, AltKeys *c3, const int32_t *rows, uint8_t *tier_out, float *out, int high_thr) {
    const int T = c1.T, dim = c1.dim;
    uint64_t key[kMaxT];
    int32_t e1[kMaxT], e2[kMaxT], alt_e[kMaxT];
    bool hit1[kMaxT], hit2[kMaxT], hit3[kMaxT], upd2[kMaxT], ins2[kMaxT], job1[kMaxT];
    uint8_t alt_tier[kMaxT];
    if (c3 && c3->batch_ready) c3->insert_batch();
    for (int i = 0; i < T; i++) {
        if (rows[i] < 0 || rows[i] >= c1.n_rows[i] || rows[i] >= c2.n_rows[i]) { set_error("evs_hostcache_request_c1c2c3: row %d of table %d is out of range", rows[i], i + 1); return EVS_EINDEX; }
        key[i] = make_key(i, rows[i]);
        c2.map.prefetch(key[i]);
        c1.map.prefetch(key[i]);
    }
    int c1_agg = 0, agg = 0;
    for (int i = 0; i < T; i++) {   // evlfu_4.cpp phase_1_find_keys_in_cache: C2 first
        e2[i] = c2.map.find(key[i]);
        hit2[i] = e2[i] >= 0;
        agg += hit2[i];
        if (hit2[i]) { __builtin_prefetch(&c2.ent[e2[i]], 1, 3); __builtin_prefetch(c2.row_of(e2[i]), 0, 3); }
    }
    for (int i = 0; i < T; i++) {   // evlfu_8.cpp:516-561 / :690-712
        e1[i] = c1.map.find(key[i]);
        hit1[i] = e1[i] >= 0;
        hit3[i] = false; upd2[i] = true; ins2[i] = false; job1[i] = false; alt_tier[i] = 0; alt_e[i] = -1;
        if (hit1[i]) {
            __builtin_prefetch(&c1.ent[e1[i]], 1, 3); __builtin_prefetch(c1.row_of(e1[i]), 0, 3);
            c1_agg++;
            upd2[i] = false;
            if (!hit2[i]) agg++;
        } else if (!hit2[i]) {
            upd2[i] = false;
            if (c3) {   // find_approximate_ev (evlfu_8.cpp:474-490): the key's alt row, if some tier holds it
                auto it = c3->members.find(key[i]);
                if (it != c3->members.end()) {
                    const uint32_t alt = it->second.alt;
                    const uint64_t akey = ((uint64_t)(alt % 100) << 32) | (uint32_t)(alt / 100);
                    int32_t ea = c1.map.find(akey);
                    if (ea >= 0) { alt_tier[i] = 1; alt_e[i] = ea; }
                    else if ((ea = c2.map.find(akey)) >= 0) { alt_tier[i] = 2; alt_e[i] = ea; }
                    if (alt_tier[i]) {
                        hit3[i] = true;
                        it->second.recent = 1;   // set_recency_flag_c3
                        c3->n_hit++;
                        agg++;
                    }
                }
            }
            if (!hit3[i]) ins2[i] = true;
        }
    }
    bool update_c2 = true;
    if (c1.map.count >= c1.cap) {   // :570-601 / :721-738 C1 full: below the threshold the odd positions' double misses go to C1
        if (agg < high_thr)
            for (int i = 0; i < T; i++)
                if (!hit2[i]) {
                    upd2[i] = false;
                    if (i % 2 == 1) { job1[i] = !(hit1[i] || hit3[i]); ins2[i] = false; }
                }
    } else {                        // :739-751 C1 still filling: every C1 miss goes to C1, C2 is left alone
        for (int i = 0; i < T; i++) if (!(hit1[i] || hit3[i])) job1[i] = true;
        update_c2 = false;
        agg = c1_agg;
    }
    for (int i = 0; i < T; i++) tier_out[i] = hit1[i] ? 1 : (hit2[i] ? 2 : (hit3[i] ? 3 : 0));
    for (int i = 0; i < T; i++) {   // prefetch the backing rows this request will fetch
        if (ins2[i] && update_c2) __builtin_prefetch(c2.backing_row(i, rows[i]), 0, 3);
        if (job1[i]) __builtin_prefetch(c1.backing_row(i, rows[i]), 0, 3);
    }
    // alt rows are read before anything moves: the request's own evictions cannot invalidate what it serves; decoded at
    // the precision of the tier that holds the row
    for (int i = 0; i < T; i++)
        if (hit3[i]) {
            Tier &src = alt_tier[i] == 1 ? c1 : c2;
            decode_row(src.row_of(alt_e[i]), src.codec, dim, out + (int64_t)i * dim);
        }
    if (update_c2) {   // evlfu_4.cpp phase_2_get_and_insert_missing_values: hits first, then the inserts
        for (int i = 0; i < T; i++)
            if (upd2[i] && hit2[i]) {
                c2.touch(e2[i], agg);
                decode_row(c2.row_of(e2[i]), c2.codec, dim, out + (int64_t)i * dim);
            }
        for (int i = 0; i < T; i++)
            if (ins2[i]) {
                const uint8_t *raw = c2.backing_row(i, rows[i]);
                const int rc = c2.evlfu_insert(key[i], raw, agg);
                if (rc) return rc - 30;
                if (c3) c3->queue_evicted(c2.last_evicted);
                decode_row(raw, c2.codec, dim, out + (int64_t)i * dim);
            }
        if (agg == T) c2.n_perfect = c2.buckets[T].len;
    }
    for (int i = 0; i < T; i++) {   // evlfu_8.cpp:629-652 / :769-785
        float *o = out + (int64_t)i * dim;
        if (hit1[i]) {
            const int32_t e = (c1.ent[e1[i]].key == key[i]) ? e1[i] : c1.map.find(key[i]);
            if (e >= 0) {
                c1.touch(e, agg);
                decode_row(c1.row_of(e), c1.codec, dim, o);
            } else {   // evicted earlier in this request (the C++ reads a dangling pointer here, evlfu_8.cpp:521-522): from storage
                decode_row(c1.backing_row(i, rows[i]), c1.codec, dim, o);
            }
        } else if (job1[i]) {
            const uint8_t *raw = c1.backing_row(i, rows[i]);
            const int rc = c1.evlfu_insert(key[i], raw, agg);
            if (rc) return rc - 40;
            if (c3) c3->queue_evicted(c1.last_evicted);
            decode_row(raw, c1.codec, dim, o);
        }
    }
    const bool perfect = agg == T;
    if (perfect) c1.n_perfect = c1.buckets[T].len;
    int served = 0;
    for (int i = 0; i < T; i++) served += tier_out[i] != 0;
    c1.n_requests++; c1.n_hits += served; c1.n_perfect_hits += perfect;   // perfectHit += request_to_*() (cache_manager.cpp:179-207)
    return 0;
}

}  // namespace host
}  // namespace evs

struct evs_hostcache { evs::host::Tier t; };
struct evs_hostaprx { evs::host::AltKeys a; };

extern "C" int evs_hostcache_create(evs_hostcache **out, int policy, int64_t capacity, int n_tables, int dim, int codec,
                                    double flush_rate, double perfect_item_cap, int flush_extra, int perfect_mode) {
    using namespace evs;
    EVS_REQUIRE(out, "evs_hostcache_create: NULL out");
    EVS_REQUIRE(policy >= 0 && policy <= 2, "evs_hostcache_create: policy %d (0 EvLFU, 1 LRU, 2 LFU)", policy);
    EVS_REQUIRE(capacity >= 1 && capacity < (1ll << 31), "evs_hostcache_create: capacity %lld", (long long)capacity);
    EVS_REQUIRE(n_tables >= 1 && n_tables <= host::kMaxT, "evs_hostcache_create: n_tables %d (1..%d)", n_tables, host::kMaxT);
    EVS_REQUIRE(dim >= 1 && dim <= host::kMaxDim && (codec != 4 || dim % 2 == 0), "evs_hostcache_create: dim %d", dim);
    EVS_REQUIRE(codec == 32 || codec == 16 || codec == 8 || codec == 4, "evs_hostcache_create: codec %d", codec);
    evs_hostcache *h = new (std::nothrow) evs_hostcache;
    if (!h) return EVS_ENOMEM;
    host::Tier &t = h->t;
    t.policy = policy; t.cap = capacity; t.T = n_tables; t.dim = dim; t.codec = codec; t.row_bytes = dim * codec / 8;
    t.flush_rate = flush_rate; t.flush_extra = flush_extra; t.perfect_mode = perfect_mode;
    t.max_perfect = (int64_t)(capacity * perfect_item_cap);   // EvLFU_C1.py:30
    t.ent = static_cast<host::Entry *>(malloc(sizeof(host::Entry) * capacity));
    t.arena = static_cast<uint8_t *>(malloc((size_t)capacity * t.row_bytes));
    t.free_stack = static_cast<int32_t *>(malloc(sizeof(int32_t) * capacity));
    if (!t.ent || !t.arena || !t.free_stack || !t.map.init(capacity)) {
        delete h;
        set_error("evs_hostcache_create: out of memory (capacity %lld)", (long long)capacity);
        return EVS_ENOMEM;
    }
    for (int64_t i = 0; i < capacity; i++) { t.free_stack[i] = (int32_t)(capacity - 1 - i); t.ent[i].key = 0; }
    t.n_free = capacity;
    *out = h;
    return EVS_OK;
}

extern "C" int evs_hostcache_destroy(evs_hostcache *c) { delete c; return EVS_OK; }

extern "C" int evs_hostcache_set_backing(evs_hostcache *c, const void *const *tables, const int64_t *n_rows) {
    using namespace evs;
    EVS_REQUIRE(c && tables && n_rows, "evs_hostcache_set_backing: NULL argument");
    for (int k = 0; k < c->t.T; k++) {
        EVS_REQUIRE(tables[k] || n_rows[k] == 0, "evs_hostcache_set_backing: table %d is NULL", k + 1);
        c->t.tables[k] = static_cast<const uint8_t *>(tables[k]);
        c->t.n_rows[k] = n_rows[k];
    }
    c->t.bound = true;
    return EVS_OK;
}

extern "C" int evs_hostcache_request(evs_hostcache *c, int64_t B, const int32_t *rows, float *out, uint8_t *hit, int approx_thres) {
    using namespace evs;
    EVS_REQUIRE(c && rows && out && hit, "evs_hostcache_request: NULL argument");
    if (!c->t.bound) { set_error("evs_hostcache_request: no backing tables (evs_hostcache_set_backing)"); return EVS_ESTATE; }
    if (c->t.error) { set_error("evs_hostcache_request: the policy hit an inconsistency earlier (%d)", c->t.error); return EVS_ESTATE; }
    host::Tier &t = c->t;
    for (int64_t b = 0; b < B; b++) {
        const int32_t *r = rows + b * t.T;
        float *o = out + b * (int64_t)t.T * t.dim;
        uint8_t *h = hit + b * t.T;
        const int rc = t.policy == 0 ? host::evlfu_request(t, r, h, o, approx_thres)
                     : t.policy == 1 ? host::lru_request(t, r, h, o) : host::lfu_request(t, r, h, o);
        if (rc == EVS_EINDEX) return rc;
        if (rc) { t.error = rc; set_error("evs_hostcache_request: policy inconsistency %d at request %lld", rc, (long long)b); return EVS_ESTATE; }
    }
    return EVS_OK;
}

extern "C" int evs_hostcache_request_c1c2c3(evs_hostcache *c1, evs_hostcache *c2, evs_hostaprx *c3, int64_t B, const int32_t *rows,
                                            float *out, uint8_t *tier, int high_agghit_threshold) {
    using namespace evs;
    EVS_REQUIRE(c1 && c2 && rows && out && tier, "evs_hostcache_request_c1c2c3: NULL argument");
    EVS_REQUIRE(c1->t.policy == 0 && c2->t.policy == 0 && c1->t.T == c2->t.T && c1->t.dim == c2->t.dim,
                "evs_hostcache_request_c1c2c3: both tiers must be EvLFU tiers of the same shape");
    EVS_REQUIRE(!c3 || c3->a.T == c1->t.T, "evs_hostcache_request_c1c2c3: the alt-key tier has another table count");
    if (!c1->t.bound || !c2->t.bound) { set_error("evs_hostcache_request_c1c2c3: no backing tables"); return EVS_ESTATE; }
    if (c3) {   // alt_of() indexes alt_tables[t][row] with rows of the tiers' tables: every table must be there and long enough
        for (int k = 0; k < c1->t.T; k++) {
            const int64_t need = c1->t.n_rows[k] > c2->t.n_rows[k] ? c1->t.n_rows[k] : c2->t.n_rows[k];
            if (need > 0 && !c3->a.alt_tables[k]) { set_error("evs_hostcache_request_c1c2c3: call evs_hostaprx_set_altkeys first (alt-key table %d is missing)", k + 1); return EVS_ESTATE; }
            if (c3->a.alt_rows[k] < need) { set_error("evs_hostcache_request_c1c2c3: alt-key table %d has %lld rows, the embedding table %lld", k + 1, (long long)c3->a.alt_rows[k], (long long)need); return EVS_EINVAL; }
        }
    }
    if (c1->t.error || c2->t.error) { set_error("evs_hostcache_request_c1c2c3: the policy hit an inconsistency earlier"); return EVS_ESTATE; }
    for (int64_t b = 0; b < B; b++) {
        const int rc = host::tiers_request(c1->t, c2->t, c3 ? &c3->a : nullptr, rows + b * c1->t.T, tier + b * c1->t.T,
                                           out + b * (int64_t)c1->t.T * c1->t.dim, high_agghit_threshold);
        if (rc == EVS_EINDEX) return rc;
        if (rc) { c1->t.error = rc; set_error("evs_hostcache_request_c1c2c3: policy inconsistency %d at request %lld", rc, (long long)b); return EVS_ESTATE; }
        if (c3 && c3->a.error) { set_error("evs_hostcache_request_c1c2c3: alt-key tier error %d", c3->a.error); return EVS_ESTATE; }
    }
    return EVS_OK;
}

extern "C" int evs_hostcache_stats(evs_hostcache *c, int64_t *out8) {
    using namespace evs;
    EVS_REQUIRE(c && out8, "evs_hostcache_stats: NULL argument");
    const host::Tier &t = c->t;
    out8[0] = t.policy == 2 ? t.least_freq : t.min_c1; out8[1] = t.n_perfect; out8[2] = t.map.count; out8[3] = t.n_flush; out8[4] = t.n_evict;
    out8[5] = t.n_requests; out8[6] = t.n_perfect_hits; out8[7] = t.n_hits;
    if (t.error) { set_error("evs_hostcache_stats: the policy hit an inconsistency (%d)", t.error); return EVS_ESTATE; }
    return EVS_OK;
}

extern "C" int evs_hostcache_reset_counters(evs_hostcache *c) {
    using namespace evs;
    EVS_REQUIRE(c, "evs_hostcache_reset_counters: NULL");
    c->t.n_requests = c->t.n_perfect_hits = c->t.n_hits = 0;
    return EVS_OK;
}

extern "C" int64_t evs_hostcache_dump(evs_hostcache *c, int64_t *triples, int64_t max_triples) {
    using namespace evs;
    if (!c) { set_error("evs_hostcache_dump: NULL"); return EVS_EINVAL; }
    const host::Tier &t = c->t;
    int64_t n = 0;
    auto emit = [&](int64_t tag, const host::List &l) {
        for (int32_t e = l.head; e >= 0; e = t.ent[e].next, n++)
            if (triples && n < max_triples) {
                triples[3 * n] = tag; triples[3 * n + 1] = (int64_t)(t.ent[e].key >> 32); triples[3 * n + 2] = (int64_t)(t.ent[e].key & 0xffffffffu);
            }
    };
    if (t.policy == 0) { for (int b = 0; b <= t.T; b++) emit(b, t.buckets[b]); }
    else if (t.policy == 1) emit(0, t.buckets[0]);
    else {
        std::vector<int64_t> fs;
        for (const auto &kv : t.freq) fs.push_back(kv.first);
        std::sort(fs.begin(), fs.end());
        for (int64_t f : fs) emit(f, t.freq.at(f));
    }
    return n;
}

extern "C" int evs_hostaprx_create(evs_hostaprx **out, int64_t capacity, int n_tables) {
    using namespace evs;
    EVS_REQUIRE(out && n_tables >= 1 && n_tables <= host::kMaxT, "evs_hostaprx_create: bad argument");
    EVS_REQUIRE(capacity >= host::kAprxBatch, "evs_hostaprx_create: capacity %lld < %d (aprx_embedding.cpp:33 asserts cap_C3 >= IO_JOB_Q_SIZE)",
                (long long)capacity, host::kAprxBatch);
    evs_hostaprx *p = new (std::nothrow) evs_hostaprx;
    if (!p) return EVS_ENOMEM;
    p->a.cap = capacity; p->a.T = n_tables;
    p->a.fifo.assign((size_t)(4 * capacity + 64), 0);
    p->a.members.reserve((size_t)capacity + host::kAprxBatch);
    *out = p;
    return EVS_OK;
}
extern "C" int evs_hostaprx_destroy(evs_hostaprx *p) { delete p; return EVS_OK; }
extern "C" int evs_hostaprx_set_altkeys(evs_hostaprx *p, const uint32_t *const *alt_tables, const int64_t *n_rows) {
    using namespace evs;
    EVS_REQUIRE(p && alt_tables && n_rows, "evs_hostaprx_set_altkeys: NULL argument");
    for (int k = 0; k < p->a.T; k++) { p->a.alt_tables[k] = alt_tables[k]; p->a.alt_rows[k] = n_rows[k]; }
    return EVS_OK;
}
extern "C" int evs_hostaprx_stats(evs_hostaprx *p, int64_t *out4) {
    using namespace evs;
    EVS_REQUIRE(p && out4, "evs_hostaprx_stats: NULL argument");
    out4[0] = (int64_t)p->a.members.size(); out4[1] = p->a.n_hit; out4[2] = p->a.n_pending; out4[3] = p->a.error;
    return EVS_OK;
}
// APRX_EV's single-key methods in order: 0 insert_altkey (aprx_embedding.cpp:278-288), 1 get_altkey_str (:341-350),
// 2 set_recency_flag_c3 (:402-411), 3 evict_one_key (:390-400)
extern "C" int evs_hostaprx_apply_ops(evs_hostaprx *p, int64_t n, const int32_t *ops, uint32_t *res) {
    using namespace evs;
    EVS_REQUIRE(p && ops && res, "evs_hostaprx_apply_ops: NULL argument");
    host::AltKeys &a = p->a;
    for (int64_t i = 0; i < n; i++) {
        const int op = ops[3 * i], tab = ops[3 * i + 1];
        const int32_t row = ops[3 * i + 2];
        EVS_REQUIRE(tab >= 1 && tab <= a.T && row >= 0 && (op != 0 || row < a.alt_rows[tab - 1]), "evs_hostaprx_apply_ops: op %lld out of range", (long long)i);
        const uint64_t key = ((uint64_t)(uint32_t)tab << 32) | (uint32_t)row;
        res[i] = 0;
        if (op == 0) {
            const uint32_t alt = a.alt_of(key);
            if ((int64_t)a.members.size() >= a.cap) a.evict_one();
            a.admit(key, alt);
            if (a.error) { set_error("evs_hostaprx_apply_ops: FIFO overflow"); return EVS_ESTATE; }
            res[i] = alt;
        } else if (op == 1) {
            auto it = a.members.find(key);
            res[i] = it != a.members.end() ? it->second.alt : 0xffffffffu;
        } else if (op == 2) {
            auto it = a.members.find(key);
            if (it != a.members.end()) it->second.recent = 1;
        } else if (op == 3) {
            a.evict_one();
        }
    }
    return EVS_OK;
}
extern "C" int64_t evs_hostaprx_dump_queue(evs_hostaprx *p, int64_t *pairs, int64_t max_pairs) {
    using namespace evs;
    if (!p) { set_error("evs_hostaprx_dump_queue: NULL"); return EVS_EINVAL; }
    int64_t n = 0;
    for (int64_t q = p->a.qhead; q < p->a.qtail; q++, n++)
        if (pairs && n < max_pairs) {
            const uint64_t key = p->a.fifo[q % p->a.fifo.size()];
            pairs[2 * n] = (int64_t)(key >> 32); pairs[2 * n + 1] = (int64_t)(key & 0xffffffffu);
        }
    return n;
}
