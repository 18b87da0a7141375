// GPU-resident EvLFU / LRU / LFU embedding cache ("C1") for gfx950.
//
// Reference semantics (cited per function below):
//   cache_algo/EvLFU_C1.py:21-166        EvLFU policy (set / update_agg_hit / update / request_to_ev_lfu)
//   cache_algo/LRU.py:14-64, LFU.py:12-95 baseline policies
//   mixed_precs_caching/evlfu_8.cpp:252-321,798-868   the C++ single-tier variant (flush constants differ)
//   cache_algo/EvLFU_C1_Cython/EvLFU.cpp:70-232       the Cython variant (0.4 / 1.0 constants)
//
// Data layout in HBM (everything the policy touches lives on the device):
//   keys[nslot] u64     open-address hash, linear probing, key = (table_1based << 32) | row, 0 = empty
//   slot_entry[nslot]   entry id of the key in that slot
//   ekey/eagg/prev/next[cap]  entry records; prev/next thread the per-priority FIFO lists
//   arena[cap * row_bytes]    the cached rows, in the table's own codec (fp32 / u16 / u8 / u4 bytes)
//   State                     list heads/tails/lengths, min_C1, n_perfect, free stack pointer, counters
//
// Two request paths:
//   * evlfu_exact_kernel -- ONE wavefront replays requests strictly in order, so the hit
//     trace, the eviction order and the final state are bit-for-bit the reference's
//     (parity gate at batch 1; any B works, B requests cost B serial policy steps).
//     Lane i owns table i: all T keys of a request are probed at once, agg_hit is
//     popcount(ballot(hit)); list surgery is done by lane 0 in table order; rows move with
//     all lanes (arena -> out for hits, backing store -> out and -> arena for inserts).
//   * cache_probe_gather_kernel (batched, snapshot semantics) -- see the section below.
#include "evs_common.h"
#include "evs_hash.h"
#include <type_traits>

#include <mutex>
#include <vector>
#include <stdlib.h>
#include <string.h>

#ifndef EVS_X_EXACT_STOP
#define EVS_X_EXACT_STOP 0
#endif
namespace evs {


enum Policy { kEvLFU = 0, kLRU = 1, kLFU = 2 };

struct CacheState {
    int cap, n_tables, dim, codec, row_bytes, policy;
    unsigned long long nslot_mask;
    int min_c1, n_perfect, max_perfect, flush_n, perfect_mode;
    int count;   // live entries
    int n_free;  // free-stack depth
    long long n_flush, n_evict, n_requests, n_perfect_hits, n_hits;
    int head[kMaxBuckets], tail[kMaxBuckets], len[kMaxBuckets];
    // LFU: lists by frequency live in lfu_head/lfu_tail/lfu_len arrays (device), least_freq here
    long long least_freq;
    int error;   // sticky: 1 = flush on an empty bucket (Python would raise), 2 = internal inconsistency
};

struct CacheArrays {
    // the map: open addressing, keys[slot] = key | entry << kMapEntryShift when the capacity fits 25 bits (`packed`: a probe is
    // ONE dependent access instead of two -- round 6), the entry also in slot_entry[slot] (larger caches read it from there)
    unsigned long long *keys;
    int *slot_entry;
    int packed;
    unsigned long long *ekey;
    int *eagg;        // EvLFU: priority bucket; LFU: unused (efreq used); LRU: 0
    long long *efreq; // LFU frequency
    int *prev, *next;
    int *free_stack;
    unsigned char *arena;
    int *lfu_head, *lfu_tail, *lfu_len;  // LFU only, indexed by frequency (< lfu_max_freq)
    long long lfu_max_freq;
};

struct CacheArgs {
    CacheState *st;
    CacheArrays a;
    const unsigned char *backing[kMaxTables];
    long long backing_rows[kMaxTables];
    const int *requests;   // (B, T) int32 row ids
    float *out;            // (B, T, d) fp32, decoded
    unsigned char *hit;    // (B, T)
    long long B;
    int approx_thres;
};


// the probing lanes must see what lane 0 wrote in the previous request.  The exact kernels are ONE workgroup (one wave) per
// launch, so workgroup scope is the scope that is needed (EVS_EXACT_SCOPE_AGENT: developer A/B -- agent-scope accesses go past
// the CU's L1 on every step of a request's chain of dependent accesses)
#ifdef EVS_EXACT_SCOPE_AGENT
#define EVS_EXACT_SCOPE __HIP_MEMORY_SCOPE_AGENT
#else
#define EVS_EXACT_SCOPE __HIP_MEMORY_SCOPE_WORKGROUP
#endif
template <typename T>
__device__ __forceinline__ T ld(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, EVS_EXACT_SCOPE); }
template <typename T>
__device__ __forceinline__ void st(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, EVS_EXACT_SCOPE); }

// lane i's value, i wave-uniform: v_readlane_b32 into a scalar register (round 6; __shfl cannot know that its index is
// uniform and goes through the LDS crossbar: thirteen ds_bpermute per table of a request)
__device__ __forceinline__ int rl(int v, int i) { return __builtin_amdgcn_readlane(v, i); }
__device__ __forceinline__ unsigned long long rl64(unsigned long long v, int i) {
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(v >> 32), i) << 32) | (unsigned)__builtin_amdgcn_readlane((int)v, i);
}
// keys are (table_1based << 32) | row with table_1based <= kMaxTables = 64: 39 bits; an entry index of at most 25 bits rides above
constexpr int kMapEntryShift = 39;
constexpr unsigned long long kMapKeyMask = (1ull << kMapEntryShift) - 1ull;
__device__ __forceinline__ unsigned long long map_word(const CacheArrays &a, unsigned long long key, int e) {
    return a.packed ? (key | ((unsigned long long)(unsigned)e << kMapEntryShift)) : key;
}
__device__ __forceinline__ unsigned long long map_key(const CacheArrays &a, unsigned long long w) { return a.packed ? (w & kMapKeyMask) : w; }
__device__ int map_find(const CacheArrays &a, unsigned long long mask, unsigned long long key, long long *empty_slot = nullptr) {
    unsigned long long i = mix64(key) & mask;
    for (;;) {
        const unsigned long long k = ld(&a.keys[i]);
        if (map_key(a, k) == key) return a.packed ? (int)(k >> kMapEntryShift) : ld(&a.slot_entry[i]);
        if (k == kEmpty) { if (empty_slot) *empty_slot = (long long)i; return -1; }   // (where an insert of this key would land)
        i = (i + 1) & mask;
    }
}
__device__ void map_put(const CacheArrays &a, unsigned long long mask, unsigned long long key, int e) {
    unsigned long long i = mix64(key) & mask;
    for (;;) {
        const unsigned long long k = ld(&a.keys[i]);
        if (k == kEmpty || map_key(a, k) == key) break;
        i = (i + 1) & mask;
    }
    st(&a.slot_entry[i], e);
    st(&a.keys[i], map_word(a, key, e));
}
// linear-probing delete with backward shift (no tombstones: the table never degrades)
__device__ void map_del(const CacheArrays &a, unsigned long long mask, unsigned long long key) {
    unsigned long long i = mix64(key) & mask;
    {   // the common case in ONE round trip (round 6): the key in its home slot, nothing behind it (load <= 0.5)
        const unsigned long long k0 = ld(&a.keys[i]), k1 = ld(&a.keys[(i + 1) & mask]);
        if (map_key(a, k0) == key && k1 == kEmpty) { st(&a.keys[i], kEmpty); return; }
    }
    for (;;) {
        const unsigned long long k = ld(&a.keys[i]);
        if (map_key(a, k) == key) break;
        if (k == kEmpty) return;
        i = (i + 1) & mask;
    }
    unsigned long long j = i;
    for (;;) {
        j = (j + 1) & mask;
        const unsigned long long kj = ld(&a.keys[j]);   // (a packed word moves with its entry)
        if (kj == kEmpty) break;
        const unsigned long long h = mix64(map_key(a, kj)) & mask;
        const bool between = (i <= j) ? (h > i && h <= j) : (h > i || h <= j);
        if (!between) {
            st(&a.slot_entry[i], ld(&a.slot_entry[j]));
            st(&a.keys[i], kj);
            i = j;
        }
    }
    st(&a.keys[i], kEmpty);
}

// intrusive FIFO lists (append tail / pop head / unlink) over entry ids
struct ListRef { int *head, *tail, *len; };
__device__ __forceinline__ void list_append(const CacheArrays &a, ListRef l, int e) {
    const int t = *l.tail;
    st(&a.prev[e], t);
    st(&a.next[e], -1);
    if (t >= 0) st(&a.next[t], e); else *l.head = e;
    *l.tail = e;
    (void)__hip_atomic_fetch_add(l.len, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (ds_add_u32: no read to wait for)
}
__device__ __forceinline__ void list_unlink(const CacheArrays &a, ListRef l, int e) {
    const int p = ld(&a.prev[e]), n = ld(&a.next[e]);
    if (p >= 0) st(&a.next[p], n); else *l.head = n;
    if (n >= 0) st(&a.prev[n], p); else *l.tail = p;
    (void)__hip_atomic_fetch_sub(l.len, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ float decode_elem(const unsigned char *row, int codec, int c) {
    if (codec == 32) return reinterpret_cast<const float *>(row)[c];
    if (codec == 16) return dec_u16(reinterpret_cast<const unsigned short *>(row)[c]);
    if (codec == 8) return dec_u8(row[c]);
    const unsigned b = row[c >> 1];
    return kU4Lut[(c & 1) ? (b & 15u) : (b >> 4)];
}

// shared (LDS) mirror of the hot part of CacheState: only lane 0 mutates it
struct Hot {
    int min_c1, n_perfect, count, n_free;
    long long n_flush, n_evict, least_freq;
    int head[kMaxBuckets], tail[kMaxBuckets], len[kMaxBuckets];
    int error;
    unsigned long long last_evicted;  // key evicted (not flushed) by the latest evlfu_set, 0 if none
};

__device__ __forceinline__ ListRef bucket(Hot &h, int b) { return ListRef{&h.head[b], &h.tail[b], &h.len[b]}; }
__device__ __forceinline__ ListRef lfu_list(const CacheArrays &a, long long f) {
    return ListRef{&a.lfu_head[f], &a.lfu_tail[f], &a.lfu_len[f]};
}

__device__ void drop_entry(const CacheArrays &a, Hot &h, unsigned long long mask, int e) {
    map_del(a, mask, ld(&a.ekey[e]));
    st(&a.ekey[e], kEmpty);
    a.free_stack[h.n_free++] = e;
    h.count--;
}

// Round 6, what lane 0's loop over a request's tables may still trust of what the lanes fetched in advance: a 1 024-bit filter
// (LDS, cleared per request) of the entries whose key or list links an operation of THIS request has changed -- an evicted
// entry (it carries another key now), the neighbours of an unlinked entry, the old tail of a list that was appended to, the
// entry moved.  An entry that is not in it (the filter has false positives, never false negatives) still has the key, the
// priority and the neighbours its lane read before the loop: no dependent access for its liveness check and its unlink,
// where the first eviction of a request used to send every later hit down two of them.  A flush sets `all`.
struct TouchMap { unsigned *bits; bool all; };
// (an LDS atomic whose result nobody reads: one ds_or_b32, nothing to wait for -- a read-modify-write in C is a ds_read the wave stalls on)
__device__ __forceinline__ void touch(TouchMap &m, int e) { if (m.bits && e >= 0) (void)__hip_atomic_fetch_or(&m.bits[((unsigned)e >> 5) & 31u], 1u << ((unsigned)e & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ bool touched(const TouchMap &m, int e) { return m.all || !m.bits || ((m.bits[((unsigned)e >> 5) & 31u] >> ((unsigned)e & 31u)) & 1u) != 0u; }

// EvLFU_C1.py:32-63 set(key, value, agg_hit); returns the entry id (row is copied later)
__device__ int evlfu_set(const CacheState &cs, const CacheArrays &a, Hot &h, unsigned long long mask,
                         unsigned long long key, int agg_hit, TouchMap *tm = nullptr) {
    const int top = cs.n_tables;
    h.last_evicted = 0;
    int reuse = -1;    // the entry an eviction has just freed: handed to the new key without a trip through the free stack
    if (h.n_perfect >= cs.max_perfect) {  // :36-44 flush the oldest of the top bucket
        for (int i = 0; i < cs.flush_n; i++) {
            const int e = h.head[top];
            if (e < 0) { if (cs.perfect_mode == 0) h.error = 1; break; }  // Python: IndexError; C++: loop guard
            list_unlink(a, bucket(h, top), e);
            drop_entry(a, h, mask, e);
        }
        // Python recounts (:43); the C++/Cython variants subtract int(rate*cap) (evlfu_8.cpp:270, EvLFU.cpp:86)
        h.n_perfect = (cs.perfect_mode == 0) ? h.len[top] : h.n_perfect - (cs.flush_n - (cs.perfect_mode == 1 ? 1 : 0));
        h.n_flush++;
        if (tm) tm->all = true;
    } else if (h.count >= cs.cap) {       // :47-56 evict the FIFO-oldest key of the lowest non-empty bucket
        while (h.len[h.min_c1] == 0) {
            h.min_c1++;
            if (h.min_c1 > top) h.min_c1 = 1;
        }
        const int e = h.head[h.min_c1];
        // the victim's neighbours and its key in ONE round trip (the key waited behind the unlink's stores)
        const int vp = ld(&a.prev[e]), vn = ld(&a.next[e]);
        const unsigned long long vk = ld(&a.ekey[e]);
        {
            ListRef l = bucket(h, h.min_c1);
            if (vp >= 0) st(&a.next[vp], vn); else *l.head = vn;
            if (vn >= 0) st(&a.prev[vn], vp); else *l.tail = vp;
            *l.len -= 1;
        }
        h.last_evicted = vk;
        map_del(a, mask, vk);
        st(&a.ekey[e], kEmpty);
        h.count--;
        h.n_evict++;
        reuse = e;     // (drop_entry would push it on the free stack and the pop below would take it straight back)
        if (tm) { touch(*tm, e); touch(*tm, vp); touch(*tm, vn); }
    }
    int e;
    if (reuse >= 0) e = reuse;
    else {
        if (h.n_free <= 0) { h.error = 2; return -1; }
        e = a.free_stack[--h.n_free];
    }
    st(&a.ekey[e], key);
    st(&a.eagg[e], agg_hit);
    map_put(a, mask, key, e);
    if (tm) { touch(*tm, *bucket(h, agg_hit).tail); touch(*tm, e); }
    list_append(a, bucket(h, agg_hit), e);
    h.count++;
    if (agg_hit < h.min_c1) h.min_c1 = agg_hit;  // :62-63
    return e;
}

// The same kernel as a SERVER (round 5; evs_cache_serve_*): it stays resident and takes one request at a time from a
// mailbox in pinned host memory -- the reference's operating point is batch 1 (cache_algo/EvLFU_C1.py:97-166,
// dlrm_s_pytorch_C1.py:227-275), where a launch and a synchronise per request cost twice what the request does.
//   request line  (host -> device, 128 bytes): four 32-byte sectors of seven row ids and a guard word (words 7, 15, 23, 31 =
//                 the sequence number) -- the host writes the ids, then the four guards; the device reads the line as one
//                 32-lane load (bus reads of 32 or 64 bytes that may be taken at different times) and accepts it when ALL
//                 guards hold the number it waits for: every sector was then read after its guard, hence after its ids,
//                 was written -- whatever the granularity the fabric delivers the line in;
//   control line  (host -> device): word 0 = stop;
//   answer line   (device -> host): the T hit flags in bytes 0..63, `done` (the sequence number served) in word 16, `alive`
//                 in word 17 -- flags first, a system-scope fence, then `done`.
// The rows go to slot (sequence % n_slots) of a ring in HBM: the caller gets DEVICE rows with no launch, copy or synchronise.
// An idle server leaves by itself after idle_ticks of the 100 MHz wall clock (a device-wide synchronise elsewhere in the
// process waits no longer than that); the host starts it again with the next request.
#ifndef EVS_X_SERVE_FENCE
#define EVS_X_SERVE_FENCE 0
#endif
#ifdef EVS_X_EXACT_TIMING   // developer build: where a request's time goes (100 MHz ticks per stage, summed over a launch)
static __device__ long long g_exact_ticks[8];
#define EVS_TICK(k) do { const long long now_ = (long long)wall_clock64(); if (lane == 0) tick_acc[k] += now_ - tick_t; tick_t = now_; } while (0)
#else
#define EVS_TICK(k) do { } while (0)
#endif
struct ServeArgs {
    volatile unsigned *req;     // request line (device address of the mapped host block)
    volatile unsigned *ctl;     // control line
    volatile unsigned *ans;     // answer line
    float *ring; int n_slots;
    long long idle_ticks;
};
// One wavefront, requests strictly in order.
// The exact kernels are ONE wavefront (launched with 64 threads): what its lanes hand each other through LDS, and what lane 0
// writes to the entry arrays for the lanes of the next request to read, is ordered by the wave's own in-order issue -- a
// wavefront-scope fence (no cache action, no wait) and a scheduling barrier say so to the compiler.  __syncthreads() here was a
// workgroup-scope release: s_waitcnt vmcnt(0) in front of every one of them, i.e. the wave waited for its own STORES to be
// acknowledged three to four times per request (round 6: ~2.5 of a request's 10 us).
#define EVS_WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
template <bool SERVE>
__device__ __forceinline__ void cache_exact_body(const CacheArgs &args, const ServeArgs &sv) {
    __shared__ Hot h;
    __shared__ int s_src[kMaxTables];    // >=0: arena entry to read the row from; -1: backing store; -2: copy of table s_from
    __shared__ int s_fill[kMaxTables];   // >=0: arena entry to fill from the backing row after the request
    __shared__ int s_from[kMaxTables];   // approximate mode: table whose vector is reused
    __shared__ const unsigned char *s_rowp[kMaxTables];
    __shared__ int s_req[kMaxTables];
    __shared__ unsigned s_touch[32];     // TouchMap::bits of the request in hand (EvLFU)
    // the tables' addresses and row counts out of LDS (round 6): indexed by a lane's table they were reads of the kernel arguments
    // -- a vector-memory round trip -- inside every request's row stage
    __shared__ const unsigned char *s_fsrc[kMaxTables];   // the request's new entries in table order: where their row comes from ...
    __shared__ unsigned char *s_fdst[kMaxTables];         // ... and where it goes (nullptr: evicted again within the request)
    __shared__ const unsigned char *s_back[kMaxTables];
    __shared__ long long s_brows[kMaxTables];
    const int lane = threadIdx.x;
    CacheState *gs = args.st;
    const CacheArrays a = args.a;
    const CacheState cs = *gs;           // immutable config fields are read from this copy
    const unsigned long long mask = cs.nslot_mask;
    const int T = cs.n_tables, d = cs.dim, rb = cs.row_bytes;
    if (lane == 0) {
        h.min_c1 = cs.min_c1; h.n_perfect = cs.n_perfect; h.count = cs.count; h.n_free = cs.n_free;
        h.n_flush = cs.n_flush; h.n_evict = cs.n_evict; h.least_freq = cs.least_freq; h.error = cs.error;
    }
    for (int b = lane; b < kMaxBuckets; b += 64) { h.head[b] = gs->head[b]; h.tail[b] = gs->tail[b]; h.len[b] = gs->len[b]; }
    for (int k = lane; k < kMaxTables; k += 64) { s_back[k] = args.backing[k]; s_brows[k] = args.backing_rows[k]; }
    EVS_WSYNC();
    // the chunk new entries' rows are moved in (table -> arena): 16 bytes when the row length, the arena and every table's base
    // are multiples of 16, else 4, else 0 = byte by byte
    static_assert(kMaxTables <= 64, "one lane per table");
    const unsigned long long align_bits = (unsigned long long)(unsigned)rb | (unsigned long long)reinterpret_cast<uintptr_t>(a.arena) |
                                          (lane < T ? (unsigned long long)reinterpret_cast<uintptr_t>(s_back[lane]) : 0ull);
    const int fill_cs = __ballot((align_bits & 15ull) != 0ull) == 0ull ? 16 : (__ballot((align_bits & 3ull) != 0ull) == 0ull ? 4 : 0);
    const int fill_cpr = fill_cs ? rb / fill_cs : 1;            // chunks per row
    const float fill_inv_cpr = 1.0f / (float)fill_cpr;          // (chunk index -> row of the list: exact for the 256 chunks it is used on)
    long long n_hits = 0, n_perfect_hits = 0;
    unsigned warm_acc = 0u;            // (what the row warm-up loads return: kept alive to the kernel's end, never meaningful)
#if EVS_X_EXACT_STOP == 1
    return;
#endif

    unsigned serve_seq = 0;            // SERVE: the last sequence number served
    long long served = 0;
#ifdef EVS_X_EXACT_TIMING
    long long tick_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tick_t = (long long)wall_clock64();
#endif
    if constexpr (SERVE) {
        serve_seq = __hip_atomic_load(const_cast<unsigned *>(sv.ans) + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (lane == 0) __hip_atomic_store(const_cast<unsigned *>(sv.ans) + 17, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // alive
    }
    for (long long rq = 0; SERVE || rq < args.B; rq++) {
        int row;
        float *out;
        unsigned char *hitp;
        if constexpr (SERVE) {
            // ---- wait for the next request: the whole line in one load (lane = word), both guards = the number awaited ----
            const unsigned want = (serve_seq + 1u) & 0x7fffffffu;
            const long long t0 = (long long)wall_clock64();
            unsigned word = 0u;
            bool leave = false;
            for (;;) {
                // ONE load per poll (round 6: the control word was a second dependent load, and the request waited out both):
                // lanes 0..31 the request line, lane 32 the control word in the line behind it
                word = __hip_atomic_load(const_cast<unsigned *>(sv.req) + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                // words 7, 15, 23, 31: one guard per 32-byte sector (bit 31 of a guard: the ids are given by ADDRESS, below)
                if ((__ballot((word & 0x7fffffffu) == want) & 0x80808080ull) == 0x80808080ull) break;
                const unsigned stop = (unsigned)__builtin_amdgcn_readlane((int)word, 32);
                if (stop != 0u || (long long)wall_clock64() - t0 > sv.idle_ticks) { leave = true; break; }
                __builtin_amdgcn_s_sleep(4);
            }
            if (leave) break;
            // id of table t: word t + t / 7 (seven ids, then the sector's guard) ...
            const int src_lane = lane + lane / 7;
            row = __shfl((int)word, src_lane & 31);
            // ... or, round 6 (evs_cache_serve_request_dev: the reference's loop has the ids on the DEVICE -- dlrm_wrap moved lS_i
            // there -- and fetching them back costs the plug-in a copy and a synchronise per request): words 0..2 = the address of
            // table 0's first index and the elements between two tables' rows; one agent-scope load per lane
            if (((unsigned)rl((int)word, 7) >> 31) != 0u) {
                const unsigned long long pa = ((unsigned long long)(unsigned)rl((int)word, 1) << 32) | (unsigned)rl((int)word, 0);
                const long long stride = (long long)(unsigned)rl((int)word, 2);
                const long long *ip = reinterpret_cast<const long long *>((uintptr_t)pa) + (long long)(lane < T ? lane : 0) * stride;
                row = (int)__hip_atomic_load(ip, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // ... and, round 6 (evs_cache_serve_request_to), where the rows go: a DEVICE address of the caller's (words 3, 4 beside
            // an id address; words 29, 30 -- the ids of tables 26, 27 -- beside ids when T <= 26), 0 = this request's ring slot
            unsigned long long out_a = 0ull;
            if (((unsigned)rl((int)word, 7) >> 31) != 0u) out_a = ((unsigned long long)(unsigned)rl((int)word, 4) << 32) | (unsigned)rl((int)word, 3);
            else if (T <= 26) out_a = ((unsigned long long)(unsigned)rl((int)word, 30) << 32) | (unsigned)rl((int)word, 29);
            if (lane >= T) row = 0;
            out = out_a ? reinterpret_cast<float *>((uintptr_t)out_a) : sv.ring + (long long)(want % (unsigned)sv.n_slots) * T * d;
            hitp = nullptr;   // (flags go to the answer line, below)
        } else {
            row = lane < T ? args.requests[rq * T + lane] : 0;   // the only read of the ids (they may live in host memory)
            out = args.out + rq * (long long)T * d;
            hitp = args.hit + rq * T;
        }
        EVS_TICK(0);                    // (waiting for / reading the request)
        unsigned char my_flag = 0;      // this lane's table: the hit flag of the request in hand
        bool churn = true;              // an insert of this request may have evicted / flushed entries (LRU / LFU: always assumed)
        bool rows_known = false;        // (wave-uniform) s_rowp has been written by the policy step itself: the resolve step below is skipped
        if (lane < T) s_req[lane] = row;
        const unsigned long long key = ((unsigned long long)(lane + 1) << 32) | (unsigned)row;
        const bool row_ok = lane < T && row >= 0 && row < s_brows[lane < T ? lane : 0];
        if (lane < T) { s_src[lane] = -1; s_fill[lane] = -1; s_from[lane] = -1; }

        if (cs.policy == kEvLFU) {
            // ---- probe all T keys at once; agg_hit = popcount(ballot)  (EvLFU_C1.py:105-120) ----
            long long ins_slot = -1;   // a missing key: the empty slot its probe walk ended on
            int e = (lane < T && row_ok) ? map_find(a, mask, key, &ins_slot) : -1;
            const unsigned long long hit_mask = __ballot(e >= 0);
            EVS_TICK(1);                // (the probe)
            const int agg_hit = __popcll(hit_mask);
            const bool pick_random = args.approx_thres > 0 && agg_hit >= args.approx_thres;  // :122-125
            bool my_hit = e >= 0;
            // every lane fetches its own entry's key and priority now (one round trip for all T); the serial
            // loop below re-reads them only after an insert of THIS request may have evicted / reused an entry
            // (round 6: NO load of this block sits behind `e >= 0 ? ... : ...` any more -- a load in a divergent branch is waited
            //  for at the branch's end, and the seven fetches below were seven round trips one after the other: 1.25 us of a
            //  10 us request, tools/exact_stage_probe.py; a lane with nothing to fetch reads entry 0 and drops it)
            const int e0 = e >= 0 ? e : 0;
            const unsigned long long pre_key_ = ld(&a.ekey[e0]);
            const int pre_agg_ = ld(&a.eagg[e0]);
            // Round 5, the same idea for everything else lane 0's serial loop used to fetch one dependent access at a time
            // (a request with six misses was ~15 round trips of one lane): every lane fetches NOW what its key will need --
            // a hit whose priority will rise: its list neighbours; a miss: the free entry it will be given (the stack is
            // popped in table order: the r-th miss of the request gets free_stack[n_free - 1 - r]) -- and knows the map slot
            // its insert lands in (the empty slot its probe ended on; two new keys that ended on the SAME slot: the later
            // one probes again).  The loop then only stores -- until something it cannot have foreseen happens (an eviction
            // or a flush rewrites the map, the stack and the lists: from there on the request takes the old path).
            const bool my_miss = lane < T && row_ok && e < 0 && !pick_random;
            const unsigned long long miss_mask = __ballot(my_miss);
            const int my_rank = __popcll(miss_mask & ((1ull << lane) - 1ull));
            const int n_free0 = h.n_free;
            const bool has_free = my_miss && n_free0 - 1 - my_rank >= 0;
            const int pre_free_ = a.free_stack[has_free ? n_free0 - 1 - my_rank : 0];
            // (asked for with the key and the priority, not behind them: a hit that does not move has fetched two words for nothing,
            //  one that moves has saved a dependent round trip -- round 6)
            const int pre_prev_ = ld(&a.prev[e0]);
            const int pre_next_ = ld(&a.next[e0]);
            // ... and the first and the last word of the row this lane's table will be served from (arena row of a hit, table row
            // of a miss: a random line of a multi-GB table, i.e. a TLB miss on top of the DRAM access): asked for HERE, behind the
            // fetches above (vector-memory operations complete in issue order: in front of them they would hold them up), their
            // latency runs under those fetches and the policy step instead of in front of the row stage
            const unsigned char *warm_p = (lane < T && row_ok) ? (e >= 0 ? a.arena + (long long)e * rb : s_back[lane] + (long long)row * rb) : a.arena;
            // (plain loads whose value is looked at when the NEXT request gets here: a volatile load is waited for on the spot)
            const unsigned warm_now = (unsigned)warm_p[0] ^ (unsigned)warm_p[rb > 0 ? rb - 1 : 0];
#ifdef EVS_X_PRE_TIMING
            EVS_TICK(6);
#endif
            const unsigned long long pre_key = e >= 0 ? pre_key_ : 0ull;
            const int pre_agg = e >= 0 ? pre_agg_ : 0, pre_free = has_free ? pre_free_ : -1;
            const int pre_prev = e >= 0 ? pre_prev_ : -1, pre_next = e >= 0 ? pre_next_ : -1;
            bool slot_clash = false;   // an earlier miss of this request ended on the same empty slot
            // (over the misses only, their slot read with v_readlane: the loop over all T tables with two __shfl -- the LDS
            //  crossbar, its index not known to be uniform -- was 2.2 us of the request)
            for (unsigned long long mm = miss_mask; mm; mm &= mm - 1ull) {
                const int j = __builtin_amdgcn_readfirstlane(__builtin_ctzll(mm));
                const long long sj = (long long)rl64((unsigned long long)ins_slot, j);
                slot_clash |= my_miss && j < lane && sj == ins_slot;
            }
#ifdef EVS_X_PRE_TIMING
            if (slot_clash && lane == 77) h.error = 9;
            EVS_TICK(7);
#endif
            bool dirty = false;
            bool fast_ok = !pick_random;   // nothing unforeseen has happened to the map / the stack yet
            if (lane < 32) s_touch[lane] = 0u;
            TouchMap tm{s_touch, false};   // (lane 0's: which entries an operation of this request has changed)
#if EVS_X_EXACT_STOP == 2
            if (pre_key != 12345ull || pre_agg != -7) return;
#endif
#ifdef EVS_X_EXACT_TIMING
            if (pre_key == 1ull && pre_agg == -77 && pre_free == -77 && pre_prev == -77 && pre_next == -77) h.error = 9;   // (the timing build waits for the fetches HERE)
#endif
            EVS_TICK(2);                // (what the lanes fetch in advance)
            // ---- policy update, table order, one lane (EvLFU_C1.py:135-161) ----
            // Round 6: lane 0 visits only the tables whose key CHANGES something -- a miss, or a hit whose priority rises.  A hit
            // that stays where it is needs the loop for one thing only: an insert of this same request may have evicted its entry
            // (:90-94).  That takes an eviction or a flush, and whether this request can cause one is known now: it inserts
            // n_miss keys into count entries of cap, with n_perfect against max_perfect.  If it cannot, the hits that do not move
            // are served by all lanes at once (their source is their entry) and the loop runs over the bits of `work`.
            const int n_miss = __popcll(miss_mask);
            const bool calm = !pick_random && (n_miss == 0 || (h.count + n_miss <= cs.cap && h.n_perfect < cs.max_perfect));
            const bool moves = e >= 0 && pre_agg < agg_hit;
            unsigned long long work = __ballot(lane < T && row_ok && (e < 0 || moves));
            if (!calm) work = __ballot(lane < T && row_ok);
            if (calm && e >= 0 && !moves && lane < T) s_src[lane] = e;
            // Round 6, the calm request with ALL LANES AT ONCE.  Lane 0's loop costs a lone wave ~0.5 us per table it visits
            // (~100 dependent instructions each: 6 of a request's 16 us, tools/exact_stage_probe.py).  When nothing can be evicted,
            // every miss has the free entry and the map slot it fetched in advance, and no two hits that move are neighbours in
            // their list, the request's operations commute except for ONE thing, the order of the appends to bucket agg_hit --
            // table order, i.e. lane order: lane l's entry goes behind the entry of the appending lane before it.  So: every mover
            // unlinks itself (its neighbours were fetched before anything changed; non-adjacent unlinks touch different words),
            // every miss fills its entry and its map slot, every appender links itself between its neighbours in lane order, the
            // first one behind the bucket's old tail, and the counters move by LDS atomics.  Anything else: the loop below.
            const unsigned long long mover_mask = __ballot(lane < T && row_ok && moves);
            bool par_ok = calm && __ballot(my_miss && (slot_clash || pre_free < 0)) == 0ull;
            if (par_ok) {
                bool adj = false;
                for (unsigned long long mm = mover_mask; mm; mm &= mm - 1ull) {
                    const int j = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(mm));
                    const int ej = rl(e, j);
                    adj |= moves && j != lane && (pre_prev == ej || pre_next == ej);
                }
                par_ok = __ballot(lane < T && adj) == 0ull;
            }
            if (par_ok) {
                const bool mover = lane < T && row_ok && moves;
                const unsigned long long app_mask = mover_mask | miss_mask;    // who appends to bucket agg_hit, in lane order
                const int n_app = __popcll(app_mask);
                const int x = mover ? e : pre_free;                              // the entry this lane appends
                const int tail0 = h.tail[agg_hit];
                const unsigned long long below = app_mask & ((1ull << lane) - 1ull), above = lane < 63 ? (app_mask >> (lane + 1)) : 0ull;
                const int lane_prev = below ? 63 - (int)__builtin_clzll(below) : 0, lane_next = above ? lane + 1 + (int)__builtin_ctzll(above) : 0;
                const int x_prev = __shfl(x, lane_prev), x_next = __shfl(x, lane_next);
                if (mover) {   // out of its list (update_agg_hit, :65-78) ...
                    ListRef l = bucket(h, pre_agg);
                    if (pre_prev >= 0) st(&a.next[pre_prev], pre_next); else *l.head = pre_next;
                    if (pre_next >= 0) st(&a.prev[pre_next], pre_prev); else *l.tail = pre_prev;
                    (void)__hip_atomic_fetch_sub(l.len, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    st(&a.eagg[e], agg_hit);
                }
                if (my_miss) {   // ... or set() without an eviction (:57-63): the entry and the map slot fetched in advance
                    st(&a.ekey[pre_free], key);
                    st(&a.eagg[pre_free], agg_hit);
                    st(&a.slot_entry[ins_slot], pre_free);
                    st(&a.keys[ins_slot], map_word(a, key, pre_free));
                    s_fill[lane] = pre_free;
                }
                if (mover || my_miss) {   // ... and behind the appender before it
                    st(&a.prev[x], below ? x_prev : tail0);
                    st(&a.next[x], above ? x_next : -1);
                    if (!below) { if (tail0 >= 0) st(&a.next[tail0], x); else h.head[agg_hit] = x; }
                    if (!above) h.tail[agg_hit] = x;
                    if (mover) s_src[lane] = e;
                }
                if (lane == 0) {
                    h.last_evicted = 0;
                    if (n_app) h.len[agg_hit] += n_app;
                    if (n_miss) { h.n_free -= n_miss; h.count += n_miss; if (agg_hit < h.min_c1) h.min_c1 = agg_hit; }
                }
                // every lane knows where its row comes from -- a hit's entry (moved or not: nothing is evicted on this path), a
                // miss's table row, nothing for an id out of range: the very address warmed above -- so the resolve step's chain of
                // dependent LDS reads (source, request row, table base: 0.4 us) is not needed
                if (lane < T) s_rowp[lane] = row_ok ? warm_p : nullptr;
                rows_known = true;
                work = 0ull;
            }
            int last_hit_table = -1;
            while (work) {
                const int i = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(work));   // (said to be uniform: a readlane whose index the compiler takes for divergent becomes a loop)
                work &= work - 1ull;
                const int ei = rl(e, i);
                const unsigned long long ki = rl64(key, i);
                const bool oki = rl((int)row_ok, i) != 0;
                const bool hiti = (hit_mask >> i) & 1ull;
                const unsigned long long pk = rl64(pre_key, i);
                const int pa = rl(pre_agg, i);
                const int pfree = rl(pre_free, i), pprev = rl(pre_prev, i), pnext = rl(pre_next, i);
                const long long pslot = (long long)rl64((unsigned long long)ins_slot, i);
                const bool pclash = rl((int)slot_clash, i) != 0;
                int src = -1, fill = -1, from = -1;
                if (lane == 0 && oki) {
                    if (hiti) {
                        // update() -> update_agg_hit (:65-78); the entry may have been evicted by an
                        // earlier insert of this same request (:90-94): then re-fetch and set().  What the lanes fetched
                        // before the loop -- key, priority, list neighbours -- stands unless an operation of this request
                        // has touched the entry (TouchMap): no dependent access then
                        const bool stale = touched(tm, ei);
                        const bool alive = (stale ? ld(&a.ekey[ei]) : pk) == ki;
                        if (alive) {
                            const int old = stale ? ld(&a.eagg[ei]) : pa;
                            if (old < agg_hit) {
                                int up = pprev, un = pnext;
                                if (stale) { up = ld(&a.prev[ei]); un = ld(&a.next[ei]); }
                                ListRef l = bucket(h, old);
                                if (up >= 0) st(&a.next[up], un); else *l.head = un;
                                if (un >= 0) st(&a.prev[un], up); else *l.tail = up;
                                (void)__hip_atomic_fetch_sub(l.len, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                touch(tm, up); touch(tm, un); touch(tm, ei); touch(tm, *bucket(h, agg_hit).tail);
                                list_append(a, bucket(h, agg_hit), ei);
                                st(&a.eagg[ei], agg_hit);
                            }
                            src = ei;
                        } else {
                            fill = evlfu_set(cs, a, h, mask, ki, agg_hit, &tm);
                            dirty = true; fast_ok = false;
                        }
                        last_hit_table = i;
                    } else if (pick_random) {  // :142-152: the miss reuses the previous hit's vector
                        from = last_hit_table;
                        src = -2;
                    } else if (fast_ok && !pclash && pfree >= 0 && (calm || (h.n_perfect < cs.max_perfect && h.count < cs.cap))) {
                        // evlfu_set without an eviction or a flush, on what was fetched above: stores only
                        h.last_evicted = 0;
                        (void)__hip_atomic_fetch_sub(&h.n_free, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        st(&a.ekey[pfree], ki);
                        st(&a.eagg[pfree], agg_hit);
                        st(&a.slot_entry[pslot], pfree);
                        st(&a.keys[pslot], map_word(a, ki, pfree));
                        touch(tm, *bucket(h, agg_hit).tail); touch(tm, pfree);
                        list_append(a, bucket(h, agg_hit), pfree);
                        (void)__hip_atomic_fetch_add(&h.count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        (void)__hip_atomic_fetch_min(&h.min_c1, agg_hit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        fill = pfree;   // (nothing was evicted: no later hit's entry can have been reused -- `dirty` stays as it is)
                    } else {
                        // (an eviction / a flush pushes entries back on the stack and rewrites map slots: what the later keys of
                        //  this request fetched in advance is void from here on)
                        fast_ok = false;   // (also behind a slot clash: the key lands wherever ITS probe ends now -- maybe on the slot a later miss fetched)
                        fill = evlfu_set(cs, a, h, mask, ki, agg_hit, &tm);
                        dirty = true;
                    }
                    s_src[i] = src; s_fill[i] = fill; s_from[i] = from;
                }
#ifdef EVS_X_EXACT_TIMING
#if !defined(EVS_X_ROWS_TIMING) && !defined(EVS_X_PRE_TIMING)
                { const long long now_ = (long long)wall_clock64(); if (lane == 0) { tick_acc[hiti ? 6 : 7] += now_ - tick_t; } tick_t = now_; }
#endif
#endif
            }
            EVS_TICK(3);                // (lane 0's loop: what is left of it beside the per-kind ticks inside)
            churn = rl((int)dirty, 0) != 0;
            warm_acc ^= warm_now;   // (the warm-up loads are looked at here, with the policy step behind them)
            if (lane == 0 && agg_hit == T) h.n_perfect = h.len[T];  // :163-165
            if (pick_random) my_hit = lane < T;                       // misses are reported as hits (:152)
            my_flag = my_hit ? 1 : 0;
            int nh = __popcll(__ballot(my_hit && lane < T));
            n_hits += nh;
            n_perfect_hits += (nh == T);
        } else if (cs.policy == kLRU) {
            // LRU.py:38-64: key by key; hit -> move to MRU end (:24-28); miss -> evict LRU head if full (:14-20)
            int nh = 0;
            for (int i = 0; i < T; i++) {
                const unsigned long long ki = __shfl(key, i);
                const bool oki = __shfl((int)row_ok, i) != 0;
                int hit_i = 0;
                if (lane == 0 && oki) {
                    const int ei = map_find(a, mask, ki);
                    if (ei >= 0) {
                        list_unlink(a, bucket(h, 0), ei);
                        list_append(a, bucket(h, 0), ei);
                        s_src[i] = ei;
                        hit_i = 1;
                    } else {
                        if (h.count >= cs.cap) {
                            const int v = h.head[0];
                            list_unlink(a, bucket(h, 0), v);
                            drop_entry(a, h, mask, v);
                            h.n_evict++;
                        }
                        const int en = a.free_stack[--h.n_free];
                        st(&a.ekey[en], ki);
                        map_put(a, mask, ki, en);
                        list_append(a, bucket(h, 0), en);
                        h.count++;
                        s_fill[i] = en;
                    }
                }
                hit_i = __shfl(hit_i, 0);
                if (lane == i) my_flag = (unsigned char)hit_i;
                nh += hit_i;
                // the fill must land before a later key of this request can evict/reuse the entry:
                // rows are moved after the loop from the BACKING store for every miss, so order is moot
            }
            n_hits += nh;
            n_perfect_hits += (nh == T);
        } else {
            // LFU.py:69-95: FIFO inside a frequency; hit moves f -> f+1 (:19-34); miss evicts head of least_freq (:36-51)
            int nh = 0;
            for (int i = 0; i < T; i++) {
                const unsigned long long ki = __shfl(key, i);
                const bool oki = __shfl((int)row_ok, i) != 0;
                int hit_i = 0;
                if (lane == 0 && oki) {
                    const int ei = map_find(a, mask, ki);
                    if (ei >= 0) {
                        const long long f = a.efreq[ei];
                        list_unlink(a, lfu_list(a, f), ei);
                        if (a.lfu_len[h.least_freq] == 0) h.least_freq += 1;
                        if (f + 1 >= a.lfu_max_freq) { h.error = 3; }
                        else {
                            a.efreq[ei] = f + 1;
                            list_append(a, lfu_list(a, f + 1), ei);
                        }
                        s_src[i] = ei;
                        hit_i = 1;
                    } else {
                        if (h.count >= cs.cap) {
                            const int v = a.lfu_head[h.least_freq];
                            if (v < 0) { h.error = 2; }
                            else {
                                list_unlink(a, lfu_list(a, h.least_freq), v);
                                drop_entry(a, h, mask, v);
                                h.n_evict++;
                            }
                        }
                        const int en = a.free_stack[--h.n_free];
                        st(&a.ekey[en], ki);
                        a.efreq[en] = 1;
                        map_put(a, mask, ki, en);
                        list_append(a, lfu_list(a, 1), en);
                        h.count++;
                        h.least_freq = 1;
                        s_fill[i] = en;
                    }
                }
                hit_i = __shfl(hit_i, 0);
                if (lane == i) my_flag = (unsigned char)hit_i;
                nh += hit_i;
            }
            n_hits += nh;
            n_perfect_hits += (nh == T);
        }
        if constexpr (!SERVE) { if (lane < T) hitp[lane] = my_flag; }
        EVS_WSYNC();

        // ---- rows: hits from the arena, misses from the backing store; then fill inserted entries ----
#if EVS_X_EXACT_STOP == 3
        return;
#endif
        if (!rows_known && lane < T) {   // lane i resolves where row i comes from ...
            const int i = lane;
            int src = s_src[i];
            int tsrc = i;
            bool zero = false;
            if (src == -2) {  // approximate mode: vector of the previous hit (or zeros when there was none)
                tsrc = s_from[i];
                if (tsrc < 0) { zero = true; tsrc = i; } else src = s_src[tsrc];
            }
            const int rrow = s_req[tsrc];
            const bool ok = rrow >= 0 && rrow < s_brows[tsrc];
            const unsigned char *rowp = nullptr;
            if (!zero) {
                if (src >= 0) rowp = a.arena + (long long)src * rb;
                else if (ok) rowp = s_back[tsrc] + (long long)rrow * rb;
            }
            s_rowp[i] = rowp;
        }
        EVS_WSYNC();
#ifdef EVS_X_ROWS_TIMING
        EVS_TICK(6);
#endif
        // Round 6: the rows of the NEW entries (table -> arena, below) are asked for here, with the rows that go out -- one round
        // trip for both; their stores stay behind the out stream (an entry a later key of this request evicted and refilled may
        // still be the source of an earlier hit's row).  ALL lanes move them, a chunk (16 or 4 bytes, what rows and bases are
        // aligned to) each: the new entries are listed (source, destination) in table order, chunk c of the list's j-th row is
        // lane (j * chunks_per_row + c) mod 64's.  (The first form -- lane t moved table t's row, nine 16-byte loads each -- kept
        // its registers in scratch and waited for every load before the next: ~2 of a request's 8 us.)
        const int fe_t = lane < T ? s_fill[lane] : -1;
        const unsigned long long fill_mask = __ballot(fe_t >= 0);
        const int n_fill = __popcll(fill_mask);
        const int fill_rank = __popcll(fill_mask & ((1ull << lane) - 1ull));
        if (fe_t >= 0) {
            s_fsrc[fill_rank] = s_back[lane] + (long long)s_req[lane] * rb;
            s_fdst[fill_rank] = a.arena + (long long)fe_t * rb;
        }
        EVS_WSYNC();
        const int n_chunks = fill_cs ? n_fill * fill_cpr : 0;
        u32x4_t fill_q[2];   // (a plain vector type: an array of HIP's uint4 -- a struct of unions -- is kept in scratch, with a wait per load)
        unsigned fill_w[4];
        const unsigned char *const fill_dummy = a.arena;   // (what a lane with no chunk reads: 16 valid, aligned bytes)
        if (fill_cs == 16) {
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int idx = lane + 64 * k;
                const int j = __float2int_rz(((float)idx + 0.5f) * fill_inv_cpr);
                const bool on = idx < n_chunks;
                const unsigned char *sp = on ? s_fsrc[on ? j : 0] + (idx - j * fill_cpr) * 16 : fill_dummy;
                fill_q[k] = *reinterpret_cast<const u32x4_t *>(sp);
            }
        } else if (fill_cs == 4) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int idx = lane + 64 * k;
                const int j = __float2int_rz(((float)idx + 0.5f) * fill_inv_cpr);
                const bool on = idx < n_chunks;
                const unsigned char *sp = on ? s_fsrc[on ? j : 0] + (idx - j * fill_cpr) * 4 : fill_dummy;
                fill_w[k] = *reinterpret_cast<const unsigned *>(sp);
            }
        }
        // ... and the T*d elements leave as one flat, independent stream (not T dependent row trips).  Round 6: sixteen elements
        // per lane are asked for before the first is looked at -- the loop over idx was one dependent element load per trip, 15
        // trips for Kaggle's 936 floats: 7 of a request's 17 us (tools/exact_stage_probe.py)
        {
            const int n_el = T * d;
            auto put = [&](int idx, float v) {
                // SERVE: written through (agent scope) -- the ring is read by launches that start while this kernel is still
                // resident, and the alternative is a system-scope fence (an L2 write-back) per request
                if constexpr (SERVE) __hip_atomic_store(out + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else out[idx] = v;
            };
            // branch-free inside (a load behind a divergent branch makes the compiler wait for everything in flight at the branch's
            // end): a lane past the last element re-reads the last one, an absent row reads the arena's first bytes, both
            // discarded by the select below; the codec is wave-uniform -- one loop per codec, nothing to decide inside
            int ti = lane / d, tc = lane - ti * d;     // (table, column) of this lane's first element; + 64 per step
            const int step_i = 64 / d, step_c = 64 - step_i * d;
            const int last_i = T - 1, last_c = d - 1;
            const unsigned char *el_p[16];
            bool have[16];
            int col[16];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const bool in = lane + 64 * k < n_el;
                const int i_ = in ? ti : last_i, c_ = in ? tc : last_c;
                const unsigned char *rowp = s_rowp[i_];
                have[k] = in & (rowp != nullptr);
                el_p[k] = rowp ? rowp : a.arena;
                col[k] = c_;
                ti += step_i; tc += step_c;
                if (tc >= d) { tc -= d; ti++; }
            }
            float val[16];
            if (cs.codec == 32) {
#pragma unroll
                for (int k = 0; k < 16; k++) val[k] = reinterpret_cast<const float *>(el_p[k])[col[k]];
            } else if (cs.codec == 16) {
                unsigned short r_[16];
#pragma unroll
                for (int k = 0; k < 16; k++) r_[k] = reinterpret_cast<const unsigned short *>(el_p[k])[col[k]];
#pragma unroll
                for (int k = 0; k < 16; k++) val[k] = dec_u16(r_[k]);
            } else if (cs.codec == 8) {
                unsigned char r_[16];
#pragma unroll
                for (int k = 0; k < 16; k++) r_[k] = el_p[k][col[k]];
#pragma unroll
                for (int k = 0; k < 16; k++) val[k] = dec_u8(r_[k]);
            } else {
                unsigned char r_[16];
#pragma unroll
                for (int k = 0; k < 16; k++) r_[k] = el_p[k][col[k] >> 1];
#pragma unroll
                for (int k = 0; k < 16; k++) val[k] = kU4Lut[(col[k] & 1) ? (r_[k] & 15u) : (r_[k] >> 4)];
            }
#ifndef EVS_X_NO_ROWS_WAIT
            __builtin_amdgcn_s_waitcnt(0x0F70);
#endif
#ifdef EVS_X_ROWS_TIMING
            EVS_TICK(7);
#endif
#pragma unroll
            for (int k = 0; k < 16; k++)
                if (lane + 64 * k < n_el) put(lane + 64 * k, have[k] ? val[k] : 0.f);
            for (int idx = lane + 1024; idx < n_el; idx += 64) {   // (more than 1 024 elements: the rest one at a time)
                const int i = idx / d, c = idx - i * d;
                const unsigned char *rowp = s_rowp[i];
                put(idx, rowp ? decode_elem(rowp, cs.codec, c) : 0.f);
            }
        }
        EVS_WSYNC();
        EVS_TICK(4);                    // (the rows that go out)
#if EVS_X_EXACT_STOP == 4
        return;
#endif
        if constexpr (SERVE) {
            // ---- the answer: hit flags, then (behind a wait that also puts the ring rows where every later launch sees
            // them) the sequence number the host is polling for.  Round 6: BEFORE the new entries' rows are moved into the
            // arena (below) -- that is this server's own bookkeeping, in program order in front of its next request, and it
            // now runs while the answer travels ----
            served++;
            serve_seq = (serve_seq + 1u) & 0x7fffffffu;
            if (lane < T) const_cast<volatile unsigned char *>(reinterpret_cast<volatile unsigned char *>(sv.ans))[lane] = my_flag;
            // (round 6: the ring rows were written THROUGH, agent scope; what is left to do before the sequence number goes out is to
            //  wait for those stores and the flags -- a system-scope fence here was an L2 write-back per request)
#if EVS_X_SERVE_FENCE == 1      // developer A/B: the round-5 form, a system-scope fence
            __threadfence_system();
#elif EVS_X_SERVE_FENCE == 2    // agent-scope fence
            __threadfence();
#else
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
#endif
#if EVS_X_SERVE_FENCE == 0
            if (lane == 0) __hip_atomic_store(const_cast<unsigned *>(sv.ans) + 16, serve_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (behind the wait above; a release here would be the write-back again)
#else
            if (lane == 0) __hip_atomic_store(const_cast<unsigned *>(sv.ans) + 16, serve_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
        }
        {   // every new entry's row, table -> arena
            // the entry may already have been evicted again by a later key of this request (only if this request evicted at all):
            // its row is not written then
            if (churn) {
                const unsigned long long ki = ((unsigned long long)(lane + 1) << 32) | (unsigned)s_req[lane < T ? lane : 0];
                if (fe_t >= 0 && ld(&a.ekey[fe_t]) != ki) s_fdst[fill_rank] = nullptr;
                EVS_WSYNC();
            }
            if (fill_cs == 16) {
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const int idx = lane + 64 * k;
                    const int j = __float2int_rz(((float)idx + 0.5f) * fill_inv_cpr);
                    unsigned char *dp = idx < n_chunks ? s_fdst[idx < n_chunks ? j : 0] : nullptr;
                    if (dp) *reinterpret_cast<u32x4_t *>(dp + (idx - j * fill_cpr) * 16) = fill_q[k];
                }
                for (int idx = lane + 128; idx < n_chunks; idx += 64) {    // (more than 128 chunks: the rest one at a time)
                    const int j = idx / fill_cpr, c = idx - j * fill_cpr;
                    unsigned char *dp = s_fdst[j];
                    if (dp) *reinterpret_cast<u32x4_t *>(dp + c * 16) = *reinterpret_cast<const u32x4_t *>(s_fsrc[j] + c * 16);
                }
            } else if (fill_cs == 4) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int idx = lane + 64 * k;
                    const int j = __float2int_rz(((float)idx + 0.5f) * fill_inv_cpr);
                    unsigned char *dp = idx < n_chunks ? s_fdst[idx < n_chunks ? j : 0] : nullptr;
                    if (dp) *reinterpret_cast<unsigned *>(dp + (idx - j * fill_cpr) * 4) = fill_w[k];
                }
                for (int idx = lane + 256; idx < n_chunks; idx += 64) {
                    const int j = idx / fill_cpr, c = idx - j * fill_cpr;
                    unsigned char *dp = s_fdst[j];
                    if (dp) *reinterpret_cast<unsigned *>(dp + c * 4) = *reinterpret_cast<const unsigned *>(s_fsrc[j] + c * 4);
                }
            } else if (fe_t >= 0 && s_fdst[fill_rank]) {   // rows or bases not even 4-byte aligned: lane t moves table t's, a byte at a time
                const unsigned char *rowp = s_fsrc[fill_rank];
                unsigned char *dst = s_fdst[fill_rank];
                for (int c = 0; c < rb; c++) dst[c] = rowp[c];
            }
        }
        __threadfence_block();   // (the rows are read back by this block only: a workgroup-scope fence; an agent-scope one is an L2 write-back per request, tools/atomic_probe.hip)
        EVS_WSYNC();
        EVS_TICK(5);                    // (the new entries' rows)
    }

#ifdef EVS_X_EXACT_TIMING
    if (lane == 0) for (int k = 0; k < 8; k++) g_exact_ticks[k] += tick_acc[k];
#endif
    asm volatile("" :: "v"(warm_acc));
    if (lane == 0) {
        gs->min_c1 = h.min_c1; gs->n_perfect = h.n_perfect; gs->count = h.count; gs->n_free = h.n_free;
        gs->n_flush = h.n_flush; gs->n_evict = h.n_evict; gs->least_freq = h.least_freq; gs->error = h.error;
        gs->n_requests = cs.n_requests + (SERVE ? served : args.B);
        gs->n_hits = cs.n_hits + n_hits;
        gs->n_perfect_hits = cs.n_perfect_hits + n_perfect_hits;
    }
    for (int b = lane; b < kMaxBuckets; b += 64) { gs->head[b] = h.head[b]; gs->tail[b] = h.tail[b]; gs->len[b] = h.len[b]; }
    if constexpr (SERVE) {   // the state is back in HBM: only now may the host start another server (or anything else) on it
        __threadfence_system();
        if (lane == 0) __hip_atomic_store(const_cast<unsigned *>(sv.ans) + 17, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ void __launch_bounds__(64) cache_exact_kernel(const CacheArgs args) { cache_exact_body<false>(args, ServeArgs{}); }
__global__ void __launch_bounds__(64) cache_serve_kernel(const CacheArgs args, const ServeArgs sv) { cache_exact_body<true>(args, sv); }


// ------------------------------------------------------------------------------------------
// a9: two-tier request, C1 (main precision) + C2 (secondary precision)
// mixed_precs_caching/evlfu_8.cpp:669-796 request_to_c1_c2 with evlfu_4.cpp phase_1 / phase_2
// as the C2 half.  One wavefront, requests in order; both tiers use the exact EvLFU machinery
// above.  Eviction victims are FIFO-oldest (the C++ takes unordered_set::begin()); a C1 hit whose
// entry was evicted earlier in the same request is served from storage (the C++ dereferences a
// dangling pointer there, evlfu_8.cpp:521-522).
// ------------------------------------------------------------------------------------------
// ---- a12: alt-key ("approximate embedding") tier, C3 -- deterministic re-specification --------------
// mixed_precs_caching/aprx_embedding.cpp (map key -> {alt_key, recency flag}, FIFO with second-chance
// eviction, batches of 50 evicted keys) and evlfu_8.cpp:474-490,533-558 (lookup on a double miss).
// The reference fills this tier from asynchronous threads; here a full batch is inserted at the start
// of the next request (oracle/evstore_oracle.c:orc_c1c2c3_request is the same specification).
constexpr int kAprxBatch = 50;   // IO_JOB_Q_SIZE (aprx_embedding.hpp:29)
struct AprxState {
    int cap, count, n_free, n_pending, batch_ready, error;
    unsigned long long mask;
    long long qcap, qhead, qtail, n_hit;
    unsigned long long pending[kAprxBatch];
};
struct AprxArrays {
    AprxState *st;
    unsigned long long *keys; int *slot_entry;       // hash: key -> entry
    unsigned long long *ekey; unsigned *ealt; unsigned char *eflag; int *free_stack;
    unsigned long long *queue;
    const unsigned *alt_tables[kMaxTables];           // alt_key[row] = alt_row*100 + alt_table (native endian)
    long long alt_rows[kMaxTables];
};
__device__ int aprx_find(const AprxArrays &x, unsigned long long mask, unsigned long long key) {
    unsigned long long i = mix64(key) & mask;
    for (;;) {
        const unsigned long long k = ld(&x.keys[i]);
        if (k == key) return ld(&x.slot_entry[i]);
        if (k == kEmpty) return -1;
        i = (i + 1) & mask;
    }
}
__device__ void aprx_put(const AprxArrays &x, unsigned long long mask, unsigned long long key, int e) {
    unsigned long long i = mix64(key) & mask;
    for (;;) {
        const unsigned long long k = ld(&x.keys[i]);
        if (k == kEmpty || k == key) break;
        i = (i + 1) & mask;
    }
    st(&x.slot_entry[i], e);
    st(&x.keys[i], key);
}
__device__ void aprx_del(const AprxArrays &x, unsigned long long mask, unsigned long long key) {
    unsigned long long i = mix64(key) & mask;
    for (;;) {
        const unsigned long long k = ld(&x.keys[i]);
        if (k == key) break;
        if (k == kEmpty) return;
        i = (i + 1) & mask;
    }
    unsigned long long j = i;
    for (;;) {
        j = (j + 1) & mask;
        const unsigned long long kj = ld(&x.keys[j]);
        if (kj == kEmpty) break;
        const unsigned long long h = mix64(kj) & mask;
        const bool between = (i <= j) ? (h > i && h <= j) : (h > i || h <= j);
        if (!between) { st(&x.slot_entry[i], ld(&x.slot_entry[j])); st(&x.keys[i], kj); i = j; }
    }
    st(&x.keys[i], kEmpty);
}
__device__ void aprx_evict_one(const AprxArrays &x, AprxState &s) {  // recency_aware_eviction (aprx_embedding.cpp:360-388)
    while (s.qhead < s.qtail) {
        const unsigned long long key = x.queue[s.qhead % s.qcap];
        const int e = aprx_find(x, s.mask, key);
        if (e >= 0) {
            if (ld(&x.eflag[e])) {  // second chance
                st(&x.eflag[e], (unsigned char)0);
                x.queue[s.qtail % s.qcap] = key; s.qtail++;
                s.qhead++;
            } else {
                aprx_del(x, s.mask, key);
                x.free_stack[s.n_free++] = e;
                s.count--; s.qhead++;
                return;
            }
        } else {
            s.qhead++;  // stale duplicate
        }
    }
}
__device__ void aprx_insert_batch(const AprxArrays &x, AprxState &s) {  // insert_altkey_batched_obj (:308-324)
    const int n_erase = s.count + kAprxBatch - s.cap;
    for (int i = 0; i < n_erase; i++) aprx_evict_one(x, s);
    for (int i = 0; i < kAprxBatch; i++) {
        const unsigned long long key = s.pending[i];
        const int t = (int)(key >> 32) - 1;
        const unsigned alt = x.alt_tables[t][(unsigned)(key & 0xffffffffull)];
        if (s.qtail - s.qhead >= s.qcap) { s.error = 1; return; }
        x.queue[s.qtail % s.qcap] = key; s.qtail++;
        int e = aprx_find(x, s.mask, key);
        if (e < 0) {
            if (s.n_free <= 0) { s.error = 2; return; }
            e = x.free_stack[--s.n_free];
            st(&x.ekey[e], key);
            aprx_put(x, s.mask, key, e);
            s.count++;
        }
        st(&x.ealt[e], alt);
        st(&x.eflag[e], (unsigned char)0);
    }
    s.n_pending = 0; s.batch_ready = 0;
}
__device__ __forceinline__ void aprx_queue_key(AprxState &s, unsigned long long key) {
    if (!key) return;
    if (s.n_pending < kAprxBatch) s.pending[s.n_pending++] = key;
    if (s.n_pending == kAprxBatch) s.batch_ready = 1;
}
// APRX_EV's public single-key methods in order, one lane (the tier is sequential by definition): op 0 insert_altkey
// (aprx_embedding.cpp:278-288), 1 get_altkey_str (:341-350), 2 set_recency_flag_c3 (:402-411), 3 evict_one_key (:390-400).
// These are the operations of the tier that ARE pinned to the compiled reference (tests/golden/aprx_ops.npz); the
// request path above uses the same device functions.
__global__ void aprx_ops_kernel(AprxArrays x, long long n, const int *ops, unsigned *res) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    AprxState s = *x.st;
    for (long long i = 0; i < n && !s.error; i++) {
        const int op = ops[3 * i], t1 = ops[3 * i + 1], row = ops[3 * i + 2];
        unsigned out = 0;
        if (t1 < 1 || t1 > kMaxTables || row < 0 || (op == 0 && (long long)row >= x.alt_rows[t1 - 1])) { s.error = 3; break; }
        const unsigned long long key = ((unsigned long long)(unsigned)t1 << 32) | (unsigned)row;
        if (op == 0) {
            const unsigned alt = x.alt_tables[t1 - 1][row];
            if (s.count >= s.cap) aprx_evict_one(x, s);
            if (s.qtail - s.qhead >= s.qcap) { s.error = 1; break; }
            x.queue[s.qtail % s.qcap] = key; s.qtail++;
            int e = aprx_find(x, s.mask, key);
            if (e < 0) {
                if (s.n_free <= 0) { s.error = 2; break; }
                e = x.free_stack[--s.n_free];
                st(&x.ekey[e], key);
                aprx_put(x, s.mask, key, e);
                s.count++;
            }
            st(&x.ealt[e], alt);
            st(&x.eflag[e], (unsigned char)0);
            out = alt;
        } else if (op == 1) {
            const int e = aprx_find(x, s.mask, key);
            out = e >= 0 ? ld(&x.ealt[e]) : 0xffffffffu;
        } else if (op == 2) {
            const int e = aprx_find(x, s.mask, key);
            if (e >= 0) st(&x.eflag[e], (unsigned char)1);
        } else if (op == 3) {
            aprx_evict_one(x, s);
        }
        res[i] = out;
    }
    *x.st = s;
}

struct TierArgs {
    CacheState *st;
    CacheArrays a;
    const unsigned char *backing[kMaxTables];
    long long backing_rows[kMaxTables];
};
struct C1C2Args {
    TierArgs t1, t2;
    AprxArrays c3;            // c3.st == nullptr: no alt-key tier
    const int *requests;
    float *out;
    unsigned char *tier_out;  // (B,T): 1 = C1 hit, 2 = served by a C2 hit, 0 = miss
    long long B;
    int threshold;            // high_agghit_threshold (evlfu_8.hpp:70)
};

__device__ void hot_load(Hot &h, const CacheState &cs, const CacheState *gs, int lane) {  // arrays straight from memory: a dynamic index into the by-value copy would put it in scratch
    if (lane == 0) {
        h.min_c1 = cs.min_c1; h.n_perfect = cs.n_perfect; h.count = cs.count; h.n_free = cs.n_free;
        h.n_flush = cs.n_flush; h.n_evict = cs.n_evict; h.least_freq = cs.least_freq; h.error = cs.error;
    }
    for (int b = lane; b < kMaxBuckets; b += 64) { h.head[b] = gs->head[b]; h.tail[b] = gs->tail[b]; h.len[b] = gs->len[b]; }
}
__device__ void hot_store(const Hot &h, CacheState *gs, int lane) {
    if (lane == 0) {
        gs->min_c1 = h.min_c1; gs->n_perfect = h.n_perfect; gs->count = h.count; gs->n_free = h.n_free;
        gs->n_flush = h.n_flush; gs->n_evict = h.n_evict; gs->least_freq = h.least_freq; gs->error = h.error;
    }
    for (int b = lane; b < kMaxBuckets; b += 64) { gs->head[b] = h.head[b]; gs->tail[b] = h.tail[b]; gs->len[b] = h.len[b]; }
}
__device__ __forceinline__ void touch(const CacheArrays &a, Hot &h, int e, int agg) {  // update_agg_hit
    const int old = ld(&a.eagg[e]);
    if (old < agg) {
        list_unlink(a, bucket(h, old), e);
        list_append(a, bucket(h, agg), e);
        st(&a.eagg[e], agg);
    }
}

__global__ void __launch_bounds__(64) cache_c1c2_kernel(const C1C2Args args) {
    __shared__ Hot h1, h2;
    __shared__ AprxState s3;
    __shared__ int s_alt_e[kMaxTables], s_alt_tier[kMaxTables];
    const bool has_c3 = args.c3.st != nullptr;
    // per key: which tier's row to output (1/2), from its arena entry (>=0) or its backing store (-1);
    // and the entry to fill afterwards per tier
    __shared__ int s_tier[kMaxTables], s_src[kMaxTables], s_fill1[kMaxTables], s_fill2[kMaxTables], s_req[kMaxTables];
    const int lane = threadIdx.x;
    const CacheState cs1 = *args.t1.st, cs2 = *args.t2.st;
    const CacheArrays a1 = args.t1.a, a2 = args.t2.a;
    const unsigned long long m1 = cs1.nslot_mask, m2 = cs2.nslot_mask;
    const int T = cs1.n_tables, d = cs1.dim;
    hot_load(h1, cs1, args.t1.st, lane);
    hot_load(h2, cs2, args.t2.st, lane);
    if (has_c3 && lane == 0) s3 = *args.c3.st;
    __syncthreads();
    long long n_perfect_req = 0, n_hits = 0;

    for (long long rq = 0; rq < args.B; rq++) {
        const int row = lane < T ? args.requests[rq * T + lane] : 0;
        if (lane < T) s_req[lane] = row;
        const unsigned long long key = ((unsigned long long)(lane + 1) << 32) | (unsigned)row;
        const bool ok = lane < T && row >= 0 && row < args.t1.backing_rows[lane < T ? lane : 0] &&
                        row < args.t2.backing_rows[lane < T ? lane : 0];
        if (has_c3 && lane == 0 && s3.batch_ready) aprx_insert_batch(args.c3, s3);  // a full batch becomes visible here
        __syncthreads();
        const int e2 = ok ? map_find(a2, m2, key) : -1;  // evlfu_4.cpp phase_1_find_keys_in_cache
        const int e1 = ok ? map_find(a1, m1, key) : -1;
        const unsigned long long okm = __ballot(ok);
        const unsigned long long hit2 = __ballot(e2 >= 0), hit1 = __ballot(e1 >= 0);
        // alt-key probe for double misses (find_approximate_ev, evlfu_8.cpp:474-490): C1 first, then C2
        int alt_e = -1, alt_tier = 0;
        if (has_c3 && ok && e1 < 0 && e2 < 0) {
            const int e3 = aprx_find(args.c3, s3.mask, key);
            if (e3 >= 0) {
                const unsigned alt = ld(&args.c3.ealt[e3]);
                const unsigned long long akey = ((unsigned long long)(alt % 100u) << 32) | (alt / 100u);
                int ea = map_find(a1, m1, akey);
                if (ea >= 0) { alt_tier = 1; alt_e = ea; }
                else { ea = map_find(a2, m2, akey); if (ea >= 0) { alt_tier = 2; alt_e = ea; } }
                if (alt_tier) st(&args.c3.eflag[e3], (unsigned char)1);  // set_recency_flag_c3
            }
        }
        const unsigned long long hit3 = __ballot(alt_tier != 0);
        if (lane < T) { s_alt_e[lane] = alt_e; s_alt_tier[lane] = alt_tier; }
        int agg = __popcll(hit2) + __popcll(hit1 & ~hit2) + __popcll(hit3);  // c1_c2_agg_hit (evlfu_8.cpp:513-546)
        const unsigned long long hit1e = hit1 | hit3;        // c1_arr_record_hit is piggybacked for alt hits (:547)
        unsigned long long upd2 = ~hit1 & hit2 & okm;        // C1 miss, C2 hit: C2 serves and updates
        unsigned long long ins2 = ~hit1 & ~hit2 & ~hit3 & okm;  // double miss without an alt row: C2 inserts (Cond 259) ...
        unsigned long long job1 = 0;
        bool update_c2 = true;
        if (h1.count >= cs1.cap) {                           // :570-586 C1 full
            if (agg < args.threshold) {                      // ... unless C1 takes the odd ones
                const unsigned long long odd = 0xAAAAAAAAAAAAAAAAull;
                job1 = ~hit2 & ~hit1e & odd & okm;
                ins2 &= ~odd;
            }
        } else {                                             // :587-601 C1 not full: all C1 misses go to C1
            job1 = ~hit1e & okm;
            update_c2 = false;
            agg = __popcll(hit1);
        }
        if (lane == 0 && has_c3) s3.n_hit += __popcll(hit3);
        if (lane < T) {
            s_tier[lane] = 0; s_src[lane] = -1; s_fill1[lane] = -1; s_fill2[lane] = -1;
            args.tier_out[rq * T + lane] = (hit1 >> lane) & 1 ? 1 : ((hit2 >> lane) & 1 ? 2 : ((hit3 >> lane) & 1 ? 3 : 0));
        }
        __syncthreads();
        for (int i = 0; i < T; i++) {  // C2 first (phase 2), updates before inserts (evlfu_4.cpp:374-400)
            const int e2i = __shfl(e2, i);
            if (lane == 0 && update_c2 && ((upd2 >> i) & 1)) {
                touch(a2, h2, e2i, agg);
                s_tier[i] = 2; s_src[i] = e2i;
            }
        }
        for (int i = 0; i < T; i++) {
            const unsigned long long ki = __shfl(key, i);
            if (lane == 0 && update_c2 && ((ins2 >> i) & 1)) {
                s_fill2[i] = evlfu_set(cs2, a2, h2, m2, ki, agg);
                if (has_c3) aprx_queue_key(s3, h2.last_evicted);  // evlfu_8.cpp:617-620
                s_tier[i] = 2; s_src[i] = -1;
            }
        }
        if (lane == 0 && update_c2 && agg == T) h2.n_perfect = h2.len[T];
        for (int i = 0; i < T; i++) {  // C1 loop (evlfu_8.cpp:769-785)
            const int e1i = __shfl(e1, i);
            const unsigned long long ki = __shfl(key, i);
            if (lane == 0) {
                if ((hit1 >> i) & 1) {
                    s_tier[i] = 1;
                    if (ld(&a1.ekey[e1i]) == ki) { touch(a1, h1, e1i, agg); s_src[i] = e1i; }
                    else s_src[i] = -1;  // evicted earlier in this request
                } else if ((hit3 >> i) & 1) {
                    s_tier[i] = 2 + s_alt_tier[i];  // 3: alt row in C1's arena, 4: in C2's arena
                    s_src[i] = s_alt_e[i];
                } else if ((job1 >> i) & 1) {
                    s_fill1[i] = evlfu_set(cs1, a1, h1, m1, ki, agg);
                    if (has_c3) aprx_queue_key(s3, h1.last_evicted);  // evlfu_8.cpp:654-658
                    s_tier[i] = 1; s_src[i] = -1;
                }
            }
        }
        if (lane == 0 && agg == T) h1.n_perfect = h1.len[T];
        n_perfect_req += (agg == T);
        n_hits += __popcll((hit1 | hit2 | hit3) & okm);
        __syncthreads();

        float *out = args.out + rq * (long long)T * d;
        for (int i = 0; i < T; i++) {
            const int tier = s_tier[i], src = s_src[i];
            const int rrow = s_req[i];
            const unsigned char *rowp = nullptr;
            int codec = cs1.codec;
            if (tier == 1) rowp = src >= 0 ? a1.arena + (long long)src * cs1.row_bytes
                                           : args.t1.backing[i] + (long long)rrow * cs1.row_bytes;
            else if (tier == 2) {
                codec = cs2.codec;
                rowp = src >= 0 ? a2.arena + (long long)src * cs2.row_bytes
                                : args.t2.backing[i] + (long long)rrow * cs2.row_bytes;
            } else if (tier == 3) {  // alt-key hit: the ALT row, decoded at the precision of the tier holding it
                rowp = a1.arena + (long long)src * cs1.row_bytes;
            } else if (tier == 4) {
                codec = cs2.codec;
                rowp = a2.arena + (long long)src * cs2.row_bytes;
            }
            for (int c = lane; c < d; c += 64) out[i * d + c] = rowp ? decode_elem(rowp, codec, c) : 0.f;
        }
        __syncthreads();
        for (int i = 0; i < T; i++) {
            const unsigned long long ki = ((unsigned long long)(i + 1) << 32) | (unsigned)s_req[i];
            const int f1 = s_fill1[i], f2 = s_fill2[i];
            if (f1 >= 0 && ld(&a1.ekey[f1]) == ki) {
                const unsigned char *rowp = args.t1.backing[i] + (long long)s_req[i] * cs1.row_bytes;
                unsigned char *dst = a1.arena + (long long)f1 * cs1.row_bytes;
                for (int c = lane; c < cs1.row_bytes; c += 64) dst[c] = rowp[c];
            }
            if (f2 >= 0 && ld(&a2.ekey[f2]) == ki) {
                const unsigned char *rowp = args.t2.backing[i] + (long long)s_req[i] * cs2.row_bytes;
                unsigned char *dst = a2.arena + (long long)f2 * cs2.row_bytes;
                for (int c = lane; c < cs2.row_bytes; c += 64) dst[c] = rowp[c];
            }
        }
        __threadfence_block();   // (as in the one-tier kernel)
        __syncthreads();
    }
    hot_store(h1, args.t1.st, lane);
    hot_store(h2, args.t2.st, lane);
    if (has_c3 && lane == 0) *args.c3.st = s3;
    if (lane == 0) {
        args.t1.st->n_requests = cs1.n_requests + args.B;
        args.t1.st->n_perfect_hits = cs1.n_perfect_hits + n_perfect_req;
        args.t1.st->n_hits = cs1.n_hits + n_hits;
    }
}


// ------------------------------------------------------------------------------------------
// Batched path (throughput): snapshot semantics.
//
// EvLFU is defined request by request; replaying a 16 384-request batch through one wavefront
// would take milliseconds.  The batched path keeps what the policy is FOR and relaxes what only a
// sequential machine can give:
//   * every key of the batch is probed against the table as it was when the batch started
//     (64 keys per wave step, one lane per key; agg_hit of a request = popcount of its lanes'
//     ballot) and its row is served at once -- arena row for a hit, backing row for a miss:
//     values are always exactly the table rows;
//   * hits raise their entry's priority to max(old, agg_hit) with atomicMax (monotone, like
//     update_agg_hit); misses are de-duplicated through the hash itself (CAS on the slot) and
//     inserted with priority = max agg_hit over the requests that missed them;
//   * room is made by evicting the lowest priorities first (a histogram of the 27 priorities gives
//     the cut; inside the cut priority the scan order decides, not strict FIFO), and the EvLFU flush
//     (top priority holding >= 95 % of the capacity -> drop 30 % of it) is applied per batch.
// Invariants (tests/test_gpu_cache.py): rows exact; no key twice; size <= capacity; the priority
// histogram matches the entries; priorities never decrease while resident.  The hit rate tracks
// the sequential oracle's within a few percent on Zipf streams.
// The priority lists of the exact path are NOT maintained here: a cache object is used either
// exactly (evs_cache_request) or batched (evs_cache_lookup_batch), never both.
// ------------------------------------------------------------------------------------------

struct BatchState {
    int n_miss, n_new, n_free, count, n_tomb;
    int cnt[kMaxBuckets];
    int pstar, need, ticket, flush_t, ticket_t, pos_ticket, do_rebuild, n_assign;
    int hand, win;   // eviction scans the entry window [hand, hand + win) (mod cap), then the hand moves on
    int n_orphan;    // file mode: missed keys of staged tables the full hash could not take (served from staging, never cached)
    int flush_gone;  // sampled flush: entries the scan kernel removed, folded into the counters by its finish kernel
    long long batch_id, n_hits, n_requests, n_perfect_hits, n_evict, n_flush;
};

struct BatchArgs {
    BatchState *bs;
    CacheArrays a;
    unsigned long long *slots;   // packed hash words (see above)
    int *estamp; int stamp;      // entries with estamp[e] == stamp are not evicted in the running batch: its hits when the miss
    int stamp_hits;              // tier is host memory (stamp_hits), and what the sampled update inserted in it
    const unsigned long long *other_slots; unsigned long long other_mask;   // two-tier: the other tier's hash (keys it holds are skipped)
    // two-tier, sampled update: the probe stamps (hash of) every key it routes to C1 into a filter; C2's update skips a key
    // whose filter word carries this batch's stamp -- the two tiers' updates no longer depend on each other (one launch).
    // A collision only skips an insert (the key is not cached this time); 2^20 words against ~10^4 marks per batch.
    unsigned *route_filter; unsigned route_mask, route_stamp;
    int *eslot;            // hash slot of each entry
    // per (request, table) position: bit 31 = valid miss, bit 30 = hit (of a miss: the hinted slot is a tombstone), bits 24..29 = agg_hit of the request,
    // bits 0..23 = (empty hash slot the probe of a miss ended on) >> hint_shift.  One 4-byte store per key in
    // K1 (with the 8-byte row address) instead of five streams: the stores were 11 us of its 18.7.
    unsigned *miss_info;
    int hint_shift;
    int *new_slot;                                 // unique new keys (hash slots), 256 per K2 block
    int *block_cnt, *block_base;                   // per K2 block: number of new keys, exclusive scan
    int *part1, *part2;                            // replica rows of K1 / K5 totals (kReplicas x kPartCols each)
    int *host_tomb;                                // mapped host word: tombstone count after this batch
    int g1, g2;
    long long *row_ptrs;                           // (B,T) address of each key's row (arena / backing / 0)
    int *row_ids;                                  // (B,T), instead of row_ptrs when the rows-in-registers consumer follows: bit 30 = arena entry, else the table row, -1 = none
    const unsigned char *backing[kMaxTables];
    long long backing_rows[kMaxTables];
    const int *requests; float *out; unsigned char *hit;
    long long B;
    unsigned long long mask;
    int cap, T, d, codec, row_bytes, max_perfect, flush_n, nslot;
    // file-backed miss tier (evs_filetier.hip): tables in staged_mask have no device-visible address; the rows of the
    // batch's new keys are copied by the host into `staging` (row i = new key i of the batch, pinned + mapped)
    unsigned staged_mask;
    const unsigned char *staging;
    unsigned long long *new_keys;   // new key i of the batch (table_1based << 32 | row), for the host's reader pool;
                                    // orphans (n_orphan) are listed from the END of the array downwards
    long long stage_rows;           // rows of `staging` / entries of new_keys (= B * T)
    int *slot_stage;                // hash slot -> index of the new key it holds in this batch
    int rebuild;                    // after this batch's close the host 1: rebuilds the hash (no tombstones remain), 2: sweeps it
    // sampled update, tables in HBM: K1 block j lists its misses itself (records of 16 bytes: row, table | agg << 8 | hint
    // kind << 16, hinted slot, request position) at miss_rec + j * list_cap, their number in list_cnt[j]; the update
    // kernel runs one wave per list instead of one thread per (request, table) position
    uint4 *miss_rec; int *list_cnt; int list_cap;
    // sampled update with the alt-key tier attached: kReplicas victim lists of vict_cap keys each in evicted_keys, their
    // lengths in vict_cnt (zeroed by the consumer kernel)
    int *vict_cnt, *vict_other; int vict_cap;
    // ... or the evicting thread makes its victim a member of the alt-key set itself, behind its own stores (no lists, no
    // insert launch): the set's key words / number of sets / counters
    unsigned long long *c3_tags; long long c3_nset; long long *c3_stat;
    unsigned long long *evicted_keys;   // (alt-key tier attached) the key each free-stack position held before K4 put it there; bit 63 = flushed, not evicted
    int tomb_parity;                // sampled update: parity of this batch (its tombstones are kTomb1 when odd); -1 otherwise
    SaGeom sa; SaUniverse sau;      // set-associative policy (evs_hash.h): this tier's set records (sa.tags == nullptr otherwise), arena row = set * sa.ways + way
};

// One atomic per BLOCK instead of one per thread: every thread of the block calls this in uniform
// control flow; threads with flag set get consecutive indices starting at the value the block
// reserved from *counter (a 426k-way contended counter would cost milliseconds).
__device__ __forceinline__ int block_reserve(int *counter, bool flag, int *s_tot /* >= 8 ints of LDS */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const unsigned long long m = __ballot(flag);
    const int rank = __popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) s_tot[wave] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < nw; w++) { const int c = s_tot[w]; s_tot[w] = tot; tot += c; }
        s_tot[nw] = tot ? atomicAdd(counter, tot) : 0;
    }
    __syncthreads();
    return s_tot[nw] + s_tot[wave] + rank;
}

// Same with a per-thread COUNT: thread gets the base index of its n consecutive tickets.
__device__ __forceinline__ int block_reserve_n(int *counter, int n, int *s_tot /* >= 8 ints of LDS */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int incl = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    __syncthreads();
    if (lane == 63) s_tot[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < nw; w++) { const int c = s_tot[w]; s_tot[w] = tot; tot += c; }
        s_tot[nw] = tot ? atomicAdd(counter, tot) : 0;
    }
    __syncthreads();
    return s_tot[nw] + s_tot[wave] + incl - n;
}

// Shared counters: a device-scope atomic on ONE address costs ~10 ns and same-address atomics serialise, so
// 1 664 blocks each adding their hit totals and histogram moves took 29 us of a 46 us kernel.  The batch
// kernels therefore add their per-block totals into kReplicas replica rows (block j -> row j % kReplicas,
// 160 B apart, fire-and-forget atomics: a handful per address) that the next single-block kernel folds and
// clears: K1 -> K3 (plan), K5 -> K6b (end-of-batch bookkeeping).  (Per-block rows folded by one block were
// no better: a thread walking 277 rows is 277 dependent round trips.)
constexpr int kPartCols = 40;   // 0..32 priority-histogram deltas, 36 recycled tombstones, 37 dropped, 38 hits, 39 perfect requests
constexpr int kReplicas = 32;
constexpr int kProbeGridMax = 8192;   // one request pair per wave up to B = 65 536, then the blocks loop

// K1: one 32-lane half-wave per request (T <= 32): probe, agg_hit by ballot, priority bump, miss record,
// and the address of every key's row ((B,T) pointer table consumed by the fused interaction kernel in
// pointer mode, or by cache_rows_from_ptrs_kernel when the rows are wanted).  Misses go to the fixed slot
// (request, table) of the miss arrays.  A block walks requests blockIdx*8, +gridDim*8, ...
__global__ void __launch_bounds__(256) cache_batch_probe_gather_kernel(const BatchArgs args) {
    __shared__ int s_delta[kMaxBuckets];
    __shared__ int s_sum[2];   // hits / perfect requests of this block
    __shared__ int s_list_n;   // list mode: misses this block has listed so far
    for (int i = threadIdx.x; i < kMaxBuckets; i += blockDim.x) s_delta[i] = 0;
    if (threadIdx.x < 2) s_sum[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_list_n = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, half = lane >> 5, hl = lane & 31;
    const int T = args.T;
    const long long req_stride = (long long)gridDim.x * 8;
    for (long long req = (long long)blockIdx.x * 8 + (threadIdx.x >> 6) * 2 + half; req - half - (threadIdx.x >> 6) * 2 < args.B;
         req += req_stride) {
        const bool req_on = req < args.B;
        const bool key_on = req_on && hl < T;
        int row = key_on ? args.requests[req * T + hl] : -1;
        const bool ok = key_on && row >= 0 && row < args.backing_rows[hl < T ? hl : 0];
        const unsigned long long key = ((unsigned long long)(hl + 1) << 32) | (unsigned)row;
        unsigned long long end_slot = 0;
        bool hint_tomb = false;
        int e = -1;
        unsigned sa_w = 0u, sa_tag1 = 0u;
        int sa_way = -1;
        if (args.sa.tags) {   // set-associative policy: one line, the priority inside the word
            unsigned set = 0u;
            sa_split(args.sa, sa_perm(args.sau, args.sau.row_base[hl] + (ok ? (unsigned)row : 0u)), set, sa_tag1);
            if (!ok) set = 0u;
            SaLine line;
            int way;
            if (args.sa.ways == 8u) { sa_load<8>(args.sa, set, line); way = sa_find<8>(args.sa, line, sa_tag1, sa_w); }
            else { sa_load<0>(args.sa, set, line); way = sa_find<0>(args.sa, line, sa_tag1, sa_w); }
            if (ok && way >= 0) e = (int)sa_entry(args.sa, set, (unsigned)way, sa_w);
            end_slot = set; sa_way = way;
        } else {
            e = ok ? probe_ro(args.slots, args.mask, key, end_slot, args.tomb_parity == 1 ? kTomb : args.tomb_parity == 0 ? kTomb1 : kTomb, &hint_tomb) : -1;
            if (e == kPending) e = -1;
        }
        const unsigned long long hm = __ballot(e >= 0);
        const unsigned hmask = (unsigned)(half ? (hm >> 32) : hm);
        const int agg = __popc(hmask);
        if (args.sa.tags) {
            if (e >= 0 && sa_prio(sa_w) < agg) {
                const int old = sa_raise(args.sa, sa_ways_ptr(args.sa, (unsigned)end_slot) + sa_way, sa_w, agg);
                if (old >= 0) { atomicSub(&s_delta[old], 1); atomicAdd(&s_delta[agg], 1); }
            }
        } else {
        // monotone max like update_agg_hit; the plain read first keeps hot entries (thousands of requests of
        // one batch hit the same few rows of the tiny tables) from serialising on one atomic address
#ifndef EVS_X_NOPRIO   // developer A/B (timing only): what the priority read per hit costs the probe
        if (e >= 0 && args.a.eagg[e] < agg) {
            const int old = atomicMax(&args.a.eagg[e], agg);
            if (old < agg) { atomicSub(&s_delta[old], 1); atomicAdd(&s_delta[agg], 1); }
        }
#endif
        }
        if (args.stamp_hits && e >= 0) args.estamp[e] = args.stamp;
        // where the row lives right now: arena for a hit, backing store for a miss, 0 for a bad row id
        const unsigned char *src = nullptr;
        if (e >= 0) src = args.a.arena + (long long)e * args.row_bytes;
        else if (ok) src = args.backing[hl] + (long long)row * args.row_bytes;
        if (key_on) {
            // bit 30: hit -- or, for a miss (bit 31), "the hinted slot is a tombstone" (what the insert's first CAS expects)
            const unsigned info = ((ok && e < 0) ? 0x80000000u : 0u) | ((e >= 0 || (ok && hint_tomb)) ? 0x40000000u : 0u) | ((unsigned)agg << 24) |
                                  (unsigned)(end_slot >> args.hint_shift);
            if (!args.miss_rec) args.miss_info[req * T + hl] = info;
            else if (args.hit) args.hit[req * T + hl] = e >= 0;
            if (args.row_ids) args.row_ids[req * T + hl] = e >= 0 ? (int)(0x40000000u | (unsigned)e) : (ok ? row : -1);
            else args.row_ptrs[req * T + hl] = (long long)src;
        }
        if (args.miss_rec) {   // list mode: the half-wave's misses, packed, behind the block's earlier ones
            const bool is_miss = ok && e < 0;
            const unsigned long long mm = __ballot(is_miss);
            const unsigned mh = (unsigned)(half ? (mm >> 32) : mm);
            int base = 0;
            if (hl == 0 && mh) base = atomicAdd(&s_list_n, __popc(mh));
            base = __shfl(base, half * 32, 64);
            if (is_miss) {
                const int at = base + __popc(mh & ((1u << hl) - 1u));
                args.miss_rec[(long long)blockIdx.x * args.list_cap + at] =
                    make_uint4((unsigned)row, (unsigned)hl | ((unsigned)agg << 8) | (hint_tomb ? 0x10000u : 0u),
                               (unsigned)(end_slot >> args.hint_shift), args.sa.tags ? sa_tag1 : (unsigned)(req * T + hl));   // (set-associative lists: the set and the tag ARE the key)
            }
        }
        if (req_on && hl == 0) { atomicAdd(&s_sum[0], agg); if (agg == T) atomicAdd(&s_sum[1], 1); }
    }
    __syncthreads();
    if (threadIdx.x < kPartCols) {
        const int i = threadIdx.x;
        const int v = i <= T ? s_delta[i] : i == 38 ? s_sum[0] : i == 39 ? s_sum[1] : 0;
        if (v) atomicAdd(&args.part1[(blockIdx.x % kReplicas) * kPartCols + i], v);
    }
    if (args.miss_rec && threadIdx.x == 0) args.list_cnt[blockIdx.x] = s_list_n;
}

// rows (B,T,d) fp32 from the pointer table (only when the caller wants the pooled rows themselves)
__global__ void __launch_bounds__(256) cache_rows_from_ptrs_kernel(const long long *row_ptrs, float *out, long long B,
                                                                   int T, int d, int codec) {
    if (codec == 32 && (d & 3) == 0) {
        const int lpr = d >> 2;
        const long long n = B * T * lpr;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
            const long long rowi = i / lpr;
            const int piece = (int)(i - rowi * lpr);
            const long long b = rowi / T;
            const int t = (int)(rowi - b * T);
            const long long p = row_ptrs[b * T + t];
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p) v = reinterpret_cast<const float4 *>(p)[piece];
            reinterpret_cast<float4 *>(out)[i] = v;
        }
    } else {
        const long long n = B * T * d;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
            const long long rowi = i / d;
            const int c = (int)(i - rowi * d);
            const long long b = rowi / T;
            const int t = (int)(rowi - b * T);
            const long long p = row_ptrs[b * T + t];
            out[i] = p ? decode_elem(reinterpret_cast<const unsigned char *>(p), codec, c) : 0.f;
        }
    }
}
// ---- two-tier batched probe (see evs_cache_lookup_batch_c1c2) ----------------------------------------------
// Batched form of the alt-key tier C3 (aprx_embedding.cpp: a map key -> {alt key, recency flag} with second-chance FIFO
// eviction).  The alt key of a key is a pure function of the key (the alt-key tables), so what the tier holds is a SET
// of keys with one flag each.  For the batched path the set is kSetWays-way set-associative: set = hash(key) % nset, one
// 64-byte line of key words (bit 38 = the recency flag), second chance within the set (a newcomer takes an empty way,
// else an unflagged one; when every way is flagged the flags are cleared and it takes the way its hash names).  No
// global FIFO, no tombstones, every operation one line and one CAS -- the exact FIFO order lives in the batch-1
// machine above (and is what tests/golden/aprx_ops.npz pins); the batched forms have no reference counterpart.
struct TwoTierArgs {
    unsigned char *row_tier;   // (B,T): 1 = row in C1's codec, 2 = row in C2's codec, 0 = no row
    unsigned char *tier_out;   // (B,T) user output: 1 = C1 hit, 2 = C2 hit, 3 = alt-key hit (the alt row is served), 0 = miss
    int threshold;             // high_agghit_threshold (evlfu_8.hpp:70)
    unsigned *route_filter; unsigned route_mask, route_stamp;   // see BatchArgs::route_filter (nullptr: not used)
    C3Batch c3;
};

__global__ void __launch_bounds__(256) cache_batch_probe2_kernel(const BatchArgs a1, const BatchArgs a2, const TwoTierArgs tt) {
    __shared__ int s_d1[kMaxBuckets], s_d2[kMaxBuckets];
    __shared__ int s_sum[4];   // C1 hits, C2 hits, perfect requests, alt-key hits
    __shared__ int s_list_n[2];   // list mode (as K1): the misses this block has listed for C1 / for C2
    for (int i = threadIdx.x; i < kMaxBuckets; i += blockDim.x) { s_d1[i] = 0; s_d2[i] = 0; }
    if (threadIdx.x < 4) s_sum[threadIdx.x] = 0;
    if (threadIdx.x < 2) s_list_n[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, half = lane >> 5, hl = lane & 31;
    const int T = a1.T;
    // snapshot: the policy kernels of this batch run later.  (Sampled update: a large cache stops a fraction of a per
    // cent short of its capacity -- the last free entries are not worth hunting -- so "full" has that much slack there.)
    const bool c1_full = a1.bs->count >= a1.cap - (a1.tomb_parity >= 0 && a1.cap > 65536 ? a1.cap / 256 : 0);
    const bool sa = a1.sa.tags != nullptr;   // (both tiers or neither)
    const long long req_stride = (long long)gridDim.x * 8;
    for (long long req = (long long)blockIdx.x * 8 + (threadIdx.x >> 6) * 2 + half; req - half - (threadIdx.x >> 6) * 2 < a1.B;
         req += req_stride) {
        const bool req_on = req < a1.B;
        const bool key_on = req_on && hl < T;
        int row = key_on ? a1.requests[req * T + hl] : -1;
        const bool ok = key_on && row >= 0 && row < a1.backing_rows[hl < T ? hl : 0] && row < a2.backing_rows[hl < T ? hl : 0];
        const unsigned long long key = ((unsigned long long)(hl + 1) << 32) | (unsigned)row;
        unsigned long long end1 = 0, end2 = 0;
        bool ht1 = false, ht2 = false;   // sampled update: is the hinted slot a (re-usable) tombstone
        int e1 = -1, e2 = -1;
        unsigned w1 = 0u, w2 = 0u, tg1 = 0u, tg2 = 0u;
        int y1 = -1, y2 = -1;
        bool c1_room = !c1_full;
        if (sa) {   // set-associative tiers: both tiers' ways of the key's set (a pair: ONE line) in one round trip; "C1 has room" = the key's C1 set has a free way
            const unsigned px = sa_perm(a1.sau, a1.sau.row_base[hl] + (ok ? (unsigned)row : 0u));
            unsigned s1 = 0u, s2 = 0u;
            sa_split(a1.sa, px, s1, tg1);
            sa_split(a2.sa, px, s2, tg2);
            if (!ok) { s1 = 0u; s2 = 0u; }
            SaLine l1, l2;
            if (a1.sa.ways == 8u && a2.sa.ways == 8u) {   // (the reference's 1 : 2 pair: way counts known to the compiler)
                sa_load<8>(a1.sa, s1, l1);
                sa_load<8>(a2.sa, s2, l2);
                y1 = sa_find<8>(a1.sa, l1, tg1, w1); y2 = sa_find<8>(a2.sa, l2, tg2, w2);
                c1_room = sa_has_free<8>(a1.sa, l1);
            } else {
                sa_load<0>(a1.sa, s1, l1);
                sa_load<0>(a2.sa, s2, l2);
                y1 = sa_find<0>(a1.sa, l1, tg1, w1); y2 = sa_find<0>(a2.sa, l2, tg2, w2);
                c1_room = sa_has_free<0>(a1.sa, l1);
            }
            if (ok && y1 >= 0) e1 = (int)sa_entry(a1.sa, s1, (unsigned)y1, w1);
            else if (ok && y2 >= 0) e2 = (int)sa_entry(a2.sa, s2, (unsigned)y2, w2);
            end1 = s1; end2 = s2;
        } else {
            e1 = ok ? probe_ro(a1.slots, a1.mask, key, end1, a1.tomb_parity == 1 ? kTomb : a1.tomb_parity == 0 ? kTomb1 : kTomb, &ht1) : -1;
            if (e1 == kPending) e1 = -1;
            e2 = (ok && e1 < 0) ? probe_ro(a2.slots, a2.mask, key, end2, a2.tomb_parity == 1 ? kTomb : a2.tomb_parity == 0 ? kTomb1 : kTomb, &ht2) : -1;
            if (e2 == kPending) e2 = -1;
        }
        // alt-key probe for a double miss (find_approximate_ev, evlfu_8.cpp:474-490): the key is in C3 and its alt row
        // is resident in C1, else in C2 -> that row is served; the request's agg_hit counts it, nothing is inserted
        int alt_tier = 0, ea = -1;
        if (tt.c3.tags && ok && e1 < 0 && e2 < 0) {
            const long long w3 = c3_find(tt.c3, key);
            if (w3 >= 0) {
                const unsigned alt = tt.c3.alt_tables[hl][row];
                const unsigned at = alt % 100u, ar = alt / 100u;
                if (at >= 1 && at <= (unsigned)T && (long long)ar < a1.backing_rows[at - 1] && (long long)ar < a2.backing_rows[at - 1]) {
                    const unsigned long long akey = ((unsigned long long)at << 32) | ar;
                    unsigned long long es;
                    ea = sa ? sa_lookup(a1.sau, a1.sa, (int)at - 1, ar) : probe_ro(a1.slots, a1.mask, akey, es);
                    if (ea >= 0) alt_tier = 1;
                    else { ea = sa ? sa_lookup(a2.sau, a2.sa, (int)at - 1, ar) : probe_ro(a2.slots, a2.mask, akey, es); if (ea >= 0) alt_tier = 2; }
                    if (alt_tier) atomicOr(&tt.c3.tags[w3], kC3Flag);   // set_recency_flag_c3
                }
            }
        }
        const unsigned long long hm = __ballot(e1 >= 0 || e2 >= 0 || alt_tier != 0);
        const int agg = __popc((unsigned)(half ? (hm >> 32) : hm));
        if (sa) {   // the priority rides in the key word: one atomicMax on it
            if (e1 >= 0 && sa_prio(w1) < agg) {
                const int old = sa_raise(a1.sa, sa_ways_ptr(a1.sa, (unsigned)end1) + y1, w1, agg);
                if (old >= 0) { atomicSub(&s_d1[old], 1); atomicAdd(&s_d1[agg], 1); }
            }
            if (e2 >= 0 && sa_prio(w2) < agg) {
                const int old = sa_raise(a2.sa, sa_ways_ptr(a2.sa, (unsigned)end2) + y2, w2, agg);
                if (old >= 0) { atomicSub(&s_d2[old], 1); atomicAdd(&s_d2[agg], 1); }
            }
        } else {
        if (e1 >= 0 && a1.a.eagg[e1] < agg) {
            const int old = atomicMax(&a1.a.eagg[e1], agg);
            if (old < agg) { atomicSub(&s_d1[old], 1); atomicAdd(&s_d1[agg], 1); }
        }
        if (e2 >= 0 && a2.a.eagg[e2] < agg) {
            const int old = atomicMax(&a2.a.eagg[e2], agg);
            if (old < agg) { atomicSub(&s_d2[old], 1); atomicAdd(&s_d2[agg], 1); }
        }
        }
        // host-memory / file-backed miss tiers: the consumers run BEHIND the policy updates (every missing row crosses the bus
        // once), so what this batch is served out of the arenas -- its hits and the alt rows -- must not be evicted by it
        if (a1.stamp_hits) {
            if (e1 >= 0) a1.estamp[e1] = a1.stamp;
            else if (e2 >= 0) a2.estamp[e2] = a2.stamp;
            else if (alt_tier == 1) a1.estamp[ea] = a1.stamp;
            else if (alt_tier == 2) a2.estamp[ea] = a2.stamp;
        }
        // evlfu_8.cpp:570-601: where a double miss goes
        const bool miss = ok && e1 < 0 && e2 < 0 && alt_tier == 0;
        const int dest = c1_room ? 1 : (agg < tt.threshold ? ((hl & 1) ? 1 : 2) : 2);
        const unsigned char *src = nullptr;
        int codec_of = 0;
        if (e1 >= 0) { src = a1.a.arena + (long long)e1 * a1.row_bytes; codec_of = 1; }
        else if (e2 >= 0) { src = a2.a.arena + (long long)e2 * a2.row_bytes; codec_of = 2; }
        else if (alt_tier == 1) { src = a1.a.arena + (long long)ea * a1.row_bytes; codec_of = 1; }   // the ALT row, at the precision of the tier holding it
        else if (alt_tier == 2) { src = a2.a.arena + (long long)ea * a2.row_bytes; codec_of = 2; }
        else if (miss && dest == 1) { src = a1.backing[hl] + (long long)row * a1.row_bytes; codec_of = 1; }
        else if (miss) { src = a2.backing[hl] + (long long)row * a2.row_bytes; codec_of = 2; }
        if (tt.route_filter && miss && dest == 1 && (hl & 1)) tt.route_filter[mix64(key) & tt.route_mask] = tt.route_stamp;
        if (key_on) {
            const long long m = req * T + hl;
            // (bit 30: hit -- or, of a miss, "the hinted slot is a tombstone", as in K1)
            if (!a1.miss_rec) {
                a1.miss_info[m] = ((miss && dest == 1) ? 0x80000000u : 0u) | ((e1 >= 0 || (miss && dest == 1 && ht1)) ? 0x40000000u : 0u) | ((unsigned)agg << 24) |
                                  (unsigned)(end1 >> a1.hint_shift);
                a2.miss_info[m] = ((miss && dest == 2) ? 0x80000000u : 0u) | ((e2 >= 0 || (miss && dest == 2 && ht2)) ? 0x40000000u : 0u) | ((unsigned)agg << 24) |
                                  (unsigned)(end2 >> a2.hint_shift);
            }
            a1.row_ptrs[m] = (long long)src;
            tt.row_tier[m] = (unsigned char)codec_of;
            tt.tier_out[m] = e1 >= 0 ? 1 : (e2 >= 0 ? 2 : (alt_tier ? 3 : 0));
        }
        if (a1.miss_rec) {   // list mode: the half-wave's misses of each tier, packed, behind the block's earlier ones (as K1)
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const bool is_miss = miss && dest == k + 1;
                const unsigned long long mm = __ballot(is_miss);
                const unsigned mh = (unsigned)(half ? (mm >> 32) : mm);
                int base = 0;
                if (hl == 0 && mh) base = atomicAdd(&s_list_n[k], __popc(mh));
                base = __shfl(base, half * 32, 64);
                if (is_miss) {
                    const BatchArgs &ak = k ? a2 : a1;
                    const int at = base + __popc(mh & ((1u << hl) - 1u));
                    ak.miss_rec[(long long)blockIdx.x * ak.list_cap + at] =
                        make_uint4((unsigned)row, (unsigned)hl | ((unsigned)agg << 8) | ((k ? ht2 : ht1) ? 0x10000u : 0u),
                                   (unsigned)((k ? end2 : end1) >> ak.hint_shift), sa ? (k ? tg2 : tg1) : (unsigned)(req * T + hl));
                }
            }
        }
        const unsigned long long h1 = __ballot(e1 >= 0), h2 = __ballot(e2 >= 0), h3 = __ballot(alt_tier != 0);
        if (req_on && hl == 0) {
            atomicAdd(&s_sum[0], __popc((unsigned)(half ? (h1 >> 32) : h1)));
            atomicAdd(&s_sum[1], __popc((unsigned)(half ? (h2 >> 32) : h2)));
            if (agg == T) atomicAdd(&s_sum[2], 1);
            if (h3) atomicAdd(&s_sum[3], __popc((unsigned)(half ? (h3 >> 32) : h3)));
        }
    }
    __syncthreads();
    if (threadIdx.x < kPartCols) {
        const int i = threadIdx.x;
        const int v1 = i <= T ? s_d1[i] : i == 38 ? s_sum[0] : i == 39 ? s_sum[2] : 0;
        const int v2 = i <= T ? s_d2[i] : i == 38 ? s_sum[1] : 0;
        if (v1) atomicAdd(&a1.part1[(blockIdx.x % kReplicas) * kPartCols + i], v1);
        if (v2) atomicAdd(&a2.part1[(blockIdx.x % kReplicas) * kPartCols + i], v2);
    }
    if (threadIdx.x == 0 && tt.c3.tags && s_sum[3]) atomicAdd(reinterpret_cast<unsigned long long *>(&tt.c3.stat[1]), (unsigned long long)s_sum[3]);
    if (a1.miss_rec && threadIdx.x == 0) { a1.list_cnt[blockIdx.x] = s_list_n[0]; a2.list_cnt[blockIdx.x] = s_list_n[1]; }
}

// rows (B,T,d) fp32 from the pointer table, each row decoded with the codec of the tier that serves it
// (row_tier == NULL: every row in codec1).  d % 4 == 0: one thread per 4-element chunk, raw chunk in (16 / 8 / 4 /
// 2 bytes), float4 out, decode through per-block LDS tables (evs_common.h); other d: element by element.
__global__ void __launch_bounds__(256) cache_rows_from_ptrs2_kernel(const long long *row_ptrs, const unsigned char *row_tier,
                                                                    float *out, long long B, int T, int d, int codec1, int codec2) {
    __shared__ float s_lut16[CodecLut<16>::kEntries], s_lut8[CodecLut<8>::kEntries], s_lut4[CodecLut<4>::kEntries];
    if (codec1 == 16 || codec2 == 16) codec_lut_init<16>(s_lut16);
    if (codec1 == 8 || codec2 == 8) codec_lut_init<8>(s_lut8);
    if (codec1 == 4 || codec2 == 4) codec_lut_init<4>(s_lut4);
    __syncthreads();
    if ((d & 3) == 0) {
        const int cpr = d >> 2;
        const long long n = B * T * cpr;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
            const long long rowi = i / cpr;
            const int c = (int)(i - rowi * cpr);
            const long long p = row_ptrs[rowi];
            const int which = row_tier ? row_tier[rowi] : 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p && which) {
                const int codec = which == 1 ? codec1 : codec2;
                const unsigned char *row = reinterpret_cast<const unsigned char *>(p);
                if (codec == 32) v = reinterpret_cast<const float4 *>(row)[c];
                else if (codec == 16) { const uint2 w = reinterpret_cast<const uint2 *>(row)[c]; v = dec_chunk<16>(w.x, w.y, s_lut16); }
                else if (codec == 8) v = dec_chunk<8>(reinterpret_cast<const unsigned *>(row)[c], 0u, s_lut8);
                else v = dec_chunk<4>(reinterpret_cast<const unsigned short *>(row)[c], 0u, s_lut4);
            }
            reinterpret_cast<float4 *>(out)[i] = v;
        }
        return;
    }
    const long long n = B * T * d;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long rowi = i / d;
        const int c = (int)(i - rowi * d);
        const long long p = row_ptrs[rowi];
        const int which = row_tier ? row_tier[rowi] : 1;
        out[i] = (p && which) ? decode_elem(reinterpret_cast<const unsigned char *>(p), which == 1 ? codec1 : codec2, c) : 0.f;
    }
}

__global__ void iota_kernel(long long *p, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = i;
}

// K2: de-duplicate the misses through the hash (first CAS on an empty slot wins).  One thread per (request,
// table) position; block j lists its unique new keys in new_slot[j*256 ...] and their number in block_cnt[j].
constexpr int kCutDensity = 4;
constexpr long long kCheapWindow = 262144;

// K3, the plan: one block folds K1's partial rows, scans K2's per-block counts into list bases, and thread 0 decides
// how many entries must go, the priority cut and the scan window.  (Folding it into K2 as "the last block to finish
// plans" was built and measured: the returning ticket atomics -- even two-level -- and the serial tail cost more than
// the launch saves: K2 + K3 23.9 us apart, 27 us merged.  Same for the close behind K5: 16.8 us apart, 21 us merged.)
__global__ void __launch_bounds__(256) cache_batch_plan_kernel(const BatchArgs args) {
    __shared__ BatchState sb;
    __shared__ long long s_col[kPartCols];
    __shared__ int s_scan[256];
    __shared__ int s_plan_cnt[kMaxBuckets];
    const int T = args.T;
    {
        const int nw = (int)(sizeof(BatchState) / sizeof(int));
        int *dst = reinterpret_cast<int *>(&sb);
        const int *src = reinterpret_cast<const int *>(args.bs);
        for (int i = threadIdx.x; i < nw; i += blockDim.x) dst[i] = src[i];
    }
    if (threadIdx.x < kPartCols) s_col[threadIdx.x] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < kReplicas * kPartCols; i += blockDim.x) {   // fold and clear K1's replica rows
        const int v = args.part1[i];
        if (v) { atomicAdd(reinterpret_cast<unsigned long long *>(&s_col[i % kPartCols]), (unsigned long long)(long long)v); args.part1[i] = 0; }
    }
    // exclusive scan of block_cnt[0..g2).  The counts come in through LDS with coalesced, independent loads (a thread
    // walking its own run of global entries is `per` DEPENDENT round trips, twice: that was most of this kernel's 11 us);
    // each thread then scans its contiguous run out of LDS.  Tiles of kScanTile counts; `carry` chains the tiles.
    constexpr int kScanTile = 8192;
    __shared__ int s_cnt[kScanTile];
    int carry = 0;
    for (int t0 = 0; t0 < args.g2; t0 += kScanTile) {
        const int nt = args.g2 - t0 < kScanTile ? args.g2 - t0 : kScanTile;
        for (int i = threadIdx.x; i < nt; i += blockDim.x) s_cnt[i] = args.block_cnt[t0 + i];
        __syncthreads();
        const int per = (nt + 255) / 256;
        const int r0 = threadIdx.x * per, r1 = r0 + per < nt ? r0 + per : nt;
        int mine = 0;
        for (int r = r0; r < r1; r++) mine += s_cnt[r];
        s_scan[threadIdx.x] = mine;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const int v = (int)threadIdx.x >= o ? s_scan[threadIdx.x - o] : 0;
            __syncthreads();
            s_scan[threadIdx.x] += v;
            __syncthreads();
        }
        int run = carry + s_scan[threadIdx.x] - mine;
        for (int r = r0; r < r1; r++) { const int c = s_cnt[r]; s_cnt[r] = run; run += c; }
        const int tile_total = s_scan[255];
        __syncthreads();
        for (int i = threadIdx.x; i < nt; i += blockDim.x) args.block_base[t0 + i] = s_cnt[i];
        carry += tile_total;
        __syncthreads();
    }
    if (threadIdx.x == 0) s_scan[255] = carry;   // (read below as the number of new keys)
    __syncthreads();
    BatchState *b = &sb;
    if (threadIdx.x == 0) {
        for (int p = 0; p <= T; p++) b->cnt[p] += (int)s_col[p];
        b->n_hits += s_col[38]; b->n_perfect_hits += s_col[39];
        b->n_tomb -= (int)s_col[36];   // tombstones K2 recycled
        if (b->n_tomb < 0) b->n_tomb = 0;
        b->n_new = s_scan[255];
        // planning copy of the histogram (the real one is updated by the evict / assign kernels) -- in LDS: as a local
        // array it is indexed dynamically, lands in scratch memory, and a kernel with a scratch segment pays for its set-up
        // at every dispatch (this kernel: 12 -> ~6 us)
        int *cnt = s_plan_cnt;
        for (int p = 0; p <= T; p++) cnt[p] = sb.cnt[p];
        b->pstar = -1; b->need = 0; b->ticket = 0; b->flush_t = 0; b->ticket_t = 0; b->pos_ticket = 0; b->win = 0;
        if (cnt[T] >= args.max_perfect && b->n_new > 0) {  // EvLFU flush (EvLFU_C1.py:36-44), once per batch
            b->flush_t = args.flush_n < cnt[T] ? args.flush_n : cnt[T];
            cnt[T] -= b->flush_t;
            b->n_flush++;
        }
        const int n_new = b->n_new < args.cap ? b->n_new : args.cap;
        b->n_assign = n_new;
        int need = n_new - (b->n_free + b->flush_t);
        const int live = b->count - b->flush_t;
        if (need > live) need = live;
        if (need > 0) {
            // The cut: the first priority whose cumulative count covers `need` -- unless the candidates would then be so
            // sparse that the window below becomes a large part of the arena (a 40-60 us scan, measured every other
            // batch at the 10 % Kaggle cache, where the priorities below the mode hold 30-60 k entries against 49 k
            // evictions per batch): then the cut moves up until the candidates are kCutDensity x need (or the window is
            // cheap, or the top priority is reached).  The hand still reaches every low-priority entry within one lap.
            long long acc = 0;
            for (int p = 0; p <= T; p++) {
                acc += cnt[p];
                if (acc < need) continue;
                if (p == T || acc >= (long long)need * kCutDensity || (long long)need * args.cap / acc * 2 <= kCheapWindow) {
                    b->pstar = p;
                    break;
                }
            }
            // candidates (priority <= pstar) are `acc` of the cap entry indices: scan twice the expected span
            long long w = (long long)need * args.cap / (acc > 0 ? acc : 1) * 2 + 16384;
            if (w > args.cap) w = args.cap;
            b->need = need; b->win = (int)w;
            b->n_evict += need;
        }
        if (b->flush_t > 0) b->win = args.cap;  // the flush takes its victims from the whole arena
        b->do_rebuild = (b->n_tomb + (need > 0 ? need : 0) + b->flush_t) > args.nslot / 8;
    }
    __syncthreads();
    {
        const int nw = (int)(sizeof(BatchState) / sizeof(int));
        const int *src = reinterpret_cast<const int *>(&sb);
        int *dst = reinterpret_cast<int *>(args.bs);
        for (int i = threadIdx.x; i < nw; i += blockDim.x) dst[i] = src[i];
    }
}

__global__ void __launch_bounds__(256) cache_batch_insert_kernel(const BatchArgs args) {
    __shared__ int s_tot[8];
    const long long n = args.B * args.T;
    const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned info = m < n ? args.miss_info[m] : 0u;
    if (m < n && args.hit) args.hit[m] = (info >> 30) & ~(info >> 31) & 1u;
    bool is_new = false, recycled = false;
    int slot = -1;
    bool wanted = (info & 0x80000000u) != 0;
    unsigned long long key = 0;
    if (wanted) {
        key = ((unsigned long long)(m % args.T + 1) << 32) | (unsigned)args.requests[m];
        if (args.other_slots) {   // two-tier: the other tier took this key in this very batch
            unsigned long long es;
            if (probe_ro(args.other_slots, args.other_mask, key, es) != -1) wanted = false;
        }
    }
    if (wanted) {
        const int agg = (int)((info >> 24) & 63u);
        // Walk from the key's HOME slot and take the first slot that is empty OR a tombstone.  Recycling tombstones is
        // what keeps the chains short in steady state: an insert that only ever took empty slots left every evicted
        // entry's slot dead until the next rebuild, and after a few thousand batches at capacity the probe kernel ran
        // 3x slower (17 -> 53 us at B = 16 384) although the table was never more than 55 % occupied.  Every copy of a
        // key inserted in this batch walks the same sequence and claims with a CAS, so duplicates still meet: the
        // loser of a race re-reads the slot and finds either its own key (a duplicate: fold the priority) or another
        // key (walk on).  The key is known to be absent from the table (K1 walked to an empty slot without finding it).
        // The walk starts at the first reusable slot K1's probe saw on the chain (every copy of a key carries the same hint:
        // one snapshot), or at the home slot when the hint had to be shortened (tables above 2^24 slots).
        unsigned long long i = args.hint_shift == 0 ? (unsigned long long)(info & 0xffffffu) : (mix64(key) & args.mask);
        const unsigned long long mine = make_word(key, kFieldPend + (unsigned)agg);
        bool placed = false;
        for (unsigned long long steps = 0; steps <= args.mask; steps++) {   // one lap at most: a full table drops the key
            unsigned long long w = args.slots[i];
            if (w == kEmpty || w == kTomb) {
                const unsigned long long prev = atomicCAS(&args.slots[i], w, mine);
                if (prev == w) { is_new = true; slot = (int)i; placed = true; recycled = (w == kTomb); break; }  // this thread owns the new key
                w = prev;
            }
            if ((w & kKeyMask) == key && w != kTomb) { atomicMax(&args.slots[i], mine); placed = true; break; }  // duplicate miss of this batch
            i = (i + 1) & args.mask;
        }
        // file mode: a dropped key of a STAGED table has no address the consumer could read -- the host stages its row
        // for this one position (listed from the end of new_keys; row_ptrs carries -(1 + orphan index) to the patch kernel)
        if (!placed && args.new_keys && ((args.staged_mask >> (int)(m % args.T)) & 1u)) {
            const int oi = atomicAdd(&args.bs->n_orphan, 1);
            args.new_keys[args.stage_rows - 1 - oi] = key;
            args.row_ptrs[m] = -(long long)(1 + oi);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long bm = __ballot(is_new);
    if (lane == 0) s_tot[wave] = __popcll(bm);
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < 4; w++) { if (w < wave) base += s_tot[w]; tot += s_tot[w]; }
    if (is_new) args.new_slot[(long long)blockIdx.x * 256 + base + __popcll(bm & ((1ull << lane) - 1ull))] = slot;
    if (threadIdx.x == 0) args.block_cnt[blockIdx.x] = tot;
    // tombstones taken back into use: the plan subtracts them from the count that decides on a rebuild (replica column 36)
    const unsigned long long rm = __ballot(recycled);
    if (lane == 0 && rm) atomicAdd(&args.part1[(blockIdx.x % kReplicas) * kPartCols + 36], __popcll(rm));
}

// K4: walk the entry window [hand, hand + win) and evict up to `need` entries whose priority is at or
// below the cut (and `flush_t` entries of the top priority when the EvLFU flush fires).  The hand
// moves on afterwards, so inside the low priorities the oldest fills go first -- the clock-hand
// stand-in for the reference's per-bucket FIFO -- and a batch touches ~2 * need / density entries
// instead of the whole arena.  Each thread owns kEvictPerThread consecutive entries so that the
// shared counters (tickets, free-stack top) see one atomic per 2048 entries, not one per entry.
constexpr int kEvictPerThread = 8;
__global__ void __launch_bounds__(256) cache_batch_evict_kernel(const BatchArgs args) {
    __shared__ int s_tot[8];
    __shared__ int s_delta[kMaxBuckets];
    __shared__ int s_tomb;
    BatchState *b = args.bs;
    const int pstar = b->pstar, need = b->need, flush_t = b->flush_t, T = args.T, win = b->win, hand = b->hand;
    const int n_free0 = b->n_free;
    const long long per_pass = (long long)gridDim.x * blockDim.x * kEvictPerThread;
    if (win <= 0 || (long long)blockIdx.x * blockDim.x * kEvictPerThread >= win) return;  // block-uniform
    for (int i = threadIdx.x; i < kMaxBuckets; i += blockDim.x) s_delta[i] = 0;
    if (threadIdx.x == 0) s_tomb = 0;
    __syncthreads();
    int n_tombs = 0;
    const int passes = (int)((win + per_pass - 1) / per_pass);
    for (int it = 0; it < passes; it++) {
        const long long o0 = it * per_pass + ((long long)blockIdx.x * blockDim.x + threadIdx.x) * kEvictPerThread;
        int prio[kEvictPerThread], ent[kEvictPerThread];
        int n_t = 0, n_c = 0;
#pragma unroll
        for (int j = 0; j < kEvictPerThread; j++) {
            const long long o = o0 + j;
            long long e = hand + o;
            if (e >= args.cap) e -= args.cap;
            ent[j] = (int)e;
            prio[j] = (o < win && args.a.ekey[e] != kEmpty && !(args.estamp && args.estamp[e] == args.stamp)) ? args.a.eagg[e] : -1;
            n_t += (prio[j] == T && flush_t > 0);
        }
        // no flush (the usual batch): a victim's ticket is also its place on the free stack, n_free + [0, need).
        // Flush batches: the planned number of flush victims may not be reached (hits of the running batch are pinned
        // with a host-memory miss tier), so both kinds of victims take their places from one more counter -- a gap
        // in the stack would hand out stale, possibly live, entries (found by tools/fuzz_cache.py).
        unsigned victim = 0, fvict = 0;
        int tk_t = 0;
        if (flush_t > 0) {  // uniform
            tk_t = block_reserve_n(&b->ticket_t, n_t, s_tot);
            int k = tk_t;
#pragma unroll
            for (int j = 0; j < kEvictPerThread; j++)
                if (prio[j] == T) { if (k++ < flush_t) fvict |= 1u << j; }
        }
#pragma unroll
        for (int j = 0; j < kEvictPerThread; j++)
            n_c += (prio[j] >= 0 && !((fvict >> j) & 1) && prio[j] <= pstar);
        int tk_c = block_reserve_n(&b->ticket, n_c, s_tot);
#pragma unroll
        for (int j = 0; j < kEvictPerThread; j++) {
            if ((fvict >> j) & 1) victim |= 1u << j;
            else if (prio[j] >= 0 && prio[j] <= pstar) { if (tk_c < need) victim |= 1u << j; tk_c++; }
        }
        int pos = n_free0 + (tk_c - n_c);                 // no flush: victims of this thread are a prefix of its tickets
        if (flush_t > 0) pos = n_free0 + block_reserve_n(&b->pos_ticket, __popc(victim), s_tot);   // uniform
        // (Giving a slot whose successor is empty back as EMPTY instead of a tombstone halves the tombstones and the
        // rebuilds -- ~3 us per batch amortised -- but the successor read is one more dependent round trip for every
        // victim: +6 us in this kernel.  Measured, not kept.)  Two passes so that the 8 victims' loads are independent.
        int sl[kEvictPerThread];
#pragma unroll
        for (int j = 0; j < kEvictPerThread; j++) sl[j] = ((victim >> j) & 1) ? args.eslot[ent[j]] : 0;
#pragma unroll
        for (int j = 0; j < kEvictPerThread; j++) {
            if (!((victim >> j) & 1)) continue;
            const int e = ent[j];
            args.slots[sl[j]] = kTomb;
            n_tombs++;
            if (args.evicted_keys) args.evicted_keys[pos] = (args.a.ekey[e] & kKeyMask) | (((fvict >> j) & 1) ? (1ull << 63) : 0ull);
            args.a.ekey[e] = kEmpty;
            atomicSub(&s_delta[prio[j]], 1);
            args.a.free_stack[pos++] = e;
        }
    }
    if (n_tombs) atomicAdd(&s_tomb, n_tombs);
    __syncthreads();
    int gone = 0;
    for (int i = threadIdx.x; i <= T; i += blockDim.x)
        if (s_delta[i]) { atomicAdd(&b->cnt[i], s_delta[i]); gone -= s_delta[i]; }
    if (gone) atomicSub(&b->count, gone);
    if (threadIdx.x == 0 && s_tomb) atomicAdd(&b->n_tomb, s_tomb);
}

// K6, end-of-batch bookkeeping (one block): folds K5's partial rows, then thread 0 closes the batch.  The state comes
// in and goes out with parallel loads / stores; thread 0 works on the LDS copy (walking the fields in global memory
// was ~10 dependent round trips).
__global__ void __launch_bounds__(256) cache_batch_close_kernel(const BatchArgs args) {
    __shared__ int s_col[kPartCols];
    __shared__ BatchState sb;
    const int nw = (int)(sizeof(BatchState) / sizeof(int));
    {
        int *dst = reinterpret_cast<int *>(&sb);
        const int *src = reinterpret_cast<const int *>(args.bs);
        for (int i = threadIdx.x; i < nw; i += blockDim.x) dst[i] = src[i];
    }
    if (threadIdx.x < kPartCols) s_col[threadIdx.x] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < kReplicas * kPartCols; i += blockDim.x) {   // fold and clear K5's replica rows
        const int v = args.part2[i];
        if (v) { atomicAdd(&s_col[i % kPartCols], v); args.part2[i] = 0; }
    }
    __syncthreads();
    BatchState *b = &sb;
    if ((int)threadIdx.x <= args.T && s_col[threadIdx.x]) b->cnt[threadIdx.x] += s_col[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        b->n_free += b->flush_t > 0 ? b->pos_ticket : (b->ticket < b->need ? b->ticket : b->need);
        const int take = b->n_assign < b->n_free ? b->n_assign : b->n_free;
        b->n_free -= take; b->count += take;
        b->n_tomb += s_col[37];
        if (args.rebuild == 1) b->n_tomb = 0;
        if (b->need > 0 && b->win > 0) { long long h = (long long)b->hand + b->win; b->hand = (int)(h % args.cap); }
        b->n_miss = 0; b->n_new = 0; b->n_orphan = 0; b->batch_id++;
        b->n_requests += args.B;
        args.host_tomb[0] = b->n_tomb;
        args.host_tomb[2] = (int)b->batch_id;   // the ordinal of this close: the host ignores reports older than its last sweep / rebuild
    }
    __syncthreads();
    {
        const int *src = reinterpret_cast<const int *>(&sb);
        int *dst = reinterpret_cast<int *>(args.bs);
        for (int i = threadIdx.x; i < nw; i += blockDim.x) dst[i] = src[i];
    }
}

// Rebuild the hash without tombstones when they pile up (the host reads the count the close left in mapped memory):
// clear, then every entry back into its chain.
__global__ void __launch_bounds__(256) cache_batch_clear_kernel(const BatchArgs args) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < args.nslot; i += (long long)gridDim.x * blockDim.x)
        args.slots[i] = kEmpty;
}
__global__ void __launch_bounds__(256) cache_batch_reinsert_kernel(const BatchArgs args) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < args.cap; e += gridDim.x * blockDim.x) {
        const unsigned long long key = args.a.ekey[e] & kKeyMask;   // (the sampled update keeps a batch stamp above the key)
        if (key == kEmpty) continue;
        unsigned long long i = mix64(key) & args.mask;
        const unsigned long long w = make_word(key, (unsigned)e);
        for (unsigned long long steps = 0; steps <= args.mask; steps++) {   // entries <= cap < nslot/2: always ends early
            if (args.slots[i] == kEmpty && atomicCAS(&args.slots[i], kEmpty, w) == kEmpty) break;
            i = (i + 1) & args.mask;
        }
        args.eslot[e] = (int)i;
    }
}

// K5: give every unique new key an entry and fill its arena row from the backing store: block j serves the
// keys K2's block j listed, 16 lanes per key (lane 0 does the bookkeeping, all of them move 16-byte pieces
// of the row).  New key i of the batch (list base + rank) takes free_stack[n_free - 1 - i] -- no atomics.
__global__ void __launch_bounds__(256) cache_batch_assign_kernel(const BatchArgs args) {
    __shared__ int s_delta[kMaxBuckets];
    __shared__ int s_drop;
    BatchState *b = args.bs;
    for (int i = threadIdx.x; i < kMaxBuckets; i += blockDim.x) s_delta[i] = 0;
    if (threadIdx.x == 0) s_drop = 0;
    __syncthreads();
    // the free stack as K4 left it: what was free plus the victims it handed out
    const int n_free = b->n_free + (b->flush_t > 0 ? b->pos_ticket : (b->ticket < b->need ? b->ticket : b->need));
    const int n_assign = b->n_assign < n_free ? b->n_assign : n_free;
    const int sub = threadIdx.x & 15;
    const int my_cnt = args.block_cnt[blockIdx.x], my_base = args.block_base[blockIdx.x];
    for (int local = threadIdx.x >> 4; local < my_cnt; local += 16) {
        const int i = my_base + local;
        const int slot = args.new_slot[(long long)blockIdx.x * 256 + local];
        if (i >= n_assign) {  // no room: forget the key
            // (file mode: the key stays pending until the consumers have read its staged row -- cache_batch_unstage_kernel)
            if (sub == 0 && !args.staging) { args.slots[slot] = kTomb; atomicAdd(&s_drop, 1); }
            continue;
        }
        const int e = args.a.free_stack[n_free - 1 - i];
        const unsigned long long w = args.slots[slot];
        const unsigned long long key = w & kKeyMask;
        // every lane of the group must have read the pending word before lane 0 replaces it
        __builtin_amdgcn_wave_barrier();
        if (sub == 0) {
            const int agg = (int)((unsigned)(w >> kKeyBits) - kFieldPend);
            args.a.ekey[e] = key; args.a.eagg[e] = agg; args.eslot[e] = slot;
            args.slots[slot] = make_word(key, (unsigned)e);
            atomicAdd(&s_delta[agg], 1);
        }
        const int t = (int)(key >> 32) - 1;
        const unsigned char *srow = ((args.staged_mask >> t) & 1u) ? args.staging + (long long)i * args.row_bytes
                                                                 : args.backing[t] + (long long)(unsigned)(key & 0xffffffffull) * args.row_bytes;
        unsigned char *drow = args.a.arena + (long long)e * args.row_bytes;
        if ((args.row_bytes & 15) == 0) {
            for (int c = sub * 16; c < args.row_bytes; c += 256) *reinterpret_cast<float4 *>(drow + c) = *reinterpret_cast<const float4 *>(srow + c);
        } else {
            for (int c = sub; c < args.row_bytes; c += 16) drow[c] = srow[c];
        }
    }
    __syncthreads();
    if (threadIdx.x < kPartCols) {
        const int i = threadIdx.x;
        const int v = i <= args.T ? s_delta[i] : i == 37 ? s_drop : 0;
        if (v) atomicAdd(&args.part2[(blockIdx.x % kReplicas) * kPartCols + i], v);
    }
}

// ---- the sampled policy update: K2..K5 as ONE kernel --------------------------------------------------------------
// The plan-based update above is five dependent launches whose work is tiny and whose time is round trips (insert ->
// plan -> evict -> assign -> close: ~50 of the batch's ~100 us at B = 16 384).  This form removes every dependency
// BETWEEN new keys instead: the thread that wins the hash slot of a new key also finds that key's victim by itself --
// it samples one aligned group of kSampleGroup entries (chosen by a hash of the key and the batch number), takes a
// free entry if the group has one, else the entry of lowest priority, claims it with a CAS on its key word, unhooks
// the old key from the hash, and hands the entry to the new key (only then does it claim the key's hash slot: see the
// kernel).  No plan, no tickets, no free stack, no scan: the
// eviction decision is "lowest priority of 8 sampled entries" (the sampled-LFU of Redis) instead of "lowest priority
// in a clock-hand window"; tests/test_gpu_cache.py holds both against the sequential oracle's hit rate.
//   Entries inserted or hit in this batch carry the batch stamp and are never victims.  Inserts and evictions now
// run side by side, so a slot freed by an eviction must not become claimable in the launch that freed it: a copy of
// a key that arrives late would take it although another copy already sits further down the chain (two entries for
// one key -- tools/fuzz_cache.py found it).  Tombstones therefore come in two values, by batch parity; a batch writes
// its own value and re-uses only the other one (K1's hint follows the same rule).  What threads tell each other inside the launch travels as agent-scope atomics only (the 8 XCDs'
// L2s are not coherent with each other): the key word of an entry (free / claimed / key), its stamp, its priority,
// the hash words.  A key whose group and every following group hold nothing claimable is dropped (tiny caches).
constexpr unsigned long long kClaimed = ~0ull;   // ekey of an entry a thread owns while it moves it
constexpr int kSampleGroup = 8;
constexpr int kFreeHunt = 64;     // groups a new key looks at for a FREE entry while the cache is not full
constexpr unsigned kStampMask = (1u << (64 - kKeyBits)) - 1u;   // the batch stamp rides in the 26 bits above the key
// Words that threads of ONE launch hand to each other are written and read with read-modify-write atomics only (those
// are performed where all 8 XCDs meet; plain and sc1 accesses may be served by an XCD's own L2).
__device__ __forceinline__ unsigned long long ld_agent(unsigned long long *p) { return atomicOr(p, 0ull); }
__device__ __forceinline__ void st_agent(int *p, int v) { atomicExch(p, v); }
__device__ __forceinline__ void st_agent(unsigned long long *p, unsigned long long v) { atomicExch(p, v); }

// One sampled group: key words, priorities and hash slots of kSampleGroup consecutive entries (plain loads, all
// independent: one round trip).  Entries that may not be taken read as kClaimed: those filled in this batch (their
// key word carries this batch's stamp -- so no second look is needed after the CAS: a CAS that expects an older word
// fails on them), and, with a host-memory miss tier, those hit in it (stamped by K1, an earlier launch).
struct SampleWindow {
    unsigned long long k[kSampleGroup];
    int pr[kSampleGroup], sl[kSampleGroup];
};
__device__ __forceinline__ void sampled_load(const BatchArgs &args, long long e0, SampleWindow &w) {
    const unsigned stamp26 = (unsigned)args.stamp & kStampMask;
#pragma unroll
    for (int j = 0; j < kSampleGroup; j++) {
        const bool on = e0 + j < args.cap;
        w.k[j] = on ? args.a.ekey[e0 + j] : kClaimed;
        w.pr[j] = on ? args.a.eagg[e0 + j] : 0;
        w.sl[j] = on ? args.eslot[e0 + j] : 0;
        const bool hit_now = on && args.stamp_hits && args.estamp[e0 + j] == args.stamp;
        // a word with this batch's stamp: filled in this launch -- or taken and given back in it (key part 0): its
        // row may still be on its way out of another XCD's L2, so it is not handed out again before the next batch
        if (w.k[j] != kEmpty && ((unsigned)(w.k[j] >> kKeyBits) == stamp26 || hit_now)) w.k[j] = kClaimed;
    }
}
// Take the best entry of the window: a free one, else the lowest priority.  Returns the entry (its key word is kClaimed,
// it belongs to the caller; if it held a key, that key's hash word is already a tombstone) or -1.
// One exit, flags instead of continue / break / return out of the nested loops: the early-exit form of this function
// came back from hipcc 7.2 returning a garbage entry index on the evict path (found with guards on the GPU).
__device__ __forceinline__ int sampled_take(const BatchArgs &args, long long e0, const SampleWindow &w, int &old_prio,
                                            unsigned long long &old_key, bool free_only = false) {
    unsigned tried = 0;
    bool more = true;
    int result = -1;
#pragma unroll 1
    for (int r = 0; r < kSampleGroup && more && result < 0; r++) {
        int best = -1, bp = 0x7fffffff, bs = 0;
        unsigned long long bk = 0;
#pragma unroll
        for (int j = 0; j < kSampleGroup; j++) {
            const int pj = (w.k[j] & kKeyMask) == kEmpty ? -1 : w.pr[j];
            const bool cand = !((tried >> j) & 1u) && w.k[j] != kClaimed && pj < bp && (!free_only || pj < 0);
            best = cand ? j : best; bp = cand ? pj : bp; bk = cand ? w.k[j] : bk; bs = cand ? w.sl[j] : bs;
        }
        more = best >= 0;
        if (more) {
            tried |= 1u << best;
            const long long e = e0 + best;
            if (atomicCAS(&args.a.ekey[e], bk, kClaimed) == bk) {   // else somebody else was faster
                if ((bk & kKeyMask) != kEmpty) args.slots[bs] = args.tomb_parity ? kTomb1 : kTomb;   // (plain: a walker that still sees the old key walks on, as it would have)
                old_prio = bp;
                old_key = bk & kKeyMask;
                result = (int)e;
            }
        }
    }
    return result;
}

// One thread copies one row.  All loads first: a load / store pair per 16 bytes is one dependent round trip each (the
// compiler cannot prove the two rows apart) and cost 11 us of the update kernel; fixed piece counts keep the pieces in
// registers (a runtime-indexed array, a switch over the sizes inside one kernel, a float4[N] with constant indices and a
// struct of named float4 members all went to scratch or were promoted to LDS under hipcc 7.2: 32 -> 43-52 us), so the
// update kernel is compiled per row size and the pieces are plain local variables.
// replica columns of part2 in this mode: 0..32 histogram deltas, 33 count delta (entries taken from the free ones minus
// victims given back), 34 evictions, 35 tombstone delta (evicted - recycled)
//
// Order inside a thread: the hash slot first (a pending word: the copy of a key that wins it is the one that inserts the
// key, with its request's agg_hit as the priority; the others stop there), then the entry, then nothing but stores --
// three dependent round trips: the miss record, {slot CAS, victim group, source row}, the CAS on the entry.
// one missed key of the batch: victim, row, key word, hash slot (see the comment above); m = its position in the batch.
// PIECES pieces of type U make one row (0: any row size, byte by byte)
// the alt-key set's fill: one evicted key becomes a member (second chance within its set)
struct C3Set { unsigned long long *tags; long long nset; long long *stat; };
__device__ __forceinline__ void c3_insert_key(const C3Set &c3, unsigned long long key) {
    const long long base = c3_base(key, c3.nset);
    bool done = false;
    for (int attempt = 0; attempt < 4 && !done; attempt++) {   // (a CAS lost to another newcomer of the same set: look again)
        unsigned long long w[kSetWays];
        int present = 0, empty = -1, plain = -1;
#pragma unroll
        for (int j = 0; j < kSetWays; j++) {
            // (agent-scope loads: coherent across the XCDs' L2s like the returning atomicOr(.., 0) this used to be, but eight
            //  independent loads of one line instead of eight read-modify-writes at the memory side)
            w[j] = __hip_atomic_load(&c3.tags[base + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            present |= (w[j] & kKeyMask) == key;
            if (w[j] == 0ull && empty < 0) empty = j;
            if (w[j] != 0ull && !(w[j] & kC3Flag) && plain < 0) plain = j;
        }
        if (present) done = true;
        else {
            int way = empty >= 0 ? empty : plain;
            if (way < 0) {   // every way has had its second chance now
#pragma unroll
                for (int j = 0; j < kSetWays; j++) atomicAnd(&c3.tags[base + j], ~kC3Flag);
                way = (int)((key >> 3) % kSetWays);
            }
            unsigned long long expect = 0ull;
#pragma unroll
            for (int j = 0; j < kSetWays; j++) expect = j == way ? w[j] : expect;
            if (empty < 0 && plain < 0) expect &= ~kC3Flag;
            if (atomicCAS(&c3.tags[base + way], expect, key) == expect) {
                if (expect == 0ull) atomicAdd(reinterpret_cast<unsigned long long *>(&c3.stat[0]), 1ull);
                done = true;
            }
        }
    }
}

// (TAIL: one more piece of another type behind the PIECES pieces -- a 36-byte row is 2 x 16 + 4 bytes, an 18-byte row 16 + 2:
//  3 or 2 memory instructions each way instead of 9; global loads and stores of 16 bytes need no more than the row's own
//  4- or 2-byte alignment on this part)
struct NoTail {};
template <int PIECES, typename U, typename TAIL = NoTail>
__device__ __forceinline__ void sampled_insert_one(const BatchArgs &args, int t, unsigned row, int agg, unsigned hint_slot, bool hint_tomb,
                                                   long long m, int *s_delta, int *s_stat, unsigned long long *s_vict = nullptr,
                                                   int *s_nvict = nullptr) {
    const unsigned long long key = ((unsigned long long)(t + 1) << 32) | row;
    // two tiers: the other tier took this key in this very batch.  (Only an odd table index can be routed both ways by two
    // requests of one batch -- probe2's rule sends an even one to C2 whatever its request's agg_hit once C1 is full, and
    // nothing to C2 before that -- so only those keys pay for the look into the other tier's hash.)
    if (args.route_filter && (t & 1) && args.route_filter[mix64(key) & args.route_mask] == args.route_stamp) return;
    if (args.other_slots && (t & 1)) {
        unsigned long long es;
        if (probe_ro(args.other_slots, args.other_mask, key, es) != -1) return;
    }
    const unsigned long long my_tomb = args.tomb_parity ? kTomb1 : kTomb, old_tomb = args.tomb_parity ? kTomb : kTomb1;
    // The victim group depends on the key, the batch and the POSITION in the batch: copies of one missing key must
    // not all start on the same 8 entries.  Its loads, and the source row, go out together with the slot CAS.
    const long long n_groups = ((long long)args.cap + kSampleGroup - 1) / kSampleGroup;
    long long g = (long long)(mix64(key ^ ((unsigned long long)(unsigned)args.stamp * 0x9e3779b97f4a7c15ull) ^ ((unsigned long long)m << 40)) % (unsigned long long)n_groups);
    SampleWindow win;
    sampled_load(args, g * kSampleGroup, win);
    const unsigned char *srow = args.backing[t] + (long long)row * args.row_bytes;
    U r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11, r12, r13, r14, r15;
    TAIL rt;
    // Tables in HBM: the source row is requested HERE, together with the slot CAS and the victim group (one round trip for
    // the three).  Host-memory tables (stamp_hits): only the copy of a key that wins the slot AND finds an entry reads the
    // row -- each missing row crosses the bus once per batch, whatever the number of requests that miss it.
    const bool late_row = args.stamp_hits != 0;
#define EVS_SAMPLED_LOAD_ROW()                                                                                     \
    do {                                                                                                           \
        if constexpr (PIECES > 0) r0 = reinterpret_cast<const U *>(srow)[0];                                       \
        if constexpr (PIECES > 1) r1 = reinterpret_cast<const U *>(srow)[1];                                       \
        if constexpr (PIECES > 2) r2 = reinterpret_cast<const U *>(srow)[2];                                       \
        if constexpr (PIECES > 3) r3 = reinterpret_cast<const U *>(srow)[3];                                       \
        if constexpr (PIECES > 4) r4 = reinterpret_cast<const U *>(srow)[4];                                       \
        if constexpr (PIECES > 5) r5 = reinterpret_cast<const U *>(srow)[5];                                       \
        if constexpr (PIECES > 6) r6 = reinterpret_cast<const U *>(srow)[6];                                       \
        if constexpr (PIECES > 7) r7 = reinterpret_cast<const U *>(srow)[7];                                       \
        if constexpr (PIECES > 8) r8 = reinterpret_cast<const U *>(srow)[8];                                       \
        if constexpr (PIECES > 9) r9 = reinterpret_cast<const U *>(srow)[9];                                       \
        if constexpr (PIECES > 10) r10 = reinterpret_cast<const U *>(srow)[10];                                    \
        if constexpr (PIECES > 11) r11 = reinterpret_cast<const U *>(srow)[11];                                    \
        if constexpr (PIECES > 12) r12 = reinterpret_cast<const U *>(srow)[12];                                    \
        if constexpr (PIECES > 13) r13 = reinterpret_cast<const U *>(srow)[13];                                    \
        if constexpr (PIECES > 14) r14 = reinterpret_cast<const U *>(srow)[14];                                    \
        if constexpr (PIECES > 15) r15 = reinterpret_cast<const U *>(srow)[15];                                    \
        if constexpr (!std::is_same<TAIL, NoTail>::value) rt = *reinterpret_cast<const TAIL *>(srow + PIECES * sizeof(U)); \
    } while (0)
    if (!late_row) EVS_SAMPLED_LOAD_ROW();
    // 1. The hash slot, with a PENDING word: that is what de-duplicates the copies of a key (the loser of the CAS sees
    //    the key and stops -- it takes no entry; an earlier form claimed the entry first and gave it back, which let a
    //    batch with hundreds of copies of its hot missing keys evict half of a small cache for nothing).  The walk starts
    //    at the first reusable slot K1's probe saw on the chain (every copy of a key carries the same hint: one
    //    snapshot), or at the home slot when the hint had to be shortened (tables above 2^24 slots); K1 also said
    //    whether that slot was empty or a tombstone, so the first access is the CAS itself.
    const unsigned long long pend = make_word(key, kFieldPend + (unsigned)agg);
    unsigned long long i = args.hint_shift == 0 ? (unsigned long long)hint_slot : (mix64(key) & args.mask);
    unsigned long long w = (args.hint_shift == 0 && hint_tomb) ? old_tomb : kEmpty;
    int placed = 0;   // 1: the slot is this thread's, 2: another copy of the key was faster
    bool recycled = false;
    for (unsigned long long steps = 0; steps <= args.mask && !placed; steps++) {   // one lap at most: a full table drops the key
        if (w == kEmpty || w == old_tomb) {
            const unsigned long long prev = atomicCAS(&args.slots[i], w, pend);
            if (prev == w) { placed = 1; recycled = (w == old_tomb); }
            else w = prev;
        }
        if (!placed) {
            if ((w & kKeyMask) == key) placed = 2;
            else { i = (i + 1) & args.mask; w = args.slots[i]; }   // somebody else's key, or not for re-use: next slot (a plain read: the CAS is the arbiter)
        }
    }
    if (placed != 1) return;
    if (recycled) atomicSub(&s_stat[2], 1);
    // 2. The entry.  While the cache still has free entries (count of the last close) a key whose group has none looks
    //    at a few more groups for one before it evicts anything: a small cache fills to exactly its capacity that way
    //    (the two-tier routing asks "is C1 full"); a large one stops hunting once less than 1/256 of it is free.
    int old_prio = -1, e = -1;
    unsigned long long old_key = 0;
    const int n_free = args.cap - args.bs->count;
    if (n_free > 0 && (args.cap <= 65536 || n_free > args.cap / 256)) {
        e = sampled_take(args, g * kSampleGroup, win, old_prio, old_key, true);
        long long gf = g;
        for (int hunt = 1; hunt < kFreeHunt && e < 0 && hunt < n_groups; hunt++) {
            gf = gf + 1 == n_groups ? 0 : gf + 1;
            SampleWindow wf;
            sampled_load(args, gf * kSampleGroup, wf);
            e = sampled_take(args, gf * kSampleGroup, wf, old_prio, old_key, true);
        }
    }
    for (long long tries = 0; tries < n_groups && e < 0; tries++) {
        if (tries) sampled_load(args, g * kSampleGroup, win);
        e = sampled_take(args, g * kSampleGroup, win, old_prio, old_key);
        g = g + 1 == n_groups ? 0 : g + 1;
    }
    if (e < 0) {   // nothing claimable anywhere (a cache smaller than the batch's keys): the key is not kept
        args.slots[i] = my_tomb;
        atomicAdd(&s_stat[2], 1);
        return;
    }
    if (old_prio >= 0) {
        atomicSub(&s_delta[old_prio], 1); atomicAdd(&s_stat[1], 1); atomicAdd(&s_stat[2], 1);
        if (s_vict) s_vict[atomicAdd(s_nvict, 1)] = old_key;   // for the alt-key tier
    } else atomicAdd(&s_stat[0], 1);
    const bool c3_inline = old_prio >= 0 && args.c3_tags != nullptr;
    if (late_row) EVS_SAMPLED_LOAD_ROW();
#undef EVS_SAMPLED_LOAD_ROW
    // 3. Everything else is stores nobody waits for: the row, the priority, the slot index, the key word (with this
    //    batch's stamp: nobody takes the entry away again), the final hash word.
    unsigned char *drow = args.a.arena + (long long)e * args.row_bytes;
    if constexpr (PIECES > 0) {
        if constexpr (PIECES > 0) reinterpret_cast<U *>(drow)[0] = r0;
        if constexpr (PIECES > 1) reinterpret_cast<U *>(drow)[1] = r1;
        if constexpr (PIECES > 2) reinterpret_cast<U *>(drow)[2] = r2;
        if constexpr (PIECES > 3) reinterpret_cast<U *>(drow)[3] = r3;
        if constexpr (PIECES > 4) reinterpret_cast<U *>(drow)[4] = r4;
        if constexpr (PIECES > 5) reinterpret_cast<U *>(drow)[5] = r5;
        if constexpr (PIECES > 6) reinterpret_cast<U *>(drow)[6] = r6;
        if constexpr (PIECES > 7) reinterpret_cast<U *>(drow)[7] = r7;
        if constexpr (PIECES > 8) reinterpret_cast<U *>(drow)[8] = r8;
        if constexpr (PIECES > 9) reinterpret_cast<U *>(drow)[9] = r9;
        if constexpr (PIECES > 10) reinterpret_cast<U *>(drow)[10] = r10;
        if constexpr (PIECES > 11) reinterpret_cast<U *>(drow)[11] = r11;
        if constexpr (PIECES > 12) reinterpret_cast<U *>(drow)[12] = r12;
        if constexpr (PIECES > 13) reinterpret_cast<U *>(drow)[13] = r13;
        if constexpr (PIECES > 14) reinterpret_cast<U *>(drow)[14] = r14;
        if constexpr (PIECES > 15) reinterpret_cast<U *>(drow)[15] = r15;
        if constexpr (!std::is_same<TAIL, NoTail>::value) *reinterpret_cast<TAIL *>(drow + PIECES * sizeof(U)) = rt;
    } else if ((args.row_bytes & 15) == 0) { for (int c = 0; c < args.row_bytes; c += 16) *reinterpret_cast<float4 *>(drow + c) = *reinterpret_cast<const float4 *>(srow + c); }
    else { for (int c = 0; c < args.row_bytes; c++) drow[c] = srow[c]; }
    args.a.eagg[e] = agg;
    args.eslot[e] = (int)i;
    // (plain stores: whoever looks at these words during the launch either sees the old value -- a claimed entry, the
    //  pending word of this key -- or the new one, and both mean "not yours"; 255 k returning-or-not atomics per batch
    //  at the memory side were a third of this kernel)
    args.a.ekey[e] = key | ((unsigned long long)((unsigned)args.stamp & kStampMask) << kKeyBits);
    args.slots[i] = make_word(key, (unsigned)e);
    atomicAdd(&s_delta[agg], 1);
    if (c3_inline) c3_insert_key(C3Set{args.c3_tags, args.c3_nset, args.c3_stat}, old_key & kKeyMask);   // what this thread evicted -> the alt-key set
}

// the piece types of a row as compiler vector types (loads through an address-space-1 pointer want trivially copyable
// non-class types; HIP's float4 / uint2 are classes)
template <typename U> struct SaNative { using type = U; };
template <> struct SaNative<float4> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct SaNative<uint2> { typedef unsigned type __attribute__((ext_vector_type(2))); };
template <> struct SaNative<NoTail> { using type = int; };
// Set-associative policy (evs_hash.h, batch policy 2): one missed key of the batch.  The record carries the key's set; its
// line (kSaWays key words: key | batch stamp | priority) and the source row go out together, then ONE CAS on the way of the
// lowest priority (free ways first, lowest way index among equals; ways filled in this batch excepted), then stores
// nobody waits for.  Two dependent round trips behind the record.
// Copies of one missing key: every copy looks at the same words and ranks them the same way, so two copies that both
// believe the key is absent choose the same way and the CAS lets one through; the loser's CAS returns the word that beat
// it -- the key itself (fold the priority, done) or another key of this batch (that way is out, next best).  The line
// is read plainly: what a plain read can return is a word as it stood at the start of the launch or one written in it,
// and a way changes at most once per launch (old -> stamped), which is all the argument needs (docs/HISTORY.md 3.4).
template <int PIECES, typename U, typename TAIL = NoTail, int W = 0>
__device__ __forceinline__ void sa_insert_one(const BatchArgs &args, const unsigned char *table, int t, unsigned row, int agg, unsigned set, unsigned tag1, int *s_delta, int *s_stat) {
    constexpr int NW = W > 0 ? W : kSaMaxWays;   // ways the scans below walk (W = 0: any geometry, masked by args.sa.ways)
    const unsigned long long key = ((unsigned long long)(t + 1) << 32) | row;
    // two tiers: the other tier took this key in this very batch (only an odd table index can be routed both ways by two
    // requests of one batch: see sampled_insert_one)
    if (args.route_filter && (t & 1) && args.route_filter[mix64(key) & args.route_mask] == args.route_stamp) return;
    unsigned *tags = sa_ways_ptr(args.sa, set);
    // The set's line FIRST, then the source row, in that order in the instruction stream: vector-memory loads return in
    // order, so the CAS can go out when the line is there (an Infinity-Cache hit: the probe read it 20 us ago) while the
    // row -- a random line of a multi-GB table -- is still on its way.  The table base comes out of LDS: as a generic
    // pointer its loads would be flat_load (they count on lgkmcnt too and the compiler drains everything in front of
    // the CAS); through an address-space-1 pointer they are global_load.
    SaLine line;
    sa_load<W>(args.sa, set, line);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char *srow = table + (long long)row * args.row_bytes;
    using NU = typename SaNative<U>::type;
    using NT = typename SaNative<TAIL>::type;
    typedef const __attribute__((address_space(1))) NU *gsrc_t;
    const gsrc_t gs = reinterpret_cast<gsrc_t>(reinterpret_cast<uintptr_t>(srow));
    NU r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11, r12, r13, r14, r15;
    if constexpr (PIECES > 0) r0 = gs[0];
    if constexpr (PIECES > 1) r1 = gs[1];
    if constexpr (PIECES > 2) r2 = gs[2];
    if constexpr (PIECES > 3) r3 = gs[3];
    if constexpr (PIECES > 4) r4 = gs[4];
    if constexpr (PIECES > 5) r5 = gs[5];
    if constexpr (PIECES > 6) r6 = gs[6];
    if constexpr (PIECES > 7) r7 = gs[7];
    if constexpr (PIECES > 8) r8 = gs[8];
    if constexpr (PIECES > 9) r9 = gs[9];
    if constexpr (PIECES > 10) r10 = gs[10];
    if constexpr (PIECES > 11) r11 = gs[11];
    if constexpr (PIECES > 12) r12 = gs[12];
    if constexpr (PIECES > 13) r13 = gs[13];
    if constexpr (PIECES > 14) r14 = gs[14];
    if constexpr (PIECES > 15) r15 = gs[15];
    NT rt;
    if constexpr (!std::is_same<TAIL, NoTail>::value) {
        typedef const __attribute__((address_space(1))) NT *gtail_t;
        rt = *reinterpret_cast<gtail_t>(reinterpret_cast<uintptr_t>(srow + PIECES * sizeof(U)));
    }
    __builtin_amdgcn_sched_barrier(0);
    const SaGeom &g = args.sa;
    const unsigned cur = sa_cur_stamp(g, args.stamp);
    unsigned w[NW];
#pragma unroll
    for (int j = 0; j < NW; j++) w[j] = sa_way_word(line, j);
    int way = -1, old_prio = -1;
    unsigned old_word = 0u;
    bool done = false;
#pragma unroll 1
    for (int attempt = 0; attempt <= NW && !done; attempt++) {
        int best = -1, bp = 0x7fffffff, dup = -1;
        unsigned bw = 0u, dw = 0u;
#pragma unroll
        for (int j = 0; j < NW; j++) {
            const bool on = W > 0 || (unsigned)j < g.ways;
            const bool is_dup = on && (w[j] & g.tag_mask) == tag1;
            dup = is_dup ? j : dup; dw = is_dup ? w[j] : dw;
            const int pj = w[j] == 0u ? -1 : sa_prio(w[j]);
            const bool cand = on && (w[j] == 0u || sa_stamp(g, w[j]) != cur) && pj < bp;
            best = cand ? j : best; bp = cand ? pj : bp; bw = cand ? w[j] : bw;
        }
        if (dup >= 0) {   // another copy of the key got here first: its priority is the maximum over the copies
            if (sa_prio(dw) < agg) {   // (a CAS, not a maximum: see sa_raise)
                const int old = sa_raise(g, &tags[dup], dw, agg);
                if (old >= 0) { atomicSub(&s_delta[old], 1); atomicAdd(&s_delta[agg], 1); }
            }
            done = true;
        } else if (best < 0) {
            done = true;   // every way of the set was filled in this batch: the key is not kept
        } else {
            // (two-copy arena: the new row goes to the copy the victim's word does NOT name, and the word says so)
            const unsigned prev = atomicCAS(&tags[best], bw, sa_word(g, tag1, cur, agg, sa_sel(g, bw) ^ 1u));
            if (prev == bw) { way = best; old_prio = bp; old_word = bw; done = true; }
            else {
#pragma unroll
                for (int j = 0; j < NW; j++) w[j] = j == best ? prev : w[j];
            }
        }
    }
    if (way < 0) return;
    if (old_prio >= 0) { atomicSub(&s_delta[old_prio], 1); atomicAdd(&s_stat[1], 1); }
    else atomicAdd(&s_stat[0], 1);
    atomicAdd(&s_delta[agg], 1);
    unsigned char *drow = args.a.arena + (long long)sa_entry(g, set, (unsigned)way, (sa_sel(g, old_word) ^ 1u) << kSaSelShift) * args.row_bytes;
    if constexpr (PIECES > 0) {
        if constexpr (PIECES > 0) reinterpret_cast<NU *>(drow)[0] = r0;
        if constexpr (PIECES > 1) reinterpret_cast<NU *>(drow)[1] = r1;
        if constexpr (PIECES > 2) reinterpret_cast<NU *>(drow)[2] = r2;
        if constexpr (PIECES > 3) reinterpret_cast<NU *>(drow)[3] = r3;
        if constexpr (PIECES > 4) reinterpret_cast<NU *>(drow)[4] = r4;
        if constexpr (PIECES > 5) reinterpret_cast<NU *>(drow)[5] = r5;
        if constexpr (PIECES > 6) reinterpret_cast<NU *>(drow)[6] = r6;
        if constexpr (PIECES > 7) reinterpret_cast<NU *>(drow)[7] = r7;
        if constexpr (PIECES > 8) reinterpret_cast<NU *>(drow)[8] = r8;
        if constexpr (PIECES > 9) reinterpret_cast<NU *>(drow)[9] = r9;
        if constexpr (PIECES > 10) reinterpret_cast<NU *>(drow)[10] = r10;
        if constexpr (PIECES > 11) reinterpret_cast<NU *>(drow)[11] = r11;
        if constexpr (PIECES > 12) reinterpret_cast<NU *>(drow)[12] = r12;
        if constexpr (PIECES > 13) reinterpret_cast<NU *>(drow)[13] = r13;
        if constexpr (PIECES > 14) reinterpret_cast<NU *>(drow)[14] = r14;
        if constexpr (PIECES > 15) reinterpret_cast<NU *>(drow)[15] = r15;
        if constexpr (!std::is_same<TAIL, NoTail>::value) *reinterpret_cast<NT *>(drow + PIECES * sizeof(U)) = rt;   // (non-temporal stores for the row: measured, no difference)
    } else if ((args.row_bytes & 15) == 0) { for (int c = 0; c < args.row_bytes; c += 16) *reinterpret_cast<float4 *>(drow + c) = *reinterpret_cast<const float4 *>(srow + c); }
    else { for (int c = 0; c < args.row_bytes; c++) drow[c] = srow[c]; }
    // three tiers: what this thread evicted becomes a member of the alt-key set (evlfu_8.cpp:617-620,654-658), behind its own stores
    if (old_prio >= 0 && args.c3_tags != nullptr) c3_insert_key(C3Set{args.c3_tags, args.c3_nset, args.c3_stat}, sa_key_of(args.sau, g, set, old_word));
}

template <int PIECES, typename U, typename TAIL = NoTail>
__global__ void __launch_bounds__(256) cache_batch_sampled_kernel(const BatchArgs args) {
    __shared__ int s_delta[kMaxBuckets];
    __shared__ int s_stat[3];
    __shared__ unsigned long long s_vict[256];   // keys this block evicted (alt-key tier attached)
    __shared__ int s_nvict, s_vbase;
    for (int i = threadIdx.x; i < kMaxBuckets; i += blockDim.x) s_delta[i] = 0;
    if (threadIdx.x < 3) s_stat[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_nvict = 0;
    __syncthreads();
    const long long n = args.B * args.T;
    const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned info = m < n ? args.miss_info[m] : 0u;
    if (m < n && args.hit) args.hit[m] = (info >> 30) & ~(info >> 31) & 1u;
    if (info & 0x80000000u)
        sampled_insert_one<PIECES, U, TAIL>(args, (int)(m % args.T), (unsigned)args.requests[m], (int)((info >> 24) & 63u), info & 0xffffffu,
                                   (info & 0x40000000u) != 0, m, s_delta, s_stat, args.vict_cnt ? s_vict : nullptr, &s_nvict);
    __syncthreads();
    if (args.vict_cnt && s_nvict) {   // the block's victims behind one of the kReplicas lists: one atomic per block
        const int r = blockIdx.x % kReplicas;
        if (threadIdx.x == 0) s_vbase = atomicAdd(&args.vict_cnt[r], s_nvict);
        __syncthreads();
        for (int i = threadIdx.x; i < s_nvict; i += blockDim.x)
            if (s_vbase + i < args.vict_cap) args.evicted_keys[(long long)r * args.vict_cap + s_vbase + i] = s_vict[i];
    }
    if (threadIdx.x < kPartCols) {
        const int i = threadIdx.x;
        const int v = i <= args.T ? s_delta[i] : (i >= 33 && i <= 35) ? s_stat[i - 33] : 0;
        if (v) atomicAdd(&args.part2[(blockIdx.x % kReplicas) * kPartCols + i], v);
    }
}

// List form (tables in HBM): block j takes the misses K1's block j listed, one wave, a record per lane -- 2 048 waves of
// mostly busy lanes instead of 6 656 waves with one lane in eight busy, and everything a miss needs arrives with its
// record (the per-position form reads the probe word, then the request row, then starts).
constexpr int kListVictMax = 512;   // (the folded two-tier probe lists 16 T <= 432 records per block)
template <int PIECES, typename U, typename TAIL>
__device__ __forceinline__ void sampled_list_block(const BatchArgs &args, int bid);
template <int PIECES, typename U, typename TAIL = NoTail>
__global__ void __launch_bounds__(256) cache_batch_sampled_list_kernel(const BatchArgs args) { sampled_list_block<PIECES, U, TAIL>(args, (int)blockIdx.x); }
// the set-associative policy's update: the same lists, sa_insert_one per record.  The list's length, the lane's first
// record and the table base addresses (into LDS: a per-lane index into the kernel arguments is a memory access) are
// asked for together -- three dependent round trips per record: those, {set line, source row}, the CAS.
template <int PIECES, typename U, typename TAIL>
__device__ __forceinline__ void sa_list_block(const BatchArgs &args, int bid);
template <int PIECES, typename U, typename TAIL = NoTail>
__global__ void __launch_bounds__(256) cache_batch_sa_list_kernel(const BatchArgs args) { sa_list_block<PIECES, U, TAIL>(args, (int)blockIdx.x); }
// both tiers of a two- / three-tier lookup in one launch: blocks [0, g1) take C1's lists, [g1, 2 g1) C2's (the route filter
// makes the two updates independent, as for the sampled pair)
template <int P1, typename U1, typename T1, int P2, typename U2, typename T2>
__global__ void __launch_bounds__(256) cache_batch_sa_list2_kernel(const BatchArgs args1, const BatchArgs args2) {
    if ((int)blockIdx.x < args1.g1) sa_list_block<P1, U1, T1>(args1, (int)blockIdx.x);
    else sa_list_block<P2, U2, T2>(args2, (int)blockIdx.x - args1.g1);
}
template <int PIECES, typename U, typename TAIL>
__device__ __forceinline__ void sa_list_block(const BatchArgs &args, int bid) {
    __shared__ int s_delta[kMaxBuckets];
    __shared__ int s_stat[3];
    __shared__ const unsigned char *s_table[32];
    for (int i = threadIdx.x; i < kMaxBuckets; i += blockDim.x) s_delta[i] = 0;
    if (threadIdx.x < 3) s_stat[threadIdx.x] = 0;
    const uint4 *rec = args.miss_rec + (long long)bid * args.list_cap;
    const int nw = (int)blockDim.x >> 6;
    const int i0 = ((int)threadIdx.x & 63) * nw + ((int)threadIdx.x >> 6);
    const int n = args.list_cnt[bid];
    uint4 r = rec[i0 < args.list_cap ? i0 : 0];
    if (threadIdx.x < 32) s_table[threadIdx.x] = args.backing[threadIdx.x];
    __syncthreads();
    // (the usual geometry -- 8-way sets: a tier alone, both tiers of the reference's 1 : 2 pair -- with the way count known to the compiler)
    auto run = [&](auto wc) {
        constexpr int W = decltype(wc)::value;
        for (int i = i0; i < n; i += 64 * nw) {
            if (i != i0) r = rec[i];
            const int t = (int)(r.y & 0xffu);
            sa_insert_one<PIECES, U, TAIL, W>(args, s_table[t & 31], t, r.x, (int)((r.y >> 8) & 0xffu), r.z, r.w, s_delta, s_stat);
        }
    };
    if (args.sa.ways == 8u) run(std::integral_constant<int, 8>{});
    else run(std::integral_constant<int, 0>{});
    __syncthreads();
    if (threadIdx.x < kPartCols) {
        const int i = threadIdx.x;
        const int v = i <= args.T ? s_delta[i] : (i >= 33 && i <= 35) ? s_stat[i - 33] : 0;
        if (v) atomicAdd(&args.part2[(bid % kReplicas) * kPartCols + i], v);
    }
}
// both tiers of a two-tier lookup in one launch (their updates are independent once the probe has stamped the route
// filter): blocks [0, g1) take C1's lists, [g1, 2 g1) C2's
template <int P1, typename U1, typename T1, int P2, typename U2, typename T2>
__global__ void __launch_bounds__(256) cache_batch_sampled_list2_kernel(const BatchArgs args1, const BatchArgs args2) {
    if ((int)blockIdx.x < args1.g1) sampled_list_block<P1, U1, T1>(args1, (int)blockIdx.x);
    else sampled_list_block<P2, U2, T2>(args2, (int)blockIdx.x - args1.g1);
}
template <int PIECES, typename U, typename TAIL>
__device__ __forceinline__ void sampled_list_block(const BatchArgs &args, int bid) {
    __shared__ int s_delta[kMaxBuckets];
    __shared__ int s_stat[3];
    __shared__ unsigned long long s_vict[kListVictMax];   // keys this block evicted (alt-key tier attached: lists of at most kListVictMax records)
    __shared__ int s_nvict, s_vbase;
    for (int i = threadIdx.x; i < kMaxBuckets; i += blockDim.x) s_delta[i] = 0;
    if (threadIdx.x < 3) s_stat[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_nvict = 0;
    __syncthreads();
    const int n = args.list_cnt[bid];
    const uint4 *rec = args.miss_rec + (long long)bid * args.list_cap;
    // (records dealt round-robin over the block's waves: a 50-record list on two waves is 25 busy lanes each, not one wave
    //  of 50 and an idle one)
    const int nw = (int)blockDim.x >> 6;
    for (int i = ((int)threadIdx.x & 63) * nw + ((int)threadIdx.x >> 6); i < n; i += 64 * nw) {
        const uint4 r = rec[i];
        sampled_insert_one<PIECES, U, TAIL>(args, (int)(r.y & 0xffu), r.x, (int)((r.y >> 8) & 0xffu), r.z, (r.y & 0x10000u) != 0, (long long)r.w,
                                   s_delta, s_stat, args.vict_cnt ? s_vict : nullptr, &s_nvict);
    }
    __syncthreads();
    if (args.vict_cnt && s_nvict) {   // the block's victims behind one of the kReplicas lists: one atomic per block
        const int r = bid % kReplicas;
        if (threadIdx.x == 0) s_vbase = atomicAdd(&args.vict_cnt[r], s_nvict);
        __syncthreads();
        for (int i = threadIdx.x; i < s_nvict; i += blockDim.x)
            if (s_vbase + i < args.vict_cap) args.evicted_keys[(long long)r * args.vict_cap + s_vbase + i] = s_vict[i];
    }
    if (threadIdx.x < kPartCols) {
        const int i = threadIdx.x;
        const int v = i <= args.T ? s_delta[i] : (i >= 33 && i <= 35) ? s_stat[i - 33] : 0;
        if (v) atomicAdd(&args.part2[(bid % kReplicas) * kPartCols + i], v);
    }
}

// close of a sampled batch (one block): folds K1's and the update's replica rows into the state
// (It only folds counters, and nothing the probe or the update read depends on them: it runs every kCloseEvery-th batch,
// before a sweep / rebuild, and before anything reads the state -- 5 us per batch become 0.6.)
struct CloseArgs {
    BatchState *bs; int *part1, *part2; int *host_words;   // host_words: [0] tombstones, [1] flush wanted, [2] ordinal of the close
    int T, cap, max_perfect, rebuild, n_batches;
    long long n_requests;
};
constexpr int kCloseEvery = 8;
__device__ __forceinline__ void sampled_close_block(const CloseArgs &args);
__global__ void __launch_bounds__(256) cache_batch_sampled_close_kernel(const CloseArgs args) { sampled_close_block(args); }
// two tiers closed by one launch (block 0: C1, block 1: C2): the second launch of a pair costs as much as the first
__global__ void __launch_bounds__(256) cache_batch_sampled_close2_kernel(const CloseArgs args0, const CloseArgs args1) {
    if (blockIdx.x == 0) sampled_close_block(args0); else sampled_close_block(args1);
}
__device__ __forceinline__ void sampled_close_block(const CloseArgs &args) {
    __shared__ long long s_col[2][kPartCols];
    __shared__ BatchState sb;
    const int nw = (int)(sizeof(BatchState) / sizeof(int));
    {
        int *dst = reinterpret_cast<int *>(&sb);
        const int *src = reinterpret_cast<const int *>(args.bs);
        for (int i = threadIdx.x; i < nw; i += blockDim.x) dst[i] = src[i];
    }
    if (threadIdx.x < 2 * kPartCols) s_col[threadIdx.x / kPartCols][threadIdx.x % kPartCols] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < kReplicas * kPartCols; i += blockDim.x) {
        const int v1 = args.part1[i], v2 = args.part2[i];
        if (v1) { atomicAdd(reinterpret_cast<unsigned long long *>(&s_col[0][i % kPartCols]), (unsigned long long)(long long)v1); args.part1[i] = 0; }
        if (v2) { atomicAdd(reinterpret_cast<unsigned long long *>(&s_col[1][i % kPartCols]), (unsigned long long)(long long)v2); args.part2[i] = 0; }
    }
    __syncthreads();
    BatchState *b = &sb;
    if ((int)threadIdx.x <= args.T) b->cnt[threadIdx.x] += (int)(s_col[0][threadIdx.x] + s_col[1][threadIdx.x]);
    __syncthreads();
    if (threadIdx.x == 0) {
        b->n_hits += s_col[0][38]; b->n_perfect_hits += s_col[0][39];
        b->count += (int)s_col[1][33];
        b->n_free = args.cap - b->count;
        b->n_evict += s_col[1][34];
        b->n_tomb += (int)s_col[1][35];
        if (b->n_tomb < 0) b->n_tomb = 0;
        if (args.rebuild == 1) b->n_tomb = 0;
        b->batch_id += args.n_batches;
        b->n_requests += args.n_requests;
        b->ticket_t = 0;
        args.host_words[0] = b->n_tomb;
        args.host_words[2] = (int)b->batch_id;   // the ordinal of the last closed batch: the host ignores reports older than its last sweep / rebuild
        args.host_words[1] = b->cnt[args.T] >= args.max_perfect ? 1 : 0;   // EvLFU flush (EvLFU_C1.py:36-44): the host launches it
        args.host_words[3] = b->count >= args.cap - (args.cap > 65536 ? args.cap / 256 : 0) ? 1 : 0;   // "full" as the two-tier routing reads it
    }
    __syncthreads();
    {
        const int *src = reinterpret_cast<const int *>(&sb);
        int *dst = reinterpret_cast<int *>(args.bs);
        for (int i = threadIdx.x; i < nw; i += blockDim.x) dst[i] = src[i];
    }
}

// EvLFU flush in sampled mode (rare: the top bucket has reached max_perfect): flush_n entries of the top priority go,
// first come first served over the arena.  Runs alone on the stream, between two batches (the close of a batch raises a
// flag in mapped host memory; the next batched call, or a stats / dump call, runs the flush first).
__global__ void __launch_bounds__(256) cache_batch_sampled_flush_kernel(BatchState *b, CacheArrays a, unsigned long long *slots,
                                                                        const int *eslot, int cap, int T, int flush_n, int max_perfect) {
    __shared__ int s_tot[8];
    __shared__ int s_gone;
    // Nothing in this launch writes cnt[T] (the finish kernel behind it does), so every block reads the same value:
    // one decision and one `want` for the whole grid.  The flag that asked for this launch is raised by a close and
    // cleared by the host without synchronising -- a close already in flight can raise it again after the flush has
    // run: that second launch finds the top bucket below max_perfect and does nothing.
    const int top_n = b->cnt[T];
    if (top_n < max_perfect) return;
    if (threadIdx.x == 0) s_gone = 0;
    const int want = flush_n < top_n ? flush_n : top_n;
    __syncthreads();
    for (long long e0 = (long long)blockIdx.x * blockDim.x; e0 < cap; e0 += (long long)gridDim.x * blockDim.x) {   // block-uniform trip count
        const long long e = e0 + threadIdx.x;
        const bool top = e < cap && (a.ekey[e] & kKeyMask) != kEmpty && a.eagg[e] == T;
        const int tk = block_reserve(&b->ticket_t, top, s_tot);
        if (top && tk < want) {
            slots[eslot[e]] = kTomb;
            a.ekey[e] = kEmpty;
            atomicAdd(&s_gone, 1);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_gone) atomicAdd(&b->flush_gone, s_gone);
}
// the same for the set-associative policy: the top-priority ways, first come first served over the sets
__global__ void __launch_bounds__(256) cache_batch_sa_flush_kernel(BatchState *b, const SaGeom g, int T, int flush_n, int max_perfect) {
    __shared__ int s_tot[8];
    __shared__ int s_gone;
    const int top_n = b->cnt[T];
    if (top_n < max_perfect) return;
    if (threadIdx.x == 0) s_gone = 0;
    const int want = flush_n < top_n ? flush_n : top_n;
    __syncthreads();
    const long long n_ent = ((long long)g.nset << g.sub_shift) * g.ways;
    for (long long e0 = (long long)blockIdx.x * blockDim.x; e0 < n_ent; e0 += (long long)gridDim.x * blockDim.x) {   // block-uniform trip count
        const long long e = e0 + threadIdx.x;
        const unsigned es = (unsigned)((e < n_ent ? e : 0) / g.ways);
        unsigned *wp = sa_ways_ptr(g, es) + (unsigned)((e < n_ent ? e : 0) - (long long)es * g.ways);
        const unsigned w = e < n_ent ? *wp : 0u;
        const bool top = w != 0u && sa_prio(w) == T;
        const int tk = block_reserve(&b->ticket_t, top, s_tot);
        if (top && tk < want) { *wp = 0u; atomicAdd(&s_gone, 1); }
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_gone) atomicAdd(&b->flush_gone, s_gone);
}
// ... and its finish: one thread folds what the scan removed into the counters and counts the flush -- only if it
// removed anything (n_flush is what batch_stats reports).
__global__ void cache_batch_sampled_flush_finish_kernel(BatchState *b, int T) {
    const int gone = b->flush_gone;
    b->flush_gone = 0;
    b->ticket_t = 0;
    if (gone <= 0) return;
    b->cnt[T] -= gone; b->count -= gone; b->n_free += gone; b->n_tomb += gone;
    b->n_flush += 1;
}

// Tombstone sweep (between two batches, instead of most rebuilds): a tombstone whose successor is EMPTY ends its chain
// and can be EMPTY itself -- and so can the tombstones in front of it.  One thread per run of kSweepRun slots, walking
// backwards from the state of the slot behind its run (read before anybody could have changed it to "empty" only
// makes the answer conservative).  One pass over the slot array (64 MB at the 10 % Kaggle cache: ~20 us) against
// ~170 us for clear + re-insert.
constexpr int kSweepRun = 16;
__global__ void __launch_bounds__(256) cache_batch_sweep_kernel(BatchState *b, unsigned long long *slots, long long nslot) {
    __shared__ int s_freed;
    if (threadIdx.x == 0) s_freed = 0;
    __syncthreads();
    int freed = 0;
    for (long long r0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * kSweepRun; r0 < nslot; r0 += (long long)gridDim.x * blockDim.x * kSweepRun) {
        unsigned long long w[kSweepRun];
#pragma unroll
        for (int j = 0; j < kSweepRun; j++) w[j] = r0 + j < nslot ? slots[r0 + j] : kEmpty;
        const long long nx = r0 + kSweepRun < nslot ? r0 + kSweepRun : 0;
        bool next_empty = slots[nx] == kEmpty;
#pragma unroll
        for (int j = kSweepRun - 1; j >= 0; j--) {
            if (r0 + j < nslot) {
                const bool tomb = w[j] >= kTomb1;
                if (tomb && next_empty) { slots[r0 + j] = kEmpty; freed++; }
                else next_empty = w[j] == kEmpty;
            }
        }
    }
    if (freed) atomicAdd(&s_freed, freed);
    __syncthreads();
    if (threadIdx.x == 0 && s_freed) atomicSub(&b->n_tomb, s_freed);
}

// Alt-key tier, batched fill: the keys K4 EVICTED from this tier in this batch (not the flushed ones: evlfu_8.cpp:617-620,
// 654-658 queue last_evicted only) become members of C3.  One thread per victim; positions [n_free, n_free + victims)
// of the free stack are K4's (the state is the one K3 / K4 left: the close has not run yet).
__global__ void __launch_bounds__(256) c3_batch_insert_kernel(const BatchArgs args, const C3Batch c3) {
    const BatchState *b = args.bs;
    const int n_vict = b->flush_t > 0 ? b->pos_ticket : (b->ticket < b->need ? b->ticket : b->need);
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_vict) return;
    const unsigned long long rec = args.evicted_keys[b->n_free + i];
    if (rec >> 63) return;
    const unsigned long long key = rec & kKeyMask;
    const long long base = c3_base(key, c3.nset);
    for (int attempt = 0; attempt < 4; attempt++) {   // (a CAS lost to another newcomer of the same set: look again)
        unsigned long long w[kSetWays];
        int present = 0, empty = -1, plain = -1;
#pragma unroll
        for (int j = 0; j < kSetWays; j++) {
            w[j] = atomicOr(&c3.tags[base + j], 0ull);
            present |= (w[j] & kKeyMask) == key;
            if (w[j] == 0ull && empty < 0) empty = j;
            if (w[j] != 0ull && !(w[j] & kC3Flag) && plain < 0) plain = j;
        }
        if (present) return;
        int way = empty >= 0 ? empty : plain;
        if (way < 0) {   // every way has had its second chance now
#pragma unroll
            for (int j = 0; j < kSetWays; j++) atomicAnd(&c3.tags[base + j], ~kC3Flag);
            way = (int)((key >> 3) % kSetWays);
        }
        unsigned long long expect = 0ull;
#pragma unroll
        for (int j = 0; j < kSetWays; j++) expect = j == way ? w[j] : expect;
        if (empty < 0 && plain < 0) expect &= ~kC3Flag;
        if (atomicCAS(&c3.tags[base + way], expect, key) == expect) {
            if (expect == 0ull) atomicAdd(reinterpret_cast<unsigned long long *>(&c3.stat[0]), 1ull);
            return;
        }
    }
}

// the same fill from the victim lists of the sampled update (kReplicas lists; the kernel empties them)
// (one launch for the victims of both tiers: the first half of the grid takes C1's lists, the second C2's -- the launches
//  are chains of four dependent round trips each, 12.5 us apiece whatever the list length)
__global__ void __launch_bounds__(256) c3_batch_insert_lists_kernel(const BatchArgs args1, const BatchArgs args2, const C3Batch c3) {
    // (args.vict_cnt points at this batch's half of the 2 x kReplicas list lengths -- halves by batch parity; the other
    //  half, consumed one batch ago, is zeroed here for the next batch: no launch just to reset 32 counters)
    const int half_grid = (int)gridDim.x / 2;
    const bool second = (int)blockIdx.x >= half_grid;
    const BatchArgs &args = second ? args2 : args1;
    const int bid = (int)blockIdx.x - (second ? half_grid : 0);
    const int r = bid % kReplicas, part = bid / kReplicas, parts = half_grid / kReplicas;
    int n = args.vict_cnt[r];
    if (n > args.vict_cap) n = args.vict_cap;
    for (int i = part * blockDim.x + threadIdx.x; i < n; i += parts * blockDim.x)
        c3_insert_key(C3Set{c3.tags, c3.nset, c3.stat}, args.evicted_keys[(long long)r * args.vict_cap + i]);
    if (part == 0 && threadIdx.x == 0) args.vict_other[r] = 0;
}

// Host-memory miss tier: after the fill (K5) every missed key that got an entry is served from its ARENA row, so
// each missing row crosses the bus once (the de-duplicated fetch of K5) instead of once per request that asked
// for it plus once for the fill.  One thread per (request, table) position; keys that found no room keep their
// host address.
__global__ void __launch_bounds__(256) cache_batch_patch_ptrs_kernel(const BatchArgs args) {
    const long long n = args.B * args.T;
    const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n) return;
    const unsigned info = args.miss_info[m];
    if (!(info & 0x80000000u)) return;
    const unsigned long long key = ((unsigned long long)(m % args.T + 1) << 32) | (unsigned)args.requests[m];
    unsigned long long end_slot;
    if (args.staging && args.row_ptrs[m] < 0) {   // orphan of the full hash: its own staged row
        args.row_ptrs[m] = (long long)(args.staging + (args.stage_rows + args.row_ptrs[m]) * args.row_bytes);
        return;
    }
    const int e = probe_ro(args.slots, args.mask, key, end_slot);
    if (e >= 0) args.row_ptrs[m] = (long long)(args.a.arena + (long long)e * args.row_bytes);
    else if (e == kPending && args.staging && ((args.staged_mask >> (int)(m % args.T)) & 1u))   // no room in the cache: the staged copy
        args.row_ptrs[m] = (long long)(args.staging + (long long)args.slot_stage[end_slot] * args.row_bytes);
}

// The same for a PAIR of tiers over host-memory / file-backed tables (evs_cache_lookup_batch_c1c2[c3]; the reference's tiers
// read their misses from files inside the request: evlfu_8.cpp:380-414 get_from_file).  Both tiers' positions live in ONE
// (B,T) pointer table (a1.row_ptrs) with a class byte each (row_tier); a position is a miss of the tier its request's
// routing chose.  After both tiers' updates: the key has an entry in its tier -> that arena row; it is pending there with a
// staged row (no room) -> the staged row; a C2 position whose key another request of the batch routed to C1 (only odd
// tables can go both ways) -> C1's copy, at C1's precision; none of these (no room under the sampled update, whose
// tables are all device-visible) -> the backing row as the probe left it.
__global__ void __launch_bounds__(256) cache_batch_patch_ptrs2_kernel(const BatchArgs a1, const BatchArgs a2, unsigned char *row_tier) {
    const long long n = a1.B * a1.T;
    const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n) return;
    const int k = (a1.miss_info[m] & 0x80000000u) ? 1 : ((a2.miss_info[m] & 0x80000000u) ? 2 : 0);
    if (!k) return;
    const BatchArgs &own = k == 1 ? a1 : a2;
    const int t = (int)(m % a1.T);
    const unsigned long long key = ((unsigned long long)(t + 1) << 32) | (unsigned)a1.requests[m];
    if (own.staging && a1.row_ptrs[m] < 0) {   // orphan of the full hash: its own staged row
        a1.row_ptrs[m] = (long long)(own.staging + (own.stage_rows + a1.row_ptrs[m]) * own.row_bytes);
        return;
    }
    unsigned long long es;
    int e = probe_ro(own.slots, own.mask, key, es);
    if (e >= 0) { a1.row_ptrs[m] = (long long)(own.a.arena + (long long)e * own.row_bytes); return; }
    if (e == kPending && own.staging && ((own.staged_mask >> t) & 1u)) {
        a1.row_ptrs[m] = (long long)(own.staging + (long long)own.slot_stage[es] * own.row_bytes);
        return;
    }
    if (k == 2) {
        e = probe_ro(a1.slots, a1.mask, key, es);
        if (e >= 0) { a1.row_ptrs[m] = (long long)(a1.a.arena + (long long)e * a1.row_bytes); row_tier[m] = 1; return; }
        if (e == kPending && a1.staging && ((a1.staged_mask >> t) & 1u)) {
            a1.row_ptrs[m] = (long long)(a1.staging + (long long)a1.slot_stage[es] * a1.row_bytes);
            row_tier[m] = 1;
        }
    }
}

// File mode, after K2 / K3: the batch's new keys as one list for the host's reader pool (new key i = list base of
// K2's block + rank, the same numbering K5 uses), and for every new key the hash slot -> i map the patch kernel needs.
__global__ void __launch_bounds__(256) cache_batch_list_kernel(const BatchArgs args) {
    const int my_cnt = args.block_cnt[blockIdx.x], my_base = args.block_base[blockIdx.x];
    for (int local = threadIdx.x; local < my_cnt; local += blockDim.x) {
        const int slot = args.new_slot[(long long)blockIdx.x * 256 + local];
        args.new_keys[my_base + local] = args.slots[slot] & kKeyMask;
        args.slot_stage[slot] = my_base + local;
    }
}
// File mode, after the consumers: new keys that found no room stop being pending (K5 tombstones them at once in the
// other modes; here their staged rows had to stay addressable through the hash until the batch was served).
__global__ void __launch_bounds__(256) cache_batch_unstage_kernel(const BatchArgs args) {
    __shared__ int s_drop;
    if (threadIdx.x == 0) s_drop = 0;
    __syncthreads();
    BatchState *b = args.bs;
    const int n_free = b->n_free + (b->flush_t > 0 ? b->pos_ticket : (b->ticket < b->need ? b->ticket : b->need));
    const int n_assign = b->n_assign < n_free ? b->n_assign : n_free;
    const int my_cnt = args.block_cnt[blockIdx.x], my_base = args.block_base[blockIdx.x];
    for (int local = threadIdx.x; local < my_cnt; local += blockDim.x)
        if (my_base + local >= n_assign) { args.slots[args.new_slot[(long long)blockIdx.x * 256 + local]] = kTomb; atomicAdd(&s_drop, 1); }
    __syncthreads();
    if (threadIdx.x == 0 && s_drop) atomicAdd(&args.part2[(blockIdx.x % kReplicas) * kPartCols + 37], s_drop);
}

}  // namespace evs

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct evs_aprx {
    evs::AprxArrays x{};
    long long nslot = 0, cap = 0;
    int n_tables = 0;
    bool has_alt = false;
    // batched form: the set-associative key set (see C3Batch)
    unsigned long long *tags = nullptr;
    long long nset = 0;
    long long *bstat = nullptr;   // device: [0] members, [1] alt hits served
    int used = 0;                 // 0 fresh, 1 exact machine, 2 batched form
};
struct evs_cache {
    evs::CacheState host;      // configuration mirror
    evs::CacheState *st = nullptr;
    evs::CacheArrays a{};
    long long nslot = 0;
    long long bnslot = 0;   // slots of the batched path's hash (its own, sparser table: see evs_cache_create)
    const unsigned char *backing[evs::kMaxTables] = {nullptr};
    long long backing_rows[evs::kMaxTables] = {0};
    bool has_backing = false;
    // batched path
    evs::BatchState *bs = nullptr;
    int *eslot = nullptr, *new_slot = nullptr;
    unsigned *miss_info = nullptr;
    int *block_cnt = nullptr, *block_base = nullptr, *part1 = nullptr, *part2 = nullptr;
    unsigned long long *bslots = nullptr;
    long long *row_ptrs = nullptr, *iota = nullptr;
    uint4 *miss_rec = nullptr; int *list_cnt = nullptr;   // sampled update, list form
    unsigned char *row_tier = nullptr;   // two-tier batched lookup: which tier's codec decodes each row
    long long max_batch = 0;
    int used = 0;  // 0 fresh, 1 exact path, 2 batched path
    int *estamp = nullptr;      // allocated when the backing tables live in host memory
    bool host_backing = false;
    long long stamp_counter = 0;
    int *host_tomb = nullptr;   // mapped host word: tombstones after the last finished batch (read without a sync)
    int *host_tomb_dev = nullptr;
    // device memory comes in three slabs (create / first batched call / per-batch buffers): one allocation each, carved
    // at 256-byte boundaries.  (Fifteen separate hipMallocs left the batched path bimodal from process to process --
    // 105 or 160 us per 16 384-request batch on the same box -- depending on how the driver happened to back them.)
    void *slab_create = nullptr, *slab_batch = nullptr, *slab_perbatch = nullptr;
    // file-backed miss tier
    evs_filetier *ft = nullptr;
    unsigned staged_mask = 0;
    unsigned long long *new_keys = nullptr, *new_keys_host = nullptr;
    unsigned char *staging = nullptr, *staging_dev = nullptr;
    int *slot_stage = nullptr;
    long long staging_rows = 0;
    long long n_staged_rows = 0;   // rows the reader pool has fetched so far (statistics)
    // fork / join of one batch: the hash part of the policy update (K2, K4) on a side stream under the consumer
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int fork_mode = -1;            // -1: not decided yet (EVS_CACHE_FORK, default off), 0 / 1
    long long batch_calls = 0, last_sweep_call = -100, last_hk_call = 0, last_flush_call = 0;
    int pending_batches = 0; long long pending_requests = 0;   // sampled update: batches whose counters the close has not folded yet
    unsigned long long *evicted_keys = nullptr;   // batched three-tier lookup: what K4 evicted, for the alt-key tier
    unsigned *route_filter = nullptr;   // two-tier sampled update (held by C1): see BatchArgs::route_filter
    unsigned long long *vict_keys = nullptr; int *vict_cnt = nullptr; long long vict_cap = 0;   // ... what the sampled update evicted (kReplicas lists)   // tombstone housekeeping: a sweep first, a rebuild if that was not enough
    int batch_policy = -1;         // policy update of the batched path: 0 plan-based (K2..K6), 1 sampled (one kernel), 2 set-associative (evs_hash.h); -1: EVS_CACHE_POLICY, default sampled
    // set-associative policy (evs_hash.h): this tier's view of the set records and the key universe; the words themselves
    // are one allocation a tier PAIR shares (ONE 128-byte record per set: both tiers' ways)
    evs::SaGeom sa{};
    evs::SaUniverse sau{};
    struct SaShared { unsigned *tags = nullptr; size_t bytes = 0; int refs = 0; } *sa_mem = nullptr;
    unsigned char *sa_arena = nullptr;   // two-copy arena of a dual geometry (a.arena points here then)
    unsigned char *arena_create = nullptr;   // the one-row-per-entry arena evs_cache_create made (freed when sa_arena replaces it)
    // the exact policy as a resident server (evs_cache_serve_*): a mailbox in mapped host memory, its own stream
    unsigned *mbox = nullptr, *mbox_dev = nullptr;   // 3 lines of 128 bytes: request, control, answer
    // (Round 6 tried the request + control lines in DEVICE memory, written by the host through the large-BAR aperture:
    //  tools/mailbox_probe.hip's bare echo makes a round trip in 1.8 us against 2.4, the server's p50 did not move -- 14.3-14.9
    //  against 14.4-14.8 us over four runs each --, so the lines stay where an ordinary store reaches them.)
    volatile unsigned *req_host = nullptr;           // the host's view of the request + control lines
    unsigned *req_dev = nullptr;                     // the device's
    hipStream_t serve_stream = nullptr;
    bool serving = false;
    unsigned serve_seq = 0;
    float *serve_ring = nullptr; int serve_slots = 0, serve_thres = -1;
    long long serve_idle_ticks = 0;
    // ordering of the server against everything else that touches the exact state or reads the ring:
    //   exact_done: recorded behind every exact-path launch on the CALLER's stream; a server started afterwards waits for it
    //               on its own stream (two kernels must never mutate the lists / the map / the free stack at once);
    //   slot_done[s]: recorded by evs_cache_serve_consumed on the stream that READS ring slot s; the request that is about
    //               to hand slot s out again waits for it on the host (the server overwrites the slot in host order).
    hipEvent_t exact_done = nullptr; bool exact_pending = false;
    std::vector<hipEvent_t> slot_done; std::vector<char> slot_busy;
    int inline_mode = -1;          // the update inside the probe launch (evs_fused_rf.hip): -1 not decided (EVS_CACHE_INLINE, default on), 0 / 1
};

// ---- set-associative geometry (host) ----
namespace {
constexpr unsigned kSaSingleWays = 8;    // a tier alone: 8 ways = a 32-byte record (EVS_SA_WAYS=16: 64 bytes, developer A/B)
bool sa_make_universe(const long long *rows1, const long long *rows2, int T, evs::SaUniverse &u) {
    unsigned long long n = 0;
    if (T > 32) return false;
    for (int t = 0; t < 32; t++) {
        u.row_base[t] = (unsigned)n;
        if (t < T) n += (unsigned long long)std::max<long long>(rows1[t], rows2 ? rows2[t] : 0);
        if (n >= (1ull << 32)) return false;
    }
    int b = 1;
    while ((1ull << b) < n) b++;
    u.mask = b == 32 ? 0xffffffffu : (1u << b) - 1u;
    u.half = (unsigned)(b + 1) / 2u;
    u.n_tables = T;
    return true;
}
bool sa_make_geom(evs::SaGeom &g, unsigned nset, unsigned ways, unsigned w_off, unsigned line_words, const evs::SaUniverse &u, unsigned sub_shift = 0, unsigned dual = 0) {
    if (nset < 1 || ways < 1 || ways > (unsigned)evs::kSaMaxWays || (w_off & 3u) || (sub_shift && (ways & 3u))) return false;
    g.tags = nullptr; g.nset = nset; g.ways = ways; g.w_off = w_off; g.line_words = line_words; g.sub_shift = sub_shift;
    unsigned l = 0;
    while ((1ull << l) < nset) l++;
    g.div_l = l;
    g.div_m = nset > 1 ? (unsigned)(((1ull << 32) * ((1ull << l) - nset)) / nset + 1ull) : 0u;
    const unsigned max_tag1 = ((u.mask / nset) >> sub_shift) + 1u;
    unsigned tb = 1;
    while (tb < 32 && (max_tag1 >> tb)) tb++;
    if (tb > 22) return false;   // (at least 4 stamp bits)
    if (tb > 21) dual = 0;       // (... also beside the copy-select bit: such a tier keeps one arena row per way and the two-launch update)
    g.tag_bits = tb; g.tag_mask = (1u << tb) - 1u;
    g.dual = dual; g.stamp_mask = (1u << (26u - dual - tb)) - 1u;
    return true;
}
unsigned sa_single_ways() {
    static int w = -1;
    if (w < 0) { const char *e = getenv("EVS_SA_WAYS"); w = (e && atoi(e) == 16) ? 16 : (int)kSaSingleWays; }
    return (unsigned)w;
}
// can this cache take the set-associative form on its own (universe below 2^32 keys, tags that fit the word)
bool sa_single_feasible(const evs_cache *c, evs::SaUniverse *u_out = nullptr, evs::SaGeom *g_out = nullptr) {
    evs::SaUniverse u; evs::SaGeom g;
    const unsigned ways = sa_single_ways();
    if (!c->has_backing || c->host.cap < (long long)ways) return false;
    if (!sa_make_universe(c->backing_rows, nullptr, c->host.n_tables, u)) return false;
    // a tier alone keeps two arena rows per way (evs_hash.h: the two-copy arena; EVS_SA_DUAL=0: one, developer A/B)
    static const unsigned dual_on = (getenv("EVS_SA_DUAL") && getenv("EVS_SA_DUAL")[0] == '0') ? 0u : 1u;
    if (!sa_make_geom(g, (unsigned)(c->host.cap / ways), ways, 0, ways, u, 0, dual_on)) return false;
    if (u_out) *u_out = u;
    if (g_out) *g_out = g;
    return true;
}
// the geometry of a C1 + C2 pair that shares its set records: nset records of 128 bytes, C1's ways at word 0, C2's behind
// them.  nset = max(cap1 / 8, ceil(cap2 / 16)); C1 min(cap1 / nset, 16) ways; C2 TWO sub-sets of 8 ways when cap2 >= 16 nset
// (the reference's 1 : 2 split, evlfu_8.cpp:63-78: C1 8 ways + C2 2 x 8, every probe an 8-way search of 32 bytes), else
// min(cap2 / nset, 16) ways in one.  false: no such geometry (a tier would get fewer than 4 ways) -- each tier then keeps
// records of its own (two line requests per key)
bool sa_pair_geometry(const evs_cache *c1, const evs_cache *c2, evs::SaUniverse &u, evs::SaGeom &g1, evs::SaGeom &g2) {
    static const bool on = !(getenv("EVS_SA_PAIR") && getenv("EVS_SA_PAIR")[0] == '0');
    if (!on || !c1->has_backing || !c2->has_backing || c1->host.n_tables != c2->host.n_tables) return false;
    const long long cap1 = c1->host.cap, cap2 = c2->host.cap;
    long long nset = std::max<long long>(std::max<long long>(cap1 / 8, (cap2 + 15) / 16), 1);
    const long long w1 = std::min<long long>(cap1 / nset, evs::kSaMaxWays);
    long long w2 = std::min<long long>(cap2 / nset, evs::kSaMaxWays);
    unsigned sub2 = 0;
    if (w2 == 16) { w2 = 8; sub2 = 1; }
    if (w1 < 4 || w2 < 4) return false;
    if (!sa_make_universe(c1->backing_rows, c2->backing_rows, c1->host.n_tables, u)) return false;
    const unsigned off2 = (unsigned)((w1 + 3) / 4 * 4);
    return sa_make_geom(g1, (unsigned)nset, (unsigned)w1, 0, 32, u) && sa_make_geom(g2, (unsigned)nset, (unsigned)w2, off2, 32, u, sub2);
}
int sa_alloc(evs_cache *c, evs_cache *partner, hipStream_t st) {   // the set records of c (shared with partner), zeroed on st
    auto *m = new evs_cache::SaShared();
    m->bytes = (size_t)c->sa.nset * c->sa.line_words * 4;
    static const long long pad_mb = getenv("EVS_SA_PAD_MB") ? atoll(getenv("EVS_SA_PAD_MB")) : 0;   // developer A/B: allocation size vs page size
    const size_t alloc = std::max<size_t>(m->bytes, (size_t)pad_mb << 20);
    if (hipMalloc(&m->tags, alloc) != hipSuccess) { (void)hipGetLastError(); delete m; return EVS_ENOMEM; }
    if (hipMemsetAsync(m->tags, 0, m->bytes, st) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(m->tags); delete m; return EVS_EHIP; }
    for (evs_cache *t : {c, partner}) {   // two-copy arenas (nothing is resident yet: the batched path starts here)
        if (!t || !t->sa.dual) continue;
        const size_t ab = ((size_t)t->sa.nset << t->sa.sub_shift) * t->sa.ways * 2u * (size_t)t->host.row_bytes;
        if (hipMalloc(&t->sa_arena, ab) != hipSuccess) {
            (void)hipGetLastError(); (void)hipFree(m->tags); delete m;
            if (c->sa_arena) { (void)hipFree(c->sa_arena); c->sa_arena = nullptr; }
            return EVS_ENOMEM;
        }
        t->a.arena = t->sa_arena;
        if (t->arena_create) { (void)hipFree(t->arena_create); t->arena_create = nullptr; }   // (nothing is resident in it: the batched path starts here)
    }
    m->refs = partner ? 2 : 1;
    c->sa_mem = m; c->sa.tags = m->tags;
    if (partner) { partner->sa_mem = m; partner->sa.tags = m->tags; }
    return EVS_OK;
}
void sa_release(evs_cache *c) {
    if (!c->sa_mem) return;
    if (--c->sa_mem->refs == 0) { (void)hipFree(c->sa_mem->tags); delete c->sa_mem; }
    c->sa_mem = nullptr; c->sa.tags = nullptr;
}
// host restatement of sa_key_of (evs_hash.h)
unsigned long long sa_key_of_host(const evs::SaUniverse &u, const evs::SaGeom &g, unsigned es, unsigned w) {
    const unsigned q = (((w & g.tag_mask) - 1u) << g.sub_shift) | (es & ((1u << g.sub_shift) - 1u));
    unsigned x = q * g.nset + (es >> g.sub_shift);
    x ^= x >> u.half; x = (x * evs::kSaInv2) & u.mask;
    x ^= x >> u.half; x = (x * evs::kSaInv1) & u.mask;
    int t = 0;
    for (int k = 1; k < u.n_tables; k++) if (x >= u.row_base[k]) t = k;
    return ((unsigned long long)(t + 1) << 32) | (x - u.row_base[t]);
}
}  // namespace


extern "C" int evs_cache_destroy(evs_cache *c) {
    if (!c) return EVS_OK;
    void *ptrs[] = {c->slab_create, c->slab_batch, c->slab_perbatch, c->estamp, c->new_keys, c->slot_stage, c->evicted_keys, c->vict_keys, c->vict_cnt, c->route_filter, c->sa_arena, c->arena_create};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (c->host_tomb) (void)hipHostFree(c->host_tomb);
    if (c->new_keys_host) (void)hipHostFree(c->new_keys_host);
    if (c->staging) (void)hipHostFree(c->staging);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->mbox) {
        c->req_host[32] = 1u;   // stop
        __atomic_thread_fence(__ATOMIC_SEQ_CST);
        if (c->serve_stream) { (void)hipStreamSynchronize(c->serve_stream); (void)hipStreamDestroy(c->serve_stream); }
        (void)hipHostFree(c->mbox);
    }
    if (c->exact_done) (void)hipEventDestroy(c->exact_done);
    for (hipEvent_t e : c->slot_done) if (e) (void)hipEventDestroy(e);
    sa_release(c);
    delete c;
    return EVS_OK;
}

// one device allocation carved into 256-byte aligned pieces
namespace {
struct SlabPlan {
    struct Item { void **where; size_t bytes; };
    std::vector<Item> items;
    template <typename P> void add(P **where, long long bytes) { items.push_back({reinterpret_cast<void **>(where), (size_t)(bytes > 0 ? bytes : 1)}); }
    size_t total() const { size_t t = 0; for (auto &i : items) t += (i.bytes + 255) & ~(size_t)255; return t; }
    bool carve(void **slab_out) const {
        void *base = nullptr;
        if (hipMalloc(&base, total()) != hipSuccess) { (void)hipGetLastError(); return false; }
        size_t off = 0;
        for (auto &i : items) { *i.where = static_cast<char *>(base) + off; off += (i.bytes + 255) & ~(size_t)255; }
        *slab_out = base;
        return true;
    }
};
}  // namespace

extern "C" int evs_cache_create(evs_cache **out, int policy, int64_t capacity, int n_tables, int dim, int codec,
                                double flush_rate, double perfect_item_cap, int flush_extra, int perfect_mode) {
    using namespace evs;
    EVS_REQUIRE(out, "evs_cache_create: NULL out");
    EVS_REQUIRE(policy >= 0 && policy <= 2, "evs_cache_create: policy %d", policy);
    EVS_REQUIRE(capacity >= 1 && capacity < (1ll << 30), "evs_cache_create: capacity %lld", (long long)capacity);
    EVS_REQUIRE(n_tables >= 1 && n_tables <= kMaxTables, "evs_cache_create: n_tables %d (max %d)", n_tables, kMaxTables);
    EVS_REQUIRE(codec == 32 || codec == 16 || codec == 8 || codec == 4, "evs_cache_create: codec %d", codec);
    EVS_REQUIRE(dim >= 1 && (codec != 4 || dim % 2 == 0), "evs_cache_create: dim %d", dim);
    evs_cache *c = new evs_cache();
    CacheState &h = c->host;
    h = CacheState{};
    h.cap = (int)capacity; h.n_tables = n_tables; h.dim = dim; h.codec = codec; h.row_bytes = dim * codec / 8;
    h.policy = policy;
    long long nslot = 16;
    while (nslot < capacity * 2 + 8) nslot <<= 1;
    c->nslot = nslot;
    // The batched path keeps its own hash, twice as sparse (load <= 0.25 instead of <= 0.5) while that still fits the 24-bit
    // slot hints: deletions leave tombstones, and how often they have to be swept / the table rebuilt (~170 us at 3.4 M
    // entries) is a matter of tombstones per slot -- 16 us per batch amortised at load 0.4, a third of that here.
    c->bnslot = (nslot * 2 <= (1ll << 24)) ? nslot * 2 : nslot;
    if (const char *e = getenv("EVS_CACHE_HASH_SCALE")) { if (e[0] == '1') c->bnslot = nslot; }   // developer A/B: the denser table
    h.nslot_mask = (unsigned long long)(nslot - 1);
    h.min_c1 = 0; h.n_perfect = 0;
    h.max_perfect = (int)(capacity * perfect_item_cap);          // EvLFU_C1.py:30
    h.flush_n = (int)(flush_rate * capacity) + (flush_extra ? 1 : 0);  // :40 (+1) / evlfu_8.cpp:256 (+0)
    h.perfect_mode = perfect_mode;
    h.n_free = (int)capacity; h.least_freq = 1;
    for (int b = 0; b < kMaxBuckets; b++) { h.head[b] = -1; h.tail[b] = -1; h.len[b] = 0; }
    c->a.lfu_max_freq = policy == kLFU ? (1ll << 22) : 1;
    c->a.packed = capacity <= (1ll << 25) ? 1 : 0;
    {
        SlabPlan sp;
        sp.add(&c->st, sizeof(CacheState));
        sp.add(&c->a.keys, nslot * 8);
        sp.add(&c->a.slot_entry, nslot * 4);
        sp.add(&c->a.ekey, capacity * 8);
        sp.add(&c->a.eagg, capacity * 4);
        sp.add(&c->a.efreq, capacity * 8);
        sp.add(&c->a.prev, capacity * 4);
        sp.add(&c->a.next, capacity * 4);
        sp.add(&c->a.free_stack, capacity * 4);
        sp.add(&c->a.lfu_head, c->a.lfu_max_freq * 4);
        sp.add(&c->a.lfu_tail, c->a.lfu_max_freq * 4);
        sp.add(&c->a.lfu_len, c->a.lfu_max_freq * 4);
        if (!sp.carve(&c->slab_create)) {
            set_error("evs_cache_create: hipMalloc(%lld bytes) failed", (long long)sp.total());
            evs_cache_destroy(c);
            return EVS_ENOMEM;
        }
        // the row arena is an allocation of its own: a set-associative tier that installs its two-copy arena (sa_alloc) gives
        // this one back instead of holding three rows per entry
        if (hipMalloc(reinterpret_cast<void **>(&c->arena_create), (size_t)capacity * (size_t)h.row_bytes) != hipSuccess) {
            (void)hipGetLastError();
            set_error("evs_cache_create: hipMalloc(%lld bytes) for the row arena failed", (long long)capacity * h.row_bytes);
            evs_cache_destroy(c);
            return EVS_ENOMEM;
        }
        c->a.arena = c->arena_create;
    }
    EVS_HIP_CHECK(hipMemset(c->a.keys, 0, nslot * 8));
    EVS_HIP_CHECK(hipMemset(c->a.ekey, 0, capacity * 8));
    EVS_HIP_CHECK(hipMemset(c->a.lfu_head, 0xff, c->a.lfu_max_freq * 4));
    EVS_HIP_CHECK(hipMemset(c->a.lfu_tail, 0xff, c->a.lfu_max_freq * 4));
    EVS_HIP_CHECK(hipMemset(c->a.lfu_len, 0, c->a.lfu_max_freq * 4));
    std::vector<int> fs(capacity);
    for (int64_t i = 0; i < capacity; i++) fs[i] = (int)(capacity - 1 - i);  // pops hand out 0,1,2,...
    EVS_HIP_CHECK(hipMemcpy(c->a.free_stack, fs.data(), capacity * 4, hipMemcpyHostToDevice));
    EVS_HIP_CHECK(hipMemcpy(c->st, &h, sizeof h, hipMemcpyHostToDevice));
    *out = c;
    return EVS_OK;
}

// The one-launch form of a set-associative tier (the policy update INSIDE the probe + interaction launch, hit flag = "served
// from the cache"): on = 1 (the default; EVS_CACHE_INLINE=0 changes the default), 0 = the two-launch chain with strict
// snapshot flags.  May be switched between batches.
extern "C" int evs_cache_set_inline_update(evs_cache *c, int on) {
    using namespace evs;
    EVS_REQUIRE(c && (on == 0 || on == 1), "evs_cache_set_inline_update: bad argument");
    c->inline_mode = on;
    return EVS_OK;
}

static int serve_pause(evs_cache *c);
static int exact_launched(evs_cache *c, hipStream_t st);
extern "C" int evs_cache_set_backing(evs_cache *c, const void *const *tables, const int64_t *n_rows) {
    using namespace evs;
    EVS_REQUIRE(c && tables && n_rows, "evs_cache_set_backing: NULL argument");
    { const int prc = serve_pause(c); if (prc) return prc; }   // (a server inside its idle window holds the OLD table pointers)
    for (int k = 0; k < c->host.n_tables; k++) {
        EVS_REQUIRE(tables[k] || n_rows[k] == 0, "evs_cache_set_backing: table %d is NULL", k);
        c->backing[k] = reinterpret_cast<const unsigned char *>(tables[k]);
        c->backing_rows[k] = n_rows[k];
    }
    c->has_backing = true;
    // where the miss tier lives decides the order of the batched pipeline (see cache_batch_impl)
    c->host_backing = false;
    for (int k = 0; k < c->host.n_tables && !c->host_backing; k++) {
        if (!tables[k]) continue;
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, tables[k]) == hipSuccess) c->host_backing = attr.type == hipMemoryTypeHost;
        else (void)hipGetLastError();
    }
    return EVS_OK;
}

// The miss tier is a set of ev-table-N.bin files (evs_filetier_open): registered tables are read by the kernels over the
// bus like pinned host tables; the others ("staged") are served through the host's reader pool -- batched lookups only.
extern "C" int evs_cache_set_file_backing(evs_cache *c, evs_filetier *ft) {
    using namespace evs;
    EVS_REQUIRE(c && ft, "evs_cache_set_file_backing: NULL argument");
    { const int prc = serve_pause(c); if (prc) return prc; }
    EVS_REQUIRE(filetier_tables(ft) == c->host.n_tables, "evs_cache_set_file_backing: the tier has %d tables, the cache %d",
                filetier_tables(ft), c->host.n_tables);
    EVS_REQUIRE(filetier_row_bytes(ft) == c->host.row_bytes, "evs_cache_set_file_backing: row size %lld, the cache's rows have %d bytes",
                filetier_row_bytes(ft), c->host.row_bytes);
    c->staged_mask = 0;
    for (int k = 0; k < c->host.n_tables; k++) {
        c->backing[k] = reinterpret_cast<const unsigned char *>(filetier_dev(ft, k));
        c->backing_rows[k] = filetier_rows(ft, k);
        if (!c->backing[k] && c->backing_rows[k] > 0) c->staged_mask |= 1u << k;
    }
    c->ft = ft;
    c->has_backing = true;
    c->host_backing = true;
    return EVS_OK;
}

extern "C" int64_t evs_cache_staged_rows(evs_cache *c) { return c ? c->n_staged_rows : 0; }

extern "C" int evs_cache_set_batch_policy(evs_cache *c, int policy) {
    using namespace evs;
    EVS_REQUIRE(c && policy >= 0 && policy <= 2, "evs_cache_set_batch_policy: bad argument");
    EVS_REQUIRE(policy != 2 || c->host.cap >= (long long)kSaSingleWays, "evs_cache_set_batch_policy: the set-associative policy needs a capacity of at least %d entries", (int)kSaSingleWays);
    if (c->used == 2) { set_error("evs_cache_set_batch_policy: the batched path is already in use"); return EVS_ESTATE; }
    c->batch_policy = policy;
    return EVS_OK;
}

static int serve_pause(evs_cache *c);
extern "C" int evs_cache_request(evs_cache *c, int64_t B, const int32_t *rows, float *out, uint8_t *hit,
                                 int approx_thres, void *stream) {
    using namespace evs;
    EVS_REQUIRE(c, "evs_cache_request: NULL cache");
    if (!c->has_backing) { set_error("evs_cache_request: call evs_cache_set_backing first"); return EVS_ESTATE; }
    if (c->staged_mask) { set_error("evs_cache_request: a file-backed tier with staged tables serves batched lookups only"); return EVS_ESTATE; }
    if (B == 0) return EVS_OK;
    EVS_REQUIRE(B > 0 && rows && out && hit, "evs_cache_request: NULL argument");
    CacheArgs args;
    args.st = c->st; args.a = c->a;
    for (int k = 0; k < kMaxTables; k++) { args.backing[k] = c->backing[k]; args.backing_rows[k] = c->backing_rows[k]; }
    args.requests = rows; args.out = out; args.hit = hit; args.B = B; args.approx_thres = approx_thres;
    if (c->used == 2) { set_error("evs_cache_request: this cache is used through the batched path"); return EVS_ESTATE; }
    c->used = 1;
    { const int prc = serve_pause(c); if (prc) return prc; }
    hipLaunchKernelGGL(cache_exact_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), args);
    EVS_HIP_CHECK(hipGetLastError());
    return exact_launched(c, reinterpret_cast<hipStream_t>(stream));
}

// ---- the exact policy as a resident server (cache_serve_kernel) ------------------------------------------------------------
static void serve_launch(evs_cache *c) {
    using namespace evs;
    CacheArgs args;
    args.st = c->st; args.a = c->a;
    for (int k = 0; k < kMaxTables; k++) { args.backing[k] = c->backing[k]; args.backing_rows[k] = c->backing_rows[k]; }
    args.requests = nullptr; args.out = nullptr; args.hit = nullptr; args.B = 0; args.approx_thres = c->serve_thres;
    ServeArgs sv;
    sv.req = c->req_dev; sv.ctl = c->req_dev + 32; sv.ans = c->mbox_dev + 64;
    sv.ring = c->serve_ring; sv.n_slots = c->serve_slots; sv.idle_ticks = c->serve_idle_ticks;
    // an exact-path launch of the caller's (evs_cache_request, evs_cache_request_c1c2[c3]) may still be running on ITS stream:
    // the server starts behind it
    if (c->exact_pending) { (void)hipStreamWaitEvent(c->serve_stream, c->exact_done, 0); c->exact_pending = false; }
    hipLaunchKernelGGL(cache_serve_kernel, dim3(1), dim3(64), 0, c->serve_stream, args, sv);
}
// behind every exact-path launch: where a server started later has to wait (only caches that have a server pay the record)
static int exact_launched(evs_cache *c, hipStream_t st) {
    if (!c->serve_stream) return EVS_OK;
    if (!c->exact_done) EVS_HIP_CHECK(hipEventCreateWithFlags(&c->exact_done, hipEventDisableTiming));
    EVS_HIP_CHECK(hipEventRecord(c->exact_done, st));
    c->exact_pending = true;
    return EVS_OK;
}
// anything else that reads or writes the exact state first sends the server home (it writes the state back on its way out);
// the next evs_cache_serve_request starts it again
static int serve_pause(evs_cache *c) {
    if (!c || !c->mbox || !c->serve_stream) return EVS_OK;
    if (hipStreamQuery(c->serve_stream) == hipSuccess) return EVS_OK;   // nothing running
    (void)hipGetLastError();
    c->req_host[32] = 1u;
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    const hipError_t e = hipStreamSynchronize(c->serve_stream);
    c->req_host[32] = 0u;
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    return e == hipSuccess ? EVS_OK : EVS_EHIP;
}
extern "C" int evs_cache_serve_start(evs_cache *c, int approx_thres, float *ring, int n_slots, int64_t idle_us) {
    using namespace evs;
    EVS_REQUIRE(c && ring && n_slots >= 1 && idle_us >= 1, "evs_cache_serve_start: bad argument");
    if (!c->has_backing) { set_error("evs_cache_serve_start: call evs_cache_set_backing first"); return EVS_ESTATE; }
    if (c->staged_mask) { set_error("evs_cache_serve_start: a file-backed tier with staged tables serves batched lookups only"); return EVS_ESTATE; }
    if (c->used == 2) { set_error("evs_cache_serve_start: this cache is used through the batched path"); return EVS_ESTATE; }
    EVS_REQUIRE(c->host.n_tables <= 28, "evs_cache_serve_start: at most 28 tables (the request line holds 4 x 7 ids)");
    c->used = 1;
    if (!c->mbox) {
        EVS_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&c->mbox), 3 * 128, hipHostMallocMapped));
        memset(c->mbox, 0, 3 * 128);
        EVS_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void **>(&c->mbox_dev), c->mbox, 0));
        c->req_host = c->mbox; c->req_dev = c->mbox_dev;
        // a stream of the highest priority: streams share a handful of hardware queues, round-robin per priority level, and a
        // queue runs its commands in order -- a copy or a kernel of the caller's that lands on the queue of the resident server
        // waits until the server goes home idle (measured through the plug-in loop: 299 us per request instead of 104 once a
        // process had made a few more streams).  Nothing else of this library uses the level.
        int prio_lo = 0, prio_hi = 0;
        EVS_HIP_CHECK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        EVS_HIP_CHECK(hipStreamCreateWithPriority(&c->serve_stream, hipStreamNonBlocking, prio_hi));
        c->serve_seq = 0;
    } else {
        const int rc = serve_pause(c);
        if (rc) return rc;
    }
    for (size_t k = 0; k < c->slot_busy.size(); k++)      // (a new ring: whoever still reads the old one finishes first)
        if (c->slot_busy[k]) { (void)hipEventSynchronize(c->slot_done[k]); c->slot_busy[k] = 0; }
    c->slot_done.resize((size_t)n_slots, nullptr); c->slot_busy.assign((size_t)n_slots, 0);
    c->serve_ring = ring; c->serve_slots = n_slots; c->serve_thres = approx_thres;
    c->serve_idle_ticks = idle_us * 100;   // wall_clock64(): 100 MHz
    c->serving = true;
    EVS_HIP_CHECK(hipDeviceSynchronize());   // (the ring and the tables' contents are in place before the server's first read)
    return EVS_OK;
}
// one request (T row ids on the host) -> T hit flags on the host, the T x d fp32 rows in slot *slot_out of the ring (device).
// Blocks until the server has answered (no launch, no copy, no synchronise: two cache-line hand-overs over the bus).
static int serve_request_impl(evs_cache *c, const int32_t *rows, const int64_t *ids_dev, int64_t ids_stride, float *out_dev, uint8_t *hit, int *slot_out);
extern "C" int evs_cache_serve_request(evs_cache *c, const int32_t *rows, uint8_t *hit, int *slot_out) {
    using namespace evs;
    EVS_REQUIRE(c && rows && hit && slot_out, "evs_cache_serve_request: NULL argument");
    return serve_request_impl(c, rows, nullptr, 0, nullptr, hit, slot_out);
}
// the same request with the rows delivered to a DEVICE buffer of the caller's (T x d floats) instead of a ring slot: no copy out
// of the ring, nothing to hand back (evs_cache_serve_consumed).  Exactly one of rows (host) / ids_dev (device, as
// evs_cache_serve_request_dev) is given; at most 26 tables.  Nothing orders the server's stores against the caller's streams:
// out_dev must not be in use by work still pending when this is called (a freshly allocated block of a stream-ordered
// allocator counts as in use until that stream is idle), and is complete when the call returns.
extern "C" int evs_cache_serve_request_to(evs_cache *c, const int32_t *rows, const int64_t *ids_dev, int64_t ids_stride, float *out_dev, uint8_t *hit) {
    using namespace evs;
    EVS_REQUIRE(c && out_dev && hit && ((rows != nullptr) != (ids_dev != nullptr)) && ids_stride >= 0 && ids_stride < (1ll << 32),
                "evs_cache_serve_request_to: bad argument");
    EVS_REQUIRE(c->host.n_tables <= 26, "evs_cache_serve_request_to: at most 26 tables (the line's last two id words carry the address)");
    int slot = 0;
    return serve_request_impl(c, rows, ids_dev, ids_stride, out_dev, hit, &slot);
}
// the same request with the T row ids given by ADDRESS: ids_dev[t * ids_stride] (int64, device memory: the (T, B) lS_i of the
// reference's loop as dlrm_wrap left it on the device, element 0 of each row, dlrm_s_pytorch_C1.py:236-239) -- the server reads
// them itself.  The ids must be COMPLETE when this is called (no stream orders the mailbox).
extern "C" int evs_cache_serve_request_dev(evs_cache *c, const int64_t *ids_dev, int64_t ids_stride, uint8_t *hit, int *slot_out) {
    using namespace evs;
    EVS_REQUIRE(c && ids_dev && hit && slot_out && ids_stride >= 0 && ids_stride < (1ll << 32), "evs_cache_serve_request_dev: bad argument");
    return serve_request_impl(c, nullptr, ids_dev, ids_stride, nullptr, hit, slot_out);
}
static int serve_request_impl(evs_cache *c, const int32_t *rows, const int64_t *ids_dev, int64_t ids_stride, float *out_dev, uint8_t *hit, int *slot_out) {
    using namespace evs;
    if (!c->serving) { set_error("evs_cache_serve_request: call evs_cache_serve_start first"); return EVS_ESTATE; }
    volatile unsigned *req = c->req_host, *ans = c->mbox + 64;
    const int T = c->host.n_tables;
    const unsigned want = (c->serve_seq + 1u) & 0x7fffffffu;
    // the ring slot this request's rows go to: whoever still READS it (evs_cache_serve_consumed) finishes first
    const int slot = (int)(want % (unsigned)c->serve_slots);
    if (!out_dev && c->slot_busy[(size_t)slot]) {
        EVS_HIP_CHECK(hipEventSynchronize(c->slot_done[(size_t)slot]));
        c->slot_busy[(size_t)slot] = 0;
    }
    c->serve_seq = want;
    const unsigned long long oa = (unsigned long long)reinterpret_cast<uintptr_t>(out_dev);
    // seven ids and a guard per 32-byte sector: whatever granularity the bus delivers the line in, ids are accepted only from
    // a sector whose guard (written after them) holds the number awaited
    unsigned guard = want;
    if (rows) {
        for (int t = 0; t < T; t++) req[t + t / 7] = (unsigned)rows[t];
        if (T <= 26) { req[29] = (unsigned)oa; req[30] = (unsigned)(oa >> 32); }   // (the id words of tables 26, 27: where the rows go, 0 = the ring)
    } else {   // by address: words 0..2 (3, 4: where the rows go), and bit 31 of the guards says so
        const unsigned long long pa = (unsigned long long)reinterpret_cast<uintptr_t>(ids_dev);
        req[0] = (unsigned)pa; req[1] = (unsigned)(pa >> 32); req[2] = (unsigned)ids_stride;
        req[3] = (unsigned)oa; req[4] = (unsigned)(oa >> 32);
        guard |= 0x80000000u;
    }
    __atomic_thread_fence(__ATOMIC_RELEASE);
    req[7] = guard; req[15] = guard; req[23] = guard; req[31] = guard;
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    long long spins = 0;
    // a request that fails leaves host and device agreeing on the sequence number: the one the device last ANSWERED
    auto fail = [&](int rc) { c->serve_seq = ans[16]; return rc; };
    while (ans[16] != want) {
        if ((spins & 63) == 0 && ans[17] == 0u) {   // no server (never started, went home idle, or was paused): start one
            const hipError_t q = hipStreamQuery(c->serve_stream);
            if (q == hipSuccess) {
                serve_launch(c);
                if (hipGetLastError() != hipSuccess) { set_error("evs_cache_serve_request: the server could not be started"); return fail(EVS_EHIP); }
            }
            else if (q != hipErrorNotReady) { (void)hipGetLastError(); return fail(EVS_EHIP); }
            else (void)hipGetLastError();
        }
        if (++spins > (1ll << 31)) {
            // send a server that may still be there home (it re-reads the answer line's number when it starts again), then agree
            (void)serve_pause(c);
            set_error("evs_cache_serve_request: the server did not answer");
            return fail(EVS_EHIP);
        }
        __builtin_ia32_pause();
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    const volatile unsigned char *fl = reinterpret_cast<const volatile unsigned char *>(ans);
    for (int t = 0; t < T; t++) hit[t] = fl[t];
    *slot_out = out_dev ? -1 : slot;
    return EVS_OK;
}
// The caller has ENQUEUED its reads of ring slot `slot` (a copy, a kernel) on `stream`: the slot is handed out again only
// after they have finished.  Without this call the contract is stream order, not host order: the rows of a slot are valid
// until n_slots - 1 more requests have been POSTED, whatever the caller's stream has executed by then.
extern "C" int evs_cache_serve_consumed(evs_cache *c, int slot, void *stream) {
    using namespace evs;
    EVS_REQUIRE(c && c->serving && slot >= 0 && slot < c->serve_slots, "evs_cache_serve_consumed: bad argument");
    hipEvent_t &e = c->slot_done[(size_t)slot];
    if (!e) EVS_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    EVS_HIP_CHECK(hipEventRecord(e, reinterpret_cast<hipStream_t>(stream)));
    c->slot_busy[(size_t)slot] = 1;
    return EVS_OK;
}
#ifdef EVS_X_EXACT_TIMING
extern "C" __attribute__((visibility("default"))) int evs_x_exact_ticks(long long *out8, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(evs::g_exact_ticks), 64) != hipSuccess) return -2;
    if (reset) { long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(evs::g_exact_ticks), z, 64) != hipSuccess) return -3; }
    return 0;
}
#endif
extern "C" int evs_cache_serve_stop(evs_cache *c) {
    using namespace evs;
    EVS_REQUIRE(c, "evs_cache_serve_stop: NULL cache");
    const int rc = serve_pause(c);
    c->serving = false;
    return rc;
}

// out8: [min_C1, n_perfect, size, n_flush, n_evict, n_requests, n_perfect_hits, n_hits]; returns the sticky error
extern "C" int evs_cache_stats(evs_cache *c, int64_t *out8, void *stream) {
    using namespace evs;
    EVS_REQUIRE(c && out8, "evs_cache_stats: NULL argument");
    { const int prc = serve_pause(c); if (prc) return prc; }
    CacheState h;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    EVS_HIP_CHECK(hipMemcpyAsync(&h, c->st, sizeof h, hipMemcpyDeviceToHost, st));
    EVS_HIP_CHECK(hipStreamSynchronize(st));
    out8[0] = h.min_c1; out8[1] = h.n_perfect; out8[2] = h.count; out8[3] = h.n_flush; out8[4] = h.n_evict;
    out8[5] = h.n_requests; out8[6] = h.n_perfect_hits; out8[7] = h.n_hits;
    if (h.error) {
        set_error("cache policy error %d (1: flush popped an empty bucket, 2: internal, 3: LFU frequency overflow)", h.error);
        return EVS_ESTATE;
    }
    return EVS_OK;
}

extern "C" int evs_cache_reset_counters(evs_cache *c, void *stream) {
    using namespace evs;
    EVS_REQUIRE(c, "evs_cache_reset_counters: NULL cache");
    { const int prc = serve_pause(c); if (prc) return prc; }
    CacheState h;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    EVS_HIP_CHECK(hipMemcpyAsync(&h, c->st, sizeof h, hipMemcpyDeviceToHost, st));
    EVS_HIP_CHECK(hipStreamSynchronize(st));
    h.n_requests = 0; h.n_perfect_hits = 0; h.n_hits = 0;
    EVS_HIP_CHECK(hipMemcpyAsync(c->st, &h, sizeof h, hipMemcpyHostToDevice, st));
    EVS_HIP_CHECK(hipStreamSynchronize(st));
    return EVS_OK;
}

// Resident keys in list order: triples (bucket|freq|0, table_1based, row); returns the count (may exceed max).
extern "C" int64_t evs_cache_dump(evs_cache *c, int64_t *triples, int64_t max_triples, void *stream) {
    using namespace evs;
    if (!c) return EVS_EINVAL;
    if (serve_pause(c)) return EVS_EHIP;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (hipStreamSynchronize(st) != hipSuccess) return EVS_EHIP;
    CacheState h;
    const int64_t cap = c->host.cap;
    std::vector<unsigned long long> ekey(cap);
    std::vector<int> next(cap);
    if (hipMemcpy(&h, c->st, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return EVS_EHIP;
    if (hipMemcpy(ekey.data(), c->a.ekey, cap * 8, hipMemcpyDeviceToHost) != hipSuccess) return EVS_EHIP;
    if (hipMemcpy(next.data(), c->a.next, cap * 4, hipMemcpyDeviceToHost) != hipSuccess) return EVS_EHIP;
    int64_t n = 0;
    auto walk = [&](int head, int64_t tag) {
        for (int e = head; e >= 0; e = next[e]) {
            if (n < max_triples && triples) {
                triples[3 * n] = tag;
                triples[3 * n + 1] = (int64_t)(ekey[e] >> 32);
                triples[3 * n + 2] = (int64_t)(ekey[e] & 0xffffffffull);
            }
            n++;
            if (n > cap + 1) return;  // corrupted list guard
        }
    };
    if (c->host.policy == kLFU) {
        const int64_t F = c->a.lfu_max_freq;
        std::vector<int> heads(F);
        if (hipMemcpy(heads.data(), c->a.lfu_head, F * 4, hipMemcpyDeviceToHost) != hipSuccess) return EVS_EHIP;
        for (int64_t f = 1; f < F && n < h.count; f++) walk(heads[f], f);
    } else {
        for (int b = 0; b <= c->host.n_tables; b++) walk(h.head[b], b);
    }
    return n;
}

extern "C" int evs_cache_request_c1c2c3(evs_cache *c1, evs_cache *c2, evs_aprx *c3, int64_t B, const int32_t *rows,
                                        float *out, uint8_t *tier, int high_agghit_threshold, void *stream);

extern "C" int evs_cache_request_c1c2(evs_cache *c1, evs_cache *c2, int64_t B, const int32_t *rows, float *out,
                                      uint8_t *tier, int high_agghit_threshold, void *stream) {
    return evs_cache_request_c1c2c3(c1, c2, nullptr, B, rows, out, tier, high_agghit_threshold, stream);
}

// ---- batched path --------------------------------------------------------------------------
// allocate / grow the batched-path state of one cache and describe it in `a` (everything but the per-call outputs)
static int resolved_batch_policy(evs_cache *c, bool single_tier = true);
static int batch_prepare(evs_cache *c, int64_t B, const int32_t *rows, hipStream_t st, evs::BatchArgs &a, const char *who) {
    using namespace evs;
    EVS_REQUIRE(c, "%s: NULL cache", who);
    EVS_REQUIRE(c->host.policy == kEvLFU, "%s: EvLFU only", who);
    EVS_REQUIRE(c->host.n_tables <= 32, "%s: at most 32 tables", who);
    if (!c->has_backing) { set_error("%s: call evs_cache_set_backing first", who); return EVS_ESTATE; }
    if (c->used == 1) { set_error("%s: this cache is used through the exact path", who); return EVS_ESTATE; }
    EVS_REQUIRE(c->host.cap <= kMaxBatchedCap, "%s: capacity above %lld entries needs wider hash words", who, kMaxBatchedCap);
    EVS_REQUIRE(B > 0 && B < (1ll << 31) / 32 && rows, "%s: bad argument", who);
    const int T = c->host.n_tables;
    const long long cap = c->host.cap;
    const bool sa = resolved_batch_policy(c) == 2;   // set-associative policy: no hash, no entry arrays (the set records: c->sa)
    if (sa && !c->sa.tags) {   // (a tier that starts out in a pair got its geometry there: batch_c1c2_impl)
        if (!sa_single_feasible(c, &c->sau, &c->sa)) {
            set_error("%s: the set-associative batch policy needs fewer than 2^32 rows over all tables and a capacity of at least rows / 2^19 entries", who);
            return EVS_EINVAL;
        }
        const int arc = sa_alloc(c, nullptr, st);
        if (arc) { set_error("%s: allocating the set records failed", who); return arc; }
    }
    const long long hash_words = sa ? 1 : c->bnslot;
    if (!c->bs) {   // first batched call: all-or-nothing, so a failed allocation leaves the cache as it was
        BatchState h{};
        h.n_free = (int)cap;
        BatchState *bs = nullptr;
        int *eslot = nullptr, *part1 = nullptr, *part2 = nullptr, *host_tomb = nullptr, *host_tomb_dev = nullptr;
        unsigned long long *bslots = nullptr;
        void *slab = nullptr;
        SlabPlan sp;
        sp.add(&bs, sizeof(BatchState)); sp.add(&eslot, sa ? 4 : cap * 4); sp.add(&bslots, hash_words * 8);
        sp.add(&part1, kReplicas * kPartCols * 4); sp.add(&part2, kReplicas * kPartCols * 4);
        const bool ok =
            sp.carve(&slab) && hipMemcpy(bs, &h, sizeof h, hipMemcpyHostToDevice) == hipSuccess &&   // blocking copy
            hipMemsetAsync(bslots, 0, hash_words * 8, st) == hipSuccess &&                               // ordered on the caller's stream
            hipMemsetAsync(part1, 0, kReplicas * kPartCols * 4, st) == hipSuccess &&
            hipMemsetAsync(part2, 0, kReplicas * kPartCols * 4, st) == hipSuccess &&
            hipHostMalloc(reinterpret_cast<void **>(&host_tomb), 4 * sizeof(int), hipHostMallocMapped) == hipSuccess &&   // [0] tombstones, [1] flush wanted, [2] which close reported [0]
            hipHostGetDevicePointer(reinterpret_cast<void **>(&host_tomb_dev), host_tomb, 0) == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();
            if (slab) (void)hipFree(slab);
            if (host_tomb) (void)hipHostFree(host_tomb);
            set_error("%s: allocating the batched-path state failed (capacity %lld)", who, cap);
            return EVS_ENOMEM;
        }
        host_tomb[0] = 0; host_tomb[1] = 0; host_tomb[2] = 0; host_tomb[3] = 0;
        c->slab_batch = slab;
        c->bs = bs; c->eslot = eslot; c->bslots = bslots; c->part1 = part1; c->part2 = part2;
        c->host_tomb = host_tomb; c->host_tomb_dev = host_tomb_dev;
    }
    const long long g2 = (B * T + 255) / 256;
    if (B > c->max_batch) {   // grow the per-batch buffers: the new slab first, the old one goes only when the new one exists
        unsigned *miss_info = nullptr;
        int *new_slot = nullptr, *block_cnt = nullptr, *block_base = nullptr;
        long long *row_ptrs = nullptr, *iota = nullptr;
        unsigned char *row_tier = nullptr;
        uint4 *miss_rec = nullptr; int *list_cnt = nullptr;
        void *slab = nullptr;
        SlabPlan sp;
        long long g1n = (B + 7) / 8; if (g1n > kProbeGridMax) g1n = kProbeGridMax;
        const long long iters = (B + 8 * g1n - 1) / (8 * g1n);
        sp.add(&miss_info, B * T * 4); sp.add(&new_slot, g2 * 256 * 4); sp.add(&block_cnt, g2 * 4); sp.add(&block_base, g2 * 4);
        sp.add(&row_ptrs, B * T * 8); sp.add(&row_tier, B * T); sp.add(&iota, B * 8);
        sp.add(&miss_rec, g1n * iters * 8 * T * 16);   // K1's per-block miss lists (sampled update)
        // one counter per list: K1 has g1n lists, the probe folded into the consumer one per 16-sample block -- more than
        // g1n once B > 16 * kProbeGridMax (the fold is bounded by rf_max_batch(), which EVS_FUSED_RF_MAX_B can raise)
        sp.add(&list_cnt, std::max<long long>(g1n, (B + 15) / 16) * 4);
        if (!sp.carve(&slab)) {
            set_error("%s: allocating the buffers of a %lld-request batch failed", who, (long long)B);
            return EVS_ENOMEM;
        }
        if (c->slab_perbatch) {
            EVS_HIP_CHECK(hipStreamSynchronize(st));
            (void)hipFree(c->slab_perbatch);
        }
        c->slab_perbatch = slab;
        c->miss_info = miss_info; c->new_slot = new_slot; c->block_cnt = block_cnt; c->block_base = block_base;
        c->row_ptrs = row_ptrs; c->row_tier = row_tier; c->iota = iota; c->miss_rec = miss_rec; c->list_cnt = list_cnt;
        hipLaunchKernelGGL(iota_kernel, dim3(256), dim3(256), 0, st, c->iota, (long long)B);
        c->max_batch = B;
    }
    c->used = 2;
    a.bs = c->bs; a.a = c->a; a.eslot = c->eslot; a.slots = c->bslots;
    a.miss_info = c->miss_info; a.new_slot = c->new_slot;
    a.hint_shift = 0;
    while (!sa && (c->bnslot >> a.hint_shift) > (1ll << 24)) a.hint_shift++;   // (set-associative policy: the records carry the whole set index)
    a.row_ptrs = c->row_ptrs; a.row_ids = nullptr;
    for (int k = 0; k < kMaxTables; k++) { a.backing[k] = c->backing[k]; a.backing_rows[k] = c->backing_rows[k]; }
    a.requests = rows; a.out = nullptr; a.hit = nullptr; a.B = B; a.mask = (unsigned long long)(c->bnslot - 1);
    a.cap = (int)cap; a.T = T; a.d = c->host.dim; a.codec = c->host.codec; a.row_bytes = c->host.row_bytes;
    a.max_perfect = c->host.max_perfect; a.flush_n = c->host.flush_n; a.nslot = (int)c->bnslot;
    a.block_cnt = c->block_cnt; a.block_base = c->block_base; a.part1 = c->part1; a.part2 = c->part2;
    long long g1 = (B + 7) / 8; if (g1 > kProbeGridMax) g1 = kProbeGridMax;
    a.g1 = (int)g1; a.g2 = (int)g2;
    a.host_tomb = c->host_tomb_dev;
    a.estamp = nullptr; a.stamp = 0; a.stamp_hits = 0;
    a.other_slots = nullptr; a.other_mask = 0; a.route_filter = nullptr; a.route_mask = 0; a.route_stamp = 0;
    a.staged_mask = 0; a.staging = nullptr; a.new_keys = nullptr; a.slot_stage = nullptr; a.stage_rows = 0;
    // Tombstone housekeeping, from the count the close of an EARLIER batch left in mapped host memory (no synchronisation:
    // a few batches late is as good): past nslot / 8 the
    // table is swept (tombstones at the end of their chains become empty again); if the count is back there within
    // three calls the sweep did not help enough and the hash is rebuilt.
    a.evicted_keys = nullptr; a.vict_cnt = nullptr; a.vict_other = nullptr; a.vict_cap = 0; a.c3_tags = nullptr; a.c3_nset = 0; a.c3_stat = nullptr; a.miss_rec = nullptr; a.list_cnt = nullptr; a.list_cap = 0;
    a.tomb_parity = -1;
    a.sa = c->sa; a.sau = c->sau;
    if (!sa) a.sa.tags = nullptr;
    a.rebuild = 0;
    c->batch_calls++;   // = the ordinal of this call's close
    if (!sa) {
        volatile int *hk = reinterpret_cast<volatile int *>(c->host_tomb);
        const long long seq = hk[2];
        // (the device runs behind the host: a report is only news if its close came after the last sweep / rebuild)
        if (seq > c->last_hk_call && hk[0] > (int)(c->bnslot / 8)) {
            if (seq - c->last_sweep_call <= 3) a.rebuild = 1;
            else { a.rebuild = 2; c->last_sweep_call = c->batch_calls; }
            c->last_hk_call = c->batch_calls;
        }
    }
    return EVS_OK;
}

// K2..K5 of one cache (its misses are described by a.miss_info)
static void batch_policy_a(evs_cache *c, const evs::BatchArgs &a, hipStream_t st) {   // K2, K3
    using namespace evs;
    hipLaunchKernelGGL(cache_batch_insert_kernel, dim3((unsigned)a.g2), dim3(256), 0, st, a);
    hipLaunchKernelGGL(cache_batch_plan_kernel, dim3(1), dim3(256), 0, st, a);
}
static void batch_policy_b(evs_cache *c, const evs::BatchArgs &a, hipStream_t st) {   // K4, K5
    using namespace evs;
    const int wide = kNumCu * 8;
    long long ne = ((long long)a.cap + 255) / 256; if (ne > wide) ne = wide;
    hipLaunchKernelGGL(cache_batch_evict_kernel, dim3((unsigned)ne), dim3(256), 0, st, a);
    hipLaunchKernelGGL(cache_batch_assign_kernel, dim3((unsigned)a.g2), dim3(256), 0, st, a);
}
static void batch_policy(evs_cache *c, const evs::BatchArgs &a, hipStream_t st) {
    batch_policy_a(c, a, st);
    batch_policy_b(c, a, st);
}

// File mode between K3 and K4: list the batch's new keys, hand them to the host's reader pool, which copies their rows
// out of the file mappings into the pinned staging buffer (row i = new key i).  Two small device-to-host copies and
// one synchronise per batch: this is the capacity tier (tables larger than what may be pinned), not the fast one.
extern "C" int evs_filetier_fetch(evs_filetier *ft, int64_t n, const uint64_t *keys, void *dst, uint32_t skip_mask);
static int batch_stage_prepare(evs_cache *c, evs::BatchArgs &a, hipStream_t st) {   // before K2: buffers and arguments
    using namespace evs;
    const long long want = a.B * a.T;
    if (want > c->staging_rows) {
        EVS_HIP_CHECK(hipStreamSynchronize(st));
        if (c->new_keys) (void)hipFree(c->new_keys);
        if (c->new_keys_host) (void)hipHostFree(c->new_keys_host);
        if (c->staging) (void)hipHostFree(c->staging);
        c->new_keys = nullptr; c->new_keys_host = nullptr; c->staging = nullptr; c->staging_rows = 0;
        EVS_HIP_CHECK(hipMalloc(&c->new_keys, want * 8));
        EVS_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&c->new_keys_host), want * 8 + 8, hipHostMallocDefault));
        EVS_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&c->staging), want * (long long)a.row_bytes, hipHostMallocMapped));
        EVS_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void **>(&c->staging_dev), c->staging, 0));
        c->staging_rows = want;
    }
    if (!c->slot_stage) EVS_HIP_CHECK(hipMalloc(&c->slot_stage, c->bnslot * 4));
    a.staged_mask = c->staged_mask; a.staging = c->staging_dev; a.new_keys = c->new_keys; a.slot_stage = c->slot_stage;
    a.stage_rows = c->staging_rows;
    return EVS_OK;
}
static int batch_stage_rows(evs_cache *c, evs::BatchArgs &a, hipStream_t st) {      // between K3 and K4
    using namespace evs;
    hipLaunchKernelGGL(cache_batch_list_kernel, dim3((unsigned)a.g2), dim3(256), 0, st, a);
    int cnt[2] = {0, 0};
    EVS_HIP_CHECK(hipMemcpyAsync(&cnt[0], &a.bs->n_new, sizeof(int), hipMemcpyDeviceToHost, st));
    EVS_HIP_CHECK(hipMemcpyAsync(&cnt[1], &a.bs->n_orphan, sizeof(int), hipMemcpyDeviceToHost, st));
    EVS_HIP_CHECK(hipStreamSynchronize(st));
    const long long rows = c->staging_rows;
    long long n_new = cnt[0] < rows ? cnt[0] : rows;
    long long n_orph = cnt[1] < rows - n_new ? cnt[1] : rows - n_new;
    if (n_new > 0) EVS_HIP_CHECK(hipMemcpyAsync(c->new_keys_host, c->new_keys, (size_t)n_new * 8, hipMemcpyDeviceToHost, st));
    if (n_orph > 0) EVS_HIP_CHECK(hipMemcpyAsync(c->new_keys_host + (rows - n_orph), c->new_keys + (rows - n_orph), (size_t)n_orph * 8,
                                                 hipMemcpyDeviceToHost, st));
    if (n_new > 0 || n_orph > 0) EVS_HIP_CHECK(hipStreamSynchronize(st));
    uint32_t skip = 0;
    for (int k = 0; k < a.T; k++) if (!((c->staged_mask >> k) & 1u)) skip |= 1u << k;
    const long long lo[2] = {0, rows - n_orph}, len[2] = {n_new, n_orph};
    for (int part = 0; part < 2; part++) {
        if (len[part] <= 0) continue;
        const int rc = evs_filetier_fetch(c->ft, len[part], reinterpret_cast<const uint64_t *>(c->new_keys_host + lo[part]),
                                          c->staging + lo[part] * (long long)a.row_bytes, skip);
        if (rc) return rc;
        for (long long i = 0; i < len[part]; i++) c->n_staged_rows += ((c->staged_mask >> ((int)(c->new_keys_host[lo[part] + i] >> 32) - 1)) & 1u);
    }
    return EVS_OK;
}

// K6: close the batch; the hash is rebuilt without tombstones when they exceed nslot/8 -- the host learns the count
// from a mapped word K6b writes (one or two batches stale, which only means slightly longer probe chains)
static void batch_housekeeping(evs_cache *c, const evs::BatchArgs &a, hipStream_t st) {
    using namespace evs;
    const int wide = kNumCu * 8;
    if (a.rebuild == 1) {   // decided in batch_prepare, so that the close already reports zero tombstones
        long long ne = ((long long)a.cap + 255) / 256; if (ne > wide) ne = wide;
        hipLaunchKernelGGL(cache_batch_clear_kernel, dim3((unsigned)wide), dim3(256), 0, st, a);
        hipLaunchKernelGGL(cache_batch_reinsert_kernel, dim3((unsigned)ne), dim3(256), 0, st, a);
    } else if (a.rebuild == 2) {
        long long nb = (c->bnslot / kSweepRun + 255) / 256; if (nb > wide) nb = wide; if (nb < 1) nb = 1;
        hipLaunchKernelGGL(cache_batch_sweep_kernel, dim3((unsigned)nb), dim3(256), 0, st, c->bs, c->bslots, c->bnslot);
    }
}
static void batch_close(evs_cache *c, evs::BatchArgs a, hipStream_t st) {
    using namespace evs;
    hipLaunchKernelGGL(cache_batch_close_kernel, dim3(1), dim3(256), 0, st, a);
    batch_housekeeping(c, a, st);
}
// threads per miss list (EVS_CACHE_LIST_WAVES overrides: developer A/B)
static unsigned sampled_list_threads(const evs::BatchArgs &a) {
    static int w = -1;
    if (w < 0) { const char *e = getenv("EVS_CACHE_LIST_WAVES"); w = e ? atoi(e) : 0; }
    if (w >= 1 && w <= 4) return 64u * (unsigned)w;
    return a.list_cap > 8 * a.T ? 128u : 64u;
}
// the sampled update, compiled per row size (row_bytes = PIECES pieces of 16 / 8 / 4 / 2 bytes, at most 16 of them)
template <int PIECES, typename U, typename TAIL = evs::NoTail>
static void launch_sampled_update_t(const evs::BatchArgs &a, hipStream_t st) {
    using namespace evs;
    // K1 listed the misses: one wave per list (two when the lists are those of the folded probe's 16-sample blocks: twice as long)
    if (a.miss_rec) hipLaunchKernelGGL((cache_batch_sampled_list_kernel<PIECES, U, TAIL>), dim3((unsigned)a.g1), dim3(sampled_list_threads(a)), 0, st, a);
    else hipLaunchKernelGGL((cache_batch_sampled_kernel<PIECES, U, TAIL>), dim3((unsigned)a.g2), dim3(256), 0, st, a);
}
static void launch_sampled_update(const evs::BatchArgs &a, hipStream_t st) {
    switch (a.row_bytes) {
    case 144: launch_sampled_update_t<9, float4>(a, st); break;     // d = 36 fp32
    case 256: launch_sampled_update_t<16, float4>(a, st); break;    // d = 64 fp32
    case 128: launch_sampled_update_t<8, float4>(a, st); break;     // d = 32 fp32, d = 64 u16
    case 64: launch_sampled_update_t<4, float4>(a, st); break;      // d = 16 fp32, d = 32 u16, d = 64 u8
    case 32: launch_sampled_update_t<2, float4>(a, st); break;
    case 16: launch_sampled_update_t<1, float4>(a, st); break;
    case 72: launch_sampled_update_t<4, float4, uint2>(a, st); break;            // d = 36 u16: 4 x 16 + 8
    case 36: launch_sampled_update_t<2, float4, unsigned>(a, st); break;         // d = 36 u8: 2 x 16 + 4
    case 18: launch_sampled_update_t<1, float4, unsigned short>(a, st); break;   // d = 36 u4: 16 + 2
    case 8: launch_sampled_update_t<1, uint2>(a, st); break;
    default: launch_sampled_update_t<0, float4>(a, st); break;
    }
}

// the set-associative policy's update (always the list form), compiled per row size like the sampled one
template <int PIECES, typename U, typename TAIL = evs::NoTail>
static void launch_sa_update_t(const evs::BatchArgs &a, hipStream_t st) {
    hipLaunchKernelGGL((evs::cache_batch_sa_list_kernel<PIECES, U, TAIL>), dim3((unsigned)a.g1), dim3(sampled_list_threads(a)), 0, st, a);
}
static void launch_sa_update(const evs::BatchArgs &a, hipStream_t st) {
    switch (a.row_bytes) {
    case 144: launch_sa_update_t<9, float4>(a, st); break;
    case 256: launch_sa_update_t<16, float4>(a, st); break;
    case 128: launch_sa_update_t<8, float4>(a, st); break;
    case 64: launch_sa_update_t<4, float4>(a, st); break;
    case 32: launch_sa_update_t<2, float4>(a, st); break;
    case 16: launch_sa_update_t<1, float4>(a, st); break;
    case 72: launch_sa_update_t<4, float4, uint2>(a, st); break;
    case 36: launch_sa_update_t<2, float4, unsigned>(a, st); break;
    case 18: launch_sa_update_t<1, float4, unsigned short>(a, st); break;
    case 8: launch_sa_update_t<1, uint2>(a, st); break;
    default: launch_sa_update_t<0, float4>(a, st); break;
    }
}

// ... and of a set-associative pair
static bool launch_sa_update_pair(const evs::BatchArgs &a1, const evs::BatchArgs &a2, hipStream_t st) {
    using namespace evs;
    static const bool on = !(getenv("EVS_CACHE_PAIR") && getenv("EVS_CACHE_PAIR")[0] == '0');
    if (!on || a1.g1 != a2.g1 || sampled_list_threads(a1) != sampled_list_threads(a2)) return false;
    const dim3 grid((unsigned)(2 * a1.g1)), block(sampled_list_threads(a1));
    if (a1.row_bytes == 36 && a2.row_bytes == 18)
        hipLaunchKernelGGL((cache_batch_sa_list2_kernel<2, float4, unsigned, 1, float4, unsigned short>), grid, block, 0, st, a1, a2);
    else if (a1.row_bytes == 32 && a2.row_bytes == 16)
        hipLaunchKernelGGL((cache_batch_sa_list2_kernel<2, float4, NoTail, 1, float4, NoTail>), grid, block, 0, st, a1, a2);
    else if (a1.row_bytes == 16 && a2.row_bytes == 8)
        hipLaunchKernelGGL((cache_batch_sa_list2_kernel<1, float4, NoTail, 1, uint2, NoTail>), grid, block, 0, st, a1, a2);
    else if (a1.row_bytes == 144 && a2.row_bytes == 36)
        hipLaunchKernelGGL((cache_batch_sa_list2_kernel<9, float4, NoTail, 2, float4, unsigned>), grid, block, 0, st, a1, a2);
    else return false;
    return true;
}

// both tiers' list updates as one launch (the pairs of row sizes a u8 C1 + u4 C2 make); false: no merged kernel for the pair
static bool launch_sampled_update_pair(const evs::BatchArgs &a1, const evs::BatchArgs &a2, hipStream_t st) {
    using namespace evs;
    static const bool on = !(getenv("EVS_CACHE_PAIR") && getenv("EVS_CACHE_PAIR")[0] == '0');
    if (!on || !a1.miss_rec || !a2.miss_rec || a1.g1 != a2.g1 || sampled_list_threads(a1) != sampled_list_threads(a2)) return false;
    const dim3 grid((unsigned)(2 * a1.g1)), block(sampled_list_threads(a1));
    if (a1.row_bytes == 36 && a2.row_bytes == 18)
        hipLaunchKernelGGL((cache_batch_sampled_list2_kernel<2, float4, unsigned, 1, float4, unsigned short>), grid, block, 0, st, a1, a2);
    else if (a1.row_bytes == 32 && a2.row_bytes == 16)
        hipLaunchKernelGGL((cache_batch_sampled_list2_kernel<2, float4, NoTail, 1, float4, NoTail>), grid, block, 0, st, a1, a2);
    else if (a1.row_bytes == 16 && a2.row_bytes == 8)
        hipLaunchKernelGGL((cache_batch_sampled_list2_kernel<1, float4, NoTail, 1, uint2, NoTail>), grid, block, 0, st, a1, a2);
    else return false;
    return true;
}

// The batch policy of a cache that was not given one (evs_cache_set_batch_policy) is decided at its first batched call:
// EVS_CACHE_POLICY = plan | sampled | setassoc, else the set-associative form where it applies -- a single tier whose
// tables the kernels read in place (HBM), at least one full set -- and the sampled update everywhere else (host-memory
// and file-backed miss tiers, the two- / three-tier lookups).
static int resolved_batch_policy(evs_cache *c, bool single_tier) {
    if (c->batch_policy < 0) {
        const char *e = getenv("EVS_CACHE_POLICY");
        const bool sa_ok = single_tier && !c->host_backing && !c->ft && sa_single_feasible(c);
        if (e && e[0] == 'p') c->batch_policy = 0;
        else if (e && e[0] == 's' && e[1] == 'a') c->batch_policy = 1;
        else c->batch_policy = sa_ok ? 2 : 1;
    }
    return c->batch_policy;
}
// sampled policy: fold the counters of the batches since the last close
static evs::CloseArgs sampled_close_args(evs_cache *c, int rebuild) {
    evs::CloseArgs ca;
    ca.bs = c->bs; ca.part1 = c->part1; ca.part2 = c->part2; ca.host_words = c->host_tomb_dev;
    ca.T = c->host.n_tables; ca.cap = (int)c->host.cap; ca.max_perfect = c->host.max_perfect; ca.rebuild = rebuild;
    ca.n_batches = c->pending_batches; ca.n_requests = c->pending_requests;
    return ca;
}
static void sampled_close_pending(evs_cache *c, int rebuild, hipStream_t st) {
    using namespace evs;
    if ((c->batch_policy != 1 && c->batch_policy != 2) || !c->bs || c->pending_batches == 0) return;
    const CloseArgs ca = sampled_close_args(c, rebuild);
    hipLaunchKernelGGL(cache_batch_sampled_close_kernel, dim3(1), dim3(256), 0, st, ca);
    c->pending_batches = 0; c->pending_requests = 0;
}
// ... of both tiers of a two-tier lookup, one launch
static void sampled_close_pending2(evs_cache *c1, int rebuild1, evs_cache *c2, int rebuild2, hipStream_t st) {
    using namespace evs;
    const bool on1 = (c1->batch_policy == 1 || c1->batch_policy == 2) && c1->bs && c1->pending_batches,
               on2 = (c2->batch_policy == 1 || c2->batch_policy == 2) && c2->bs && c2->pending_batches;
    if (!on1 || !on2) { sampled_close_pending(c1, rebuild1, st); sampled_close_pending(c2, rebuild2, st); return; }
    const CloseArgs ca1 = sampled_close_args(c1, rebuild1), ca2 = sampled_close_args(c2, rebuild2);
    hipLaunchKernelGGL(cache_batch_sampled_close2_kernel, dim3(2), dim3(256), 0, st, ca1, ca2);
    c1->pending_batches = 0; c1->pending_requests = 0;
    c2->pending_batches = 0; c2->pending_requests = 0;
}
// sampled policy: the EvLFU flush the close of an earlier batch asked for (a flag in mapped host memory)
static void sampled_flush_if_wanted(evs_cache *c, hipStream_t st) {
    using namespace evs;
    if ((c->batch_policy != 1 && c->batch_policy != 2) || !c->host_tomb || !c->bs) return;
    volatile int *flags = reinterpret_cast<volatile int *>(c->host_tomb);
    if (!flags[1]) return;
    flags[1] = 0;
    c->last_flush_call = c->batch_calls;   // "full" reports of closes before this call are void
    const int wide = kNumCu * 8;
    long long nf = ((long long)c->host.cap + 255) / 256; if (nf > wide) nf = wide;
    if (c->batch_policy == 2) {
        hipLaunchKernelGGL(cache_batch_sa_flush_kernel, dim3((unsigned)nf), dim3(256), 0, st, c->bs, c->sa,
                           c->host.n_tables, c->host.flush_n, c->host.max_perfect);
        hipLaunchKernelGGL(cache_batch_sampled_flush_finish_kernel, dim3(1), dim3(1), 0, st, c->bs, c->host.n_tables);
        return;
    }
    hipLaunchKernelGGL(cache_batch_sampled_flush_kernel, dim3((unsigned)nf), dim3(256), 0, st, c->bs, c->a, c->bslots, c->eslot,
                       (int)c->host.cap, c->host.n_tables, c->host.flush_n, c->host.max_perfect);
    hipLaunchKernelGGL(cache_batch_sampled_flush_finish_kernel, dim3(1), dim3(1), 0, st, c->bs, c->host.n_tables);
}

static int cache_batch_impl(evs_cache *c, int64_t B, const int32_t *rows, float *out, uint8_t *hit, const float *x,
                            int64_t x_stride, int itself, float *R, void *stream) {
    using namespace evs;
    if (B == 0) return EVS_OK;
    EVS_REQUIRE(hit, "evs_cache_lookup_batch: NULL hit");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    BatchArgs a;
    const int prc = batch_prepare(c, B, rows, st, a, "evs_cache_lookup_batch");
    if (prc) return prc;
    const int T = c->host.n_tables;
    const long long cap = c->host.cap;
    const int wide = kNumCu * 8;
    a.out = out; a.hit = hit;
    // Host-memory miss tier: hits of this batch are pinned (stamp), the policy update runs FIRST, the missed keys
    // that got an entry are re-pointed at their arena rows and only then do the consumers read -- every missing
    // row crosses the bus once (K5's de-duplicated fetch).  Miss tier in HBM: consumers first, as measured best.
    const bool host_tier = c->host_backing;
    if (host_tier) {
        if (!c->estamp) {
            EVS_HIP_CHECK(hipMalloc(&c->estamp, cap * 4));
            EVS_HIP_CHECK(hipMemsetAsync(c->estamp, 0, cap * 4, st));
        }
        a.estamp = c->estamp; a.stamp_hits = 1;
        a.stamp = (int)(++c->stamp_counter % 0x7ffffffe) + 1;   // never 0, distinct for consecutive batches
    }
    auto consumers = [&]() -> int {
        if (out) {
            long long nb = (B * T * (long long)c->host.dim / 4 + 255) / 256; if (nb > wide) nb = wide; if (nb < 1) nb = 1;
            if (c->host.codec == 32)
                hipLaunchKernelGGL(cache_rows_from_ptrs_kernel, dim3((unsigned)nb), dim3(256), 0, st, c->row_ptrs, out,
                                   (long long)B, T, c->host.dim, c->host.codec);
            else   // reduced precision: chunked decode through LDS tables
                hipLaunchKernelGGL(cache_rows_from_ptrs2_kernel, dim3((unsigned)nb), dim3(256), 0, st, c->row_ptrs,
                                   (const unsigned char *)nullptr, out, (long long)B, T, c->host.dim, c->host.codec, c->host.codec);
        }
        if (R && c->host.codec != 32) {
            // a reduced-precision tier: every row decoded inside the interaction kernel (evs_mixed.hip; all rows of one class --
            // u8 / u4 rows take the (u8, u4) rows-in-registers kernel with every row in its first / second class)
            const int rc = c->host.codec == 4
                ? interact_from_mixed_rows(B, T, c->host.dim, x, x_stride, c->row_ptrs, nullptr, 8, 4, itself, R, st, 2)   // every row in the pair's second class
                : interact_from_mixed_rows(B, T, c->host.dim, x, x_stride, c->row_ptrs, nullptr, c->host.codec,
                                           c->host.codec == 8 ? 4 : c->host.codec, itself, R, st);
            if (rc) return rc;
        } else
        if (R) {
            const int rc = a.row_ids
                ? fused_interact_from_row_ids(B, T, c->host.dim, x, x_stride, a.row_ids, c->a.arena,
                                              reinterpret_cast<const void *const *>(c->backing), itself, R, st)
                : fused_interact_from_row_ptrs(B, T, c->host.dim, x, x_stride, (const int64_t *)c->row_ptrs,
                                               (const int64_t *)c->iota, itself, R, st);
            if (rc) return rc;
        }
        return EVS_OK;
    };
    // The interaction alone as the consumer, tables in HBM, a shape the rows-in-registers kernel takes: K1 writes 4-byte
    // row ids (arena entry / table row) instead of 8-byte addresses and that kernel reads them (22.9 -> ~20 us at
    // B = 16 384).  The 8-byte address table is the (B,T) int64 buffer either way: the ids use its first half.
    if (R && !out && !host_tier && !(c->ft && c->staged_mask) && c->host.codec == 32 && fused_row_ids_supported(B, T, c->host.dim)) {
        bool small = true;
        for (int k = 0; k < T; k++) small = small && c->backing_rows[k] < (1ll << 30);
        if (small) a.row_ids = reinterpret_cast<int *>(c->row_ptrs);
    }
    if (R && c->host.codec == 32)
        EVS_REQUIRE(evs_fused_dim_supported(c->host.dim) && T + 1 <= EVS_MAX_FEATURES,
                    "evs_cache_lookup_interact: needs a fused-kernel dimension and T <= 31");
    if (R && c->host.codec != 32)
        EVS_REQUIRE((c->host.dim == 16 || c->host.dim == 32 || c->host.dim == 36) && T + 1 <= EVS_MAX_FEATURES && !host_tier,
                    "evs_cache_lookup_interact: a reduced-precision cache needs d in {16, 32, 36}, T <= 31 and its tables in HBM");
    // consumers of the snapshot read the rows BEFORE the policy kernels move anything
    const bool file_mode = c->ft && c->staged_mask;
    if (resolved_batch_policy(c) == 2) {
        // Set-associative policy (evs_hash.h): probe (one line per key) -> consumers -> ONE update kernel (one line, one
        // CAS and the row per new key) -> counters folded every kCloseEvery-th batch.  No hash, no tombstones, no sweeps.
        if (host_tier || c->ft) { set_error("evs_cache_lookup_batch: the set-associative batch policy reads its miss tier in place from HBM (no host-memory / file-backed tables)"); return EVS_ESTATE; }
        a.stamp = (int)(++c->stamp_counter % 0x7ffffffe) + 1;
        sampled_flush_if_wanted(c, st);
        a.miss_rec = c->miss_rec; a.list_cnt = c->list_cnt;
        a.list_cap = (int)((B + 8 * (long long)a.g1 - 1) / (8 * (long long)a.g1)) * 8 * T;
        static const bool fold_on = !(getenv("EVS_CACHE_FOLD") && getenv("EVS_CACHE_FOLD")[0] == '0');
        bool small_tabs = true;   // (tile entries carry a row id in 30 bits)
        for (int k = 0; k < T; k++) small_tabs = small_tabs && c->backing_rows[k] < (1ll << 30);
        // a reduced-precision tier (the reference's one-layer 16 / 8 / 4-bit builds): the probe folded into ITS consumer too
        static const bool foldq_on = !(getenv("EVS_CACHE_FOLDQ") && getenv("EVS_CACHE_FOLDQ")[0] == '0');
        const bool foldq = fold_on && foldq_on && c->host.codec != 32 && R && !out && c->sa.ways == 8 && c->sa.sub_shift == 0 && small_tabs &&
                           (cap << c->sa.dual) < (1ll << 30) && fused_probe_codec_supported(B, T, c->host.dim, c->host.codec);
        const bool fold = (fold_on && a.row_ids && R && !out && c->sa.ways == 8 && (cap << c->sa.dual) < (1ll << 30)) || foldq;   // (the folded probes are compiled for 8-way sets; tile entries carry an arena row in 30 bits)
        // the update inside the probe launch too: fp32 rows, the folded fp32 launch, a two-copy arena (EVS_CACHE_INLINE=0: every
        // batch updated by a launch of its own behind it -- the round-4 chain, strict snapshot flags)
        if (c->inline_mode < 0) c->inline_mode = (getenv("EVS_CACHE_INLINE") && getenv("EVS_CACHE_INLINE")[0] == '0') ? 0 : 1;
        // (a reduced-precision tier takes the same form in its own consumer: evs_fused_rfq.hip, PROBE)
        // (a way stamped by the RUNNING batch is hidden from this launch's probers; the stamp is the batch number modulo
        //  2^stamp_bits, so an entry last touched exactly k * 2^stamp_bits batches ago is hidden too -- a miss served from its
        //  table, exact as ever, but a lost hit: the form is taken only where that is one batch in 256 or rarer, i.e. never for
        //  the tiny set counts whose tags leave the stamp a few bits)
        const bool inl = fold && c->inline_mode == 1 && c->sa.dual && c->sa.stamp_mask >= 255u &&
                         (foldq || (c->host.codec == 32 && (c->host.row_bytes & 15) == 0));
        if (fold) {
            ProbeArgs pa;
            pa.slots = nullptr; pa.mask = 0; pa.reusable_tomb = kTomb; pa.eagg = nullptr;
            pa.requests = rows; pa.hit = hit;
            pa.miss_rec = a.miss_rec; pa.list_cnt = a.list_cnt; pa.list_cap = 16 * T;
            pa.part1 = a.part1; pa.hint_shift = 0; pa.T = T;
            pa.sa = c->sa; pa.sau = c->sau;
            a.list_cap = 16 * T;
            a.g1 = (int)((B + 15) / 16);
            if (inl) {
                pa.miss_rec = nullptr; pa.list_cnt = nullptr;
                pa.pend_stamp = (unsigned)a.stamp & c->sa.stamp_mask;
                pa.arena_w = c->a.arena; pa.row_bytes = c->host.row_bytes; pa.part2 = a.part2;
                static const int xf = getenv("EVS_X_INL") ? atoi(getenv("EVS_X_INL")) : 0;
                pa.xflags = xf;
            }
            const int rc = fused_probe_interact(B, T, c->host.dim, x, x_stride, pa, c->a.arena,
                                                reinterpret_cast<const void *const *>(c->backing), c->backing_rows, itself, R, st, c->host.codec);
            if (rc) return rc;
            if (inl) {   // nothing is left to do behind the launch
                c->pending_batches++; c->pending_requests += B;
                if (c->pending_batches >= kCloseEvery) sampled_close_pending(c, 0, st);
                EVS_HIP_CHECK(hipGetLastError());
                return EVS_OK;
            }
        } else {
            hipLaunchKernelGGL(cache_batch_probe_gather_kernel, dim3((unsigned)a.g1), dim3(256), 0, st, a);
            const int rc = consumers(); if (rc) return rc;
        }
        launch_sa_update(a, st);
        c->pending_batches++; c->pending_requests += B;
        if (c->pending_batches >= kCloseEvery) sampled_close_pending(c, 0, st);
        EVS_HIP_CHECK(hipGetLastError());
        return EVS_OK;
    }
    if (resolved_batch_policy(c) == 1 && !file_mode) {
        // Sampled policy update: probe -> consumers -> ONE update kernel -> close.  (Host-memory miss tier: the update
        // first -- it fetches each missing row once -- then the re-pointed consumers.)
        if (!host_tier) a.stamp = (int)(++c->stamp_counter % 0x7ffffffe) + 1;   // (host tier: set above, with the hit stamps)
        a.tomb_parity = a.stamp & 1;
        sampled_flush_if_wanted(c, st);   // EvLFU flush the close of an earlier batch asked for: before this batch's probe
        if (!host_tier) {   // K1 lists the misses itself (the patch kernel of the host tier reads the per-position records)
            a.miss_rec = c->miss_rec; a.list_cnt = c->list_cnt;
            a.list_cap = (int)((B + 8 * (long long)a.g1 - 1) / (8 * (long long)a.g1)) * 8 * T;
        }
        if (c->fork_mode < 0) {
            const char *e = getenv("EVS_CACHE_FORK");
            c->fork_mode = (e && e[0] == '1') ? 1 : 0;
        }
        // EVS_CACHE_FORK=1: the update runs on a side stream UNDER the consumer.  What makes that legal: with the hits of
        // the batch stamped (K1) the update never takes an entry the consumer reads, and the consumer reads nothing
        // else the update writes.
        const bool fork = c->fork_mode == 1 && !host_tier && (R || out);
        if (fork) {
            if (!c->estamp) {
                EVS_HIP_CHECK(hipMalloc(&c->estamp, cap * 4));
                EVS_HIP_CHECK(hipMemsetAsync(c->estamp, 0, cap * 4, st));
            }
            a.estamp = c->estamp; a.stamp_hits = 1;
            if (!c->side) {
                EVS_HIP_CHECK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
                EVS_HIP_CHECK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
                EVS_HIP_CHECK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
            }
        }
        // the probe inside the consumer: one launch instead of two when the consumer is the rows-in-registers kernel
        // (EVS_CACHE_FOLD=0: the two-launch form)
        static const bool fold_on = !(getenv("EVS_CACHE_FOLD") && getenv("EVS_CACHE_FOLD")[0] == '0');
        const bool fold = fold_on && a.row_ids && a.miss_rec && !fork && R && !out;
        if (fold) {
            ProbeArgs pa;
            pa.slots = a.slots; pa.mask = a.mask; pa.reusable_tomb = a.tomb_parity ? kTomb : kTomb1;
            pa.eagg = a.a.eagg; pa.requests = rows; pa.hit = hit;
            pa.miss_rec = a.miss_rec; pa.list_cnt = a.list_cnt; pa.list_cap = 16 * T;
            pa.part1 = a.part1; pa.hint_shift = a.hint_shift; pa.T = T;
            pa.sa = evs::SaGeom{}; pa.sau = c->sau;
            a.list_cap = 16 * T;
            a.g1 = (int)((B + 15) / 16);   // the update kernel runs one wave per list: here a list per 16-sample block
            const int rc = fused_probe_interact(B, T, c->host.dim, x, x_stride, pa, c->a.arena,
                                                reinterpret_cast<const void *const *>(c->backing), c->backing_rows, itself, R, st);
            if (rc) return rc;
        } else
        hipLaunchKernelGGL(cache_batch_probe_gather_kernel, dim3((unsigned)a.g1), dim3(256), 0, st, a);
        hipStream_t su = st;
        if (fold) {
        } else if (fork) {
            EVS_HIP_CHECK(hipEventRecord(c->ev_fork, st));
            EVS_HIP_CHECK(hipStreamWaitEvent(c->side, c->ev_fork, 0));
            su = c->side;
        } else if (!host_tier) { const int rc = consumers(); if (rc) return rc; }
        launch_sampled_update(a, su);
        if (fork) {
            EVS_HIP_CHECK(hipEventRecord(c->ev_join, c->side));
            const int rc = consumers(); if (rc) return rc;
            EVS_HIP_CHECK(hipStreamWaitEvent(st, c->ev_join, 0));
        }
        if (host_tier) {
            hipLaunchKernelGGL(cache_batch_patch_ptrs_kernel, dim3((unsigned)a.g2), dim3(256), 0, st, a);
            const int rc = consumers(); if (rc) return rc;
        }
        c->pending_batches++; c->pending_requests += B;
        if (c->pending_batches >= kCloseEvery || a.rebuild) sampled_close_pending(c, a.rebuild, st);
        batch_housekeeping(c, a, st);
        EVS_HIP_CHECK(hipGetLastError());
        return EVS_OK;
    }
    hipLaunchKernelGGL(cache_batch_probe_gather_kernel, dim3((unsigned)a.g1), dim3(256), 0, st, a);
    if (c->fork_mode < 0) {
        const char *e = getenv("EVS_CACHE_FORK");
        c->fork_mode = (e && e[0] == '1') ? 1 : 0;
    }
    // Fork / join inside the batch (EVS_CACHE_FORK=1, off by default): K2, K3 and K4 only touch the hash, the
    // priorities and the free stack -- nothing the consumer reads (row addresses, arena and backing rows) -- so they
    // can run on a side stream UNDER the consumer, K5 (which overwrites arena rows) joining behind both.  Same
    // snapshot semantics, same results -- and slower on this stack: 148 us per batch against 125 (300 unseen batches,
    // B = 16 384): both sides slow down when they overlap (consumer 22 -> 25 us, K2 27 -> 37) and the two
    // cross-stream event waits cost more than the overlap returns.
    const bool fork = c->fork_mode == 1 && !host_tier && !file_mode && (R || out);
    if (fork) {
        if (!c->side) {
            EVS_HIP_CHECK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
            EVS_HIP_CHECK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
            EVS_HIP_CHECK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
        }
        EVS_HIP_CHECK(hipEventRecord(c->ev_fork, st));
        EVS_HIP_CHECK(hipStreamWaitEvent(c->side, c->ev_fork, 0));
        batch_policy_a(c, a, c->side);
        {
            const int widef = kNumCu * 8;
            long long ne = ((long long)a.cap + 255) / 256; if (ne > widef) ne = widef;
            hipLaunchKernelGGL(cache_batch_evict_kernel, dim3((unsigned)ne), dim3(256), 0, c->side, a);
        }
        EVS_HIP_CHECK(hipEventRecord(c->ev_join, c->side));
        const int rc = consumers(); if (rc) return rc;
        EVS_HIP_CHECK(hipStreamWaitEvent(st, c->ev_join, 0));
        hipLaunchKernelGGL(cache_batch_assign_kernel, dim3((unsigned)a.g2), dim3(256), 0, st, a);
        batch_close(c, a, st);
        EVS_HIP_CHECK(hipGetLastError());
        return EVS_OK;
    }
    if (!host_tier) { const int rc = consumers(); if (rc) return rc; }
    if (file_mode) { const int rc = batch_stage_prepare(c, a, st); if (rc) return rc; }
    batch_policy_a(c, a, st);
    if (file_mode) { const int rc = batch_stage_rows(c, a, st); if (rc) return rc; }
    batch_policy_b(c, a, st);
    if (host_tier) {
        hipLaunchKernelGGL(cache_batch_patch_ptrs_kernel, dim3((unsigned)a.g2), dim3(256), 0, st, a);
        const int rc = consumers(); if (rc) return rc;
    }
    if (file_mode) hipLaunchKernelGGL(cache_batch_unstage_kernel, dim3((unsigned)a.g2), dim3(256), 0, st, a);
    batch_close(c, a, st);
    EVS_HIP_CHECK(hipGetLastError());
    return EVS_OK;
}

// ---- batched two-tier lookup (C1 main precision + C2 secondary precision), snapshot semantics ------------------
// The throughput form of request_to_c1_c2 (mixed_precs_caching/evlfu_8.cpp:669-796), no reference counterpart
// (the reference is batch-1): every key of the batch is probed in C1, then in C2, against the tiers as they
// stand when the call starts; agg_hit of a request = keys found in either tier; a hit is served (and its priority
// raised) in the tier that holds it; a double miss is routed by the reference's rule evaluated on the snapshot --
// C1 not full: C1; C1 full and agg_hit < threshold: odd table index -> C1, even -> C2; else C2 -- served from the
// destination tier's backing table at that tier's precision and inserted there once per batch (a key two requests
// route differently goes to C1).  Each tier then runs the single-tier policy update (K2-K6) on its own misses.
static int batch_c1c2_impl(evs_cache *c1, evs_cache *c2, evs_aprx *c3, int64_t B, const int32_t *rows, float *out, uint8_t *tier,
                           int high_agghit_threshold, const float *x, int64_t x_stride, int itself, float *R, void *stream);

extern "C" int evs_cache_lookup_batch_c1c2(evs_cache *c1, evs_cache *c2, int64_t B, const int32_t *rows, float *out,
                                           uint8_t *tier, int high_agghit_threshold, void *stream) {
    EVS_REQUIRE(out || B == 0, "evs_cache_lookup_batch_c1c2: NULL out");
    return batch_c1c2_impl(c1, c2, nullptr, B, rows, out, tier, high_agghit_threshold, nullptr, 0, 0, nullptr, stream);
}

// ... with the interaction as the consumer: R = interact_features(x, rows served by the two tiers), every row decoded
// from the precision of the tier that serves it INSIDE the interaction kernel (evs_mixed.hip) -- no fp32 (B,T,d) rows
extern "C" int evs_cache_lookup_interact_c1c2(evs_cache *c1, evs_cache *c2, int64_t B, const int32_t *rows, const float *x,
                                              int64_t x_stride, int itself, float *R, uint8_t *tier,
                                              int high_agghit_threshold, void *stream) {
    EVS_REQUIRE((x && R) || B == 0, "evs_cache_lookup_interact_c1c2: NULL x / R");
    EVS_REQUIRE(x_stride % 4 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0, "evs_cache_lookup_interact_c1c2: x must be 16-byte aligned, stride %% 4 == 0");
    return batch_c1c2_impl(c1, c2, nullptr, B, rows, nullptr, tier, high_agghit_threshold, x, x_stride, itself, R, stream);
}

// ---- batched THREE-tier lookup: the two-tier snapshot lookup with the alt-key tier C3 (request_to_c1_c2_c3,
// evlfu_8.cpp:492-667, as a throughput form; no reference counterpart).  A double miss whose key is a member of C3 and
// whose alt row is resident in C1 (else C2) when the call starts is served that row (tier code 3), decoded at the
// precision of the tier holding it; its recency flag is set, the request's agg_hit counts it, nothing is inserted for
// it.  The keys this batch's policy update EVICTS from C1 / C2 become members of C3 (visible from the next batch on).
extern "C" int evs_cache_lookup_batch_c1c2c3(evs_cache *c1, evs_cache *c2, evs_aprx *c3, int64_t B, const int32_t *rows,
                                             float *out, uint8_t *tier, int high_agghit_threshold, void *stream) {
    EVS_REQUIRE(out || B == 0, "evs_cache_lookup_batch_c1c2c3: NULL out");
    EVS_REQUIRE(c3, "evs_cache_lookup_batch_c1c2c3: NULL alt-key tier");
    return batch_c1c2_impl(c1, c2, c3, B, rows, out, tier, high_agghit_threshold, nullptr, 0, 0, nullptr, stream);
}
extern "C" int evs_cache_lookup_interact_c1c2c3(evs_cache *c1, evs_cache *c2, evs_aprx *c3, int64_t B, const int32_t *rows,
                                                const float *x, int64_t x_stride, int itself, float *R, uint8_t *tier,
                                                int high_agghit_threshold, void *stream) {
    EVS_REQUIRE((x && R) || B == 0, "evs_cache_lookup_interact_c1c2c3: NULL x / R");
    EVS_REQUIRE(c3, "evs_cache_lookup_interact_c1c2c3: NULL alt-key tier");
    EVS_REQUIRE(x_stride % 4 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0, "evs_cache_lookup_interact_c1c2c3: x must be 16-byte aligned, stride %% 4 == 0");
    return batch_c1c2_impl(c1, c2, c3, B, rows, nullptr, tier, high_agghit_threshold, x, x_stride, itself, R, stream);
}
// out4 (host): [members, alt hits served, capacity of the sets, 0]; pairs (host, may be NULL): (table_1based, row, flag) triples
extern "C" int64_t evs_aprx_batch_dump(evs_aprx *p, int64_t *triples, int64_t max_triples, int64_t *out4, void *stream) {
    using namespace evs;
    if (!p) { set_error("evs_aprx_batch_dump: NULL tier"); return EVS_EINVAL; }
    if (hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)) != hipSuccess) return EVS_EHIP;
    const long long n = p->nset * kSetWays;
    std::vector<unsigned long long> tags(n);
    long long st[2] = {0, 0};
    if (hipMemcpy(tags.data(), p->tags, n * 8, hipMemcpyDeviceToHost) != hipSuccess) return EVS_EHIP;
    if (hipMemcpy(st, p->bstat, sizeof st, hipMemcpyDeviceToHost) != hipSuccess) return EVS_EHIP;
    int64_t m = 0;
    for (long long i = 0; i < n; i++) {
        if (!tags[i]) continue;
        const unsigned long long key = tags[i] & kKeyMask;
        if (triples && m < max_triples) { triples[3 * m] = (int64_t)(key >> 32); triples[3 * m + 1] = (int64_t)(key & 0xffffffffull); triples[3 * m + 2] = (tags[i] & kC3Flag) ? 1 : 0; }
        m++;
    }
    if (out4) { out4[0] = st[0]; out4[1] = st[1]; out4[2] = n; out4[3] = 0; }
    return m;
}

static int batch_c1c2_impl(evs_cache *c1, evs_cache *c2, evs_aprx *c3, int64_t B, const int32_t *rows, float *out, uint8_t *tier,
                           int high_agghit_threshold, const float *x, int64_t x_stride, int itself, float *R, void *stream) {
    using namespace evs;
    if (B == 0) return EVS_OK;
    EVS_REQUIRE(c1 && c2 && tier, "evs_cache_lookup_batch_c1c2: NULL argument");
    if (R) EVS_REQUIRE((c1->host.dim == 16 || c1->host.dim == 32 || c1->host.dim == 36) && c1->host.n_tables + 1 <= EVS_MAX_FEATURES,
                       "evs_cache_lookup_interact_c1c2: d must be 16, 32 or 36 and T <= 31");
    EVS_REQUIRE(c1->host.n_tables == c2->host.n_tables && c1->host.dim == c2->host.dim,
                "evs_cache_lookup_batch_c1c2: the tiers must agree on n_tables and dim");
    // Miss tiers outside HBM (BASELINE configs[4] composed: the reference's C1 / C2 read their misses from files inside the
    // request, evlfu_8.cpp:380-414 get_from_file + the reader pool :191-250, :603-625): pinned host tables
    // (evs_cache_set_backing) or file-backed ones (evs_cache_set_file_backing) under EITHER tier.  The pair then runs the
    // host-tier order -- probe (hits and served alt rows stamped), both tiers' updates (each missing row fetched ONCE into its
    // tier's arena: over the bus by the update kernel, or by the host's reader pool for staged tables), a patch kernel that
    // points every missed position at the arena / staged copy, and only then the consumers.
    const bool host2 = c1->host_backing || c2->host_backing;
    const bool file2 = (c1->ft && c1->staged_mask) || (c2->ft && c2->staged_mask);
    // a pair that was given no policy takes the set-associative form where both tiers can (tables in HBM, at least one full
    // set each), the sampled update over host-memory tables, the plan-based one when a tier has STAGED tables (its reader
    // pool works between plan and assign); a tier with a policy of its own decides for the other; one set-associative and
    // one hashed tier: refused
    {
        const bool free1 = c1->batch_policy < 0, free2 = c2->batch_policy < 0;
        const bool sa_ok = !host2 && !c1->ft && !c2->ft && (c1->sa.tags || sa_single_feasible(c1)) && (c2->sa.tags || sa_single_feasible(c2));
        if (file2) {
            if (free1) c1->batch_policy = 0;
            if (free2) c2->batch_policy = 0;
        } else if (free1 && free2 && !(getenv("EVS_CACHE_POLICY"))) c1->batch_policy = c2->batch_policy = sa_ok ? 2 : 1;
        else if (free1 != free2) {
            evs_cache *set = free1 ? c2 : c1, *unset = free1 ? c1 : c2;
            unset->batch_policy = (set->batch_policy == 2 && !sa_ok) ? 1 : set->batch_policy;
        }
        const int p1 = resolved_batch_policy(c1, true), p2 = resolved_batch_policy(c2, true);
        EVS_REQUIRE((p1 == 2) == (p2 == 2), "evs_cache_lookup_batch_c1c2: both tiers take the set-associative batch policy, or neither (C1 %d, C2 %d)", p1, p2);
        if (p1 == 2 && host2) { set_error("evs_cache_lookup_batch_c1c2: the set-associative batch policy reads its miss tier in place from HBM (no host-memory / file-backed tables)"); return EVS_ESTATE; }
        EVS_REQUIRE(p1 != 2 || sa_ok, "evs_cache_lookup_batch_c1c2: the set-associative batch policy needs both tiers' tables in HBM and at least %d entries each", (int)kSaSingleWays);
        if (file2 && (p1 != 0 || p2 != 0)) { set_error("evs_cache_lookup_batch_c1c2: a pair with staged file-backed tables takes the plan-based batch policy on both tiers (C1 %d, C2 %d)", p1, p2); return EVS_ESTATE; }
        EVS_REQUIRE(!host2 || (p1 == p2), "evs_cache_lookup_batch_c1c2: tiers over host-memory / file-backed tables take the same batch policy (C1 %d, C2 %d)", p1, p2);
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // two set-associative tiers that start out together share their set records: ONE 128-byte record per set holds C1's and
    // C2's ways, so a key's two probes are one line request (a tier that was used alone before keeps records of its own)
    if (resolved_batch_policy(c1) == 2 && !c1->sa.tags && !c2->sa.tags && !c1->bs && !c2->bs) {
        SaUniverse u; SaGeom g1, g2;
        if (sa_pair_geometry(c1, c2, u, g1, g2)) {
            c1->sau = u; c2->sau = u; c1->sa = g1; c2->sa = g2;
            const int arc = sa_alloc(c1, c2, st);
            if (arc) { c1->sa = SaGeom{}; c2->sa = SaGeom{}; set_error("evs_cache_lookup_batch_c1c2: allocating the set records failed"); return arc; }
        }
    }
    // (The pair's update folded into the probe launch as well -- the thread that routes a double miss claims a way of the
    //  destination tier, whose ways it has to read AGAIN once the request's agg_hit has decided the destination, a look at the
    //  other tier's set behind the claim so that a key routed both ways ends up in one tier, the rows copied table -> arena at
    //  the end of the block -- was built and measured in round 5: 48.5-49.3 us per batch against 43.0-44.2 for probe launch +
    //  update launch on the same box (three tiers 60.4-61.7 against 57.2).  A single tier has the missed row in the registers
    //  of the lanes that gather it and its set's ways at hand; a pair has neither.  The pair keeps its update launch.)
    BatchArgs a1, a2;
    int rc = batch_prepare(c1, B, rows, st, a1, "evs_cache_lookup_batch_c1c2");
    if (rc) return rc;
    rc = batch_prepare(c2, B, rows, st, a2, "evs_cache_lookup_batch_c1c2");
    if (rc) return rc;
    const int T = c1->host.n_tables;
    const int wide = kNumCu * 8;
    if (host2) {
        for (int k = 0; k < 2; k++) {
            evs_cache *c = k ? c2 : c1;
            BatchArgs &a = k ? a2 : a1;
            if (!c->estamp) {
                EVS_HIP_CHECK(hipMalloc(&c->estamp, (long long)c->host.cap * 4));
                EVS_HIP_CHECK(hipMemsetAsync(c->estamp, 0, (long long)c->host.cap * 4, st));
            }
            a.estamp = c->estamp; a.stamp_hits = 1;
            a.stamp = (int)(++c->stamp_counter % 0x7ffffffe) + 1;   // never 0, distinct for consecutive batches
        }
    }
    TwoTierArgs tt;
    tt.row_tier = c1->row_tier; tt.tier_out = tier; tt.threshold = high_agghit_threshold;
    tt.route_filter = nullptr; tt.route_mask = 0; tt.route_stamp = 0;
    tt.c3.tags = nullptr; tt.c3.nset = 0; tt.c3.stat = nullptr;
    for (int k = 0; k < kMaxTables; k++) { tt.c3.alt_tables[k] = nullptr; tt.c3.alt_rows[k] = 0; }
    if (c3) {
        EVS_REQUIRE(c3->n_tables == T, "evs_cache_lookup_batch_c1c2c3: the alt-key tier was made for %d tables", c3->n_tables);
        if (!c3->has_alt) { set_error("evs_cache_lookup_batch_c1c2c3: call evs_aprx_set_altkeys first"); return EVS_ESTATE; }
        if (c3->used == 1) { set_error("evs_cache_lookup_batch_c1c2c3: this alt-key tier is used through the exact path"); return EVS_ESTATE; }
        c3->used = 2;
        for (evs_cache *c : {c1, c2})
            if (!c->evicted_keys) EVS_HIP_CHECK(hipMalloc(&c->evicted_keys, (long long)c->host.cap * 8));
        a1.evicted_keys = c1->evicted_keys; a2.evicted_keys = c2->evicted_keys;
        tt.c3.tags = c3->tags; tt.c3.nset = c3->nset; tt.c3.stat = c3->bstat;
        for (int k = 0; k < T; k++) {
            tt.c3.alt_tables[k] = c3->x.alt_tables[k]; tt.c3.alt_rows[k] = c3->x.alt_rows[k];
            EVS_REQUIRE(c3->x.alt_rows[k] >= c1->backing_rows[k] && c3->x.alt_rows[k] >= c2->backing_rows[k],
                        "evs_cache_lookup_batch_c1c2c3: alt-key table %d is shorter than the embedding table", k);
        }
    }
    // both tiers on the sampled update (the default): the probe follows the parity rule of each tier's tombstones
    // sa2: both tiers set-associative (evs_hash.h) -- the same flow as sampled2 (probe lists the misses per tier, one update
    // kernel per tier, lazy closes) over set lines instead of hashes; no tombstones, no housekeeping
    const bool sa2 = resolved_batch_policy(c1) == 2 && resolved_batch_policy(c2) == 2;
    const bool sampled2 = sa2 || (resolved_batch_policy(c1) == 1 && resolved_batch_policy(c2) == 1);
    // the pair's probe permutes a key ONCE, with C1's universe, for both tiers (a1.sau); C2's inserts, evictions and alt-key
    // look-ups use its own.  Tiers that did not start out together (each built its universe alone) must have built the same one
    // -- same row bases, same width -- or C2's (set, tag) would not map back to the key: refused instead of served wrong.
    if (sa2 && memcmp(&c1->sau, &c2->sau, sizeof(SaUniverse)) != 0) {
        set_error("evs_cache_lookup_batch_c1c2: the tiers were built over different key universes (table row counts differ); make both over the same tables");
        return EVS_ESTATE;
    }
    constexpr unsigned kRouteWords = 1u << 20;
    static const bool route_on_env = !(getenv("EVS_CACHE_ROUTEFILTER") && getenv("EVS_CACHE_ROUTEFILTER")[0] == '0');
    const bool route_on = route_on_env || sa2;   // (the set-associative update has no cross-tier hash look-up to fall back on)
    if (sampled2 && route_on && !c1->route_filter) {
        EVS_HIP_CHECK(hipMalloc(&c1->route_filter, kRouteWords * 4));
        EVS_HIP_CHECK(hipMemsetAsync(c1->route_filter, 0, kRouteWords * 4, st));
    }
    if (sampled2) {
        for (int k = 0; k < 2; k++) {
            evs_cache *c = k ? c2 : c1;
            BatchArgs &a = k ? a2 : a1;
            if (!host2) a.stamp = (int)(++c->stamp_counter % 0x7ffffffe) + 1;   // (host tiers: set above, with the hit stamps)
            a.tomb_parity = sa2 ? -1 : (a.stamp & 1);
            sampled_flush_if_wanted(c, st);
            static const bool c3_inline_on = !(getenv("EVS_CACHE_C3INLINE") && getenv("EVS_CACHE_C3INLINE")[0] == '0');
            if (c3 && (c3_inline_on || sa2)) {   // the evicting thread inserts its victim into the alt-key set itself
                a.c3_tags = c3->tags; a.c3_nset = c3->nset; a.c3_stat = c3->bstat;
            } else
            if (c3) {   // victim lists for the alt-key tier: kReplicas lists, a block adds at most 256 keys to one of them
                const long long need = ((long long)a.g2 + kReplicas - 1) / kReplicas * 256;
                if (need > c->vict_cap) {
                    EVS_HIP_CHECK(hipStreamSynchronize(st));
                    if (c->vict_keys) (void)hipFree(c->vict_keys);
                    c->vict_keys = nullptr; c->vict_cap = 0;
                    EVS_HIP_CHECK(hipMalloc(&c->vict_keys, need * kReplicas * 8));
                    c->vict_cap = need;
                }
                if (!c->vict_cnt) {
                    EVS_HIP_CHECK(hipMalloc(&c->vict_cnt, 2 * kReplicas * 4));
                    EVS_HIP_CHECK(hipMemsetAsync(c->vict_cnt, 0, 2 * kReplicas * 4, st));
                }
                a.evicted_keys = c->vict_keys; a.vict_cap = (int)c->vict_cap;
                a.vict_cnt = c->vict_cnt + (a.stamp & 1) * kReplicas; a.vict_other = c->vict_cnt + (1 - (a.stamp & 1)) * kReplicas;
            }
        }
    }
    if (sampled2) {
        // the probe lists each tier's misses per block (as K1 does for one tier); with the alt-key tier attached a list
        // kernel block stages its victims in LDS: lists of at most kListVictMax records
        const long long lc = (B + 8 * (long long)a1.g1 - 1) / (8 * (long long)a1.g1) * 8 * T;
        static const bool list_on = !(getenv("EVS_CACHE_LIST2") && getenv("EVS_CACHE_LIST2")[0] == '0');
        if (sa2 || (list_on && !host2 && a1.g1 == a2.g1 && (!c3 || lc <= kListVictMax))) {   // (host tiers: the patch kernel reads the per-position records)
            a1.miss_rec = c1->miss_rec; a1.list_cnt = c1->list_cnt; a1.list_cap = (int)lc;
            a2.miss_rec = c2->miss_rec; a2.list_cnt = c2->list_cnt; a2.list_cap = (int)lc;
        }
    }
    // The probe inside the consumer (the (u8, u4) rows-in-registers kernel of evs_mixed.hip): one launch instead of two, and
    // the (address, class) pairs never leave the block.  EVS_CACHE_FOLD2=0: the two-launch form.
    if (sampled2 && c1->route_filter) {
        tt.route_filter = c1->route_filter; tt.route_mask = kRouteWords - 1; tt.route_stamp = (unsigned)a1.stamp;
        a2.route_filter = c1->route_filter; a2.route_mask = kRouteWords - 1; a2.route_stamp = (unsigned)a1.stamp;
    }
    static const bool fold2_on = !(getenv("EVS_CACHE_FOLD2") && getenv("EVS_CACHE_FOLD2")[0] == '0');
    const bool fold2 = fold2_on && sampled2 && !host2 && a1.miss_rec && R && !out && B <= 65536 && T <= 32 &&
                       (!sa2 || (c1->sa.ways == 8 && c2->sa.ways == 8 && (c1->host.cap << c1->sa.dual) < (1ll << 30) && (c2->host.cap << c2->sa.dual) < (1ll << 30))) &&   // (the folded probe is compiled for 8-way sets)
                       mixed84_supported(T, c1->host.dim, c1->host.codec, c2->host.codec);
    if (fold2) {
        Probe2Args pa;
        for (int k = 0; k < 2; k++) {
            evs_cache *c = k ? c2 : c1;
            BatchArgs &a = k ? a2 : a1;
            TierProbe &tp = k ? pa.t2 : pa.t1;
            a.g1 = (int)((B + 15) / 16);   // the update kernel runs one block per list: here a list per 16-sample block
            a.list_cap = 16 * T;
            tp.slots = a.slots; tp.mask = a.mask; tp.reusable_tomb = a.tomb_parity ? kTomb : kTomb1;
            tp.eagg = a.a.eagg; tp.arena = a.a.arena; tp.row_bytes = a.row_bytes;
            for (int t = 0; t < 32; t++) { tp.backing[t] = t < T ? a.backing[t] : nullptr; tp.backing_rows[t] = t < T ? a.backing_rows[t] : 0; }
            tp.miss_rec = a.miss_rec; tp.list_cnt = a.list_cnt; tp.part1 = a.part1; tp.hint_shift = a.hint_shift;
            tp.count = &a.bs->count; tp.cap = a.cap; tp.full_slack = a.cap > 65536 ? a.cap / 256 : 0;
            tp.sa = a.sa;
            (void)c;
        }
        pa.requests = rows; pa.tier_out = tier; pa.threshold = high_agghit_threshold; pa.T = T; pa.list_cap = 16 * T;
        pa.c3 = tt.c3;
        pa.sau = a1.sau;
        pa.route_filter = tt.route_filter; pa.route_mask = tt.route_mask; pa.route_stamp = tt.route_stamp;
        rc = probe2_interact_mixed84(B, T, c1->host.dim, x, x_stride, pa, itself, R, st);
        if (rc) return rc;
    } else
    hipLaunchKernelGGL(cache_batch_probe2_kernel, dim3((unsigned)a1.g1), dim3(256), 0, st, a1, a2, tt);
    if (host2) {
        // host-memory / file-backed miss tiers: updates first, then the re-pointed consumers
        a2.row_ptrs = c1->row_ptrs;   // both tiers' positions live in C1's pointer table
        if (sampled2) {
            if (!a2.route_filter) { a2.other_slots = c1->bslots; a2.other_mask = (unsigned long long)(c1->bnslot - 1); }
            launch_sampled_update(a1, st);   // per-position forms (a.miss_rec == NULL); the row of a new key is read by the one thread that inserts it
            launch_sampled_update(a2, st);
            if (c3 && !a1.c3_tags) hipLaunchKernelGGL(c3_batch_insert_lists_kernel, dim3(2 * kReplicas * 8), dim3(256), 0, st, a1, a2, tt.c3);
        } else {
            for (int k = 0; k < 2; k++) {
                evs_cache *c = k ? c2 : c1;
                BatchArgs &a = k ? a2 : a1;
                const bool staged = c->ft && c->staged_mask;
                if (staged) { rc = batch_stage_prepare(c, a, st); if (rc) return rc; }
                if (k) { a2.other_slots = c1->bslots; a2.other_mask = (unsigned long long)(c1->bnslot - 1); }   // a key C1 just took is not inserted in C2 too
                batch_policy_a(c, a, st);
                if (staged) { rc = batch_stage_rows(c, a, st); if (rc) return rc; }   // the reader pool: each new key's row out of its file, once
                batch_policy_b(c, a, st);
                if (c3) hipLaunchKernelGGL(c3_batch_insert_kernel, dim3((unsigned)a.g2), dim3(256), 0, st, a, tt.c3);   // what this tier evicted
            }
        }
        hipLaunchKernelGGL(cache_batch_patch_ptrs2_kernel, dim3((unsigned)a1.g2), dim3(256), 0, st, a1, a2, c1->row_tier);
        if (out) {
            long long nb = (B * T * (long long)c1->host.dim / 4 + 255) / 256; if (nb > wide) nb = wide; if (nb < 1) nb = 1;
            hipLaunchKernelGGL(cache_rows_from_ptrs2_kernel, dim3((unsigned)nb), dim3(256), 0, st, c1->row_ptrs, c1->row_tier, out,
                               (long long)B, T, c1->host.dim, c1->host.codec, c2->host.codec);
        }
        if (R) {
            rc = interact_from_mixed_rows(B, T, c1->host.dim, x, x_stride, c1->row_ptrs, c1->row_tier, c1->host.codec, c2->host.codec,
                                          itself, R, st);
            if (rc) return rc;
        }
        if (sampled2) {
            c1->pending_batches++; c1->pending_requests += B;
            c2->pending_batches++; c2->pending_requests += B;
            volatile int *rep1 = reinterpret_cast<volatile int *>(c1->host_tomb);
            const bool c1_known_full = rep1 && rep1[3] == 1 && (long long)rep1[2] >= c1->last_flush_call && c1->last_flush_call < c1->batch_calls;
            if (!c1_known_full || c1->pending_batches >= kCloseEvery || c2->pending_batches >= kCloseEvery || a1.rebuild || a2.rebuild)
                sampled_close_pending2(c1, a1.rebuild, c2, a2.rebuild, st);
            batch_housekeeping(c1, a1, st);
            batch_housekeeping(c2, a2, st);
        } else {
            if (c1->ft && c1->staged_mask) hipLaunchKernelGGL(cache_batch_unstage_kernel, dim3((unsigned)a1.g2), dim3(256), 0, st, a1);
            if (c2->ft && c2->staged_mask) hipLaunchKernelGGL(cache_batch_unstage_kernel, dim3((unsigned)a2.g2), dim3(256), 0, st, a2);
            batch_close(c1, a1, st);
            batch_close(c2, a2, st);
        }
        EVS_HIP_CHECK(hipGetLastError());
        return EVS_OK;
    }
    if (out) {
        long long nb = (B * T * (long long)c1->host.dim / 4 + 255) / 256; if (nb > wide) nb = wide; if (nb < 1) nb = 1;
        hipLaunchKernelGGL(cache_rows_from_ptrs2_kernel, dim3((unsigned)nb), dim3(256), 0, st, c1->row_ptrs, c1->row_tier, out,
                           (long long)B, T, c1->host.dim, c1->host.codec, c2->host.codec);
    }
    if (R && !fold2) {
        rc = interact_from_mixed_rows(B, T, c1->host.dim, x, x_stride, c1->row_ptrs, c1->row_tier, c1->host.codec, c2->host.codec,
                                      itself, R, st);
        if (rc) return rc;
    }
    if (sampled2) {
        // one update kernel per tier (per-position form: the probe wrote a record per (request, table) and tier); the
        // routing reads each tier's entry count, so the counters are folded every batch here
        if (sa2) {
            if (!launch_sa_update_pair(a1, a2, st)) {
                launch_sa_update(a1, st);
                launch_sa_update(a2, st);
            }
        } else {
        if (!a2.route_filter) { a2.other_slots = c1->bslots; a2.other_mask = (unsigned long long)(c1->bnslot - 1); }   // a key C1 just took is not inserted in C2 too
        if (!(a2.route_filter && launch_sampled_update_pair(a1, a2, st))) {
            launch_sampled_update(a1, st);
            launch_sampled_update(a2, st);
        }
        }
        if (c3 && !a1.c3_tags) hipLaunchKernelGGL(c3_batch_insert_lists_kernel, dim3(2 * kReplicas * 8), dim3(256), 0, st, a1, a2, tt.c3);   // what the two tiers evicted
        c1->pending_batches++; c1->pending_requests += B;
        c2->pending_batches++; c2->pending_requests += B;
        // The close only folds counters, and the one thing the next probe reads of them is "is C1 full" (the routing rule).
        // Once a close that ran after C1's last flush has reported it full -- nothing but a flush takes entries away -- the
        // counters are folded every kCloseEvery-th batch as in the single-tier path (5 us per batch become 0.6); while C1
        // fills, every batch.  (EVS_CACHE_LAZY2=0: every batch.)
        static const bool lazy2_on = !(getenv("EVS_CACHE_LAZY2") && getenv("EVS_CACHE_LAZY2")[0] == '0');
        volatile int *rep1 = reinterpret_cast<volatile int *>(c1->host_tomb);
        // (set-associative tiers: the routing reads the key's own set, not the entry count -- lazy from the first batch)
        const bool c1_known_full = sa2 || (lazy2_on && rep1 && rep1[3] == 1 && (long long)rep1[2] >= c1->last_flush_call && c1->last_flush_call < c1->batch_calls);
        if (!c1_known_full || c1->pending_batches >= kCloseEvery || c2->pending_batches >= kCloseEvery || a1.rebuild || a2.rebuild)
            sampled_close_pending2(c1, a1.rebuild, c2, a2.rebuild, st);
        if (!sa2) {
            batch_housekeeping(c1, a1, st);
            batch_housekeeping(c2, a2, st);
        }
        EVS_HIP_CHECK(hipGetLastError());
        return EVS_OK;
    }
    batch_policy(c1, a1, st);
    if (c3) hipLaunchKernelGGL(c3_batch_insert_kernel, dim3((unsigned)a1.g2), dim3(256), 0, st, a1, tt.c3);   // what C1 evicted
    a2.other_slots = c1->bslots; a2.other_mask = (unsigned long long)(c1->bnslot - 1);   // a key C1 just took is not inserted in C2 too
    batch_policy(c2, a2, st);
    if (c3) hipLaunchKernelGGL(c3_batch_insert_kernel, dim3((unsigned)a2.g2), dim3(256), 0, st, a2, tt.c3);   // what C2 evicted
    batch_close(c1, a1, st);
    batch_close(c2, a2, st);
    EVS_HIP_CHECK(hipGetLastError());
    return EVS_OK;
}

extern "C" int evs_cache_lookup_batch(evs_cache *c, int64_t B, const int32_t *rows, float *out, uint8_t *hit,
                                      void *stream) {
    using namespace evs;
    EVS_REQUIRE(out || B == 0, "evs_cache_lookup_batch: NULL out");
    return cache_batch_impl(c, B, rows, out, hit, nullptr, 0, 0, nullptr, stream);
}

extern "C" int evs_cache_lookup_interact(evs_cache *c, int64_t B, const int32_t *rows, const float *x, int64_t x_stride,
                                         int itself, float *R, uint8_t *hit, void *stream) {
    using namespace evs;
    EVS_REQUIRE((x && R) || B == 0, "evs_cache_lookup_interact: NULL x / R");
    EVS_REQUIRE(x_stride % 4 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0, "evs_cache_lookup_interact: x must be 16-byte aligned, stride % 4 == 0");
    return cache_batch_impl(c, B, rows, nullptr, hit, x, x_stride, itself, R, stream);
}

// out8 (host): [size, n_free, n_tomb, n_flush, n_evict, n_requests, n_perfect_hits, n_hits]; hist: n_tables+1 priority counts
extern "C" int evs_cache_batch_stats(evs_cache *c, int64_t *out8, int64_t *hist, void *stream) {
    using namespace evs;
    EVS_REQUIRE(c && out8 && c->bs, "evs_cache_batch_stats: the batched path has not been used");
    BatchState h;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    sampled_close_pending(c, 0, st);
    EVS_HIP_CHECK(hipStreamSynchronize(st));
    sampled_flush_if_wanted(c, st);   // what is reported is the state the next batch will see
    EVS_HIP_CHECK(hipMemcpyAsync(&h, c->bs, sizeof h, hipMemcpyDeviceToHost, st));
    EVS_HIP_CHECK(hipStreamSynchronize(st));
    out8[0] = h.count; out8[1] = h.n_free; out8[2] = h.n_tomb; out8[3] = h.n_flush; out8[4] = h.n_evict;
    out8[5] = h.n_requests; out8[6] = h.n_perfect_hits; out8[7] = h.n_hits;
    if (hist) for (int p = 0; p <= c->host.n_tables; p++) hist[p] = h.cnt[p];
    return EVS_OK;
}

// resident (priority, table_1based, row) triples, unordered (host); returns the count
extern "C" int64_t evs_cache_batch_dump(evs_cache *c, int64_t *triples, int64_t max_triples, void *stream) {
    using namespace evs;
    if (!c || !c->bs) return EVS_EINVAL;
    sampled_close_pending(c, 0, reinterpret_cast<hipStream_t>(stream));
    if (hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)) != hipSuccess) return EVS_EHIP;
    sampled_flush_if_wanted(c, reinterpret_cast<hipStream_t>(stream));
    if (hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)) != hipSuccess) return EVS_EHIP;
    const int64_t cap = c->host.cap;
    int64_t n = 0;
    if (c->batch_policy == 2 && c->sa.tags) {   // set-associative policy: (set, tag) is the key, the priority rides in the word
        const evs::SaGeom &g = c->sa;
        std::vector<unsigned> tags((size_t)g.nset * g.line_words);
        if (hipMemcpy(tags.data(), g.tags, tags.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return EVS_EHIP;
        for (unsigned es = 0; es < (g.nset << g.sub_shift); es++)
            for (unsigned w = 0; w < g.ways; w++) {
                const unsigned word = tags[(size_t)(es >> g.sub_shift) * g.line_words + g.w_off + (es & ((1u << g.sub_shift) - 1u)) * g.ways + w];
                if (!word) continue;
                const unsigned long long key = sa_key_of_host(c->sau, g, es, word);
                if (n < max_triples && triples) { triples[3 * n] = (int64_t)(word >> evs::kSaPrioShift); triples[3 * n + 1] = (int64_t)(key >> 32); triples[3 * n + 2] = (int64_t)(key & 0xffffffffull); }
                n++;
            }
        return n;
    }
    std::vector<unsigned long long> ekey(cap);
    std::vector<int> eagg(cap);
    if (hipMemcpy(ekey.data(), c->a.ekey, cap * 8, hipMemcpyDeviceToHost) != hipSuccess) return EVS_EHIP;
    if (hipMemcpy(eagg.data(), c->a.eagg, cap * 4, hipMemcpyDeviceToHost) != hipSuccess) return EVS_EHIP;
    for (int64_t e = 0; e < cap; e++) {
        const unsigned long long key = ekey[e] & evs::kKeyMask;
        if (!key) continue;   // (the sampled update keeps a batch stamp above the key)
        if (n < max_triples && triples) { triples[3 * n] = eagg[e]; triples[3 * n + 1] = (int64_t)(key >> 32); triples[3 * n + 2] = (int64_t)(key & 0xffffffffull); }
        n++;
    }
    return n;
}

// ---- a12: alt-key tier object --------------------------------------------------------------------
extern "C" int evs_aprx_destroy(evs_aprx *p) {
    if (!p) return EVS_OK;
    void *ptrs[] = {p->x.st, p->x.keys, p->x.slot_entry, p->x.ekey, p->x.ealt, p->x.eflag, p->x.free_stack, p->x.queue, p->tags, p->bstat};
    for (void *q : ptrs) if (q) (void)hipFree(q);
    delete p;
    return EVS_OK;
}

extern "C" int evs_aprx_create(evs_aprx **out, int64_t capacity, int n_tables) {
    using namespace evs;
    EVS_REQUIRE(out && capacity >= kAprxBatch && capacity < (1ll << 30) && n_tables >= 1 && n_tables <= kMaxTables,
                "evs_aprx_create: capacity must be >= %d (aprx_embedding.cpp:33) and n_tables <= %d", kAprxBatch, kMaxTables);
    evs_aprx *p = new evs_aprx();
    p->cap = capacity; p->n_tables = n_tables;
    long long nslot = 16;
    while (nslot < capacity * 2 + 8) nslot <<= 1;
    p->nslot = nslot;
    AprxState h{};
    h.cap = (int)capacity; h.n_free = (int)capacity; h.mask = (unsigned long long)(nslot - 1); h.qcap = 4 * capacity + 64;
#define EVS_ALLOC(ptr, bytes)                                                                   \
    do {                                                                                        \
        if (hipMalloc(reinterpret_cast<void **>(&(ptr)), (bytes)) != hipSuccess) {              \
            set_error("evs_aprx_create: hipMalloc(%lld bytes) failed", (long long)(bytes));     \
            evs_aprx_destroy(p);                                                                \
            return EVS_ENOMEM;                                                                  \
        }                                                                                       \
    } while (0)
    EVS_ALLOC(p->x.st, sizeof(AprxState));
    EVS_ALLOC(p->x.keys, nslot * 8);
    EVS_ALLOC(p->x.slot_entry, nslot * 4);
    EVS_ALLOC(p->x.ekey, capacity * 8);
    EVS_ALLOC(p->x.ealt, capacity * 4);
    EVS_ALLOC(p->x.eflag, capacity);
    EVS_ALLOC(p->x.free_stack, capacity * 4);
    EVS_ALLOC(p->x.queue, h.qcap * 8);
    p->nset = capacity / kSetWays > 0 ? capacity / kSetWays : 1;   // the batched form's sets: at most `capacity` members
    EVS_ALLOC(p->tags, p->nset * kSetWays * 8);
    EVS_ALLOC(p->bstat, 2 * 8);
#undef EVS_ALLOC
    EVS_HIP_CHECK(hipMemset(p->tags, 0, p->nset * kSetWays * 8));
    EVS_HIP_CHECK(hipMemset(p->bstat, 0, 2 * 8));
    EVS_HIP_CHECK(hipMemset(p->x.keys, 0, nslot * 8));
    std::vector<int> fs(capacity);
    for (int64_t i = 0; i < capacity; i++) fs[i] = (int)(capacity - 1 - i);
    EVS_HIP_CHECK(hipMemcpy(p->x.free_stack, fs.data(), capacity * 4, hipMemcpyHostToDevice));
    EVS_HIP_CHECK(hipMemcpy(p->x.st, &h, sizeof h, hipMemcpyHostToDevice));
    *out = p;
    return EVS_OK;
}

extern "C" int evs_aprx_set_altkeys(evs_aprx *p, const uint32_t *const *alt_tables, const int64_t *n_rows) {
    using namespace evs;
    EVS_REQUIRE(p && alt_tables && n_rows, "evs_aprx_set_altkeys: NULL argument");
    for (int k = 0; k < p->n_tables; k++) {
        EVS_REQUIRE(alt_tables[k] || n_rows[k] == 0, "evs_aprx_set_altkeys: table %d is NULL", k);
        p->x.alt_tables[k] = alt_tables[k];
        p->x.alt_rows[k] = n_rows[k];
    }
    p->has_alt = true;
    return EVS_OK;
}

extern "C" int evs_aprx_apply_ops(evs_aprx *p, int64_t n, const int32_t *ops, uint32_t *res, void *stream) {
    using namespace evs;
    EVS_REQUIRE(p && (n == 0 || (ops && res)) && n >= 0, "evs_aprx_apply_ops: bad argument");
    if (!p->has_alt) { set_error("evs_aprx_apply_ops: call evs_aprx_set_altkeys first"); return EVS_ESTATE; }
    if (n == 0) return EVS_OK;
    hipLaunchKernelGGL(aprx_ops_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), p->x, (long long)n, ops, res);
    EVS_HIP_CHECK(hipGetLastError());
    return EVS_OK;
}

// the FIFO front to back as (table_1based, row) pairs on the host, stale duplicates included; returns the length
extern "C" int64_t evs_aprx_dump_queue(evs_aprx *p, int64_t *pairs, int64_t max_pairs, void *stream) {
    using namespace evs;
    if (!p) { set_error("evs_aprx_dump_queue: NULL tier"); return EVS_EINVAL; }
    AprxState h;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (hipMemcpyAsync(&h, p->x.st, sizeof h, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return EVS_EHIP;
    std::vector<unsigned long long> q((size_t)h.qcap);
    if (hipMemcpy(q.data(), p->x.queue, (size_t)h.qcap * 8, hipMemcpyDeviceToHost) != hipSuccess) return EVS_EHIP;
    int64_t n = 0;
    for (long long i = h.qhead; i < h.qtail; i++, n++)
        if (pairs && n < max_pairs) {
            pairs[2 * n] = (int64_t)(q[(size_t)(i % h.qcap)] >> 32);
            pairs[2 * n + 1] = (int64_t)(q[(size_t)(i % h.qcap)] & 0xffffffffull);
        }
    return n;
}

// out4 (host): [size, n_hit, n_pending, error]
extern "C" int evs_aprx_stats(evs_aprx *p, int64_t *out4, void *stream) {
    using namespace evs;
    EVS_REQUIRE(p && out4, "evs_aprx_stats: NULL argument");
    AprxState h;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    EVS_HIP_CHECK(hipMemcpyAsync(&h, p->x.st, sizeof h, hipMemcpyDeviceToHost, st));
    EVS_HIP_CHECK(hipStreamSynchronize(st));
    out4[0] = h.count; out4[1] = h.n_hit; out4[2] = h.n_pending; out4[3] = h.error;
    return EVS_OK;
}

extern "C" int evs_cache_request_c1c2c3(evs_cache *c1, evs_cache *c2, evs_aprx *c3, int64_t B, const int32_t *rows,
                                        float *out, uint8_t *tier, int high_agghit_threshold, void *stream) {
    using namespace evs;
    EVS_REQUIRE(c1 && c2, "evs_cache_request_c1c2c3: NULL cache");
    if (c3) {
        if (c3->used == 2) { evs::set_error("evs_cache_request_c1c2c3: this alt-key tier is used through the batched path"); return EVS_ESTATE; }
        c3->used = 1;
    }
    EVS_REQUIRE(c1->host.policy == kEvLFU && c2->host.policy == kEvLFU, "evs_cache_request_c1c2c3: both tiers must be EvLFU");
    EVS_REQUIRE(c1->host.n_tables == c2->host.n_tables && c1->host.dim == c2->host.dim,
                "evs_cache_request_c1c2c3: the tiers disagree on n_tables/dim");
    if (!c1->has_backing || !c2->has_backing) { set_error("evs_cache_request_c1c2c3: set the backing tables of both tiers first"); return EVS_ESTATE; }
    if (c3 && !c3->has_alt) { set_error("evs_cache_request_c1c2c3: call evs_aprx_set_altkeys first"); return EVS_ESTATE; }
    EVS_REQUIRE(!c3 || c3->n_tables == c1->host.n_tables, "evs_cache_request_c1c2c3: alt-key tier has a different n_tables");
    if (B == 0) return EVS_OK;
    EVS_REQUIRE(B > 0 && rows && out && tier, "evs_cache_request_c1c2c3: NULL argument");
    if (c1->used == 2 || c2->used == 2) { set_error("evs_cache_request_c1c2c3: a tier is used through the batched path"); return EVS_ESTATE; }
    { const int p1 = serve_pause(c1), p2 = serve_pause(c2); if (p1 || p2) return EVS_EHIP; }
    c1->used = c2->used = 1;
    C1C2Args args;
    evs_cache *cs[2] = {c1, c2};
    TierArgs *ts[2] = {&args.t1, &args.t2};
    for (int t = 0; t < 2; t++) {
        ts[t]->st = cs[t]->st; ts[t]->a = cs[t]->a;
        for (int k = 0; k < kMaxTables; k++) { ts[t]->backing[k] = cs[t]->backing[k]; ts[t]->backing_rows[k] = cs[t]->backing_rows[k]; }
    }
    if (c3) args.c3 = c3->x; else { args.c3 = AprxArrays{}; args.c3.st = nullptr; }
    args.requests = rows; args.out = out; args.tier_out = tier; args.B = B; args.threshold = high_agghit_threshold;
    hipLaunchKernelGGL(cache_c1c2_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), args);
    EVS_HIP_CHECK(hipGetLastError());
    { const int r1 = exact_launched(c1, reinterpret_cast<hipStream_t>(stream)), r2 = exact_launched(c2, reinterpret_cast<hipStream_t>(stream)); if (r1 || r2) return EVS_EHIP; }
    return EVS_OK;
}
