// The batched cache path's hash (gfx950 only): constants, word layout and the read-only probe -- shared by evs_cache.hip
// (probe, update and housekeeping kernels) and evs_fused_rf.hip (the probe folded into the interaction kernel).
#pragma once
#include "evs_common.h"

namespace evs {

constexpr int kMaxBuckets = 65;     // EvLFU priorities 0..n_tables
constexpr unsigned long long kEmpty = 0ull;
__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
constexpr unsigned long long kTomb = ~0ull;
// second tombstone value (sampled policy update: tombstones written in batches of odd parity, see there); neither
// value can equal a key (table field 63)
constexpr unsigned long long kTomb1 = ~0ull - 1ull;
constexpr int kPending = -2;
// The batched path keeps its own hash: ONE 8-byte word per slot = key (38 bits: (table+1) << 32 | row)
// | entry index (26 bits), so a probe is a single random access and the table for 3.4 M entries is
// 128 MiB (load <= 0.25; it stays inside the 256 MiB Infinity Cache).  While a
// key is being inserted the entry field carries kFieldPend + its pending priority, so the duplicates of
// one batch fold their agg_hit into the word with a 64-bit atomicMax.  kEmpty = 0 and kTomb = ~0 are
// not valid keys (table+1 is in 1..32).
constexpr int kKeyBits = 38;
constexpr unsigned long long kKeyMask = (1ull << kKeyBits) - 1ull;
constexpr unsigned kFieldPend = (1u << (64 - kKeyBits)) - 128u;
constexpr long long kMaxBatchedCap = (long long)kFieldPend - 1;
__device__ __forceinline__ unsigned long long make_word(unsigned long long key, unsigned field) {
    return key | ((unsigned long long)field << kKeyBits);
}

// Read-only probe.  Found: the entry (or kPending) and end_slot = the key's slot.  Not found: -1 and end_slot = the
// first slot of the chain an insert of this key may take -- the first tombstone the walk passed, else the empty word
// it ended on (K2 starts its claim there instead of walking the chain again).
__device__ __forceinline__ int probe_ro(const unsigned long long *slots, unsigned long long mask, unsigned long long key,
                                        unsigned long long &end_slot, unsigned long long reusable_tomb = kTomb,
                                        bool *hint_is_tomb = nullptr) {
    unsigned long long i = mix64(key) & mask;
    long long first_tomb = -1;
    if (hint_is_tomb) *hint_is_tomb = false;
    // bounded: a batch with more unique new keys than free slots (tiny cache, huge batch) can leave the table without
    // an empty word until the next rebuild -- the walk then ends after one lap instead of never
    for (unsigned long long steps = 0; steps <= mask; steps++) {
        const unsigned long long w = slots[i];
        if ((w & kKeyMask) == key) {
            end_slot = i;
            const unsigned f = (unsigned)(w >> kKeyBits);
            return f >= kFieldPend ? kPending : (int)f;
        }
        if (w == kEmpty) {
            end_slot = first_tomb >= 0 ? (unsigned long long)first_tomb : i;
            if (hint_is_tomb) *hint_is_tomb = first_tomb >= 0;
            return -1;
        }
        if (w == reusable_tomb && first_tomb < 0) first_tomb = (long long)i;
        i = (i + 1) & mask;  // tombstones and other keys: keep walking
    }
    end_slot = first_tomb >= 0 ? (unsigned long long)first_tomb : i;
    if (hint_is_tomb) *hint_is_tomb = first_tomb >= 0;
    return -1;
}


// ---- set-associative form of the batched cache (batch policy 2, "setassoc") ------------------------------------------
// No hash chain and no entry arrays: key -> set = (mix64(key) >> 32) * nset >> 32, a set is kSaWays consecutive 8-byte
// words = 64 bytes of ONE 128-byte line, way w of set s owns arena row s * kSaWays + w.  A word = key (38 bits) | batch stamp
// (19 bits: the batch that filled the way) | priority (7 bits, the EvLFU agg_hit maximum) -- the priority sits in the top
// bits, so "raise the priority of this key" is one 64-bit atomicMax on the word.  A probe is one line, an insert is that
// line + one CAS + the row; nothing ever moves, so there are no tombstones, no sweeps and no rebuilds.  The victim of a
// new key is the lowest priority of ITS OWN set (free ways first), ways filled in the running batch excepted.
#ifndef EVS_SA_WAYS
#define EVS_SA_WAYS 8
#endif
constexpr int kSaWays = EVS_SA_WAYS;   // 8 or 16.  Measured at the 10 % Kaggle cache, B = 16 384: 16 ways (one whole 128-byte line per set) 42.2 us per
                                        // batch, hit rate 0.8839; 8 ways (64 bytes) 37.3 us, 0.8827 (sequential oracle 0.884): the set line is read by
                                        // one thread per key, 16-byte pieces through the CU's 64 B/clk vector-memory path, and that is what the 5 us are
constexpr int kSaPrioShift = 57;
constexpr unsigned kSaStampMask = (1u << (kSaPrioShift - kKeyBits)) - 1u;
constexpr unsigned long long kSaLowMask = (1ull << kSaPrioShift) - 1ull;
__device__ __forceinline__ unsigned sa_set_of(unsigned long long key, unsigned nset) {
    return (unsigned)(((mix64(key) >> 32) * (unsigned long long)nset) >> 32);
}
__device__ __forceinline__ unsigned long long sa_word(unsigned long long key, unsigned stamp, int prio) {
    return key | ((unsigned long long)(stamp & kSaStampMask) << kKeyBits) | ((unsigned long long)prio << kSaPrioShift);
}
__device__ __forceinline__ int sa_prio(unsigned long long w) { return (int)(w >> kSaPrioShift); }
__device__ __forceinline__ unsigned sa_stamp(unsigned long long w) { return (unsigned)(w >> kKeyBits) & kSaStampMask; }
// the set's line, as 16-byte loads that all go out before the first one is looked at
struct SaLine { ulonglong2 v[kSaWays / 2]; };
__device__ __forceinline__ void sa_load(const unsigned long long *tags, unsigned set, SaLine &l) {
    const ulonglong2 *p = reinterpret_cast<const ulonglong2 *>(tags + (unsigned long long)set * kSaWays);
#pragma unroll
    for (int j = 0; j < kSaWays / 2; j++) l.v[j] = p[j];
}
// does the set have a free way
__device__ __forceinline__ bool sa_has_free(const SaLine &l) {
    bool f = false;
#pragma unroll
    for (int j = 0; j < kSaWays / 2; j++) f = f || l.v[j].x == kEmpty || l.v[j].y == kEmpty;
    return f;
}
// way holding `key` (-1: none) and its word
__device__ __forceinline__ int sa_find(const SaLine &l, unsigned long long key, unsigned long long &word) {
    int way = -1;
    word = 0ull;
#pragma unroll
    for (int j = 0; j < kSaWays / 2; j++) {
        const bool a = (l.v[j].x & kKeyMask) == key, b = (l.v[j].y & kKeyMask) == key;
        way = a ? 2 * j : b ? 2 * j + 1 : way;
        word = a ? l.v[j].x : b ? l.v[j].y : word;
    }
    return way;
}

// entry of `key` in a set-associative tier (-1: not resident)
__device__ __forceinline__ int sa_lookup(const unsigned long long *tags, unsigned nset, unsigned long long key) {
    const unsigned set = sa_set_of(key, nset);
    SaLine l;
    sa_load(tags, set, l);
    unsigned long long w;
    const int way = sa_find(l, key, w);
    return way >= 0 ? (int)(set * (unsigned)kSaWays + (unsigned)way) : -1;
}

constexpr int kMaxTables = 64;      // one lane per table (exact path)

// Batched form of the alt-key tier C3 (see evs_cache.hip): a kSetWays-way set-associative set of key words
constexpr int kSetWays = 8;
constexpr unsigned long long kC3Flag = 1ull << kKeyBits;
struct C3Batch {
    unsigned long long *tags;   // nset x kSetWays key words; nullptr: no alt-key tier
    long long nset;
    long long *stat;            // [0] members, [1] alt hits served
    const unsigned *alt_tables[kMaxTables];
    long long alt_rows[kMaxTables];
};
__device__ __forceinline__ long long c3_find(const C3Batch &c3, unsigned long long key) {
    const long long base = (long long)(mix64(key * 0x9e3779b97f4a7c15ull) % (unsigned long long)c3.nset) * kSetWays;
    long long found = -1;
#pragma unroll
    for (int w = 0; w < kSetWays; w++)
        if ((c3.tags[base + w] & kKeyMask) == key) found = base + w;
    return found;
}


// Folded two-tier probe (evs_mixed.hip, PROBE variant of the (u8, u4) consumer): what cache_batch_probe2_kernel needs of
// each tier, by value
struct TierProbe {
    const unsigned long long *slots; unsigned long long mask;
    unsigned long long reusable_tomb;     // the tombstone value this batch may re-use (parity rule)
    int *eagg;
    const unsigned char *arena;
    int row_bytes;
    const unsigned char *backing[32];
    long long backing_rows[32];
    uint4 *miss_rec; int *list_cnt;       // per block: the misses routed to this tier (see BatchArgs::miss_rec)
    int *part1;                           // replica rows of the hit / histogram totals (32 x 40 ints)
    int hint_shift;
    const int *count; int cap, full_slack;   // entries resident at the last close; "full" = count >= cap - full_slack
    unsigned long long *tags; unsigned sa_nset;   // set-associative tier (sa_nset != 0): the sets' key words (slots / eagg unused)
};
struct Probe2Args {
    TierProbe t1, t2;
    const int *requests;                  // (B,T) int32 row ids
    unsigned char *tier_out;              // (B,T): 1 = C1 hit, 2 = C2 hit, 3 = alt-key hit, 0 = miss
    int threshold, T, list_cap;
    unsigned *route_filter; unsigned route_mask, route_stamp;   // keys routed to C1 are stamped here (evs_cache.hip: BatchArgs::route_filter)
    C3Batch c3;
};
// evs_mixed.hip: is there a (u8, u4) rows-in-registers consumer for the shape; the two-tier probe + the interaction over
// the rows it finds as ONE launch (blocks of 16 samples: miss lists of 16 T records per block and tier)
bool mixed84_supported(int T, int d, int codec1, int codec2);
int probe2_interact_mixed84(long long B, int T, int d, const float *x, long long x_stride, const Probe2Args &probe, int itself,
                            float *R, hipStream_t st);

// Folded probe (evs_fused_rf.hip, PROBE variant): what cache_batch_probe_gather_kernel needs of a cache, by value
struct ProbeArgs {
    const unsigned long long *slots; unsigned long long mask;
    unsigned long long reusable_tomb;     // the tombstone value this batch may re-use (parity rule)
    int *eagg;                            // priorities (monotone max of agg_hit)
    const int *requests;                  // (B,T) int32 row ids
    unsigned char *hit;                   // (B,T) out, may be NULL
    uint4 *miss_rec; int *list_cnt; int list_cap;   // per block: its misses as 16-byte records (see BatchArgs::miss_rec)
    int *part1;                           // replica rows of the hit / histogram totals
    int hint_shift, T;
    unsigned long long *tags; unsigned sa_nset;   // set-associative form (sa_nset != 0): the sets' key words instead of slots / eagg
};

// evs_fused.hip: interaction over x + the T rows the cache serves, the probe folded into the kernel (fp32 rows; is there a
// kernel for the shape: fused_row_ids_supported)
int fused_probe_interact(int64_t B, int T, int d, const float *x, int64_t x_stride, const ProbeArgs &probe, const void *arena,
                         const void *const *tables, const long long *table_rows, int itself, float *R, hipStream_t st);

}  // namespace evs
