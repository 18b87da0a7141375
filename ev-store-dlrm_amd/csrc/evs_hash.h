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
// No hash chain and no entry arrays.  Round 4: the way words are 32 bits and a tier PAIR shares its set records.
//   key universe   gid = row_base[table] + row (a dense number < N = all rows of all tables), x = P(gid) with P a
//                  bijection of [0, 2^b), 2^b >= N (two rounds of odd multiply + xorshift on b bits)
//   set / tag      set = x % nset, tag = x / nset: (set, tag) IS the key -- quotienting, no false positives, and the tag is
//                  only log2(2^b / nset) bits (9 for the 2 % Kaggle tiers), so a way fits a 32-bit word:
//                  [31:26] priority (EvLFU agg_hit maximum: "raise the priority" = one 32-bit atomicMax)
//                  [25:tb] batch stamp (the batch that filled the way, mod 2^(26-tb)) | [tb-1:0] tag + 1 (0 = empty way)
//   record         `line_words` consecutive words per set index.  A tier keeps 2^sub_shift SUB-SETS of `ways` ways in a record
//                  (words [w_off + sub * ways, + ways)); the sub-set is the low bits of the quotient x / nset, the tag the
//                  rest: effective set es = set * 2^sub_shift + sub, way w of it owns arena row es * ways + w.
//                  One tier alone: one sub-set of 8 ways = 32 bytes.  A C1 + C2 pair that starts out together: ONE 128-byte
//                  record per set index -- C1's 8 ways, then C2's (two 8-way sub-sets for the reference's 1 : 2 split,
//                  evlfu_8.cpp:63-78) -- so a key's two tier probes are one line request instead of two, and each tier
//                  still searches 8 ways.
// A probe is one line, an insert is that line + one CAS + the row; nothing ever moves: no tombstones, sweeps, rebuilds.
// The victim of a new key is the lowest priority of ITS OWN set (free ways first), ways filled in the running batch excepted.
//   two-copy arena (round 5, `dual` = 1: a tier alone): every way owns TWO arena rows and bit 25 of its word (`sel`, taken from
//                  the stamp field) says which one is live.  A replacement writes the new row into the OTHER copy and flips
//                  the bit with the CAS that installs the word, so the row a reader found through the old word is never
//                  overwritten by the replacement that retires it -- what lets a batch's policy update run INSIDE the launch
//                  that is still probing (evs_fused_rf.hip: the thread that misses a key inserts it): a prober that read the
//                  old word reads the old copy, intact; one that reads the new word sees the running batch's stamp, treats
//                  the way as a miss (the row may not be there yet) and serves the key from its table.
constexpr int kSaMaxWays = 16;          // per tier (4 x 16-byte pieces)
constexpr int kSaPrioShift = 26;
constexpr unsigned kSaLowMask = (1u << kSaPrioShift) - 1u;
constexpr unsigned kSaMul1 = 0x9E3779B1u, kSaMul2 = 0x85EBCA6Bu;   // odd: bijections mod 2^b
constexpr unsigned kSaInv1 = 0x0E8B2F51u, kSaInv2 = 0xA5CB9243u;   // their inverses mod 2^32 (checked at build time below)
static_assert((unsigned)(kSaMul1 * kSaInv1) == 1u && (unsigned)(kSaMul2 * kSaInv2) == 1u, "modular inverses");
struct SaUniverse {                     // the key universe of a cache (or of a pair): by value in kernel arguments
    unsigned row_base[32];              // gid of row 0 of table t
    unsigned mask;                      // 2^b - 1
    unsigned half;                      // xorshift distance, 2 * half >= b (so the xorshift is its own inverse)
    int n_tables;
};
struct SaGeom {                         // one tier's view of the set records
    unsigned *tags;                     // nset records of line_words words (nullptr: not a set-associative tier)
    unsigned nset;
    unsigned div_m, div_l;              // x / nset = (umulhi(x, div_m) + ((x - umulhi(x, div_m)) >> 1)) >> (div_l - 1)   (nset >= 2)
    unsigned tag_mask;                  // (1 << tb) - 1
    unsigned tag_bits;
    unsigned line_words, w_off, ways;
    unsigned sub_shift;                 // log2 of the sub-sets per record (0: one)
    unsigned dual;                      // 1: two arena rows per way, bit 25 of the word selects the live one (see above); else 0
    unsigned stamp_mask;                // (1 << stamp bits) - 1, stamp bits = 26 - dual - tag_bits
};
constexpr unsigned kSaSelShift = 25;
constexpr unsigned kSaNoStamp = 0xffffffffu;   // "no update in flight" (a stamp field never holds it)
// gid = row_base[table] + row.  (The kernels keep row_base in an LDS table: a per-lane index into the kernel arguments is a
// vector-memory round trip in front of the set loads.)
__device__ __forceinline__ unsigned sa_perm(const SaUniverse &u, unsigned gid) {
    unsigned x = gid;
    x = (x * kSaMul1) & u.mask; x ^= x >> u.half;
    x = (x * kSaMul2) & u.mask; x ^= x >> u.half;
    return x;
}
__device__ __forceinline__ unsigned sa_unperm(const SaUniverse &u, unsigned x) {   // -> gid
    x ^= x >> u.half; x = (x * kSaInv2) & u.mask;
    x ^= x >> u.half; x = (x * kSaInv1) & u.mask;
    return x;
}
// x -> (set index = x % nset, quotient x / nset); tiers of a pair share nset: one division for both
__device__ __forceinline__ void sa_divmod(const SaGeom &g, unsigned x, unsigned &set, unsigned &q) {
    q = x;
    if (g.nset > 1u) {
        const unsigned hi = __umulhi(x, g.div_m);
        q = (hi + ((x - hi) >> 1)) >> (g.div_l - 1u);
    }
    set = x - q * g.nset;
}
// (set index, quotient) -> (effective set, tag + 1) of one tier
__device__ __forceinline__ void sa_place(const SaGeom &g, unsigned set, unsigned q, unsigned &es, unsigned &tag1) {
    es = (set << g.sub_shift) | (q & ((1u << g.sub_shift) - 1u));
    tag1 = (q >> g.sub_shift) + 1u;
}
// x -> (effective set, tag + 1)
__device__ __forceinline__ void sa_split(const SaGeom &g, unsigned x, unsigned &es, unsigned &tag1) {
    unsigned set, q;
    sa_divmod(g, x, set, q);
    sa_place(g, set, q, es, tag1);
}
// sel: which arena copy the word's row lives in (dual geometries; always 0 otherwise)
__device__ __forceinline__ unsigned sa_word(const SaGeom &g, unsigned tag1, unsigned stamp, int prio, unsigned sel = 0u) {
    return tag1 | ((stamp & g.stamp_mask) << g.tag_bits) | ((sel & g.dual) << kSaSelShift) | ((unsigned)prio << kSaPrioShift);
}
__device__ __forceinline__ int sa_prio(unsigned w) { return (int)(w >> kSaPrioShift); }
__device__ __forceinline__ unsigned sa_stamp(const SaGeom &g, unsigned w) { return (w >> g.tag_bits) & g.stamp_mask; }
__device__ __forceinline__ unsigned sa_cur_stamp(const SaGeom &g, int stamp) { return (unsigned)stamp & g.stamp_mask; }
__device__ __forceinline__ unsigned sa_bump(unsigned w, int agg) { return (w & kSaLowMask) | ((unsigned)agg << kSaPrioShift); }
__device__ __forceinline__ unsigned sa_sel(const SaGeom &g, unsigned w) { return (w >> kSaSelShift) & g.dual; }
// arena row of (effective set, way) as the word w describes it
__device__ __forceinline__ unsigned sa_entry(const SaGeom &g, unsigned es, unsigned way, unsigned w) {
    return ((es * g.ways + way) << g.dual) | sa_sel(g, w);
}
// raise the priority of the way that held word w when the prober looked, to agg: a compare-and-swap, not a maximum -- a
// replacement that got there first must not be undone by the bumped OLD word winning a numeric comparison (with the update
// of the batch before running in the same launch that is a live race; without it the CAS succeeds first time).
// Returns the priority the raise replaced, or -1 when it changed nothing (the way was raised past agg, or replaced).
#ifdef EVS_X_LOG
static __device__ unsigned long long g_xlog[1 << 20];
static __device__ unsigned g_xlog_n;
__device__ __forceinline__ void xlog(unsigned type, unsigned *wp, unsigned oldw, unsigned neww) {
    const unsigned i = atomicAdd(&g_xlog_n, 1u);
    if (i < (1u << 18)) { g_xlog[4 * i] = type; g_xlog[4 * i + 1] = (unsigned long long)wp; g_xlog[4 * i + 2] = oldw; g_xlog[4 * i + 3] = neww; }
}
#else
__device__ __forceinline__ void xlog(unsigned, unsigned *, unsigned, unsigned) {}
#endif
__device__ __forceinline__ int sa_raise(const SaGeom &g, unsigned *wp, unsigned w, int agg) {
    for (int spin = 0; spin < 64; spin++) {
        const unsigned prev = atomicCAS(wp, w, sa_bump(w, agg));
        if (prev == w) { xlog(1, wp, w, sa_bump(w, agg)); return sa_prio(w); }
        if (((prev ^ w) & kSaLowMask) != 0u || sa_prio(prev) >= agg) return -1;   // another key's word, or high enough already
        w = prev;   // the same key, raised by somebody else but still below agg
    }
    return -1;
}
// a tier's ways of one set: 16-byte loads that all go out before the first one is looked at.
// W (template): the tier's way count when the caller knows it at compile time (8 or 16: the one-tier form, the 8 + 16 pair) --
// exactly W / 4 loads and W-way loops; W = 0: any geometry (kSaMaxWays-way loops masked by g.ways; the pieces past the tier's
// last one re-read it UNCONDITIONALLY: a conditional load is control flow around it with a wait each).
struct SaLine { uint4 v[kSaMaxWays / 4]; };
__device__ __forceinline__ unsigned *sa_ways_ptr(const SaGeom &g, unsigned es) {   // es: effective set (sa_split)
    return g.tags + (unsigned long long)(es >> g.sub_shift) * g.line_words + g.w_off + (es & ((1u << g.sub_shift) - 1u)) * g.ways;
}
template <int W = 0>
__device__ __forceinline__ void sa_load(const SaGeom &g, unsigned set, SaLine &l) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) u32x4 *gp_t;
    const gp_t p = reinterpret_cast<gp_t>(reinterpret_cast<uintptr_t>(sa_ways_ptr(g, set)));
    if constexpr (W > 0) {
#pragma unroll
        for (int j = 0; j < W / 4; j++) { const u32x4 v = p[j]; l.v[j] = make_uint4(v.x, v.y, v.z, v.w); }
    } else {
        const unsigned last = ((g.ways + 3u) >> 2) - 1u;
#pragma unroll
        for (int j = 0; j < kSaMaxWays / 4; j++) {
            const u32x4 v = p[(unsigned)j < last ? (unsigned)j : last];
            l.v[j] = make_uint4(v.x, v.y, v.z, v.w);
        }
    }
}
__device__ __forceinline__ unsigned sa_way_word(const SaLine &l, int j) {   // j: compile-time constant in unrolled loops
    const uint4 &v = l.v[j >> 2];
    return (j & 3) == 0 ? v.x : (j & 3) == 1 ? v.y : (j & 3) == 2 ? v.z : v.w;
}
// does the set have a free way (among this tier's ways)
template <int W = 0>
__device__ __forceinline__ bool sa_has_free(const SaGeom &g, const SaLine &l) {
    bool f = false;
#pragma unroll
    for (int j = 0; j < (W > 0 ? W : kSaMaxWays); j++) f = f || ((W > 0 || (unsigned)j < g.ways) && sa_way_word(l, j) == 0u);
    return f;
}
// way holding tag1 (-1: none) and its word.  pend: the stamp of an update in flight (kSaNoStamp: none) -- a way it has
// filled is not there yet for this reader
template <int W = 0>
__device__ __forceinline__ int sa_find(const SaGeom &g, const SaLine &l, unsigned tag1, unsigned &word, unsigned pend = kSaNoStamp) {
    int way = -1;
    word = 0u;
#pragma unroll
    for (int j = 0; j < (W > 0 ? W : kSaMaxWays); j++) {
        const unsigned w = sa_way_word(l, j);
        const bool m = (W > 0 || (unsigned)j < g.ways) && (w & g.tag_mask) == tag1 && sa_stamp(g, w) != pend;   // (tag1 >= 1: an empty way never matches)
        way = m ? j : way;
        word = m ? w : word;
    }
    return way;
}
// entry (arena row) of key (t, row) in a set-associative tier, -1: not resident
template <int W = -1>   // W: the tier's way count when the caller knows it (-1: looked up in g)
__device__ __forceinline__ int sa_lookup(const SaUniverse &u, const SaGeom &g, int t, unsigned row) {
    unsigned set, tag1, w;
    sa_split(g, sa_perm(u, u.row_base[t & 31] + row), set, tag1);
    SaLine l;
    int way;
    if constexpr (W > 0) { sa_load<W>(g, set, l); way = sa_find<W>(g, l, tag1, w); }
    else if (g.ways == 8u) { sa_load<8>(g, set, l); way = sa_find<8>(g, l, tag1, w); }
    else { sa_load<0>(g, set, l); way = sa_find<0>(g, l, tag1, w); }
    return way >= 0 ? (int)sa_entry(g, set, (unsigned)way, w) : -1;
}
// the key (table_1based << 32 | row) a way word of set `set` stands for
__device__ __forceinline__ unsigned long long sa_key_of(const SaUniverse &u, const SaGeom &g, unsigned es, unsigned w) {
    const unsigned q = (((w & g.tag_mask) - 1u) << g.sub_shift) | (es & ((1u << g.sub_shift) - 1u));
    const unsigned gid = sa_unperm(u, q * g.nset + (es >> g.sub_shift));
    int t = 0;
    for (int k = 1; k < u.n_tables; k++) t = gid >= u.row_base[k] ? k : t;
    return ((unsigned long long)(t + 1) << 32) | (gid - u.row_base[t]);
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
// first word of a key's set: the high half of the hash scaled into [0, nset) by a multiply (nset < 2^32; a 64-bit modulo by a
// run-time value is a ~150-instruction software division, in the head of every block of the three-tier probe)
__device__ __forceinline__ long long c3_base(unsigned long long key, long long nset) {
    return (long long)__umulhi((unsigned)(mix64(key * 0x9e3779b97f4a7c15ull) >> 32), (unsigned)nset) * kSetWays;
}
__device__ __forceinline__ long long c3_find(const C3Batch &c3, unsigned long long key) {
    const long long base = c3_base(key, c3.nset);
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
    SaGeom sa;                            // set-associative tier (sa.tags != nullptr): the set records (slots / eagg unused)
};
struct Probe2Args {
    TierProbe t1, t2;
    const int *requests;                  // (B,T) int32 row ids
    unsigned char *tier_out;              // (B,T): 1 = C1 hit, 2 = C2 hit, 3 = alt-key hit, 0 = miss
    int threshold, T, list_cap;
    unsigned *route_filter; unsigned route_mask, route_stamp;   // keys routed to C1 are stamped here (evs_cache.hip: BatchArgs::route_filter)
    C3Batch c3;
    SaUniverse sau;                       // set-associative tiers: the key universe both share
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
    SaGeom sa; SaUniverse sau;            // set-associative form (sa.tags != nullptr): the set records instead of slots / eagg
    // The policy update folded in too (round 5; a set-associative fp32 tier alone with a two-copy arena): arena_w != nullptr --
    // a thread that misses a key claims a way of the key's set itself (stamp pend_stamp = this batch's) and the lanes that
    // gather the key's row store it into the arena; miss_rec / list_cnt may then be NULL (no lists, no update launch).
    // pend_stamp: ways carrying it are misses for this launch's probe (kSaNoStamp: none are).
    unsigned pend_stamp = kSaNoStamp;
    unsigned char *arena_w = nullptr; int row_bytes = 0;
    int *part2 = nullptr;                 // replica rows of the inserts' totals (as the update kernels write them)
    int xflags = 0;                       // developer A/B (timing only): 1 = no arena stores, 2 = no claims
};

// One missed key into its set (the per-record step of cache_batch_sa_list_kernel restated for callers that hold the source
// row's address; 8-way sets): the set's ways, a duplicate folds its priority, else ONE CAS on the way of the lowest priority
// (free ways first, lowest index among equals, ways filled with stamp `cur` excepted; a lost CAS refreshes that way from what
// it returned and ranks again).  The new row belongs in the copy the victim's word does not name; the CALLER copies it (the
// folded launch does that at its end).  s_delta: priority histogram moves; s_stat: [0] free ways taken, [1] evictions.
// Split in two so that the first CAS can travel under other work: sa_claim_issue ranks and sends it (false: nothing is on
// its way -- a duplicate, or no way to take), sa_claim_finish looks at what came back and carries on from there.
// (Ways of equal priority: the lowest index goes.  The way filled longest ago instead -- the stamp field as an age, the reference's
//  FIFO inside a bucket at batch granularity -- was built and measured: the same hit rate at full size (0.8785 against the sequential
//  oracle's 0.8839 either way) and 41 us per batch instead of 34, the folded launch's head sits at its 128 registers.)
struct SaPick { int best, bp, dup; unsigned bw, dw; };
__device__ __forceinline__ SaPick sa_pick8(const SaGeom &g, unsigned cur, const unsigned (&w)[8], unsigned tag1) {
    SaPick p{-1, 0x7fffffff, -1, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const bool is_dup = (w[j] & g.tag_mask) == tag1;
        p.dup = is_dup ? j : p.dup; p.dw = is_dup ? w[j] : p.dw;
        const int pj = w[j] == 0u ? -1 : sa_prio(w[j]);
        const bool cand = (w[j] == 0u || sa_stamp(g, w[j]) != cur) && pj < p.bp;
        p.best = cand ? j : p.best; p.bp = cand ? pj : p.bp; p.bw = cand ? w[j] : p.bw;
    }
    return p;
}
__device__ __forceinline__ bool sa_claim_issue(const SaGeom &g, unsigned cur, unsigned set, unsigned tag1, int agg, const unsigned (&w)[8],
                                               SaPick &pk, unsigned &prev, int *s_delta) {
    unsigned *tags = sa_ways_ptr(g, set);
    pk = sa_pick8(g, cur, w, tag1);
    if (pk.dup >= 0) {
        if (sa_prio(pk.dw) < agg) {
            const int old = sa_raise(g, &tags[pk.dup], pk.dw, agg);
            if (old >= 0) { atomicSub(&s_delta[old], 1); atomicAdd(&s_delta[agg], 1); }
        }
        return false;
    }
    if (pk.best < 0) return false;   // every way of the set was filled by this very update: the key is not kept
    prev = atomicCAS(&tags[pk.best], pk.bw, sa_word(g, tag1, cur, agg, sa_sel(g, pk.bw) ^ 1u));
    return true;
}
// -> the arena entry to fill, or -1 (a duplicate folded, or the key turned away)
__device__ __forceinline__ int sa_claim_finish(const SaGeom &g, unsigned cur, unsigned set, unsigned tag1, int agg, unsigned (&w)[8],
                                               SaPick pk, unsigned prev, int *s_delta, int *s_stat) {
    // (every CAS sent is looked at: the loop ends on a success, on "nothing to take", or -- a set under heavy fire: lost CASes
    //  to raises do not use up ways -- after 24 lost ones with the last answer checked)
#pragma unroll 1
    for (int attempt = 0; ; attempt++) {
        if (prev == pk.bw) {
            xlog(2, sa_ways_ptr(g, set) + pk.best, pk.bw, sa_word(g, tag1, cur, agg, sa_sel(g, pk.bw) ^ 1u));
            if (pk.bp >= 0) { atomicSub(&s_delta[pk.bp], 1); atomicAdd(&s_stat[1], 1); }
            else atomicAdd(&s_stat[0], 1);
            atomicAdd(&s_delta[agg], 1);
            return (int)sa_entry(g, set, (unsigned)pk.best, (sa_sel(g, pk.bw) ^ 1u) << kSaSelShift);
        }
        if (attempt == 24) return -1;
#pragma unroll
        for (int j = 0; j < 8; j++) w[j] = j == pk.best ? prev : w[j];
        if (!sa_claim_issue(g, cur, set, tag1, agg, w, pk, prev, s_delta)) return -1;
    }
}
__device__ __forceinline__ int sa_claim_way(const SaGeom &g, unsigned cur, unsigned set, unsigned tag1, int agg, const SaLine &line,
                                            int *s_delta, int *s_stat) {
    unsigned w[8];
#pragma unroll
    for (int j = 0; j < 8; j++) w[j] = sa_way_word(line, j);
    SaPick pk; unsigned prev = 0u;
    if (!sa_claim_issue(g, cur, set, tag1, agg, w, pk, prev, s_delta)) return -1;
    return sa_claim_finish(g, cur, set, tag1, agg, w, pk, prev, s_delta, s_stat);
}

// evs_fused.hip: interaction over x + the T rows the cache serves, the probe folded into the kernel (fp32 rows; is there a
// kernel for the shape: fused_row_ids_supported)
int fused_probe_interact(int64_t B, int T, int d, const float *x, int64_t x_stride, const ProbeArgs &probe, const void *arena,
                         const void *const *tables, const long long *table_rows, int itself, float *R, hipStream_t st, int codec = 32);
bool fused_probe_codec_supported(int64_t B, int T, int d, int codec);   // a reduced-precision tier: is there a folded kernel

}  // namespace evs
