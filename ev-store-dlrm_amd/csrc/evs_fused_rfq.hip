// Fused embedding gather + pairwise-dot interaction over REDUCED-PRECISION tables (u16 / u8 / u4 codes), rows in flight
// in registers.
//
// Same computation and the same bits as emb_interact_dot_lds_kernel<CODEC, ..., BAG1> (evs_fused.hip):
//     R[b] = [ x[b] | strict-lower(T[b] T[b]^T) ],  T[b] = [x[b]; dec(W_0[idx_0[b]]); ...; dec(W_{F-2}[idx_{F-2}[b]])]
// (dlrm_s_pytorch.py:407-461 apply_emb over the decoded mixed-precision tables of mixed_precs_caching/evlfu_{16,8,4}.cpp,
//  one index per bag -- the Criteo collate, dlrm_data_pytorch.py:407-408 -- then :483-516 interact_features), one wavefront
// per sample, the same decoders (evs_common.h) and the same MFMA 16x16x4 f32 chains in the same order.
//
// Why a kernel of its own.  The LDS-DMA loop keeps ONE sample per wave in flight; with encoded rows that sample is 26 x 72
// (u16), 36 (u8) or 18 (u4) bytes, so a CU has half, a quarter, an eighth of the fp32 bytes on the wire and the launch is
// bound by memory LATENCY at a third of the HBM rate (B = 16 384, d = 36: u16 25.7 us where the fp32 launch takes 19.3 us
// for twice the bytes).  An encoded row is small enough to be fetched directly in the MFMA operand mapping -- lane
// (r16, q) of row r16 + 16 rr needs the CQ chunks [q CQ, (q + 1) CQ) of that row (CQ x {8, 4, 2} contiguous bytes: one
// load) and the REM trailing chunk (one more) -- so there is no transpose through LDS at all, and a d = 36 u16 sample in
// flight is 13 VGPRs (16 for fp32 in evs_fused_rf.hip).  A block owns one 16-sample chunk; every wave requests the rows of
// its 4 samples at once and consumes them in order under counted s_waitcnt vmcnt (straight-line code, see evs_fused_rf.hip).
// x (fp32) travels one float per lane and is spread to the operand mapping through a 256-byte LDS slot per wave.
#include "evs_fused.h"

#include <stdlib.h>
#include <type_traits>

namespace evs {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// blocks per CU the register allocation aims at.  Batches above 16 K samples are several generations of blocks, and what
// bounds them is the number of samples in flight per CU: 96 VGPRs = 5 blocks (u16), 80 = 6 (u8 / u4).  Measured at
// B = 65 536, d = 36 u16: 105 VGPRs (4 blocks) 73 us, 95 (5 blocks) 62.7 us.
#ifndef EVS_RFQ_LB16
#define EVS_RFQ_LB16 4
#endif
#ifndef EVS_RFQ_LB8
#define EVS_RFQ_LB8 6
#endif
#ifndef EVS_RFQ_LB4
#define EVS_RFQ_LB4 5   // u4, folded with aligned windows: 84 VGPRs in the lS_o form (6 blocks per CU = 80 would spill four; 5 vs 6 blocks measured equal)
#endif
#ifndef EVS_RFQ_LBP8
#define EVS_RFQ_LBP8 5   // the u8 PROBE form (a cache tier's head in front: set words, claims): at 80 registers it spills five; 5 blocks per
#endif                   // CU against 6: 31.4 vs 32.7 us per batch at B = 16 384, 94.3 vs 96.8 at 65 536 (4 blocks: 31.5 / 94.5)
constexpr int rfq_min_blocks(int codec, bool probe = false) {
    return codec == 16 ? EVS_RFQ_LB16 : codec == 4 ? EVS_RFQ_LB4 : probe ? EVS_RFQ_LBP8 : EVS_RFQ_LB8;
}
#ifndef EVS_RFQ_FOLD
#define EVS_RFQ_FOLD 1   // developer A/B: 0 = the tail chunk of a d = 36 u8 / u4 row as a load of its own
#endif
#ifndef EVS_RFQ_FOLD4A
#define EVS_RFQ_FOLD4A 1   // developer A/B: 0 = u4 windows at 4 q whatever the row's alignment
#endif
#ifndef EVS_RFQ_FOLD4
#define EVS_RFQ_FOLD4 1  // u4, one index per bag declared, F > 16: the folded form needs 83 VGPRs; at the 80 of six blocks per CU two
#endif                   // spills put scratch traffic and vmcnt(0) waits into the counted sequence (B = 65 536: 60.5 vs 55.0 us),
                         // so that one instantiation keeps the separate tail load; the other u4 forms fold
#ifndef EVS_RFQ_U4PAIR
#define EVS_RFQ_U4PAIR 1   // developer A/B: 0 = u4 decoded nibble by nibble through the 16-entry table
#endif
#ifndef EVS_OUT_CPOL
#define EVS_OUT_CPOL 2   // nt: R is written once and streams out (see evs_fused.hip)
#endif
#ifndef EVS_RFQ_I8
#define EVS_RFQ_I8 1     // u8, d = 36, F > 16: the row x row dot products on the INTEGER matrix pipe (see the I8 path below); 0 = fp32 chains
#endif
typedef int i32x4 __attribute__((ext_vector_type(4)));

// N raw bytes at p (global address space: a flat load would force every later wait to vmcnt(0)) -> w[0 .. max(1, N/4))
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
template <int N>
__device__ __forceinline__ void load_raw(unsigned long long p, unsigned (&w)[N >= 4 ? N / 4 : 1]) {
    if constexpr (N == 12) {
        const u32x3 v = *reinterpret_cast<const __attribute__((address_space(1), aligned(4))) u32x3 *>((uintptr_t)p);
        w[0] = v[0]; w[1] = v[1]; w[2] = v[2];
    } else if constexpr (N == 16) {
        const u32x4 v = *reinterpret_cast<const __attribute__((address_space(1))) u32x4 *>((uintptr_t)p);
        w[0] = v[0]; w[1] = v[1]; w[2] = v[2]; w[3] = v[3];
    } else if constexpr (N == 8) {
        const u32x2 v = *reinterpret_cast<const __attribute__((address_space(1))) u32x2 *>((uintptr_t)p);
        w[0] = v[0]; w[1] = v[1];
    } else if constexpr (N == 4) {
        w[0] = *reinterpret_cast<const __attribute__((address_space(1))) unsigned *>((uintptr_t)p);
    } else {
        static_assert(N == 2, "raw pieces of 16, 12, 8, 4 or 2 bytes");
        w[0] = *reinterpret_cast<const __attribute__((address_space(1))) unsigned short *>((uintptr_t)p);
    }
}

// the mirror of load_raw: N raw bytes w[...] -> p (the PROBE form copies the rows of missed keys into the tier's arena)
template <int N>
__device__ __forceinline__ void store_raw(unsigned long long p, const unsigned (&w)[N >= 4 ? N / 4 : 1]) {
    if constexpr (N == 12) {
        *reinterpret_cast<__attribute__((address_space(1), aligned(4))) u32x3 *>((uintptr_t)p) = (u32x3){w[0], w[1], w[2]};
    } else if constexpr (N == 16) {
        *reinterpret_cast<__attribute__((address_space(1), aligned(4))) u32x4 *>((uintptr_t)p) = (u32x4){w[0], w[1], w[2], w[3]};
    } else if constexpr (N == 8) {
        *reinterpret_cast<__attribute__((address_space(1), aligned(4))) u32x2 *>((uintptr_t)p) = (u32x2){w[0], w[1]};
    } else if constexpr (N == 4) {
        *reinterpret_cast<__attribute__((address_space(1), aligned(2))) unsigned *>((uintptr_t)p) = w[0];
    } else {
        static_assert(N == 2, "raw pieces of 16, 12, 8, 4 or 2 bytes");
        *reinterpret_cast<__attribute__((address_space(1))) unsigned short *>((uintptr_t)p) = (unsigned short)w[0];
    }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// two u16 codes of the main range (<= 65 000) -> two floats: the same value and the same three roundings as dec_u16_main
// (evs_common.h) -- (float)v - 32 500.0f is exact, like the integer subtraction there -- written on float pairs so that
// the subtraction, the product and the fma are one packed instruction each (v_pk_add/mul/fma_f32)
__device__ __forceinline__ f32x2 dec_u16_main2(unsigned w) {
    f32x2 f = {(float)(w & 0xffffu), (float)(w >> 16)};
    f = f - 32500.0f;
    const f32x2 lo = f * 5.052425e-13f;
    return __builtin_elementwise_fma(f, (f32x2){2e-05f, 2e-05f}, lo);
}
// one chunk (4 elements).  FAST (u16 only): the caller has checked that no lane of the wave holds a tail code (> 65 000)
template <int CODEC, bool FAST>
__device__ __forceinline__ float4 dec_chunk_q(unsigned w0, unsigned w1, const float *lut) {
    if constexpr (CODEC == 16 && FAST) {
        const f32x2 a = dec_u16_main2(w0), b = dec_u16_main2(w1);
        return make_float4(a[0], a[1], b[0], b[1]);
    } else if constexpr (CODEC == 16) {
        return make_float4(dec_code<16>(w0 & 0xffffu, lut), dec_code<16>(w0 >> 16, lut), dec_code<16>(w1 & 0xffffu, lut), dec_code<16>(w1 >> 16, lut));
    } else {
        return dec_chunk<CODEC>(w0, w1, lut);
    }
}
// element e (0..3, per lane) of a chunk
template <int CODEC, bool FAST>
__device__ __forceinline__ float dec_elem_q(unsigned w0, unsigned w1, int e, const float *lut) {
    if constexpr (CODEC == 16) {
        const unsigned v = (((e & 2) ? w1 : w0) >> (16 * (e & 1))) & 0xffffu;
        if constexpr (FAST) return dec_u16_main(v); else return dec_code<16>(v, lut);
    } else if constexpr (CODEC == 8) {
        return lut[(w0 >> (8 * e)) & 0xffu];
    } else {   // u4: element 2j is the HIGH nibble of byte j
        return lut[(w0 >> (8 * (e >> 1) + ((e & 1) ? 0 : 4))) & 15u];
    }
}

// CHECK (offsets given, whole batches: nnz == B, B or B + 1 offsets): the block also loads the offsets of its 16 samples and
// checks that every bag is exactly {idx[b]} (offsets[b] == b and the bag ends at b + 1) -- the Criteo collate's arange
// offsets then cost one more line per table and block, not a kernel.  A block's results depend on its own bags only, so a
// block that finds anything else pools ITS samples in a slow loop straight from global memory (general semantics: empty
// bags, several indices summed in index order, bad offsets / indices skipped and flagged -- the arithmetic of the general
// loop of evs_fused.hip) and feeds the same MFMA + output code.  No flag, no second launch.
// PROBE (round 4; a single reduced-precision tier -- the reference's one-layer evlfu_16 / _8 / _4 builds -- under the
// set-associative batch policy): the kernel reads the REQUEST rows and probes the tier's set records itself in its head, as
// emb_interact_rf_kernel<..., PROBE> does for an fp32 tier (evs_fused_rf.hip): hits read the arena (bit 30 of the tile
// entry), misses the table, hit flags / miss lists / statistics go where cache_batch_probe_gather_kernel would have put them.
template <int CODEC, int CQ, int REM, int NT, bool CHECK = false, bool PROBE = false>
__global__ void __launch_bounds__(256, rfq_min_blocks(CODEC, PROBE)) emb_interact_rfq_kernel(const FusedArgs args) {
    static_assert(!(CHECK && PROBE), "the offsets check belongs to the plain launch");
    // optimistic launches (offsets given, see offsets_arange_kernel in evs_fused.hip): this is the one-index-per-bag loop
    if constexpr (!CHECK && !PROBE) {
        if (args.opt_flag && *args.opt_flag == args.opt_id) return;
    }
    constexpr int NR = NT;
    constexpr int NC = CQ + REM;
    constexpr int d = 4 * (4 * CQ + REM);
    constexpr int NROWS = 16 * NT;
    constexpr int D = 4;                          // samples per wave, all in flight at once
    constexpr int kChunkBytes = CODEC / 2;        // 4 elements
    constexpr int row_bytes = d * CODEC / 8;
    constexpr int kMainBytes = CQ * kChunkBytes;  // this lane's contiguous share of a row
    constexpr int kMainDw = kMainBytes >= 4 ? kMainBytes / 4 : 1;
    constexpr int kRemDw = kChunkBytes >= 4 ? kChunkBytes / 4 : 1;
    // FOLD (d = 36, u8 / u4; round 3): the trailing chunk rides in the main load.  The launch's rate is set by the number of
    // random LINE REQUESTS, not by bytes (tools/sector_probe.hip: ~50 G lines/s from HBM whatever the size of the piece), and a
    // separate 4- or 2-byte load of the row's tail is a second request for the same line.  u8: every lane loads 12 bytes at
    // 8 q (k-slot 3: bytes 24..35 = its two chunks + the tail); u4: 8 bytes at 4 q, k-slot 3 at byte 10 (bytes 10..17: its
    // chunks in the upper half + the tail) -- never past the row.  The tail travels from k-slot 3 to the other three by
    // ds_bpermute at decode time.
    constexpr bool FOLD = REM == 1 && CQ == 2 && (CODEC == 8 || (CODEC == 4 && (EVS_RFQ_FOLD4 || CHECK || NT == 1))) && EVS_RFQ_FOLD;
    constexpr int kLoadDw = FOLD ? kMainDw + 1 : kMainDw;
    static_assert(d <= 64, "x travels one float per lane");
    // I8 (round 5; u8 rows, d = 36, F > 16): the u8 decoder is affine up to its roundings -- dec(u) = fl(fl(fl(u / 254) * 2) - 1),
    // within 6e-8 of (u - 127) / 127 = (s + 1) / 127 with s = u - 128 -- so for two u8 rows
    //     f_i . f_j = (P_ij + S_i + S_j + d) / 127^2,   P_ij = sum_k s_ik s_jk,  S_i = sum_k s_ik     (all exact in int32)
    // and P is ONE v_mfma_i32_16x16x64_i8 per 16 x 16 tile of pairs (the raw bytes XOR 0x80 are the operand: no decode at all;
    // k-slot q carries bytes [8 q, 8 q + 8), slot 3 the tail too), S_i rides along as the product with a row of ones (tile row
    // F, free for F <= 28).  Three integer MFMAs per sample replace 27 fp32 ones (1/32 of their matrix time each) and the
    // table look-ups of 18 elements per lane.  The 26 pairs with x (fp32) stay fp32: on the vector units, x . f_j =
    // sum_k (x_k / 127) (s_jk + 1) as packed fp32 fmas over this lane's 12 bytes per row, summed over the four k-slots.
    // One rounding for N / 127^2 (N < 2^24 converts exactly) against ~36 of the fp32 chain: closer to the exact value of the
    // reference's own arithmetic than the chain (worst case over 800 k random pairs 0.27 of rtol 1e-5 + atol 2e-6, the chain
    // 0.30: tests/test_gpu_parity.py, the bound in DESIGN.md 3.3).
    constexpr bool I8 = CODEC == 8 && FOLD && NT == 2 && EVS_RFQ_I8 != 0;
    __shared__ int s_S[I8 ? 4 : 1][32];            // I8: the rows' byte sums S_i of the sample in hand, per wave
    __shared__ int s_idx[PROBE ? 1024 : 512];     // [32 features][16 samples]: row id, sample id (x), -1 = no row (PROBE: a second plane,
                                                  // the arena entry a missed key's row is copied into on the way -- the update folded in)
    __shared__ __attribute__((aligned(16))) float s_x[4][64];
    __shared__ float s_lut[CodecLut<CODEC>::kEntries];
    // u4: a BYTE decodes to two elements at once (element 2j = the high nibble): one 8-byte table read and one byte
    // extraction per two elements instead of two reads and four shift / mask operations -- vector instructions add to the
    // matrix time on this part (docs/HISTORY.md 3.2c), and the nibble form made u4 slower than u8 for half the bytes
    constexpr bool PAIR4 = CODEC == 4 && EVS_RFQ_U4PAIR;
    __shared__ float2 s_lut2[PAIR4 ? 256 : 1];
    constexpr int OUT_MAX = ((d + NROWS * (NROWS + 1) / 2 + 63) / 64) * 64;
    __shared__ __attribute__((aligned(16))) float s_out[4][OUT_MAX + 16];

    const int lane = threadIdx.x & (kWave - 1);
    const int r16 = lane & 15;
    const int q = lane >> 4;
    const int F = args.F, itself = args.itself;
    const int out_row = d + args.P;
    const int64_t B = args.B;
    const FusedArgs *ka = (const FusedArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float *my_out = s_out[wave_in_block];
    float *my_x = s_x[wave_in_block];
    const unsigned long long zeros_p = (unsigned long long)reinterpret_cast<uintptr_t>(args.zeros);
    // a page of the code that decodes to exactly 0.0f (u16 32 500, u8 127, u4 7): what a lane reads for a row that does not
    // exist (no such feature, index out of range) -- no select behind the decoder
    const unsigned long long zc_p = (unsigned long long)reinterpret_cast<uintptr_t>(args.zero_codes);

    const int64_t blk_first = (int64_t)blockIdx.x * 16;
    const int64_t blk_end = blk_first + 16 < B ? blk_first + 16 : B;
    if (blk_first >= blk_end) return;       // block-uniform
    const int blk_n = (int)(blk_end - blk_first);
    const int n_samples = blk_n > wave_in_block ? (blk_n - wave_in_block + 3) / 4 : 0;

    // ---- index tile: thread e (and e + 256) owns element (feature e >> 4, sample e & 15) -------------------------------
    bool bad = false, my_ragged = false, fast_bad = false;
    if constexpr (PROBE) {
        // the cache probe folded in (the set-associative form of evs_fused_rf.hip's PROBE head): thread e owns key
        // (table (e >> 4) - 1, sample e & 15) and its twin e + 256
        __shared__ int s_agg[16];                 // hits per request of the chunk
        __shared__ int s_pdelta[kMaxBuckets];     // priority histogram moves
        __shared__ int s_psum[2];                 // hits / perfect requests
        __shared__ int s_nlist;                   // misses listed
        __shared__ unsigned s_nrows[32], s_sa_base[32];
        __shared__ int s_udelta[kMaxBuckets];     // the update folded in (ProbeArgs::arena_w, see evs_fused_rf.hip): the inserts' totals
        __shared__ int s_ustat[2];
        const ProbeArgs &pa = args.probe;
        const int T = pa.T;
        const bool ins = pa.arena_w != nullptr;   // block-uniform
        for (int i = threadIdx.x; i < kMaxBuckets; i += blockDim.x) { s_pdelta[i] = 0; s_udelta[i] = 0; }
        if (threadIdx.x < 2) s_ustat[threadIdx.x] = 0;
        if (threadIdx.x < 16) s_agg[threadIdx.x] = 0;
        if (threadIdx.x < 2) s_psum[threadIdx.x] = 0;
        if (threadIdx.x == 0) s_nlist = 0;
        if (threadIdx.x < 32) {   // per-table facts the probe indexes by LANE: out of LDS, not out of the kernel arguments
            s_nrows[threadIdx.x] = (unsigned)ka->n_rows[threadIdx.x];                     // feature f's rows (f = table + 1)
            s_sa_base[threadIdx.x] = ka->probe.sau.row_base[(threadIdx.x + 31) & 31];   // feature f = table f - 1
        }
        codec_lut_init<CODEC>(s_lut);
        if constexpr (PAIR4) s_lut2[threadIdx.x] = make_float2(u4_value(threadIdx.x >> 4), u4_value(threadIdx.x & 15u));
        __syncthreads();
        const int64_t bs = blk_first + (threadIdx.x & 15);
        int prow[2], pe[2], pprio[2], pway[2];
        bool pact[2], pok[2];
        unsigned pset[2], ptag[2], pw0[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int f = ((int)threadIdx.x >> 4) + 16 * h;
            pact[h] = f >= 1 && f < F && bs < blk_end;
            const int *rp = pact[h] ? pa.requests + bs * (int64_t)T + (f - 1) : reinterpret_cast<const int *>(args.dummy_i64);
            prow[h] = *reinterpret_cast<const __attribute__((address_space(1))) int *>(reinterpret_cast<uintptr_t>(rp));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int f = ((int)threadIdx.x >> 4) + 16 * h;
            pok[h] = pact[h] & (prow[h] >= 0) & ((unsigned)prow[h] < s_nrows[f & 31]);
            sa_split(pa.sa, sa_perm(pa.sau, s_sa_base[f & 31] + (pok[h] ? (unsigned)prow[h] : 0u)), pset[h], ptag[h]);
            if (!pok[h]) pset[h] = 0u;
        }
        SaLine line[2];
        unsigned lw[2][8];   // the two keys' set ways (kept for the claim of a missed key)
#pragma unroll
        for (int h = 0; h < 2; h++) sa_load<8>(pa.sa, pset[h], line[h]);
        __builtin_amdgcn_sched_barrier(0);   // both keys' set records in one round trip
#pragma unroll
        for (int h = 0; h < 2; h++) {
            unsigned w;
            const int way = sa_find<8>(pa.sa, line[h], ptag[h], w, pa.pend_stamp);
            const bool found = pok[h] && way >= 0;
            pe[h] = found ? (int)sa_entry(pa.sa, pset[h], (unsigned)way, w) : -1;
            pway[h] = way;
            pprio[h] = found ? sa_prio(w) : 0x7fffffff;
            pw0[h] = w;
#pragma unroll
            for (int j = 0; j < 8; j++) lw[h][j] = sa_way_word(line[h], j);
            if (found) atomicAdd(&s_agg[threadIdx.x & 15], 1);
        }
        __syncthreads();
        const int agg = s_agg[threadIdx.x & 15];
        // a missed key claims a way of its set here (evs_fused_rf.hip's PROBE head has the reasoning): the CAS travels with the raises
        int uwon[2] = {-1, -1};
        unsigned uprev[2] = {0u, 0u};
        SaPick upk[2] = {{-1, 0, -1, 0u, 0u}, {-1, 0, -1, 0u, 0u}};
        bool uwait[2] = {false, false};
        if (ins) {
#pragma unroll
            for (int h = 0; h < 2; h++)
                if (pok[h] && pe[h] < 0) uwait[h] = sa_claim_issue(pa.sa, pa.pend_stamp, pset[h], ptag[h], agg, lw[h], upk[h], uprev[h], s_udelta);
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int f = ((int)threadIdx.x >> 4) + 16 * h;
            if (pe[h] >= 0 && pprio[h] < agg) {   // monotone max like update_agg_hit
                const int old = sa_raise(pa.sa, sa_ways_ptr(pa.sa, pset[h]) + pway[h], pw0[h], agg);
                if (old >= 0) { atomicSub(&s_pdelta[old], 1); atomicAdd(&s_pdelta[agg], 1); }
            }
            int v = -1;
            if (f == 0) v = bs < blk_end ? (int)bs : -1;                 // x: the sample number
            else if (pact[h]) v = pe[h] >= 0 ? (int)(0x40000000u | (unsigned)pe[h]) : (pok[h] ? prow[h] : -1);
            s_idx[(int)threadIdx.x + 256 * h] = v;
            if (ins) {   // where the gathered row of a missed key goes (-1: nowhere)
                if (uwait[h]) uwon[h] = sa_claim_finish(pa.sa, pa.pend_stamp, pset[h], ptag[h], agg, lw[h], upk[h], uprev[h], s_udelta, s_ustat);
                s_idx[512 + (int)threadIdx.x + 256 * h] = uwon[h];
            }
            if (pact[h]) {
                const int64_t m = bs * (int64_t)T + (f - 1);
                if (pa.hit) pa.hit[m] = pe[h] >= 0;
                if (!pok[h]) bad = true;
                if (pa.miss_rec != nullptr && pok[h] && pe[h] < 0) {
                    const int at = atomicAdd(&s_nlist, 1);
                    pa.miss_rec[(int64_t)blockIdx.x * pa.list_cap + at] = make_uint4((unsigned)prow[h], (unsigned)(f - 1) | ((unsigned)agg << 8), pset[h], ptag[h]);
                }
            }
            if (f == 1 && bs < blk_end) { atomicAdd(&s_psum[0], agg); if (agg == T) atomicAdd(&s_psum[1], 1); }
        }
        __syncthreads();
        if (threadIdx.x < 40) {   // the block's totals into one of the replica rows (folded by the cache's close)
            const int i = threadIdx.x;
            const int v = i <= T ? s_pdelta[i] : i == 38 ? s_psum[0] : i == 39 ? s_psum[1] : 0;
            if (v) atomicAdd(&pa.part1[(blockIdx.x % 32) * 40 + i], v);
        }
        if (ins && threadIdx.x < 40) {   // the inserts' totals, as the update kernels leave them (folded by the cache's close)
            const int i = threadIdx.x;
            const int v = i <= T ? s_udelta[i] : i == 33 ? s_ustat[0] : i == 34 ? s_ustat[1] : 0;
            if (v) atomicAdd(&pa.part2[(blockIdx.x % 32) * 40 + i], v);
        }
        if (pa.list_cnt != nullptr && threadIdx.x == 0) pa.list_cnt[blockIdx.x] = s_nlist;
    } else
    {
        const int64_t bs = blk_first + (threadIdx.x & 15);
        int64_t v[2], o0[2], o1[2];
        bool own_end[2] = {false, false}, live_bag[2] = {false, false}, at_nnz[2] = {false, false};
        int64_t nzv[2] = {0, 0};
        int kind[2];
        unsigned nr[2];
        // the per-feature launch arguments of this wave's 4 + 4 features come through the scalar cache (wave-uniform
        // addresses: s_load) and are picked per lane -- a per-lane read of the kernarg segment would be one more
        // vector-memory round trip in front of the index loads
        const int sel = lane >> 4;
        typedef long long ll4 __attribute__((ext_vector_type(4)));
        const long long m0 = -(long long)(sel == 0), m1 = -(long long)(sel == 1), m2 = -(long long)(sel == 2), m3 = -(long long)(sel == 3);
        auto pick = [&](const int64_t *arr4) -> int64_t {   // arr4[sel], arr4 wave-uniform: one s_load_dwordx8, a bit blend
            const ll4 a = *reinterpret_cast<const ll4 *>(arr4);   // (selects here become branches around the loads)
            return (int64_t)((a[0] & m0) | (a[1] & m1) | (a[2] & m2) | (a[3] & m3));
        };
        // (every scalar load first, unconditionally and back to back: one wait for all of them)
        int64_t k_ip[2], k_nr[2], k_op[2] = {0, 0}, k_ol[2] = {0, 0}, k_nz[2] = {0, 0};
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int f0 = 4 * wave_in_block + 16 * h;      // wave-uniform; this lane's feature is f0 + sel (< EVS_MAX_FEATURES)
            k_ip[h] = pick(reinterpret_cast<const int64_t *>(&ka->indices[f0]));
            k_nr[h] = pick(&ka->n_rows[f0]);
            if constexpr (CHECK) {
                k_op[h] = pick(reinterpret_cast<const int64_t *>(&ka->offsets[f0]));
                k_ol[h] = pick(&ka->off_len[f0]);
                k_nz[h] = pick(&ka->nnz[f0]);
            }
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int f = 4 * wave_in_block + 16 * h + sel;
            const bool tab = f >= 1 && f < F;
            const int64_t *ip = tab ? reinterpret_cast<const int64_t *>(k_ip[h]) : nullptr;
            kind[h] = f >= F ? 0 : (f == 0 ? 1 : 2);
            nr[h] = tab ? (unsigned)k_nr[h] : 0u;
            const int64_t *ap = (ip && bs < blk_end) ? ip + bs : args.dummy_i64;
            v[h] = *reinterpret_cast<const __attribute__((address_space(1))) int64_t *>(reinterpret_cast<uintptr_t>(ap));
            if constexpr (CHECK) {
                // bag b ends where bag b + 1 starts: that is the neighbour lane's offset, except behind the block's last sample
                const bool on = ip && bs < blk_end;
                const int64_t *op = reinterpret_cast<const int64_t *>(k_op[h]);
                const int64_t ol = k_ol[h], nz = k_nz[h];
                const bool own = on && ((threadIdx.x & 15) == 15 || bs + 1 >= blk_end);
#ifdef EVS_XQ_NOOFF   // developer A/B (timing only): the check without its loads
                const int64_t *p0 = args.dummy_i64 + (op == nullptr), *p1 = args.dummy_i64 + (ol == 1);
#else
                const int64_t *p0 = on ? op + bs : args.dummy_i64;
                const int64_t *p1 = (own && bs + 1 < ol) ? op + bs + 1 : args.dummy_i64;
#endif
                o0[h] = *reinterpret_cast<const __attribute__((address_space(1))) int64_t *>(reinterpret_cast<uintptr_t>(p0));
                o1[h] = *reinterpret_cast<const __attribute__((address_space(1))) int64_t *>(reinterpret_cast<uintptr_t>(p1));
                own_end[h] = own; live_bag[h] = on; at_nnz[h] = own && !(bs + 1 < ol); nzv[h] = nz;
            }
        }
        if constexpr (CHECK) {
            // a bag ends where the next one starts = the next lane's o0, which that lane checks itself: no exchange between lanes
#pragma unroll
            for (int h = 0; h < 2; h++) {
                if (at_nnz[h]) o1[h] = nzv[h];              // the last bag ends at nnz
                const bool ok = (o0[h] == bs) & (!own_end[h] | (o1[h] == bs + 1));
                my_ragged |= live_bag[h] & !ok;
            }
        }
        codec_lut_init<CODEC>(s_lut);
        if constexpr (PAIR4) s_lut2[threadIdx.x] = make_float2(u4_value(threadIdx.x >> 4), u4_value(threadIdx.x & 15u));   // (256 threads)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const bool live = kind[h] != 0 && bs < blk_end;
            const int64_t val = kind[h] == 2 ? v[h] : bs;     // x: the sample number
            const bool in_range = kind[h] == 1 || (uint64_t)val < (uint64_t)nr[h];
            // (CHECK: an index whose own bag is not {idx[b]} may sit at a position no bag refers to -- the slow loop, which
            //  this block then runs, has the verdict on it)
            if constexpr (CHECK) fast_bad |= live & !in_range;   // (counts only if the block stays on the one-index code)
            else bad |= live & !in_range;
            s_idx[(int)threadIdx.x + 256 * h] = (live & in_range) ? (int)val : -1;
        }
    }
    // ---- this lane's rows in the MFMA operand mapping: row r16 + 16 rr, chunks [q CQ, (q + 1) CQ) and the REM tail -------
    unsigned long long fbase[NR];
    unsigned fscale[NR];
#pragma unroll
    for (int rr = 0; rr < NR; rr++) {
        const int f = r16 + 16 * rr;
        // (row 0 -- x, fp32, spread from LDS below -- and the rows past F read the zero-code page like an absent row)
        const bool on = f >= 1 && f < F;
        fbase[rr] = on ? (unsigned long long)reinterpret_cast<uintptr_t>(ka->src[f]) : zc_p;
#ifdef EVS_X_STRIDE     // developer probe (tools/stride_probe.py; timing only): rows every EVS_X_STRIDE bytes
        fscale[rr] = on ? (unsigned)EVS_X_STRIDE : 0u;
#else
        fscale[rr] = on ? (unsigned)row_bytes : 0u;
#endif
    }
    const unsigned long long xbase = (unsigned long long)reinterpret_cast<uintptr_t>(ka->src[0]);
    const unsigned xscale = (unsigned)(ka->stride[0] * 4);
    bool blk_ragged = false;
#ifdef EVS_XQ_NOOR   // developer A/B (timing only): plain barrier, no verdict
    __syncthreads();
    if (my_ragged && args.B == 12345) bad = true;
#else
    if constexpr (CHECK) blk_ragged = __syncthreads_or(my_ragged); else __syncthreads();
#endif
#ifdef EVS_XQ_NOSLOW   // developer A/B (timing only): no slow loop in the kernel
    blk_ragged = false;
#endif
#ifdef EVS_XQ_NOOFF
    blk_ragged = blk_ragged && args.B == 12345;
#endif
    if (!blk_ragged) bad |= fast_bad;

    constexpr int kOob = 0x7ffffff0;
    auto flush_out = [&](int64_t bp, bool on) {
#ifdef EVS_XQ_NOSTORE   // developer A/B (timing only): R is never written
        on = false;
#endif
        float *Rb = args.R + (on ? bp : 0) * (int64_t)out_row;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(Rb, 0, on ? out_row * 4 : 0, 0x00020000);
        const int n4 = out_row >> 2;   // whole 16-byte pieces; the 0..3 trailing floats go as dwords
        // (an rfq launch has F <= kTileMaxF: at most d + 28 * 29 / 2 floats -- the generic 32-row bound would issue a
        //  third store that the bounds check always drops)
        constexpr int kFlushMaxRow = ((d + kTileMaxF * (kTileMaxF + 1) / 2 + 3) / 4) * 4;
#pragma unroll
        for (int h = 0; h < (kFlushMaxRow + 255) / 256; h++) {
            const int e4 = lane + 64 * h;
            const float4 v = reinterpret_cast<const float4 *>(my_out)[e4 < n4 ? e4 : 0];
            u32x4 u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
            __builtin_amdgcn_raw_buffer_store_b128(u, rs, e4 < n4 ? 16 * e4 : kOob, 0, EVS_OUT_CPOL);
        }
        {
            const int e = 4 * n4 + (lane & 3);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(my_out[e]), rs, lane < (out_row & 3) ? 4 * e : kOob, 0, EVS_OUT_CPOL);
        }
    };

    auto chunk_words = [&](const auto &w, int c, unsigned &w0, unsigned &w1) {
        if constexpr (CODEC == 16) { w0 = w[2 * c]; w1 = w[2 * c + 1]; }
        else if constexpr (CODEC == 8) { w0 = w[c]; w1 = 0; }
        else { w0 = (w[0] >> (16 * c)) & 0xffffu; w1 = 0; }
    };
    // row 0 of the operand tile is x: plain fp32 chunks, spread from one float per lane through the wave's LDS slot
    auto spread_x = [&](float4 (&a)[NR][NC]) {
        {
#pragma unroll
            for (int c = 0; c < CQ; c++) {
                const float4 xa = *reinterpret_cast<const float4 *>(my_x + (q * CQ + c) * 4);
                if (r16 == 0) a[0][c] = xa;
            }
#pragma unroll
            for (int t = 0; t < REM; t++) {
                const float xe = my_x[4 * CQ * 4 + 4 * t + q];
                if (r16 == 0) a[0][CQ + t].x = xe;
            }
        }
    };
    // the interaction of one sample (operands in the MFMA layout) and the staging of its output row
    auto interact_stage = [&](const float4 (&a)[NR][NC], float xv) {
        f32x4 c00 = {0.f, 0.f, 0.f, 0.f}, c10 = {0.f, 0.f, 0.f, 0.f}, c11 = {0.f, 0.f, 0.f, 0.f};
#ifdef EVS_XQ_NOMFMA   // developer A/B (timing only, wrong R): three adds per operand chunk instead of the matrix-core work
#pragma unroll
        for (int c = 0; c < NC; c++) { c00[c & 3] += a[0][c].x + a[0][c].y; c10[c & 3] += a[NR - 1][c].z; c11[c & 3] += a[NR - 1][c].w + a[0][c].z; }
#else
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const float e0[4] = {a[0][c].x, a[0][c].y, a[0][c].z, a[0][c].w};
            const float e1[4] = {a[NR - 1][c].x, a[NR - 1][c].y, a[NR - 1][c].z, a[NR - 1][c].w};
            if (c < CQ) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(e0[e], e0[e], c00, 0, 0, 0);
                    if constexpr (NT == 2) {
                        c10 = __builtin_amdgcn_mfma_f32_16x16x4f32(e1[e], e0[e], c10, 0, 0, 0);
                        c11 = __builtin_amdgcn_mfma_f32_16x16x4f32(e1[e], e1[e], c11, 0, 0, 0);
                    }
                }
            } else {   // the REM trailing chunks are held by all four k-slots; slot q contributes element q
                const float s0 = e0[0], s1 = e1[0];
                c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(s0, s0, c00, 0, 0, 0);
                if constexpr (NT == 2) {
                    c10 = __builtin_amdgcn_mfma_f32_16x16x4f32(s1, s0, c10, 0, 0, 0);
                    c11 = __builtin_amdgcn_mfma_f32_16x16x4f32(s1, s1, c11, 0, 0, 0);
                }
            }
        }
#endif
        // stage the output row: x passthrough, then the packed lower triangle straight from the accumulators;
        // never-stored elements go to a dump slot behind the row
        const int dump = 4 * (OUT_MAX + r16);
        *reinterpret_cast<float *>(reinterpret_cast<char *>(my_out) + (lane < d ? 4 * lane : dump)) = xv;
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const int i = 4 * q + v;
            const int zo00 = (i < F && r16 < i + itself) ? 4 * (d + (i * (i - 1 + 2 * itself)) / 2 + r16) : dump;
            *reinterpret_cast<float *>(reinterpret_cast<char *>(my_out) + zo00) = c00[v];
            if constexpr (NT == 2) {
                const int gi = 16 + i;
                const int base = (gi * (gi - 1 + 2 * itself)) / 2;
                const int zo10 = gi < F ? 4 * (d + base + r16) : dump;
                const int zo11 = (gi < F && 16 + r16 < gi + itself) ? 4 * (d + base + 16 + r16) : dump;
                *reinterpret_cast<float *>(reinterpret_cast<char *>(my_out) + zo10) = c10[v];
                *reinterpret_cast<float *>(reinterpret_cast<char *>(my_out) + zo11) = c11[v];
            }
        }
    };

    if constexpr (CHECK) {
        if (blk_ragged) {   // block-uniform, rare: this block's samples with general bag semantics, pooled straight from global memory
            for (int u = 0; u < n_samples; u++) {
                const int64_t b = blk_first + wave_in_block + 4 * (int64_t)u;
                float4 a[NR][NC];
#pragma unroll
                for (int rr = 0; rr < NR; rr++) {
                    const int f = r16 + 16 * rr;
#pragma unroll
                    for (int c = 0; c < NC; c++) a[rr][c] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (f < 1 || f >= F) continue;
                    const int64_t *ip = ka->indices[f], *op = ka->offsets[f];
                    const unsigned long long src = (unsigned long long)reinterpret_cast<uintptr_t>(ka->src[f]);
                    const int64_t nnz = ka->nnz[f];
                    int64_t s0 = op[b];
                    int64_t e0 = (b + 1 < ka->off_len[f]) ? op[b + 1] : nnz;
                    if (!((s0 >= 0) & (e0 >= s0) & (e0 <= nnz))) { bad = true; s0 = e0 = 0; }
                    const uint64_t n_rows = (uint64_t)ka->n_rows[f];
                    for (int64_t j = s0; j < e0; j++) {
                        const int64_t r = ip[j];
                        if ((uint64_t)r >= n_rows) { bad = true; continue; }   // skipped; a skipped FIRST row counts as zeros
                        const unsigned long long row = src + (uint64_t)r * (uint64_t)row_bytes;
                        unsigned wm[kMainDw], wr[REM > 0 ? REM : 1][kRemDw];
                        load_raw<kMainBytes>(row + q * kMainBytes, wm);
#pragma unroll
                        for (int t = 0; t < REM; t++) load_raw<kChunkBytes>(row + 4 * kMainBytes + t * kChunkBytes, wr[t]);
#pragma unroll
                        for (int c = 0; c < NC; c++) {
                            unsigned w0, w1;
                            float4 t4;
                            if (c < CQ) { chunk_words(wm, c, w0, w1); t4 = dec_chunk_q<CODEC, false>(w0, w1, s_lut); }
                            else {
                                w0 = wr[c - CQ][0]; w1 = CODEC == 16 ? wr[c - CQ][kRemDw - 1] : 0u;
                                t4 = make_float4(dec_elem_q<CODEC, false>(w0, w1, q, s_lut), 0.f, 0.f, 0.f);
                            }
                            if (j == s0) { a[rr][c] = t4; continue; }
                            a[rr][c].x = __fadd_rn(a[rr][c].x, t4.x); a[rr][c].y = __fadd_rn(a[rr][c].y, t4.y);
                            a[rr][c].z = __fadd_rn(a[rr][c].z, t4.z); a[rr][c].w = __fadd_rn(a[rr][c].w, t4.w);
                        }
                    }
                }
                const float xv = *reinterpret_cast<const float *>((uintptr_t)(xbase + (unsigned long long)b * (unsigned long long)xscale + 4 * (lane < d ? lane : 0)));
                my_x[lane] = xv;
                spread_x(a);
                interact_stage(a, xv);
                flush_out(b, true);
            }
            if (bad) atomicOr(args.err, 1);
            return;
        }
    }

    // ---- request the rows of all D samples of this wave ----------------------------------------------------------------
    unsigned rmain[D][NR][kLoadDw], rrem[D][NR][REM > 0 ? REM : 1][kRemDw];
    float rx[D];
    const unsigned fold_off = FOLD ? (CODEC == 8 ? 8u * (unsigned)q : (q < 3 ? 4u * (unsigned)q : 10u)) : 0u;
    // u4, folded: an 18-byte row starts 2 bytes off a dword boundary when its index is odd, and an 8-byte load off the
    // boundary is more than one request.  k-slots 0..2 therefore start their window 2 bytes EARLIER on odd rows (the
    // previous row's last two bytes: in bounds -- an odd row has a predecessor) and shift; k-slot 3 keeps bytes 10..17.
    constexpr bool FOLD4A = FOLD && CODEC == 4 && EVS_RFQ_FOLD4A;
    unsigned odd_rows = 0;   // bit u * NR + rr
#pragma unroll
    for (int u = 0; u < D; u++) {
        const int m = wave_in_block + 4 * u;              // block-local sample
        const unsigned phantom = u < n_samples ? 0u : 0xffffffffu;   // past this wave's samples: every lane reads the zero page
#pragma unroll
        for (int rr = 0; rr < NR; rr++) {
            const int iv = s_idx[(r16 + 16 * rr) * 16 + m];
            // branch-free (bit blends, no selects over the LDS reads -- see evs_fused_rf.hip): -1 -> the zero page
#ifdef EVS_XQ_NOROWS   // developer A/B (timing only, wrong R): every row is the zero-code page -- no random access at all
            const unsigned neg = 0xffffffffu | (unsigned)iv;
#else
            const unsigned neg = (unsigned)(iv >> 31) | phantom;
#endif
            unsigned idx = (unsigned)iv & ~neg;
            unsigned long long rbase = fbase[rr];
            if constexpr (PROBE) {   // bit 30: a row of the tier's arena (bit blend, no select over the LDS reads)
                const unsigned long long in_arena = 0ull - (unsigned long long)((idx >> 30) & 1u);
                rbase ^= (rbase ^ (unsigned long long)reinterpret_cast<uintptr_t>(args.arena)) & in_arena;
                idx &= 0x3fffffffu;
            }
            const unsigned long long p = rbase + (unsigned long long)idx * (unsigned long long)fscale[rr];
            const unsigned long long m64 = ((unsigned long long)neg << 32) | neg;
            const unsigned long long pa = p ^ ((p ^ zc_p) & m64);
            if constexpr (FOLD4A) {
                const unsigned par = idx & 1u;                    // (an absent row: index 0, the zero-code page)
                odd_rows |= par << (u * NR + rr);
                load_raw<4 * kLoadDw>(pa + fold_off - (q < 3 ? 2u * par : 0u), rmain[u][rr]);
            } else if constexpr (FOLD) {
                load_raw<4 * kLoadDw>(pa + fold_off, rmain[u][rr]);
            } else {
                load_raw<kMainBytes>(pa + q * kMainBytes, rmain[u][rr]);
#pragma unroll
                for (int t = 0; t < REM; t++) load_raw<kChunkBytes>(pa + 4 * kMainBytes + t * kChunkBytes, rrem[u][rr][t]);
            }
        }
        {
            const unsigned long long p = xbase + (unsigned long long)(unsigned)(blk_first + m) * (unsigned long long)xscale + 4 * (lane < d ? lane : 0);
#ifdef EVS_XQ_NOX      // developer A/B (timing only, wrong R): x is never read
            const unsigned long long m64 = ~0ull | p;
#else
            const unsigned long long m64 = ((unsigned long long)phantom << 32) | phantom;
#endif
            const unsigned long long pa = p ^ ((p ^ zeros_p) & m64);
            rx[u] = *reinterpret_cast<const __attribute__((address_space(1))) float *>((uintptr_t)pa);
        }
    }
    __builtin_amdgcn_sched_barrier(0);   // keep every request above the first consume


    const bool ins_blk = PROBE && args.probe.arena_w != nullptr;   // block-uniform: missed keys' rows go into the tier's arena on the way
#pragma unroll
    for (int u = 0; u < D; u++) {
        const int64_t b = blk_first + wave_in_block + 4 * (int64_t)u;   // wave-uniform
        if constexpr (PROBE) {
            if (ins_blk) {   // the rows of the ways this block claimed (s_idx's second plane): from the registers that gathered them
#pragma unroll
                for (int rr = 0; rr < NR; rr++) {
                    const int e = s_idx[512 + (r16 + 16 * rr) * 16 + wave_in_block + 4 * u];
                    if (e >= 0) {
                        const unsigned long long dst = (unsigned long long)reinterpret_cast<uintptr_t>(args.probe.arena_w) + (unsigned long long)(unsigned)e * (unsigned)row_bytes;
                        if constexpr (FOLD && CODEC == 8) {
                            store_raw<12>(dst + 8u * (unsigned)q, rmain[u][rr]);   // (bytes 8 q .. 8 q + 11: the overlaps carry the same bytes)
                        } else if constexpr (FOLD) {   // u4: k-slots 0..2 hold bytes 4 q .. 4 q + 3 in a window that starts 2 bytes early on odd rows
                            const unsigned par = FOLD4A ? (odd_rows >> (u * NR + rr)) & 1u : 0u;
                            const unsigned long long w01 = ((unsigned long long)rmain[u][rr][1] << 32) | rmain[u][rr][0];
                            const unsigned v0[1] = {q < 3 ? (unsigned)(w01 >> (16u * par)) : rmain[u][rr][0]};
                            store_raw<4>(dst + (q < 3 ? 4u * (unsigned)q : 10u), v0);
                            const unsigned v1[1] = {rmain[u][rr][1]};
                            if (q == 3) store_raw<4>(dst + 14u, v1);
                        } else {
                            store_raw<kMainBytes>(dst + (unsigned)(q * kMainBytes), rmain[u][rr]);
                            if (q == 3) {
#pragma unroll
                                for (int t = 0; t < REM; t++) store_raw<kChunkBytes>(dst + 4 * kMainBytes + t * kChunkBytes, rrem[u][rr][t]);
                            }
                        }
                    }
                }
            }
        }
        my_x[lane] = rx[u];
        if constexpr (I8) {
            flush_out(b - 4, u > 0 && u - 1 < n_samples);    // sample u-1 leaves under this sample's work
            // (three phases kept apart by scheduling barriers -- the x pairs, the integer products, the output row -- so that
            //  the twelve scaled x values, the operands and the accumulators are never all alive at once: the launch wants six
            //  blocks per CU = 80 registers, and a spill costs more than the fp32 chains did)
            // -- phase 1: the pairs with x, on the vector units: x' = x / 127 over this lane's bytes
            {
                float zx[2], xx = 0.f;
                float xs[12];
                const float4 xa = *reinterpret_cast<const float4 *>(my_x + 8 * q), xb = *reinterpret_cast<const float4 *>(my_x + 8 * q + 4);
                const float4 xt = *reinterpret_cast<const float4 *>(my_x + 32);
                const float r127 = 1.0f / 127.0f, rt = q == 3 ? r127 : 0.f;   // (the tail belongs to k-slot 3)
                xs[0] = xa.x * r127; xs[1] = xa.y * r127; xs[2] = xa.z * r127; xs[3] = xa.w * r127;
                xs[4] = xb.x * r127; xs[5] = xb.y * r127; xs[6] = xb.z * r127; xs[7] = xb.w * r127;
                xs[8] = xt.x * rt; xs[9] = xt.y * rt; xs[10] = xt.z * rt; xs[11] = xt.w * rt;
                if (itself) {   // x . x (the diagonal, when it is kept) = 127^2 * sum x' x'
                    float t = xs[0] * xs[0];
#pragma unroll
                    for (int e = 1; e < 12; e++) t = __builtin_fmaf(xs[e], xs[e], t);
                    t += __shfl_xor(t, 16);
                    t += __shfl_xor(t, 32);
                    xx = t * 16129.0f;
                }
                float x1 = 0.f;   // sum of this lane's x' (the "+ 1" of every s + 1)
#pragma unroll
                for (int e = 0; e < 12; e++) x1 += xs[e];
#pragma unroll
                for (int rr = 0; rr < 2; rr++) {
                    f32x2 acc = {x1, 0.f};
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        const unsigned w = rmain[u][rr][k] ^ 0x80808080u;   // s = u - 128 as signed bytes
                        const f32x2 s01 = {(float)(signed char)(w & 0xffu), (float)(signed char)((w >> 8) & 0xffu)};
                        const f32x2 s23 = {(float)(signed char)((w >> 16) & 0xffu), (float)(signed char)(w >> 24)};
                        acc = __builtin_elementwise_fma(s01, (f32x2){xs[4 * k], xs[4 * k + 1]}, acc);
                        acc = __builtin_elementwise_fma(s23, (f32x2){xs[4 * k + 2], xs[4 * k + 3]}, acc);
                    }
                    float t = acc[0] + acc[1];
                    t += __shfl_xor(t, 16);
                    t += __shfl_xor(t, 32);
                    zx[rr] = t;   // x . f_row for rows r16 (rr = 0) and 16 + r16
                }
                // ... staged at once (the x column of the output row; k-slot 0 writes): nothing of phase 1 lives on
                const int dump = 4 * (OUT_MAX + r16);
                *reinterpret_cast<float *>(reinterpret_cast<char *>(my_out) + (lane < d ? 4 * lane : dump)) = rx[u];
                const int i0 = r16, i1 = 16 + r16;
                const int zo0 = (q == 0 && i0 < F && 0 < i0 + itself) ? 4 * (d + (i0 * (i0 - 1 + 2 * itself)) / 2) : dump;
                const int zo1 = (q == 0 && i1 < F) ? 4 * (d + (i1 * (i1 - 1 + 2 * itself)) / 2) : dump;
                *reinterpret_cast<float *>(reinterpret_cast<char *>(my_out) + zo0) = i0 == 0 ? xx : zx[0];
                *reinterpret_cast<float *>(reinterpret_cast<char *>(my_out) + zo1) = zx[1];
            }
            __builtin_amdgcn_sched_barrier(0);
            // -- phase 2: the integer operands (this lane's 8 -- k-slot 3: 12 -- bytes of rows r16 and 16 + r16 as signed bytes;
            // tile row F is a row of ones: S_i = its products) and the three products
            i32x4 c00 = {0, 0, 0, 0}, c10 = {0, 0, 0, 0}, c11 = {0, 0, 0, 0};
            {
                i32x4 op[2];
#pragma unroll
                for (int rr = 0; rr < 2; rr++) {
                    const bool ones = (r16 + 16 * rr) == F;
                    const unsigned w0 = ones ? 0x01010101u : rmain[u][rr][0] ^ 0x80808080u;
                    const unsigned w1 = ones ? 0x01010101u : rmain[u][rr][1] ^ 0x80808080u;
                    const unsigned w2 = q == 3 ? (ones ? 0x01010101u : rmain[u][rr][2] ^ 0x80808080u) : 0u;
                    op[rr] = (i32x4){(int)w0, (int)w1, (int)w2, 0};
                }
                c00 = __builtin_amdgcn_mfma_i32_16x16x64_i8(op[0], op[0], c00, 0, 0, 0);
                c10 = __builtin_amdgcn_mfma_i32_16x16x64_i8(op[1], op[0], c10, 0, 0, 0);
                c11 = __builtin_amdgcn_mfma_i32_16x16x64_i8(op[1], op[1], c11, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // -- phase 3: S_i = row F of the products (tile row F - 16 of c10 / c11) -> the wave's LDS slot -> every lane's rows
            // and column; then the output row: x, the x column (phase 1), the packed lower triangle (integer products)
            {
                const int fr = F - 16, fq = fr >> 2, fv = fr & 3;   // wave-uniform
                const int s0 = fv == 0 ? c10[0] : fv == 1 ? c10[1] : fv == 2 ? c10[2] : c10[3];
                const int s1 = fv == 0 ? c11[0] : fv == 1 ? c11[1] : fv == 2 ? c11[2] : c11[3];
                if (q == fq) { s_S[wave_in_block][r16] = s0 + d; s_S[wave_in_block][16 + r16] = s1 + d; }   // (d folded into the column's share)
            }
            // the other lanes' reads below are ordered behind these lanes' writes by the hardware (one wave, LDS in order) but NOT
            // by the language: without a fence the compiler reads s_S first in the lanes that did not write it (it did, for the
            // first sample of every wave)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const float rz = 1.0f / 16129.0f;
            const int dump = 4 * (OUT_MAX + r16);
            const int sc0 = s_S[wave_in_block][r16], sc1 = s_S[wave_in_block][16 + r16];   // S_col + d
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const int i = 4 * q + v;
                const int si0 = s_S[wave_in_block][i] - d, si1 = s_S[wave_in_block][16 + i] - d;   // S_row
                // (column 0 and row 0 are x's: written above)
                const int zo00 = (i < F && r16 < i + itself && r16 > 0) ? 4 * (d + (i * (i - 1 + 2 * itself)) / 2 + r16) : dump;
                *reinterpret_cast<float *>(reinterpret_cast<char *>(my_out) + zo00) = (float)(c00[v] + si0 + sc0) * rz;
                const int gi = 16 + i;
                const int base = (gi * (gi - 1 + 2 * itself)) / 2;
                const int zo10 = (gi < F && r16 > 0) ? 4 * (d + base + r16) : dump;
                const int zo11 = (gi < F && 16 + r16 < gi + itself) ? 4 * (d + base + 16 + r16) : dump;
                *reinterpret_cast<float *>(reinterpret_cast<char *>(my_out) + zo10) = (float)(c10[v] + si1 + sc0) * rz;
                *reinterpret_cast<float *>(reinterpret_cast<char *>(my_out) + zo11) = (float)(c11[v] + si1 + sc1) * rz;
            }
            continue;
        }
        // operands: a[rr][c] = the 4 elements of chunk c (c < CQ: this k-slot's own chunks); of the REM trailing chunks
        // k-slot q feeds only element q to the matrix core: a[rr][CQ + t].x holds it, nothing else is decoded
        float4 a[NR][NC];
        if constexpr (FOLD) {
#pragma unroll
            for (int rr = 0; rr < NR; rr++) {
                unsigned tailw;
                if constexpr (CODEC == 8) {
                    tailw = rmain[u][rr][2];                                  // k-slot 3: bytes 32..35
                } else {
                    const unsigned lo = rmain[u][rr][0], hi = rmain[u][rr][1];
                    tailw = hi >> 16;                                         // k-slot 3: bytes 16..17
                    if constexpr (FOLD4A) {
                        const bool odd = (odd_rows >> (u * NR + rr)) & 1u;
                        rmain[u][rr][0] = (q == 3 || odd) ? __builtin_amdgcn_alignbit(hi, lo, 16) : lo;
                    } else
                    rmain[u][rr][0] = q < 3 ? lo : __builtin_amdgcn_alignbit(hi, lo, 16);   // ... and its chunks, bytes 12..15
                }
                rrem[u][rr][0][0] = (unsigned)__builtin_amdgcn_ds_bpermute(4 * (48 + r16), (int)tailw);
            }
        }
        auto decode = [&](auto fast) {
            constexpr bool FAST = decltype(fast)::value;
#pragma unroll
            for (int rr = 0; rr < NR; rr++) {
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    unsigned w0, w1;
                    if (c < CQ) chunk_words(rmain[u][rr], c, w0, w1);
                    else { w0 = rrem[u][rr][c - CQ][0]; w1 = CODEC == 16 ? rrem[u][rr][c - CQ][kRemDw - 1] : 0u; }
#ifdef EVS_XQ_NODEC   // developer A/B (timing only, wrong R): no decode
                    a[rr][c] = make_float4(__uint_as_float(w0), __uint_as_float(w1), __uint_as_float(w0 ^ w1), __uint_as_float(w0 + w1));
#else
                    if constexpr (PAIR4) {
                        if (c < CQ) {
                            const float2 p0 = s_lut2[w0 & 0xffu], p1 = s_lut2[(w0 >> 8) & 0xffu];
                            a[rr][c] = make_float4(p0.x, p0.y, p1.x, p1.y);
                        } else a[rr][c] = make_float4(dec_elem_q<CODEC, FAST>(w0, w1, q, s_lut), 0.f, 0.f, 0.f);
                    } else {
                    if (c < CQ) a[rr][c] = dec_chunk_q<CODEC, FAST>(w0, w1, s_lut);
                    else a[rr][c] = make_float4(dec_elem_q<CODEC, FAST>(w0, w1, q, s_lut), 0.f, 0.f, 0.f);
                    }
#endif
                }
            }
        };
        if constexpr (CODEC == 16) {
            // tail codes (> 65 000: |x| > 0.65, decoded through the LDS table) do not occur in tables encoded from trained
            // embeddings: ONE wave-uniform test per sample (packed u16 max over the lane's 12 raw words) picks the decoder
            u16x2 mx = __builtin_bit_cast(u16x2, rmain[u][0][0]);
#pragma unroll
            for (int rr = 0; rr < NR; rr++) {
#pragma unroll
                for (int k = 0; k < kMainDw; k++) mx = __builtin_elementwise_max(mx, __builtin_bit_cast(u16x2, rmain[u][rr][k]));
#pragma unroll
                for (int t = 0; t < REM; t++)
#pragma unroll
                    for (int k = 0; k < kRemDw; k++) mx = __builtin_elementwise_max(mx, __builtin_bit_cast(u16x2, rrem[u][rr][t][k]));
            }
            const bool tail = mx[0] > 65000 || mx[1] > 65000;
            if (__builtin_amdgcn_ballot_w64(tail) == 0ull) decode(std::true_type{}); else decode(std::false_type{});
        } else {
            decode(std::false_type{});
        }
        spread_x(a);
        const float xv = rx[u];
        flush_out(b - 4, u > 0 && u - 1 < n_samples);    // sample u-1 leaves under the MFMAs of sample u
        interact_stage(a, xv);
    }
    flush_out(blk_first + wave_in_block + 12, n_samples == 4);
    if (bad) atomicOr(args.err, 1);
}

static int rfq_mode() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("EVS_FUSED_RFQ"); v = e ? atoi(e) : 1; }   // developer switch: 0 = the LDS-DMA loop
    return v;
}
static int64_t rfq_max_batch() {
    static int64_t v = -1;
    if (v < 0) { const char *e = getenv("EVS_FUSED_RFQ_MAX_B"); v = e ? atoll(e) : (1ll << 40); }
    return v;
}

template <int CODEC, int CQ, int REM>
static void launch_rfq_nt(const FusedArgs &a, hipStream_t st) {
    const unsigned blocks = (unsigned)((a.B + 15) / 16);
    if (a.bag1 == 4) {
        if (a.F > 16) hipLaunchKernelGGL((emb_interact_rfq_kernel<CODEC, CQ, REM, 2, true>), dim3(blocks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((emb_interact_rfq_kernel<CODEC, CQ, REM, 1, true>), dim3(blocks), dim3(256), 0, st, a);
        return;
    }
    if (a.F > 16) hipLaunchKernelGGL((emb_interact_rfq_kernel<CODEC, CQ, REM, 2>), dim3(blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((emb_interact_rfq_kernel<CODEC, CQ, REM, 1>), dim3(blocks), dim3(256), 0, st, a);
}
template <int CODEC>
static bool launch_rfq_d(const FusedArgs &a, hipStream_t st) {
    switch (a.d) {
    case 16: launch_rfq_nt<CODEC, 1, 0>(a, st); return true;
    case 32: launch_rfq_nt<CODEC, 2, 0>(a, st); return true;
    case 36: launch_rfq_nt<CODEC, 2, 1>(a, st); return true;
    default: return false;
    }
}

// is there a kernel for this launch (bag1: 1 no offsets, 2 the one-index-per-bag leg of the optimistic triple, 4 offsets
// given and checked by the kernel itself)
bool rfq_supported(const FusedArgs &a, int codec) {
    if (codec != 16 && codec != 8 && codec != 4) return false;
    if (!rfq_mode() || !a.enc_lds || a.F > kTileMaxF || a.B > rfq_max_batch() || a.B >= (1ll << 31)) return false;
    return a.d == 16 || a.d == 32 || a.d == 36;
}

// the PROBE form: a single reduced-precision set-associative tier (8-way sets), d in {16, 32, 36}
template <int CODEC>
static bool launch_rfq_probe_d(const FusedArgs &a, hipStream_t st) {
    const unsigned blocks = (unsigned)((a.B + 15) / 16);
    const bool nt2 = a.F > 16;
    switch (a.d) {
    case 16:
        if (nt2) hipLaunchKernelGGL((emb_interact_rfq_kernel<CODEC, 1, 0, 2, false, true>), dim3(blocks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((emb_interact_rfq_kernel<CODEC, 1, 0, 1, false, true>), dim3(blocks), dim3(256), 0, st, a);
        return true;
    case 32:
        if (nt2) hipLaunchKernelGGL((emb_interact_rfq_kernel<CODEC, 2, 0, 2, false, true>), dim3(blocks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((emb_interact_rfq_kernel<CODEC, 2, 0, 1, false, true>), dim3(blocks), dim3(256), 0, st, a);
        return true;
    case 36:
        if (nt2) hipLaunchKernelGGL((emb_interact_rfq_kernel<CODEC, 2, 1, 2, false, true>), dim3(blocks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((emb_interact_rfq_kernel<CODEC, 2, 1, 1, false, true>), dim3(blocks), dim3(256), 0, st, a);
        return true;
    default: return false;
    }
}
bool rfq_probe_supported(int64_t B, int F, int d, int codec) {
    return (codec == 16 || codec == 8 || codec == 4) && rfq_mode() && F <= kTileMaxF && B < (1ll << 30) && (d == 16 || d == 32 || d == 36);
}
bool launch_rfq_probe(const FusedArgs &a, int codec, hipStream_t st) {
    if (!rfq_probe_supported(a.B, a.F, a.d, codec) || !a.probe.sa.tags || a.probe.sa.ways != 8u || !a.arena) return false;
    FusedArgs b = a;
    b.zero_codes = zero_code_page(codec);
    if (!b.zero_codes) return false;
    switch (codec) {
    case 16: return launch_rfq_probe_d<16>(b, st);
    case 8: return launch_rfq_probe_d<8>(b, st);
    default: return launch_rfq_probe_d<4>(b, st);
    }
}

bool launch_rfq(const FusedArgs &a, int codec, hipStream_t st) {
    if (!rfq_supported(a, codec) || (a.bag1 != 1 && a.bag1 != 2 && a.bag1 != 4)) return false;
    FusedArgs b = a;
    b.zero_codes = zero_code_page(codec);
    if (!b.zero_codes) return false;

    switch (codec) {
    case 16: return launch_rfq_d<16>(b, st);
    case 8: return launch_rfq_d<8>(b, st);
    default: return launch_rfq_d<4>(b, st);
    }
}

}  // namespace evs
