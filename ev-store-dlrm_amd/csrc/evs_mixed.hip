// Interaction consumer for rows of MIXED precision: R[b] = [x[b] | strict-lower(T T^T)], T = [x[b]; row_1; ...; row_T] where
// row_k is given by ADDRESS and a codec class -- the batched two-tier cache lookup (evs_cache.hip: C1 in the main
// precision, C2 in the secondary one, mixed_precs_caching/evlfu_8.cpp:669-796) serves every key from the tier that
// holds it, so the 26 rows of one sample come in two codecs (and from four places: either arena, either backing
// table).  Before this kernel the batched path decoded them to an fp32 (B,T,d) tensor in HBM (cache_rows_from_ptrs2)
// and ran the dense interaction over it: 2 x 3.7 KB per sample of traffic that exists only to change the format.
//
// One wavefront per sample.  Stage: the F * d/4 four-element chunks of the sample are dealt over the lanes (lane p of
// pass `it` owns chunk p % (d/4) of row p / (d/4)): pointer + class -> raw chunk (16 / 8 / 4 / 2 bytes) -> decode through
// the per-block LDS tables (evs_common.h, bit-exact with the reference's decoders) -> the fp32 image of the sample in
// the wave's LDS slot, in the layout the fp32 kernels use.  Then the same v_mfma_f32_16x16x4_f32 chains, the staged
// output row and 16-byte stores as evs_fused_rf.hip.
#include <type_traits>
#include "evs_common.h"
#include "evs_hash.h"

#include <stdlib.h>

namespace evs {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct MixedArgs {
    const float *x; long long x_stride;     // floats between samples
    const long long *row_ptrs;              // (B, T) row addresses (0 = no row: zeros)
    const unsigned char *row_class;         // (B, T) 1 = codec1, 2 = codec2, 0 = zeros
    float *R;
    long long B;
    int T, itself, P, codec1, codec2;
    int default_class;                      // row_class == NULL: the class of every row
};

template <int CQ, int REM, int NT>
__global__ void __launch_bounds__(256) interact_mixed_rows_kernel(const MixedArgs args) {
    constexpr int NR = NT, NC = CQ + REM;
    constexpr int d = 4 * (4 * CQ + REM);
    constexpr int NCH = d / 4;              // 4-element chunks per row
    constexpr int RPI = 64 / NCH;           // rows per KiB of the image
    constexpr int NROWS = 16 * NT;
    constexpr int NJ = (NROWS + RPI - 1) / RPI;
    constexpr int row_bytes = d * 4;
    __shared__ __attribute__((aligned(16))) char s_rows[4][NJ * 1024];
    constexpr int OUT_MAX = ((d + NROWS * (NROWS + 1) / 2 + 63) / 64) * 64;
    __shared__ __attribute__((aligned(16))) float s_out[4][OUT_MAX + 16];
    __shared__ float s_lut16[CodecLut<16>::kEntries], s_lut8[CodecLut<8>::kEntries], s_lut4[CodecLut<4>::kEntries];
    codec_lut_init<16>(s_lut16);
    codec_lut_init<8>(s_lut8);
    codec_lut_init<4>(s_lut4);
    __syncthreads();

    const int lane = threadIdx.x & (kWave - 1);
    const int r16 = lane & 15, q = lane >> 4;
    const int T = args.T, F = T + 1, itself = args.itself;
    const int out_row = d + args.P;
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    char *my_lds = s_rows[wave_in_block];
    float *my_out = s_out[wave_in_block];
    int lds_off[NR];
#pragma unroll
    for (int rr = 0; rr < NR; rr++) {
        const int row = r16 + 16 * rr;
        lds_off[rr] = (row / RPI) * 1024 + (row % RPI) * row_bytes + q * CQ * 16;
    }
    constexpr int kRemOff = 4 * CQ * 16;
    constexpr int kOob = 0x7ffffff0;
    constexpr int NIT = (NROWS * NCH + 63) / 64;
    const long long waves_total = (long long)gridDim.x * 4;
    for (long long b = (long long)blockIdx.x * 4 + wave_in_block; b < args.B; b += waves_total) {
        // ---- stage: decode the sample's rows into the fp32 image ----
        float4 v[NIT];
        int dst[NIT];
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int p = lane + 64 * it;
            const int r = p / NCH, c = p - r * NCH;
            v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            dst[it] = r < NROWS ? (r / RPI) * 1024 + (r % RPI) * row_bytes + c * 16 : -1;
            if (r == 0) {
                v[it] = reinterpret_cast<const float4 *>(args.x + b * args.x_stride)[c];
            } else if (r < F) {
                const long long ptr = args.row_ptrs[b * T + (r - 1)];
                const int cls = args.row_class ? args.row_class[b * T + (r - 1)] : args.default_class;
                const int codec = cls == 1 ? args.codec1 : args.codec2;
                if (ptr && cls) {
                    const unsigned char *row = reinterpret_cast<const unsigned char *>(ptr);
                    if (codec == 32) v[it] = reinterpret_cast<const float4 *>(row)[c];
                    else if (codec == 16) { const uint2 w = reinterpret_cast<const uint2 *>(row)[c]; v[it] = dec_chunk<16>(w.x, w.y, s_lut16); }
                    else if (codec == 8) v[it] = dec_chunk<8>(reinterpret_cast<const unsigned *>(row)[c], 0u, s_lut8);
                    else v[it] = dec_chunk<4>(reinterpret_cast<const unsigned short *>(row)[c], 0u, s_lut4);
                }
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; it++)
            if (dst[it] >= 0) *reinterpret_cast<float4 *>(my_lds + dst[it]) = v[it];
        // ---- operands, interaction, output row (as evs_fused_rf.hip) ----
        float4 a[NR][NC];
#pragma unroll
        for (int rr = 0; rr < NR; rr++) {
#pragma unroll
            for (int c = 0; c < CQ; c++) a[rr][c] = *reinterpret_cast<const float4 *>(my_lds + lds_off[rr] + c * 16);
#pragma unroll
            for (int m = 0; m < REM; m++)
                a[rr][CQ + m] = *reinterpret_cast<const float4 *>(my_lds + lds_off[rr] - q * CQ * 16 + kRemOff + m * 16);
        }
        float xv[(d + 63) / 64];
#pragma unroll
        for (int h = 0; h < (d + 63) / 64; h++) {
            const int e = lane + 64 * h;
            xv[h] = reinterpret_cast<const float *>(my_lds)[e < d ? e : 0];
        }
        f32x4 c00 = {0.f, 0.f, 0.f, 0.f}, c10 = c00, c11 = c00;
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
            } else {
                const float s0 = q == 0 ? e0[0] : q == 1 ? e0[1] : q == 2 ? e0[2] : e0[3];
                const float s1 = q == 0 ? e1[0] : q == 1 ? e1[1] : q == 2 ? e1[2] : e1[3];
                c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(s0, s0, c00, 0, 0, 0);
                if constexpr (NT == 2) {
                    c10 = __builtin_amdgcn_mfma_f32_16x16x4f32(s1, s0, c10, 0, 0, 0);
                    c11 = __builtin_amdgcn_mfma_f32_16x16x4f32(s1, s1, c11, 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int h = 0; h < (d + 63) / 64; h++) {
            const int e = lane + 64 * h;
            my_out[e < d ? e : OUT_MAX + r16] = xv[h];
        }
#pragma unroll
        for (int vv = 0; vv < 4; vv++) {
            const int i = 4 * q + vv;
            const int dump = 4 * (OUT_MAX + r16);
            const int zo00 = (i < F && r16 < i + itself) ? 4 * (d + (i * (i - 1 + 2 * itself)) / 2 + r16) : dump;
            *reinterpret_cast<float *>(reinterpret_cast<char *>(my_out) + zo00) = c00[vv];
            if constexpr (NT == 2) {
                const int gi = 16 + i;
                const int base = (gi * (gi - 1 + 2 * itself)) / 2;
                const int zo10 = gi < F ? 4 * (d + base + r16) : dump;
                const int zo11 = (gi < F && 16 + r16 < gi + itself) ? 4 * (d + base + 16 + r16) : dump;
                *reinterpret_cast<float *>(reinterpret_cast<char *>(my_out) + zo10) = c10[vv];
                *reinterpret_cast<float *>(reinterpret_cast<char *>(my_out) + zo11) = c11[vv];
            }
        }
        float *Rb = args.R + b * (long long)out_row;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(Rb, 0, out_row * 4, 0x00020000);
        const int n4 = out_row >> 2;
#pragma unroll
        for (int h = 0; h < (OUT_MAX + 255) / 256; h++) {
            if (h * 64 < n4) {
                const int e4 = lane + 64 * h;
                const float4 w = reinterpret_cast<const float4 *>(my_out)[e4 < n4 ? e4 : 0];
                u32x4 u = {__float_as_uint(w.x), __float_as_uint(w.y), __float_as_uint(w.z), __float_as_uint(w.w)};
                __builtin_amdgcn_raw_buffer_store_b128(u, rs, e4 < n4 ? 16 * e4 : kOob, 0, 2);
            }
        }
        if (out_row & 3) {
            const int e = 4 * n4 + (lane & 3);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(my_out[e]), rs, lane < (out_row & 3) ? 4 * e : kOob, 0, 2);
        }
    }
}

// ---- codec pair (8, 4), the reference's as-shipped tiers: rows in flight in registers ------------------------------------
// The kernel above takes one sample per wave at a time: pointer -> row chunk -> decode -> LDS image -> operands, two
// dependent round trips per sample with nothing else of the wave in flight (34 us at B = 16 384).  This form is
// evs_fused_rfq.hip's: a block owns 16 samples, their (address, class) pairs go through an LDS tile, every wave requests
// the rows of its 4 samples at once, each lane exactly the bytes it feeds the matrix core (row r16 + 16 rr, chunks
// [q CQ, (q + 1) CQ) and the REM tail), and consumes them in order under counted s_waitcnt vmcnt.  The two classes differ
// in row size (d bytes / d/2 bytes), so a lane loads the WIDER class's piece for either class -- for a u4 row from a
// start clamped into the row, shifted into place afterwards (no byte outside the row is read) -- and decodes through ONE
// LDS table: entries 0..255 the u8 values, 256..271 the u4 values, the index picked per lane.  Absent rows read the
// u8 zero-code page (code 127 = 0.0f exactly).
struct Mixed84Args {
    MixedArgs m;
    const void *zero_codes8;
    const void *zeros;
    Probe2Args p;   // PROBE variant
};

template <int N>
__device__ __forceinline__ void mixed_load_raw(unsigned long long p, unsigned (&w)[N >= 4 ? N / 4 : 1]) {
    if constexpr (N == 12) {
        typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
        const u32x3 v = *reinterpret_cast<const __attribute__((address_space(1), aligned(4))) u32x3 *>((uintptr_t)p);
        w[0] = v[0]; w[1] = v[1]; w[2] = v[2];
    } else if constexpr (N == 8) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 v = *reinterpret_cast<const __attribute__((address_space(1))) u32x2 *>((uintptr_t)p);
        w[0] = v[0]; w[1] = v[1];
    } else if constexpr (N == 4) {
        w[0] = *reinterpret_cast<const __attribute__((address_space(1))) unsigned *>((uintptr_t)p);
    } else {
        static_assert(N == 2, "raw pieces of 12, 8, 4 or 2 bytes");
        w[0] = *reinterpret_cast<const __attribute__((address_space(1))) unsigned short *>((uintptr_t)p);
    }
}

// PROBE: the two-tier probe (cache_batch_probe2_kernel, evs_cache.hip) folded into the head -- thread e (and e + 256) owns key
// (table (e >> 4) - 1, sample e & 15): both hashes, the alt-key tier, agg_hit per request (LDS), priority bumps, the routing of
// double misses, tier codes, each tier's miss list of the block, hit statistics -- and the (address, class) pairs go straight
// into the LDS tile instead of through two (B,T) arrays and a launch boundary.
// blocks per CU the PROBE form is compiled for: at 5 (96 VGPRs) it spills six registers -- scratch traffic and vmcnt(0) waits
// inside the counted sequence; at 4 it needs 99 and spills nothing (two tiers + interaction 46.9 -> 44.6-45.8 us per batch)
#ifndef EVS_MIXED_LB_PROBE
#define EVS_MIXED_LB_PROBE 4
#endif
#ifndef EVS_MIXED_FOLD
#define EVS_MIXED_FOLD 1   // developer A/B: 0 = the tail chunk of a d = 36 row as a load of its own
#endif
template <int CQ, int REM, int NT, bool PROBE = false>
__global__ void __launch_bounds__(256, PROBE ? EVS_MIXED_LB_PROBE : 5) interact_mixed84_kernel(const Mixed84Args margs) {
    const MixedArgs &args = margs.m;
    constexpr int NR = NT, NC = CQ + REM;
    constexpr int d = 4 * (4 * CQ + REM);
    constexpr int NROWS = 16 * NT;
    constexpr int D = 4;
    constexpr int RB2 = d / 2;                      // bytes of a u4 row (a u8 row: d)
    constexpr int W1 = CQ * 4, W2 = CQ * 2;         // this lane's main piece per class
    constexpr int WM = W1;                          // loaded width (the wider class)
    constexpr int WMdw = WM >= 4 ? WM / 4 : 1;
    static_assert(RB2 >= WM && (REM == 0 || RB2 >= 4), "the clamped window must fit a u4 row");
    __shared__ unsigned long long s_ptr[512];       // [32 features][16 samples]: row address (feature 0 = x: unused)
    __shared__ unsigned char s_cls[512];
    __shared__ __attribute__((aligned(16))) float s_x[4][64];
    // the decode table: the 256 u8 values, the sixteen u4 values behind them.  (Round 6: sixteen copies side by side, lane l
    // reading copy l & 15 -- at most a 2-way bank conflict where 32 lanes reading a 256-entry table at RANDOM indices collide ~3.5
    // ways -- was built and measured: SQ_LDS_BANK_CONFLICT 2.92 M -> 3.42 M per launch, the same 39.2 us.  The codes of real
    // tables are not random: embeddings sit around zero, i.e. around code 127 / nibble 7, a dozen distinct entries per
    // instruction, which the single table serves conflict-free and with broadcasts; the copies turn equal indices in different
    // lanes into different addresses.  The table is not where this kernel's LDS conflicts come from.)
    constexpr int kLutCopies = 1;
    constexpr unsigned kU4Base = 256u * kLutCopies * 4u;       // byte offset of the u4 entries
    constexpr int kLutShift = 2;                               // log2 of the bytes between two entries
    __shared__ float s_lut[(256 + 16) * kLutCopies];
    constexpr int OUT_MAX = ((d + NROWS * (NROWS + 1) / 2 + 63) / 64) * 64;
    __shared__ __attribute__((aligned(16))) float s_out[4][OUT_MAX + 16];

    const int lane = threadIdx.x & (kWave - 1);
    const int r16 = lane & 15, q = lane >> 4;
    const int T = args.T, F = T + 1, itself = args.itself;
    const int out_row = d + args.P;
    const long long B = args.B;
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float *my_out = s_out[wave_in_block];
    float *my_x = s_x[wave_in_block];
    const unsigned long long zc_p = (unsigned long long)reinterpret_cast<uintptr_t>(margs.zero_codes8);
    const unsigned long long zeros_p = (unsigned long long)reinterpret_cast<uintptr_t>(margs.zeros);

    const long long blk_first = (long long)blockIdx.x * 16;
    const long long blk_end = blk_first + 16 < B ? blk_first + 16 : B;
    if (blk_first >= blk_end) return;
    const int blk_n = (int)(blk_end - blk_first);
    const int n_samples = blk_n > wave_in_block ? (blk_n - wave_in_block + 3) / 4 : 0;
    // (Round 6: the four blocks that share a CU started 1 / 2 / 3 us apart -- do their probe heads, which wait for memory, and
    //  their consume loops, which issue, overlap better than in lock-step?  39.2 -> 40.1 / 41.3 / 43.4 us per batch: every
    //  microsecond of stagger is a microsecond added.  A block's own chain -- head, then its four samples per wave -- is what the
    //  launch waits for; tools/variants.sh st100@evs_mixed.)

    // ---- the tile: the block's 16 x T (address, class) pairs are one contiguous run of each array ----------------------
    for (int i = threadIdx.x; i < 512; i += blockDim.x) { s_ptr[i] = zc_p; s_cls[i] = 1; }
    for (int i = threadIdx.x; i < 256 + 16; i += blockDim.x) {   // (no read of the __constant__ table: a memory round trip in the head)
        s_lut[i] = i < 256 ? dec_u8((unsigned)i) : u4_value((unsigned)(i - 256));
    }
    // PROBE: per-table facts the probe indexes by LANE (a per-lane index into the kernel arguments is a vector-memory round
    // trip, and two of them behind a short-circuit && are two DEPENDENT ones in front of the set loads): rows both tiers'
    // tables have, dense row number of row 0 (set-associative tiers)
    __shared__ long long s_nrows[PROBE ? 32 : 1];
    __shared__ unsigned s_sa_base[PROBE ? 32 : 1];
    __shared__ unsigned long long s_back[PROBE ? 64 : 1];   // the tables the two tiers read their misses from (t1: [k], t2: [32 + k])
    __shared__ unsigned long long s_alt[PROBE ? 32 : 1];    // the alt-key tables (three tiers)
    if constexpr (PROBE) {
        if (threadIdx.x < 32) {
            const long long r1 = margs.p.t1.backing_rows[threadIdx.x], r2 = margs.p.t2.backing_rows[threadIdx.x];
            s_nrows[threadIdx.x] = r1 < r2 ? r1 : r2;
            s_sa_base[threadIdx.x] = margs.p.sau.row_base[threadIdx.x];
            s_back[threadIdx.x] = (unsigned long long)reinterpret_cast<uintptr_t>(margs.p.t1.backing[threadIdx.x]);
            s_back[32 + threadIdx.x] = (unsigned long long)reinterpret_cast<uintptr_t>(margs.p.t2.backing[threadIdx.x]);
            s_alt[threadIdx.x] = (unsigned long long)reinterpret_cast<uintptr_t>(margs.p.c3.alt_tables[threadIdx.x]);
        }
    }
    __syncthreads();
    __shared__ int s_agg[16];                        // PROBE: hits per request of the chunk
    __shared__ int s_d1[kMaxBuckets], s_d2[kMaxBuckets];   // priority histogram moves of the two tiers
    __shared__ int s_sum[4];                         // C1 hits, C2 hits, perfect requests, alt-key hits
    __shared__ int s_list_n[2];                      // misses listed for C1 / C2
    if constexpr (PROBE) {
        const Probe2Args &pa = margs.p;
        for (int i = threadIdx.x; i < kMaxBuckets; i += blockDim.x) { s_d1[i] = 0; s_d2[i] = 0; }
        if (threadIdx.x < 16) s_agg[threadIdx.x] = 0;
        if (threadIdx.x < 4) s_sum[threadIdx.x] = 0;
        if (threadIdx.x < 2) s_list_n[threadIdx.x] = 0;
        __syncthreads();
        const bool c1_full = *pa.t1.count >= pa.t1.cap - pa.t1.full_slack;   // snapshot, as probe2
        const long long bs = blk_first + (threadIdx.x & 15);
        int prow[2], e1[2], e2[2], ea[2], alt_tier[2];
        bool act[2], ok[2], ht1[2], ht2[2], c1_room[2];
        unsigned long long key[2], end1[2], end2[2], w1[2], w2[2];
        unsigned tg1[2] = {0u, 0u}, tg2[2] = {0u, 0u};
        int yw[2] = {-1, -1};                       // set-associative tiers: the way of the hit (in the tier that holds the key)
        const bool sa = pa.t1.sa.tags != nullptr;   // set-associative tiers (evs_hash.h): both tiers or neither
#pragma unroll
        for (int h = 0; h < 2; h++) {   // the thread's two request rows side by side
            const int f = ((int)threadIdx.x >> 4) + 16 * h;
            act[h] = f >= 1 && f < F && bs < blk_end;
            const int *rp = act[h] ? pa.requests + bs * (long long)T + (f - 1) : pa.requests;
            prow[h] = *reinterpret_cast<const __attribute__((address_space(1))) int *>(reinterpret_cast<uintptr_t>(rp));
        }
        __builtin_amdgcn_sched_barrier(0);   // both request rows on the wire before either is looked at
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int f = ((int)threadIdx.x >> 4) + 16 * h, k = f >= 1 ? f - 1 : 0;
            const long long nr = s_nrows[k];   // (unconditional read, bitwise tests: no branch between the two keys' loads)
            ok[h] = act[h] & (prow[h] >= 0) & ((long long)prow[h] < nr);
            key[h] = ((unsigned long long)f << 32) | (unsigned)prow[h];
            end1[h] = end2[h] = 0; ht1[h] = ht2[h] = false;
            e1[h] = e2[h] = -1; w1[h] = w2[h] = 0ull; c1_room[h] = !c1_full;
        }
        if (sa) {
            // both tiers' set lines of both keys in ONE round trip (the hashed form walks C1, then C2, then reads the
            // priority); "C1 has room" is a property of the key's own C1 set: a free way
            // (a pair that shares its set records: both tiers' ways sit in ONE 128-byte line)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int f = ((int)threadIdx.x >> 4) + 16 * h;
                const unsigned px = sa_perm(pa.sau, s_sa_base[(f + 31) & 31] + (ok[h] ? (unsigned)prow[h] : 0u));
                unsigned s1, s2, sx, qx;
                sa_divmod(pa.t1.sa, px, sx, qx);
                sa_place(pa.t1.sa, sx, qx, s1, tg1[h]);
                if (pa.t2.sa.nset == pa.t1.sa.nset) sa_place(pa.t2.sa, sx, qx, s2, tg2[h]);   // (a pair that shares its records: one division)
                else sa_split(pa.t2.sa, px, s2, tg2[h]);
                end1[h] = ok[h] ? s1 : 0u;
                end2[h] = ok[h] ? s2 : 0u;
            }
            // (the probe is folded into this kernel for 8-way tiers only -- C1 8 ways + C2 8-way sub-sets, the reference's 1 : 2
            //  pair in one 128-byte record; the host checks: evs_cache.hip -- so the way counts are compile-time constants)
            SaLine l1[2], l2[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                sa_load<8>(pa.t1.sa, (unsigned)end1[h], l1[h]);
                sa_load<8>(pa.t2.sa, (unsigned)end2[h], l2[h]);
            }
            __builtin_amdgcn_sched_barrier(0);   // all the set loads of the thread's two keys in ONE round trip
#pragma unroll
            for (int h = 0; h < 2; h++) {
                unsigned x1, x2;
                const int y1 = sa_find<8>(pa.t1.sa, l1[h], tg1[h], x1), y2 = sa_find<8>(pa.t2.sa, l2[h], tg2[h], x2);   // (a pair's update is a launch of its own: no stamp is in flight here)
                w1[h] = x1; w2[h] = x2;
                if (ok[h] && y1 >= 0) { e1[h] = (int)sa_entry(pa.t1.sa, (unsigned)end1[h], (unsigned)y1, x1); yw[h] = y1; }
                else if (ok[h] && y2 >= 0) { e2[h] = (int)sa_entry(pa.t2.sa, (unsigned)end2[h], (unsigned)y2, x2); yw[h] = y2; }
                c1_room[h] = sa_has_free<8>(pa.t1.sa, l1[h]);
            }
        } else {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            e1[h] = ok[h] ? probe_ro(pa.t1.slots, pa.t1.mask, key[h], end1[h], pa.t1.reusable_tomb, &ht1[h]) : -1;
            if (e1[h] == kPending) e1[h] = -1;
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            e2[h] = (ok[h] && e1[h] < 0) ? probe_ro(pa.t2.slots, pa.t2.mask, key[h], end2[h], pa.t2.reusable_tomb, &ht2[h]) : -1;
            if (e2[h] == kPending) e2[h] = -1;
        }
        }
        bool dm[2];
#pragma unroll
        for (int h = 0; h < 2; h++) { alt_tier[h] = 0; ea[h] = -1; dm[h] = pa.c3.tags != nullptr && ok[h] && e1[h] < 0 && e2[h] < 0; }
        if (pa.c3.tags != nullptr) {   // (wave-uniform) alt-key probe for the double misses (find_approximate_ev, evlfu_8.cpp:474-490), as in probe2
            typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
            typedef const __attribute__((address_space(1))) u64x2 *gtag_t;
            long long cb[2], w3[2];
            unsigned alt[2];
            u64x2 cw[2][kSetWays / 2];
            // the key's set of the alt-key tier AND its alt-table entry in one round trip (the entry does not wait for the
            // membership verdict), both keys of the thread side by side; lanes without a double miss read word 0 / their own request
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int f = ((int)threadIdx.x >> 4) + 16 * h;
                cb[h] = dm[h] ? c3_base(key[h], pa.c3.nset) : 0ll;
                const unsigned *ap = dm[h] ? reinterpret_cast<const unsigned *>((uintptr_t)s_alt[(f + 31) & 31]) + prow[h] : reinterpret_cast<const unsigned *>(pa.requests);
                alt[h] = *reinterpret_cast<const __attribute__((address_space(1))) unsigned *>(reinterpret_cast<uintptr_t>(ap));
                const gtag_t tp = reinterpret_cast<gtag_t>(reinterpret_cast<uintptr_t>(pa.c3.tags + cb[h]));
#pragma unroll
                for (int w = 0; w < kSetWays / 2; w++) cw[h][w] = tp[w];
            }
            __builtin_amdgcn_sched_barrier(0);
            unsigned at[2], ar[2];
            bool va[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                w3[h] = -1;
#pragma unroll
                for (int w = 0; w < kSetWays; w++)
                    if ((cw[h][w >> 1][w & 1] & kKeyMask) == key[h]) w3[h] = cb[h] + w;
                at[h] = alt[h] % 100u; ar[h] = alt[h] / 100u;
                va[h] = dm[h] && w3[h] >= 0 && at[h] >= 1 && at[h] <= (unsigned)T;
                va[h] = va[h] && (long long)ar[h] < s_nrows[(at[h] - 1u) & 31u];   // (rows BOTH tiers' tables have)
            }
            if (sa) {
                // both tiers' ways of the alt key in ONE round trip (a shared record: one line), as for the key itself
                unsigned a1[2], a2[2], ag1[2], ag2[2];
                SaLine la1[2], la2[2];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const unsigned px = sa_perm(pa.sau, s_sa_base[(at[h] - 1u) & 31u] + (va[h] ? ar[h] : 0u));
                    unsigned sx, qx;
                    sa_divmod(pa.t1.sa, px, sx, qx);
                    sa_place(pa.t1.sa, sx, qx, a1[h], ag1[h]);
                    if (pa.t2.sa.nset == pa.t1.sa.nset) sa_place(pa.t2.sa, sx, qx, a2[h], ag2[h]);
                    else sa_split(pa.t2.sa, px, a2[h], ag2[h]);
                    if (!va[h]) { a1[h] = 0u; a2[h] = 0u; }
                    sa_load<8>(pa.t1.sa, a1[h], la1[h]);
                    sa_load<8>(pa.t2.sa, a2[h], la2[h]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    unsigned xw, xw2;
                    const int y1 = sa_find<8>(pa.t1.sa, la1[h], ag1[h], xw), y2 = sa_find<8>(pa.t2.sa, la2[h], ag2[h], xw2);
                    if (va[h] && y1 >= 0) { ea[h] = (int)sa_entry(pa.t1.sa, a1[h], (unsigned)y1, xw); alt_tier[h] = 1; }
                    else if (va[h] && y2 >= 0) { ea[h] = (int)sa_entry(pa.t2.sa, a2[h], (unsigned)y2, xw2); alt_tier[h] = 2; }
                }
            } else {
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    if (va[h]) {
                        const unsigned long long akey = ((unsigned long long)at[h] << 32) | ar[h];
                        unsigned long long es;
                        ea[h] = probe_ro(pa.t1.slots, pa.t1.mask, akey, es);
                        if (ea[h] >= 0) alt_tier[h] = 1;
                        else { ea[h] = probe_ro(pa.t2.slots, pa.t2.mask, akey, es); if (ea[h] >= 0) alt_tier[h] = 2; }
                    }
                }
            }
#pragma unroll
            for (int h = 0; h < 2; h++)
                if (alt_tier[h]) atomicOr(&pa.c3.tags[w3[h]], kC3Flag);   // set_recency_flag_c3
        }
#pragma unroll
        for (int h = 0; h < 2; h++)
            if (e1[h] >= 0 || e2[h] >= 0 || alt_tier[h]) atomicAdd(&s_agg[threadIdx.x & 15], 1);
        __syncthreads();
        const int agg = s_agg[threadIdx.x & 15];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int f = ((int)threadIdx.x >> 4) + 16 * h, k = f >= 1 ? f - 1 : 0;
            if (sa) {
                if (e1[h] >= 0 && sa_prio((unsigned)w1[h]) < agg) {
                    const int old = sa_raise(pa.t1.sa, sa_ways_ptr(pa.t1.sa, (unsigned)end1[h]) + yw[h], (unsigned)w1[h], agg);
                    if (old >= 0) { atomicSub(&s_d1[old], 1); atomicAdd(&s_d1[agg], 1); }
                }
                if (e2[h] >= 0 && sa_prio((unsigned)w2[h]) < agg) {
                    const int old = sa_raise(pa.t2.sa, sa_ways_ptr(pa.t2.sa, (unsigned)end2[h]) + yw[h], (unsigned)w2[h], agg);
                    if (old >= 0) { atomicSub(&s_d2[old], 1); atomicAdd(&s_d2[agg], 1); }
                }
            } else {
            if (e1[h] >= 0 && pa.t1.eagg[e1[h]] < agg) {
                const int old = atomicMax(&pa.t1.eagg[e1[h]], agg);
                if (old < agg) { atomicSub(&s_d1[old], 1); atomicAdd(&s_d1[agg], 1); }
            }
            if (e2[h] >= 0 && pa.t2.eagg[e2[h]] < agg) {
                const int old = atomicMax(&pa.t2.eagg[e2[h]], agg);
                if (old < agg) { atomicSub(&s_d2[old], 1); atomicAdd(&s_d2[agg], 1); }
            }
            }
            // evlfu_8.cpp:570-601: where a double miss goes
            const bool miss = ok[h] && e1[h] < 0 && e2[h] < 0 && alt_tier[h] == 0;
            const int dest = c1_room[h] ? 1 : (agg < pa.threshold ? ((k & 1) ? 1 : 2) : 2);
            const unsigned char *src = nullptr;
            int codec_of = 0;
            if (e1[h] >= 0) { src = pa.t1.arena + (long long)((unsigned long long)(unsigned)e1[h] * (unsigned long long)d); codec_of = 1; }
            else if (e2[h] >= 0) { src = pa.t2.arena + (long long)((unsigned long long)(unsigned)e2[h] * (unsigned long long)RB2); codec_of = 2; }
            else if (alt_tier[h] == 1) { src = pa.t1.arena + (long long)((unsigned long long)(unsigned)ea[h] * (unsigned long long)d); codec_of = 1; }
            else if (alt_tier[h] == 2) { src = pa.t2.arena + (long long)((unsigned long long)(unsigned)ea[h] * (unsigned long long)RB2); codec_of = 2; }
            else if (miss && dest == 1) { src = reinterpret_cast<const unsigned char *>((uintptr_t)s_back[k & 31]) + (long long)prow[h] * d; codec_of = 1; }
            else if (miss) { src = reinterpret_cast<const unsigned char *>((uintptr_t)s_back[32 + (k & 31)]) + (long long)prow[h] * RB2; codec_of = 2; }
            if (pa.route_filter && miss && dest == 1 && (k & 1)) pa.route_filter[mix64(key[h]) & pa.route_mask] = pa.route_stamp;
            if (act[h]) {
                const long long m = bs * (long long)T + k;
                if (src) { s_ptr[f * 16 + ((int)threadIdx.x & 15)] = (unsigned long long)reinterpret_cast<uintptr_t>(src); s_cls[f * 16 + ((int)threadIdx.x & 15)] = (unsigned char)codec_of; }
                pa.tier_out[m] = e1[h] >= 0 ? 1 : (e2[h] >= 0 ? 2 : (alt_tier[h] ? 3 : 0));
                if (miss) {
                    const TierProbe &tp = dest == 1 ? pa.t1 : pa.t2;
                    const int at = atomicAdd(&s_list_n[dest - 1], 1);
                    tp.miss_rec[(long long)blockIdx.x * pa.list_cap + at] =
                        make_uint4((unsigned)prow[h], (unsigned)k | ((unsigned)agg << 8) | ((dest == 1 ? ht1[h] : ht2[h]) ? 0x10000u : 0u),
                                   (unsigned)((dest == 1 ? end1[h] : end2[h]) >> tp.hint_shift), sa ? (dest == 1 ? tg1[h] : tg2[h]) : (unsigned)m);
                }
                if (e1[h] >= 0) atomicAdd(&s_sum[0], 1);
                if (e2[h] >= 0) atomicAdd(&s_sum[1], 1);
                if (alt_tier[h]) atomicAdd(&s_sum[3], 1);
            }
            if (f == 1 && bs < blk_end && agg == T) atomicAdd(&s_sum[2], 1);
        }
    } else {
        const int n = blk_n * T;
        const long long base = blk_first * T;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int i = (int)threadIdx.x + 256 * h;
            const bool on = i < n;
            const long long pv = *reinterpret_cast<const __attribute__((address_space(1))) long long *>(
                reinterpret_cast<uintptr_t>(on ? args.row_ptrs + base + i : args.row_ptrs));
            const unsigned char cv = args.row_class ? *reinterpret_cast<const __attribute__((address_space(1))) unsigned char *>(
                reinterpret_cast<uintptr_t>(on ? args.row_class + base + i : args.row_class)) : (unsigned char)args.default_class;
            if (on && pv && cv) {
                const int sidx = i / T, k = i - sidx * T;
                s_ptr[(k + 1) * 16 + sidx] = (unsigned long long)pv;
                s_cls[(k + 1) * 16 + sidx] = cv;
            }
        }
    }
    __syncthreads();
    if constexpr (PROBE) {   // the block's totals into one replica row per tier (folded by the caches' close), its list lengths
        const Probe2Args &pa = margs.p;
        if (threadIdx.x < 40) {
            const int i = threadIdx.x;
            const int v1 = i <= T ? s_d1[i] : i == 38 ? s_sum[0] : i == 39 ? s_sum[2] : 0;
            const int v2 = i <= T ? s_d2[i] : i == 38 ? s_sum[1] : 0;
            if (v1) atomicAdd(&pa.t1.part1[(blockIdx.x % 32) * 40 + i], v1);
            if (v2) atomicAdd(&pa.t2.part1[(blockIdx.x % 32) * 40 + i], v2);
        }
        if (threadIdx.x == 0) {
            pa.t1.list_cnt[blockIdx.x] = s_list_n[0]; pa.t2.list_cnt[blockIdx.x] = s_list_n[1];
            if (pa.c3.tags && s_sum[3]) atomicAdd(reinterpret_cast<unsigned long long *>(&pa.c3.stat[1]), (unsigned long long)s_sum[3]);
        }
    }

    constexpr int kOob = 0x7ffffff0;
    // lane-invariant pieces of the flush and of the staging, computed ONCE (round 6, as evs_fused_rf.hip has had them since round
    // 3: left to itself the compiler re-derives the flush offsets and the twelve accumulator offsets -- multiplies, compares,
    // exec-mask regions, ~120 vector instructions -- for every sample: two thirds of this loop's vector work)
    constexpr int NFL = (OUT_MAX + 255) / 256;
    int fl_lds[NFL], fl_off[NFL], fl_tail_lds, fl_tail_off;
    {
        const int n4 = out_row >> 2;
#pragma unroll
        for (int h = 0; h < NFL; h++) {
            const int e4 = lane + 64 * h;
            fl_lds[h] = 16 * (e4 < n4 ? e4 : 0);
            fl_off[h] = e4 < n4 ? 16 * e4 : kOob;
            asm volatile("" : "+v"(fl_lds[h]), "+v"(fl_off[h]));
        }
        fl_tail_lds = 4 * (4 * n4 + (lane & 3));
        fl_tail_off = lane < (out_row & 3) ? fl_tail_lds : kOob;
        asm volatile("" : "+v"(fl_tail_lds), "+v"(fl_tail_off));
    }
    int zo00h[4], zo10h[4], zo11h[4], xv_offh;
    {
        const int dump0 = 4 * (OUT_MAX + r16);
#pragma unroll
        for (int vv = 0; vv < 4; vv++) {
            const int i = 4 * q + vv;
            zo00h[vv] = (i < F && r16 < i + itself) ? 4 * (d + (i * (i - 1 + 2 * itself)) / 2 + r16) : dump0;
            const int gi = 16 + i;
            const int base = (gi * (gi - 1 + 2 * itself)) / 2;
            zo10h[vv] = (NT == 2 && gi < F) ? 4 * (d + base + r16) : dump0;
            zo11h[vv] = (NT == 2 && gi < F && 16 + r16 < gi + itself) ? 4 * (d + base + 16 + r16) : dump0;
            asm volatile("" : "+v"(zo00h[vv]), "+v"(zo10h[vv]), "+v"(zo11h[vv]));
        }
        xv_offh = lane < d ? 4 * lane : dump0;
        asm volatile("" : "+v"(xv_offh));
    }
    auto flush_out = [&](long long bp, bool on) {
        float *Rb = args.R + (on ? bp : 0) * (long long)out_row;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(Rb, 0, on ? out_row * 4 : 0, 0x00020000);
#pragma unroll
        for (int h = 0; h < NFL; h++) {
            const float4 v = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(my_out) + fl_lds[h]);
            u32x4 u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
            __builtin_amdgcn_raw_buffer_store_b128(u, rs, fl_off[h], 0, 2);
        }
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(*reinterpret_cast<const float *>(reinterpret_cast<const char *>(my_out) + fl_tail_lds)),
                                              rs, fl_tail_off, 0, 2);
    };

    // ---- request the rows of all D samples of this wave -----------------------------------------------------------------
    // class 2 (u4): the WM-byte window starts at min(q W2, RB2 - WM); the wanted bytes sit sh2 bytes into it
    const int st2 = q * W2 < RB2 - WM ? q * W2 : RB2 - WM;
    const int sh2 = (q * W2 - st2) * 8;                        // bits
    constexpr int rst2 = REM ? (4 * W2 + 4 <= RB2 ? 4 * W2 : RB2 - 4) : 0;   // REM chunk of a u4 row: 2 bytes at 4 W2, loaded as 4
    constexpr int rsh2 = REM ? (4 * W2 - rst2) * 8 : 0;
    // FOLD (d = 36; see evs_fused_rfq.hip): the launch pays per random LINE REQUEST, and the tail chunk as a load of its own is
    // a second request for the row's line.  Every lane loads 12 bytes instead: a u8 row at 8 q (k-slot 3: bytes 24..35 = its
    // chunks + the tail), a u4 row at min(4 q, 6) (k-slot 2 and 3: bytes 6..17 -- the wanted 4 bytes sit 2 / 6 bytes into the
    // window, the tail in its last two); the tail reaches the other k-slots by ds_bpermute at decode time.
    constexpr bool FOLD = CQ == 2 && REM == 1 && EVS_MIXED_FOLD;
    constexpr int LDdw = FOLD ? 3 : WMdw;
    const int fst2 = 4 * q < 6 ? 4 * q : 6;
    unsigned rmain[D][NR][LDdw], rrem[D][NR][REM > 0 ? REM : 1];
    float rx[D];
    unsigned cls2 = 0;   // bit u * NR + rr: the row is of class 2
#pragma unroll
    for (int u = 0; u < D; u++) {
        const int m = wave_in_block + 4 * u;
        const bool phantom = u >= n_samples;
#pragma unroll
        for (int rr = 0; rr < NR; rr++) {
            const int row = r16 + 16 * rr;
            unsigned long long p = s_ptr[row * 16 + m];
            const bool c2 = s_cls[row * 16 + m] == 2 && !phantom;
            if (phantom) p = zc_p;
            if constexpr (FOLD) {
                mixed_load_raw<12>(p + (c2 ? fst2 : q * W1), rmain[u][rr]);
            } else {
            mixed_load_raw<WM>(p + (c2 ? st2 : q * W1), rmain[u][rr]);
            }
            if constexpr (REM > 0 && !FOLD) {
                unsigned t[1];
                mixed_load_raw<4>(p + (c2 ? rst2 : 4 * W1), t);
                rrem[u][rr][0] = t[0];
            }
            cls2 |= (c2 ? 1u : 0u) << (u * NR + rr);
        }
        {
            const unsigned long long p = phantom ? zeros_p : (unsigned long long)reinterpret_cast<uintptr_t>(args.x + (blk_first + m) * args.x_stride + (lane < d ? lane : 0));
            rx[u] = *reinterpret_cast<const __attribute__((address_space(1))) float *>((uintptr_t)p);
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    // one 4-element chunk: u8 -> bytes of w8, u4 -> nibbles of the low 16 bits of w4 (element 2j = the HIGH nibble of byte j).
    // Branch-free (round 6): written as `c2 ? index4 : index8` per element the compiler put every select behind an exec-mask
    // region -- branches, s_and_saveexec / s_or pairs: 3.0 M scalar and 6.4 M vector wave-instructions per launch.  Here the
    // four nibbles are spread into the four bytes of a word (two masks and a byte permute), the class picks the word by a
    // bit mask (m: all ones for class 2) and the table base by the same mask, and an element is one bit-field extract and one
    // shift-add.
    constexpr unsigned lut_lane = 0u;
    auto dec = [&](unsigned w8, unsigned w4, unsigned m) -> float4 {
        const unsigned t0 = w4 & 0x0F0Fu, t1 = (w4 >> 4) & 0x0F0Fu;
        const unsigned b4 = __builtin_amdgcn_perm(t0, t1, 0x05010400u);   // bytes: e0 = t1.b0, e1 = t0.b0, e2 = t1.b1, e3 = t0.b1
        const unsigned w = (b4 & m) | (w8 & ~m);
        const unsigned base = lut_lane + (m & kU4Base);
        const char *lut = reinterpret_cast<const char *>(s_lut);
        return make_float4(*reinterpret_cast<const float *>(lut + ((w & 255u) << kLutShift) + base),
                           *reinterpret_cast<const float *>(lut + (((w >> 8) & 255u) << kLutShift) + base),
                           *reinterpret_cast<const float *>(lut + (((w >> 16) & 255u) << kLutShift) + base),
                           *reinterpret_cast<const float *>(lut + ((w >> 24) << kLutShift) + base));
    };

#pragma unroll
    for (int u = 0; u < D; u++) {
        const long long b = blk_first + wave_in_block + 4 * (long long)u;
        my_x[lane] = rx[u];
        float4 a[NR][NC];
#pragma unroll
        for (int rr = 0; rr < NR; rr++) {
            const unsigned m2 = 0u - ((cls2 >> (u * NR + rr)) & 1u);   // all ones: the row is of class 2
            // class 2: the wanted W2 bytes, shifted down to bit 0 of a 32-bit word
            unsigned lo4;
            if constexpr (FOLD) {
                // the u4 window of 12 bytes: k-slots 0 / 1 want its first word, 2 the word 2 bytes in, 3 the word 6 bytes in
                const unsigned w0 = rmain[u][rr][0], w1 = rmain[u][rr][1], w2 = rmain[u][rr][2];
                lo4 = q < 2 ? w0 : __builtin_amdgcn_alignbit(q == 2 ? w1 : w2, q == 2 ? w0 : w1, 16);
                // the tail: k-slot 3 holds it -- u8 bytes 32..35 = its third word, u4 bytes 16..17 = the top half of it
                rrem[u][rr][0] = (unsigned)__builtin_amdgcn_ds_bpermute(4 * (48 + r16), (int)(((w2 >> 16) & m2) | (w2 & ~m2)));
            } else
            if constexpr (WMdw == 2) lo4 = (unsigned)((((unsigned long long)rmain[u][rr][1] << 32) | rmain[u][rr][0]) >> sh2);
            else lo4 = rmain[u][rr][0] >> sh2;
#pragma unroll
            for (int c = 0; c < CQ; c++) {
                const unsigned w8 = rmain[u][rr][WMdw == 2 ? c : 0];
                a[rr][c] = dec(w8, lo4 >> (16 * c), m2);
            }
            if constexpr (REM > 0) {   // k-slot q feeds only element q of the REM chunk to the matrix core
                const unsigned w8 = rrem[u][rr][0], w4 = FOLD ? rrem[u][rr][0] : rrem[u][rr][0] >> rsh2;
                const unsigned i8 = (w8 >> (8 * q)) & 255u;
                const unsigned i4 = (w4 >> (8 * (q >> 1) + ((q & 1) ? 0 : 4))) & 15u;
                const unsigned ix = (i4 & m2) | (i8 & ~m2);
                a[rr][CQ] = make_float4(*reinterpret_cast<const float *>(reinterpret_cast<const char *>(s_lut) + (ix << kLutShift) + lut_lane + (m2 & kU4Base)), 0.f, 0.f, 0.f);
            }
        }
        {   // row 0 is x: plain fp32 chunks, spread through the wave's LDS slot
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
        const float xv = rx[u];
        flush_out(b - 4, u > 0 && u - 1 < n_samples);
        f32x4 c00 = {0.f, 0.f, 0.f, 0.f}, c10 = c00, c11 = c00;
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
            } else {
                c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(e0[0], e0[0], c00, 0, 0, 0);
                if constexpr (NT == 2) {
                    c10 = __builtin_amdgcn_mfma_f32_16x16x4f32(e1[0], e0[0], c10, 0, 0, 0);
                    c11 = __builtin_amdgcn_mfma_f32_16x16x4f32(e1[0], e1[0], c11, 0, 0, 0);
                }
            }
        }
        *reinterpret_cast<float *>(reinterpret_cast<char *>(my_out) + xv_offh) = xv;
#pragma unroll
        for (int vv = 0; vv < 4; vv++) {
            *reinterpret_cast<float *>(reinterpret_cast<char *>(my_out) + zo00h[vv]) = c00[vv];
            if constexpr (NT == 2) {
                *reinterpret_cast<float *>(reinterpret_cast<char *>(my_out) + zo10h[vv]) = c10[vv];
                *reinterpret_cast<float *>(reinterpret_cast<char *>(my_out) + zo11h[vv]) = c11[vv];
            }
        }
    }
    flush_out(blk_first + wave_in_block + 12, n_samples == 4);
}

template <int CQ, int REM>
static void launch_mixed84(const Mixed84Args &a, bool probe, hipStream_t st) {
    const unsigned blocks = (unsigned)((a.m.B + 15) / 16);
    if (probe) {
        if (a.m.T + 1 > 16) hipLaunchKernelGGL((interact_mixed84_kernel<CQ, REM, 2, true>), dim3(blocks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((interact_mixed84_kernel<CQ, REM, 1, true>), dim3(blocks), dim3(256), 0, st, a);
        return;
    }
    if (a.m.T + 1 > 16) hipLaunchKernelGGL((interact_mixed84_kernel<CQ, REM, 2>), dim3(blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((interact_mixed84_kernel<CQ, REM, 1>), dim3(blocks), dim3(256), 0, st, a);
}

// is there a (u8, u4) rows-in-registers kernel for the shape (the two-tier lookup folds its probe into it)
bool mixed84_supported(int T, int d, int codec1, int codec2) {
    static const bool rfq_on = !(getenv("EVS_MIXED_RFQ") && getenv("EVS_MIXED_RFQ")[0] == '0');
    return rfq_on && codec1 == 8 && codec2 == 4 && T + 1 <= 28 && (d == 16 || d == 32 || d == 36);
}
// the two-tier probe and the interaction over the rows it finds, one launch (evs_cache.hip fills `probe`)
int probe2_interact_mixed84(long long B, int T, int d, const float *x, long long x_stride, const Probe2Args &probe, int itself,
                            float *R, hipStream_t st) {
    Mixed84Args ma;
    const int F = T + 1;
    ma.m.x = x; ma.m.x_stride = x_stride; ma.m.row_ptrs = nullptr; ma.m.row_class = nullptr; ma.m.R = R; ma.m.B = B; ma.m.T = T;
    ma.m.itself = itself ? 1 : 0; ma.m.P = itself ? F * (F + 1) / 2 : F * (F - 1) / 2; ma.m.codec1 = 8; ma.m.codec2 = 4; ma.m.default_class = 1;
    ma.zero_codes8 = zero_code_page(8); ma.zeros = zero_page(); ma.p = probe;
    if (!ma.zero_codes8 || !ma.zeros) return EVS_EHIP;
    switch (d) {
    case 16: launch_mixed84<1, 0>(ma, true, st); break;
    case 32: launch_mixed84<2, 0>(ma, true, st); break;
    case 36: launch_mixed84<2, 1>(ma, true, st); break;
    default: set_error("probe2_interact_mixed84: no kernel for d=%d", d); return EVS_EINVAL;
    }
    EVS_HIP_CHECK(hipGetLastError());
    return EVS_OK;
}

template <auto K>
static void launch_mixed(const MixedArgs &a, hipStream_t st) {
    static int per_cu = 0;
    if (!per_cu) {
        int n = 0;
        const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, K, 256, 0);
        per_cu = (e == hipSuccess && n > 0) ? n : 2;
    }
    long long blocks = (a.B + 3) / 4;
    const long long cap = (long long)kNumCu * per_cu;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(K, dim3((unsigned)blocks), dim3(256), 0, st, a);
}

// evs_cache.hip: R over x + T rows (address, class) -- d in {16, 32, 36}, T <= 31
int interact_from_mixed_rows(long long B, int T, int d, const float *x, long long x_stride, const long long *row_ptrs,
                             const unsigned char *row_class, int codec1, int codec2, int itself, float *R, hipStream_t st,
                             int default_class) {
    MixedArgs a;
    a.default_class = default_class == 2 ? 2 : 1;
    const int F = T + 1;
    a.x = x; a.x_stride = x_stride; a.row_ptrs = row_ptrs; a.row_class = row_class; a.R = R; a.B = B; a.T = T;
    a.itself = itself ? 1 : 0; a.P = itself ? F * (F + 1) / 2 : F * (F - 1) / 2; a.codec1 = codec1; a.codec2 = codec2;
    const bool nt2 = F > 16;
    if (mixed84_supported(T, d, codec1, codec2)) {
        Mixed84Args ma;
        ma.m = a; ma.zero_codes8 = zero_code_page(8); ma.zeros = zero_page(); ma.p = Probe2Args{};
        if (ma.zero_codes8 && ma.zeros) {
            switch (d) {
            case 16: launch_mixed84<1, 0>(ma, false, st); break;
            case 32: launch_mixed84<2, 0>(ma, false, st); break;
            default: launch_mixed84<2, 1>(ma, false, st); break;
            }
            EVS_HIP_CHECK(hipGetLastError());
            return EVS_OK;
        }
    }
    switch (d) {
    case 16: if (nt2) launch_mixed<interact_mixed_rows_kernel<1, 0, 2>>(a, st); else launch_mixed<interact_mixed_rows_kernel<1, 0, 1>>(a, st); break;
    case 32: if (nt2) launch_mixed<interact_mixed_rows_kernel<2, 0, 2>>(a, st); else launch_mixed<interact_mixed_rows_kernel<2, 0, 1>>(a, st); break;
    case 36: if (nt2) launch_mixed<interact_mixed_rows_kernel<2, 1, 2>>(a, st); else launch_mixed<interact_mixed_rows_kernel<2, 1, 1>>(a, st); break;
    default: set_error("interact_from_mixed_rows: no kernel for d=%d", d); return EVS_EINVAL;
    }
    EVS_HIP_CHECK(hipGetLastError());
    return EVS_OK;
}

}  // namespace evs
