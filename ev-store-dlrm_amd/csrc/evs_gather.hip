// Fused multi-table EmbeddingBag(sum) gather for gfx950 (MI355X).
//
// Replaces the per-table Python loop of DLRM_Net.apply_emb (reference
// dlrm_s_pytorch.py:407-461: T separate nn.EmbeddingBag launches) by ONE launch
// over all T tables.  HBM-bound random-row gather; design notes:
//   * a "lane group" of LPR = d/4 lanes owns one bag; each lane carries 4
//     consecutive elements of the pooled row, so an fp32 row is read as
//     LPR x 16-byte loads (global_load_dwordx4) and written as LPR x 16-byte
//     stores; u16 / u8 / u4 rows are read as 8 / 4 / 2-byte pieces and decoded
//     in registers (decode-on-load, fp32 accumulate).
//   * every lane group works on UNROLL bags at once so that UNROLL independent
//     row fetches are in flight per lane (bag size 1 -- Criteo -- is a pure
//     gather: latency is hidden by memory-level parallelism, not by the loop).
//   * work items are ordered table-major and split over the 8 XCDs in
//     contiguous ranges (evs_common.h:xcd_range) so that each XCD's 4 MiB L2
//     caches hot rows of ~T/8 tables instead of all T.
//   * summation order is the index order, fp32, multiply and add unfused
//     (library is built with -ffp-contract=off) -> bit-exact with the oracle.
#include "evs_common.h"
#include <stdlib.h>

namespace evs {

struct GatherArgs {
    const void *table[EVS_MAX_TABLES_PER_LAUNCH];
    const int64_t *indices[EVS_MAX_TABLES_PER_LAUNCH];
    const int64_t *offsets[EVS_MAX_TABLES_PER_LAUNCH];
    const float *row_w[EVS_MAX_TABLES_PER_LAUNCH];
    int64_t n_rows[EVS_MAX_TABLES_PER_LAUNCH];
    int64_t nnz[EVS_MAX_TABLES_PER_LAUNCH];
    // row-split tables (sharded.py "rowsplit"): table[t] holds the global rows [row_lo, row_lo + n_rows) of a table of
    // row_total rows; an index outside that range but inside the table belongs to another rank and contributes nothing,
    // silently (whole tables: row_lo = 0, row_total = n_rows -- every valid index is in range)
    int64_t row_lo[EVS_MAX_TABLES_PER_LAUNCH];
    int64_t row_total[EVS_MAX_TABLES_PER_LAUNCH];
    float *out;
    int64_t out_tstride, out_bstride;
    // peer-major output (the all-to-all send buffer): bag b goes to out + (b / bags_per_peer) * out_pstride + t * out_tstride
    // + (b % bags_per_peer) * out_bstride; bags_per_peer >= B: one "peer", the plain layout
    int64_t out_pstride;
    unsigned bags_per_peer;
    // p2p exchange (evs_embedding_bag_sum_p2p): device array of one entry per peer -- what to ADD (in floats) to the local-
    // layout address of peer q's block so that it lands in peer q's receive buffer (IPC-mapped); NULL: the local layout
    const int64_t *peer_delta;
    int64_t B;
    int64_t chunks_per_table;  // ceil(B / bags-per-wave-item)
    int T;
    int d;
    int *err;
    const void *zeros;   // the zero page (rows-in-registers gather: what a lane reads for a bag that has no row)
};

template <int CODEC>
struct RowPiece;  // 4 consecutive elements of a row -> float4

template <>
struct RowPiece<32> {
    static constexpr int kBytes = 16;
    __device__ static __forceinline__ float4 load(const void *row, int piece, const float *) {
        return reinterpret_cast<const float4 *>(row)[piece];
    }
};
template <>
struct RowPiece<16> {
    static constexpr int kBytes = 8;
    __device__ static __forceinline__ float4 load(const void *row, int piece, const float *lut) {
        const uint2 v = reinterpret_cast<const uint2 *>(row)[piece];  // 4 native-endian ushorts
        return dec_chunk<16>(v.x, v.y, lut);
    }
};
template <>
struct RowPiece<8> {
    static constexpr int kBytes = 4;
    __device__ static __forceinline__ float4 load(const void *row, int piece, const float *lut) {
        return dec_chunk<8>(reinterpret_cast<const unsigned *>(row)[piece], 0u, lut);
    }
};
template <>
struct RowPiece<4> {
    static constexpr int kBytes = 2;
    __device__ static __forceinline__ float4 load(const void *row, int piece, const float *lut) {
        return dec_chunk<4>(reinterpret_cast<const unsigned short *>(row)[piece], 0u, lut);  // byte0 | byte1<<8
    }
};

// LPR = lanes per row (d = 4*LPR).  LPR_T > 0: compile-time, LPR_T == 0: runtime (args.d/4).
template <int CODEC, int LPR_T, int UNROLL, bool BAG1>
__global__ void __launch_bounds__(256) embedding_bag_sum_kernel(const GatherArgs args) {
    // BAG1: offsets == NULL for every table = one index per bag (bag b reads indices[b]): no offsets loads in front
    // of the index load, no wave-wide trip count -- the row gather of the sharded step's send buffer.
    __shared__ float s_lut[CodecLut<CODEC>::kEntries];
    if constexpr (CODEC != 32) {
        codec_lut_init<CODEC>(s_lut);
        __syncthreads();
    }
    const int LPR = LPR_T > 0 ? LPR_T : args.d / 4;
    const int RPW = kWave / LPR;  // bags per wave per unroll slot
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const int slot = lane / LPR;
    const int piece = lane - slot * LPR;
    const bool lane_on = slot < RPW;
    const int64_t B = args.B;
    const int64_t row_bytes = (int64_t)LPR * RowPiece<CODEC>::kBytes;
    const int bags_per_item = RPW * UNROLL;

    const int64_t n_items = (int64_t)args.T * args.chunks_per_table;
    const XcdRange xr = xcd_range(n_items, 4, wave);
    bool bad = false;

    for (int64_t item = xr.first; item < xr.end; item += xr.stride) {
        // wave-uniform table id -> per-table descriptors come from scalar loads
        const int t = __builtin_amdgcn_readfirstlane((int)(item / args.chunks_per_table));
        const int64_t chunk = item - (int64_t)t * args.chunks_per_table;
        const int64_t b0 = chunk * bags_per_item;
        const char *__restrict__ W = reinterpret_cast<const char *>(args.table[t]);
        const int64_t *__restrict__ idx = args.indices[t];
        const int64_t *__restrict__ off = args.offsets[t];
        const float *__restrict__ rw = args.row_w[t];
        const int64_t n_rows = args.n_rows[t];
        const int64_t nnz = args.nnz[t];
        const int64_t row_lo = args.row_lo[t], row_total = args.row_total[t];

        int64_t s[UNROLL], len[UNROLL];
        // offsets given: idx[b] is requested together with offsets[b] / offsets[b + 1] on the bet that bag b is {idx[b]} (the
        // Criteo collate's arange offsets, dlrm_data_pytorch.py:407-408); a bag that starts at b takes its first index from
        // there -- two dependent round trips in front of the rows instead of three -- any other bag reads idx[start] as before
        int64_t spec[UNROLL], bpos[UNROLL];
        int64_t maxlen = 0;
        if constexpr (BAG1) {
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                const int64_t b = b0 + (int64_t)u * RPW + slot;
                s[u] = b;
                len[u] = (lane_on && b < B) ? 1 : 0;
            }
            maxlen = 1;
        } else {
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                const int64_t b = b0 + (int64_t)u * RPW + slot;
                s[u] = 0;
                len[u] = 0;
                bpos[u] = b;
                spec[u] = -1;
                if (lane_on && b < B && b < nnz) spec[u] = idx[b];
                if (lane_on && b < B) {
                    const int64_t st = off[b];
                    const int64_t en = (b + 1 < B) ? off[b + 1] : nnz;
                    if (st >= 0 && en >= st && en <= nnz) {
                        s[u] = st;
                        len[u] = en - st;
                    } else {
                        bad = true;
                    }
                }
                maxlen = len[u] > maxlen ? len[u] : maxlen;
            }
            // wave-wide max bag length (all lanes must run the same trip count)
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                const int64_t o = __shfl_xor(maxlen, m, kWave);
                maxlen = o > maxlen ? o : maxlen;
            }
        }

        float4 acc[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);

        for (int64_t j = 0; j < maxlen; j++) {
            int64_t r[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                r[u] = -1;
                if (j < len[u]) {
                    int64_t v;
                    if constexpr (BAG1) v = idx[s[u] + j];
                    else if (j == 0 && s[u] == bpos[u]) v = spec[u];
                    else v = idx[s[u] + j];
                    if (v >= 0 && v < row_total) {
                        v -= row_lo;
                        if (v >= 0 && v < n_rows) r[u] = v;   // (else: another rank's rows)
                    } else {
                        bad = true;
                    }
                }
            }
            float4 v[UNROLL];
            float w[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                w[u] = 1.0f;
                if (r[u] >= 0) {
                    v[u] = RowPiece<CODEC>::load(W + r[u] * row_bytes, piece, s_lut);
                    if (rw) w[u] = rw[r[u]];
                }
            }
            if (rw) {
#pragma unroll
                for (int u = 0; u < UNROLL; u++) {
                    if (r[u] >= 0) {
                        acc[u].x = __fadd_rn(acc[u].x, __fmul_rn(v[u].x, w[u]));
                        acc[u].y = __fadd_rn(acc[u].y, __fmul_rn(v[u].y, w[u]));
                        acc[u].z = __fadd_rn(acc[u].z, __fmul_rn(v[u].z, w[u]));
                        acc[u].w = __fadd_rn(acc[u].w, __fmul_rn(v[u].w, w[u]));
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < UNROLL; u++) {
                    if (r[u] >= 0) {
                        acc[u].x = __fadd_rn(acc[u].x, v[u].x);
                        acc[u].y = __fadd_rn(acc[u].y, v[u].y);
                        acc[u].z = __fadd_rn(acc[u].z, v[u].z);
                        acc[u].w = __fadd_rn(acc[u].w, v[u].w);
                    }
                }
            }
        }

        float *__restrict__ out = args.out + (int64_t)t * args.out_tstride + (int64_t)piece * 4;
        const bool peers = (int64_t)args.bags_per_peer < B;   // kernel-uniform
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const int64_t b = b0 + (int64_t)u * RPW + slot;
            if (lane_on && b < B) {
                int64_t o = b * args.out_bstride;
                if (peers) {
                    const unsigned q = (unsigned)b / args.bags_per_peer;
                    o = (int64_t)q * args.out_pstride + (b - (int64_t)q * args.bags_per_peer) * args.out_bstride;
                    if (args.peer_delta) o += args.peer_delta[q];
                }
                *reinterpret_cast<float4 *>(out + o) = acc[u];
            }
        }
    }
    if (bad) atomicOr(args.err, 1);
}

// any d (not a multiple of 4, or > 256): one element per lane, one bag per wave iteration.
template <int CODEC>
__global__ void __launch_bounds__(256) embedding_bag_sum_scalar_kernel(const GatherArgs args) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const int d = args.d;
    const int64_t n_items = (int64_t)args.T * args.B;
    const XcdRange xr = xcd_range(n_items, 4, wave);
    bool bad = false;
    for (int64_t item = xr.first; item < xr.end; item += xr.stride) {
        const int t = __builtin_amdgcn_readfirstlane((int)(item / args.B));
        const int64_t b = item - (int64_t)t * args.B;
        const int64_t nnz = args.nnz[t];
        const int64_t *off = args.offsets[t];  // NULL: one index per bag
        int64_t st = off ? off[b] : b;
        int64_t en = off ? ((b + 1 < args.B) ? off[b + 1] : nnz) : b + 1;
        if (!(st >= 0 && en >= st && en <= nnz)) { bad = true; en = st = 0; }
        const unsigned char *W = reinterpret_cast<const unsigned char *>(args.table[t]);
        const float *rw = args.row_w[t];
        int64_t ob = b * args.out_bstride;
        if ((int64_t)args.bags_per_peer < args.B) {
            const unsigned q = (unsigned)b / args.bags_per_peer;
            ob = (int64_t)q * args.out_pstride + (b - (int64_t)q * args.bags_per_peer) * args.out_bstride;
            if (args.peer_delta) ob += args.peer_delta[q];
        }
        float *out = args.out + (int64_t)t * args.out_tstride + ob;
        for (int c = lane; c < d; c += kWave) {
            float acc = 0.f;
            for (int64_t j = st; j < en; j++) {
                int64_t r = args.indices[t][j];
                if (r < 0 || r >= args.row_total[t]) { bad = true; continue; }
                r -= args.row_lo[t];
                if (r < 0 || r >= args.n_rows[t]) continue;   // another rank's rows
                float v;
                if (CODEC == 32) v = reinterpret_cast<const float *>(W)[r * d + c];
                else if (CODEC == 16) v = dec_u16(reinterpret_cast<const unsigned short *>(W)[r * d + c]);
                else if (CODEC == 8) v = dec_u8(W[r * d + c]);
                else {
                    const unsigned byte = W[r * (d / 2) + c / 2];
                    v = kU4Lut[(c & 1) ? (byte & 15u) : (byte >> 4)];
                }
                if (rw) v = __fmul_rn(v, rw[r]);
                acc = __fadd_rn(acc, v);
            }
            out[c] = acc;
        }
    }
    if (bad) atomicOr(args.err, 1);
}

template <int CODEC, int LPR_T, int UNROLL>
static void launch_vec(const GatherArgs &a, int lpr, hipStream_t stream, bool bag1) {
    GatherArgs args = a;
    const int rpw = kWave / lpr;
    const int bags_per_item = rpw * UNROLL;
    args.chunks_per_table = (a.B + bags_per_item - 1) / bags_per_item;
    const int64_t n_items = (int64_t)a.T * args.chunks_per_table;
    int64_t blocks = (n_items + 3) / 4;
    static const int cap_per_cu = getenv("EVS_GATHER_BLOCKS_PER_CU") ? atoi(getenv("EVS_GATHER_BLOCKS_PER_CU")) : 8;   // (developer A/B)
    const int64_t cap = (int64_t)kNumCu * cap_per_cu;  // 8 blocks of 256 threads per CU
    if (blocks > cap) blocks = cap;
    blocks = round_up((int)blocks, kNumXcd);
    if (bag1)
        hipLaunchKernelGGL((embedding_bag_sum_kernel<CODEC, LPR_T, UNROLL, true>), dim3((unsigned)blocks), dim3(256), 0,
                           stream, args);
    else
        hipLaunchKernelGGL((embedding_bag_sum_kernel<CODEC, LPR_T, UNROLL, false>), dim3((unsigned)blocks), dim3(256),
                           0, stream, args);
}

// ---- one index per bag, fp32, d in {16, 32, 36}: rows in flight in registers (round 3) ----------------------------------
// The gather of the Criteo layout is a pure row copy: bag b of table t is row idx[t][b].  The grid-stride kernel above keeps
// UNROLL = 4 rows in flight per lane group and pays an index round trip per item; this form is the fused kernel's
// (evs_fused_rf.hip) without its interaction: a block owns 16 samples of ALL tables, the 16 x T indices arrive in ONE round
// trip (thread e owns (table e >> 4, sample e & 15), and e + 256), go through a 2 KB LDS tile, and every wave requests the
// rows of its 4 samples at once -- LPRD lanes per row, RPI = 64 / LPRD rows per load instruction, NJ instructions per
// sample, 4 NJ in flight per lane -- and stores them as they arrive (straight-line code: counted vmcnt waits).
// CHECK (offsets given): the same threads also load offsets[b] (and the neighbour's, = where bag b ends) and check that
// every bag of the block is exactly {idx[b]}; a block that finds anything else pools its 16 x T bags the general way
// (index order, unfused fp32 adds, bad offsets / indices skipped and flagged: the arithmetic of the kernel above).
// Row ranges (row_lo / row_total) and the peer-major output of the sharded step are honoured.
typedef float gr_f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) gr_f32x4 *gr_gf4_t;
typedef const __attribute__((address_space(1))) int64_t *gr_gi64_t;

// CODEC (round 4): the reduced-precision row formats through the same mapping -- a lane's piece of a row is 4 consecutive
// elements = 8 / 4 / 2 encoded bytes (u16 / u8 / u4), in flight raw, decoded through the per-block LDS tables when it is
// stored as 16 bytes of fp32; a row that is not there reads the page of the code that decodes to 0.0f (zero_code_page).
// One request per row and instruction as for fp32 (the grid-stride kernel keeps 4 rows per lane group in flight and pays
// an index round trip per item: u8 at B = 65 536 132 us against 62 here).
template <int CODEC> struct GrPiece { typedef gr_f32x4 type; };
template <> struct GrPiece<16> { typedef unsigned type __attribute__((ext_vector_type(2))); };
template <> struct GrPiece<8> { typedef unsigned type; };
template <> struct GrPiece<4> { typedef unsigned short type; };
template <int CODEC, int LPRD, int NJ, bool CHECK>
__global__ void __launch_bounds__(256, (NJ > 5 ? 2 : 4)) gather_rows_kernel(const GatherArgs args) {
    constexpr int RPI = 64 / LPRD;
    constexpr int PB = CODEC / 2;                     // bytes of a lane's piece (4 elements)
    constexpr int row_bytes = LPRD * PB;
    typedef typename GrPiece<CODEC>::type piece_t;
    typedef const __attribute__((address_space(1))) piece_t *gpiece_t;
    static_assert(NJ * RPI <= 32, "one tile row per table");
    __shared__ float s_lut[CodecLut<CODEC>::kEntries];
    if constexpr (CODEC != 32) codec_lut_init<CODEC>(s_lut);   // (the barrier behind the index tile covers it)
    __shared__ int s_idx[32 * 16];                    // [table][sample]: row of the table's local range, -1 = nothing to read
    __shared__ unsigned long long s_base[32];
    __shared__ unsigned long long s_obase[32];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int T = args.T;
    const int64_t B = args.B;
    const GatherArgs *ka = (const GatherArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    const int64_t blk_first = (int64_t)blockIdx.x * 16;
    const int64_t blk_end = blk_first + 16 < B ? blk_first + 16 : B;
    if (blk_first >= blk_end) return;   // block-uniform
    if (threadIdx.x < 32) {             // every array has 32 entries: unconditional loads, one round trip
        const int t = (int)threadIdx.x;
        s_base[t] = (unsigned long long)reinterpret_cast<uintptr_t>(ka->table[t]);
        s_obase[t] = (unsigned long long)reinterpret_cast<uintptr_t>(args.out) + (unsigned long long)((int64_t)t * args.out_tstride * 4);
    }
    // ---- the block's index tile ---------------------------------------------------------------------------------------
    bool bad = false, ragged = false;
    {
        const int64_t bs = blk_first + (threadIdx.x & 15);
        int64_t v[2], o0[2] = {0, 0}, o1[2] = {0, 0};
        int64_t nr[2], lo[2], tot[2], nz[2];
        bool on[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int t = ((int)threadIdx.x >> 4) + 16 * h;   // < 32
            on[h] = t < T && bs < blk_end;
            const int64_t *ip = ka->indices[t], *op = ka->offsets[t];
            nr[h] = ka->n_rows[t]; lo[h] = ka->row_lo[t]; tot[h] = ka->row_total[t]; nz[h] = ka->nnz[t];
            const int64_t *ap = (on[h] && ip && bs < nz[h]) ? ip + bs : reinterpret_cast<const int64_t *>(&ka->B);
            v[h] = *reinterpret_cast<gr_gi64_t>(reinterpret_cast<uintptr_t>(ap));
            if (!(on[h] && ip && bs < nz[h])) v[h] = -1;
            if constexpr (CHECK) {
                const int64_t *p0 = (on[h] && op) ? op + bs : reinterpret_cast<const int64_t *>(&ka->B);
                // where bag b ends: the next bag's start (the neighbour lane has it, except behind the block's last sample)
                const bool own = on[h] && ((threadIdx.x & 15) == 15 || bs + 1 >= blk_end);
                const int64_t *p1 = (own && op && bs + 1 < B) ? op + bs + 1 : reinterpret_cast<const int64_t *>(&ka->B);
                o0[h] = *reinterpret_cast<gr_gi64_t>(reinterpret_cast<uintptr_t>(p0));
                o1[h] = *reinterpret_cast<gr_gi64_t>(reinterpret_cast<uintptr_t>(p1));
            }
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int t = ((int)threadIdx.x >> 4) + 16 * h;
            int id = -1;
            if (on[h]) {
                if constexpr (CHECK) {
                    // (a bag ends where the next one starts = the next lane's o0, which that lane checks itself)
                    const bool own = (threadIdx.x & 15) == 15 || bs + 1 >= blk_end;
                    const int64_t en = bs + 1 < B ? o1[h] : nz[h];
                    ragged |= !((o0[h] == bs) & (!own | (en == bs + 1)));
                }
                if (v[h] >= 0 && v[h] < tot[h]) {
                    const int64_t r = v[h] - lo[h];
                    if (r >= 0 && r < nr[h]) id = (int)r;   // (else: another rank's rows, silently)
                } else {
                    bad = true;
                }
            }
            s_idx[t * 16 + ((int)threadIdx.x & 15)] = id;
        }
    }
    const bool peers = (int64_t)args.bags_per_peer < B;   // kernel-uniform
    auto out_off = [&](int64_t b) -> int64_t {            // element offset of bag b inside a table's block of the output
        if (!peers) return b * args.out_bstride;
        const unsigned q = (unsigned)b / args.bags_per_peer;
        return (int64_t)q * args.out_pstride + (b - (int64_t)q * args.bags_per_peer) * args.out_bstride + (args.peer_delta ? args.peer_delta[q] : 0);
    };
    if constexpr (CHECK) {
        if (__syncthreads_or(ragged)) {
            // ---- rare: this block's bags with general semantics, one (bag, 16-byte piece) per thread and trip ----------------
            bad = false;   // (the verdict on an index belongs to the bag that holds it)
            const int n_work = T * 16 * LPRD;
            for (int wk = (int)threadIdx.x; wk < n_work; wk += 256) {
                const int piece = wk % LPRD, bag = wk / LPRD;
                const int t = bag >> 4;
                const int64_t b = blk_first + (bag & 15);
                if (b >= blk_end) continue;
                const int64_t *ip = ka->indices[t], *op = ka->offsets[t];
                const int64_t nzt = ka->nnz[t];
                int64_t st = op[b], en = (b + 1 < B) ? op[b + 1] : nzt;
                if (!(st >= 0 && en >= st && en <= nzt)) { bad = true; st = en = 0; }
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                const char *W = reinterpret_cast<const char *>(ka->table[t]);
                const int64_t tot_t = ka->row_total[t], lo_t = ka->row_lo[t], nr_t = ka->n_rows[t];
                for (int64_t j = st; j < en; j++) {
                    int64_t r = ip[j];
                    if (r < 0 || r >= tot_t) { bad = true; continue; }
                    r -= lo_t;
                    if (r < 0 || r >= nr_t) continue;
                    const float4 x = RowPiece<CODEC>::load(W + r * row_bytes, piece, s_lut);
                    acc.x = __fadd_rn(acc.x, x.x); acc.y = __fadd_rn(acc.y, x.y); acc.z = __fadd_rn(acc.z, x.z); acc.w = __fadd_rn(acc.w, x.w);
                }
                *reinterpret_cast<float4 *>(args.out + (int64_t)t * args.out_tstride + out_off(b) + piece * 4) = acc;
            }
            if (bad) atomicOr(args.err, 1);
            return;
        }
    } else {
        __syncthreads();
    }
#ifndef EVS_GR_FLAT
#define EVS_GR_FLAT 1
#endif
    // ---- table-major output (consecutive bags of a table are consecutive rows: the reference's list of T (B, d) tensors, the
    // blocks of the all-to-all send buffer): the FLAT mapping (round 5).  The block's 16 rows of one table are one aligned run
    // of 16 LPRD 16-byte pieces; piece P of the block = (table P / (16 LPRD), bag (P % (16 LPRD)) / LPRD, piece P % LPRD), and
    // instruction i of wave w takes pieces [256 i + 64 w, + 64): every store instruction is 1 024 contiguous bytes that start on a
    // line -- whole lines only.  The sample-major mapping below writes a table-major output as 144-byte runs, two partial lines
    // each, and the launch is priced by its line requests, loads and stores alike (tools/gather_ablate.sh: u8 at B = 65 536,
    // loads only 33 us, stores only 58, both 109).  (The same over the sample-major interaction tile -- the block's 16 bags as one
    // address range, chunks aligned to its lines -- was built and measured equal, 113 against 112 us at B = 65 536 fp32: the mapping
    // below already writes that layout as 1 008-byte runs.)
    if (EVS_GR_FLAT && args.out_bstride == 4 * LPRD) {   // kernel-uniform
        constexpr int PT = 16 * LPRD;
        const int n_pieces = T * PT;
        const int blk_n = (int)(blk_end - blk_first);
        piece_t fring[4 * NJ];
#pragma unroll
        for (int i = 0; i < 4 * NJ; i++) {
            const int P = 256 * i + 64 * wave + lane;
            const int Pc = P < n_pieces ? P : 0;
            const int t = Pc / PT, q = Pc - t * PT, sm = q / LPRD, piece = q - sm * LPRD;
            const int iv = (P < n_pieces && sm < blk_n) ? s_idx[t * 16 + sm] : -1;
            const unsigned long long neg = 0ull - (unsigned long long)((unsigned)iv >> 31);
            const unsigned long long p = s_base[t] + (unsigned long long)(piece * PB) + (unsigned long long)((unsigned)iv & 0x7fffffffu) * (unsigned long long)row_bytes;
            const unsigned long long zp = (unsigned long long)reinterpret_cast<uintptr_t>(args.zeros) + (unsigned long long)(piece * PB);
            fring[i] = *reinterpret_cast<gpiece_t>((uintptr_t)(p ^ ((p ^ zp) & neg)));
            if ((i + 1) % NJ == 0) __builtin_amdgcn_sched_barrier(0);   // (address temporaries of NJ requests alive at a time, not of all 4 NJ)
        }
#pragma unroll
        for (int i = 0; i < 4 * NJ; i++) {
            int P = 256 * i + 64 * wave + lane;
            asm volatile("" : "+v"(P));   // (computed again, not carried from the request loop: 4 NJ x {table, bag, piece} in registers spill the fp32 forms)
            const int Pc = P < n_pieces ? P : 0;
            const int t = Pc / PT, q = Pc - t * PT, sm = q / LPRD, piece = q - sm * LPRD;
            gr_f32x4 o;
            if constexpr (CODEC == 32) o = fring[i];
            else {
                float4 f;
                if constexpr (CODEC == 16) f = dec_chunk<16>(fring[i].x, fring[i].y, s_lut);
                else f = dec_chunk<CODEC>((unsigned)fring[i], 0u, s_lut);
                o.x = f.x; o.y = f.y; o.z = f.z; o.w = f.w;
            }
            if (P < n_pieces && sm < blk_n)
                *reinterpret_cast<gr_f32x4 *>((uintptr_t)(s_obase[t] + (unsigned long long)(out_off(blk_first + sm) * 4) + (unsigned long long)(piece * 16))) = o;
        }
        if (bad) atomicOr(args.err, 1);
        return;
    }
    // ---- the rows of this wave's samples wave, wave + 4, wave + 8, wave + 12 ----------------------------------------------
    const int r0 = lane / LPRD, piece16 = (lane - r0 * LPRD) * 16, piece_in = (lane - r0 * LPRD) * PB;
    const bool lane_on = r0 < RPI;
    const unsigned long long zeros_p = (unsigned long long)reinterpret_cast<uintptr_t>(args.zeros) + (unsigned long long)piece_in;
    unsigned long long fbase[NJ], obase[NJ];
    int tile[NJ];
    bool t_on[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int t = (lane_on ? r0 : 0) + j * RPI;   // < 32
        t_on[j] = lane_on && t < T;
        fbase[j] = s_base[t] + (unsigned long long)piece_in;
        obase[j] = s_obase[t] + (unsigned long long)piece16;
        tile[j] = t * 16 + wave;
    }
    piece_t ring[4][NJ];
#pragma unroll
    for (int n = 0; n < 4; n++) {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int iv = s_idx[tile[j] + 4 * n];
            const unsigned long long neg = 0ull - (unsigned long long)((unsigned)iv >> 31);   // -1 -> the zero page (bit blend: no branch around the load)
            const unsigned long long p = fbase[j] + (unsigned long long)((unsigned)iv & 0x7fffffffu) * (unsigned long long)row_bytes;
#ifdef EVS_GR_NOLOAD   // (developer ablation: every row = the zero page)
            ring[n][j] = *reinterpret_cast<gpiece_t>((uintptr_t)(zeros_p + 0 * (p ^ neg)));
#else
            ring[n][j] = *reinterpret_cast<gpiece_t>((uintptr_t)(p ^ ((p ^ zeros_p) & neg)));
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int n = 0; n < 4; n++) {
        const int64_t b = blk_first + wave + 4 * n;   // wave-uniform
        if (b < blk_end) {
            const unsigned long long ob = (unsigned long long)(out_off(b) * 4);
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                gr_f32x4 o;
                if constexpr (CODEC == 32) o = ring[n][j];
                else {
                    float4 f;
                    if constexpr (CODEC == 16) f = dec_chunk<16>(ring[n][j].x, ring[n][j].y, s_lut);
                    else f = dec_chunk<CODEC>((unsigned)ring[n][j], 0u, s_lut);
                    o.x = f.x; o.y = f.y; o.z = f.z; o.w = f.w;
                }
#ifdef EVS_GR_NOSTORE   // (developer ablation: only values that cannot occur are stored)
                if (t_on[j] && o.x == 12345.678f) {
#else
                if (t_on[j]) {
#endif
                    // (ordinary stores: non-temporal ones run at half the rate for this pattern, tools/store_pattern_probe.hip)
                    *reinterpret_cast<gr_f32x4 *>((uintptr_t)(obase[j] + ob)) = o;
                }
            }
        }
    }
    if (bad) atomicOr(args.err, 1);
}

// ---- multi-hot bags, fp32, unweighted: rows through LDS, every lane busy (round 3) ------------------------------------------
// The grid-stride kernel above gives a lane group a BAG: with bags of 1..10 indices every lane of a wave loops to the
// longest of its 28 bags, an index load and a dependent row load per trip -- half the lanes idle, twice the memory
// instructions, 2 n dependent round trips for a bag of n.  Here a lane group takes a LOOKUP: a block owns 64 consecutive
// bags of one table = one contiguous run of that table's index array (offsets valid and monotone -- checked; a block that
// finds anything else pools its bags one by one, flagged as the kernel above flags them).  The run is walked in tiles of
// 16 load instructions per block: the tile's indices in one coalesced round trip, its rows in the next (LPRD lanes per row,
// every lane of every instruction busy), into LDS; then thread (bag, 16-byte piece) adds its bag's rows of the tile IN
// INDEX ORDER (sequential fp32 adds from LDS: bit-exact with the oracle's loop), carrying its partial sum across tiles.
// SEL (long bags: an average above 16 indices): the reduction skips an invalid row with a SELECT of the old sum and is unrolled
// four deep, so that the LDS reads of a long bag pipeline (a branch per row: 352 us for bags of ~38 indices, the select
// 218); short bags keep the branch (74 vs 81 us at ~5 indices per bag)
template <int LPRD, bool SEL>
__global__ void __launch_bounds__(256) bag_sum_flat_kernel(const GatherArgs args) {
    constexpr int RPI = 64 / LPRD;              // rows per load instruction
    constexpr int kInstr = 4;                   // load instructions per wave and tile
    constexpr int TILE = 4 * kInstr * RPI;      // rows per tile (d = 36: 112 rows = 16 KB)
    constexpr int BAGS = 64;
    constexpr int row_bytes = LPRD * 16;
    constexpr int kItems = (BAGS * LPRD + 255) / 256;   // (bag, piece) sums per thread
    __shared__ __attribute__((aligned(16))) float4 s_tile[TILE * LPRD];
    __shared__ unsigned char s_ok[TILE];
    __shared__ int64_t s_st[BAGS + 1];          // bag starts; [nb] = the end of the last bag
    __shared__ int s_bad;
    const int64_t B = args.B;
    const int chunks = (int)((B + BAGS - 1) / BAGS);
    const int t = (int)(blockIdx.x / chunks);
    const int64_t b0 = (int64_t)(blockIdx.x - (unsigned)t * chunks) * BAGS;
    const int nb = (int)(B - b0 < BAGS ? B - b0 : BAGS);
    const GatherArgs *ka = (const GatherArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    const int64_t *idx = ka->indices[t], *off = ka->offsets[t];
    const char *W = reinterpret_cast<const char *>(ka->table[t]);
    const int64_t nnz = ka->nnz[t], n_rows = ka->n_rows[t], row_lo = ka->row_lo[t], row_total = ka->row_total[t];
    if (threadIdx.x == 0) s_bad = 0;
    if ((int)threadIdx.x <= nb) {
        const int64_t b = b0 + threadIdx.x;
        s_st[threadIdx.x] = b < B ? off[b] : nnz;
    }
    __syncthreads();
    bool bad = false;
    if ((int)threadIdx.x < nb) {                 // every bag of the chunk a valid [start, end) inside the index array?
        const int64_t st = s_st[threadIdx.x], en = s_st[threadIdx.x + 1];
        if (!(st >= 0 && en >= st && en <= nnz)) atomicOr(&s_bad, 1);
    }
    __syncthreads();
    const bool peers = (int64_t)args.bags_per_peer < B;
    auto out_ptr = [&](int64_t b, int piece) -> float4 * {
        int64_t o = b * args.out_bstride;
        if (peers) {
            const unsigned q = (unsigned)b / args.bags_per_peer;
            o = (int64_t)q * args.out_pstride + (b - (int64_t)q * args.bags_per_peer) * args.out_bstride;
            if (args.peer_delta) o += args.peer_delta[q];
        }
        return reinterpret_cast<float4 *>(args.out + (int64_t)t * args.out_tstride + o + piece * 4);
    };
    if (s_bad) {   // rare: a bad offset in the chunk -- bag by bag, the bad ones empty and flagged
        for (int wk = (int)threadIdx.x; wk < nb * LPRD; wk += 256) {
            const int i = wk / LPRD, piece = wk - i * LPRD;
            const int64_t b = b0 + i;
            int64_t st = off[b], en = (b + 1 < B) ? off[b + 1] : nnz;
            if (!(st >= 0 && en >= st && en <= nnz)) { bad = true; st = en = 0; }
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int64_t j = st; j < en; j++) {
                int64_t r = idx[j];
                if (r < 0 || r >= row_total) { bad = true; continue; }
                r -= row_lo;
                if (r < 0 || r >= n_rows) continue;
                const float4 x = reinterpret_cast<const float4 *>(W + r * row_bytes)[piece];
                acc.x = __fadd_rn(acc.x, x.x); acc.y = __fadd_rn(acc.y, x.y); acc.z = __fadd_rn(acc.z, x.z); acc.w = __fadd_rn(acc.w, x.w);
            }
            *out_ptr(b, piece) = acc;
        }
        if (bad) atomicOr(args.err, 1);
        return;
    }
    const int lane = threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    const int g = lane / LPRD, piece = lane - g * LPRD;
    const bool lane_on = g < RPI;
    const int64_t lo = s_st[0], hi = s_st[nb];
    float4 acc[kItems];
    int my_bag[kItems], my_piece[kItems];
#pragma unroll
    for (int k = 0; k < kItems; k++) {
        const int wk = (int)threadIdx.x + 256 * k;
        acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        my_bag[k] = wk < nb * LPRD ? wk / LPRD : -1;
        my_piece[k] = wk - (wk / LPRD) * LPRD;
    }
    for (int64_t base = lo; base < hi; base += TILE) {
        // the tile's indices (one round trip), then its rows (the next): slot = (wave * kInstr + k) * RPI + g
        int64_t v[kInstr];
#pragma unroll
        for (int k = 0; k < kInstr; k++) {
            const int64_t e = base + (wave * kInstr + k) * RPI + g;
            v[k] = (lane_on && e < hi) ? idx[e] : -2;
        }
#pragma unroll
        for (int k = 0; k < kInstr; k++) {
            const int slot = (wave * kInstr + k) * RPI + g;
            int64_t r = -1;
            if (v[k] >= 0 && v[k] < row_total) { r = v[k] - row_lo; if (r < 0 || r >= n_rows) r = -1; }
            else if (v[k] != -2) bad = true;
            if (lane_on) {
                if (r >= 0) s_tile[slot * LPRD + piece] = reinterpret_cast<const float4 *>(W + r * row_bytes)[piece];
                if (piece == 0) s_ok[slot] = r >= 0;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kItems; k++) {
            if (my_bag[k] >= 0) {
                int64_t a = s_st[my_bag[k]], b = s_st[my_bag[k] + 1];
                a = a > base ? a : base;
                b = b < base + TILE ? b : base + TILE;
                const int s0 = (int)(a - base), s1 = (int)(b - base);
                if constexpr (SEL) {
#pragma unroll 4
                    for (int slot = s0; slot < s1; slot++) {   // index order
                        const bool ok = s_ok[slot] != 0;
                        const float4 x = s_tile[slot * LPRD + my_piece[k]];
                        const float4 n4 = make_float4(__fadd_rn(acc[k].x, x.x), __fadd_rn(acc[k].y, x.y), __fadd_rn(acc[k].z, x.z), __fadd_rn(acc[k].w, x.w));
                        acc[k].x = ok ? n4.x : acc[k].x; acc[k].y = ok ? n4.y : acc[k].y; acc[k].z = ok ? n4.z : acc[k].z; acc[k].w = ok ? n4.w : acc[k].w;
                    }
                } else {
                    for (int slot = s0; slot < s1; slot++) {   // index order
                        if (s_ok[slot]) {
                            const float4 x = s_tile[slot * LPRD + my_piece[k]];
                            acc[k].x = __fadd_rn(acc[k].x, x.x); acc[k].y = __fadd_rn(acc[k].y, x.y);
                            acc[k].z = __fadd_rn(acc[k].z, x.z); acc[k].w = __fadd_rn(acc[k].w, x.w);
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < kItems; k++)
        if (my_bag[k] >= 0) *out_ptr(b0 + my_bag[k], my_piece[k]) = acc[k];
    if (bad) atomicOr(args.err, 1);
}

static bool launch_bag_sum_flat(const GatherArgs &a, bool vec_ok, hipStream_t stream) {
    static const bool on = !(getenv("EVS_GATHER_FLAT") && getenv("EVS_GATHER_FLAT")[0] == '0');
    if (!on || !vec_ok || !(a.d == 16 || a.d == 32 || a.d == 36 || a.d == 64)) return false;
    int64_t nnz = 0;
    for (int k = 0; k < a.T; k++) { if (a.row_w[k] || !a.offsets[k]) return false; nnz += a.nnz[k]; }
    // very long bags: a tile holds the rows of one or two bags and as few (bag, piece) threads do all the adding -- beyond an
    // average of 128 indices the grid-stride kernel's lane group per bag takes over (at ~38: 218 us here, 229 there)
    static const int max_avg = getenv("EVS_GATHER_FLAT_MAXAVG") ? atoi(getenv("EVS_GATHER_FLAT_MAXAVG")) : 128;
    if (nnz > (int64_t)max_avg * a.B * (int64_t)a.T) return false;
    const bool sel = nnz > 16 * a.B * (int64_t)a.T;
    const int64_t chunks = (a.B + 63) / 64;
    if (chunks * a.T >= (1ll << 31)) return false;
    const dim3 grid((unsigned)(chunks * a.T)), block(256);
#define EVS_FLAT(L) do { if (sel) hipLaunchKernelGGL((bag_sum_flat_kernel<L, true>), grid, block, 0, stream, a); \
                         else hipLaunchKernelGGL((bag_sum_flat_kernel<L, false>), grid, block, 0, stream, a); return true; } while (0)
    switch (a.d) {
    case 16: EVS_FLAT(4);
    case 32: EVS_FLAT(8);
    case 36: EVS_FLAT(9);
    default: EVS_FLAT(16);
    }
#undef EVS_FLAT
}

// ---- long bags (round 4): a lane group per BAG, K rows of it in flight, a register accumulator in index order ------------
// The reference's own benchmark shape (bench/dlrm_s_benchmark.sh:20-45: 8 tables x 1 M rows x d = 64, 100 indices per bag,
// mb 2 048) and every bag much longer than the tile kernel above likes: there a tile of 64-112 rows is one or two bags and
// as few (bag, piece) threads add all of it out of LDS (0.17 of peak at 38 indices per bag).  Here the d / 4 lanes that
// hold a pooled row own the bag: the bag's next K indices arrive as ONE coalesced load of the group (lane l takes index l),
// each is handed to the whole group (ds_bpermute), the K rows are requested back to back -- K x 64 / LPR rows in flight
// per wave -- and added to the accumulator IN INDEX ORDER as they return (straight-line code: counted vmcnt waits), the
// following chunk's indices already on the wire.  An index of another rank's rows (row ranges) or out of the table reads
// the zero page: x + 0.0f is x for every x the accumulator can hold (it starts at +0.0f and +0.0f + -0.0f = +0.0f), so the
// sums keep the bits of the oracle's loop.  Ragged bags: a wave runs to its longest bag, the others add zero pages.
template <int LPR, int K>
__global__ void __launch_bounds__(256) bag_sum_long_kernel(const GatherArgs args) {
    constexpr int G = 64 / LPR;                     // bags per wave
    constexpr int row_bytes = LPR * 16;
    constexpr int NIL = (K + LPR - 1) / LPR;        // index loads per lane and chunk
    const int lane = threadIdx.x & 63;
    const int g = lane / LPR, piece = lane - g * LPR;
    const bool lane_on = g < G;
    const int64_t B = args.B;
    const int64_t cpt = (B + G - 1) / G;            // wave items per table
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= cpt * args.T) return;               // wave-uniform
    const int t = __builtin_amdgcn_readfirstlane((int)(item / cpt));
    const int64_t b = (item - (int64_t)t * cpt) * G + g;
    const char *__restrict__ W = reinterpret_cast<const char *>(args.table[t]);
    const int64_t *__restrict__ idx = args.indices[t];
    const int64_t *__restrict__ off = args.offsets[t];
    const int64_t n_rows = args.n_rows[t], nnz = args.nnz[t], row_lo = args.row_lo[t], row_total = args.row_total[t];
    const char *zeros = reinterpret_cast<const char *>(args.zeros);
    bool bad = false;
    int64_t st = 0;
    int len = 0;
    if (lane_on && b < B) {
        const int64_t s0 = off[b], e0 = (b + 1 < B) ? off[b + 1] : nnz;
        if (s0 >= 0 && e0 >= s0 && e0 <= nnz && e0 - s0 < (1ll << 30)) { st = s0; len = (int)(e0 - s0); }
        else bad = true;
    }
    int maxlen = len;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { const int o = __shfl_xor(maxlen, m, 64); maxlen = o > maxlen ? o : maxlen; }
    // the row id (-1: nothing to add) of index e of this lane's bag
    auto row_of = [&](int e, bool in_chunk) -> int {
        int r = -1;
        if (in_chunk && e < len) {
            const int64_t v = idx[st + e];
            if (v >= 0 && v < row_total) { const int64_t q = v - row_lo; if (q >= 0 && q < n_rows) r = (int)q; }
            else bad = true;
        }
        return r;
    };
    int cur[NIL], nxt[NIL];
#pragma unroll
    for (int i = 0; i < NIL; i++) cur[i] = row_of(piece + LPR * i, piece + LPR * i < K);
    gr_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int gbase = g * LPR;
    for (int pos = 0; pos < maxlen; pos += K) {
#pragma unroll
        for (int i = 0; i < NIL; i++) nxt[i] = row_of(pos + K + piece + LPR * i, piece + LPR * i < K);   // the next chunk's indices: on the wire under this chunk's rows
        gr_f32x4 row[K];
#pragma unroll
        for (int j = 0; j < K; j++) {
            const int r = __shfl(cur[j / LPR], gbase + (j % LPR), 64);
            const char *p = r >= 0 ? W + (int64_t)r * row_bytes : zeros;
            row[j] = *reinterpret_cast<const __attribute__((address_space(1))) gr_f32x4 *>(reinterpret_cast<uintptr_t>(p + piece * 16));
        }
        __builtin_amdgcn_sched_barrier(0);   // all K requests out before the first add (the scheduler otherwise keeps 4 in flight)
#pragma unroll
        for (int j = 0; j < K; j++) {   // index order, unfused fp32 adds
            acc.x = __fadd_rn(acc.x, row[j].x); acc.y = __fadd_rn(acc.y, row[j].y);
            acc.z = __fadd_rn(acc.z, row[j].z); acc.w = __fadd_rn(acc.w, row[j].w);
        }
#pragma unroll
        for (int i = 0; i < NIL; i++) cur[i] = nxt[i];
    }
    if (lane_on && b < B) {
        int64_t o = b * args.out_bstride;
        if ((int64_t)args.bags_per_peer < B) {
            const unsigned q = (unsigned)b / args.bags_per_peer;
            o = (int64_t)q * args.out_pstride + (b - (int64_t)q * args.bags_per_peer) * args.out_bstride;
            if (args.peer_delta) o += args.peer_delta[q];
        }
        *reinterpret_cast<float4 *>(args.out + (int64_t)t * args.out_tstride + o + piece * 4) = make_float4(acc.x, acc.y, acc.z, acc.w);
    }
    if (bad) atomicOr(args.err, 1);
}

static bool launch_bag_sum_long(const GatherArgs &a, bool vec_ok, hipStream_t stream) {
    static const bool on = !(getenv("EVS_GATHER_LONG") && getenv("EVS_GATHER_LONG")[0] == '0');
    if (!on || !vec_ok || !(a.d == 16 || a.d == 32 || a.d == 36 || a.d == 64 || a.d == 128) || !zero_page()) return false;
    int64_t nnz = 0;
    for (int k = 0; k < a.T; k++) {
        if (a.row_w[k] || !a.offsets[k] || a.n_rows[k] >= (1ll << 31)) return false;
        nnz += a.nnz[k];
    }
    // from an average of 2 indices per bag on (measured, ragged 1..10-index bags at B = 16 384: 62 vs 75 us for the tile kernel; 9 per bag: 29 vs 48;
    // fixed 100 per bag: 72 vs 500) -- below that the batch is mostly one-index bags and the tile kernel keeps every lane busy
    static const int min_avg = getenv("EVS_GATHER_LONG_MINAVG") ? atoi(getenv("EVS_GATHER_LONG_MINAVG")) : 2;
    if (nnz < (int64_t)min_avg * a.B * (int64_t)a.T) return false;
    const int lpr = a.d / 4, G = 64 / lpr;
    const int64_t items = (int64_t)a.T * ((a.B + G - 1) / G);
    if ((items + 3) / 4 >= (1ll << 31)) return false;
    GatherArgs g = a;
    g.zeros = zero_page();
    const dim3 grid((unsigned)((items + 3) / 4)), block(256);
    switch (a.d) {
    case 16: hipLaunchKernelGGL((bag_sum_long_kernel<4, 8>), grid, block, 0, stream, g); break;
    case 32: hipLaunchKernelGGL((bag_sum_long_kernel<8, 8>), grid, block, 0, stream, g); break;
    case 36: hipLaunchKernelGGL((bag_sum_long_kernel<9, 9>), grid, block, 0, stream, g); break;
    case 64: hipLaunchKernelGGL((bag_sum_long_kernel<16, 16>), grid, block, 0, stream, g); break;
    default: hipLaunchKernelGGL((bag_sum_long_kernel<32, 16>), grid, block, 0, stream, g); break;
    }
    return true;
}

// is there a rows-in-registers gather for the launch, and launch it
template <int CODEC>
static bool launch_gather_rows(const GatherArgs &a, bool vec_ok, hipStream_t stream, bool bag1) {
    static const bool on = !(getenv("EVS_GATHER_RF") && getenv("EVS_GATHER_RF")[0] == '0');
    const void *zp = CODEC == 32 ? zero_page() : zero_code_page(CODEC);
    if (!on || !vec_ok || a.T > 32 || !(a.d == 16 || a.d == 32 || a.d == 36 || a.d == 64) || !zp) return false;
    for (int k = 0; k < a.T; k++) {
        if (a.row_w[k] || a.n_rows[k] >= (1ll << 31)) return false;
        if (!bag1 && a.nnz[k] != a.B) return false;   // offsets given: the bet is on whole batches of one-index bags
    }
    const int rpi = 64 / (a.d / 4);
    const int nj = (a.T + rpi - 1) / rpi;
    const dim3 grid((unsigned)((a.B + 15) / 16)), block(256);
    GatherArgs g = a;
    g.zeros = zp;
#define EVS_GR(L, N) \
    do { \
        if (bag1) hipLaunchKernelGGL((gather_rows_kernel<CODEC, L, N, false>), grid, block, 0, stream, g); \
        else hipLaunchKernelGGL((gather_rows_kernel<CODEC, L, N, true>), grid, block, 0, stream, g); \
        return true; \
    } while (0)
    if (a.d == 16) { if (nj <= 1) EVS_GR(4, 1); EVS_GR(4, 2); }
    if (a.d == 64) {   // 16 lanes per row, 4 rows per instruction: T <= 28 (7 instructions per sample: two blocks per CU)
        if (nj <= 2) EVS_GR(16, 2); if (nj <= 4) EVS_GR(16, 4); if (nj <= 7) EVS_GR(16, 7);
        return false;
    }
    if (a.d == 32) { if (nj <= 1) EVS_GR(8, 1); if (nj <= 2) EVS_GR(8, 2); if (nj <= 3) EVS_GR(8, 3); EVS_GR(8, 4); }
    if (nj <= 1) EVS_GR(9, 1); if (nj <= 2) EVS_GR(9, 2); if (nj <= 3) EVS_GR(9, 3); if (nj <= 4) EVS_GR(9, 4);
    return false;   // (d = 36, T in 29..32: a fifth instruction would address tile rows past 31 -- the grid-stride kernel)
#undef EVS_GR
}

template <int CODEC>
static void launch_codec(const GatherArgs &a, bool vec_ok, hipStream_t stream, bool bag1) {
    const int d = a.d;
    if (vec_ok && d % 4 == 0 && d <= 256) {
        const int lpr = d / 4;
        switch (lpr) {
        case 4: launch_vec<CODEC, 4, 4>(a, lpr, stream, bag1); return;    // d = 16
        case 8: launch_vec<CODEC, 8, 4>(a, lpr, stream, bag1); return;    // d = 32
        case 9: launch_vec<CODEC, 9, 4>(a, lpr, stream, bag1); return;    // d = 36 (every EVStore script)
        case 16: launch_vec<CODEC, 16, 4>(a, lpr, stream, bag1); return;  // d = 64
        case 32: launch_vec<CODEC, 32, 4>(a, lpr, stream, bag1); return;  // d = 128
        default: launch_vec<CODEC, 0, 4>(a, lpr, stream, bag1); return;
        }
    }
    GatherArgs args = a;
    args.chunks_per_table = 0;
    int64_t blocks = ((int64_t)a.T * a.B + 3) / 4;
    const int64_t cap = (int64_t)kNumCu * 8;
    if (blocks > cap) blocks = cap;
    blocks = round_up((int)blocks, kNumXcd);
    hipLaunchKernelGGL((embedding_bag_sum_scalar_kernel<CODEC>), dim3((unsigned)blocks), dim3(256), 0, stream, args);
}

}  // namespace evs

static int bag_sum_impl(int T, int64_t B, int d, int codec, const void *const *tables,
                        const int64_t *n_rows, const int64_t *row_lo, const int64_t *row_total,
                        const int64_t *const *indices, const int64_t *const *offsets, const int64_t *nnz,
                        const float *const *row_weights, float *out, int64_t out_table_stride,
                        int64_t out_bag_stride, int64_t out_peer_stride, int64_t bags_per_peer, const int64_t *peer_delta,
                        void *stream) {
    using namespace evs;
    const char *who = "evs_embedding_bag_sum";
    EVS_REQUIRE(T >= 0 && B >= 0 && d > 0, "%s: bad shape T=%d B=%lld d=%d", who, T, (long long)B, d);
    EVS_REQUIRE(codec == 32 || codec == 16 || codec == 8 || codec == 4, "%s: codec %d", who, codec);
    EVS_REQUIRE(codec != 4 || d % 2 == 0, "%s: 4-bit rows need an even d (got %d)", who, d);
    if (T == 0 || B == 0) return EVS_OK;
    EVS_REQUIRE(tables && n_rows && indices && nnz && out, "%s: NULL argument", who);
    EVS_REQUIRE(B < (1ll << 31), "%s: B %lld", who, (long long)B);
    EVS_REQUIRE((row_lo == nullptr) == (row_total == nullptr), "%s: row_lo and row_total go together", who);
    if (bags_per_peer <= 0 || bags_per_peer >= B) { bags_per_peer = B; out_peer_stride = 0; }
    // offsets == NULL, or every offsets[k] == NULL: one index per bag (bag b of table k reads indices[k][b])
    bool bag1 = offsets == nullptr;
    if (!bag1) {
        int n_null = 0;
        for (int k = 0; k < T; k++) n_null += offsets[k] == nullptr;
        EVS_REQUIRE(n_null == 0 || n_null == T, "%s: offsets NULL for %d of %d tables (all or none)", who, n_null, T);
        bag1 = n_null == T;
    }
    int *err = index_error_flag();
    if (!err) return EVS_EHIP;
    bool vec_ok = (reinterpret_cast<uintptr_t>(out) % 16 == 0) && out_table_stride % 4 == 0 &&
                  out_bag_stride % 4 == 0 && out_peer_stride % 4 == 0;
    for (int k = 0; k < T; k++) {
        EVS_REQUIRE(n_rows[k] >= 0 && nnz[k] >= 0, "%s: table %d has negative size", who, k);
        EVS_REQUIRE(tables[k] || n_rows[k] == 0, "%s: table %d is NULL", who, k);
        EVS_REQUIRE(!bag1 || nnz[k] >= B, "%s: table %d has %lld indices for %lld one-index bags", who, k,
                    (long long)nnz[k], (long long)B);
        EVS_REQUIRE(indices[k] || nnz[k] == 0, "%s: indices[%d] is NULL", who, k);
        EVS_REQUIRE(!row_lo || (row_lo[k] >= 0 && row_lo[k] + n_rows[k] <= row_total[k]),
                    "%s: table %d holds rows [%lld, %lld) of %lld", who, k, (long long)(row_lo ? row_lo[k] : 0),
                    (long long)(row_lo ? row_lo[k] + n_rows[k] : 0), (long long)(row_total ? row_total[k] : 0));
        if (reinterpret_cast<uintptr_t>(tables[k]) % 16 != 0) vec_ok = false;
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    for (int k0 = 0; k0 < T; k0 += EVS_MAX_TABLES_PER_LAUNCH) {
        const int n = (T - k0 < EVS_MAX_TABLES_PER_LAUNCH) ? T - k0 : EVS_MAX_TABLES_PER_LAUNCH;
        GatherArgs a;
        for (int k = 0; k < EVS_MAX_TABLES_PER_LAUNCH; k++) {
            const bool on = k < n;
            a.table[k] = on ? tables[k0 + k] : nullptr;
            a.indices[k] = on ? indices[k0 + k] : nullptr;
            a.offsets[k] = (on && !bag1) ? offsets[k0 + k] : nullptr;
            a.row_w[k] = (on && row_weights) ? row_weights[k0 + k] : nullptr;
            a.n_rows[k] = on ? n_rows[k0 + k] : 0;
            a.nnz[k] = on ? nnz[k0 + k] : 0;
            a.row_lo[k] = (on && row_lo) ? row_lo[k0 + k] : 0;
            a.row_total[k] = on ? (row_total ? row_total[k0 + k] : n_rows[k0 + k]) : 0;
        }
        a.out = out + (int64_t)k0 * out_table_stride;
        a.out_tstride = out_table_stride;
        a.out_bstride = out_bag_stride;
        a.out_pstride = out_peer_stride;
        a.bags_per_peer = (unsigned)bags_per_peer;
        a.peer_delta = (bags_per_peer < B) ? peer_delta : nullptr;
        a.B = B;
        a.T = n;
        a.d = d;
        a.err = err;
        a.chunks_per_table = 0;
        a.zeros = nullptr;
        if ((codec == 32 && launch_gather_rows<32>(a, vec_ok, st, bag1)) || (codec == 16 && launch_gather_rows<16>(a, vec_ok, st, bag1)) ||
            (codec == 8 && launch_gather_rows<8>(a, vec_ok, st, bag1)) || (codec == 4 && launch_gather_rows<4>(a, vec_ok, st, bag1))) {
            EVS_HIP_CHECK(hipGetLastError());
            continue;
        }
        if (codec == 32 && !bag1 && launch_bag_sum_long(a, vec_ok, st)) { EVS_HIP_CHECK(hipGetLastError()); continue; }
        if (codec == 32 && !bag1 && launch_bag_sum_flat(a, vec_ok, st)) { EVS_HIP_CHECK(hipGetLastError()); continue; }
        switch (codec) {
        case 32: launch_codec<32>(a, vec_ok, st, bag1); break;
        case 16: launch_codec<16>(a, vec_ok, st, bag1); break;
        case 8: launch_codec<8>(a, vec_ok, st, bag1); break;
        default: launch_codec<4>(a, vec_ok, st, bag1); break;
        }
        EVS_HIP_CHECK(hipGetLastError());
    }
    return EVS_OK;
}

extern "C" int evs_embedding_bag_sum_sharded(int T, int64_t B, int d, int codec, const void *const *tables,
                                             const int64_t *n_rows, const int64_t *row_lo, const int64_t *row_total,
                                             const int64_t *const *indices, const int64_t *const *offsets, const int64_t *nnz,
                                             const float *const *row_weights, float *out, int64_t out_table_stride,
                                             int64_t out_bag_stride, int64_t out_peer_stride, int64_t bags_per_peer,
                                             void *stream) {
    return bag_sum_impl(T, B, d, codec, tables, n_rows, row_lo, row_total, indices, offsets, nnz, row_weights, out, out_table_stride,
                        out_bag_stride, out_peer_stride, bags_per_peer, nullptr, stream);
}

// p2p exchange: the sharded pooling launch writing every peer's block straight into that peer's receive buffer.  peer_delta:
// DEVICE array of ceil(B / bags_per_peer) entries (floats, multiples of 4): added to the local-layout address of peer q's block
// -- i.e. (peer q's mapped receive address of this rank's block) - (out + q * out_peer_stride).
extern "C" int evs_embedding_bag_sum_p2p(int T, int64_t B, int d, int codec, const void *const *tables,
                                         const int64_t *n_rows, const int64_t *row_lo, const int64_t *row_total,
                                         const int64_t *const *indices, const int64_t *const *offsets, const int64_t *nnz,
                                         const float *const *row_weights, float *out, int64_t out_table_stride,
                                         int64_t out_bag_stride, int64_t out_peer_stride, int64_t bags_per_peer,
                                         const int64_t *peer_delta, void *stream) {
    using namespace evs;
    EVS_REQUIRE(peer_delta && bags_per_peer > 0, "evs_embedding_bag_sum_p2p: needs a peer table and bags_per_peer");
    return bag_sum_impl(T, B, d, codec, tables, n_rows, row_lo, row_total, indices, offsets, nnz, row_weights, out, out_table_stride,
                        out_bag_stride, out_peer_stride, bags_per_peer, peer_delta, stream);
}

extern "C" int evs_embedding_bag_sum(int T, int64_t B, int d, int codec, const void *const *tables,
                                     const int64_t *n_rows, const int64_t *const *indices,
                                     const int64_t *const *offsets, const int64_t *nnz,
                                     const float *const *row_weights, float *out,
                                     int64_t out_table_stride, int64_t out_bag_stride, void *stream) {
    return evs_embedding_bag_sum_sharded(T, B, d, codec, tables, n_rows, nullptr, nullptr, indices, offsets, nnz, row_weights, out,
                                         out_table_stride, out_bag_stride, 0, 0, stream);
}

// ---- row-split tables, receiver side: which partial holds the row ---------------------------------------------------------
namespace evs {
struct RouteArgs {
    const int64_t *idx[EVS_MAX_TABLES_PER_LAUNCH];
    int64_t *dst[EVS_MAX_TABLES_PER_LAUNCH];
    int64_t n_rows[EVS_MAX_TABLES_PER_LAUNCH];
    int64_t row_off[64];
    int64_t B;
    int T, world;
};
__global__ void __launch_bounds__(256) rowsplit_route_kernel(const RouteArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.B * a.T) return;
    const int t = (int)(i / a.B);
    const int64_t b = i - (int64_t)t * a.B;
    const int64_t v = a.idx[t][b], n = a.n_rows[t];
    // rank r holds rows [r*n/W, (r+1)*n/W): the largest r with r*n/W <= v  (sharded.row_owner)
    int64_t r = n > 0 ? ((v + 1) * a.world + (n - 1)) / n - 1 : 0;
    r = r < 0 ? 0 : (r >= a.world ? a.world - 1 : r);   // (an index outside the table: the pool launch flagged it)
    a.dst[t][b] = a.row_off[r] + b;
}
}  // namespace evs

extern "C" int evs_rowsplit_route(int T, int64_t B, int world, const int64_t *const *indices, const int64_t *n_rows,
                                  const int64_t *row_off, int64_t *const *dst, void *stream) {
    using namespace evs;
    EVS_REQUIRE(T >= 0 && T <= EVS_MAX_TABLES_PER_LAUNCH && B >= 0 && world >= 1 && world <= 64,
                "evs_rowsplit_route: T=%d (<= %d) B=%lld world=%d (<= 64)", T, EVS_MAX_TABLES_PER_LAUNCH, (long long)B, world);
    if (T == 0 || B == 0) return EVS_OK;
    EVS_REQUIRE(indices && n_rows && row_off && dst, "evs_rowsplit_route: NULL argument");
    RouteArgs a;
    for (int k = 0; k < EVS_MAX_TABLES_PER_LAUNCH; k++) {
        a.idx[k] = k < T ? indices[k] : nullptr;
        a.dst[k] = k < T ? dst[k] : nullptr;
        a.n_rows[k] = k < T ? n_rows[k] : 0;
        EVS_REQUIRE(k >= T || (a.idx[k] && a.dst[k]), "evs_rowsplit_route: table %d has a NULL pointer", k);
    }
    for (int r = 0; r < 64; r++) a.row_off[r] = r < world ? row_off[r] : 0;
    a.B = B; a.T = T; a.world = world;
    const int64_t n = B * T;
    hipLaunchKernelGGL(rowsplit_route_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    EVS_HIP_CHECK(hipGetLastError());
    return EVS_OK;
}

extern "C" int evs_embedding_bag_sum_stacked(int T, int64_t B, int d, int codec, const void *const *tables,
                                             const int64_t *n_rows, const int64_t *indices_base,
                                             int64_t indices_row_stride, int64_t nnz_per_table,
                                             const int64_t *offsets_base, int64_t offsets_row_stride,
                                             const float *const *row_weights, float *out,
                                             int64_t out_table_stride, int64_t out_bag_stride, void *stream) {
    using namespace evs;
    EVS_REQUIRE(T >= 0 && T <= 4096, "evs_embedding_bag_sum_stacked: bad T=%d", T);
    if (T == 0 || B == 0) return EVS_OK;
    EVS_REQUIRE(indices_base, "evs_embedding_bag_sum_stacked: NULL argument");
    // offsets_base == NULL: one index per bag for every table (see evs_embedding_bag_sum)
    const int64_t *idx[4096];
    const int64_t *off[4096];
    int64_t nnz[4096];
    for (int k = 0; k < T; k++) {
        idx[k] = indices_base + (int64_t)k * indices_row_stride;
        off[k] = offsets_base ? offsets_base + (int64_t)k * offsets_row_stride : nullptr;
        nnz[k] = nnz_per_table;
    }
    return evs_embedding_bag_sum(T, B, d, codec, tables, n_rows, idx, offsets_base ? off : nullptr, nnz, row_weights, out,
                                 out_table_stride, out_bag_stride, stream);
}

extern "C" int evs_check_index_errors(void *stream) {
    using namespace evs;
    int *err = index_error_flag();
    if (!err) return EVS_EHIP;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int h = 0;
    EVS_HIP_CHECK(hipMemcpyAsync(&h, err, sizeof(int), hipMemcpyDeviceToHost, st));
    EVS_HIP_CHECK(hipStreamSynchronize(st));
    if (h) {
        EVS_HIP_CHECK(hipMemsetAsync(err, 0, sizeof(int), st));
        if (h & 2) { set_error("a p2p exchange wait ran out of patience (a peer did not signal): evs_p2p_sync"); return EVS_ESTATE; }
        set_error("embedding index out of range (or malformed offsets) in a previous launch");
        return EVS_EINDEX;
    }
    return EVS_OK;
}
