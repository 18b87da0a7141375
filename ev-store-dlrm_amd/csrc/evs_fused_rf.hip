// Fused embedding gather + pairwise-dot interaction, "rows in flight in registers" form of the bag-1 index-tile loop.
//
// Same computation and the same bits as emb_interact_dot_lds_kernel<..., BAG1, TILE> (evs_fused.hip):
//     R[b] = [ x[b] | strict-lower(T[b] T[b]^T) ],  T[b] = [x[b]; W_0[idx_0[b]]; ...; W_{F-2}[idx_{F-2}[b]]]
// (dlrm_s_pytorch.py:407-461 apply_emb with one index per bag -- the Criteo collate, dlrm_data_pytorch.py:407-408 --
//  followed by :483-516 interact_features), one wavefront per sample, MFMA 16x16x4 f32 chains, output row staged
// in LDS and written one iteration later.  What changes is WHERE a sample's rows wait while they travel:
//
//   LDS-DMA loop:  the rows of sample k+1 are DMA'd into the wave's single 4 KiB LDS slot while sample k computes: one
//                  sample in flight per wave, and a 16 384-sample launch (4 samples per wave) is a chain of 4 memory
//                  round trips per wave plus fill and drain.
//   this kernel:   a block owns ONE 16-sample chunk; the rows of all 4 samples of a wave are requested at once and
//                  travel in REGISTERS, in the DMA-shaped mapping (LPRD = d/4 consecutive lanes fetch one row as d*4
//                  contiguous bytes with one global_load_dwordx4 each; 64/LPRD rows per instruction, every line
//                  requested once): 16 VGPRs hold one d=36 sample.  The LDS slot is only the transpose buffer between
//                  that mapping and the MFMA operand mapping (ds_write_b128 x 4, ds_read_b128 x 6 per sample).  The
//                  code is straight-line, so hipcc's own counted s_waitcnt vmcnt(15..12) consumes sample u while
//                  samples u+1.. stay in flight (with any branch around a memory operation the waitcnt pass takes the
//                  minimum over paths and drains everything: measured, see docs/HISTORY.md 3.2b).
// Used for batches whose blocks are all resident at once (B <= 16 x 4 x 256 = 16 384); larger batches keep the LDS-DMA
// loop, whose steady state is bound by the per-CU vector-memory / LDS pipes rather than by latency (docs/HISTORY.md 3.2).
// Measured (same box, A/B by EVS_FUSED_RF): B = 16 384: 20.5 -> 18.9 us, B = 8 192: 12.5 -> 11.4 us.
//
// Row addresses need no cross-lane traffic here: the index tile in LDS has one row PER FEATURE (x and dense features
// carry the sample number as their "index"), and a lane of the DMA mapping reads the tile entry of the row it fetches.
#include <fcntl.h>
#include <unistd.h>
#include "evs_fused.h"

#include <stdlib.h>
#include <string.h>

namespace evs {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) f32x4 *gf4_t;

#ifndef EVS_RF_DEPTH
#define EVS_RF_DEPTH 4
#endif
#ifndef EVS_RF_LB
#define EVS_RF_LB 4
#endif
#ifndef EVS_OUT_CPOL
#define EVS_OUT_CPOL 2   // nt: R is written once and streams out (see evs_fused.hip)
#endif

// MLP: the first top-MLP layer (dlrm_s_pytorch.py:601-605: p = apply_mlp(z, top_l); its first nn.Linear + ReLU) fused
// behind the interaction.  The 16 output rows of the block's chunk stay in LDS (s_R, zero-padded to kp columns) and are
// the A operand (M = 16 samples) of Z1 = act(R W1^T + b1): 16 x 16 output tiles, one v_mfma_f32_16x16x4_f32 chain of
// kp / 4 steps each, the four waves taking the n1 / 16 tiles in turn; W1 arrives zero-padded and row-aligned (w1p) and
// is read straight from L2 (the whole 0.8 MB matrix is re-read per 16 samples: 0.8 GB of L2 traffic at B = 16 384, and
// 100 MFMAs per tile -- with the layer the kernel is matrix-core-bound, not HBM-bound).  k-slot q of the MFMA owns the
// contiguous columns [q * kp/4, (q+1) * kp/4) of both operands, so they are read as 16-byte pieces.  R itself is
// written only when asked for.
constexpr int kMlpRowStride = 452;   // floats per staged row: kp <= 448 (F <= 28, d <= 36, diagonal kept) + 4 (bank spread)

// IDS: the cache tier's consumer -- features 1..F-1 come as one (B, F-1) int32 table of row ids (FusedArgs::row_ids): bit 30
// says "row of the cache arena", else the row of the feature's own table; 4 bytes per key instead of an 8-byte address,
// and the rows travel exactly as in the plain launch.
// PROBE (implies IDS): the kernel reads the REQUEST rows and probes the cache's hash itself in its head -- what
// cache_batch_probe_gather_kernel does as a launch of its own (hash probe, agg_hit per request, priority bump, hit flags,
// the block's miss list for the update kernel, hit statistics) -- and goes on with the ids it found.
// CHECK (lS_o given, whole batches: nnz == B, B or B + 1 offsets -- what the reference's loop passes): the block also loads
// the offsets of its 16 samples and checks that every bag is exactly {idx[b]} (offsets[b] == b and the bag ends at b + 1:
// the neighbour lane's offset, one extra load behind the block's last sample).  A block that finds anything else pools ITS
// samples in a slow loop straight from global memory (general semantics: empty bags, several indices summed in index
// order, bad offsets / indices skipped and flagged -- the arithmetic of evs_fused.hip's general loop) and feeds the same
// MFMA + output code: no flag, no second launch, as in the index-tile loop of evs_fused.hip and in evs_fused_rfq.hip.
// SERVE (round 6; evs_emb_interact_serve_*): the same body run by a RESIDENT grid, once per (descriptor, chunk) -- what varies
// from batch to batch (x, the (T, B) index / offsets arrays, R, B) comes from the descriptor `sd`, everything else (tables,
// shapes) from a FusedArgs the server keeps in device memory (`ka` points at it, `args` is a copy of its scalar fields); R is
// stored with agent-scope write-through stores (sc1): a resident kernel has no end-of-kernel release that would write its L2
// back for the launches (on other XCDs) that read R afterwards.
// what a RESIDENT grid reads of a batch's inputs (indices, offsets, x) goes through agent-scope loads: the grid never passes a
// kernel boundary, so nothing invalidates its L1 / L2 between two batches that reuse the same addresses (an acquire fence per
// descriptor does -- buffer_inv sc1 by a thousand blocks: measured, +27 us per batch)
template <bool COHERENT>
__device__ __forceinline__ int64_t rf_ld_i64(const int64_t *p) {
    typedef const __attribute__((address_space(1))) int64_t *g_t;
    if constexpr (COHERENT) return __hip_atomic_load(reinterpret_cast<g_t>(reinterpret_cast<uintptr_t>(p)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *reinterpret_cast<g_t>(reinterpret_cast<uintptr_t>(p));
}
template <bool COHERENT>
__device__ __forceinline__ float rf_ld_f32(const float *p) {
    typedef const __attribute__((address_space(1))) float *g_t;
    if constexpr (COHERENT) return __hip_atomic_load(reinterpret_cast<g_t>(reinterpret_cast<uintptr_t>(p)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *reinterpret_cast<g_t>(reinterpret_cast<uintptr_t>(p));
}
struct RfServeDesc {
    const float *x; const int64_t *idx; const int64_t *off; float *R;
    int64_t B, x_stride, idx_stride, off_stride;
};
#ifdef EVS_X_PT   // developer instrumentation (tools/probe_stage_probe.py): per block, 100 MHz ticks from the block's entry to stage k, summed over launches
__device__ unsigned long long g_pt[1024][16];     // (a row per block: launches do not overlap, a block adds to its own words)
#define EVS_PT(k) do { if (threadIdx.x == 0) g_pt[blockIdx.x & 1023][k] += (unsigned long long)((long long)wall_clock64() - pt_t0); } while (0)
#define EVS_PTW(k) do { __builtin_amdgcn_s_waitcnt(0x0F70); EVS_PT(k); } while (0)     // behind everything this wave has asked for
#else
#define EVS_PT(k) do { } while (0)
#define EVS_PTW(k) do { } while (0)
#endif
template <int CQ, int REM, int NT, int D, bool MLP, bool IDS, bool PROBE, bool CHECK, bool SERVE>
__device__ __forceinline__ void rf_body(const FusedArgs &args, const FusedArgs *ka, const RfServeDesc &sd, const int blk_in, const bool srv_first = false) {
    static_assert(!CHECK || (!MLP && !IDS && !PROBE), "the offsets check belongs to the plain launch");
    static_assert(!SERVE || (CHECK && !MLP && !IDS && !PROBE), "the resident form serves the drop-in call (lS_o given)");
#ifdef EVS_X_PT
    const long long pt_t0 = (long long)wall_clock64();
    if (threadIdx.x == 0) g_pt[blockIdx.x & 1023][15] += 1ull;
#endif
#ifndef EVS_X_SRV
#define EVS_X_SRV 0     // developer A/B of the resident form (timing only): 2 = R stores without sc1, 4 = workers sleep longer between polls
#endif
    constexpr int kCpol = (SERVE && !(EVS_X_SRV & 2)) ? (EVS_OUT_CPOL | 16) : EVS_OUT_CPOL;   // (aux bit 4 = sc1 on gfx940+: agent scope, write-through)
    constexpr int NR = NT;
    constexpr int NC = CQ + REM;
    constexpr int d = 4 * (4 * CQ + REM);
    constexpr int LPRD = d / 4;             // lanes per fp32 row (16 B each)
    constexpr int RPI = 64 / LPRD;          // rows per load instruction
    constexpr int NROWS = 16 * NT;
    constexpr int MAXF = NT == 2 ? kTileMaxF : 16;
    constexpr int NJ = (MAXF + RPI - 1) / RPI;   // load instructions per sample
    constexpr int row_bytes = d * 4;
    static_assert(NJ * RPI <= 32, "a tile row per fetched row");
    __shared__ __attribute__((aligned(16))) char s_rows[4][NJ * 1024];   // per wave: transpose buffer (DMA image of one sample)
    __shared__ int s_idx[2 * 512];                                        // [2][32 features][16 samples]: row id, sample id (dense), -1 = zeros
    __shared__ const int64_t *s_tile_p[32];
    __shared__ unsigned s_tile_nr[32];
    __shared__ int s_tile_kind[32];                                       // 0 absent, 1 dense (x, received pooled vectors), 2 table
    __shared__ unsigned long long s_feat_base[32];                        // per feature: first row / bytes between rows -- read per
    __shared__ unsigned s_feat_scale[32];                                 // load instead of living in 12 VGPRs per lane
    __shared__ unsigned s_sa_base[PROBE ? 32 : 1];                        // PROBE, set-associative cache: dense row number of row 0 of feature f's table
    // PROBE, the update folded in as well (ProbeArgs::arena_w): totals of the inserts this block makes
    __shared__ int s_udelta[PROBE ? kMaxBuckets : 1];
    __shared__ int s_ustat[PROBE ? 2 : 1];
    __shared__ unsigned long long s_srv_src[SERVE ? 32 : 1];              // SERVE: the tables' addresses and row counts, read from the template ONCE per
    __shared__ unsigned s_srv_nr[SERVE ? 32 : 1];                         // block (srv_first) -- the launch form pays that round trip per chunk
    __shared__ const int64_t *s_tile_o[CHECK ? 32 : 1];                   // CHECK: offsets arrays, their readable entries, nnz
    __shared__ int64_t s_tile_ol[CHECK ? 32 : 1], s_tile_nz[CHECK ? 32 : 1];
    constexpr int OUT_MAX = ((d + NROWS * (NROWS + 1) / 2 + 63) / 64) * 64;
    // staged output rows: one slot per wave (flushed an iteration later), or with MLP all 16 rows of the chunk
    constexpr int kOutRows = MLP ? 16 : 4;
    constexpr int kOutStride = MLP ? kMlpRowStride : OUT_MAX + 16;
    __shared__ __attribute__((aligned(16))) float s_out[kOutRows][kOutStride];
    __shared__ float s_dump[MLP ? 4 * 16 : 1];   // MLP: where the never-stored accumulator elements go

    // (SERVE: the resident grid runs this body in a loop; every lane-invariant table below -- staging offsets, flush offsets,
    //  operand addresses -- derives from the thread index, and left alone the compiler hoists all of them out of the loop and
    //  keeps them alive across the head: 128 VGPRs and 65 spilled.  An opaque copy of the index per call keeps them where the
    //  launch form has them.)
    unsigned tid_x = threadIdx.x;
    if constexpr (SERVE) asm volatile("" : "+v"(tid_x));
    const int lane = tid_x & (kWave - 1);
    const int r16 = lane & 15;
    const int q = lane >> 4;
    const int F = args.F, itself = args.itself;
    const int out_row = d + args.P;
    const int64_t B = SERVE ? sd.B : args.B;
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(tid_x >> 6));
    char *my_lds = s_rows[wave_in_block];
    float *my_out = s_out[wave_in_block];   // (MLP: re-pointed per sample)
    const char *zeros_l = reinterpret_cast<const char *>(args.zeros);

    // ---- the block's sample range and the feature table of the tile ------------------------------------
    // K batches in one launch (multi_n > 0; plain and CHECK launches): block i = chunk i % multi_cpb of batch i / multi_cpb;
    // sample numbers below are the batch's own, x / indices / offsets / R come from the batch's entries
    const int64_t per = args.tile_per;
    int blk_id = blk_in, batch_k = 0;
    if constexpr (!MLP && !IDS && !PROBE) {
        if (args.multi_n > 0) { batch_k = blk_id / args.multi_cpb; blk_id -= batch_k * args.multi_cpb; }
    }
    const bool multi = SERVE || (!MLP && !IDS && !PROBE && args.multi_n > 0);
    float *const R_base = SERVE ? sd.R : (multi ? ka->multi_R[batch_k] : args.R);
    const int64_t blk_first = (int64_t)blk_id * per;
    const int64_t blk_end = blk_first + per < B ? blk_first + per : B;
    if (blk_first >= blk_end) return;       // block-uniform
    const int blk_n = (int)(blk_end - blk_first);
    const int n_samples = blk_n > wave_in_block ? (blk_n - wave_in_block + 3) / 4 : 0;
    // Round 6: the stacked form of the call (FusedArgs::stk, and every batch of the resident dispatcher) -- its index / offsets
    // loads need nothing of the feature table below, so they go out FIRST and the table's round trip to the kernel arguments
    // runs under them: one dependent round trip less in the head of every block.
    const bool stk = SERVE || (!MLP && !IDS && !PROBE && args.stk != 0 && args.multi_n == 0);
    const int64_t *const stk_idx = SERVE ? sd.idx : args.stk_idx, *const stk_off = SERVE ? sd.off : args.stk_off;
    const int64_t stk_istride = SERVE ? sd.idx_stride : args.stk_idx_stride, stk_ostride = SERVE ? sd.off_stride : args.stk_off_stride;
    const int64_t stk_ol = SERVE ? sd.B : args.stk_off_len;
    // ---- index tiles: thread e (and e + 256) owns tile element (feature e >> 4, sample-in-chunk e & 15) ----
    bool bad = false, my_ragged = false;
    bool oob[2] = {false, false};
    int64_t tile_v[2] = {-1, -1};
    int64_t tile_o0[2] = {0, 0}, tile_o1[2] = {0, 0};   // CHECK: offsets[b] and where bag b ends
    const int64_t *dummy_i = args.dummy_i64;   // any readable int64 (lanes with nothing to load read it)
    auto tile_load = [&](int c) {       // chunk c of the block -> registers; no branch, no use of the value before tile_store
        const int64_t bs = blk_first + 16 * (int64_t)c + (tid_x & 15);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int f = ((int)tid_x >> 4) + 16 * h;
            // (stk: x + tables behind ONE (T, B) index array -- the Criteo collate's -- whose rows are at a fixed stride: the address is
            //  arithmetic, nothing of the block's feature table is needed, and the loads leave BEFORE that table is read)
            const bool table = (stk ? (f >= 1 && f < F) : s_tile_kind[f] == 2) && bs < blk_end;
            if constexpr (IDS) {
                const int *ap = table ? args.row_ids + bs * (int64_t)(F - 1) + (f - 1) : reinterpret_cast<const int *>(dummy_i);
                tile_v[h] = *reinterpret_cast<const __attribute__((address_space(1))) int *>(reinterpret_cast<uintptr_t>(ap));
            } else {
                const int64_t *ap = table ? (stk ? stk_idx + (int64_t)(f - 1) * stk_istride : s_tile_p[f]) + bs : dummy_i;
                // (explicitly global: a flat load would force every later wait to vmcnt(0))
                tile_v[h] = rf_ld_i64<SERVE>(ap);
            }
            if constexpr (CHECK) {
                const int64_t *op = stk ? stk_off + (int64_t)(f - 1) * stk_ostride : s_tile_o[f];
                const bool own = table && ((tid_x & 15) == 15 || bs + 1 >= blk_end);
                const int64_t *p0 = table ? op + bs : dummy_i;
                const int64_t *p1 = (own && bs + 1 < (stk ? stk_ol : s_tile_ol[f])) ? op + bs + 1 : dummy_i;
                tile_o0[h] = rf_ld_i64<SERVE>(p0);
                tile_o1[h] = rf_ld_i64<SERVE>(p1);
            }
        }
    };
    if constexpr (!PROBE) { if (stk) tile_load(0); }
    if (tid_x < 32) {
        // branch-free on purpose: every FusedArgs array has EVS_MAX_FEATURES = 32 entries (those past F are NULL / 0), so
        // lane f reads entry f of each of them UNCONDITIONALLY -- all the loads leave together, one round trip -- and the
        // tests below are selects on the values.  (Written as `f < F ? ka->x[f] : 0` the compiler puts each load behind
        // its own branch and wait: three to six DEPENDENT round trips to the kernel arguments in front of the first index
        // load, in every block of every launch -- the larger part of the launch's fixed cost.)
        const int f = (int)tid_x;
        const bool on = f < F;
        const int64_t *ip = nullptr;
        unsigned long long src = 0ull;
        int64_t nr = 0, stride = 0;
        const int64_t *op = nullptr;
        int64_t ol = 0, nz = 0;
        if constexpr (SERVE) {
            if (srv_first) {   // (block-uniform) the one round trip to the template this block ever makes
                s_srv_src[f] = (unsigned long long)reinterpret_cast<uintptr_t>(ka->src[f]);
                s_srv_nr[f] = (unsigned)ka->n_rows[f];
            }
            src = s_srv_src[f]; nr = (int64_t)s_srv_nr[f];
            ol = sd.B; nz = sd.B;
        } else {
            ip = ka->indices[f];
            src = (unsigned long long)reinterpret_cast<uintptr_t>(ka->src[f]);
            nr = ka->n_rows[f];
            stride = ka->stride[f];
            if constexpr (CHECK) { op = ka->offsets[f]; ol = ka->off_len[f]; nz = ka->nnz[f]; }
        }
        unsigned long long mx = 0ull;
        const int64_t *mi = nullptr, *mo = nullptr;
        if constexpr (SERVE) {
            mx = (unsigned long long)reinterpret_cast<uintptr_t>(sd.x); mi = sd.idx; mo = sd.off;
        } else if constexpr (!MLP && !IDS && !PROBE) {
            const int kk = multi ? batch_k : 0;   // (entry 0 is always readable)
            mx = (unsigned long long)reinterpret_cast<uintptr_t>(ka->multi_x[kk]);
            mi = ka->multi_idx[kk];
            if constexpr (CHECK) mo = ka->multi_off[kk];
        }
        const int64_t m_istride = SERVE ? sd.idx_stride : args.multi_idx_stride, m_ostride = SERVE ? sd.off_stride : args.multi_off_stride;
        if (multi) ip = (f >= 1 && on) ? mi + (int64_t)(f - 1) * m_istride : nullptr;
        if (!on) ip = nullptr;
        const bool table = (IDS || PROBE) ? (f >= 1 && on) : ip != nullptr;
        s_tile_p[f] = ip;
        s_tile_nr[f] = on ? (unsigned)nr : 0u;
        s_tile_kind[f] = !on ? 0 : (table ? 2 : 1);
        s_feat_base[f] = !on ? 0ull : ((multi && f == 0) ? mx : src);
        s_feat_scale[f] = !on ? 0u : (table ? (unsigned)row_bytes : (unsigned)(((SERVE && f == 0) ? sd.x_stride : stride) * 4));
        if constexpr (PROBE) s_sa_base[f] = ka->probe.sau.row_base[(f + 31) & 31];   // (feature f = table f - 1; unconditional read, as above)
        if constexpr (CHECK) {
            s_tile_o[f] = table ? (multi ? mo + (int64_t)(f - 1) * m_ostride : op) : nullptr;
            s_tile_ol[f] = table ? ol : 0;
            s_tile_nz[f] = table ? nz : 0;
        }
    }
    __syncthreads();

    // ---- DMA-shaped mapping: for load j this lane fetches piece dma_piece of row j*RPI + dma_r0.  Lanes past the last
    // whole row of an instruction (lane 63 at d = 36) mirror the last piece: the same 16 bytes, one request, and their
    // copy lands in the padding of the 1 KiB image block.  Rows >= F have tile entries -1 and read the zero page.
    const int lane_eff = lane < RPI * LPRD ? lane : RPI * LPRD - 1;
    const int dma_piece16 = (lane_eff % LPRD) * 16;
    const int dma_r0 = lane_eff / LPRD;
    // ---- MFMA operand mapping (as the LDS-DMA loop: row r at (r / RPI) KiB + (r % RPI) * row_bytes) -------
    int lds_off[NR];
#pragma unroll
    for (int rr = 0; rr < NR; rr++) {
        const int row = r16 + 16 * rr;
        lds_off[rr] = (row / RPI) * 1024 + (row % RPI) * row_bytes + q * CQ * 16;
    }
    constexpr int kRemOff = 4 * CQ * 16;
    int rem_off[NR];   // element q of the first trailing chunk of this lane's tile rows
#pragma unroll
    for (int rr = 0; rr < NR; rr++) rem_off[rr] = lds_off[rr] - q * CQ * 16 + kRemOff + q * 4;
    constexpr int kOob = 0x7ffffff0;
    // (`on` false: a zero-length buffer resource, the hardware drops every store -- the loop body has no branch around
    //  its vector-memory operations, see the main loop)
#ifndef EVS_RF_OLDFLUSH
    // lane-invariant pieces of the flush, computed once: which 16-byte piece of the staged row this lane moves in store h
    // (an rf launch has F <= kTileMaxF: at most d + 28 * 29 / 2 = 442 floats, two store instructions of 64 x 16 bytes
    // cover them -- the generic 32-row bound would issue a third that the bounds check always drops), its offset in R's
    // row (kOob: dropped by the buffer bounds check), and the same for the 0..3 trailing floats
    constexpr int kFlushMaxRow = MLP ? OUT_MAX : ((d + MAXF * (MAXF + 1) / 2 + 3) / 4) * 4;
    constexpr int NFL = (kFlushMaxRow + 255) / 256;
    int fl_lds[NFL], fl_off[NFL], fl_tail_lds = 0, fl_tail_off = 0;   // filled by fill_hoists(), behind the row requests
    auto fill_flush = [&]() {
        const int n4 = out_row >> 2;
#pragma unroll
        for (int h = 0; h < NFL; h++) {
            const int e4 = lane + 64 * h;
            fl_lds[h] = 16 * (e4 < n4 ? e4 : 0);
            fl_off[h] = e4 < n4 ? 16 * e4 : kOob;
            asm volatile("" : "+v"(fl_lds[h]), "+v"(fl_off[h]));   // (keep them in registers: the compiler otherwise re-derives them per sample)
        }
        fl_tail_lds = 4 * (4 * (out_row >> 2) + (lane & 3));
        fl_tail_off = lane < (out_row & 3) ? fl_tail_lds : kOob;
        asm volatile("" : "+v"(fl_tail_lds), "+v"(fl_tail_off));
    };
    auto flush_out = [&](int64_t bp, bool on) {
        if constexpr (MLP) on = on && args.write_r;
        float *Rb = R_base + (on ? bp : 0) * (int64_t)out_row;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(Rb, 0, on ? out_row * 4 : 0, 0x00020000);
#pragma unroll
        for (int h = 0; h < NFL; h++) {
            const float4 v = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(my_out) + fl_lds[h]);
            u32x4 u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
            __builtin_amdgcn_raw_buffer_store_b128(u, rs, fl_off[h], 0, kCpol);
        }
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(*reinterpret_cast<const float *>(reinterpret_cast<const char *>(my_out) + fl_tail_lds)),
                                              rs, fl_tail_off, 0, kCpol);
    };
#else
    auto flush_out = [&](int64_t bp, bool on) {
        if constexpr (MLP) on = on && args.write_r;
        float *Rb = R_base + (on ? bp : 0) * (int64_t)out_row;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(Rb, 0, on ? out_row * 4 : 0, 0x00020000);
        const int n4 = out_row >> 2;   // whole 16-byte pieces; the 0..3 trailing floats go as dwords
        // (always the same number of store instructions: out-of-range lanes and whole out-of-range instructions are
        //  dropped by the buffer bounds check, and the waitcnt bookkeeping stays static)
#pragma unroll
        for (int h = 0; h < (OUT_MAX + 255) / 256; h++) {
            const int e4 = lane + 64 * h;
            const float4 v = reinterpret_cast<const float4 *>(my_out)[e4 < n4 ? e4 : 0];
            u32x4 u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
            __builtin_amdgcn_raw_buffer_store_b128(u, rs, e4 < n4 ? 16 * e4 : kOob, 0, kCpol);
        }
        {
            const int e = 4 * n4 + (lane & 3);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(my_out[e]), rs, lane < (out_row & 3) ? 4 * e : kOob, 0, kCpol);
        }
    };

#endif

    auto tile_store = [&](int c) {      // registers -> tile buffer c & 1
        const int64_t bs = blk_first + 16 * (int64_t)c + (tid_x & 15);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int f = ((int)tid_x >> 4) + 16 * h;
            const int kind = s_tile_kind[f];
            const bool live = kind != 0 && bs < blk_end && c >= 0;
            const int64_t v = kind == 2 ? tile_v[h] : bs;       // dense features (x, received pooled vectors): the sample number
            const bool in_range = kind == 1 || (IDS ? v >= 0 : (uint64_t)v < (uint64_t)s_tile_nr[f]);   // (IDS: the probe kernel checked the row ids)
            if constexpr (CHECK) oob[h] = live & !in_range;   // (whether it counts is verify()'s call: see there)
            else bad |= live & !in_range;
            s_idx[(c & 1) * 512 + (int)tid_x + 256 * h] = (live & in_range) ? (int)v : -1;
        }
    };

    // CHECK: the verdict on the offsets -- is every bag of the chunk exactly {idx[b]}?  (Taken in front of the row requests.
    // Behind them -- rows asked for on the bet that it is, compares and the block barrier under their round trip, the slow
    // loop at the end of the kernel overwriting what the one-index code produced -- was built and measured: 22.3 instead of
    // 20.7 us at B = 16 384; the offsets pairs then live across the 64 row registers, 128 VGPRs and spills.)
    // A bag ends where the next one starts, and the next one's start is its own lane's o0 (same feature, next sample): every lane
    // checks that ITS bag starts at its own position, the lane of the chunk's last sample also where that bag ends -- no
    // exchange between lanes (round 4: the shuffle and the second 64-bit compare per key were 10 VALU instructions per
    // sample in front of the row requests).
    bool fast_bad = false;   // out-of-range indices seen by the one-index code; they count only if the block stays on it
    auto verify = [&]() {
        const int64_t bs = blk_first + (tid_x & 15);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int f = ((int)tid_x >> 4) + 16 * h;
            // (an index whose own bag is not {idx[b]} may sit at a position no bag refers to: the slow loop, which this
            //  block then runs, has the verdict on it)
            const bool table = s_tile_kind[f] == 2 && bs < blk_end;
            const bool own = table && ((tid_x & 15) == 15 || bs + 1 >= blk_end);
            int64_t o1 = tile_o1[h];
            if (!(bs + 1 < s_tile_ol[f])) o1 = s_tile_nz[f];   // the last bag ends at nnz
            const bool ok = (tile_o0[h] == bs) & (!own | (o1 == bs + 1));
            my_ragged |= table & !ok;
            fast_bad |= oob[h];
        }
    };

    // ---- the rows of this wave's sample n -> registers (D samples in flight) ----------------------------
    f32x4 ring[D][NJ];
    // per load j this lane always fetches a row of the SAME feature (tile row dma_r0 + j * RPI): where that feature's rows
    // start (plus this lane's 16-byte piece) and how far apart they are, read from the block's feature table once
    unsigned long long fbase[NJ];
    unsigned fscale[NJ];
    int idx_lds[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int r = dma_r0 + j * RPI;               // < 32: tile rows >= F hold -1
        fbase[j] = s_feat_base[r] + (unsigned long long)dma_piece16;
        fscale[j] = s_feat_scale[r];
        idx_lds[j] = 4 * (r * 16 + wave_in_block);    // byte address of tile entry (row r, sample wave_in_block) in s_idx
    }
    const unsigned long long zeros_piece = (unsigned long long)reinterpret_cast<uintptr_t>(zeros_l) + (unsigned long long)dma_piece16;
    auto issue = [&](int n, f32x4 (&slot)[NJ]) {
        // block-local sample m = wave_in_block + 4 n (< 16: one chunk per block): tile buffer 0, entry m of each row
        const unsigned phantom = (unsigned)n < (unsigned)n_samples ? 0u : 0xffffffffu;   // past this wave's samples: every lane reads the zero page
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int iv = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(s_idx) + idx_lds[j] + 16 * n);
            // branch-free on purpose (bit blend, not a select: the compiler turns a select over these LDS reads into
            // control flow and serialises the four loads): -1 -> the zero page
            const unsigned neg = (unsigned)(iv >> 31) | phantom;
            unsigned long long base = fbase[j];
            unsigned idx = (unsigned)iv & ~neg;
            if constexpr (IDS || PROBE) {   // bit 30: a row of the cache arena (bit blend, as below: no select over LDS reads)
                const unsigned long long in_arena = 0ull - (unsigned long long)((idx >> 30) & 1u);
                base ^= (base ^ ((unsigned long long)reinterpret_cast<uintptr_t>(args.arena) + (unsigned long long)dma_piece16)) & in_arena;
                idx &= 0x3fffffffu;
            }
            const unsigned long long p = base + (unsigned long long)idx * (unsigned long long)fscale[j];
            const unsigned long long m64 = ((unsigned long long)neg << 32) | neg;
            const unsigned long long pa = p ^ ((p ^ zeros_piece) & m64);
            slot[j] = *reinterpret_cast<gf4_t>((uintptr_t)pa);
        }
    };
    // a: this lane's CQ k-slot chunks of tile rows r16 (+ 16); rem: element q of each of the REM trailing chunks (all four
    // k-slots hold those chunks, slot q contributes element q -- read as ONE float from the image, not selected out of four)
    auto interact = [&](const float4 (&a)[NR][CQ > 0 ? CQ : 1], const float (&rem)[NR][REM > 0 ? REM : 1], f32x4 &c00, f32x4 &c10, f32x4 &c11) {
        c00 = f32x4{0.f, 0.f, 0.f, 0.f}; c10 = f32x4{0.f, 0.f, 0.f, 0.f}; c11 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < CQ; c++) {
            const float e0[4] = {a[0][c].x, a[0][c].y, a[0][c].z, a[0][c].w};
            const float e1[4] = {a[NR - 1][c].x, a[NR - 1][c].y, a[NR - 1][c].z, a[NR - 1][c].w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(e0[e], e0[e], c00, 0, 0, 0);
                if constexpr (NT == 2) {
                    c10 = __builtin_amdgcn_mfma_f32_16x16x4f32(e1[e], e0[e], c10, 0, 0, 0);
                    c11 = __builtin_amdgcn_mfma_f32_16x16x4f32(e1[e], e1[e], c11, 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int m = 0; m < REM; m++) {
            const float s0 = rem[0][m], s1 = rem[NR - 1][m];
            c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(s0, s0, c00, 0, 0, 0);
            if constexpr (NT == 2) {
                c10 = __builtin_amdgcn_mfma_f32_16x16x4f32(s1, s0, c10, 0, 0, 0);
                c11 = __builtin_amdgcn_mfma_f32_16x16x4f32(s1, s1, c11, 0, 0, 0);
            }
        }
    };

    // ---- where the accumulators go in the staged output row: lane-invariant, computed ONCE (relative to my_out; left to
    // itself the compiler re-derives the twelve offsets -- multiplies, compares, exec-mask regions -- for every sample)
    int zo00h[4], zo10h[4], zo11h[4];
    int xv_off[(d + 63) / 64];
    auto fill_stage = [&]() {
        const int dump0 = 4 * (OUT_MAX + r16);
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const int i = 4 * q + v;
            zo00h[v] = (i < F && r16 < i + itself) ? 4 * (d + (i * (i - 1 + 2 * itself)) / 2 + r16) : dump0;
            const int gi = 16 + i;
            const int base = (gi * (gi - 1 + 2 * itself)) / 2;
            zo10h[v] = (NT == 2 && gi < F) ? 4 * (d + base + r16) : dump0;
            zo11h[v] = (NT == 2 && gi < F && 16 + r16 < gi + itself) ? 4 * (d + base + 16 + r16) : dump0;
            asm volatile("" : "+v"(zo00h[v]), "+v"(zo10h[v]), "+v"(zo11h[v]));
        }
#pragma unroll
        for (int h = 0; h < (d + 63) / 64; h++) {
            const int e = lane + 64 * h;
            xv_off[h] = e < d ? 4 * e : 4 * (OUT_MAX + r16);
        }
    };

    auto flush_slow = [&](int64_t bp) {   // the slow block's flush: offsets computed in place (the hoisted ones are not filled yet)
        float *Rb = R_base + bp * (int64_t)out_row;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(Rb, 0, out_row * 4, 0x00020000);
        const int n4 = out_row >> 2;
#pragma unroll
        for (int h = 0; h < NFL; h++) {
            const int e4 = lane + 64 * h;
            const float4 v = reinterpret_cast<const float4 *>(my_out)[e4 < n4 ? e4 : 0];
            u32x4 u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
            __builtin_amdgcn_raw_buffer_store_b128(u, rs, e4 < n4 ? 16 * e4 : kOob, 0, kCpol);
        }
        const int e = 4 * n4 + (lane & 3);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(my_out[e]), rs, lane < (out_row & 3) ? 4 * e : kOob, 0, kCpol);
    };
    // ---- CHECK, rare: a block that finds a bag other than {idx[b]} pools ITS samples with the general semantics --------
    auto slow_block = [&]() {
        for (int u = 0; u < n_samples; u++) {
            const int64_t b = blk_first + wave_in_block + 4 * (int64_t)u;
            float4 a[NR][NC];
#pragma unroll
            for (int rr = 0; rr < NR; rr++) {
                const int f = r16 + 16 * rr;
#pragma unroll
                for (int c = 0; c < NC; c++) a[rr][c] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (f >= F) continue;
                // byte offset of chunk c inside a row: this lane's k-slot chunks, then the shared remainder chunks
                auto chunk_at = [&](const char *row, int c) -> float4 {
                    return *reinterpret_cast<const float4 *>(row + (c < CQ ? (q * CQ + c) * 16 : kRemOff + (c - CQ) * 16));
                };
                const int64_t *ip = s_tile_p[f];                                            // (multi: this batch's arrays)
                const char *src = reinterpret_cast<const char *>((uintptr_t)s_feat_base[f]);
                if (!ip) {   // dense feature (x, received pooled vectors)
                    const char *row = src + (uint64_t)b * (uint64_t)(((SERVE && f == 0) ? sd.x_stride : ka->stride[f]) * 4);
#pragma unroll
                    for (int c = 0; c < NC; c++) {
                        if constexpr (SERVE) {   // (x: fresh from memory, see rf_ld_*)
                            const float *fp = reinterpret_cast<const float *>(row + (c < CQ ? (q * CQ + c) * 16 : kRemOff + (c - CQ) * 16));
                            a[rr][c] = make_float4(rf_ld_f32<true>(fp), rf_ld_f32<true>(fp + 1), rf_ld_f32<true>(fp + 2), rf_ld_f32<true>(fp + 3));
                        } else a[rr][c] = chunk_at(row, c);
                    }
                    continue;
                }
                const int64_t *op = s_tile_o[f];
                const int64_t nnz = SERVE ? sd.B : ka->nnz[f];
                int64_t s0 = rf_ld_i64<SERVE>(op + b);
                int64_t e0 = (b + 1 < (SERVE ? sd.B : ka->off_len[f])) ? rf_ld_i64<SERVE>(op + b + 1) : nnz;
                if (!((s0 >= 0) & (e0 >= s0) & (e0 <= nnz))) { bad = true; s0 = e0 = 0; }
                const uint64_t n_rows = SERVE ? (uint64_t)s_srv_nr[f] : (uint64_t)ka->n_rows[f];
                for (int64_t j = s0; j < e0; j++) {
                    const int64_t r = rf_ld_i64<SERVE>(ip + j);
                    if ((uint64_t)r >= n_rows) { bad = true; continue; }   // skipped; a skipped FIRST row counts as zeros
                    const char *row = src + (uint64_t)r * (uint64_t)row_bytes;
#pragma unroll
                    for (int c = 0; c < NC; c++) {
                        const float4 t = chunk_at(row, c);
                        if (j == s0) { a[rr][c] = t; continue; }
                        a[rr][c].x = __fadd_rn(a[rr][c].x, t.x); a[rr][c].y = __fadd_rn(a[rr][c].y, t.y);
                        a[rr][c].z = __fadd_rn(a[rr][c].z, t.z); a[rr][c].w = __fadd_rn(a[rr][c].w, t.w);
                    }
                }
            }
            float xv[(d + 63) / 64];   // x[b] (feature 0, dense) for the passthrough columns
#pragma unroll
            for (int h = 0; h < (d + 63) / 64; h++) {
                const int e = lane + 64 * h;
                xv[h] = rf_ld_f32<SERVE>(reinterpret_cast<const float *>(reinterpret_cast<const char *>((uintptr_t)s_feat_base[0]) + (uint64_t)b * (uint64_t)((SERVE ? sd.x_stride : ka->stride[0]) * 4)) + (e < d ? e : 0));
            }
            f32x4 c00, c10, c11;
            {
                float4 aq[NR][CQ > 0 ? CQ : 1];
                float ar[NR][REM > 0 ? REM : 1];
#pragma unroll
                for (int rr = 0; rr < NR; rr++) {
#pragma unroll
                    for (int c = 0; c < CQ; c++) aq[rr][c] = a[rr][c];
#pragma unroll
                    for (int m = 0; m < REM; m++) {
                        const float4 t = a[rr][CQ + m];
                        ar[rr][m] = q == 0 ? t.x : q == 1 ? t.y : q == 2 ? t.z : t.w;
                    }
                }
                interact(aq, ar, c00, c10, c11);
            }
            const int dump = 4 * (OUT_MAX + r16);
#pragma unroll
            for (int h = 0; h < (d + 63) / 64; h++) {
                const int e = lane + 64 * h;
                *reinterpret_cast<float *>(reinterpret_cast<char *>(my_out) + (e < d ? 4 * e : dump)) = xv[h];
            }
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
            flush_slow(b);
        }
        if (bad) atomicOr(args.err, 1);
    };

    static_assert(D == 4, "a block owns one 16-sample chunk: 4 samples per wave");
    if constexpr (PROBE) {
        // ---- the cache probe, folded in: thread e (and e + 256) owns key (table (e >> 4) - 1, sample e & 15) -------------
        __shared__ int s_agg[16];                 // hits per request of the chunk
        __shared__ int s_pdelta[kMaxBuckets];     // priority histogram moves
        __shared__ int s_psum[2];                 // hits / perfect requests
        __shared__ int s_nlist;                   // misses listed
        const ProbeArgs &pa = args.probe;
        const int T = pa.T;
        for (int i = tid_x; i < kMaxBuckets; i += blockDim.x) s_pdelta[i] = 0;
        if (tid_x < 16) s_agg[tid_x] = 0;
        if (tid_x < 2) s_psum[tid_x] = 0;
        if (tid_x == 0) s_nlist = 0;
        __syncthreads();
        int pe[2], prow[2], pprio[2], pway[2];
        unsigned phint[2], ptag[2];
        bool pok[2], ptomb[2], pact[2];
        unsigned long long pkey[2], phome[2], pw0[2];
        const int64_t bs = blk_first + (tid_x & 15);
        // ---- the policy update folded in too (round 5; ProbeArgs::arena_w != nullptr: a set-associative fp32 tier alone with a
        // two-copy arena, evs_hash.h).  A thread that misses a key claims a way of the key's set right here -- it holds the
        // set's ways already: rank, ONE CAS, beside the priority raises -- and the lanes that gather the key's row from its
        // table for the interaction store it into the arena on the way (s_idx's second tile buffer carries the arena row to
        // the consume loop).  No miss lists, no update launch, no second read of anything.  What makes it legal inside the
        // launch that is still probing: the new word carries THIS batch's stamp (= pa.pend_stamp: a miss for every prober of
        // this launch -- the row may not be there yet) and names the arena copy the retired word does not, so a prober that
        // read the old word reads a row nobody is writing.
        const bool ins = pa.arena_w != nullptr;   // block-uniform
        if (ins) {
            for (int i = tid_x; i < kMaxBuckets; i += blockDim.x) s_udelta[i] = 0;
            if (tid_x < 2) s_ustat[tid_x] = 0;
        }
        // the thread's two keys side by side, one round trip per step for both: request rows, home slots, priorities
        // (a key whose home slot holds neither it nor nothing walks its chain with probe_ro: rare at load <= 0.25)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int f = ((int)tid_x >> 4) + 16 * h;
            pact[h] = f >= 1 && f < F && bs < blk_end;
            const int *rp = pact[h] ? pa.requests + bs * (int64_t)T + (f - 1) : reinterpret_cast<const int *>(dummy_i);
            prow[h] = *reinterpret_cast<const __attribute__((address_space(1))) int *>(reinterpret_cast<uintptr_t>(rp));
        }
        EVS_PTW(1);          // (the request rows are here)
        unsigned lw[2][8];   // the two keys' set ways (kept for the claim of a missed key)
        const bool sa = pa.sa.tags != nullptr;   // set-associative cache (evs_hash.h): one line per key, the priority inside the way word
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int f = ((int)tid_x >> 4) + 16 * h;
            const unsigned nrf = s_tile_nr[f & 31];
            pok[h] = pact[h] & (prow[h] >= 0) & ((unsigned)prow[h] < nrf);
            pkey[h] = ((unsigned long long)f << 32) | (unsigned)prow[h];   // table_1based = f
        }
        if (sa) {
            unsigned pset[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int f = ((int)tid_x >> 4) + 16 * h;
                sa_split(pa.sa, sa_perm(pa.sau, s_sa_base[f & 31] + (pok[h] ? (unsigned)prow[h] : 0u)), pset[h], ptag[h]);
                if (!pok[h]) pset[h] = 0u;
            }
            // (the probe is folded into this kernel for 8-way tiers only -- the host checks: evs_cache.hip -- so the way count
            //  is a compile-time constant: exactly two 16-byte loads and an 8-way search per key)
            SaLine line[2];
#pragma unroll
            for (int h = 0; h < 2; h++) sa_load<8>(pa.sa, pset[h], line[h]);
            __builtin_amdgcn_sched_barrier(0);   // both keys' set lines in one round trip
            EVS_PTW(2);          // (the set lines are here)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                unsigned w;
                const int way = sa_find<8>(pa.sa, line[h], ptag[h], w, pa.pend_stamp);
                const bool found = pok[h] && way >= 0;
                pe[h] = found ? (int)sa_entry(pa.sa, pset[h], (unsigned)way, w) : -1;
                pway[h] = way;
                pprio[h] = found ? sa_prio(w) : 0x7fffffff;
                pw0[h] = w; phint[h] = pset[h]; ptomb[h] = false;
#pragma unroll
                for (int j = 0; j < 8; j++) lw[h][j] = sa_way_word(line[h], j);
                if (found) atomicAdd(&s_agg[tid_x & 15], 1);
            }
        } else {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            phome[h] = mix64(pkey[h]) & pa.mask;
            pw0[h] = pa.slots[pok[h] ? phome[h] : 0];
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            pe[h] = -1; phint[h] = 0; ptomb[h] = false; pprio[h] = 0x7fffffff;
            if (pok[h]) {
                unsigned long long end_slot = phome[h];
                bool ht = false;
                int e = -1;
                if ((pw0[h] & kKeyMask) == pkey[h]) {
                    const unsigned fld = (unsigned)(pw0[h] >> kKeyBits);
                    e = fld >= kFieldPend ? -1 : (int)fld;
                } else if (pw0[h] != kEmpty) {
                    e = probe_ro(pa.slots, pa.mask, pkey[h], end_slot, pa.reusable_tomb, &ht);
                    if (e == kPending) e = -1;
                }
                pe[h] = e; phint[h] = (unsigned)(end_slot >> pa.hint_shift); ptomb[h] = ht;
            }
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if (pe[h] >= 0) {
                atomicAdd(&s_agg[tid_x & 15], 1);
                pprio[h] = pa.eagg[pe[h]];   // asked for now: it travels while the block meets
            }
        }
        }
        __syncthreads();
        EVS_PT(3);
        const int agg = s_agg[tid_x & 15];
        // a missed key's claim: duplicate / victim / the CAS sent here, looked at behind the raises below (one round trip for both).
        // (Looked at behind the ROW requests instead -- the claim kept across them, block barriers that order LDS only -- was
        // built and measured on one box: 34.4 against 32.5 us per batch, the kernel sits at its 128 registers; the new rows as
        // predicated buffer stores instead of stores under an exec mask: 33.3.  The lean form of the first -- only the CAS's answer,
        // the victim's word and the way carried across the requests, a lost CAS re-reading its set: 128 registers, no spill --
        // measured EQUAL, 34.8-35.2 against 34.8-35.5 on its box: the CAS's round trip is not what the launch waits for.
        // tools/cache_lib_ab.sh)
        int uwon[2] = {-1, -1};
        unsigned uprev[2] = {0u, 0u};
        SaPick upk[2] = {{-1, 0, -1, 0u, 0u}, {-1, 0, -1, 0u, 0u}};
        bool uwait[2] = {false, false};
        if (ins) {
#pragma unroll
            for (int h = 0; h < 2; h++)
                if (pok[h] && pe[h] < 0 && !(pa.xflags & 2)) uwait[h] = sa_claim_issue(pa.sa, pa.pend_stamp, phint[h], ptag[h], agg, lw[h], upk[h], uprev[h], s_udelta);
        }
        EVS_PT(10);          // (the claims are ranked and sent)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int f = ((int)tid_x >> 4) + 16 * h;
            // monotone max like update_agg_hit; the plain read first keeps hot entries from serialising on one address
            if (pe[h] >= 0 && pprio[h] < agg) {
                int old;
                if (sa) old = sa_raise(pa.sa, sa_ways_ptr(pa.sa, phint[h]) + pway[h], (unsigned)pw0[h], agg);
                else { old = atomicMax(&pa.eagg[pe[h]], agg); old = old < agg ? old : -1; }
                if (old >= 0) { atomicSub(&s_pdelta[old], 1); atomicAdd(&s_pdelta[agg], 1); }
            }
            int v = -1;
            if (f == 0) v = bs < blk_end ? (int)bs : -1;                 // x: the sample number
            else if (pact[h]) v = pe[h] >= 0 ? (int)(0x40000000u | (unsigned)pe[h]) : (pok[h] ? prow[h] : -1);
            s_idx[(int)tid_x + 256 * h] = v;
            if (ins) {   // where the gathered row of a missed key goes (-1: nowhere)
                if (uwait[h]) uwon[h] = sa_claim_finish(pa.sa, pa.pend_stamp, phint[h], ptag[h], agg, lw[h], upk[h], uprev[h], s_udelta, s_ustat);
                s_idx[512 + (int)tid_x + 256 * h] = uwon[h];
            }
            if (pact[h]) {
                const int64_t m = bs * (int64_t)T + (f - 1);
                if (pa.hit) pa.hit[m] = pe[h] >= 0;
                if (pa.miss_rec != nullptr && pok[h] && pe[h] < 0) {
                    const int at = atomicAdd(&s_nlist, 1);
                    pa.miss_rec[(int64_t)blockIdx.x * pa.list_cap + at] =
                        make_uint4((unsigned)prow[h], (unsigned)(f - 1) | ((unsigned)agg << 8) | (ptomb[h] ? 0x10000u : 0u), phint[h], sa ? ptag[h] : (unsigned)m);
                }
            }
            if (f == 1 && bs < blk_end) { atomicAdd(&s_psum[0], agg); if (agg == T) atomicAdd(&s_psum[1], 1); }
        }
        EVS_PT(11);          // (raises done, claims looked at, the tile written)
        __syncthreads();
        EVS_PT(12);
        if (tid_x < 40) {   // the block's totals into one of the replica rows (folded by the cache's close)
            const int i = tid_x;
            const int v = i <= T ? s_pdelta[i] : i == 38 ? s_psum[0] : i == 39 ? s_psum[1] : 0;
            if (v) atomicAdd(&pa.part1[(blockIdx.x % 32) * 40 + i], v);
        }
        if (ins && tid_x < 40) {   // the inserts' totals, as the update kernels leave them (folded by the cache's close)
            const int i = tid_x;
            const int v = i <= T ? s_udelta[i] : i == 33 ? s_ustat[0] : i == 34 ? s_ustat[1] : 0;
            if (v) atomicAdd(&pa.part2[(blockIdx.x % 32) * 40 + i], v);
        }
        if (pa.list_cnt != nullptr && tid_x == 0) pa.list_cnt[blockIdx.x] = s_nlist;
        EVS_PTW(5);          // (everything the head has sent is answered; the head is over)
    } else {
        if (!stk) tile_load(0);
        tile_store(0);
        if constexpr (CHECK) {
            verify();
            if (__syncthreads_or(my_ragged)) {   // block-uniform, rare: this block's samples with general bag semantics
                slow_block();
                return;
            }
            bad |= fast_bad;
            EVS_PT(5);
        } else {
            __syncthreads();
        }
    }
    // scheduling barriers (EVS_RF_SB, developer A/B; bit 1: the four samples' requests leave in sample order, bit 0: the
    // stores of sample u - 1 leave inside iteration u).  Left alone, hipcc 7.2 interleaves the requests of samples 0 and 1
    // and sinks EVERY output store below the last MFMA of the last sample; pinned is 0.1-0.3 us faster at B = 16 384.
#ifndef EVS_RF_SB
#define EVS_RF_SB 3
#endif
#pragma unroll
    for (int u = 0; u < D; u++) {
        issue(u, ring[u]);      // (samples past the block's end: every lane reads the zero page)
        if constexpr ((EVS_RF_SB & 2) != 0) __builtin_amdgcn_sched_barrier(0);
    }
    float x_srv[SERVE ? D : 1][SERVE ? (d + 63) / 64 : 1];   // SERVE: x[b] once more, fresh from memory (rf_ld_*), asked for with the row requests
    if constexpr (SERVE) {
#pragma unroll
        for (int u = 0; u < D; u++) {
            const int64_t b = blk_first + wave_in_block + 4 * (int64_t)u;
#pragma unroll
            for (int h = 0; h < (d + 63) / 64; h++) {
                const int e = lane + 64 * h;
                const float *xp = (u < n_samples && e < d) ? sd.x + b * sd.x_stride + e : reinterpret_cast<const float *>(zeros_l);
                x_srv[u][h] = rf_ld_f32<true>(xp);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);   // the scheduler would otherwise sink three of the four requests below the first consume
    // the lane-invariant staging / flush offsets: computed here, under the row requests' round trip, not in front of them
    // (16 waves per CU start in step: every instruction in front of the first load is paid by all of them at once)
    fill_stage();
    fill_flush();
    EVS_PT(6);               // (the row requests are out)
    const bool ins_blk = PROBE && args.probe.arena_w != nullptr && !(args.probe.xflags & 1);   // block-uniform: missed keys' rows go into the cache arena on the way
#pragma unroll
    for (int u = 0; u < D; u++) {
        const int64_t b = blk_first + wave_in_block + 4 * (int64_t)u;   // wave-uniform
        // the image of sample u: what the row DMA of the LDS loop would have left in the slot
#pragma unroll
        for (int j = 0; j < NJ; j++) *reinterpret_cast<f32x4 *>(my_lds + j * 1024 + lane * 16) = ring[u][j];
        if constexpr (SERVE) {   // row 0 of the image = x[b]: the copy that came through the agent-scope loads
#pragma unroll
            for (int h = 0; h < (d + 63) / 64; h++) {
                const int e = lane + 64 * h;
                if (e < d) reinterpret_cast<float *>(my_lds)[e] = x_srv[u][h];
            }
        }
        if constexpr (PROBE) {
            if (ins_blk) {   // the rows of the ways this block claimed: from the registers that gathered them into the arena
#pragma unroll
                for (int j = 0; j < NJ; j++) {
                    const int e = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(s_idx) + 2048 + idx_lds[j] + 16 * u);
                    if (e >= 0)
                        *reinterpret_cast<f32x4 *>(args.probe.arena_w + (unsigned long long)(unsigned)e * (unsigned)row_bytes + (unsigned)dma_piece16) = ring[u][j];
                }
            }
        }
        float4 a[NR][CQ > 0 ? CQ : 1];
        float ar[NR][REM > 0 ? REM : 1];
#pragma unroll
        for (int rr = 0; rr < NR; rr++) {
#pragma unroll
            for (int c = 0; c < CQ; c++) a[rr][c] = *reinterpret_cast<const float4 *>(my_lds + lds_off[rr] + c * 16);
#pragma unroll
            for (int m = 0; m < REM; m++)
                ar[rr][m] = *reinterpret_cast<const float *>(my_lds + rem_off[rr] + m * 16);
        }
        float xv[(d + 63) / 64];   // x[b] is row 0 of the image
#pragma unroll
        for (int h = 0; h < (d + 63) / 64; h++) {
            const int e = lane + 64 * h;
            xv[h] = reinterpret_cast<const float *>(my_lds)[e < d ? e : 0];
        }
        flush_out(b - 4, u > 0 && u - 1 < n_samples);    // sample u-1 leaves under the MFMAs of sample u
        f32x4 c00, c10, c11;
        interact(a, ar, c00, c10, c11);
        if constexpr (MLP) my_out = s_out[wave_in_block + 4 * u];
        // never-stored elements go to a dump slot: behind the row, or (MLP) in a block of their own
        const int dump = MLP ? (int)(reinterpret_cast<char *>(&s_dump[wave_in_block * 16 + r16]) - reinterpret_cast<char *>(my_out))
                             : 4 * (OUT_MAX + r16);
        // stage the output row: x passthrough, then the packed lower triangle straight from the accumulators
        if constexpr (!MLP) {
#pragma unroll
            for (int h = 0; h < (d + 63) / 64; h++) *reinterpret_cast<float *>(reinterpret_cast<char *>(my_out) + xv_off[h]) = xv[h];
#pragma unroll
            for (int v = 0; v < 4; v++) {
                *reinterpret_cast<float *>(reinterpret_cast<char *>(my_out) + zo00h[v]) = c00[v];
                if constexpr (NT == 2) {
                    *reinterpret_cast<float *>(reinterpret_cast<char *>(my_out) + zo10h[v]) = c10[v];
                    *reinterpret_cast<float *>(reinterpret_cast<char *>(my_out) + zo11h[v]) = c11[v];
                }
            }
        } else {
#pragma unroll
        for (int h = 0; h < (d + 63) / 64; h++) {
            const int e = lane + 64 * h;
            *reinterpret_cast<float *>(reinterpret_cast<char *>(my_out) + (e < d ? 4 * e : dump)) = xv[h];
        }
        if constexpr (MLP) {   // zero the padding columns [K, kp) of the GEMM operand
            if (lane < args.kp - out_row) my_out[out_row + lane] = 0.f;
        }
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
        }
        if constexpr ((EVS_RF_SB & 1) != 0) __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (MLP) my_out = s_out[wave_in_block + 12];
    flush_out(blk_first + wave_in_block + 12, n_samples == 4);
    EVS_PT(8);               // (wave 0's last stores are out)
    EVS_PTW(9);              // (... and acknowledged)
    if (bad) atomicOr(args.err, 1);
    if constexpr (MLP) {
        __syncthreads();   // the chunk's 16 staged rows are complete
        const int kp = args.kp, kq = kp >> 2, n1 = args.n1;
        const int n_tiles = (n1 + 15) >> 4;
        const float *arow = &s_out[r16][q * kq];                          // A: sample r16, k-slot q
        for (int nt = wave_in_block; nt < n_tiles; nt += 4) {
            const int n = 16 * nt + r16;
            const float *wrow = args.w1p + (size_t)n * kp + q * kq;       // B: output n, k-slot q (rows padded to 16 * n_tiles)
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            // software pipeline: the operand pieces of step group g + 1 (kPf steps of 4 MFMAs) are requested before the
            // MFMAs of group g issue -- W1 comes from L2 (~500 cycles), a load-use loop runs at 4 MFMAs per round trip
            // (measured at B = 16 384, n1 = 512: 229 -> 137 us; two tiles per wave sharing the A pieces: 195 us, dropped)
            constexpr int kPf = 5;
            const int nj = kq >> 2;                    // 16-byte pieces per k-slot
            f32x4 bq[kPf];
            float4 aq[kPf];
#pragma unroll
            for (int p = 0; p < kPf; p++) {
                const int j = p < nj ? p : 0;
                bq[p] = *reinterpret_cast<gf4_t>(reinterpret_cast<uintptr_t>(wrow + 4 * j));
                aq[p] = *reinterpret_cast<const float4 *>(arow + 4 * j);
            }
            for (int g = 0; g < nj; g += kPf) {
                f32x4 bc[kPf];
                float4 ac[kPf];
#pragma unroll
                for (int p = 0; p < kPf; p++) { bc[p] = bq[p]; ac[p] = aq[p]; }
#pragma unroll
                for (int p = 0; p < kPf; p++) {        // next group (clamped: the last group re-reads piece 0, unused)
                    const int j = g + kPf + p < nj ? g + kPf + p : 0;
                    bq[p] = *reinterpret_cast<gf4_t>(reinterpret_cast<uintptr_t>(wrow + 4 * j));
                    aq[p] = *reinterpret_cast<const float4 *>(arow + 4 * j);
                }
#pragma unroll
                for (int p = 0; p < kPf; p++) {
                    if (g + p < nj) {
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[p].x, bc[p][0], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[p].y, bc[p][1], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[p].z, bc[p][2], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[p].w, bc[p][3], acc, 0, 0, 0);
                    }
                }
            }
            const float bias = n < n1 ? args.b1[n] : 0.f;
#pragma unroll
            for (int v = 0; v < 4; v++) {   // lane holds Z1[sample 4q + v][output n]
                const int m = 4 * q + v;
                float z = acc[v] + bias;
                if (args.relu) z = z > 0.f ? z : 0.f;
                if (m < blk_n && n < n1) args.z1[(blk_first + m) * (int64_t)n1 + n] = z;
            }
        }
    }
}

template <int CQ, int REM, int NT, int D, bool MLP = false, bool IDS = false, bool PROBE = false, bool CHECK = false>
__global__ void __launch_bounds__(256, (MLP ? 3 : (CQ >= 4 ? 2 : EVS_RF_LB))) emb_interact_rf_kernel(const FusedArgs args) {
    rf_body<CQ, REM, NT, D, MLP, IDS, PROBE, CHECK, false>(args, (const FusedArgs *)__builtin_amdgcn_kernarg_segment_ptr(), RfServeDesc{}, (int)blockIdx.x);
}

static int rf_mode() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("EVS_FUSED_RF"); v = e ? atoi(e) : 1; }   // developer switch: 0 = the LDS-DMA loop
    return v;
}

// developer switch: bytes of dynamic LDS added to every block, i.e. fewer blocks per CU (27 KB static: 4 per CU; +20 KB: 3;
// +40 KB: 2) -- does a launch whose blocks arrive in two waves overlap its own fill and drain?
static unsigned rf_pad_lds() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("EVS_FUSED_RF_PADLDS"); v = e ? atoi(e) : 0; }
    return (unsigned)v;
}
// developer switch: samples per block (16: one generation of co-resident blocks at B = 16 384; 8 / 4: the grid arrives in two / four
// generations whose heads and bodies can overlap -- the per-block timeline of tools/probe_stage_probe.py asked for the experiment.
// Measured on the plain launch, B = 16 384: 19.2 us at 16, 28.8 at 8, 48.7 at 4 -- but a wave still issues four samples' worth of
// requests (the missing ones against the zero page; the body does not build at a depth of 2), so this prices a block's fixed cost,
// not two generations as such: EVS_FUSED_RF_PADLDS -- the same blocks, fewer per CU -- is the fair form of that question.
// Plain launches only: the cache tier's per-block buffers are sized for 16.)
static int rf_tile_per() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("EVS_FUSED_RF_TILE"); v = e ? atoi(e) : 16; if (v != 4 && v != 8 && v != 12) v = 16; }
    return v;
}
template <auto K>
static void launch_rf_grid(FusedArgs a, hipStream_t st) {
    a.tile_per = rf_tile_per();   // (default 16) one 16-sample chunk per block: 4 samples per wave, all requested at once
    hipLaunchKernelGGL(K, dim3((unsigned)((a.B + a.tile_per - 1) / a.tile_per)), dim3(256), rf_pad_lds(), st, a);
}

// the batch sizes this form is for: every block resident at once (4 blocks of 256 threads per CU at 128 VGPRs)
static int64_t rf_max_batch() {
    static int64_t v = -1;
    if (v < 0) { const char *e = getenv("EVS_FUSED_RF_MAX_B"); v = e ? atoll(e) : 16ll * kNumCu * EVS_RF_LB; }
    return v;
}

bool launch_rf_mlp(const FusedArgs &a, hipStream_t st) {
    if (a.F > kTileMaxF || a.bag1 != 1 || a.kp > kMlpRowStride - 4 || (a.kp & 15) || a.kp < a.d + a.P) return false;
    const bool nt2 = a.F > 16;
    switch (a.d) {
    case 16:
        if (nt2) launch_rf_grid<emb_interact_rf_kernel<1, 0, 2, 4, true>>(a, st); else launch_rf_grid<emb_interact_rf_kernel<1, 0, 1, 4, true>>(a, st);
        return true;
    case 32:
        if (nt2) launch_rf_grid<emb_interact_rf_kernel<2, 0, 2, 4, true>>(a, st); else launch_rf_grid<emb_interact_rf_kernel<2, 0, 1, 4, true>>(a, st);
        return true;
    case 36:
        if (nt2) launch_rf_grid<emb_interact_rf_kernel<2, 1, 2, 4, true>>(a, st); else launch_rf_grid<emb_interact_rf_kernel<2, 1, 1, 4, true>>(a, st);
        return true;
    default:
        return false;
    }
}

bool rf_ids_supported(int64_t B, int F, int d) {
    return rf_mode() && F <= kTileMaxF && B <= rf_max_batch() && (d == 16 || d == 32 || d == 36);
}
bool launch_rf_ids(const FusedArgs &a, hipStream_t st) {
    if (!rf_ids_supported(a.B, a.F, a.d) || !a.row_ids || !a.arena) return false;
    const bool nt2 = a.F > 16;
    switch (a.d) {
    case 16:
        if (nt2) launch_rf_grid<emb_interact_rf_kernel<1, 0, 2, EVS_RF_DEPTH, false, true>>(a, st); else launch_rf_grid<emb_interact_rf_kernel<1, 0, 1, EVS_RF_DEPTH, false, true>>(a, st);
        return true;
    case 32:
        if (nt2) launch_rf_grid<emb_interact_rf_kernel<2, 0, 2, EVS_RF_DEPTH, false, true>>(a, st); else launch_rf_grid<emb_interact_rf_kernel<2, 0, 1, EVS_RF_DEPTH, false, true>>(a, st);
        return true;
    default:
        if (nt2) launch_rf_grid<emb_interact_rf_kernel<2, 1, 2, EVS_RF_DEPTH, false, true>>(a, st); else launch_rf_grid<emb_interact_rf_kernel<2, 1, 1, EVS_RF_DEPTH, false, true>>(a, st);
        return true;
    }
}

bool launch_rf_probe(const FusedArgs &a, hipStream_t st) {
    if (!rf_ids_supported(a.B, a.F, a.d) || (!a.probe.slots && !a.probe.sa.tags) || !a.arena) return false;
    const bool nt2 = a.F > 16;
    switch (a.d) {
    case 16:
        if (nt2) launch_rf_grid<emb_interact_rf_kernel<1, 0, 2, EVS_RF_DEPTH, false, false, true>>(a, st); else launch_rf_grid<emb_interact_rf_kernel<1, 0, 1, EVS_RF_DEPTH, false, false, true>>(a, st);
        return true;
    case 32:
        if (nt2) launch_rf_grid<emb_interact_rf_kernel<2, 0, 2, EVS_RF_DEPTH, false, false, true>>(a, st); else launch_rf_grid<emb_interact_rf_kernel<2, 0, 1, EVS_RF_DEPTH, false, false, true>>(a, st);
        return true;
    default:
        if (nt2) launch_rf_grid<emb_interact_rf_kernel<2, 1, 2, EVS_RF_DEPTH, false, false, true>>(a, st); else launch_rf_grid<emb_interact_rf_kernel<2, 1, 1, EVS_RF_DEPTH, false, false, true>>(a, st);
        return true;
    }
}

// the same launch with lS_o given (whole batches, FusedArgs::bag1 == 3): the kernel checks the offsets of its 16 samples itself
// d = 64 (the Terabyte scripts' width, round 3): the same kernel with CQ = 4 -- 16 lanes per row, 4 rows per load, 7 loads
// per sample, 135 VGPRs (three blocks per CU).  Same box, B = 16 384: one index per bag declared 30.4 -> 25.6 us (0.59 -> 0.70 of
// peak), lS_o given 33.2 -> 27.5 us; B = 65 536: 91.5 -> 89.5 us.
static bool rf_d64() {
    static const bool on = !(getenv("EVS_FUSED_RF_D64") && getenv("EVS_FUSED_RF_D64")[0] == '0');
    return on;
}

bool launch_rf_check(const FusedArgs &a, hipStream_t st) {
    static const bool on = !(getenv("EVS_FUSED_RF_CHECK") && getenv("EVS_FUSED_RF_CHECK")[0] == '0');
    if (!on || !rf_mode() || a.F > kTileMaxF || a.bag1 != 3 || a.B > rf_max_batch()) return false;
    const bool nt2 = a.F > 16;
    switch (a.d) {
    case 16:
        if (nt2) launch_rf_grid<emb_interact_rf_kernel<1, 0, 2, EVS_RF_DEPTH, false, false, false, true>>(a, st); else launch_rf_grid<emb_interact_rf_kernel<1, 0, 1, EVS_RF_DEPTH, false, false, false, true>>(a, st);
        return true;
    case 32:
        if (nt2) launch_rf_grid<emb_interact_rf_kernel<2, 0, 2, EVS_RF_DEPTH, false, false, false, true>>(a, st); else launch_rf_grid<emb_interact_rf_kernel<2, 0, 1, EVS_RF_DEPTH, false, false, false, true>>(a, st);
        return true;
    case 36:
        if (nt2) launch_rf_grid<emb_interact_rf_kernel<2, 1, 2, EVS_RF_DEPTH, false, false, false, true>>(a, st); else launch_rf_grid<emb_interact_rf_kernel<2, 1, 1, EVS_RF_DEPTH, false, false, false, true>>(a, st);
        return true;
    case 64:
        if (!rf_d64()) return false;
        if (nt2) launch_rf_grid<emb_interact_rf_kernel<4, 0, 2, EVS_RF_DEPTH, false, false, false, true>>(a, st); else launch_rf_grid<emb_interact_rf_kernel<4, 0, 1, EVS_RF_DEPTH, false, false, false, true>>(a, st);
        return true;
    default:
        return false;
    }
}

// K batches in one launch: K * ceil(B / 16) one-chunk blocks; the hardware hands a CU the next block as one retires, so the
// drain of a batch's last blocks runs under the fill of the next batch's first ones (what two alternating streams give a
// caller, without any stream: cross-stream event waits cost more here than they return -- measured, docs/HISTORY.md 3.2d)
bool rf_multi_supported(int64_t B, int F, int d) {
    return rf_mode() && F <= kTileMaxF && B >= 1 && (d == 16 || d == 32 || d == 36 || (d == 64 && rf_d64()));
}
template <auto K>
static void launch_rf_multi_grid(FusedArgs a, hipStream_t st) {
    a.tile_per = 16;
    a.multi_cpb = (int)((a.B + 15) / 16);
    hipLaunchKernelGGL(K, dim3((unsigned)(a.multi_cpb * a.multi_n)), dim3(256), 0, st, a);
}
bool launch_rf_multi(const FusedArgs &a, hipStream_t st) {
    if (!rf_multi_supported(a.B, a.F, a.d) || a.multi_n < 1 || a.multi_n > kMultiMax || !(a.bag1 == 1 || a.bag1 == 3)) return false;
    const bool nt2 = a.F > 16;
#define EVS_RF_MULTI(CQ_, REM_)                                                                                              \
    do {                                                                                                                     \
        if (a.bag1 == 1) {                                                                                                   \
            if (nt2) launch_rf_multi_grid<emb_interact_rf_kernel<CQ_, REM_, 2, EVS_RF_DEPTH>>(a, st);                        \
            else launch_rf_multi_grid<emb_interact_rf_kernel<CQ_, REM_, 1, EVS_RF_DEPTH>>(a, st);                            \
        } else {                                                                                                             \
            if (nt2) launch_rf_multi_grid<emb_interact_rf_kernel<CQ_, REM_, 2, EVS_RF_DEPTH, false, false, false, true>>(a, st); \
            else launch_rf_multi_grid<emb_interact_rf_kernel<CQ_, REM_, 1, EVS_RF_DEPTH, false, false, false, true>>(a, st);     \
        }                                                                                                                    \
    } while (0)
    switch (a.d) {
    case 16: EVS_RF_MULTI(1, 0); return true;
    case 32: EVS_RF_MULTI(2, 0); return true;
    case 36: EVS_RF_MULTI(2, 1); return true;
    case 64: EVS_RF_MULTI(4, 0); return true;
    default: return false;
    }
#undef EVS_RF_MULTI
}

// The plain launch (one index per bag declared) also takes batches ABOVE one resident generation for d = 36 and 16: with
// the round-3 head the one-chunk blocks beat the LDS-DMA loop there too (same box, d = 36: B = 32 768 36.2 -> 32.4 us,
// 65 536 62.1 -> 60.0, 131 072 120.0 -> 114.6, 262 144 235 -> 227; d = 16 at 65 536: 42.0 -> 38.5; d = 32 loses, 48.6 ->
// 50.6, and keeps the loop).  The checked / probing / row-id forms stay at one generation (no gain measured for the
// checked form: 65.6 vs 65.7 us at 65 536).  EVS_FUSED_RF_MAX_B, when set, bounds every form.
static int64_t rf_max_batch_plain(int d) {
    if (getenv("EVS_FUSED_RF_MAX_B") || d == 32) return rf_max_batch();
    if (d == 64) return 1ll << 22;
    return 1ll << 22;
}

bool launch_rf(const FusedArgs &a, hipStream_t st) {
    if (!rf_mode() || a.F > kTileMaxF || a.bag1 != 1 || a.B > rf_max_batch_plain(a.d)) return false;
    const bool nt2 = a.F > 16;
    switch (a.d) {
    case 16:
        if (nt2) launch_rf_grid<emb_interact_rf_kernel<1, 0, 2, EVS_RF_DEPTH>>(a, st); else launch_rf_grid<emb_interact_rf_kernel<1, 0, 1, EVS_RF_DEPTH>>(a, st);
        return true;
    case 32:
        if (nt2) launch_rf_grid<emb_interact_rf_kernel<2, 0, 2, EVS_RF_DEPTH>>(a, st); else launch_rf_grid<emb_interact_rf_kernel<2, 0, 1, EVS_RF_DEPTH>>(a, st);
        return true;
    case 36:
        if (nt2) launch_rf_grid<emb_interact_rf_kernel<2, 1, 2, EVS_RF_DEPTH>>(a, st); else launch_rf_grid<emb_interact_rf_kernel<2, 1, 1, EVS_RF_DEPTH>>(a, st);
        return true;
    case 64:   // 28 VGPRs per sample in flight (135 in all: three blocks per CU); EVS_FUSED_RF_D64=0: the LDS-DMA loop
        if (!rf_d64()) return false;
        if (nt2) launch_rf_grid<emb_interact_rf_kernel<4, 0, 2, EVS_RF_DEPTH>>(a, st); else launch_rf_grid<emb_interact_rf_kernel<4, 0, 1, EVS_RF_DEPTH>>(a, st);
        return true;
    default:   // (d = 128 was built the same way -- CQ = 8, 14 loads per sample, 248 VGPRs, one or two blocks per CU -- and lost to the
               //  LDS-DMA loop: 57.5 vs 52.5 us at B = 16 384, 220 vs 182 us at 65 536; removed)
        return false;
    }
}


// ======================================================================================================================
// The fused launch as a RESIDENT DISPATCHER (round 6; evs_emb_interact_serve_*).  BASELINE's metric is "lookups/sec + p50 batch
// latency", and a launch that is waited for spends 12 of its 31 us (B = 16 384) outside the kernel: ~6 us of host time in the
// launch call, ~3 us until the command processor has dispatched 1 024 blocks, ~3 us until the completion signal is visible;
// below ~4 000 samples a launch IS that floor (7-8 us per batch whatever B).  Here the grid stays on the device and takes
// batches from a mailbox in pinned host memory (the reference's loop calls the pair once per batch: dlrm_s_pytorch.py:596-601,
// the inference loop :801-836):
//   request ring   (host -> device) kSrvSlots descriptors of 64 bytes: x, (T, B) indices, (T, B) offsets, R, B, strides, the block the
//                  batch's first chunk goes to; each 32-byte sector closed by the sequence number (written last: a sector is
//                  accepted only when its guard holds the number awaited, whatever granularity the bus delivers it in);
//   leader         block 0, one wavefront: polls the next FOUR ring slots as one 64-lane system-scope load, copies every
//                  descriptor that has arrived into device memory (agent-scope write-through stores) and publishes its number
//                  in kSrvReplicas replica lines (a thousand blocks polling ONE word cost more than the work: r04_atomic_probe);
//   workers        blocks 1 .. G-1: lane 0 polls its replica line; a new number -> an agent-scope acquire (inputs written by
//                  other launches), the descriptor, then rf_body<..., SERVE> for every chunk c of the batch with
//                  (first_block + c) % (G - 1) == this worker -- consecutive small batches land on different blocks and overlap;
//                  R leaves as agent-scope write-through stores (a resident kernel has no end-of-kernel release), every wave
//                  waits for its stores (vmcnt(0)), then ONE arrival atomic per block; the block that completes the batch
//                  writes the batch's number into the ANSWER ring in host memory, which the caller spins on;
//   leaving        the leader publishes "stop" after idle_ticks without a request (or when the host's control word says so):
//                  every block returns; the next post starts the grid again.  While it is resident the grid holds its CUs
//                  (G = 4 blocks per CU by default): kernels of other streams run when it has left.
// Same bits as the launch form (the same rf_body).  Rules for the caller: the inputs of a batch are complete when it is posted
// (post on the host after the producer's stream has been synchronised), and R may be read by anything started after evs_emb_interact_serve_wait has returned.
constexpr int kSrvSlots = 64;
constexpr int kSrvReplicas = 32;
#ifndef EVS_SRV_SUB
#define EVS_SRV_SUB 32
#endif
constexpr unsigned kSrvSub = EVS_SRV_SUB;         // sub-counters a batch's arrivals are spread over (a power of two; developer A/B: 128 -- a quarter of
                                                  // the arrivals per counter, but 128 answer words written by 128 blocks for the host to collect -- 31.2 us
                                                  // waited for at B = 16 384 against 26.0, 19.6 against 14.2 at B = 2 048; 8: 26.9 / 14.7)
constexpr unsigned kSrvAnsLine = kSrvSub;  // words per slot of the answer ring: one per sub-counter of the batch's arrivals
struct SrvDesc { unsigned w[16]; };   // w0-1 x, w2-3 idx, w4-5 off, w6 B, w7 seq | w8-9 R, w10 x_stride, w11 first block, w12 idx stride, w13 off stride, w14 -, w15 seq
struct SrvState {
    unsigned pub[kSrvReplicas][32];       // line r: word 0 = the last published sequence number, word 1 = the generation (launch number) of the grid that has LEFT behind it
    SrvDesc desc[kSrvSlots];              // the leader's copies of the descriptors
    // per slot: chunks finished (a thousand arrivals on ONE word serialise at the memory-side atomic unit: ~6 us,
    // r04_atomic_probe, and the blocks of a full batch all finish together): chunk c arrives at sub-counter c % 32 (a line each);
    // whoever completes sub-counter r writes the batch's number into word r of the slot's ANSWER line in host memory, and the
    // host takes the batch as answered when all of them hold it (the first form had a top counter the 32 completers arrived at:
    // one more dependent atomic round trip, ~1 us, in front of every answer)
    unsigned arrived[kSrvSlots][kSrvSub][32];
    unsigned prog[4096];                  // host-published front end: the last batch block b has looked at (kept across launches)
};
// HOST-PUBLISHED front end (round 6; parts whose device memory the host can address -- large BAR): the host writes a batch's
// descriptor straight into device memory through the PCIe aperture -- into its slot of a descriptor ring and into the first half
// of every replica line -- and every block's poll of its line is the whole way in: no leader reading the mailbox over the bus
// and republishing (tools/mailbox_probe.hip: post -> 32 pollers -> arrival counter -> answer 3.0 us; the leader's hop alone
// was ~3).  Fine-grained device memory (a poll of plain device memory is served by the polling XCD's L2 once the line is in it).
//   line r, words 0..15   the descriptor of the LATEST batch posted; its guards (words 7, 15) are its number    [host, through the aperture]
//           word 16       the launch number of a grid that is leaving                                           [the leader]
//   line 0, word 18       the host's request to leave                                                           [host]
// Nobody publishes "up to here" when the grid leaves -- the host posts whenever it likes, a block may have seen a batch the
// leader has not --: every block keeps ITS OWN count of batches looked at (SrvState::prog), stores it on its way out, and its
// successor in the next launch goes on from there; a batch is answered when every chunk has arrived, whichever launch ran it.
struct SrvFront {
    unsigned line[kSrvReplicas][32];
    SrvDesc desc[kSrvSlots];
};
struct SrvArgs {
    const FusedArgs *tmpl;                // tables, shapes (device memory; written before the grid starts, never while it runs)
    SrvState *st;
    SrvFront *front;                      // the host-published front end, or nullptr: the leader reads the mailbox in host memory
    volatile unsigned *req;               // request ring (device address of the mapped host block)
    volatile unsigned *ctl;               // word 0: stop
    volatile unsigned *ans;               // answer ring: kSrvSlots lines of kSrvAnsLine words (word r = the number of the batch whose sub-counter r is complete);
                                          // behind them the status line: word 0 = alive, word 1 = the last published number
    unsigned start_seq;                   // the last number published by an earlier run of the grid
    unsigned gen;                         // this launch's number (>= 1): "stop" is the leader writing it into word 1 of the replica lines --
                                          // what an earlier launch left there never matches, so nothing has to be cleared between launches
    long long idle_ticks;
};

// which block runs chunk c of a batch (the same rule on both sides).  A batch with fewer chunks than the grid has blocks goes to the
// WORKERS only (blocks 1 .. G-1, starting at the batch's `first`: consecutive small batches land on different blocks), so the
// leader stays at its mailbox; a batch with at least G chunks uses every block, the leader's too (at B = 16 384 = 1 024 chunks a
// grid of 1 024 blocks would otherwise hand one worker two chunks: the batch's tail)
__device__ __forceinline__ unsigned srv_block_of(unsigned c, unsigned n_chunks, unsigned first, unsigned G) {
    return n_chunks >= G ? (first + c) % G : 1u + (first + c) % (G - 1u);
}
// the chunks of one batch that fall to block `me`, then the arrival: every wave waits for its stores of R (written through),
// ONE atomic per block, and the block that completes the batch writes its number into the answer ring in host memory
// a batch whose chunks this block has run but not yet reported: the report (srv_arrive) waits until the block's stores of R are
// known to be complete -- which the worker learns for free from its NEXT poll (vector-memory operations complete in issue
// order: a load that has returned has every earlier store of its wave behind it), so the drain of a batch's stores runs under
// the poll for the next one instead of in front of it
struct SrvPending { unsigned k, c0, step, n_chunks; };
__device__ __forceinline__ void srv_arrive(const SrvArgs &sv, const SrvPending &p) {
    const unsigned slot = p.k % (unsigned)kSrvSlots, n_chunks = p.n_chunks;
    unsigned *ans = const_cast<unsigned *>(sv.ans) + slot * kSrvAnsLine;
    if (n_chunks == 1u) {             // (a batch of one chunk: nobody else to wait for)
        __hip_atomic_store(ans, p.k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    for (unsigned c = p.c0; c < n_chunks; c += p.step) {
        const unsigned r = c & (kSrvSub - 1u), expect = n_chunks / kSrvSub + (r < (n_chunks & (kSrvSub - 1u)) ? 1u : 0u);
        const unsigned before = __hip_atomic_fetch_add(&sv.st->arrived[slot][r][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (before + 1u == expect) {
            __hip_atomic_store(&sv.st->arrived[slot][r][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // back to zero for the slot's next use
            __hip_atomic_store(ans + r, p.k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
// the chunks of one batch that fall to block `me`; -> true: it ran some (pend says which: report them with srv_arrive once the
// stores are known to be complete)
template <int CQ, int REM, int NT>
__device__ __forceinline__ bool srv_run_batch(const SrvArgs &sv, const FusedArgs &la, const unsigned (&w)[16], unsigned k, unsigned me, unsigned G, bool &first, SrvPending &pend) {
    RfServeDesc sd;
    sd.x = reinterpret_cast<const float *>((uintptr_t)(((unsigned long long)w[1] << 32) | w[0]));
    sd.idx = reinterpret_cast<const int64_t *>((uintptr_t)(((unsigned long long)w[3] << 32) | w[2]));
    sd.off = reinterpret_cast<const int64_t *>((uintptr_t)(((unsigned long long)w[5] << 32) | w[4]));
    sd.R = reinterpret_cast<float *>((uintptr_t)(((unsigned long long)w[9] << 32) | w[8]));
    sd.B = (int64_t)w[6]; sd.x_stride = (int64_t)w[10]; sd.idx_stride = (int64_t)w[12]; sd.off_stride = (int64_t)w[13];
    const unsigned n_chunks = (w[6] + 15u) >> 4;
    unsigned c0, step;
    if (n_chunks >= G) { c0 = (me + G - w[11] % G) % G; step = G; }
    else { if (me == 0u) return false; c0 = (me - 1u + (G - 1u) - w[11] % (G - 1u)) % (G - 1u); step = G - 1u; }
    unsigned mine = 0u;
    for (unsigned c = c0; c < n_chunks; c += step) {
        rf_body<CQ, REM, NT, EVS_RF_DEPTH, false, false, false, true, true>(la, sv.tmpl, sd, (int)c, first);
        first = false;
        mine++;
        __syncthreads();
    }
    pend.k = k; pend.c0 = c0; pend.step = step; pend.n_chunks = n_chunks;
    return mine != 0u;
}

template <int CQ, int REM, int NT>
__global__ void __launch_bounds__(256, (CQ >= 4 ? 2 : EVS_RF_LB)) emb_interact_rf_serve_kernel(const SrvArgs sv) {
    SrvState *st = sv.st;
    const int lane = threadIdx.x & 63;
    const unsigned G = gridDim.x, me = blockIdx.x;
    // the scalar fields of the template (constant address space: scalar loads; the template is not written while the grid runs)
    typedef const __attribute__((address_space(4))) FusedArgs *tmpl4_t;
    const tmpl4_t t4 = reinterpret_cast<tmpl4_t>(reinterpret_cast<uintptr_t>(sv.tmpl));
    FusedArgs la;
    la.F = t4->F; la.d = t4->d; la.itself = t4->itself; la.P = t4->P; la.err = t4->err; la.zeros = t4->zeros;
    la.dummy_i64 = t4->dummy_i64; la.dummy_f32 = t4->dummy_f32; la.tile_per = 16; la.multi_n = 0; la.multi_cpb = 0;
    la.multi_idx_stride = 0; la.multi_off_stride = 0; la.R = nullptr; la.B = 0; la.bag1 = 3;
    bool first = true;     // (block-uniform) this block has not run a chunk yet: the body's one read of the template
    // what the block's wave 0 brought back from its poll: [0..63] descriptor words (leader: up to four new batches; worker: its
    // replica line -- number, stop word, the descriptor of batch `number`), [64] = batches to look at, [65] = leave afterwards
    __shared__ unsigned s_line[66];
    const bool hostpub = sv.front != nullptr;      // (grid-uniform)
    const unsigned *my_line = hostpub ? &sv.front->line[me % (unsigned)kSrvReplicas][0] : &st->pub[me % (unsigned)kSrvReplicas][0];
    // the last batch this block has looked at (host-published lines: its own count, kept in device memory across launches)
    unsigned my = hostpub ? __hip_atomic_load(&st->prog[me & 4095u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : sv.start_seq;
    SrvPending pend{0u, 0u, 1u, 0u};
    bool have_pend = false;          // (block-uniform) chunks run, their stores possibly still on their way, not yet reported
    if (me == 0u && threadIdx.x == 0) __hip_atomic_store(const_cast<unsigned *>(sv.ans) + kSrvSlots * kSrvAnsLine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // alive
    for (;;) {
        if (threadIdx.x < 64 && hostpub) {
            // ---------------- host-published lines: every block's wave 0 at its line (the leader's too) ----------------
            const long long t0 = (long long)wall_clock64();
            unsigned v = 0u, n_new = 0u;
            bool leave_now = false;
            for (;;) {
                v = __hip_atomic_load(my_line + (lane & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                const unsigned g7 = (unsigned)__builtin_amdgcn_readlane((int)v, 7), g15 = (unsigned)__builtin_amdgcn_readlane((int)v, 15);
                const unsigned gn = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
                // the line's number: both guards (a line caught between its two halves' arrival shows nothing new yet)
                const unsigned sq = (g7 == g15 && (int)(g7 - my) > 0) ? g7 : my;
                if (sq != my) { n_new = sq - my; break; }
                if (have_pend) break;     // nothing new, but chunks to report: the load above has returned, so have this wave's stores
                if (gn == sv.gen) { leave_now = true; break; }       // the leader has gone (idle, or asked to): so does this block
                if (me == 0u) {
                    const unsigned stopreq = (unsigned)__builtin_amdgcn_readlane((int)v, 18);
                    if (stopreq != 0u || (long long)wall_clock64() - t0 > sv.idle_ticks) {
                        if (lane < kSrvReplicas) __hip_atomic_store(&sv.front->line[lane][16], sv.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        leave_now = true;
                        break;
                    }
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (lane < 16) s_line[lane] = v;       // the latest descriptor (the look below checks whose it is)
            if (lane == 0) { s_line[64] = n_new; s_line[65] = leave_now ? 1u : 0u; }
        } else if (threadIdx.x < 64) {
            if (me == 0u) {
                // ---------------- the leader: wave 0 at the mailbox ----------------
                const long long t0 = (long long)wall_clock64();
                int n_new = 0;
                unsigned word = 0u;
                bool pend_only = false;
                for (;;) {
                    // the next four ring slots in one load: lane = 16 * (slot in the group) + word
                    const unsigned slot0 = (my + 1u) % (unsigned)kSrvSlots;
                    const unsigned sl = (slot0 + (unsigned)(lane >> 4)) % (unsigned)kSrvSlots;
                    word = __hip_atomic_load(const_cast<unsigned *>(sv.req) + sl * 16u + (unsigned)(lane & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    // (the control word asked for WITH the ring, not behind it: it was a second round trip over the bus per poll)
                    const unsigned stop = __hip_atomic_load(const_cast<unsigned *>(sv.ctl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    // group g holds descriptor my + 1 + g when both its guards say so
                    const unsigned want = my + 1u + (unsigned)(lane >> 4);
                    const unsigned long long gm = __ballot(((lane & 7) == 7) && word == want);   // lanes 7, 15 of every group
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const bool okg = ((gm >> (16 * g + 7)) & 1ull) && ((gm >> (16 * g + 15)) & 1ull);
                        if (okg && n_new == g) n_new = g + 1;    // consecutive ones only
                    }
                    if (n_new > 0) {
                        // the descriptors into the ring in device memory (what a worker reads when a poll brought it more than one
                        // new number), the LAST of them into every replica line behind the line's number (a worker's poll brings the
                        // descriptor with the number: one round trip less per batch) -- all without a wait in between: a reader checks
                        // the guards of what it reads against the number it is after (older: not there yet, read again)
                        if ((lane >> 4) < n_new) __hip_atomic_store(&st->desc[sl].w[lane & 15], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const unsigned wl = (unsigned)__shfl((int)word, 16 * (n_new - 1) + (lane & 15));
#pragma unroll 4
                        for (int r = 0; r < kSrvReplicas; r += 4)
                            __hip_atomic_store(&st->pub[r + (lane >> 4)][2 + (lane & 15)], wl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (lane < kSrvReplicas) __hip_atomic_store(&st->pub[lane][0], my + (unsigned)n_new, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    if (have_pend) { pend_only = true; break; }   // nothing new, but chunks to report: the load above has returned, so have this wave's stores
                    if (stop != 0u || (long long)wall_clock64() - t0 > sv.idle_ticks) break;
                    __builtin_amdgcn_s_sleep(2);
                }
                s_line[lane] = word;
                if (lane == 0) { s_line[64] = (unsigned)n_new; s_line[65] = (n_new == 0 && !pend_only) ? 1u : 0u; }
            } else {
                // ---------------- a worker: the whole replica line in one 32-lane load ----------------
                unsigned v, sq, gn;
                for (;;) {
                    v = __hip_atomic_load(my_line + (lane & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    sq = (unsigned)__builtin_amdgcn_readlane((int)v, 0); gn = (unsigned)__builtin_amdgcn_readlane((int)v, 1);   // (not __shfl: that is the LDS crossbar)
                    if (sq != my || gn == sv.gen) break;
                    if (have_pend) break;     // nothing new, but chunks to report: the load above has returned, so have this wave's stores
                    __builtin_amdgcn_s_sleep((EVS_X_SRV & 4) ? 16 : 1);
                }
                if (lane >= 2 && lane < 18) s_line[lane - 2] = v;
                if (lane == 0) { s_line[64] = sq - my; s_line[65] = gn == sv.gen ? 1u : 0u; }
            }
        }
        // (waves 1..3: nothing to poll for, their stores of the batch before drain meanwhile; wave 0's poll has returned: so have its stores)
        if (have_pend && threadIdx.x >= 64) __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        if (have_pend) { if (threadIdx.x == 0) srv_arrive(sv, pend); have_pend = false; }
        const unsigned n_look = s_line[64], leave = s_line[65];
        for (unsigned g = 0; g < n_look; g++) {
            const unsigned k = my + 1u + g;
            unsigned w[16];
            bool have;
            if ((me == 0u && !hostpub) || g + 1u == n_look) {     // the leader's own copies / the descriptor that came with the line
#pragma unroll
                for (int i = 0; i < 16; i++) w[i] = (unsigned)__builtin_amdgcn_readfirstlane((int)s_line[((me == 0u && !hostpub) ? 16 * g : 0) + i]);
                have = w[7] == k && w[15] == k;
            } else have = false;
            if (!have && (me != 0u || hostpub)) {
                // an earlier batch of a burst (or a line caught between its words and its number): the ring in device memory.  Its
                // guards say which batch the slot holds: k -- take it (both guards); an OLDER number -- the leader's copy is still on its way: read again; a LATER one -- this worker
                // lags, batch k completed without it and the ring has come round: k owed it nothing (a batch is only answered,
                // and its slot only reused, when every chunk of it has arrived).
                const unsigned *dw = hostpub ? sv.front->desc[k % (unsigned)kSrvSlots].w : st->desc[k % (unsigned)kSrvSlots].w;
                for (int tries = 0; tries < (1 << 20); tries++) {
                    // ONE load instruction for the sixteen words (lane = word): a 32-byte sector -- seven words and the guard behind
                    // them -- is sampled at one time, so a guard that holds k vouches for the words in front of it (sixteen separate
                    // loads could take words 0..6 before a write lands and the guard after it)
                    const unsigned dv = hostpub ? __hip_atomic_load(dw + (lane & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                                                : __hip_atomic_load(dw + (lane & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int i = 0; i < 16; i++) w[i] = (unsigned)__builtin_amdgcn_readlane((int)dv, i);
                    if (w[7] == k && w[15] == k) { have = true; break; }     // (the leader writes a descriptor as ONE 64-byte store)
                    if ((int)(w[7] - k) > 0 || (int)(w[15] - k) > 0) break;    // overwritten by a later batch
                }
            }
            if (have) {
                if (have_pend) {   // a second batch inside one look: the first one's stores are waited for in place (rare: a burst)
                    __builtin_amdgcn_s_waitcnt(0x0F70);
                    __syncthreads();
                    if (threadIdx.x == 0) srv_arrive(sv, pend);
                    have_pend = false;
                }
                have_pend = srv_run_batch<CQ, REM, NT>(sv, la, w, k, me, G, first, pend);
            }
        }
        my += n_look;
        __syncthreads();
        if (leave) break;
    }
    if (have_pend) {
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        if (threadIdx.x == 0) srv_arrive(sv, pend);
    }
    if (hostpub && threadIdx.x == 0) __hip_atomic_store(&st->prog[me & 4095u], my, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (me == 0u) {
        // leaving: every worker sees the stop word behind the last number; the host learns how far the grid got
        if (!hostpub && threadIdx.x < kSrvReplicas) __hip_atomic_store(&st->pub[threadIdx.x][1], sv.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x == 0) {
            __hip_atomic_store(const_cast<unsigned *>(sv.ans) + kSrvSlots * kSrvAnsLine + 1, my, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(const_cast<unsigned *>(sv.ans) + kSrvSlots * kSrvAnsLine, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

}  // namespace evs

struct evs_rf_server {
    evs::FusedArgs tmpl{};
    evs::FusedArgs *tmpl_dev = nullptr;
    evs::SrvState *st_dev = nullptr;
    evs::SrvFront *front = nullptr;  // host-published front end: fine-grained device memory the host writes through the aperture (same address on both sides), or nullptr
    unsigned *mbox = nullptr, *mbox_dev = nullptr;   // host block: request ring | control line | answer ring + status line
    hipStream_t stream = nullptr;
    unsigned posted = 0;            // the last sequence number posted
    unsigned gen = 0;               // launches of the grid so far
    unsigned next_first = 0;        // the worker the next batch's first chunk goes to
    unsigned slot_words[evs::kSrvSlots] = {};   // answer words the slot's batch in flight is answered through (min(chunks, 32))
    int n_blocks = 0, T = 0, d = 0;
    long long idle_ticks = 0;
    bool nt2 = false;
};
namespace {
constexpr size_t kSrvReqWords = (size_t)evs::kSrvSlots * 16, kSrvCtlWords = 32, kSrvAnsWords = (size_t)evs::kSrvSlots * evs::kSrvAnsLine + 16;
constexpr size_t kSrvStatus = (size_t)evs::kSrvSlots * evs::kSrvAnsLine;   // the status line behind the answer ring
// batch k of ring slot `slot` has been answered: every sub-counter's completer has written its number
inline bool srv_answered(evs_rf_server *s, unsigned slot, unsigned k) {
    volatile unsigned *a = s->mbox + kSrvReqWords + kSrvCtlWords + (size_t)slot * evs::kSrvAnsLine;
    const unsigned n = s->slot_words[slot] ? s->slot_words[slot] : 1u;
    for (unsigned r = 0; r < n; r++) if (a[r] != k) return false;
    return true;
}
inline volatile unsigned *srv_req(evs_rf_server *s) { return s->mbox; }
inline volatile unsigned *srv_ctl(evs_rf_server *s) { return s->mbox + kSrvReqWords; }
inline volatile unsigned *srv_ans(evs_rf_server *s) { return s->mbox + kSrvReqWords + kSrvCtlWords; }
void srv_launch(evs_rf_server *s) {
    using namespace evs;
    SrvArgs a;
    a.tmpl = s->tmpl_dev; a.st = s->st_dev; a.front = s->front;
    a.req = s->mbox_dev; a.ctl = s->mbox_dev + kSrvReqWords; a.ans = s->mbox_dev + kSrvReqWords + kSrvCtlWords;
    a.start_seq = srv_ans(s)[kSrvStatus + 1];   // how far the last run got (0 at first)
    a.idle_ticks = s->idle_ticks;
    a.gen = ++s->gen;
    const dim3 grid((unsigned)s->n_blocks), block(256);
    const bool nt2 = s->nt2;
    switch (s->d) {
    case 16: if (nt2) hipLaunchKernelGGL((emb_interact_rf_serve_kernel<1, 0, 2>), grid, block, 0, s->stream, a); else hipLaunchKernelGGL((emb_interact_rf_serve_kernel<1, 0, 1>), grid, block, 0, s->stream, a); break;
    case 32: if (nt2) hipLaunchKernelGGL((emb_interact_rf_serve_kernel<2, 0, 2>), grid, block, 0, s->stream, a); else hipLaunchKernelGGL((emb_interact_rf_serve_kernel<2, 0, 1>), grid, block, 0, s->stream, a); break;
    case 36: if (nt2) hipLaunchKernelGGL((emb_interact_rf_serve_kernel<2, 1, 2>), grid, block, 0, s->stream, a); else hipLaunchKernelGGL((emb_interact_rf_serve_kernel<2, 1, 1>), grid, block, 0, s->stream, a); break;
    default: if (nt2) hipLaunchKernelGGL((emb_interact_rf_serve_kernel<4, 0, 2>), grid, block, 0, s->stream, a); else hipLaunchKernelGGL((emb_interact_rf_serve_kernel<4, 0, 1>), grid, block, 0, s->stream, a); break;
    }
}
// Can this process STORE to p?  The attribute says the part has a large BAR; whether this allocation is mapped writable here is
// asked of the kernel, not found out by a fault: read(2) from /dev/zero INTO p copies four zero bytes to it, or fails with
// EFAULT -- no signal either way (the block has just been zeroed: nothing changes).
bool host_can_store(void *p) {
    const int fd = open("/dev/zero", O_RDONLY);
    if (fd < 0) return false;
    const ssize_t n = read(fd, p, 4);
    (void)close(fd);
    return n == 4;
}
// send the grid home (it leaves by itself when idle) and wait until it has gone
int srv_pause(evs_rf_server *s) {
    if (!s->stream) return EVS_OK;
    if (hipStreamQuery(s->stream) == hipSuccess) return EVS_OK;
    (void)hipGetLastError();
    // (the host-published front end: the request is a word of line 0, written through the aperture and never read back)
    volatile unsigned *stop = s->front ? reinterpret_cast<volatile unsigned *>(&s->front->line[0][18]) : srv_ctl(s);
    *stop = 1u;
    __builtin_ia32_sfence();
    const hipError_t e = hipStreamSynchronize(s->stream);
    *stop = 0u;
    __builtin_ia32_sfence();
    return e == hipSuccess ? EVS_OK : EVS_EHIP;
}
}  // namespace

extern "C" int evs_emb_interact_serve_start(evs_rf_server **out, int T, int d, const void *const *tables, const int64_t *n_rows,
                                            int itself, int n_blocks, int64_t idle_us) {
    using namespace evs;
    EVS_REQUIRE(out && tables && n_rows, "evs_emb_interact_serve_start: NULL argument");
    EVS_REQUIRE(T >= 1 && T + 1 <= kTileMaxF, "evs_emb_interact_serve_start: T=%d (the rows-in-registers kernel takes x + at most %d tables)", T, kTileMaxF - 1);
    EVS_REQUIRE(d == 16 || d == 32 || d == 36 || d == 64, "evs_emb_interact_serve_start: d=%d (16, 32, 36 or 64; fp32 tables)", d);
    EVS_REQUIRE(idle_us >= 1 && n_blocks >= 0, "evs_emb_interact_serve_start: bad argument");
    for (int t = 0; t < T; t++)
        EVS_REQUIRE(n_rows[t] >= 0 && n_rows[t] < (1ll << 31) && (n_rows[t] == 0 || (tables[t] && reinterpret_cast<uintptr_t>(tables[t]) % 16 == 0)),
                    "evs_emb_interact_serve_start: table %d (16-byte aligned, fewer than 2^31 rows)", t);
    evs_rf_server *s = new evs_rf_server();
    s->T = T; s->d = d; s->nt2 = T + 1 > 16;
    const int per_cu = d == 64 ? 2 : EVS_RF_LB;
    s->n_blocks = n_blocks > 0 ? n_blocks : kNumCu * per_cu;
    if (s->n_blocks < 2) s->n_blocks = 2;
    s->idle_ticks = idle_us * 100;   // wall_clock64(): 100 MHz
    FusedArgs &a = s->tmpl;
    const int F = T + 1;
    for (int f = 0; f < EVS_MAX_FEATURES; f++) {
        a.src[f] = nullptr; a.stride[f] = 0; a.indices[f] = nullptr; a.offsets[f] = nullptr; a.nnz[f] = 0;
        a.n_rows[f] = 0; a.row_w[f] = nullptr; a.off_len[f] = 0;
    }
    a.zeros = zero_page();
    a.err = index_error_flag();
    if (!a.zeros || !a.err) { delete s; return EVS_EHIP; }
    const int64_t *any_i64 = reinterpret_cast<const int64_t *>(a.zeros);
    for (int t = 0; t < T; t++) {
        a.src[t + 1] = n_rows[t] == 0 ? a.zeros : tables[t];
        a.indices[t + 1] = any_i64;       // (non-NULL marks a table: the kernel reads the descriptor's arrays)
        a.offsets[t + 1] = any_i64;
        a.n_rows[t + 1] = n_rows[t];
    }
    a.src[0] = a.zeros;
    a.B = 0; a.F = F; a.d = d; a.itself = itself ? 1 : 0; a.P = itself ? F * (F + 1) / 2 : F * (F - 1) / 2;
    a.dummy_i64 = any_i64; a.dummy_f32 = reinterpret_cast<const float *>(a.zeros); a.bag1 = 3; a.enc_lds = 0; a.opt_flag = nullptr; a.opt_id = 0;
    a.tile_per = 16; a.row_ids = nullptr; a.arena = nullptr; a.w1p = nullptr; a.b1 = nullptr; a.z1 = nullptr; a.n1 = 0; a.kp = 0; a.relu = 0; a.write_r = 1;
    a.zero_codes = nullptr; a.multi_n = 0; a.multi_cpb = 0; a.multi_idx_stride = 0; a.multi_off_stride = 0; a.R = nullptr;
    auto fail = [&](int rc) { (void)evs_emb_interact_serve_destroy(s); return rc; };
    if (hipMalloc(reinterpret_cast<void **>(&s->tmpl_dev), sizeof(FusedArgs)) != hipSuccess) return fail(EVS_ENOMEM);
    if (hipMalloc(reinterpret_cast<void **>(&s->st_dev), sizeof(SrvState)) != hipSuccess) return fail(EVS_ENOMEM);
    if (hipMemcpy(s->tmpl_dev, &a, sizeof(FusedArgs), hipMemcpyHostToDevice) != hipSuccess) return fail(EVS_EHIP);
    if (hipMemset(s->st_dev, 0, sizeof(SrvState)) != hipSuccess) return fail(EVS_EHIP);
    {   // The host-published front end where the host can address device memory (EVS_SERVE_PUBLISH=leader keeps the leader's
        // mailbox: developer A/B, and what parts without a large BAR run)
        const char *how = getenv("EVS_SERVE_PUBLISH");
        int dev_id = 0, large_bar = 0;
        if (!(how && !strcmp(how, "leader")) && s->n_blocks <= 4096 && hipGetDevice(&dev_id) == hipSuccess &&
            hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, dev_id) == hipSuccess && large_bar) {
            void *p = nullptr;
            if (hipExtMallocWithFlags(&p, sizeof(SrvFront), hipDeviceMallocFinegrained) == hipSuccess && hipMemset(p, 0, sizeof(SrvFront)) == hipSuccess &&
                hipDeviceSynchronize() == hipSuccess && host_can_store(p))
                s->front = reinterpret_cast<SrvFront *>(p);
            else { if (p) (void)hipFree(p); (void)hipGetLastError(); }
        } else (void)hipGetLastError();
    }
    const size_t words = kSrvReqWords + kSrvCtlWords + kSrvAnsWords;
    if (hipHostMalloc(reinterpret_cast<void **>(&s->mbox), words * 4, hipHostMallocMapped) != hipSuccess) return fail(EVS_ENOMEM);
    memset(s->mbox, 0, words * 4);
    if (hipHostGetDevicePointer(reinterpret_cast<void **>(&s->mbox_dev), s->mbox, 0) != hipSuccess) return fail(EVS_EHIP);
    int prio_lo = 0, prio_hi = 0;
    if (hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess) return fail(EVS_EHIP);
    if (hipStreamCreateWithPriority(&s->stream, hipStreamNonBlocking, prio_hi) != hipSuccess) return fail(EVS_EHIP);
    if (hipDeviceSynchronize() != hipSuccess) return fail(EVS_EHIP);   // (the template and the state are in place before the grid's first read)
    *out = s;
    return EVS_OK;
}

// Post one batch: R = interact_features(x, apply_emb(lS_o, lS_i, tables)), lS_o given and checked per 16-sample block exactly as
// evs_emb_interact_dot_stacked does.  Returns at once; *ticket names the batch for evs_emb_interact_serve_wait.
extern "C" int evs_emb_interact_serve_post(evs_rf_server *s, int64_t B, const float *x, int64_t x_stride, const int64_t *indices_base,
                                           int64_t indices_row_stride, const int64_t *offsets_base, int64_t offsets_row_stride,
                                           float *R, uint64_t *ticket) {
    using namespace evs;
    EVS_REQUIRE(s && x && indices_base && offsets_base && R && ticket, "evs_emb_interact_serve_post: NULL argument");
    EVS_REQUIRE(B >= 1 && B < (1ll << 31), "evs_emb_interact_serve_post: B=%lld", (long long)B);
    EVS_REQUIRE(x_stride >= 0 && x_stride % 4 == 0 && x_stride < (1ll << 31) && reinterpret_cast<uintptr_t>(x) % 16 == 0,
                "evs_emb_interact_serve_post: x must be 16-byte aligned with a row stride that is a multiple of 4 floats");
    EVS_REQUIRE(indices_row_stride >= 0 && indices_row_stride < (1ll << 31) && offsets_row_stride >= 0 && offsets_row_stride < (1ll << 31),
                "evs_emb_interact_serve_post: row strides of the (T, B) arrays must fit 31 bits");
    const unsigned k = s->posted + 1u;
    const unsigned slot = k % (unsigned)kSrvSlots;
    volatile unsigned *ans = srv_ans(s), *req = srv_req(s) + (size_t)slot * 16;
    // the ring holds kSrvSlots batches in flight: the slot's previous user (k - kSrvSlots) must have been answered
    if (k > (unsigned)kSrvSlots) {
        const unsigned prev = k - (unsigned)kSrvSlots;
        uint64_t t = prev;
        if (!srv_answered(s, slot, prev)) { const int rc = evs_emb_interact_serve_wait(s, t); if (rc) return rc; }
    }
    {   const uint64_t n_chunks = (uint64_t)((B + 15) / 16);
        s->slot_words[slot] = (unsigned)(n_chunks < kSrvSub ? n_chunks : kSrvSub); }
    const unsigned long long px = (unsigned long long)reinterpret_cast<uintptr_t>(x), pi = (unsigned long long)reinterpret_cast<uintptr_t>(indices_base),
                             po = (unsigned long long)reinterpret_cast<uintptr_t>(offsets_base), pr = (unsigned long long)reinterpret_cast<uintptr_t>(R);
    if (s->front) {
        // host-published: the descriptor into its ring slot in DEVICE memory, then into the first half of every replica line --
        // 64-byte lines written whole through the write-combining aperture (one bus write each), a store fence between the ring
        // and the lines (a block that finds a number in its line reads the ring for the batches before it), nothing read back
        const unsigned dsc[16] = {(unsigned)px, (unsigned)(px >> 32), (unsigned)pi, (unsigned)(pi >> 32), (unsigned)po, (unsigned)(po >> 32), (unsigned)B, k,
                                  (unsigned)pr, (unsigned)(pr >> 32), (unsigned)x_stride, s->next_first, (unsigned)indices_row_stride, (unsigned)offsets_row_stride, 0u, k};
        volatile unsigned *ring = reinterpret_cast<volatile unsigned *>(s->front->desc[slot].w);
        for (int i = 0; i < 16; i++) ring[i] = dsc[i];
        __builtin_ia32_sfence();
        for (int r = 0; r < kSrvReplicas; r++) {
            volatile unsigned *l = reinterpret_cast<volatile unsigned *>(s->front->line[r]);
            for (int i = 0; i < 16; i++) l[i] = dsc[i];
        }
        __builtin_ia32_sfence();
    } else {
    req[0] = (unsigned)px; req[1] = (unsigned)(px >> 32); req[2] = (unsigned)pi; req[3] = (unsigned)(pi >> 32);
    req[4] = (unsigned)po; req[5] = (unsigned)(po >> 32); req[6] = (unsigned)B;
    req[8] = (unsigned)pr; req[9] = (unsigned)(pr >> 32); req[10] = (unsigned)x_stride; req[11] = s->next_first;
    req[12] = (unsigned)indices_row_stride; req[13] = (unsigned)offsets_row_stride; req[14] = 0u;
    __atomic_thread_fence(__ATOMIC_RELEASE);
    req[7] = k; req[15] = k;
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    }
    s->posted = k;
    s->next_first = (unsigned)(((uint64_t)s->next_first + (uint64_t)((B + 15) / 16)) % (uint64_t)(1u << 30));
    *ticket = k;
    if (ans[kSrvStatus] == 0u) {   // nobody there (never started, or gone home idle): start the grid
        const hipError_t q = hipStreamQuery(s->stream);
        if (q == hipSuccess) { srv_launch(s); if (hipGetLastError() != hipSuccess) { set_error("evs_emb_interact_serve_post: the grid could not be started"); return EVS_EHIP; } }
        else if (q != hipErrorNotReady) { (void)hipGetLastError(); return EVS_EHIP; }
        else (void)hipGetLastError();
    }
    return EVS_OK;
}

// Spin until batch `ticket` has been answered (R complete: anything started afterwards may read it)
extern "C" int evs_emb_interact_serve_wait(evs_rf_server *s, uint64_t ticket) {
    using namespace evs;
    EVS_REQUIRE(s && ticket >= 1 && ticket <= s->posted, "evs_emb_interact_serve_wait: no such batch");
    // (only the last kSrvSlots batches have an answer word of their own: an older one was answered when its slot was re-posted)
    if ((uint64_t)s->posted - ticket >= (uint64_t)kSrvSlots) return EVS_OK;
    volatile unsigned *ans = srv_ans(s);
    const unsigned slot = (unsigned)(ticket % (unsigned)kSrvSlots);
    long long spins = 0;
    while (!srv_answered(s, slot, (unsigned)ticket)) {
        if ((++spins & 255) == 0 && ans[kSrvStatus] == 0u) {   // the grid has left (idle) with this batch still in the ring: start it again
            const hipError_t q = hipStreamQuery(s->stream);
            if (q == hipSuccess) { if (srv_answered(s, slot, (unsigned)ticket)) break; srv_launch(s); if (hipGetLastError() != hipSuccess) return EVS_EHIP; }
            else if (q != hipErrorNotReady) { (void)hipGetLastError(); return EVS_EHIP; }
            else (void)hipGetLastError();
        }
        if (spins > (1ll << 31)) { (void)srv_pause(s); set_error("evs_emb_interact_serve_wait: the grid did not answer"); return EVS_EHIP; }
        __builtin_ia32_pause();
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return EVS_OK;
}

// 1: the host publishes descriptors through the PCIe aperture (large BAR); 0: the leader reads a mailbox in host memory
extern "C" int evs_emb_interact_serve_mode(evs_rf_server *s) { return s && s->front ? 1 : 0; }

extern "C" int evs_emb_interact_serve_stop(evs_rf_server *s) {
    EVS_REQUIRE(s, "evs_emb_interact_serve_stop: NULL server");
    if (s->posted) { const int rc = evs_emb_interact_serve_wait(s, s->posted); if (rc) return rc; }
    return srv_pause(s);
}

extern "C" int evs_emb_interact_serve_destroy(evs_rf_server *s) {
    if (!s) return EVS_OK;
    if (s->mbox && s->stream) (void)srv_pause(s);
    if (s->stream) { (void)hipStreamSynchronize(s->stream); (void)hipStreamDestroy(s->stream); }
    if (s->mbox) (void)hipHostFree(s->mbox);
    if (s->tmpl_dev) (void)hipFree(s->tmpl_dev);
    if (s->st_dev) (void)hipFree(s->st_dev);
    if (s->front) (void)hipFree(s->front);
    delete s;
    return EVS_OK;
}

#ifdef EVS_X_PT
extern "C" __attribute__((visibility("default"))) int evs_x_pt(unsigned long long *out, int reset) {   // out: 16 (summed over the blocks)
    static unsigned long long h[1024 * 16], z[1024 * 16];
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(evs::g_pt), sizeof h) != hipSuccess) return -2;
    if (out) for (int k = 0; k < 16; k++) { out[k] = 0; for (int b = 0; b < 1024; b++) out[k] += h[b * 16 + k]; }
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(evs::g_pt), z, sizeof z) != hipSuccess) return -3;
    return 0;
}
#endif
#ifdef EVS_X_LOG
// developer tool (tools/dbg_inline.py): the event log of the folded update -- (type, word address, old word, new word)
extern "C" __attribute__((visibility("default"))) long long evs_x_log_fetch(unsigned long long *out, long long max_events, int reset) {
    unsigned n = 0;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(evs::g_xlog_n), 4) != hipSuccess) return -2;
    long long m = n < (1u << 18) ? n : (1u << 18);
    if (m > max_events) m = max_events;
    if (out && m && hipMemcpyFromSymbol(out, HIP_SYMBOL(evs::g_xlog), (size_t)m * 32) != hipSuccess) return -3;
    if (reset) { n = 0; if (hipMemcpyToSymbol(HIP_SYMBOL(evs::g_xlog_n), &n, 4) != hipSuccess) return -4; }
    return m;
}
#endif
