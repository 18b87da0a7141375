// Shared between the fused gather + interaction kernels (evs_fused.hip, evs_fused_rf.hip).
#pragma once
#include "evs_common.h"
#include "evs_hash.h"

namespace evs {

constexpr int kMultiMax = 8;   // batches per multi launch (more: several launches)

struct FusedArgs {
    const void *src[EVS_MAX_FEATURES];        // dense: fp32 rows; indirect: table bytes
    int64_t stride[EVS_MAX_FEATURES];         // dense: floats between samples
    const int64_t *indices[EVS_MAX_FEATURES]; // NULL => dense
    const int64_t *offsets[EVS_MAX_FEATURES];
    int64_t nnz[EVS_MAX_FEATURES];
    int64_t n_rows[EVS_MAX_FEATURES];
    int64_t off_len[EVS_MAX_FEATURES];
    const float *row_w[EVS_MAX_FEATURES];
    float *R;
    int64_t B;
    int F, d, itself, P;
    int *err;
    const int64_t *dummy_i64;  // any readable int64 (lanes with nothing to fetch read it)
    const float *dummy_f32;
    const void *zeros;         // >= 1 KiB of zero bytes (idle lanes / empty bags read rows from it)
    int bag1;                  // 1: every indirect feature has one index per bag, no offsets array;
                               // 2: offsets ARE given and the launch bets they are arange (see opt_flag);
                               // 3: offsets given, whole batch: the index-tile loop checks them chunk by chunk itself
                               // 4: (reduced precision) offsets given, whole batch: evs_fused_rfq checks them per block and pools the
                               //    blocks that fail in its own slow loop
    int *opt_flag; int opt_id; // optimistic launch triple: offsets_arange_kernel writes opt_id here when an offsets
                               // array is not arange; then the bag-1 loop runs if it is absent, the general loop if present
    int enc_lds;               // reduced precision: feature 0 is x, every other feature a 4-byte aligned table
    int tile_per;              // index-tile kernel: samples per block (block i owns [i * tile_per, (i + 1) * tile_per))
    // first top-MLP layer fused behind the interaction (evs_fused_rf.hip, MLP variant): Z1 = act(R W1^T + b1)
    const float *w1p;          // W1 zero-padded to (n1 rounded up to 16) x kp floats, kp = K rounded up to 16, K = d + P
    const float *b1;           // n1 biases
    float *z1;                 // (B, n1) out
    int n1, kp, relu, write_r; // write_r: also store R (B, d + P)
    // cache consumer (evs_fused_rf.hip, IDS variant): features 1..F-1 are given as ONE (B, F-1) int32 table of row ids --
    // bit 30 set: row (id & 0x3fffffff) of `arena`; clear: row id of the feature's own table src[f]; -1: the zero row
    const int *row_ids;
    const void *arena;
    // ... or the kernel probes the cache itself (PROBE variant): the (B, F-1) request rows in, hit flags, miss lists and
    // hit statistics out -- cache_batch_probe_gather_kernel folded into the head of this launch
    ProbeArgs probe;
    // evs_fused_rfq.hip: 1 KiB of the code that decodes to 0.0f (zero_code_page) -- the "row" of an absent table row
    const void *zero_codes;
    // evs_fused_rf.hip, K batches in ONE launch (evs_emb_interact_dot_stacked_multi): block i works on chunk i % multi_cpb of
    // batch i / multi_cpb -- feature 0 (x), the (T, B) index / offsets arrays and R of that batch come from these tables,
    // everything else (tables, shapes) is shared.  multi_n == 0: a plain launch.
    // evs_fused_rf.hip, the stacked call (x + T tables behind ONE (T, B) index array and, lS_o given, one (T, B) offsets array: what
    // evs_emb_interact_dot_stacked passes): feature f >= 1 reads indices at stk_idx + (f - 1) * stk_idx_stride -- arithmetic instead
    // of a read of indices[f] / offsets[f] out of the kernel arguments in front of the first index load.  stk == 0: not that form.
    int stk = 0;
    const int64_t *stk_idx = nullptr, *stk_off = nullptr;
    int64_t stk_idx_stride = 0, stk_off_stride = 0, stk_off_len = 0;
    int multi_n = 0, multi_cpb = 0;
    int64_t multi_idx_stride, multi_off_stride;   // elements between the rows of two tables in a batch's (T, B) arrays
    const float *multi_x[kMultiMax];
    const int64_t *multi_idx[kMultiMax];
    const int64_t *multi_off[kMultiMax];
    float *multi_R[kMultiMax];
};


constexpr int kTileMaxF = 28;   // index-tile launches: x + at most 27 tables

// evs_fused_rf.hip: the rows-in-flight-in-registers form of the bag-1 index-tile loop (fp32 tables, d in {16, 32, 36},
// F <= kTileMaxF); returns false when it has no kernel for the shape (the caller then uses the LDS-DMA loop)
bool launch_rf(const FusedArgs &a, hipStream_t st);
// ... with lS_o given (bag1 == 3: whole batches): the offsets of each block's 16 samples checked in the kernel, the blocks
// that fail pooled in its own slow loop
bool launch_rf_check(const FusedArgs &a, hipStream_t st);
// K batches in one launch (FusedArgs::multi_*): bag1 == 1 (declared one index per bag) or 3 (offsets given, checked per block)
bool rf_multi_supported(int64_t B, int F, int d);
bool launch_rf_multi(const FusedArgs &a, hipStream_t st);
// the same kernel with the first top-MLP layer behind it (any batch size); false = no kernel for the shape
bool launch_rf_mlp(const FusedArgs &a, hipStream_t st);
// the same kernel reading a (B, F-1) table of 32-bit row ids (the cache tier's consumer); rf_ids_supported: is there a
// kernel for this shape (the probe kernel has to know which table to write before the consumer is launched)
bool rf_ids_supported(int64_t B, int F, int d);
bool launch_rf_ids(const FusedArgs &a, hipStream_t st);
// ... probing the cache itself (FusedArgs::probe filled in); same shapes as launch_rf_ids; one block per 16 samples
bool launch_rf_probe(const FusedArgs &a, hipStream_t st);

// evs_fused_rfq.hip: reduced-precision tables (codec 16 / 8 / 4), encoded rows in flight in registers in the MFMA operand
// mapping; bag1 == 1 (no offsets) or 2 (the one-index-per-bag leg of the optimistic triple); false = no kernel for the shape
bool launch_rfq(const FusedArgs &a, int codec, hipStream_t st);
bool rfq_supported(const FusedArgs &a, int codec);
bool rfq_probe_supported(int64_t B, int F, int d, int codec);      // evs_fused_rfq.hip: the tier probe folded in (set-associative, 8 ways)
bool launch_rfq_probe(const FusedArgs &a, int codec, hipStream_t st);

}  // namespace evs
