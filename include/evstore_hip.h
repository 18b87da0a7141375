/* evstore_hip.h -- C ABI of libevstore_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the DLRM embedding-lookup + EvLFU cache + feature
 * interaction hot path of ucare-uchicago/ev-store-dlrm.  Plain pointers and
 * sizes only; every pointer marked "device" is HBM memory of the current HIP
 * device, every `stream` is a hipStream_t passed as void* (NULL = default
 * stream).  All entry points return 0 on success and a negative EVS_E* code
 * on failure; evs_last_error() returns a thread-local message.
 *
 * Each group cites the reference interface it replaces (paths relative to the
 * reference tree).
 */
#ifndef EVSTORE_HIP_H
#define EVSTORE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EVS_ABI_VERSION 1
#define EVS_API __attribute__((visibility("default")))

/* error codes */
#define EVS_OK 0
#define EVS_EINVAL (-1)    /* bad argument (shape, alignment, unsupported codec ...) */
#define EVS_EHIP (-2)      /* a HIP runtime call failed */
#define EVS_EINDEX (-3)    /* an index was out of range (see evs_check_index_errors) */
#define EVS_ENOMEM (-4)
#define EVS_ESTATE (-5)    /* call sequence error (e.g. cache not initialised) */
#define EVS_EIO (-6)       /* table file missing / short read */

/* row codecs = the reference's on-disk precisions (SURVEY 8(b) on-disk contracts):
 * 32: raw little-endian fp32                    (script/convert_ev_to_binary.py:58-69)
 * 16: custom ushort affine code                 (mixed_precs_caching/evlfu_16.cpp:332-356)
 *  8: uint8 affine  f=((float)u/254)*2-1        (mixed_precs_caching/evlfu_8.cpp:370-378)
 *  4: 15-entry LUT, two codes per byte, hi first (mixed_precs_caching/evlfu_4.cpp:319-341) */
#define EVS_CODEC_F32 32
#define EVS_CODEC_U16 16
#define EVS_CODEC_U8 8
#define EVS_CODEC_U4 4

#define EVS_MAX_TABLES_PER_LAUNCH 32
#define EVS_MAX_FEATURES 32

EVS_API int evs_abi_version(void);
EVS_API const char *evs_last_error(void);

/* ---------------------------------------------------------------------------
 * a1: DLRM_Net.apply_emb  (dlrm_s_pytorch.py:407-461)
 *     = T x nn.EmbeddingBag(n_k, d, mode="sum") (dlrm_s_pytorch.py:276) in ONE launch.
 *
 * For every table k and bag b:  out[k][b][:] = sum_{j in bag} decode(W_k[idx_k[j]]) (* w_k[idx_k[j]])
 * bag b of table k = indices[k][ offsets[k][b] .. offsets[k][b+1] ) and the last bag
 * runs to nnz[k] (start offsets only, exactly nn.EmbeddingBag's convention).
 * Summation is in index order, fp32, multiply and add unfused (bit-exact with
 * oracle/evstore_oracle.c:orc_embedding_bag_sum).
 *
 * The four pointer arrays and n_rows/nnz are HOST arrays of length T whose
 * elements are device pointers / sizes.  tables[k]: rows of d elements in the
 * given codec, row-major, row r at byte r*d*codec/8 -- i.e. the bytes of the
 * reference's ev-table-{k+1}.bin, 16-byte aligned.  row_weights may be NULL, or
 * hold NULL entries: row_weights[k] is the per-ROW weight vector v_W_l[k]
 * (dlrm_s_pytorch.py:426 gathers it with the indices).
 * Output element (k,b,c) is written to out[k*out_table_stride + b*out_bag_stride + c];
 * strides are in floats and must be multiples of 4, out 16-byte aligned.
 *   list-of-(B,d) layout:    table_stride = B*d, bag_stride = d
 *   (B,F,d) interaction tile: out = T + d,  table_stride = d, bag_stride = F*d
 * Out-of-range indices contribute nothing and raise a sticky device flag read
 * by evs_check_index_errors() (nn.EmbeddingBag raises; a kernel cannot).
 * offsets == NULL (or every offsets[k] == NULL) states that each bag holds ONE index
 * (offsets = arange(B), what every EVStore script feeds): bag b of table k reads
 * indices[k][b], nnz[k] >= B; the offsets loads in front of the index load go away.
 * ------------------------------------------------------------------------- */
EVS_API int evs_embedding_bag_sum(int T, int64_t B, int d, int codec,
                          const void *const *tables, const int64_t *n_rows,
                          const int64_t *const *indices, const int64_t *const *offsets,
                          const int64_t *nnz, const float *const *row_weights,
                          float *out, int64_t out_table_stride, int64_t out_bag_stride,
                          void *stream);

/* The pool step of the table-sharded forward (dlrm_s_pytorch.py:543-570 distributed_forward: every rank pools ITS
 * tables for the FULL batch, then one all-to-all): evs_embedding_bag_sum with two extensions.
 *   row-split tables (row_lo / row_total, HOST arrays of T, both NULL = whole tables): tables[k] holds the global rows
 *     [row_lo[k], row_lo[k] + n_rows[k]) of a table of row_total[k] rows.  An index inside the table but outside that
 *     range belongs to another rank: it contributes nothing, silently -- the bag's sum over THIS rank's rows is written
 *     (zeros when it has none); only an index outside [0, row_total) raises the sticky flag.
 *   peer-major output (out_peer_stride / bags_per_peer; bags_per_peer <= 0 or >= B = the plain layout): element (k, b, c)
 *     goes to out[(b / bags_per_peer) * out_peer_stride + k * out_table_stride + (b % bags_per_peer) * out_bag_stride + c]
 *     -- the block of the all_to_all_single send buffer that goes to rank b / bags_per_peer (extend_distributed.py:
 *     389-426 builds that buffer with torch.cat; here the gather writes it in place).  Strides in floats, % 4 == 0. */
EVS_API int evs_embedding_bag_sum_sharded(int T, int64_t B, int d, int codec,
                          const void *const *tables, const int64_t *n_rows,
                          const int64_t *row_lo, const int64_t *row_total,
                          const int64_t *const *indices, const int64_t *const *offsets,
                          const int64_t *nnz, const float *const *row_weights,
                          float *out, int64_t out_table_stride, int64_t out_bag_stride,
                          int64_t out_peer_stride, int64_t bags_per_peer, void *stream);

/* ---------------------------------------------------------------------------
 * a15 without a collective call: the exchange step of the table-sharded forward (dlrm_s_pytorch.py:543-570,
 * extend_distributed.py:389-426 All2All_Req, :444-465 All2All_Wait) as direct writes over xGMI.
 *   evs_embedding_bag_sum_p2p   evs_embedding_bag_sum_sharded whose peer-major blocks go STRAIGHT into the peers' receive
 *                               buffers: peer_delta (DEVICE array, one entry per peer, in floats, multiples of 4) is added to
 *                               the local-layout address of peer q's block, i.e. peer_delta[q] = (address of this rank's
 *                               block inside peer q's IPC-mapped receive buffer) - (out + q * out_peer_stride).
 *   evs_p2p_alloc / _free       receive buffers and flag words: fine-grained device memory (a peer's writes are visible to
 *                               this device's loads without a cache flush), zeroed.
 *   evs_p2p_ipc_export / _open / _close   the 64-byte IPC handle of such a buffer, and its mapping in another process
 *                               (hipIpcGetMemHandle / hipIpcOpenMemHandle: the handles cross the process group once).
 *   evs_p2p_sync                one tiny launch: release-store sig_value (system scope) into n_sig words -- flag words in the
 *                               peers' blocks: "use k of your slot holds my block" / "I am through with use k of it" -- then
 *                               wait (acquire loads, bounded: a timeout raises the sticky flag evs_check_index_errors reads,
 *                               EVS_ESTATE) until n_wait words of this rank's own block have reached wait_value.  Host arrays
 *                               of device pointers, at most 64 each.
 * The layout contract is the collective's (what all_to_all_single would have delivered, block for block); sharded.py's
 * exchange_mode = "p2p" drives it, tests/test_p2p_exchange.py pins it against the RCCL / gloo paths.
 * ------------------------------------------------------------------------- */
EVS_API int evs_embedding_bag_sum_p2p(int T, int64_t B, int d, int codec,
                          const void *const *tables, const int64_t *n_rows,
                          const int64_t *row_lo, const int64_t *row_total,
                          const int64_t *const *indices, const int64_t *const *offsets,
                          const int64_t *nnz, const float *const *row_weights,
                          float *out, int64_t out_table_stride, int64_t out_bag_stride,
                          int64_t out_peer_stride, int64_t bags_per_peer, const int64_t *peer_delta, void *stream);
EVS_API int evs_p2p_alloc(void **out, int64_t bytes);
EVS_API int evs_p2p_free(void *p);
EVS_API int evs_p2p_ipc_export(void *p, void *handle64);
EVS_API int evs_p2p_ipc_open(const void *handle64, void **out);
EVS_API int evs_p2p_ipc_close(void *p);
EVS_API int evs_p2p_sync(int n_sig, uint32_t *const *sig, uint32_t sig_value, int n_wait, const uint32_t *const *wait,
                         uint32_t wait_value, void *stream);

/* The loader's collate on the device -- collate_wrapper_criteo_offset, dlrm_data_pytorch.py:397-410 -- from the raw batch as
 * CriteoDataset.__getitem__ yields it (:372-395): x_int (B, n_dense) int32 counts and x_cat (B, T) int32 ids in device
 * memory, sample s at x_int + s * x_int_stride / x_cat + s * x_cat_stride (strides in elements: n_dense / T for separate
 * row-major arrays) -> X (B, n_dense) fp32 = log(x_int + 1) (fp32 add, then logf: within 1-2 ulp of torch.log on the host),
 * lS_i (T, B) int64 = x_cat transposed, lS_o (T, B) int64 = arange(B) per table (lS_o NULL: not written -- the fused launch
 * can be told one_index_per_bag instead).  156 bytes per sample cross the bus instead of 468 (T = 26).
 * The Terabyte binary loader's batches (script/data_loader_terabyte.py:196-236 CriteoBinDataset, :68-87 _transform_features)
 * are one (B, 40) int32 block of the file -- label, 13 counts, 26 ids per record: both strides 40, x_int = block + 1,
 * x_cat = block + 14 -- with ids taken modulo max_ind_range when it is > 0 (:71-72).  1 <= T <= 64. */
EVS_API int evs_collate_criteo_offset(int64_t B, int n_dense, int T, const int32_t *x_int, int64_t x_int_stride, const int32_t *x_cat,
                                      int64_t x_cat_stride, int max_ind_range, float *X, int64_t *lS_o, int64_t *lS_i, void *stream);

/* a16 (dlrm_wrap's per-batch H2D, dlrm_s_pytorch.py:131-147) as a two-stream pipeline: signal words that one stream of a
 * device writes and another waits for (">= value") in stream order, executed by the command processors -- no event, no host
 * wake-up between the copy stream and the compute stream (inference_loop.Prefetcher(copy_stream=True)).
 * evs_signal_alloc: 8 bytes of signal memory, zeroed (EVS_ESTATE when the device has no stream wait-value operations). */
EVS_API int evs_signal_alloc(void **out);
EVS_API int evs_signal_free(void *signal);
EVS_API int evs_stream_write_value(void *stream, void *signal, uint32_t value);
EVS_API int evs_stream_wait_value(void *stream, void *signal, uint32_t value);

/* Row-split tables, receiver side (one index per bag): sample b of table k reads the partial of the rank whose row range
 * holds indices[k][b] -- rank r holds rows [r*n/world, (r+1)*n/world) -- i.e. row  row_off[r] + b  of the partials inside
 * the receive buffer (row_off: HOST array of `world` row offsets, one per source rank).  dst[k][b] (device, int64) receives
 * that row number: the index list of the "gathered" feature the fused interaction kernel then reads.  T <= 32, world <= 64. */
EVS_API int evs_rowsplit_route(int T, int64_t B, int world, const int64_t *const *indices, const int64_t *n_rows,
                               const int64_t *row_off, int64_t *const *dst, void *stream);

/* Same operation for the stacked layout of the Criteo collate
 * (dlrm_data_pytorch.py:397-410: lS_i and lS_o are (T,B) int64 tensors, one row per
 * table): indices[k] = indices_base + k*indices_row_stride (B entries each, i.e.
 * nnz[k] = nnz_per_table), offsets[k] = offsets_base + k*offsets_row_stride.
 * Strides in elements.  Saves the caller from building four T-long pointer arrays
 * per batch.  offsets_base == NULL: one index per bag (as offsets == NULL above). */
EVS_API int evs_embedding_bag_sum_stacked(int T, int64_t B, int d, int codec,
                          const void *const *tables, const int64_t *n_rows,
                          const int64_t *indices_base, int64_t indices_row_stride,
                          int64_t nnz_per_table,
                          const int64_t *offsets_base, int64_t offsets_row_stride,
                          const float *const *row_weights,
                          float *out, int64_t out_table_stride, int64_t out_bag_stride,
                          void *stream);

/* Synchronises `stream`, returns EVS_EINDEX if any launch since the last call saw
 * an out-of-range index (and clears the flag), else 0. */
EVS_API int evs_check_index_errors(void *stream);

/* ---------------------------------------------------------------------------
 * a11: the offline encoders as a GPU batch tool
 *     (script/reduce_precision.py:26-51 u16, :140-172 + :321 u4, :270 u8;
 *      script/convert_ev_to_binary.py:31-69 byte layout)
 * src: n_rows x d fp32 (device), dst: n_rows x (d*codec/8) bytes (device), the
 * reference's ev-table-N.bin layout for that precision.  Bit-exact with the
 * reference's Python arithmetic (fp64 on the widened fp32 values).
 * ------------------------------------------------------------------------- */
EVS_API int evs_encode_table(int codec, int64_t n_rows, int d, const float *src, void *dst, void *stream);

/* Device-side address of a pinned (hipHostMalloc / torch pin_memory) host buffer, NULL when the
 * buffer is not device-accessible.  The batch-1 entry points (evs_cache_request*, ev_lookup) take
 * such pointers for the ids, the rows and the hit flags: a request of 26 ids and 936 floats then
 * crosses the bus inside the kernel -- one launch and one synchronise per request, no copy commands
 * (the reference's loop is one request at a time: cache_manager.cpp:231-237). */
EVS_API void *evs_host_device_pointer(void *host_ptr);

/* ---------------------------------------------------------------------------
 * a3: DLRM_Net.interact_features, arch_interaction_op="dot"
 *     (dlrm_s_pytorch.py:483-516)
 *
 * feats: HOST array of F device pointers; feature f of sample b is the d
 * floats at feats[f] + b*feat_strides[f]  (feats[0] = x, feats[1..] = ly).
 * R: device (B, d + P) row-major, P = F(F-1)/2 (itself=0) or F(F+1)/2 (itself=1):
 *   R[b] = [ x[b] | Z[b][i][j] for i in 0..F-1 for j in 0..i-1(+itself) ],  Z = T.T^T
 * fp32 in, fp32 accumulate on the matrix cores (v_mfma_f32_16x16x4_f32).
 * Requires F <= 32, d % 4 == 0, d <= 256.
 * ------------------------------------------------------------------------- */
EVS_API int evs_interact_dot(int64_t B, int F, int d, const float *const *feats,
                     const int64_t *feat_strides, int itself, float *R, void *stream);

/* ---------------------------------------------------------------------------
 * a4: the apply_emb -> interact_features pair of DLRM_Net.sequential_forward
 *     (dlrm_s_pytorch.py:588-601) as ONE kernel: R = interact_features(x, apply_emb(...)).
 *
 * Every row of the per-sample matrix T[b] (F x d) is described by an evs_feature:
 *   dense    (indices == NULL): the d floats at (const float*)src + b*stride
 *                               -- x, or pooled vectors received from the all-to-all;
 *   indirect (indices != NULL): the sum-pooled bag b of table `src` (row-major rows of d
 *                               elements in `codec`), exactly evs_embedding_bag_sum's
 *                               definition for one table (offsets = B bag starts, the
 *                               last bag runs to nnz; row_weights = per-ROW weights or NULL).
 * feats: HOST array of F entries, feats[0] must be dense (x).  All indirect features
 * share `codec`.  Output as evs_interact_dot.  Requires F <= 32, B < 2^31,
 * evs_fused_dim_supported(d), src 16-byte aligned, dense strides % 4 == 0.
 * The (B,F,d) intermediate is never written.  Pooled sums round exactly as in
 * evs_embedding_bag_sum; dot products are fp32 MFMA chains (rtol 1e-5 vs the oracle).
 * ------------------------------------------------------------------------- */
typedef struct evs_feature {
    const void *src;
    int64_t stride;
    const int64_t *indices;   /* NULL = dense feature -- except: n_rows > 0, nnz == 0 and offsets given = a table whose
                                 bags are all empty in this batch (an empty index array has no address) */
    const int64_t *offsets;   /* indirect: B bag starts; NULL on EVERY indirect feature = one index per bag */
    int64_t nnz;
    int64_t n_rows;
    const float *row_weights;
    int64_t offsets_len;   /* readable entries of `offsets` (>= B); 0 means B.  A batch SLICE of a longer
                              offsets array passes offsets+b0 and offsets_len = B_total-b0: the slice's last
                              bag then ends at offsets[B] instead of nnz. */
} evs_feature;

/* 1 if the fused kernel is built for this embedding dimension (16,32,36,48,64,128). */
EVS_API int evs_fused_dim_supported(int d);

EVS_API int evs_emb_interact_dot(int64_t B, int F, int d, int codec, const evs_feature *feats,
                                 int itself, float *R, void *stream);

/* Stacked-layout form (Criteo collate: indices and offsets are (T,B) int64 tensors, see
 * evs_embedding_bag_sum_stacked): feature 0 = x (B,d) with row stride x_stride floats,
 * feature k+1 = bag-sum over tables[k].  F = T + 1.  offsets_base == NULL declares ONE INDEX PER BAG
 * (the Criteo collate's offsets = arange(B), dlrm_data_pytorch.py:407-408): bag b = indices[k][b]; the
 * offsets stage of the kernel is skipped (fp32 tables, unweighted). */
EVS_API int evs_emb_interact_dot_stacked(int64_t B, int T, int d, int codec,
                                 const void *const *tables, const int64_t *n_rows,
                                 const float *x, int64_t x_stride,
                                 const int64_t *indices_base, int64_t indices_row_stride,
                                 int64_t nnz_per_table,
                                 const int64_t *offsets_base, int64_t offsets_row_stride,
                                 const float *const *row_weights, int itself, float *R, void *stream);

/* K independent batches of B samples in one call (a serving loop's queue): x, indices_base, offsets_base (NULL = one
 * index per bag for every batch) and R are HOST arrays of K device pointers, every batch with the same shape and strides.
 * fp32 tables with d in {16, 32, 36} and T <= 27: ONE launch per 8 batches -- a batch's last blocks drain under the
 * next batch's first ones, the rate a caller alternating two HIP streams gets, with no stream but `stream`; other
 * shapes: K launches.  Bit-identical to K calls of evs_emb_interact_dot_stacked either way. */
EVS_API int evs_emb_interact_dot_stacked_multi(int K, int64_t B, int T, int d, int codec, const void *const *tables,
                                               const int64_t *n_rows, const float *const *x, int64_t x_stride,
                                               const int64_t *const *indices_base, int64_t indices_row_stride,
                                               int64_t nnz_per_table, const int64_t *const *offsets_base,
                                               int64_t offsets_row_stride, int itself, float *const *R, void *stream);
/* The same call -- R = interact_features(x, apply_emb(lS_o, lS_i, emb_l, v_W_l)), lS_o given and checked per 16-sample block
 * (dlrm_s_pytorch.py:596-601; the inference loop calls it once per batch, :801-836) -- through a RESIDENT DISPATCHER (round 6):
 * the latency half of the metric.  A launched-and-waited-for batch spends ~12 us outside its kernel (the launch call, the
 * dispatch of a thousand blocks, the completion signal), and below ~4 000 samples a launch IS that floor.  serve_start keeps
 * the tables (fp32, d in {16, 32, 36, 64}, T <= 27; HBM addresses as evs_emb_interact_dot_stacked takes them) and arms a grid
 * of n_blocks blocks (0 = 4 per CU, at most 4 096) that starts with the first post and STAYS on the device: serve_post writes
 * one 64-byte descriptor (no launch: ~1 us of host time) and returns a ticket -- where the host can address device memory
 * (large BAR) straight into the lines in device memory the blocks poll, through the PCIe aperture; elsewhere
 * (or EVS_SERVE_PUBLISH=leader) into a ring in pinned host memory that the grid's leader wavefront reads over the bus and
 * republishes on the device --, every block runs the chunks that fall to it with the very body of the launched kernel (same
 * bits), R is written through to memory, the blocks that complete the batch's arrival counters write the ticket into an
 * answer line in host memory and serve_wait spins on that.  Up to 64 batches may be in flight (a post blocks on the batch 64
 * tickets back); consecutive small batches land on different blocks and overlap.
 *   WHAT THE CALLER GUARANTEES: x / lS_i / lS_o of a batch are complete when it is posted (no stream orders a post), R may be
 *   read by anything STARTED after serve_wait has returned, and nothing else needs the GPU urgently while the grid is
 *   resident -- it holds its compute units; it leaves by itself after idle_us without a post (the next post starts it again;
 *   a device-wide synchronise elsewhere waits that long), and serve_stop sends it home at once.
 * Out-of-range indices / bad offsets: skipped and flagged exactly as the launch form does (evs_check_index_errors).
 * x: 16-byte aligned, row stride a multiple of 4 floats; lS_i / lS_o: (T, B) int64 with unit inner stride, given by the address of
 * row 0 and the elements between two rows; one server is driven by ONE host thread (posts and waits are not locked). */
typedef struct evs_rf_server evs_rf_server;
EVS_API int evs_emb_interact_serve_start(evs_rf_server **out, int T, int d, const void *const *tables, const int64_t *n_rows,
                                         int itself, int n_blocks, int64_t idle_us);
EVS_API int evs_emb_interact_serve_post(evs_rf_server *s, int64_t B, const float *x, int64_t x_stride,
                                        const int64_t *indices_base, int64_t indices_row_stride,
                                        const int64_t *offsets_base, int64_t offsets_row_stride, float *R, uint64_t *ticket);
EVS_API int evs_emb_interact_serve_wait(evs_rf_server *s, uint64_t ticket);
/* which front end this server runs: 1 = host-published (descriptors through the PCIe aperture), 0 = the leader's mailbox */
EVS_API int evs_emb_interact_serve_mode(evs_rf_server *s);
EVS_API int evs_emb_interact_serve_stop(evs_rf_server *s);
EVS_API int evs_emb_interact_serve_destroy(evs_rf_server *s);
/* SURVEY 8(f).3, second half: the apply_emb -> interact_features -> FIRST top-MLP layer chain of
 * DLRM_Net.sequential_forward (dlrm_s_pytorch.py:596-605) in one launch: Z1 = act(R W1^T + b1), act = ReLU when relu != 0
 * (create_mlp, dlrm_s_pytorch.py:205-245: nn.Linear + nn.ReLU).  The interaction rows of 16 samples stay in LDS and feed
 * fp32 MFMA tiles against W1; R reaches HBM only when R != NULL.  One index per bag (indices (T,B) int64, bag b = index b),
 * fp32 tables, d in {16, 32, 36}, T <= 27.  w1_padded: W1 (n1 x K row-major, K = d + P) zero-padded to ((n1 + 15) / 16 * 16)
 * rows of kp = (K + 15) / 16 * 16 floats, 16-byte aligned; b1: n1 floats; Z1: (B, n1) fp32. */
EVS_API int evs_emb_interact_mlp1_stacked(int64_t B, int T, int d, const void *const *tables, const int64_t *n_rows,
                                          const float *x, int64_t x_stride, const int64_t *indices_base,
                                          int64_t indices_row_stride, int itself, const float *w1_padded, int kp,
                                          const float *b1, int n1, int relu, float *Z1, float *R, void *stream);

/* "cat" interaction (dlrm_s_pytorch.py:506-508): R[b] = [x[b] | ly_0[b] | ...], (B, F*d). */
EVS_API int evs_interact_cat(int64_t B, int F, int d, const float *const *feats,
                     const int64_t *feat_strides, float *R, void *stream);

/* ---------------------------------------------------------------------------
 * a6/a7/a8: the cache tier, GPU resident.
 *   EvLFU: cache_algo/EvLFU_C1.py:21-166 (init/set/update_agg_hit/update/request_to_ev_lfu)
 *   LRU:   cache_algo/LRU.py:14-64        LFU: cache_algo/LFU.py:12-95
 * policy: 0 EvLFU, 1 LRU, 2 LFU.  capacity in ENTRIES (--cache-size).  n_tables keys per
 * request (<= 64), table = position, key = (table_1based, row).
 * EvLFU constants: (flush_rate, perfect_item_cap, flush_extra, perfect_mode) =
 *   (0.3, 0.95, 1, 0) cache_algo/EvLFU_C1.py:18-19,:40,:43
 *   (0.3, 0.95, 0, 2) mixed_precs_caching/evlfu_8.hpp:50-51, evlfu_8.cpp:256-270
 *   (0.4, 1.0, 1, 1)  cache_algo/EvLFU_C1_Cython/EvLFU.cpp:12-13,:80-86
 * Rows are cached in the table's codec (arena bytes = capacity * dim*codec/8) and
 * decoded to fp32 on output.  The hash table, priority lists and arena live in HBM.
 * ------------------------------------------------------------------------- */
typedef struct evs_cache evs_cache;
EVS_API int evs_cache_create(evs_cache **out, int policy, int64_t capacity, int n_tables, int dim, int codec,
                             double flush_rate, double perfect_item_cap, int flush_extra, int perfect_mode);
EVS_API int evs_cache_destroy(evs_cache *c);
/* Miss path: tables[k] is DEVICE-ACCESSIBLE memory holding table k in the cache's codec
 * (HBM, or pinned host memory mapped into the device = the host-mmap miss tier,
 * emb_storage/mmap_file_read.py / evlfu_8.cpp:191-250); HOST arrays of n_tables entries.
 * With host-memory tables the batched lookups (one tier, or the two- / three-tier forms) fetch
 * each missing row once (de-duplicated, into the arena) and serve the whole batch from HBM; hits
 * of the running batch are never evicted by it. */
EVS_API int evs_cache_set_backing(evs_cache *c, const void *const *tables, const int64_t *n_rows);
/* B requests replayed strictly in order (exact reference semantics; B=1 is the reference's
 * request_to_ev_lfu / request_to_lru / request_to_lfu).  rows: device (B, n_tables) int32;
 * out: device (B, n_tables, dim) fp32; hit: device (B, n_tables) bytes (0/1).
 * approx_thres > 0: EvLFU approximate mode (EvLFU_C1.py:122-125,:142-152). */
EVS_API int evs_cache_request(evs_cache *c, int64_t B, const int32_t *rows, float *out, uint8_t *hit,
                              int approx_thres, void *stream);
/* a9: two-tier request, mixed_precs_caching/evlfu_8.cpp:669-796 (request_to_c1_c2): C1 = main
 * precision tier, C2 = secondary precision tier, both EvLFU caches created with the
 * mixed_precs_caching constants (0.3, 0.95, 0, 2), each with its own backing tables in its own
 * codec.  tier (device, (B,n_tables) bytes): 1 = C1 hit, 2 = C2 hit, 0 = miss.  When C1 is full and
 * the combined agg_hit < high_agghit_threshold (23, evlfu_8.hpp:70) double misses with odd table
 * index go to C1, even ones to C2; at or above the threshold all go to C2; while C1 is not full
 * every C1 miss goes to C1 and C2 is left untouched.  Victims are FIFO-oldest (the C++ evicts in
 * unordered_set order); perfect requests are counted in c1's n_perfect_hits. */
EVS_API int evs_cache_request_c1c2(evs_cache *c1, evs_cache *c2, int64_t B, const int32_t *rows, float *out,
                                   uint8_t *tier, int high_agghit_threshold, void *stream);
/* The exact policy as a RESIDENT SERVER on the GPU (round 5) -- the reference serves one request at a time
 * (cache_algo/EvLFU_C1.py:97-166, dlrm_s_pytorch_C1.py:227-275, cpp_socket_client.py:129-157), where evs_cache_request
 * costs a launch and a synchronise per request on top of the request itself.  evs_cache_serve_start arms a one-wavefront
 * kernel (the same code as evs_cache_request's: same hit flags, rows, list order) that stays on the device and takes requests
 * from a mailbox in pinned host memory: the host writes the T ids and a sequence number into one 128-byte line, the server
 * polls that line over the bus, runs the request, writes the T x dim fp32 rows into slot (sequence % n_slots) of `ring`
 * (DEVICE memory, n_slots x T x dim floats: the caller gets device rows without a launch, a copy or a synchronise) and the
 * hit flags + the sequence number into a host line the caller spins on.  evs_cache_serve_request blocks until the answer is
 * there (hit: T bytes on the host; *slot_out: which ring slot holds the rows).
 * HOW LONG A SLOT'S ROWS STAY VALID: the server overwrites slot s when the request n_slots later is POSTED -- host order,
 * not the order of the caller's stream.  A caller that enqueues its reads of the slot (a copy, a kernel) on a stream and does
 * not synchronise must say so: evs_cache_serve_consumed(c, slot, stream) records an event behind those reads, and the request
 * that is about to hand the slot out again waits for it on the host.  Without that call the caller has to have FINISHED
 * reading a slot before it posts the (n_slots - 1)-th request after it.
 * An idle server leaves by itself after idle_us (a device-wide synchronise elsewhere -- hipDeviceSynchronize,
 * torch.cuda.synchronize -- waits for it, i.e. up to idle_us longer than it otherwise would) and is
 * started again by the next request; evs_cache_stats / _dump / _request / _request_c1c2 / _set_backing / _reset_counters /
 * _destroy send it home first (it writes the policy state back on its way out), and a server started after an exact-path
 * launch waits for that launch on its own stream.  A request that fails (EVS_EHIP: the server did not answer / could not be
 * started) leaves host and device agreeing on the sequence number, so the next request is a fresh one.
 * At most 28 tables (the 128-byte request line: four 32-byte sectors of seven ids and a guard word each -- ids are accepted
 * only from a sector whose guard holds the number awaited, whatever granularity the bus delivers the line in);
 * EVS_ESTATE for a cache on the batched path. */
EVS_API int evs_cache_serve_start(evs_cache *c, int approx_thres, float *ring, int n_slots, int64_t idle_us);
EVS_API int evs_cache_serve_request(evs_cache *c, const int32_t *rows_host, uint8_t *hit_host, int *slot_out);
/* ... with the T ids given by ADDRESS (round 6): ids_dev[t * ids_stride], int64 in device memory -- element 0 of each row of the
 * (T, B) lS_i the reference's loop has moved to the GPU (dlrm_wrap, dlrm_s_pytorch.py:131-147; apply_emb_evstore then takes it back
 * with lS_i.cpu(), dlrm_s_pytorch_C1.py:233-239: a copy and a synchronise per request).  The server reads the ids itself; they
 * must be complete when the call is made. */
EVS_API int evs_cache_serve_request_dev(evs_cache *c, const int64_t *ids_dev, int64_t ids_stride, uint8_t *hit_host, int *slot_out);
/* ... with the rows delivered to a DEVICE buffer of the caller's (out_dev: T x dim floats) instead of a ring slot (round 6): what
 * the reference's loop wants is a fresh tensor per request (EvLFU_C1.py:157-161 builds 26 of them), and copying the slot into
 * one was a launch and an event per request.  Exactly one of rows_host / ids_dev (+ ids_stride, as above) is given; at most 26
 * tables (the address rides in the line's last two id words).  Nothing orders the server's stores against the caller's
 * streams: out_dev must not be in use by work still pending when the call is made (a block a stream-ordered allocator has just
 * handed out counts as in use until that stream is idle), and holds the rows when the call returns. */
EVS_API int evs_cache_serve_request_to(evs_cache *c, const int32_t *rows_host, const int64_t *ids_dev, int64_t ids_stride, float *out_dev, uint8_t *hit_host);
EVS_API int evs_cache_serve_consumed(evs_cache *c, int slot, void *stream);
EVS_API int evs_cache_serve_stop(evs_cache *c);
/* Batched EvLFU lookup with snapshot semantics (the throughput path; no reference counterpart --
 * the reference is batch-1): all B requests are probed against the cache as it is when the call
 * starts, rows are served at once (arena for hits, backing for misses: always exactly the table
 * rows), then priorities are raised (atomicMax to agg_hit), the batch's unique misses are inserted
 * and the lowest priorities are evicted to make room (+ the EvLFU flush rule per batch).
 * Same argument layout as evs_cache_request.  A cache object is driven either by
 * evs_cache_request (exact) or by evs_cache_lookup_batch, never both (EVS_ESTATE). */
EVS_API int evs_cache_lookup_batch(evs_cache *c, int64_t B, const int32_t *rows, float *out, uint8_t *hit,
                                   void *stream);
/* How the batched path of this cache makes room (before its first batched call; EVS_ESTATE afterwards).  A cache that
 * is not given a policy decides at its first batched call: EVS_CACHE_POLICY = plan | sampled | setassoc in the environment,
 * else 2 where it applies (tiers whose tables the kernels read in place from HBM, capacity >= 8: a single tier, or BOTH tiers
 * of a two- / three-tier lookup -- a pair is all set-associative or not at all) and 1 elsewhere.
 *   2 "setassoc": the cache is 8-way set-associative.  A key is its dense row number over all tables, permuted (a
 *     bijection of [0, 2^b)), then SPLIT: set = x mod nset, tag = x div nset -- (set, tag) IS the key, so a way is one
 *     32-bit word (priority 6 bits | batch stamp | tag + 1) and a set 32 bytes; way w of set s owns arena row 8 s + w.
 *     A probe reads that record; the update kernel's thread that inserts a new key takes the lowest priority OF THE KEY'S
 *     OWN SET (free ways first) with one CAS and writes the row.  No hash chains, tombstones, sweeps or entry arrays.
 *     capacity / 8 sets (up to 7 entries of the capacity unused); fewer than 2^32 rows over all tables; tables in HBM
 *     (host-memory / file-backed tables: EVS_ESTATE).
 *     A tier ALONE keeps TWO arena rows per way (round 5; bit 25 of the way word names the live one, a replacement writes the
 *     other and flips the bit with the CAS that installs the key), and on such a tier evs_cache_lookup_interact with fp32
 *     rows makes the policy update INSIDE its one launch: the thread that misses a key claims a way of the key's set right
 *     there (one CAS beside the priority raises) and the lanes that gather the key's row for the interaction store it into
 *     the arena -- no miss lists, no update launch (37-38 -> 32-33 us per 16 384-batch at the 10 % Kaggle cache).  What that
 *     changes: a way filled by the running batch carries the batch's stamp and is a MISS for every prober of the same launch
 *     (its row may not be there yet: the key is served from its table), and a key that was resident when the batch arrived
 *     can be retired by one of the batch's own inserts before a later block looks for it.  So under this form a hit flag
 *     says "served from the cache": flag = 1 => the key was resident when the batch arrived (always); flag = 0 => it was not,
 *     OR one of this batch's inserts retired it first (at most as many keys as the batch evicted; the row is then served from
 *     the table, exact as ever, and the key is inserted again like any other miss).  Keys inserted by batch k are hits from
 *     batch k + 1 on.  Everything listed under "WHAT IS THE SAME" below holds unchanged.  EVS_CACHE_INLINE=0 (environment)
 *     keeps the update as a launch of its own behind the probe: strict snapshot flags, as evs_cache_lookup_batch always has.
 *     (A single reduced-precision tier -- codec 16 / 8 / 4, d in {16, 32, 36} -- takes the same one-launch form.)
 *     A C1 + C2 pair that starts out together SHARES its set records: one
 *     128-byte line per set index holds C1's 8 ways and C2's ways (two 8-way sub-sets, picked by one more bit of the
 *     quotient, for the reference's 1 : 2 capacity split, evlfu_8.cpp:63-78), so a key's two tier probes are ONE line
 *     request; the routing rule's "while C1 is not full" (evlfu_8.cpp:570-601) is read PER KEY: a double miss goes to C1
 *     while the key's own C1 set has a free way, and by the agg_hit / odd-even rule once it has none (the hashed forms
 *     read it off the tier's entry count);
 *   1 "sampled": a hash over an entry arena; one kernel after the consumers -- the thread that inserts a new key picks
 *     that key's victim itself, the lowest priority of one sampled group of 8 entries (free entries first);
 *   0 "plan": insert -> plan -> evict -> assign -> close, the lowest priorities of a clock-hand window go, exactly as
 *     many as the batch needs (the file-backed miss tier always takes this form).
 * WHAT IS THE SAME under all three, and tested: served rows are exactly the table rows; hit flags are residency when the
 * batch arrives (snapshot; the one-launch form of policy 2: see there); no duplicate keys; size <= capacity; priorities only rise (monotone max of agg_hit,
 * cache_algo/EvLFU_C1.py:65-79); the EvLFU flush fires when the top bucket reaches max_perfect (:36-44).
 * WHAT DIFFERS from the reference's sequential EvLFU (:97-166 evicts the FIFO-oldest entry of the globally lowest
 * bucket): 0 evicts the lowest priorities of a window, 1 the lowest of 8 sampled entries, 2 the lowest of the key's set --
 * under 1 and 2 a key that was HIT in the running batch can be a victim while lower priorities exist elsewhere, and
 * under 2 a set that receives more new keys than it has ways in one batch turns the rest away (they are served from
 * the tables and not cached this time); which keys are resident after a batch depends on thread timing under 1 and 2.
 * Hit rate at the 10 % Criteo-Kaggle cache (Zipf 0.75, B = 16 384, 600 batches): 0 0.8868, 1 0.8862, 2 0.8840, the
 * sequential oracle on the same stream 0.884 (tests/test_gpu_fullsize.py asserts |batched - sequential| <= 0.01 over
 * ten batches at full size for all three; bench.py: cache_tier.hit_rate_vs_sequential_oracle). */
EVS_API int evs_cache_set_batch_policy(evs_cache *c, int policy);
/* Batched two-tier lookup, snapshot semantics: the throughput form of evs_cache_request_c1c2 (no reference
 * counterpart).  Every key is probed in C1, then in C2, against the tiers as they stand when the call starts;
 * agg_hit of a request counts keys found in either tier; a hit is served, and its priority raised, in the tier
 * that holds it; a double miss is routed by the reference's rule (evlfu_8.cpp:570-601) evaluated on the snapshot
 * -- C1 not full: C1; C1 full and agg_hit < threshold: odd table index -> C1, even -> C2; else C2 -- served from
 * the destination tier's backing table at that tier's precision and inserted there once per batch.  out: (B,T,dim)
 * fp32; tier (B,T): 1 = C1 hit, 2 = C2 hit, 0 = miss.  Both caches take the batched path from then on.
 * MISS TIERS OUTSIDE HBM (BASELINE configs[4] composed; the reference's tiers read their misses from files inside the
 * request: evlfu_8.cpp:380-414 get_from_file, reader pool :191-250, :603-625): either tier may sit over pinned host tables
 * (evs_cache_set_backing) or file-backed ones (evs_cache_set_file_backing).  The pair then runs: probe (this batch's hits
 * and served alt rows are pinned in their arenas) -> both tiers' policy updates, each missing row fetched ONCE into its
 * tier's arena (over the bus by the thread that inserts it; staged tables: by the host's reader pool, de-duplicated) ->
 * every missed position re-pointed at the arena / staged copy -> the consumers.  Policies: "sampled" over device-visible
 * tables (pinned, registered), "plan" on BOTH tiers as soon as one has staged tables; "setassoc": EVS_ESTATE.  A position
 * whose key another request of the batch routed to C1 (odd tables only) is served C1's copy at C1's precision. */
EVS_API int evs_cache_lookup_batch_c1c2(evs_cache *c1, evs_cache *c2, int64_t B, const int32_t *rows, float *out,
                                        uint8_t *tier, int high_agghit_threshold, void *stream);
/* The same two-tier snapshot lookup with the interaction as its consumer (BASELINE configs[4] end to end):
 * R = interact_features(x, rows) with every row decoded from the precision of the tier that serves it inside the
 * interaction kernel -- the fp32 (B,T,d) rows are never materialised.  d in {16, 32, 36}, T <= 31. */
EVS_API int evs_cache_lookup_interact_c1c2(evs_cache *c1, evs_cache *c2, int64_t B, const int32_t *rows, const float *x,
                                           int64_t x_stride, int itself, float *R, uint8_t *tier,
                                           int high_agghit_threshold, void *stream);
/* Batched THREE-tier lookup: the two calls above with the alt-key tier C3 (the throughput form of request_to_c1_c2_c3,
 * evlfu_8.cpp:492-667; no reference counterpart).  A double miss whose key is a member of C3 and whose alt row is
 * resident in C1 (else C2) when the call starts is served that row -- tier code 3, decoded at the precision of the
 * tier holding it; its recency flag is set, the request's agg_hit counts it, nothing is inserted for it.  The keys
 * this batch's policy update EVICTS (not flushes) from C1 / C2 become members of C3, visible from the next batch on.
 * In this form C3 is an 8-way set-associative key set with second chance inside each set (the alt key of a key is a
 * pure function of the key: the tier only has to remember WHICH keys it knows): one line and one CAS per operation,
 * no global FIFO -- the exact FIFO order is the batch-1 machine's (evs_cache_request_c1c2c3, evs_aprx_apply_ops).
 * A tier object is driven by one of the two forms, never both (EVS_ESTATE). */
typedef struct evs_aprx evs_aprx;   /* (created with evs_aprx_create, below) */
EVS_API int evs_cache_lookup_batch_c1c2c3(evs_cache *c1, evs_cache *c2, evs_aprx *c3, int64_t B, const int32_t *rows,
                                          float *out, uint8_t *tier, int high_agghit_threshold, void *stream);
EVS_API int evs_cache_lookup_interact_c1c2c3(evs_cache *c1, evs_cache *c2, evs_aprx *c3, int64_t B, const int32_t *rows,
                                             const float *x, int64_t x_stride, int itself, float *R, uint8_t *tier,
                                             int high_agghit_threshold, void *stream);
/* Members of the batched C3 as (table_1based, row, recency flag) triples (host; may be NULL), returns their number;
 * out4 (host, may be NULL): [members counted on insert, alt hits served, capacity of the sets, 0].  Synchronises. */
EVS_API int64_t evs_aprx_batch_dump(evs_aprx *p, int64_t *triples, int64_t max_triples, int64_t *out4, void *stream);
/* The same lookup feeding the interaction directly: R = interact_features(x, [rows of the 26 keys])
 * (B, d + F(F-1)/2) without materialising the rows -- the probe writes a table of row addresses
 * (arena row for a hit, backing row for a miss) that the fused MFMA kernel consumes.  fp32 caches:
 * dimensions supported by evs_fused_dim_supported (up to 16 384 requests the probe itself runs inside that
 * kernel).  16 / 8 / 4-bit caches (tables in HBM, d in {16, 32, 36}): the rows are decoded inside the
 * interaction kernel (the reference's one-layer reduced-precision builds, cache_manager.cpp:13-20). */
EVS_API int evs_cache_lookup_interact(evs_cache *c, int64_t B, const int32_t *rows, const float *x, int64_t x_stride,
                                      int itself, float *R, uint8_t *hit, void *stream);
/* The one-launch form of a set-associative tier (round 5: the policy update runs INSIDE evs_cache_lookup_interact's probe +
 * interaction launch; a hit flag then says "served from the cache": 1 => resident at arrival, 0 => not resident OR retired
 * by one of this batch's own inserts) is the default wherever its conditions hold (a tier alone, 8 ways, two-copy arena, at
 * least 8 stamp bits in the way word).  on = 0 restores the two-launch chain with strict snapshot flags for this cache, on = 1
 * the default; the environment variable EVS_CACHE_INLINE=0 only changes the default of caches that were never told.  May be
 * called between batches. */
EVS_API int evs_cache_set_inline_update(evs_cache *c, int on);
/* out8: [size, n_free, n_tombstones, n_flush, n_evict, n_requests, n_perfect_hits, n_hits];
 * hist (may be NULL): n_tables+1 resident-entry counts per priority. */
EVS_API int evs_cache_batch_stats(evs_cache *c, int64_t *out8, int64_t *hist, void *stream);
/* resident (priority, table_1based, row) triples of the batched path, unordered; returns the count. */
EVS_API int64_t evs_cache_batch_dump(evs_cache *c, int64_t *triples, int64_t max_triples, void *stream);
/* File-backed miss tier: the reference's mmap miss path (emb_storage/mmap_file_read.py:32-40,
 * reader pool mixed_precs_caching/evlfu_8.cpp:191-250) under the GPU cache.  evs_filetier_open maps every
 * ev-table-N.bin read-only (row r at byte row_bytes * r) and REGISTERS tables with the GPU (hipHostRegister, mapped:
 * kernels read them over the bus, zero-copy) smallest first while their total fits pinned_budget_bytes; the others stay
 * plain mappings ("staged"): the batched lookup lists the batch's de-duplicated new keys, a pool of host threads copies
 * those rows out of the mappings into a pinned staging buffer and the fill kernel takes them from there -- every missing
 * row is read from the file once per batch.  evs_cache_set_file_backing replaces evs_cache_set_backing; with staged
 * tables only the batched lookups (evs_cache_lookup_batch / _interact, and the two- / three-tier forms) are served.  The tier outlives the cache's use
 * of it; evs_filetier_close unmaps.  evs_filetier_fetch is the reader pool itself (host memory in, host memory out). */
typedef struct evs_filetier evs_filetier;
EVS_API int evs_filetier_open(evs_filetier **out, int n_tables, const char *const *paths, int64_t row_bytes,
                              int64_t pinned_budget_bytes);
EVS_API int evs_filetier_info(evs_filetier *ft, int64_t *n_rows, const void **dev_ptrs, int *registered, int64_t *pinned_bytes);
EVS_API int evs_filetier_fetch(evs_filetier *ft, int64_t n, const uint64_t *keys /* table_1based << 32 | row */, void *dst,
                               uint32_t skip_mask);
EVS_API int evs_filetier_close(evs_filetier *ft);
EVS_API int evs_cache_set_file_backing(evs_cache *c, evs_filetier *ft);
EVS_API int64_t evs_cache_staged_rows(evs_cache *c);   /* rows the reader pool has fetched for this cache so far */
/* a12: the alt-key ("approximate embedding") tier C3 -- mixed_precs_caching/aprx_embedding.cpp and
 * evlfu_8.cpp:474-490,492-667 (request_to_c1_c2_c3).  DETERMINISTIC RE-SPECIFICATION, parity unpinned:
 * the reference fills the tier from asynchronous threads racing the request thread.  Here keys evicted
 * (not flushed) from C1/C2 are queued; when 50 (IO_JOB_Q_SIZE) are pending the batch is inserted at the
 * start of the next request (second-chance FIFO eviction makes room).  On a double miss whose alt key is
 * mapped and whose alt row is resident in C1 (else C2), that row's vector is served (tier code 3),
 * decoded at the precision of the tier holding it; the key's recency flag is set; the request's agg_hit
 * counts it; nothing is inserted for it.  alt_tables[k]: device-accessible uint32[n_rows[k]] with
 * alt_key = alt_row*100 + alt_table_1based (the reference's alt-key files store this big-endian,
 * script/convert_altkeys_to_binary.py:27-57; convert to native order when loading). */
typedef struct evs_aprx evs_aprx;
EVS_API int evs_aprx_create(evs_aprx **out, int64_t capacity, int n_tables);
EVS_API int evs_aprx_destroy(evs_aprx *p);
EVS_API int evs_aprx_set_altkeys(evs_aprx *p, const uint32_t *const *alt_tables, const int64_t *n_rows);
EVS_API int evs_aprx_stats(evs_aprx *p, int64_t *out4 /* size, n_hit, n_pending, error */, void *stream);
/* APRX_EV's public single-key methods applied in order (the part of the tier that IS pinned to the compiled reference,
 * driven single-threaded): ops = n x (op, table_1based, row), op 0 insert_altkey (aprx_embedding.cpp:278-288: evict one
 * when full, push on the FIFO, map[key] = {alt key of the row, false}), 1 get_altkey_str (:341-350), 2 set_recency_flag_c3
 * (:402-411), 3 evict_one_key (:390-400, second chance :360-388).  res[i] (device): op 0 the alt key, op 1 the alt key or
 * 0xffffffff on a miss, else 0.  evs_aprx_dump_queue: the FIFO front to back as (table_1based, row) pairs (host),
 * stale duplicates included (print_all_keys_in_c3, :430-434); returns its length. */
EVS_API int evs_aprx_apply_ops(evs_aprx *p, int64_t n, const int32_t *ops, uint32_t *res, void *stream);
EVS_API int64_t evs_aprx_dump_queue(evs_aprx *p, int64_t *pairs, int64_t max_pairs, void *stream);
/* evs_cache_request_c1c2 with the alt-key tier (c3 may be NULL = plain two tiers). */
EVS_API int evs_cache_request_c1c2c3(evs_cache *c1, evs_cache *c2, evs_aprx *c3, int64_t B, const int32_t *rows,
                                     float *out, uint8_t *tier, int high_agghit_threshold, void *stream);
/* out8 (host): [min_C1, n_perfect, size, n_flush, n_evict, n_requests, n_perfect_hits, n_hits].
 * Synchronises the stream.  Returns EVS_ESTATE if the policy hit an inconsistency. */
EVS_API int evs_cache_stats(evs_cache *c, int64_t *out8, void *stream);
EVS_API int evs_cache_reset_counters(evs_cache *c, void *stream);
/* Resident keys in list order as (bucket | frequency | 0, table_1based, row) triples (host);
 * returns the number of resident keys (or a negative error). */
EVS_API int64_t evs_cache_dump(evs_cache *c, int64_t *triples, int64_t max_triples, void *stream);

/* ---------------------------------------------------------------------------
 * a6-a9, a12 at batch 1: the HOST engine of the exact policies (csrc/evs_hostcache.hip).
 * The reference's EVStore loop is one request at a time (--test-mini-batch-size=1, dlrm_s_pytorch_C1.py:236-239) and
 * its policies are sequential by definition; a chain of dependent pointer updates is a CPU cache hierarchy's job (one
 * wavefront replays a request in 28 us + launch + synchronise, one host core in a few).  Same policies, same constants,
 * same argument meaning as evs_cache_create / evs_cache_request / evs_cache_request_c1c2c3 / evs_aprx_* above, same
 * golden traces bit for bit -- but every pointer is HOST memory, there is no stream, and nothing touches the GPU:
 *   tables[k]   any host-readable mapping of ev-table-{k+1}.bin in the tier's codec (mmap, pinned, malloc);
 *               a miss reads ONE row from it (evlfu_8.cpp:380-414 get_from_file: fseek + fread)
 *   rows        (B, n_tables) int32, requests replayed strictly in order
 *   out         (B, n_tables, dim) fp32, decoded from the tier's codec
 *   hit / tier  (B, n_tables) bytes
 * The tier's arena (capacity x dim*codec/8 bytes) and its hash / lists live in host memory.  Errors as everywhere:
 * EVS_EINDEX for a row id outside its table, EVS_ESTATE when the policy state is inconsistent (what the Python
 * reference would raise on).  The batched (snapshot) lookups stay on the GPU tier.
 * evs_hostcache_stats out8 = [min_C1 | LFU least_freq, n_perfect, size, n_flush, n_evict, n_requests, n_perfect_hits,
 * n_hits]; evs_hostcache_dump = evs_cache_dump's triples; evs_hostaprx_* = evs_aprx_* (alt_tables: host uint32 arrays,
 * native byte order).
 * ------------------------------------------------------------------------- */
typedef struct evs_hostcache evs_hostcache;
typedef struct evs_hostaprx evs_hostaprx;
EVS_API int evs_hostcache_create(evs_hostcache **out, int policy, int64_t capacity, int n_tables, int dim, int codec,
                                 double flush_rate, double perfect_item_cap, int flush_extra, int perfect_mode);
EVS_API int evs_hostcache_destroy(evs_hostcache *c);
EVS_API int evs_hostcache_set_backing(evs_hostcache *c, const void *const *tables, const int64_t *n_rows);
EVS_API int evs_hostcache_request(evs_hostcache *c, int64_t B, const int32_t *rows, float *out, uint8_t *hit,
                                  int approx_thres);
EVS_API int evs_hostcache_request_c1c2c3(evs_hostcache *c1, evs_hostcache *c2, evs_hostaprx *c3 /* may be NULL */,
                                         int64_t B, const int32_t *rows, float *out, uint8_t *tier,
                                         int high_agghit_threshold);
EVS_API int evs_hostcache_stats(evs_hostcache *c, int64_t *out8);
EVS_API int evs_hostcache_reset_counters(evs_hostcache *c);
EVS_API int64_t evs_hostcache_dump(evs_hostcache *c, int64_t *triples, int64_t max_triples);
EVS_API int evs_hostaprx_create(evs_hostaprx **out, int64_t capacity, int n_tables);
EVS_API int evs_hostaprx_destroy(evs_hostaprx *p);
EVS_API int evs_hostaprx_set_altkeys(evs_hostaprx *p, const uint32_t *const *alt_tables, const int64_t *n_rows);
EVS_API int evs_hostaprx_stats(evs_hostaprx *p, int64_t *out4 /* size, n_hit, n_pending, error */);
EVS_API int evs_hostaprx_apply_ops(evs_hostaprx *p, int64_t n, const int32_t *ops, uint32_t *res);
EVS_API int64_t evs_hostaprx_dump_queue(evs_hostaprx *p, int64_t *pairs, int64_t max_pairs);

/* ---------------------------------------------------------------------------
 * a14: the reference's cache-manager C ABI (mixed_precs_caching/cache_manager.cpp), bound by
 * cache_algo/cpp_socket_client.py:69-83 through ctypes.  Same names, same signatures.
 *   ev_lookup: reads 26 int32 row ids (0-based; table = position), returns a pointer to the
 *   library-owned static float[26*36], table-major, valid until the next call; not re-entrant;
 *   side effect perfect-hit counter += (all 26 hit).  NULL (after printing) on a configuration error
 *   (the reference prints and exit(-1)s).
 * Configuration: the reference's five compile-time knobs (cache_manager.cpp:13-20) at run time.
 *   n_caching_layer 1 (C1), 2 (C1 + C2 = request_to_c1_c2) or 3 (+ the alt-key tier = request_to_c1_c2_c3;
 *   needs evs_manager_set_altkey_dir / EVS_ALTKEY_DIR; sizes from size_proportion "a-b-c" as evlfu_8.cpp:63-92),
 *   main_precision 32|16|8|4, secondary_precision 16|8|4,
 *   total_size in fp32-row equivalents (one tier: capacity = total_size * 32/main_precision entries;
 *   two tiers: total_size/2 each, i.e. (total/2)*32/main and (total/2)*32/secondary entries -- except an 8-bit
 *   secondary tier under a 32/16-bit main tier, which gets (total/2)*16 entries because the reference's
 *   EVLFU_8BIT constructor multiplies an already-multiplied capacity by 4 again: evlfu_32.cpp:102, evlfu_16.cpp:95,
 *   evlfu_8.cpp:93; evs_manager_tier_capacity reports what was built),
 *   ev_table_root = directory holding ev-table/binary, ev-table-16/binary, ev-table-8/binary,
 *   ev-table-4/binary (evlfu_*.hpp EV_TABLE_PATH), backing 0 = tables in HBM, 1 = pinned host.
 *   Without a call, ev_lookup reads EVS_N_CACHING_LAYER, EVS_MAIN_PRECISION, EVS_TOTAL_SIZE,
 *   EVS_EV_TABLE_ROOT, EVS_BACKING from the environment on first use.
 * ------------------------------------------------------------------------- */
EVS_API int evs_manager_configure(int n_caching_layer, int main_precision, int secondary_precision,
                                  int64_t total_size, const char *size_proportion, const char *ev_table_root,
                                  int backing);
EVS_API int evs_manager_set_altkey_dir(const char *dir);  /* n_caching_layer 3: directory of the alt-key ev-table-N.bin files */
EVS_API long long evs_manager_perfect_hit(void);
EVS_API long long evs_manager_tier_capacity(int tier);   /* entries of tier 1 | 2 | 3 as configured (0 = absent) */
EVS_API long long evs_manager_aprx_hit(void);             /* evlfu_8bit->aprx_ev_hit (cache_manager.cpp:279) */
EVS_API float *ev_lookup(int *arr);                      /* cache_manager.cpp:231 */
EVS_API float *get_ev_values(int *arr);                  /* cache_manager.cpp:257 */
EVS_API void print_perfect_hit(void);                    /* cache_manager.cpp:262 */
EVS_API int ev_lookup_based_on_list_keys(int *arr);      /* cache_manager.cpp:239 (dead in the reference) */
EVS_API void test_arr(int *arr);                         /* cache_manager.cpp:154 */
EVS_API void init_global_vars(void);                     /* cache_manager.cpp:410 (socket server: not built) */
EVS_API void start_server_threads(void);                 /* cache_manager.cpp:419 (socket server: not built) */

#ifdef __cplusplus
}
#endif
#endif /* EVSTORE_HIP_H */
