/* evstore_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference's algorithms for the embedding
 * lookup / EvLFU cache / interaction hot path.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library, and only as the checker.  The product path
 * (ev-store-dlrm_amd/) never links, imports or falls back to it.
 *
 * Parity status: PINNED.  Every function here is checked in
 * tests/test_oracle_golden.py against vectors produced by running the
 * reference itself (tests/golden/make_golden.py imports the reference's
 * Python; oracle/ref/ref_codec_dump.cpp links the reference's C++ decoders).
 *
 * Each function cites the reference file:line it restates (paths relative to
 * /root/reference).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAX_TABLES 64
#define ORC_NBUCKET_MAX 65 /* buckets 0..n_tables */

/* ------------------------------------------------------------------------ */
/* a10: codecs, bytes -> fp32                                                */
/* ------------------------------------------------------------------------ */

/* mixed_precs_caching/evlfu_8.cpp:370-378  f = ((float)u / 254) * 2 - 1, all fp32 */
void orc_decode_u8(const uint8_t *in, int64_t n, float *out) {
    for (int64_t i = 0; i < n; i++) {
        float v = (float)in[i];
        v = v / 254;
        v = v * 2;
        out[i] = v - 1;
    }
}

/* mixed_precs_caching/evlfu_4.hpp:46 (table), evlfu_4.cpp:319-341 (high nibble first).
 * Index 15 is out of bounds in the reference; the encoder never emits it
 * (script/reduce_precision.py:163).  It decodes to NaN here so a test can see it. */
static const float k_u4_lut[16] = {1.0f, 0.8f, 0.6f, 0.4f, 0.0625f, 0.00390625f, 0.0000153f, 0.0f,
                                   -0.0000153f, -0.00390625f, -0.0625f, -0.4f, -0.6f, -0.8f, -1.0f,
                                   NAN};
const float *orc_u4_lut(void) { return k_u4_lut; }

void orc_decode_u4(const uint8_t *in, int64_t nbytes, float *out) {
    for (int64_t i = 0; i < nbytes; i++) {
        int lo = in[i] % 16;
        int hi = (in[i] - lo) / 16;
        out[2 * i + 0] = k_u4_lut[hi];
        out[2 * i + 1] = k_u4_lut[lo];
    }
}

/* mixed_precs_caching/evlfu_16.cpp:332-356.  The arithmetic mixes float and
 * double exactly as the C++ does: (float)v * 0.00002 is a double product. */
void orc_decode_u16(const uint16_t *in, int64_t n, float *out) {
    for (int64_t i = 0; i < n; i++) {
        uint16_t value = in[i];
        if (value > 65000) {
            float diff = ((float)(value - 65000)) / 100;
            if (value % 2 == 1)
                out[i] = (float)(-1 * (0.65 + diff));
            else
                out[i] = (float)(0.65 + diff);
        } else {
            out[i] = (float)((((float)value) * 0.00002) - 0.65);
        }
    }
}

/* ------------------------------------------------------------------------ */
/* a11: encoders, value -> code (script/reduce_precision.py)                 */
/* ------------------------------------------------------------------------ */

/* script/reduce_precision.py:270  round(((x + 1)/2) * 254), Python round = half-to-even */
int64_t orc_encode_u8(double x) { return (int64_t)nearbyint(((x + 1) / 2) * 254); }

/* script/reduce_precision.py:26-51 convert_ev_float_to_ushort (int() truncates toward zero) */
int64_t orc_encode_u16(double value) {
    if (value < -0.65) {
        int64_t leftover = (int64_t)(-100 * (0.65 + value));
        if (leftover % 2 == 0) leftover += 1;
        return 65000 + leftover;
    } else if (value > 0.65) {
        int64_t leftover = (int64_t)(100 * (value - 0.65));
        if (leftover % 2 == 1) leftover -= 1;
        return 65000 + leftover;
    }
    return (int64_t)((value + 0.65) / 1.3 * 65000);
}

/* script/reduce_precision.py:140-172 convert_to_4bit_int_posit */
int64_t orc_encode_u4(double v) {
    static const double pos[7] = {0.8, 0.6, 0.4, 0.25, 0.015, 0.00025, 0};
    static const double neg[7] = {-1, -0.8, -0.6, -0.4, -0.25, -0.015, -0.00025};
    if (v == 0) return 7;
    if (v > 0) {
        int code = 0;
        for (int i = 0; i < 7; i++) {
            if (v >= pos[i]) return code;
            code++;
        }
        return -1;
    }
    if (v >= neg[6]) return 8;
    int code = 15;
    for (int i = 0; i < 7; i++) {
        if (v < neg[i]) return code;
        code--;
    }
    return -1;
}

void orc_encode_u8_arr(const double *x, int64_t n, int64_t *out) { for (int64_t i = 0; i < n; i++) out[i] = orc_encode_u8(x[i]); }
void orc_encode_u16_arr(const double *x, int64_t n, int64_t *out) { for (int64_t i = 0; i < n; i++) out[i] = orc_encode_u16(x[i]); }
void orc_encode_u4_arr(const double *x, int64_t n, int64_t *out) { for (int64_t i = 0; i < n; i++) out[i] = orc_encode_u4(x[i]); }

/* ------------------------------------------------------------------------ */
/* a1: EmbeddingBag(mode="sum") for one table                                */
/* dlrm_s_pytorch.py:407-461 (per-table loop), :276 (nn.EmbeddingBag sum).   */
/* offsets hold bag STARTS only; the last bag runs to nnz.  per-row weights   */
/* (v_W_l[k].gather(0, idx), :426) multiply each row before the add.          */
/* Summation order: index order, fp32, multiply and add not fused.           */
/* codec: 32 (fp32 rows), 16, 8, 4 -- rows decoded then summed (a9/a10).      */
/* returns 0, or -1 on an out-of-range index / malformed offsets.            */
/* ------------------------------------------------------------------------ */
static void decode_row(const void *W, int codec, int d, int64_t row, float *tmp) {
    switch (codec) {
    case 32: memcpy(tmp, (const float *)W + row * d, sizeof(float) * (size_t)d); break;
    case 16: orc_decode_u16((const uint16_t *)W + row * d, d, tmp); break;
    case 8: orc_decode_u8((const uint8_t *)W + row * d, d, tmp); break;
    case 4: orc_decode_u4((const uint8_t *)W + row * (d / 2), d / 2, tmp); break;
    }
}

int orc_embedding_bag_sum(const void *W, int codec, int64_t n_rows, int d, const int64_t *idx,
                          int64_t nnz, const int64_t *off, int64_t B, const float *row_weights,
                          float *out /* B*d */) {
    float tmp[1024];
    if (d > 1024) return -2;
    for (int64_t b = 0; b < B; b++) {
        int64_t s = off[b], e = (b + 1 < B) ? off[b + 1] : nnz;
        if (s < 0 || e < s || e > nnz) return -1;
        float *o = out + b * d;
        for (int c = 0; c < d; c++) o[c] = 0.0f;
        for (int64_t j = s; j < e; j++) {
            int64_t r = idx[j];
            if (r < 0 || r >= n_rows) return -1;
            decode_row(W, codec, d, r, tmp);
            if (row_weights) {
                float w = row_weights[r];
                for (int c = 0; c < d; c++) {
                    float p = tmp[c] * w;
                    o[c] = o[c] + p;
                }
            } else {
                for (int c = 0; c < d; c++) o[c] = o[c] + tmp[c];
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* a3: interact_features, arch_interaction_op="dot"                          */
/* dlrm_s_pytorch.py:483-516: T=(B,F,d); Z=T.T^T; keep (i,j) with j<i        */
/* (j<=i when arch_interaction_itself) in row-major order; R=[x | Zflat].    */
/* Dot products are accumulated in double and rounded once: a rounding-       */
/* order-free reference; compare with rtol 1e-5 (BASELINE.json north_star).  */
/* ------------------------------------------------------------------------ */
void orc_interact_dot(const float *T, int64_t B, int F, int d, int itself, float *R) {
    int off = itself ? 1 : 0;
    int64_t P = 0;
    for (int i = 0; i < F; i++) P += i + off;
    for (int64_t b = 0; b < B; b++) {
        const float *t = T + b * (int64_t)F * d;
        float *r = R + b * (d + P);
        for (int c = 0; c < d; c++) r[c] = t[c];
        int64_t p = d;
        for (int i = 0; i < F; i++)
            for (int j = 0; j < i + off; j++) {
                double acc = 0;
                for (int c = 0; c < d; c++) acc += (double)t[i * d + c] * (double)t[j * d + c];
                r[p++] = (float)acc;
            }
    }
}

/* fp32 k-ordered fused chain: what v_mfma_f32_16x16x4_f32 computes when fed
 * k in natural order (MI355X guide, "FP32-input MFMA" numerics). Kept for
 * bounding the GPU kernel's rounding against the double version above. */
void orc_interact_dot_f32chain(const float *T, int64_t B, int F, int d, int itself, float *R) {
    int off = itself ? 1 : 0;
    int64_t P = 0;
    for (int i = 0; i < F; i++) P += i + off;
    for (int64_t b = 0; b < B; b++) {
        const float *t = T + b * (int64_t)F * d;
        float *r = R + b * (d + P);
        for (int c = 0; c < d; c++) r[c] = t[c];
        int64_t p = d;
        for (int i = 0; i < F; i++)
            for (int j = 0; j < i + off; j++) {
                float acc = 0;
                for (int c = 0; c < d; c++) acc = fmaf(t[i * d + c], t[j * d + c], acc);
                r[p++] = acc;
            }
    }
}

/* ------------------------------------------------------------------------ */
/* key -> entry hash map shared by the three policies                        */
/* keys: (table_id_1based << 32) | row  -- the reference's "T-R" strings      */
/* (cache_algo/EvLFU_C1.py:108) carry exactly this pair.                      */
/* ------------------------------------------------------------------------ */
typedef struct {
    uint64_t *keys; /* 0 = empty */
    int32_t *vals;
    uint64_t mask;
    int64_t count;
} orc_map;

static uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
static void map_init(orc_map *m, int64_t cap) {
    uint64_t n = 16;
    while ((int64_t)n < cap * 2 + 8) n <<= 1;
    m->keys = (uint64_t *)calloc(n, sizeof(uint64_t));
    m->vals = (int32_t *)malloc(n * sizeof(int32_t));
    m->mask = n - 1;
    m->count = 0;
}
static void map_free(orc_map *m) { free(m->keys); free(m->vals); }
static int32_t map_get(const orc_map *m, uint64_t key) {
    uint64_t i = mix64(key) & m->mask;
    while (m->keys[i]) {
        if (m->keys[i] == key) return m->vals[i];
        i = (i + 1) & m->mask;
    }
    return -1;
}
static void map_put(orc_map *m, uint64_t key, int32_t v) {
    uint64_t i = mix64(key) & m->mask;
    while (m->keys[i] && m->keys[i] != key) i = (i + 1) & m->mask;
    if (!m->keys[i]) m->count++;
    m->keys[i] = key;
    m->vals[i] = v;
}
static void map_del(orc_map *m, uint64_t key) { /* linear probing, backward-shift delete */
    uint64_t i = mix64(key) & m->mask;
    while (m->keys[i] && m->keys[i] != key) i = (i + 1) & m->mask;
    if (!m->keys[i]) return;
    m->count--;
    uint64_t j = i;
    for (;;) {
        j = (j + 1) & m->mask;
        if (!m->keys[j]) break;
        uint64_t h = mix64(m->keys[j]) & m->mask;
        /* can entry j move into hole i?  yes iff h is cyclically outside (i, j] */
        int between = (i <= j) ? (h > i && h <= j) : (h > i || h <= j);
        if (!between) {
            m->keys[i] = m->keys[j];
            m->vals[i] = m->vals[j];
            i = j;
        }
    }
    m->keys[i] = 0;
}

/* intrusive FIFO lists over entry indices (append tail / pop head / unlink) */
typedef struct { int32_t head, tail; int64_t len; } orc_list;
static void list_init(orc_list *l) { l->head = l->tail = -1; l->len = 0; }
static void list_append(orc_list *l, int32_t *prev, int32_t *next, int32_t e) {
    prev[e] = l->tail; next[e] = -1;
    if (l->tail >= 0) next[l->tail] = e; else l->head = e;
    l->tail = e; l->len++;
}
static void list_unlink(orc_list *l, int32_t *prev, int32_t *next, int32_t e) {
    if (prev[e] >= 0) next[prev[e]] = next[e]; else l->head = next[e];
    if (next[e] >= 0) prev[next[e]] = prev[e]; else l->tail = prev[e];
    l->len--;
}

/* miss source: in-memory tables (fp32, row-major, dim floats per row) = what
 * emb_storage/file_read.py:27-33 returns for (tableId,rowId). */
typedef struct {
    int n_tables, dim;
    const float *tables[ORC_MAX_TABLES];
} orc_store;

static void store_fetch(const orc_store *s, int table1, int64_t row, float *out) {
    memcpy(out, s->tables[table1 - 1] + row * s->dim, sizeof(float) * (size_t)s->dim);
}

/* ------------------------------------------------------------------------ */
/* a6: EvLFU  (cache_algo/EvLFU_C1.py:21-166)                                */
/*   flush_rate / perfect_item_cap: 0.3 / 0.95 in Python (:18-19) and in     */
/*   mixed_precs_caching (evlfu_8.hpp:50-51); 0.4 / 1.0 in the Cython build  */
/*   (cache_algo/EvLFU_C1_Cython/EvLFU.cpp:12-13).                           */
/*   flush_extra: Python flushes int(rate*cap)+1 keys (:40); the C++ flushes */
/*   int(rate*cap) (evlfu_8.cpp:256).                                        */
/* ------------------------------------------------------------------------ */
typedef struct {
    int64_t cap;
    int n_tables, dim;
    int min_c1;
    int64_t n_perfect, max_perfect;
    double flush_rate;
    int flush_extra;
    int perfect_mode; /* after a flush: 0 = recount len(bucket top) (EvLFU_C1.py:43); 1 = subtract int(rate*cap)
                         with n+1 keys flushed (EvLFU_C1_Cython/EvLFU.cpp:80-86); 2 = subtract int(rate*cap), n keys
                         flushed, stop silently at the end of the bucket (evlfu_8.cpp:256-270) */
    int64_t n_flush, n_evict;
    uint64_t last_evicted; /* key evicted (not flushed) by the latest evlfu_set, 0 if none */
    orc_map map;
    uint64_t *ekey;
    int32_t *eagg, *prev, *next;
    int32_t *free_stack;
    int64_t n_free;
    float *vals;
    orc_list lists[ORC_NBUCKET_MAX];
    orc_store store;
} orc_evlfu;

orc_evlfu *orc_evlfu_new(int64_t cap, int n_tables, int dim, double flush_rate,
                         double perfect_item_cap, int flush_extra, int perfect_mode) {
    if (n_tables > ORC_MAX_TABLES || cap < 1) return NULL;
    orc_evlfu *c = (orc_evlfu *)calloc(1, sizeof(orc_evlfu));
    c->cap = cap; c->n_tables = n_tables; c->dim = dim; c->min_c1 = 0;
    c->flush_rate = flush_rate; c->flush_extra = flush_extra; c->perfect_mode = perfect_mode;
    c->max_perfect = (int64_t)(cap * perfect_item_cap); /* EvLFU_C1.py:30 int(cap*0.95) */
    map_init(&c->map, cap);
    c->ekey = (uint64_t *)malloc(sizeof(uint64_t) * cap);
    c->eagg = (int32_t *)malloc(sizeof(int32_t) * cap);
    c->prev = (int32_t *)malloc(sizeof(int32_t) * cap);
    c->next = (int32_t *)malloc(sizeof(int32_t) * cap);
    c->free_stack = (int32_t *)malloc(sizeof(int32_t) * cap);
    c->vals = (float *)malloc(sizeof(float) * cap * dim);
    for (int64_t i = 0; i < cap; i++) c->free_stack[i] = (int32_t)(cap - 1 - i);
    c->n_free = cap;
    for (int b = 0; b <= n_tables; b++) list_init(&c->lists[b]);
    c->store.n_tables = n_tables; c->store.dim = dim;
    return c;
}
void orc_evlfu_free(orc_evlfu *c) {
    if (!c) return;
    map_free(&c->map);
    free(c->ekey); free(c->eagg); free(c->prev); free(c->next); free(c->free_stack); free(c->vals);
    free(c);
}
void orc_evlfu_set_tables(orc_evlfu *c, const float *const *tables) {
    for (int k = 0; k < c->n_tables; k++) c->store.tables[k] = tables[k];
}

static void evlfu_drop(orc_evlfu *c, int32_t e) { /* vals_C1.pop(key) */
    map_del(&c->map, c->ekey[e]);
    c->free_stack[c->n_free++] = e;
}

/* EvLFU_C1.py:32-63 set(key, value, agg_hit) */
static int evlfu_set(orc_evlfu *c, uint64_t key, const float *value, int agg_hit) {
    int top = c->n_tables;
    c->last_evicted = 0;
    if (c->n_perfect >= c->max_perfect) { /* :36-44 flush the oldest of bucket 26 */
        int64_t n = (int64_t)(c->flush_rate * c->cap) + c->flush_extra;
        for (int64_t i = 0; i < n; i++) {
            int32_t e = c->lists[top].head;
            if (e < 0) {
                if (c->perfect_mode == 0) return -1; /* Python would raise IndexError on pop(0) */
                break;
            }
            list_unlink(&c->lists[top], c->prev, c->next, e);
            evlfu_drop(c, e);
        }
        if (c->perfect_mode == 0) c->n_perfect = c->lists[top].len;
        else c->n_perfect -= (int64_t)(c->flush_rate * c->cap);
        c->n_flush++;
    } else if (c->map.count >= c->cap) { /* :47-56 evict FIFO-oldest of lowest non-empty bucket */
        while (c->lists[c->min_c1].len == 0) {
            c->min_c1 += 1;
            if (c->min_c1 > top) c->min_c1 = 1;
        }
        int32_t e = c->lists[c->min_c1].head;
        list_unlink(&c->lists[c->min_c1], c->prev, c->next, e);
        c->last_evicted = c->ekey[e];
        evlfu_drop(c, e);
        c->n_evict++;
    }
    if (c->n_free <= 0) return -2;
    int32_t existing = map_get(&c->map, key);
    int32_t e;
    if (existing >= 0) {
        /* vals_C1[key] = ... overwrites; lists_C1 would then hold the key twice.
         * Cannot happen through request(): a present key takes the hit branch. */
        return -3;
    }
    e = c->free_stack[--c->n_free];
    c->ekey[e] = key; c->eagg[e] = agg_hit;
    memcpy(c->vals + (int64_t)e * c->dim, value, sizeof(float) * (size_t)c->dim);
    map_put(&c->map, key, e);
    list_append(&c->lists[agg_hit], c->prev, c->next, e);
    if (agg_hit < c->min_c1) c->min_c1 = agg_hit; /* :62-63 */
    return 0;
}

/* EvLFU_C1.py:97-166 request_to_ev_lfu.  rows: n_tables int32 (0-based row ids,
 * table = position).  hit_out: n_tables bytes.  out: n_tables*dim floats.
 * approx_thres > 0 enables the approximate mode (:122-125,:142-152).
 * returns agg_hit, or < 0 on an internal error. */
int orc_evlfu_request(orc_evlfu *c, const int32_t *rows, uint8_t *hit_out, float *out,
                      int approx_thres) {
    int T = c->n_tables, dim = c->dim;
    int agg_hit = 0;
    uint64_t keys[ORC_MAX_TABLES];
    uint8_t hit[ORC_MAX_TABLES];
    for (int i = 0; i < T; i++) { /* :105-120 probe */
        keys[i] = ((uint64_t)(i + 1) << 32) | (uint32_t)rows[i];
        hit[i] = map_get(&c->map, keys[i]) >= 0;
        agg_hit += hit[i];
    }
    int pick_random = (approx_thres > 0 && agg_hit >= approx_thres);
    float miss_vals[ORC_MAX_TABLES][256];
    if (dim > 256) return -9;
    if (!pick_random) /* :128 fetch all missing at once */
        for (int i = 0; i < T; i++)
            if (!hit[i]) store_fetch(&c->store, i + 1, rows[i], miss_vals[i]);
    const float *last_val = NULL; /* random_ev_value: last hit's vector (:141) */
    for (int i = 0; i < T; i++) { /* :135-161 update */
        float *o = out + (int64_t)i * dim;
        if (hit[i]) {
            int32_t e = map_get(&c->map, keys[i]); /* update() -> update_agg_hit (:65-78) */
            if (e >= 0) {
                if (c->eagg[e] < agg_hit) {
                    list_unlink(&c->lists[c->eagg[e]], c->prev, c->next, e);
                    list_append(&c->lists[agg_hit], c->prev, c->next, e);
                    c->eagg[e] = agg_hit;
                }
                memcpy(o, c->vals + (int64_t)e * dim, sizeof(float) * (size_t)dim);
            } else { /* :90-94 kicked out while inserting a previous key: re-fetch + set */
                float tmp[256];
                store_fetch(&c->store, i + 1, rows[i], tmp);
                int rc = evlfu_set(c, keys[i], tmp, agg_hit);
                if (rc) return rc - 10;
                memcpy(o, tmp, sizeof(float) * (size_t)dim);
            }
            last_val = o;
        } else if (pick_random) { /* :142-152 miss served from the previous hit, reported as hit */
            if (last_val) memcpy(o, last_val, sizeof(float) * (size_t)dim);
            else memset(o, 0, sizeof(float) * (size_t)dim); /* reference: 36 random.uniform draws */
            hit[i] = 1;
        } else {
            /* update(key, ..., missing_value): a duplicate key inserted earlier in this same
             * request is found by update_agg_hit and returned without a second insert */
            int32_t e = map_get(&c->map, keys[i]);
            if (e >= 0) {
                if (c->eagg[e] < agg_hit) {
                    list_unlink(&c->lists[c->eagg[e]], c->prev, c->next, e);
                    list_append(&c->lists[agg_hit], c->prev, c->next, e);
                    c->eagg[e] = agg_hit;
                }
                memcpy(o, c->vals + (int64_t)e * dim, sizeof(float) * (size_t)dim);
            } else {
                int rc = evlfu_set(c, keys[i], miss_vals[i], agg_hit);
                if (rc) return rc - 20;
                memcpy(o, miss_vals[i], sizeof(float) * (size_t)dim);
            }
        }
    }
    if (agg_hit == T) c->n_perfect = c->lists[T].len; /* :163-165 */
    if (hit_out) memcpy(hit_out, hit, (size_t)T);
    return agg_hit;
}

/* state dump: triples (bucket, table1, row) in bucket order then FIFO order */
int64_t orc_evlfu_dump(const orc_evlfu *c, int64_t *out, int64_t max_triples) {
    int64_t n = 0;
    for (int b = 0; b <= c->n_tables; b++)
        for (int32_t e = c->lists[b].head; e >= 0; e = c->next[e]) {
            if (n < max_triples) {
                out[3 * n + 0] = b;
                out[3 * n + 1] = (int64_t)(c->ekey[e] >> 32);
                out[3 * n + 2] = (int64_t)(c->ekey[e] & 0xffffffffu);
            }
            n++;
        }
    return n;
}
/* [min_C1, n_perfect, len(vals), n_flush, n_evict] */
void orc_evlfu_state(const orc_evlfu *c, int64_t *out5) {
    out5[0] = c->min_c1; out5[1] = c->n_perfect; out5[2] = c->map.count; out5[3] = c->n_flush;
    out5[4] = c->n_evict;
}

/* ------------------------------------------------------------------------ */
/* a7: LRU  (cache_algo/LRU.py:14-64): per key, in table order: hit -> move  */
/* to MRU end; miss -> fetch, evict LRU head if len >= cap, insert at end.   */
/* ------------------------------------------------------------------------ */
typedef struct {
    int64_t cap; int n_tables, dim;
    orc_map map; uint64_t *ekey; int32_t *prev, *next, *free_stack; int64_t n_free;
    float *vals; orc_list order; orc_store store;
} orc_lru;

orc_lru *orc_lru_new(int64_t cap, int n_tables, int dim) {
    if (n_tables > ORC_MAX_TABLES || cap < 1) return NULL;
    orc_lru *c = (orc_lru *)calloc(1, sizeof(orc_lru));
    c->cap = cap; c->n_tables = n_tables; c->dim = dim;
    map_init(&c->map, cap);
    c->ekey = (uint64_t *)malloc(sizeof(uint64_t) * cap);
    c->prev = (int32_t *)malloc(sizeof(int32_t) * cap);
    c->next = (int32_t *)malloc(sizeof(int32_t) * cap);
    c->free_stack = (int32_t *)malloc(sizeof(int32_t) * cap);
    c->vals = (float *)malloc(sizeof(float) * cap * dim);
    for (int64_t i = 0; i < cap; i++) c->free_stack[i] = (int32_t)(cap - 1 - i);
    c->n_free = cap; list_init(&c->order);
    c->store.n_tables = n_tables; c->store.dim = dim;
    return c;
}
void orc_lru_free(orc_lru *c) {
    if (!c) return;
    map_free(&c->map); free(c->ekey); free(c->prev); free(c->next); free(c->free_stack); free(c->vals); free(c);
}
void orc_lru_set_tables(orc_lru *c, const float *const *tables) {
    for (int k = 0; k < c->n_tables; k++) c->store.tables[k] = tables[k];
}
int orc_lru_request(orc_lru *c, const int32_t *rows, uint8_t *hit_out, float *out) {
    int agg = 0;
    for (int i = 0; i < c->n_tables; i++) {
        uint64_t key = ((uint64_t)(i + 1) << 32) | (uint32_t)rows[i];
        float *o = out + (int64_t)i * c->dim;
        int32_t e = map_get(&c->map, key);
        if (e >= 0) { /* LRU.py:24-28 */
            list_unlink(&c->order, c->prev, c->next, e);
            list_append(&c->order, c->prev, c->next, e);
            memcpy(o, c->vals + (int64_t)e * c->dim, sizeof(float) * (size_t)c->dim);
            hit_out[i] = 1; agg++;
        } else { /* :30-34 -> set() :14-20 */
            store_fetch(&c->store, i + 1, rows[i], o);
            if (c->map.count >= c->cap) {
                int32_t v = c->order.head;
                list_unlink(&c->order, c->prev, c->next, v);
                map_del(&c->map, c->ekey[v]);
                c->free_stack[c->n_free++] = v;
            }
            e = c->free_stack[--c->n_free];
            c->ekey[e] = key;
            memcpy(c->vals + (int64_t)e * c->dim, o, sizeof(float) * (size_t)c->dim);
            map_put(&c->map, key, e);
            list_append(&c->order, c->prev, c->next, e);
            hit_out[i] = 0;
        }
    }
    return agg;
}
int64_t orc_lru_dump(const orc_lru *c, int64_t *out, int64_t max_pairs) {
    int64_t n = 0;
    for (int32_t e = c->order.head; e >= 0; e = c->next[e]) {
        if (n < max_pairs) { out[2 * n] = (int64_t)(c->ekey[e] >> 32); out[2 * n + 1] = (int64_t)(c->ekey[e] & 0xffffffffu); }
        n++;
    }
    return n;
}

/* ------------------------------------------------------------------------ */
/* a7: LFU  (cache_algo/LFU.py:12-95): FIFO within a frequency; a hit moves  */
/* the key from freq f to f+1; a miss evicts the head of least_freq's list.   */
/* ------------------------------------------------------------------------ */
typedef struct {
    int64_t cap; int n_tables, dim; int64_t least_freq;
    orc_map map; uint64_t *ekey; int64_t *efreq; int32_t *prev, *next, *free_stack; int64_t n_free;
    float *vals; orc_list *freq_lists; int64_t n_freq_lists; orc_store store;
} orc_lfu;

static void lfu_grow(orc_lfu *c, int64_t f) {
    if (f < c->n_freq_lists) return;
    int64_t n = c->n_freq_lists ? c->n_freq_lists : 16;
    while (n <= f) n *= 2;
    c->freq_lists = (orc_list *)realloc(c->freq_lists, sizeof(orc_list) * n);
    for (int64_t i = c->n_freq_lists; i < n; i++) list_init(&c->freq_lists[i]);
    c->n_freq_lists = n;
}
orc_lfu *orc_lfu_new(int64_t cap, int n_tables, int dim) {
    if (n_tables > ORC_MAX_TABLES || cap < 1) return NULL;
    orc_lfu *c = (orc_lfu *)calloc(1, sizeof(orc_lfu));
    c->cap = cap; c->n_tables = n_tables; c->dim = dim; c->least_freq = 1;
    map_init(&c->map, cap);
    c->ekey = (uint64_t *)malloc(sizeof(uint64_t) * cap);
    c->efreq = (int64_t *)malloc(sizeof(int64_t) * cap);
    c->prev = (int32_t *)malloc(sizeof(int32_t) * cap);
    c->next = (int32_t *)malloc(sizeof(int32_t) * cap);
    c->free_stack = (int32_t *)malloc(sizeof(int32_t) * cap);
    c->vals = (float *)malloc(sizeof(float) * cap * dim);
    for (int64_t i = 0; i < cap; i++) c->free_stack[i] = (int32_t)(cap - 1 - i);
    c->n_free = cap; lfu_grow(c, 2);
    c->store.n_tables = n_tables; c->store.dim = dim;
    return c;
}
void orc_lfu_free(orc_lfu *c) {
    if (!c) return;
    map_free(&c->map); free(c->ekey); free(c->efreq); free(c->prev); free(c->next);
    free(c->free_stack); free(c->vals); free(c->freq_lists); free(c);
}
void orc_lfu_set_tables(orc_lfu *c, const float *const *tables) {
    for (int k = 0; k < c->n_tables; k++) c->store.tables[k] = tables[k];
}
int orc_lfu_request(orc_lfu *c, const int32_t *rows, uint8_t *hit_out, float *out) {
    int agg = 0;
    for (int i = 0; i < c->n_tables; i++) {
        uint64_t key = ((uint64_t)(i + 1) << 32) | (uint32_t)rows[i];
        float *o = out + (int64_t)i * c->dim;
        int32_t e = map_get(&c->map, key);
        if (e >= 0) { /* LFU.py:53-60 -> _update :19-34 */
            int64_t f = c->efreq[e];
            list_unlink(&c->freq_lists[f], c->prev, c->next, e);
            if (c->freq_lists[c->least_freq].len == 0) c->least_freq += 1;
            c->efreq[e] = f + 1;
            lfu_grow(c, f + 1);
            list_append(&c->freq_lists[f + 1], c->prev, c->next, e);
            memcpy(o, c->vals + (int64_t)e * c->dim, sizeof(float) * (size_t)c->dim);
            hit_out[i] = 1; agg++;
        } else { /* :61-65 -> set() :36-51 */
            store_fetch(&c->store, i + 1, rows[i], o);
            if (c->map.count >= c->cap) {
                int32_t v = c->freq_lists[c->least_freq].head;
                if (v < 0) return -1;
                list_unlink(&c->freq_lists[c->least_freq], c->prev, c->next, v);
                map_del(&c->map, c->ekey[v]);
                c->free_stack[c->n_free++] = v;
            }
            e = c->free_stack[--c->n_free];
            c->ekey[e] = key; c->efreq[e] = 1;
            memcpy(c->vals + (int64_t)e * c->dim, o, sizeof(float) * (size_t)c->dim);
            map_put(&c->map, key, e);
            list_append(&c->freq_lists[1], c->prev, c->next, e);
            c->least_freq = 1;
            hit_out[i] = 0;
        }
    }
    return agg;
}
int64_t orc_lfu_dump(const orc_lfu *c, int64_t *out, int64_t max_triples) {
    int64_t n = 0;
    for (int64_t f = 1; f < c->n_freq_lists; f++)
        for (int32_t e = c->freq_lists[f].head; e >= 0; e = c->next[e]) {
            if (n < max_triples) { out[3 * n] = f; out[3 * n + 1] = (int64_t)(c->ekey[e] >> 32); out[3 * n + 2] = (int64_t)(c->ekey[e] & 0xffffffffu); }
            n++;
        }
    return n;
}

/* ------------------------------------------------------------------------ */
/* a9: two-tier request, C1 (main precision) + C2 (secondary precision)      */
/* mixed_precs_caching/evlfu_8.cpp:669-796 request_to_c1_c2, with            */
/* evlfu_4.cpp:374-425 phase_1 / phase_2 as the C2 half.                     */
/* Both tiers are orc_evlfu objects built with the "cpp" constants; their     */
/* tables hold the rows already decoded at that tier's precision.             */
/* Deviations from the C++ (documented in DESIGN.md 4)  : eviction victims are    */
/* FIFO-oldest (the C++ takes unordered_set::begin(), not reproducible), and  */
/* a C1 hit whose entry was evicted earlier in the same request is served     */
/* from storage instead of through the dangling pointer (evlfu_8.cpp:521-522).*/
/* tier_out[i]: 1 = C1 hit, 2 = C2 hit (served by C2), 0 = miss.              */
/* returns 1 for a perfect request (all keys in C1 or C2), 0 otherwise.       */
/* ------------------------------------------------------------------------ */
static void evlfu_touch(orc_evlfu *c, int32_t e, int agg_hit) { /* update_agg_hit, evlfu_8.cpp:303-321 */
    if (c->eagg[e] < agg_hit) {
        list_unlink(&c->lists[c->eagg[e]], c->prev, c->next, e);
        list_append(&c->lists[agg_hit], c->prev, c->next, e);
        c->eagg[e] = agg_hit;
    }
}

int orc_c1c2_request(orc_evlfu *c1, orc_evlfu *c2, const int32_t *rows, uint8_t *tier_out, float *out,
                     int high_agghit_threshold) {
    const int T = c1->n_tables, dim = c1->dim;
    uint64_t keys[ORC_MAX_TABLES];
    int hit1[ORC_MAX_TABLES], hit2[ORC_MAX_TABLES], upd2[ORC_MAX_TABLES], ins2[ORC_MAX_TABLES], job1[ORC_MAX_TABLES];
    int c2_agg = 0, c1_agg = 0, agg;
    float tmp[256];
    for (int i = 0; i < T; i++) {
        keys[i] = ((uint64_t)(i + 1) << 32) | (uint32_t)rows[i];
        hit2[i] = map_get(&c2->map, keys[i]) >= 0; /* evlfu_4.cpp phase_1_find_keys_in_cache */
        c2_agg += hit2[i];
    }
    agg = c2_agg;
    for (int i = 0; i < T; i++) { /* evlfu_8.cpp:690-712 */
        hit1[i] = map_get(&c1->map, keys[i]) >= 0;
        upd2[i] = 1; ins2[i] = 0; job1[i] = 0;
        if (hit1[i]) {
            c1_agg++;
            upd2[i] = 0;
            if (!hit2[i]) agg++;
        } else if (!hit2[i]) {
            ins2[i] = 1;
            upd2[i] = 0;
        }
    }
    int should_update_c2 = 1;
    if (c1->map.count >= c1->cap) { /* :721-738 C1 full: below the threshold C1 takes the odd double-misses */
        if (agg < high_agghit_threshold)
            for (int i = 0; i < T; i++)
                if (!hit2[i]) {
                    upd2[i] = 0;
                    if (i % 2 == 1) { job1[i] = 1; ins2[i] = 0; }
                }
    } else { /* :739-751 C1 not full: everything C1 misses goes to C1, C2 is left alone */
        for (int i = 0; i < T; i++) if (!hit1[i]) job1[i] = 1;
        should_update_c2 = 0;
        agg = c1_agg;
    }
    for (int i = 0; i < T; i++) tier_out[i] = hit1[i] ? 1 : (hit2[i] ? 2 : 0);
    if (should_update_c2) { /* evlfu_4.cpp phase_2_get_and_insert_missing_values */
        for (int i = 0; i < T; i++)
            if (upd2[i]) {
                int32_t e = map_get(&c2->map, keys[i]);
                evlfu_touch(c2, e, agg);
                memcpy(out + (int64_t)i * dim, c2->vals + (int64_t)e * dim, sizeof(float) * (size_t)dim);
            }
        for (int i = 0; i < T; i++)
            if (ins2[i]) {
                store_fetch(&c2->store, i + 1, rows[i], tmp);
                int rc = evlfu_set(c2, keys[i], tmp, agg);
                if (rc) return rc - 30;
                memcpy(out + (int64_t)i * dim, tmp, sizeof(float) * (size_t)dim);
            }
        if (agg == T) c2->n_perfect = c2->lists[T].len;
    }
    for (int i = 0; i < T; i++) { /* evlfu_8.cpp:769-785 */
        float *o = out + (int64_t)i * dim;
        if (hit1[i]) {
            int32_t e = map_get(&c1->map, keys[i]);
            if (e >= 0) {
                evlfu_touch(c1, e, agg);
                memcpy(o, c1->vals + (int64_t)e * dim, sizeof(float) * (size_t)dim);
            } else { /* evicted earlier in this request: the C++ reads a dangling pointer here */
                store_fetch(&c1->store, i + 1, rows[i], o);
            }
        } else if (job1[i]) {
            store_fetch(&c1->store, i + 1, rows[i], tmp);
            int rc = evlfu_set(c1, keys[i], tmp, agg);
            if (rc) return rc - 40;
            memcpy(o, tmp, sizeof(float) * (size_t)dim);
        }
    }
    if (agg == T) { c1->n_perfect = c1->lists[T].len; return 1; }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* a12: the alt-key ("approximate embedding") tier, C3                       */
/* mixed_precs_caching/aprx_embedding.cpp + evlfu_8.cpp:474-490,492-667.     */
/* DETERMINISTIC RE-SPECIFICATION -- parity UNPINNED: the reference fills    */
/* this tier from 5 asynchronous threads with no lock against the request    */
/* thread (aprx_embedding.cpp:90 vs :345) and its batch-insert loops use      */
/* uninitialised counters (:293,:298,:314,:319), so its observable behaviour  */
/* depends on thread timing.  Here:                                           */
/*   * keys EVICTED (not flushed) from C1 / C2 are queued (evlfu_8.cpp:284-287,*/
/*     :617-620,:654-658); when IO_JOB_Q_SIZE=50 are pending the batch is      */
/*     inserted at the START of the next request: n_erase = size+50-cap        */
/*     second-chance evictions, then 50 x (push FIFO, map[key]={alt,false})    */
/*     (aprx_embedding.cpp:308-324, :360-388);                                 */
/*   * alt key of (table,row) = the 4-byte big-endian word r of the table's     */
/*     alt-key file = alt_row*100 + alt_table (:243-251, :344-353);            */
/*   * on a double miss, if map[key] exists and the alt row is resident in C1   */
/*     or else C2, that row's vector is served (decoded at the precision of the */
/*     tier it was found in -- the C++ decodes a 4-bit buffer as 8-bit, an      */
/*     overread), the key's recency flag is set, the request's agg_hit counts   */
/*     it as a hit, and neither tier inserts or updates anything for it.        */
/* ------------------------------------------------------------------------ */
#define ORC_APRX_BATCH 50
typedef struct {
    int64_t cap, count, n_hit;
    orc_map map;          /* key -> entry */
    uint64_t *ekey; uint32_t *ealt; uint8_t *eflag; int32_t *free_stack; int64_t n_free;
    uint64_t *queue; int64_t qcap, qhead, qtail; /* FIFO of keys, may hold stale duplicates */
    uint64_t pending[ORC_APRX_BATCH]; int n_pending; int batch_ready;
    const uint32_t *alt_tables[ORC_MAX_TABLES];
    int error;
} orc_aprx;

orc_aprx *orc_aprx_new(int64_t cap, const uint32_t *const *alt_tables, int n_tables) {
    if (cap < ORC_APRX_BATCH) return NULL; /* aprx_embedding.cpp:33 assert(cap_C3 >= IO_JOB_Q_SIZE) */
    orc_aprx *x = (orc_aprx *)calloc(1, sizeof(orc_aprx));
    x->cap = cap;
    map_init(&x->map, cap);
    x->ekey = (uint64_t *)malloc(sizeof(uint64_t) * cap);
    x->ealt = (uint32_t *)malloc(sizeof(uint32_t) * cap);
    x->eflag = (uint8_t *)malloc((size_t)cap);
    x->free_stack = (int32_t *)malloc(sizeof(int32_t) * cap);
    for (int64_t i = 0; i < cap; i++) x->free_stack[i] = (int32_t)(cap - 1 - i);
    x->n_free = cap;
    x->qcap = 4 * cap + 64;
    x->queue = (uint64_t *)malloc(sizeof(uint64_t) * x->qcap);
    for (int k = 0; k < n_tables; k++) x->alt_tables[k] = alt_tables[k];
    return x;
}
void orc_aprx_free(orc_aprx *x) {
    if (!x) return;
    map_free(&x->map); free(x->ekey); free(x->ealt); free(x->eflag); free(x->free_stack); free(x->queue); free(x);
}
static void aprx_evict_one(orc_aprx *x) { /* recency_aware_eviction, aprx_embedding.cpp:360-388 */
    while (x->qhead < x->qtail) {
        uint64_t key = x->queue[x->qhead % x->qcap];
        int32_t e = map_get(&x->map, key);
        if (e >= 0) {
            if (x->eflag[e]) { /* second chance */
                x->eflag[e] = 0;
                x->queue[x->qtail % x->qcap] = key; x->qtail++;
                x->qhead++;
            } else {
                map_del(&x->map, key);
                x->free_stack[x->n_free++] = e;
                x->count--;
                x->qhead++;
                return;
            }
        } else {
            x->qhead++; /* stale duplicate */
        }
    }
}
static void aprx_insert_batch(orc_aprx *x) { /* insert_altkey_batched_obj, aprx_embedding.cpp:308-324 */
    int64_t n_erase = x->count + ORC_APRX_BATCH - x->cap;
    for (int64_t i = 0; i < n_erase; i++) aprx_evict_one(x);
    for (int i = 0; i < ORC_APRX_BATCH; i++) {
        uint64_t key = x->pending[i];
        int t = (int)(key >> 32) - 1;
        uint32_t alt = x->alt_tables[t][(uint32_t)(key & 0xffffffffu)];
        if (x->qtail - x->qhead >= x->qcap) { x->error = 1; return; }
        x->queue[x->qtail % x->qcap] = key; x->qtail++;
        int32_t e = map_get(&x->map, key);
        if (e < 0) {
            if (x->n_free <= 0) { x->error = 2; return; }
            e = x->free_stack[--x->n_free];
            x->ekey[e] = key;
            map_put(&x->map, key, e);
            x->count++;
        }
        x->ealt[e] = alt; x->eflag[e] = 0;
    }
    x->n_pending = 0; x->batch_ready = 0;
}
static void aprx_queue_key(orc_aprx *x, uint64_t key) {
    if (!key) return;
    if (x->n_pending < ORC_APRX_BATCH) x->pending[x->n_pending++] = key;
    /* a 51st key before the batch is inserted cannot happen: the batch is inserted at the next request start
       and one request evicts at most 2*T <= 128 keys -> keep the first 50, count the overflow as dropped */
    if (x->n_pending == ORC_APRX_BATCH) x->batch_ready = 1;
}
/* The tier's public single-key methods, for the ops that CAN be pinned to the reference (driven single-threaded by
 * oracle/ref/ref_aprx_driver.cpp): op 0 insert_altkey (aprx_embedding.cpp:278-288: evict one when full, push the key on
 * the FIFO -- duplicates allowed --, map[key] = {alt from the alt-key file, false}), 1 get_altkey_str (:341-350),
 * 2 set_recency_flag_c3 (:402-411), 3 evict_one_key (:390-400).  Same aprx_evict_one / map / FIFO as the request path.
 * ops: n x (op, table_1based, row); res: n x uint32 (op 0: the alt key; op 1: the alt key or 0xffffffff; else 0). */
void orc_aprx_apply_ops(orc_aprx *x, int64_t n, const int32_t *ops, uint32_t *res) {
    for (int64_t i = 0; i < n; i++) {
        const int op = ops[3 * i];
        const uint64_t key = ((uint64_t)(uint32_t)ops[3 * i + 1] << 32) | (uint32_t)ops[3 * i + 2];
        res[i] = 0;
        if (op == 0) {
            const uint32_t alt = x->alt_tables[ops[3 * i + 1] - 1][(uint32_t)ops[3 * i + 2]];
            if (x->count >= x->cap) aprx_evict_one(x);
            if (x->qtail - x->qhead >= x->qcap) { x->error = 1; return; }
            x->queue[x->qtail % x->qcap] = key; x->qtail++;
            int32_t e = map_get(&x->map, key);
            if (e < 0) {
                if (x->n_free <= 0) { x->error = 2; return; }
                e = x->free_stack[--x->n_free];
                x->ekey[e] = key;
                map_put(&x->map, key, e);
                x->count++;
            }
            x->ealt[e] = alt; x->eflag[e] = 0;
            res[i] = alt;
        } else if (op == 1) {
            const int32_t e = map_get(&x->map, key);
            res[i] = e >= 0 ? x->ealt[e] : 0xffffffffu;
        } else if (op == 2) {
            const int32_t e = map_get(&x->map, key);
            if (e >= 0) x->eflag[e] = 1;
        } else if (op == 3) {
            aprx_evict_one(x);
        }
    }
}
/* the FIFO front to back as (table_1based, row) pairs, stale duplicates included (print_all_keys_in_c3, :430-434) */
int64_t orc_aprx_dump_queue(const orc_aprx *x, int64_t *pairs, int64_t max_pairs) {
    int64_t n = 0;
    for (int64_t q = x->qhead; q < x->qtail; q++, n++)
        if (pairs && n < max_pairs) {
            pairs[2 * n] = (int64_t)(x->queue[q % x->qcap] >> 32);
            pairs[2 * n + 1] = (int64_t)(x->queue[q % x->qcap] & 0xffffffffu);
        }
    return n;
}
void orc_aprx_state(const orc_aprx *x, int64_t *out4) { out4[0] = x->count; out4[1] = x->n_hit; out4[2] = x->n_pending; out4[3] = x->error; }

/* request_to_c1_c2_c3 (evlfu_8.cpp:492-667).  tier_out: 1 C1 hit, 2 C2 hit, 3 alt-key hit, 0 miss. */
int orc_c1c2c3_request(orc_evlfu *c1, orc_evlfu *c2, orc_aprx *c3, const int32_t *rows, uint8_t *tier_out, float *out,
                       int high_agghit_threshold) {
    const int T = c1->n_tables, dim = c1->dim;
    uint64_t keys[ORC_MAX_TABLES];
    int hit1[ORC_MAX_TABLES], hit2[ORC_MAX_TABLES], hit3[ORC_MAX_TABLES], upd2[ORC_MAX_TABLES], ins2[ORC_MAX_TABLES],
        job1[ORC_MAX_TABLES], alt_tier[ORC_MAX_TABLES];
    int32_t alt_e[ORC_MAX_TABLES];
    int c2_agg = 0, c1_agg = 0, agg;
    float tmp[256];
    if (c3->batch_ready) aprx_insert_batch(c3);
    for (int i = 0; i < T; i++) {
        keys[i] = ((uint64_t)(i + 1) << 32) | (uint32_t)rows[i];
        hit2[i] = map_get(&c2->map, keys[i]) >= 0;
        c2_agg += hit2[i];
    }
    agg = c2_agg;
    for (int i = 0; i < T; i++) {
        hit1[i] = map_get(&c1->map, keys[i]) >= 0;
        hit3[i] = 0; upd2[i] = 1; ins2[i] = 0; job1[i] = 0; alt_tier[i] = 0; alt_e[i] = -1;
        if (hit1[i]) {
            c1_agg++;
            upd2[i] = 0;
            if (!hit2[i]) agg++;
        } else if (!hit2[i]) {
            upd2[i] = 0;
            int32_t e3 = map_get(&c3->map, keys[i]); /* find_approximate_ev, evlfu_8.cpp:474-490 */
            if (e3 >= 0) {
                uint32_t alt = c3->ealt[e3];
                uint64_t akey = ((uint64_t)(alt % 100) << 32) | (uint32_t)(alt / 100);
                int32_t ea = map_get(&c1->map, akey);
                if (ea >= 0) { alt_tier[i] = 1; alt_e[i] = ea; }
                else { ea = map_get(&c2->map, akey); if (ea >= 0) { alt_tier[i] = 2; alt_e[i] = ea; } }
            }
            if (alt_tier[i]) {
                hit3[i] = 1;
                c3->eflag[e3] = 1; /* set_recency_flag_c3 */
                c3->n_hit++;
                agg++;
            } else {
                ins2[i] = 1;
            }
        }
    }
    int should_update_c2 = 1;
    if (c1->map.count >= c1->cap) {
        if (agg < high_agghit_threshold)
            for (int i = 0; i < T; i++)
                if (!hit2[i]) {
                    upd2[i] = 0;
                    if (i % 2 == 1) { job1[i] = !(hit1[i] || hit3[i]); ins2[i] = 0; }
                }
    } else {
        for (int i = 0; i < T; i++) if (!(hit1[i] || hit3[i])) job1[i] = 1;
        should_update_c2 = 0;
        agg = c1_agg;
    }
    for (int i = 0; i < T; i++) tier_out[i] = hit1[i] ? 1 : (hit2[i] ? 2 : (hit3[i] ? 3 : 0));
    /* alt-key rows are read now: the request's own evictions cannot invalidate what it serves */
    for (int i = 0; i < T; i++)
        if (hit3[i]) {
            const orc_evlfu *src = alt_tier[i] == 1 ? c1 : c2;
            memcpy(out + (int64_t)i * dim, src->vals + (int64_t)alt_e[i] * dim, sizeof(float) * (size_t)dim);
        }
    if (should_update_c2) {
        for (int i = 0; i < T; i++)
            if (upd2[i] && hit2[i]) {
                int32_t e = map_get(&c2->map, keys[i]);
                evlfu_touch(c2, e, agg);
                memcpy(out + (int64_t)i * dim, c2->vals + (int64_t)e * dim, sizeof(float) * (size_t)dim);
            }
        for (int i = 0; i < T; i++)
            if (ins2[i]) {
                store_fetch(&c2->store, i + 1, rows[i], tmp);
                int rc = evlfu_set(c2, keys[i], tmp, agg);
                if (rc) return rc - 30;
                aprx_queue_key(c3, c2->last_evicted); /* evlfu_8.cpp:617-620 */
                memcpy(out + (int64_t)i * dim, tmp, sizeof(float) * (size_t)dim);
            }
        if (agg == T) c2->n_perfect = c2->lists[T].len;
    }
    for (int i = 0; i < T; i++) {
        float *o = out + (int64_t)i * dim;
        if (hit1[i]) {
            int32_t e = map_get(&c1->map, keys[i]);
            if (e >= 0) {
                evlfu_touch(c1, e, agg);
                memcpy(o, c1->vals + (int64_t)e * dim, sizeof(float) * (size_t)dim);
            } else {
                store_fetch(&c1->store, i + 1, rows[i], o);
            }
        } else if (job1[i]) {
            store_fetch(&c1->store, i + 1, rows[i], tmp);
            int rc = evlfu_set(c1, keys[i], tmp, agg);
            if (rc) return rc - 40;
            aprx_queue_key(c3, c1->last_evicted); /* evlfu_8.cpp:654-658 */
            memcpy(o, tmp, sizeof(float) * (size_t)dim);
        }
    }
    if (agg == T) { c1->n_perfect = c1->lists[T].len; return 1; }
    return 0;
}
