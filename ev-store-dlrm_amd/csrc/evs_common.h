// Internal helpers shared by the kernels of libevstore_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/evstore_hip.h"

namespace evs {

constexpr int kWave = 64;        // CDNA wavefront
constexpr int kNumXcd = 8;       // MI355X: 8 XCDs, blocks are dealt round-robin over them
constexpr int kNumCu = 256;

void set_error(const char *fmt, ...);

#define EVS_HIP_CHECK(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            evs::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return EVS_EHIP;                                                             \
        }                                                                                \
    } while (0)

#define EVS_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            evs::set_error(__VA_ARGS__);       \
            return EVS_EINVAL;                 \
        }                                      \
    } while (0)

// sticky device-side flag for out-of-range indices (one per device)
int *index_error_flag();
// 4 KiB of zeros in HBM (one per device): the "row" read by lanes that have nothing to fetch
const void *zero_page();
// 1 KiB of the code that decodes to exactly 0.0f (one per device and codec): u16 32 500, u8 127, u4 nibble 7
const void *zero_code_page(int codec);
int *optimistic_slot(int *id_out);   // evs_api.hip

// ---- codecs: bit-exact device restatements of the reference decoders ----------------
// 8-bit: mixed_precs_caching/evlfu_8.cpp:370-378, all fp32, division kept as a division
__device__ __forceinline__ float dec_u8(unsigned v) {
    float f = (float)v;
    f = __fdiv_rn(f, 254.0f);
    f = __fmul_rn(f, 2.0f);
    return __fsub_rn(f, 1.0f);
}
// 16-bit: mixed_precs_caching/evlfu_16.cpp:332-356 -- the reference mixes float and double: (float)v * 0.00002 is a
// double product, 0.65 is subtracted in double, the result is rounded to float.  For the main range (v <= 65 000) that is
// bit for bit  fma(w, ch, w * cl)  in fp32 with w = v - 32 500, ch = fl32(0.00002), cl = fl32(0.00002 - ch): 0.65 is
// 32 500 x 0.00002 up to 1e-17, far below half a float ulp of every non-zero result, and the three fp32 roundings land
// on the same float as the three of the reference -- checked for all 65 001 codes against the double formula with exact
// rational arithmetic before it went in, and by tests/test_gpu_parity.py::test_exhaustive_decode_tables_on_gpu against the
// compiled reference decoder on every run.  Four fp32 instructions instead of two conversions and two fp64 operations
// per element (fp64 and the conversions run at a half / a quarter of the fp32 rate).
__device__ __forceinline__ float dec_u16_main(unsigned v) {
    const float w = (float)((int)v - 32500);
    return __fmaf_rn(w, 2e-05f, __fmul_rn(w, 5.052425e-13f));
}
__device__ __forceinline__ float dec_u16(unsigned v) {
    if (v > 65000u) {
        float diff = __fdiv_rn((float)(int)(v - 65000u), 100.0f);
        double m = __dadd_rn(0.65, (double)diff);
        return (v & 1u) ? (float)(-m) : (float)m;
    }
    return dec_u16_main(v);
}
// 4-bit look-up table (mixed_precs_caching/evlfu_4.hpp:46); entry 15 is out of bounds in
// the reference and never emitted by its encoder: it decodes to NaN here.
__device__ __constant__ const float kU4Lut[16] = {
    1.0f, 0.8f, 0.6f, 0.4f, 0.0625f, 0.00390625f, 0.0000153f, 0.0f,
    -0.0000153f, -0.00390625f, -0.0625f, -0.4f, -0.6f, -0.8f, -1.0f, __builtin_nanf("")};

// the same 16 values without a memory access (selects on immediates): what the table fills of the kernels use -- a per-lane
// read of kU4Lut in a kernel's head is a vector-memory round trip in front of everything else, in every block
__device__ __forceinline__ float u4_value(unsigned c) {
    const unsigned m = c > 7u ? 14u - c : c;          // magnitude index 0..6 (7: zero; c = 15 -> wraps: NaN below)
    float v = 0.0f;
    v = m == 0u ? 1.0f : v; v = m == 1u ? 0.8f : v; v = m == 2u ? 0.6f : v; v = m == 3u ? 0.4f : v;
    v = m == 4u ? 0.0625f : v; v = m == 5u ? 0.00390625f : v; v = m == 6u ? 0.0000153f : v;
    v = c > 7u ? -v : v;
    return c == 15u ? __builtin_nanf("") : (c == 7u ? 0.0f : v);
}

// Decode through a per-block LDS table: u8 -> all 256 values (the exact expression above has an
// IEEE division per element), u4 -> the 16 entries (a per-lane index into __constant__ memory is a
// vector-memory gather), u16 -> the 536 magnitudes of the v > 65000 tail (the division again); the
// u16 main range stays arithmetic (dec_u16_main: four fp32 instructions).
template <int CODEC> struct CodecLut { static constexpr int kEntries = 1; };
template <> struct CodecLut<16> { static constexpr int kEntries = 536; };
template <> struct CodecLut<8> { static constexpr int kEntries = 256; };
template <> struct CodecLut<4> { static constexpr int kEntries = 16; };

// every thread of the block calls this, then __syncthreads()
template <int CODEC>
__device__ __forceinline__ void codec_lut_init(float *lut) {
    for (int i = threadIdx.x; i < CodecLut<CODEC>::kEntries; i += blockDim.x) {
        if constexpr (CODEC == 8) lut[i] = dec_u8((unsigned)i);
        else if constexpr (CODEC == 4) lut[i] = u4_value((unsigned)i);
        else if constexpr (CODEC == 16) lut[i] = (float)__dadd_rn(0.65, (double)__fdiv_rn((float)i, 100.0f));
        else lut[i] = 0.f;
    }
}
template <int CODEC>
__device__ __forceinline__ float dec_code(unsigned v, const float *lut) {
    if constexpr (CODEC == 16) {
        const bool tail = v > 65000u;
        const float t = lut[tail ? v - 65000u : 0u];
        const float m = dec_u16_main(v);
        return tail ? ((v & 1u) ? -t : t) : m;
    } else {
        return lut[v];
    }
}
// one chunk = 4 consecutive elements: 8 / 4 / 2 raw bytes in w0 (and w1 for u16)
template <int CODEC>
__device__ __forceinline__ float4 dec_chunk(unsigned w0, unsigned w1, const float *lut) {
    if constexpr (CODEC == 16) {
        // tail codes (> 65 000: |x| > 0.65) are rare in trained tables: when no lane of the wave holds one in this chunk --
        // a wave-uniform test -- the table read and the selects of the tail are skipped for the whole wave
        const unsigned a = w0 & 0xffffu, b = w0 >> 16, c = w1 & 0xffffu, e = w1 >> 16;
        const bool tail = a > 65000u || b > 65000u || c > 65000u || e > 65000u;
        if (__builtin_amdgcn_ballot_w64(tail) == 0ull)
            return make_float4(dec_u16_main(a), dec_u16_main(b), dec_u16_main(c), dec_u16_main(e));
        return make_float4(dec_code<16>(a, lut), dec_code<16>(b, lut), dec_code<16>(c, lut), dec_code<16>(e, lut));
    } else if constexpr (CODEC == 8) {
        return make_float4(lut[w0 & 0xffu], lut[(w0 >> 8) & 0xffu], lut[(w0 >> 16) & 0xffu], lut[(w0 >> 24) & 0xffu]);
    } else {  // u4: element 2j is the HIGH nibble of byte j (script/reduce_precision.py:321)
        return make_float4(lut[(w0 >> 4) & 15u], lut[w0 & 15u], lut[(w0 >> 12) & 15u], lut[(w0 >> 8) & 15u]);
    }
}

// XCD-aware split of `n_items` (ordered so that neighbours share data, e.g. table-major)
// over the grid: blocks b and b+8 share an XCD (and its 4 MiB L2), so XCD x owns the
// contiguous item range [x*n/8, (x+1)*n/8) and its blocks stride through it.
struct XcdRange {
    int64_t begin, end, stride, first;
};
__device__ __forceinline__ XcdRange xcd_range(int64_t n_items, int units_per_block, int unit_in_block) {
    const int nb = gridDim.x;
    const int bid = blockIdx.x;
    XcdRange r;
    if (nb % kNumXcd == 0) {
        const int xcd = bid % kNumXcd, local = bid / kNumXcd, per = nb / kNumXcd;
        r.begin = n_items * xcd / kNumXcd;
        r.end = n_items * (xcd + 1) / kNumXcd;
        r.stride = (int64_t)per * units_per_block;
        r.first = r.begin + (int64_t)local * units_per_block + unit_in_block;
    } else {
        r.begin = 0;
        r.end = n_items;
        r.stride = (int64_t)nb * units_per_block;
        r.first = (int64_t)bid * units_per_block + unit_in_block;
    }
    return r;
}

// evs_fused.hip: interaction over x + T features given as absolute fp32 row addresses (cache tier)
int fused_interact_from_row_ptrs(int64_t B, int T, int d, const float *x, int64_t x_stride, const int64_t *row_ptrs,
                                 const int64_t *iota, int itself, float *R, hipStream_t st);

// evs_fused.hip / evs_fused_rf.hip: the same over ONE (B,T) table of 32-bit row ids -- bit 30 set: row (id & 0x3fffffff)
// of `arena`, clear: row id of table k, -1: the zero row (fp32 rows; is there a kernel: fused_row_ids_supported)
bool fused_row_ids_supported(int64_t B, int T, int d);
int fused_interact_from_row_ids(int64_t B, int T, int d, const float *x, int64_t x_stride, const int *row_ids,
                                const void *arena, const void *const *tables, int itself, float *R, hipStream_t st);

// evs_mixed.hip: interaction over x + T rows given as (address, codec class) pairs, decoded on the fly
int interact_from_mixed_rows(long long B, int T, int d, const float *x, long long x_stride, const long long *row_ptrs,
                             const unsigned char *row_class, int codec1, int codec2, int itself, float *R, hipStream_t st,
                             int default_class = 1);   // row_class == NULL: every row of this class (1 = codec1, 2 = codec2)

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// evs_filetier.hip (accessors for evs_cache.hip)
}  // namespace evs
struct evs_filetier;
namespace evs {
int filetier_tables(const evs_filetier *ft);
long long filetier_row_bytes(const evs_filetier *ft);
long long filetier_rows(const evs_filetier *ft, int k);
const void *filetier_dev(const evs_filetier *ft, int k);

}  // namespace evs
