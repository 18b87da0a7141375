// Offline encoders as a GPU batch tool: fp32 tables -> the reference's reduced-precision
// on-disk row formats, bit-exact with script/reduce_precision.py + script/convert_ev_to_binary.py:
//   u8   round(((x + 1) / 2) * 254), Python round = half-to-even            (reduce_precision.py:270)
//   u16  convert_ev_float_to_ushort: linear code inside +-0.65, 0.01 steps with a sign-parity
//        convention outside, int() truncates toward zero                     (reduce_precision.py:26-51)
//   u4   convert_to_4bit_int_posit thresholds, dim 2j in the HIGH nibble    (reduce_precision.py:140-172,321)
// The reference computes in Python floats (doubles) from the CSV text of the fp32 weights; the kernels widen
// the fp32 input to fp64 and do the same IEEE arithmetic (the library is built with -ffp-contract=off).
// Codes are stored like numpy's astype(uint8/uint16): modulo 2^8 / 2^16.
#include "evs_common.h"

namespace evs {

__device__ __forceinline__ long long enc_u8(double x) { return (long long)rint(((x + 1.0) / 2.0) * 254.0); }

__device__ __forceinline__ long long enc_u16(double value) {
    if (value < -0.65) {
        long long leftover = (long long)(-100.0 * (0.65 + value));
        if (leftover % 2 == 0) leftover += 1;
        return 65000 + leftover;
    } else if (value > 0.65) {
        long long leftover = (long long)(100.0 * (value - 0.65));
        if (leftover % 2 == 1) leftover -= 1;
        return 65000 + leftover;
    }
    return (long long)((value + 0.65) / 1.3 * 65000.0);
}

__device__ __forceinline__ int enc_u4(double v) {
    if (v == 0.0) return 7;
    if (v > 0.0) {
        if (v >= 0.8) return 0;
        if (v >= 0.6) return 1;
        if (v >= 0.4) return 2;
        if (v >= 0.25) return 3;
        if (v >= 0.015) return 4;
        if (v >= 0.00025) return 5;
        return 6;
    }
    if (v >= -0.00025) return 8;
    if (v < -1.0) return 15;
    if (v < -0.8) return 14;
    if (v < -0.6) return 13;
    if (v < -0.4) return 12;
    if (v < -0.25) return 11;
    if (v < -0.015) return 10;
    return 9;
}

template <int CODEC>
__global__ void __launch_bounds__(256) encode_table_kernel(const float *__restrict__ src, unsigned char *__restrict__ dst,
                                                           long long n_units) {
    // one unit = one output code (u8, u16) or one output byte = two codes (u4)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_units; i += (long long)gridDim.x * blockDim.x) {
        if constexpr (CODEC == 8) {
            dst[i] = (unsigned char)enc_u8((double)src[i]);
        } else if constexpr (CODEC == 16) {
            reinterpret_cast<unsigned short *>(dst)[i] = (unsigned short)enc_u16((double)src[i]);
        } else {
            const int hi = enc_u4((double)src[2 * i]), lo = enc_u4((double)src[2 * i + 1]);
            dst[i] = (unsigned char)(hi * 16 + lo);
        }
    }
}

}  // namespace evs

extern "C" int evs_encode_table(int codec, int64_t n_rows, int d, const float *src, void *dst, void *stream) {
    using namespace evs;
    EVS_REQUIRE(codec == 16 || codec == 8 || codec == 4, "evs_encode_table: codec %d (16, 8 or 4)", codec);
    EVS_REQUIRE(n_rows >= 0 && d >= 1 && (codec != 4 || d % 2 == 0), "evs_encode_table: n_rows=%lld d=%d", (long long)n_rows, d);
    if (n_rows == 0) return EVS_OK;
    EVS_REQUIRE(src && dst, "evs_encode_table: NULL argument");
    const long long n_units = codec == 4 ? n_rows * (long long)d / 2 : n_rows * (long long)d;
    long long nb = (n_units + 255) / 256;
    if (nb > kNumCu * 16) nb = kNumCu * 16;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    unsigned char *out = static_cast<unsigned char *>(dst);
    if (codec == 8) hipLaunchKernelGGL(encode_table_kernel<8>, dim3((unsigned)nb), dim3(256), 0, st, src, out, n_units);
    else if (codec == 16) hipLaunchKernelGGL(encode_table_kernel<16>, dim3((unsigned)nb), dim3(256), 0, st, src, out, n_units);
    else hipLaunchKernelGGL(encode_table_kernel<4>, dim3((unsigned)nb), dim3(256), 0, st, src, out, n_units);
    EVS_HIP_CHECK(hipGetLastError());
    return EVS_OK;
}
