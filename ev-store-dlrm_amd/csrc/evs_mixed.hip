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
#include "evs_common.h"

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
                const int cls = args.row_class ? args.row_class[b * T + (r - 1)] : 1;
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
                             const unsigned char *row_class, int codec1, int codec2, int itself, float *R, hipStream_t st) {
    MixedArgs a;
    const int F = T + 1;
    a.x = x; a.x_stride = x_stride; a.row_ptrs = row_ptrs; a.row_class = row_class; a.R = R; a.B = B; a.T = T;
    a.itself = itself ? 1 : 0; a.P = itself ? F * (F + 1) / 2 : F * (F - 1) / 2; a.codec1 = codec1; a.codec2 = codec2;
    const bool nt2 = F > 16;
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
