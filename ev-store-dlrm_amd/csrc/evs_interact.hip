// interact_features ("dot") on the gfx950 matrix cores.
//
// Replaces dlrm_s_pytorch.py:483-516: cat -> bmm(T, T^T) -> strict-lower-triangle
// gather (index tensors rebuilt on the host every call) -> cat, by ONE kernel that
// reads each sample's F feature rows once and writes the (d + P) output row once.
//
// One wavefront owns one sample at a time.  Z = T.T^T for F <= 32 is at most a 2x2
// grid of 16x16 tiles of which only the lower three are needed; each tile is a chain
// of d/4 v_mfma_f32_16x16x4_f32 (exact fp32: a k-ordered fmaf chain, see the MI355X
// guide "FP32-input MFMA").  The MFMA sums over k; the ORDER of k does not matter
// for the operand layout as long as A and B agree, so lane (row r, k-slot q) holds
// the KS = d/4 CONTIGUOUS elements T[r][q*KS .. q*KS+KS) -- one contiguous piece of
// the row per lane instead of a stride-4 column walk -- and MFMA step s multiplies
// elements {s, KS+s, 2KS+s, 3KS+s}.
//
// Roofline: 2*F*F*d flop vs 4*F*d + 4*(d+P) bytes per sample (52 488 flop vs 5 436 B
// at F=27, d=36): below the fp32-matrix ridge, so HBM-bound; the kernel exists to
// touch every byte exactly once, the MFMA keeps the VALU free for addressing.
#include "evs_common.h"

namespace evs {

struct InteractArgs {
    const float *feat[EVS_MAX_FEATURES];
    int64_t stride[EVS_MAX_FEATURES];
    float *R;
    int64_t B;
    int F, d, itself, P;
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KS>
__device__ __forceinline__ void load_piece(const float *__restrict__ p, float (&v)[KS]) {
    if constexpr (KS % 4 == 0) {
#pragma unroll
        for (int i = 0; i < KS / 4; i++) {
            const float4 q = reinterpret_cast<const float4 *>(p)[i];
            v[4 * i + 0] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < KS; i++) v[i] = p[i];
    }
}

// NT = 1: F <= 16 (one tile); NT = 2: F <= 32 (tiles (0,0), (1,0), (1,1)).
template <int KS, int NT>
__global__ void __launch_bounds__(256) interact_dot_kernel(const InteractArgs args) {
    const int lane = threadIdx.x & (kWave - 1);
    const int r16 = lane & 15;  // row inside a 16-row tile (A operand) = column (B operand)
    const int q = lane >> 4;    // k-slot 0..3
    const int F = args.F, d = KS * 4, itself = args.itself;
    const int64_t out_row = (int64_t)d + args.P;

    // per-lane feature pointers: dynamic index into the kernarg struct (read-only memory)
    const InteractArgs *ka = (const InteractArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    const float *p0 = nullptr, *p1 = nullptr;
    int64_t s0 = 0, s1 = 0;
    if (r16 < F) { p0 = ka->feat[r16] + q * KS; s0 = ka->stride[r16]; }
    if (NT == 2 && r16 + 16 < F) { p1 = ka->feat[r16 + 16] + q * KS; s1 = ka->stride[r16 + 16]; }

    const int64_t waves_total = (int64_t)gridDim.x * (blockDim.x / kWave);
    const int64_t wave_id = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;

    for (int64_t b = wave_id; b < args.B; b += waves_total) {
        float a0[KS], a1[KS];
#pragma unroll
        for (int i = 0; i < KS; i++) { a0[i] = 0.f; a1[i] = 0.f; }
        if (p0) load_piece<KS>(p0 + b * s0, a0);
        if (NT == 2 && p1) load_piece<KS>(p1 + b * s1, a1);

        f32x4 c00 = {0.f, 0.f, 0.f, 0.f}, c10 = {0.f, 0.f, 0.f, 0.f}, c11 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; s++) {
            c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], a0[s], c00, 0, 0, 0);
            if (NT == 2) {
                c10 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], a0[s], c10, 0, 0, 0);
                c11 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], a1[s], c11, 0, 0, 0);
            }
        }

        float *__restrict__ R = args.R + b * out_row;
        // x passthrough: feature 0 lives in the four lanes with r16 == 0
        if (r16 == 0) {
#pragma unroll
            for (int i = 0; i < KS; i++) R[q * KS + i] = a0[i];
        }
        // C/D layout of 16x16 tiles: row i = 4*(lane>>4) + v, col j = lane & 15
        float *__restrict__ Z = R + d;
        const int j0 = r16;
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const int i = 4 * q + v;
            if (i < F && j0 < i + itself) Z[(i * (i - 1 + 2 * itself)) / 2 + j0] = c00[v];
            if (NT == 2) {
                const int gi = 16 + i;
                if (gi < F) {
                    const int base = (gi * (gi - 1 + 2 * itself)) / 2;
                    Z[base + j0] = c10[v];                                            // cols 0..15 < gi
                    if (16 + j0 < gi + itself) Z[base + 16 + j0] = c11[v];            // cols 16..31
                }
            }
        }
    }
}

// Generic fallback (F > 32 or d % 4 != 0 or d > 256): one (sample, pair) per thread.
__global__ void __launch_bounds__(256) interact_dot_generic_kernel(const float *const *feat,
                                                                   const int64_t *stride, float *R, int64_t B,
                                                                   int F, int d, int itself, int P) {
    const int64_t n = B * (int64_t)(d + P);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = e / (d + P);
        const int c = (int)(e - b * (d + P));
        if (c < d) { R[e] = feat[0][b * stride[0] + c]; continue; }
        int p = c - d, i = itself ? 0 : 1;
        while (p >= i + itself) { p -= i + itself; i++; }  // row i holds i+itself pairs
        const float *x = feat[i] + b * stride[i], *y = feat[p] + b * stride[p];
        float acc = 0.f;
        for (int k = 0; k < d; k++) acc = fmaf(x[k], y[k], acc);
        R[e] = acc;
    }
}

__global__ void __launch_bounds__(256) interact_cat_kernel(const InteractArgs args) {
    const InteractArgs *ka = (const InteractArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    const int d = args.d, F = args.F;
    const int64_t n = args.B * (int64_t)F * d;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = e / ((int64_t)F * d);
        const int rem = (int)(e - b * (int64_t)F * d);
        const int f = rem / d, c = rem - f * d;
        args.R[e] = ka->feat[f][b * ka->stride[f] + c];
    }
}

template <int KS>
static void launch_ks(const InteractArgs &a, hipStream_t st) {
    int64_t blocks = (a.B + 3) / 4;
    const int64_t cap = (int64_t)kNumCu * 8;
    if (blocks > cap) blocks = cap;
    if (a.F <= 16)
        hipLaunchKernelGGL((interact_dot_kernel<KS, 1>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((interact_dot_kernel<KS, 2>), dim3((unsigned)blocks), dim3(256), 0, st, a);
}

static int fill_args(InteractArgs &a, int64_t B, int F, int d, const float *const *feats,
                     const int64_t *feat_strides, int itself, float *R) {
    for (int f = 0; f < EVS_MAX_FEATURES; f++) {
        a.feat[f] = f < F ? feats[f] : nullptr;
        a.stride[f] = f < F ? feat_strides[f] : 0;
    }
    a.R = R; a.B = B; a.F = F; a.d = d; a.itself = itself;
    a.P = itself ? F * (F + 1) / 2 : F * (F - 1) / 2;
    return 0;
}

}  // namespace evs

extern "C" int evs_interact_dot(int64_t B, int F, int d, const float *const *feats, const int64_t *feat_strides,
                                int itself, float *R, void *stream) {
    using namespace evs;
    EVS_REQUIRE(B >= 0 && F >= 1 && d >= 1, "evs_interact_dot: bad shape B=%lld F=%d d=%d", (long long)B, F, d);
    if (B == 0) return EVS_OK;
    EVS_REQUIRE(feats && feat_strides && R, "evs_interact_dot: NULL argument");
    itself = itself ? 1 : 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    bool mfma_ok = F <= EVS_MAX_FEATURES && d % 4 == 0 && d <= 256;
    for (int f = 0; f < F && mfma_ok; f++) {
        EVS_REQUIRE(feats[f], "evs_interact_dot: feats[%d] is NULL", f);
        const int ks = d / 4;
        // lanes read KS-float pieces: float4 loads need 16-byte aligned pieces
        if (ks % 4 == 0 && (reinterpret_cast<uintptr_t>(feats[f]) % 16 != 0 || feat_strides[f] % 4 != 0))
            mfma_ok = false;
    }
    if (mfma_ok) {
        InteractArgs a;
        fill_args(a, B, F, d, feats, feat_strides, itself, R);
        switch (d / 4) {
        case 4: launch_ks<4>(a, st); break;    // d = 16
        case 8: launch_ks<8>(a, st); break;    // d = 32
        case 9: launch_ks<9>(a, st); break;    // d = 36
        case 16: launch_ks<16>(a, st); break;  // d = 64
        case 32: launch_ks<32>(a, st); break;  // d = 128
        default: mfma_ok = false; break;
        }
        if (mfma_ok) {
            EVS_HIP_CHECK(hipGetLastError());
            return EVS_OK;
        }
    }
    // generic path: pointer tables go through a small device buffer
    const int P = itself ? F * (F + 1) / 2 : F * (F - 1) / 2;
    void *dev = nullptr;
    const size_t bytes = (size_t)F * (sizeof(float *) + sizeof(int64_t));
    EVS_HIP_CHECK(hipMallocAsync(&dev, bytes, st));
    EVS_HIP_CHECK(hipMemcpyAsync(dev, feats, F * sizeof(float *), hipMemcpyHostToDevice, st));
    EVS_HIP_CHECK(hipMemcpyAsync((char *)dev + F * sizeof(float *), feat_strides, F * sizeof(int64_t),
                                 hipMemcpyHostToDevice, st));
    EVS_HIP_CHECK(hipStreamSynchronize(st));  // feats/feat_strides are caller-owned host memory
    int64_t blocks = (B * (int64_t)(d + P) + 255) / 256;
    if (blocks > kNumCu * 8) blocks = kNumCu * 8;
    hipLaunchKernelGGL(interact_dot_generic_kernel, dim3((unsigned)blocks), dim3(256), 0, st,
                       (const float *const *)dev, (const int64_t *)((char *)dev + F * sizeof(float *)), R, B, F, d,
                       itself, P);
    EVS_HIP_CHECK(hipGetLastError());
    EVS_HIP_CHECK(hipFreeAsync(dev, st));
    return EVS_OK;
}

extern "C" int evs_interact_cat(int64_t B, int F, int d, const float *const *feats, const int64_t *feat_strides,
                                float *R, void *stream) {
    using namespace evs;
    EVS_REQUIRE(B >= 0 && F >= 1 && F <= EVS_MAX_FEATURES && d >= 1, "evs_interact_cat: bad shape B=%lld F=%d d=%d",
                (long long)B, F, d);
    if (B == 0) return EVS_OK;
    EVS_REQUIRE(feats && feat_strides && R, "evs_interact_cat: NULL argument");
    InteractArgs a;
    fill_args(a, B, F, d, feats, feat_strides, 0, R);
    int64_t blocks = (B * (int64_t)F * d + 255) / 256;
    if (blocks > kNumCu * 8) blocks = kNumCu * 8;
    hipLaunchKernelGGL(interact_cat_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    EVS_HIP_CHECK(hipGetLastError());
    return EVS_OK;
}
