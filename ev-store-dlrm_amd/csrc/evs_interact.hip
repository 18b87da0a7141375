// interact_features: entry point, generic fallback and the "cat" variant.
// The matrix-core "dot" kernel lives in evs_fused.hip (dense features = plain interaction).
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
    // matrix-core path = the fused kernel with every feature dense (evs_fused.hip)
    bool mfma_ok = F <= EVS_MAX_FEATURES && evs_fused_dim_supported(d) && B < (1ll << 31);
    for (int f = 0; f < F && mfma_ok; f++) {
        EVS_REQUIRE(feats[f], "evs_interact_dot: feats[%d] is NULL", f);
        if (reinterpret_cast<uintptr_t>(feats[f]) % 16 != 0 || feat_strides[f] % 4 != 0) mfma_ok = false;
    }
    if (mfma_ok) {
        evs_feature ft[EVS_MAX_FEATURES];
        for (int f = 0; f < F; f++) {
            ft[f].src = feats[f]; ft[f].stride = feat_strides[f]; ft[f].indices = nullptr; ft[f].offsets = nullptr;
            ft[f].nnz = 0; ft[f].n_rows = 0; ft[f].row_weights = nullptr; ft[f].offsets_len = 0;
        }
        return evs_emb_interact_dot(B, F, d, 32, ft, itself, R, stream);
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
