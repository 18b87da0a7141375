// The Criteo collate on the device (gfx950): the input side of the hot path.
//
// The reference's loader builds every batch on the host -- collate_wrapper_criteo_offset, dlrm_data_pytorch.py:397-410:
// X = log(X_int + 1) in fp32, lS_i = X_cat transposed to (T, B) int64, lS_o = arange(B) per table, (T, B) int64 -- and
// dlrm_wrap (dlrm_s_pytorch.py:131-147) then copies the three tensors to the device: 468 bytes per sample at T = 26 over
// the bus.  The raw batch as the dataset holds it (CriteoDataset.__getitem__, :372-395: 13 int32 counts and 26 int32 ids per
// sample) is 156 bytes per sample; this kernel makes the same three tensors from it in HBM, one launch.
//   a block owns 64 samples: the (64, T) ids and the (64, n_dense) counts are read as the contiguous runs they are, the ids
//   go through an LDS tile and leave as T rows of 64 consecutive int64 (512-byte runs), lS_o beside them.
// The same for the Terabyte binary loader (script/data_loader_terabyte.py:196-236 CriteoBinDataset + :68-87 _transform_features):
// a batch there is a (B, 40) int32 block of the file -- label, 13 counts, 26 ids per record -- so the two inputs are views of one
// array with a row stride of 40 (x_int_stride / x_cat_stride, in elements), and ids are taken modulo max_ind_range when > 0.
#include "evs_common.h"
#include <hip/hip_runtime.h>

namespace evs {

struct CollateArgs {
    const int *x_int, *x_cat;
    float *X;
    int64_t *lS_o, *lS_i;
    int64_t B, int_stride, cat_stride;
    int n_dense, T, max_ind_range;
};

__global__ void __launch_bounds__(256) collate_criteo_kernel(const CollateArgs a) {
    __shared__ int s_cat[64 * 65];   // [sample][table], one word of padding per sample
    const int64_t b0 = (int64_t)blockIdx.x * 64;
    const int nb = (int)(a.B - b0 < 64 ? a.B - b0 : 64);
    const int T = a.T, nd = a.n_dense;
    // ids: T consecutive ints per sample
    const int *cat = a.x_cat + b0 * a.cat_stride;
    for (int i = threadIdx.x; i < nb * T; i += 256) {
        const int s = i / T, t = i - s * T;
        int v = cat[(int64_t)s * a.cat_stride + t];
        if (a.max_ind_range > 0) v %= a.max_ind_range;   // (x_cat_batch % max_ind_range, data_loader_terabyte.py:71-72; ids are >= 0)
        s_cat[s * 65 + t] = v;
    }
    // dense counts -> log(x + 1) in fp32 (torch.log(torch.tensor(.., dtype=torch.float) + 1)): nb * nd consecutive elements
    const int *xi = a.x_int + b0 * a.int_stride;
    float *xo = a.X + b0 * nd;
    for (int i = threadIdx.x; i < nb * nd; i += 256) {
        const int s = i / (nd > 0 ? nd : 1), c = i - s * nd;
        xo[i] = logf(__fadd_rn((float)xi[(int64_t)s * a.int_stride + c], 1.0f));
    }
    __syncthreads();
    // table t, samples b0 .. b0 + nb: one wave per table row at a time
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = wave; t < T; t += 4) {
        if (lane < nb) {
            a.lS_i[(int64_t)t * a.B + b0 + lane] = (int64_t)s_cat[lane * 65 + t];
            if (a.lS_o) a.lS_o[(int64_t)t * a.B + b0 + lane] = b0 + lane;
        }
    }
}

}  // namespace evs

extern "C" int evs_collate_criteo_offset(int64_t B, int n_dense, int T, const int32_t *x_int, int64_t x_int_stride, const int32_t *x_cat,
                                         int64_t x_cat_stride, int max_ind_range, float *X, int64_t *lS_o, int64_t *lS_i, void *stream) {
    using namespace evs;
    EVS_REQUIRE(B >= 0 && n_dense >= 0 && T >= 1 && T <= 64, "evs_collate_criteo_offset: needs 1 <= T <= 64 (got T=%d)", T);
    EVS_REQUIRE(x_cat && lS_i && (n_dense == 0 || (x_int && X)), "evs_collate_criteo_offset: NULL argument");
    EVS_REQUIRE(x_cat_stride >= T && (n_dense == 0 || x_int_stride >= n_dense), "evs_collate_criteo_offset: a row stride below the row length");
    if (B == 0) return EVS_OK;
    CollateArgs a{x_int, x_cat, X, lS_o, lS_i, B, x_int_stride, x_cat_stride, n_dense, T, max_ind_range};
    hipLaunchKernelGGL(collate_criteo_kernel, dim3((unsigned)((B + 63) / 64)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    EVS_HIP_CHECK(hipGetLastError());
    return EVS_OK;
}
