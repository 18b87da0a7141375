// Developer probe (not product code): do v_mfma_f32_16x16x4_f32 and plain VALU work of DIFFERENT waves on the same SIMD
// overlap?  One 512-thread block per CU: waves 0-3 (one per SIMD) run a chain of MFMAs, waves 4-7 (their SIMD partners) a
// chain of v_fma_f32; timed alone and together.  together ~ max(alone) => they co-execute; ~ sum => they share the issue.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/coexec_probe tools/coexec_probe.hip && tools/_build/coexec_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ void __launch_bounds__(512) probe(float *out, int n_mfma, int n_valu, int mode) {
    const int wave = threadIdx.x >> 6;
    float a = (float)threadIdx.x * 1e-3f, b = 1.0001f;
    if (wave < 4) {
        if (!(mode & 1)) return;
        f32x4 c[CHAINS];
#pragma unroll
        for (int k = 0; k < CHAINS; k++) c[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < n_mfma; i++) {
#pragma unroll
            for (int k = 0; k < CHAINS; k++) c[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[k], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < CHAINS; k++) s += c[k][0] + c[k][1] + c[k][2] + c[k][3];
        if (s == 123.456f) out[threadIdx.x] = s;
    } else {
        if (!(mode & 2)) return;
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = a + k;
        for (int i = 0; i < n_valu; i++) {
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = __builtin_fmaf(v[k], b, a);
        }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; k++) s += v[k];
        if (s == 123.456f) out[threadIdx.x] = s;
    }
}

template <int CHAINS>
static float run(float *out, int nm, int nv, int mode) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((probe<CHAINS>), dim3(256), dim3(512), 0, 0, out, nm, nv, mode);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    return best * 1e3f;
}

int main() {
    float *out; CK(hipMalloc(&out, 4096));
    const int nm = 20000;            // x CHAINS MFMAs per wave
    for (int nv : {20000, 40000, 80000, 160000}) {   // x 8 v_fma per wave
        const float m1 = run<1>(out, nm * 3, nv, 1), m3 = run<3>(out, nm, nv, 1), v = run<3>(out, nm, nv, 2), b1 = run<1>(out, nm * 3, nv, 3), b3 = run<3>(out, nm, nv, 3);
        printf("60000 MFMA 16x16x4 f32 per wave: one chain %.0f us, three chains %.0f us (%.1f cycles each at 2.4 GHz) | %d v_fma per wave: %.0f us (%.1f cycles each) | together: one chain %.0f us, three chains %.0f us\n",
               m1, m3, m3 * 2400.0 / 60000.0, nv * 8, v, v * 2400.0 / (nv * 8.0), b1, b3);
    }
    return 0;
}
