// Developer probe (not product code): what the STORE side of apply_emb alone costs by lane mapping.  The rows-in-registers
// gather writes a (T, B, d) fp32 output; at B = 65 536, T = 26, d = 36 that is 245 MB, four fifths of the launch's bytes.
// Patterns (a block = 16 samples of all tables, 4 waves, 9 lanes x 16 B per 144-byte row, 7 rows per store instruction):
//   A  the kernel's mapping up to round 4: an instruction = ONE sample of 7 tables (7 separate 144-byte runs); wave w owns
//      samples w, w + 4, w + 8, w + 12
//   B  an instruction = 7 CONSECUTIVE samples of one table (one 1 008-byte run, two at a table boundary); wave w owns tables
//      w, w + 4, ...
//   F  a plain fill of the same bytes (every wave instruction 1 024 contiguous bytes)
// each with ordinary and non-temporal stores.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/store_pattern_probe tools/store_pattern_probe.hip && tools/_build/store_pattern_probe [B]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ void st(float *p, f32x4 v) {
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(p));
    else *reinterpret_cast<f32x4 *>(p) = v;
}

template <int PAT, bool NT>
__global__ void __launch_bounds__(256) store_kernel(float *out, int64_t B, int T, float seed) {
    constexpr int d = 36, LPRD = 9, RPI = 7;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = lane / LPRD, piece = lane - r0 * LPRD;
    const int64_t blk = (int64_t)blockIdx.x * 16;
    const f32x4 v = {seed, seed + lane, seed + wave, seed};
    if (PAT == 0) {
#pragma unroll
        for (int n = 0; n < 4; n++) {
            const int64_t b = blk + wave + 4 * n;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int t = r0 + j * RPI;
                if (r0 < RPI && t < T) st<NT>(out + ((int64_t)t * B + b) * d + piece * 4, v);
            }
        }
    } else if (PAT == 1) {
        const int TW = (T - wave + 3) / 4;           // tables of this wave: wave, wave + 4, ...
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int q = k * RPI + r0, i = q >> 4, s = q & 15;
            const int t = wave + 4 * i;
            if (r0 < RPI && i < TW) st<NT>(out + ((int64_t)t * B + blk + s) * d + piece * 4, v);
        }
    } else {
        // fill: the block's share of the bytes, wave instruction = 1 024 contiguous bytes
        const int64_t per_block = (int64_t)16 * T * d / 4;      // float4s
        f32x4 *o = reinterpret_cast<f32x4 *>(out) + (int64_t)blockIdx.x * per_block;
        for (int64_t i = threadIdx.x; i < per_block; i += 256) st<NT>(reinterpret_cast<float *>(o + i), v);
    }
}

// MIX: what the memory system does with the gather's traffic MIX, free of the kernel's structure -- per block of 16 samples
// waves 0-1 read 416 random 36-byte rows (9 lanes x 4 B, 7 rows per instruction, 15 instructions in flight) of a 1.2 GB
// table, waves 2-3 store the block's 416 output rows (pattern A).  mode 1 = reads only, 2 = stores only, 3 = both at once.
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
template <int ROWB>
__global__ void __launch_bounds__(256) mix_kernel(float *out, const unsigned char *tab, uint64_t nrows, int64_t B, int T, int mode, unsigned *sink, unsigned salt) {
    constexpr int d = 36, LPRD = 9, RPI = 7, PB = ROWB / LPRD;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = lane / LPRD, piece = lane - r0 * LPRD;
    const int64_t blk = (int64_t)blockIdx.x * 16;
    if (wave < 2) {
        if (!(mode & 1)) return;
        unsigned acc = 0;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            unsigned v[15][PB / 4];
#pragma unroll
            for (int k = 0; k < 15; k++) {
                const uint64_t row = mix64(((uint64_t)blockIdx.x * 64 + wave * 32 + h * 15 + k) * 8 + (r0 < RPI ? r0 : 0) + salt) % nrows;
                const unsigned *p = reinterpret_cast<const unsigned *>(tab + row * ROWB + piece * PB);
#pragma unroll
                for (int w = 0; w < PB / 4; w++) v[k][w] = __builtin_nontemporal_load(p + w);
            }
#pragma unroll
            for (int k = 0; k < 15; k++)
#pragma unroll
                for (int w = 0; w < PB / 4; w++) acc ^= v[k][w];
        }
        if (acc == 0x12345u) *sink = acc;
    } else {
        if (!(mode & 2)) return;
        const f32x4 v = {1.f, (float)lane, (float)wave, 2.f};
        const int w2 = wave - 2;
#pragma unroll
        for (int n = 0; n < 8; n++) {
            const int64_t b = blk + w2 + 2 * n;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int t = r0 + j * RPI;
                if (r0 < RPI && t < T) st<false>(out + ((int64_t)t * B + b) * d + piece * 4, v);
            }
        }
    }
}
template <int ROWB> static float run_mix(float *out, const unsigned char *tab, uint64_t nrows, int64_t B, int T, int mode, unsigned *sink, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL((mix_kernel<ROWB>), dim3((unsigned)(B / 16)), dim3(256), 0, 0, out, tab, nrows, B, T, mode, sink, (unsigned)i * 7919u);
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL((mix_kernel<ROWB>), dim3((unsigned)(B / 16)), dim3(256), 0, 0, out, tab, nrows, B, T, mode, sink, (unsigned)(i + 5) * 7919u);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters * 1e3f;
}

template <int PAT, bool NT> static float run(float *out, int64_t B, int T, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL((store_kernel<PAT, NT>), dim3((unsigned)(B / 16)), dim3(256), 0, 0, out, B, T, (float)i);
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL((store_kernel<PAT, NT>), dim3((unsigned)(B / 16)), dim3(256), 0, 0, out, B, T, (float)i);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters * 1e3f;
}

int main(int argc, char **argv) {
    const int T = 26;
    for (int64_t B : {(int64_t)16384, (int64_t)65536}) {
        if (argc > 1 && atoll(argv[1]) != B) continue;
        const int64_t bytes = B * T * 36 * 4;
        float *out; CK(hipMalloc(&out, bytes));
        const double mb = bytes / 1e6;
        const float a = run<0, false>(out, B, T, 50), an = run<0, true>(out, B, T, 50);
        const float b = run<1, false>(out, B, T, 50), bn = run<1, true>(out, B, T, 50);
        const float f = run<2, false>(out, B, T, 50), fn = run<2, true>(out, B, T, 50);
        printf("B=%6lld (%.0f MB): A %.1f us (%.2f TB/s)  A nt %.1f (%.2f) | B %.1f (%.2f)  B nt %.1f (%.2f) | fill %.1f (%.2f)  fill nt %.1f (%.2f)\n",
               (long long)B, mb, a, mb / a, an, mb / an, b, mb / b, bn, mb / bn, f, mb / f, fn, mb / fn);
        const uint64_t nrows = 33762577;
        unsigned char *tab; CK(hipMalloc(&tab, nrows * 144)); CK(hipMemset(tab, 1, nrows * 144));
        unsigned *sink; CK(hipMalloc(&sink, 4));
        const float r8 = run_mix<36>(out, tab, nrows, B, T, 1, sink, 50), w8 = run_mix<36>(out, tab, nrows, B, T, 2, sink, 50), m8 = run_mix<36>(out, tab, nrows, B, T, 3, sink, 50);
        const float r32 = run_mix<144>(out, tab, nrows, B, T, 1, sink, 50), m32 = run_mix<144>(out, tab, nrows, B, T, 3, sink, 50);
        printf("   mix, %lld random rows + %.0f MB of stores: 36-byte rows: reads %.1f us, stores %.1f us, both %.1f us | 144-byte rows: reads %.1f us, both %.1f us\n",
               (long long)(B * T), mb, r8, w8, m8, r32, m32);
        CK(hipFree(tab)); CK(hipFree(sink));
        CK(hipFree(out));
    }
    return 0;
}
