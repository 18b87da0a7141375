// Developer probe (not product code): how many RANDOM rows per second this part serves, by row size, alignment and load shape.
// The reduced-precision fused kernels read 26 rows of 36 (u8) / 18 (u4) / 72 (u16) bytes per sample at random offsets of
// tables far larger than the caches; this measures the ceiling of that access pattern alone -- no index loads, no decode,
// no output -- so that the kernels' fraction of the HBM peak can be read against what the memory system does for such rows.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/sector_probe tools/sector_probe.hip && tools/_build/sector_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

template <int N> struct Vec { uint32_t v[N]; };

// LPR lanes per row, DW dwords per lane (1, 2, 3, 4), U independent loads in flight per lane, ROWB row pitch in bytes
template <int LPR, int DW, int U, bool TR = false>
__global__ void __launch_bounds__(256) probe_kernel(const uint32_t *__restrict__ base, uint64_t nrows, int rowb, int iters, uint32_t *sink, int dep,
                                                    const int64_t *__restrict__ idx) {
    constexpr int RPW = 64 / LPR;                       // rows per wave instruction
    const int lane = threadIdx.x & 63;
    // TR: the matrix-core operand mapping of the rows-in-registers kernels -- row = lane % 16, piece = lane / 16 (LPR = 4)
    const int g = TR ? lane % 16 : lane / LPR, piece = TR ? lane / 16 : lane % LPR;
    const bool live = g < RPW;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        Vec<DW> r[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            uint64_t row;
            const uint64_t n = ((wave * iters + it) * U + u) * RPW + (live ? g : 0);
            if (dep) row = (uint64_t)idx[n];            // a coalesced index load in front of the row load (the kernels' shape)
            else row = (uint64_t)(((unsigned __int128)mix64(n + 0x9e3779b97f4a7c15ull) * nrows) >> 64);
            const uint32_t *p = (const uint32_t *)((const char *)base + row * (uint64_t)rowb) + piece * DW;
            if constexpr (DW == 1) { r[u].v[0] = p[0]; }
            else if constexpr (DW == 2) { const uint2 t = *(const uint2 *)p; r[u].v[0] = t.x; r[u].v[1] = t.y; }
            else if constexpr (DW == 3) { r[u].v[0] = p[0]; r[u].v[1] = p[1]; r[u].v[2] = p[2]; }
            else { const uint4 t = *(const uint4 *)p; r[u].v[0] = t.x; r[u].v[1] = t.y; r[u].v[2] = t.z; r[u].v[3] = t.w; }
        }
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int k = 0; k < DW; k++) acc ^= r[u].v[k];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

struct Case { const char *name; int lpr, dw, rowb; bool tr = false; };

template <int LPR, int DW, int U, bool TR = false>
static float run(const uint32_t *base, uint64_t nrows, int rowb, int blocks, int iters, uint32_t *sink, int dep, const int64_t *idx) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((probe_kernel<LPR, DW, U, TR>), dim3(blocks), dim3(256), 0, 0, base, nrows, rowb, iters, sink, dep, idx);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) best = std::min(best, ms);
    }
    return best;
}

int main(int argc, char **argv) {
    const double span_gb = argc > 1 ? atof(argv[1]) : 1.2;      // bytes the random rows are spread over
    const int rows_target = argc > 2 ? atoi(argv[2]) : 65536 * 26;   // rows per launch (the u8 launch at B = 65 536)
    const bool quick = getenv("QUICK") != nullptr;              // U = 8, 2 048 blocks only
    const bool kaggle = argc > 3 && !strcmp(argv[3], "kaggle"); // rows drawn as the bench draws them: feature k = i % 26, uniform within table k
    static const int64_t LN[26] = {1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27, 14992, 5461306, 10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572};
    int64_t ln_base[27] = {0};
    for (int k = 0; k < 26; k++) ln_base[k + 1] = ln_base[k] + LN[k];
    const uint64_t span = kaggle ? (uint64_t)ln_base[26] * 144 : (uint64_t)(span_gb * 1e9) & ~127ull;
    uint32_t *base, *sink;
    CK(hipMalloc(&base, span + 4096));
    CK(hipMemset(base, 1, span + 4096));
    CK(hipMalloc(&sink, 4));
    int64_t *idx;
    const size_t nidx = (size_t)rows_target * 2 + (1 << 20);
    CK(hipMalloc(&idx, nidx * 8));
    printf("span %.2f GB, about %d rows per launch; rows/s are per launch, best of 3 after one warm-up\n", span / 1e9, rows_target);
    printf("%-48s %4s %9s %10s %10s %10s\n", "case", "U", "blocks", "us", "G rows/s", "GB/s rows");
    const Case cases[] = {
        {"128 B line, aligned: 8 lanes x 16 B", 8, 4, 128}, {"64 B sector, aligned: 4 lanes x 16 B", 4, 4, 64},
        {"32 B, aligned: 2 lanes x 16 B", 2, 4, 32},        {"4 B of a random 64 B sector: 1 lane", 1, 1, 64},
        {"36 B row (u8), 4 B aligned: 9 lanes x 4 B", 9, 1, 36}, {"36 B row (u8): 3 lanes x 12 B", 3, 3, 36},
        {"18 B row pitch 20 (u4-like): 5 lanes x 4 B", 5, 1, 20}, {"72 B row (u16): 9 lanes x 8 B", 9, 2, 72},
        {"144 B row (fp32): 9 lanes x 16 B", 9, 4, 144},
        {"64 B pitch, 36 B read (padded u8): 3 x 12 B", 3, 3, 64},
        {"32 B pitch, 16 B read (padded u4): 1 x 16 B", 1, 4, 32},
        {"36 B row, 32 B read: 4 adjacent lanes x 8 B", 4, 2, 36}, {"36 B row, 32 B read: lanes l, l+16, .. x 8 B", 4, 2, 36, true},
        {"144 B row, 64 B read: 4 adjacent lanes x 16 B", 4, 4, 144}, {"144 B row, 64 B read: lanes l, l+16, .. x 16 B", 4, 4, 144, true},
        {"18 B row pitch 20, 16 B read: 4 adj x 4 B", 4, 1, 20}, {"18 B row pitch 20, 16 B read: l, l+16, .. x 4 B", 4, 1, 20, true},
        {"18 B row pitch 18 (u4): 4 adj x 4 B, 2 B aligned", 4, 1, 18}, {"36 B row pitch 36 (u8), 32 B read: 4 adj x 8 B, 4 B aligned", 4, 2, 36},
        {"18 B row pitch 18 (u4): 2 adj x 8 B, 2 B aligned", 2, 2, 18},
        {"72 B row pitch 72 (u16), 64 B read: 4 adj x 16 B, 8 B aligned", 4, 4, 72}, {"80 B pitch, 64 B read: 4 adj x 16 B, 16 B aligned", 4, 4, 80},
        {"144 B row pitch 144 (fp32), 64 B read: 4 adj x 16 B", 4, 4, 144}, {"36 B row pitch 36 (u8): 3 adj x 12 B, 4 B aligned", 3, 3, 36}, {"48 B pitch: 3 adj x 12 B... 16 B aligned rows", 3, 3, 48},
    };
    for (const Case &c : cases) {
        const uint64_t nrows = kaggle ? (uint64_t)ln_base[26] : span / c.rowb;
        const int rpw = 64 / c.lpr;
        for (int dep = kaggle ? 1 : 0; dep < (quick && !kaggle ? 1 : 2); dep++) {
            if (dep) {   // indices: uniformly random rows, generated on the host once per case
                std::vector<int64_t> h(nidx);
                uint64_t s = 88172645463325252ull;
                for (size_t i = 0; i < nidx; i++) {
                    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
                    h[i] = kaggle ? ln_base[i % 26] + (int64_t)(s % (uint64_t)LN[i % 26]) : (int64_t)(s % nrows);
                }
                CK(hipMemcpy(idx, h.data(), nidx * 8, hipMemcpyHostToDevice));
            }
            for (int U : {4, 8, 16}) {
                if (quick && U != 8) continue;
                for (int bpc : {4, 8}) {
                    if (quick && bpc != 8) continue;
                    const int blocks = 256 * bpc;
                    const int iters = std::max(1, (int)((double)rows_target / ((double)blocks * 4 * U * rpw) + 0.5));
                    const double rows = (double)blocks * 4 * iters * U * rpw;
                    if (dep && rows > (double)nidx) continue;
                    float ms = 0;
#define RUNT(L, D) (U == 4 ? run<L, D, 4, true>(base, nrows, c.rowb, blocks, iters, sink, dep, idx) : U == 8 ? run<L, D, 8, true>(base, nrows, c.rowb, blocks, iters, sink, dep, idx) \
                                                                                                      : run<L, D, 16, true>(base, nrows, c.rowb, blocks, iters, sink, dep, idx))
#define RUN(L, D) (U == 4 ? run<L, D, 4>(base, nrows, c.rowb, blocks, iters, sink, dep, idx) : U == 8 ? run<L, D, 8>(base, nrows, c.rowb, blocks, iters, sink, dep, idx) \
                                                                                                      : run<L, D, 16>(base, nrows, c.rowb, blocks, iters, sink, dep, idx))
                    if (c.tr) ms = c.dw == 1 ? RUNT(4, 1) : c.dw == 2 ? RUNT(4, 2) : RUNT(4, 4); else if (c.lpr == 4 && c.dw == 1) ms = RUN(4, 1); else if (c.lpr == 4 && c.dw == 2) ms = RUN(4, 2);
                    else if (c.lpr == 8) ms = RUN(8, 4); else if (c.lpr == 4) ms = RUN(4, 4); else if (c.lpr == 2) ms = RUN(2, 4);
                    else if (c.lpr == 1 && c.dw == 1) ms = RUN(1, 1); else if (c.lpr == 1) ms = RUN(1, 4); else if (c.lpr == 9 && c.dw == 1) ms = RUN(9, 1); else if (c.lpr == 3) ms = RUN(3, 3);
                    else if (c.lpr == 5) ms = RUN(5, 1); else if (c.lpr == 9 && c.dw == 2) ms = RUN(9, 2); else ms = RUN(9, 4);
                    printf("%-48s %4d %9d %10.1f %10.1f %10.0f  %s\n", c.name, U, blocks, ms * 1e3, rows / ms / 1e6, rows * c.lpr * c.dw * 4 / ms / 1e6,
                           kaggle ? "Kaggle table mix, index loaded" : dep ? "index loaded" : "index computed");
                }
            }
        }
    }
    return 0;
}
