// Dev micro-benchmark: what random ROW traffic costs on MI355X at the shapes of the pointwise training step (config C2:
// ComplEx WN18RR, 40 943 rows of 200 floats per table, ~10-30 k rows touched per step).  Three primitives, each as a function
// of the number of rows touched and of the table size (cache level):
//   gather : one lane group per row, float4 loads, row reduced to a scalar (what a scorer's operand fetch looks like)
//   rmw    : read a parameter row + a state row, Adagrad-style update, write both back (the sparse optimiser's floor)
//   chain  : k DEPENDENT gathers per group (row id of hop j+1 comes from hop j): the latency of one memory round trip under load
// hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o tools/_libs/gather_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <numeric>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int G>
__device__ __forceinline__ float gsum(float v) {
#pragma unroll
    for (int o = G / 2; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// rows of `nvec` float4; group of G lanes per row, NV float4 per lane
template <int G, int NV>
__global__ __launch_bounds__(256) void k_gather(const float4* __restrict__ tab, const int* __restrict__ idx, int n, int nvec,
                                                float* __restrict__ out, int rows_per_group) {
    const int gl = threadIdx.x % G;
    const long g0 = ((long)blockIdx.x * (256 / G) + threadIdx.x / G) * rows_per_group;
    float acc = 0.f;
    for (int k = 0; k < rows_per_group; ++k) {
        const long g = g0 + k;
        if (g >= n) break;
        const float4* row = tab + (long)idx[g] * nvec;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int i = v * G + gl;
            if (i < nvec) { const float4 x = row[i]; acc += x.x + x.y + x.z + x.w; }
        }
    }
    acc = gsum<G>(acc);
    if (gl == 0 && g0 < n) out[g0 / rows_per_group] = acc;
}

template <int G, int NV>
__global__ __launch_bounds__(256) void k_rmw(float4* __restrict__ p, float4* __restrict__ s, const int* __restrict__ idx, int n,
                                             int nvec) {
    const int gl = threadIdx.x % G;
    const long g = (long)blockIdx.x * (256 / G) + threadIdx.x / G;
    if (g >= n) return;
    const long base = (long)idx[g] * nvec;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int i = v * G + gl;
        if (i < nvec) {
            float4 x = p[base + i], a = s[base + i];
            const float gr = 1e-3f;
            a.x += gr * gr; a.y += gr * gr; a.z += gr * gr; a.w += gr * gr;
            x.x -= 0.01f * gr / (sqrtf(a.x) + 1e-10f); x.y -= 0.01f * gr / (sqrtf(a.y) + 1e-10f);
            x.z -= 0.01f * gr / (sqrtf(a.z) + 1e-10f); x.w -= 0.01f * gr / (sqrtf(a.w) + 1e-10f);
            p[base + i] = x; s[base + i] = a;
        }
    }
}

template <int G, int NV>
__global__ __launch_bounds__(256) void k_chain(const float4* __restrict__ tab, const int* __restrict__ idx, int n, int nvec,
                                               int rows, int hops, float* __restrict__ out) {
    const int gl = threadIdx.x % G;
    const long g = (long)blockIdx.x * (256 / G) + threadIdx.x / G;
    if (g >= n) return;
    int r = idx[g];
    float acc = 0.f;
    for (int h = 0; h < hops; ++h) {
        const float4* row = tab + (long)r * nvec;
        float a = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int i = v * G + gl;
            if (i < nvec) { const float4 x = row[i]; a += x.x + x.y + x.z + x.w; }
        }
        a = gsum<G>(a);
        acc += a;
        r = (int)((unsigned)(r * 2654435761u + (unsigned)(int)(a * 1e-9f)) % (unsigned)rows);   // next row depends on the data just read
    }
    if (gl == 0) out[g] = acc;
}

static float time_us(hipEvent_t a, hipEvent_t b, int reps) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms * 1e3f / reps; }

int main() {
    const int d = 200, nvec = d / 4;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::mt19937 rng(1);
    printf("random row traffic, rows of %d floats (%d B); G = lanes per row\n", d, d * 4);
    for (int rows : {16296, 40943, 163772, 655088}) {
        const long numel4 = (long)rows * nvec;
        float4 *tab, *st; int* idx; float* out;
        CK(hipMalloc(&tab, numel4 * 16)); CK(hipMalloc(&st, numel4 * 16));
        CK(hipMemset(tab, 0, numel4 * 16)); CK(hipMemset(st, 0, numel4 * 16));
        const int nmax = 131072;
        CK(hipMalloc(&idx, nmax * 4)); CK(hipMalloc(&out, nmax * 4));
        printf("== table %d rows = %.1f MB (x2 for rmw)\n", rows, numel4 * 16 / 1e6);
        for (int n : {8192, 32768, 131072}) {
            std::vector<int> h(n);
            if (n <= rows) {   // distinct rows (an optimiser never visits a row twice)
                std::vector<int> all(rows); std::iota(all.begin(), all.end(), 0); std::shuffle(all.begin(), all.end(), rng);
                std::copy(all.begin(), all.begin() + n, h.begin());
            } else for (auto& x : h) x = (int)(rng() % rows);
            CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
            const int reps = 20;
#define RUN(label, bytes, launch)                                                                              \
            { for (int w = 0; w < 3; ++w) { launch; }                                                          \
              CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));                                              \
              for (int r = 0; r < reps; ++r) { launch; }                                                       \
              CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));                                             \
              const float us = time_us(e0, e1, reps);                                                          \
              printf("  n=%6d %-28s %8.2f us  %7.1f GB/s\n", n, label, us, (bytes) / us / 1e3); }
            RUN("gather G=64 1 row/group", (double)n * d * 4, (k_gather<64, 1><<<(n + 3) / 4, 256>>>(tab, idx, n, nvec, out, 1)))
            RUN("gather G=32 1 row/group", (double)n * d * 4, (k_gather<32, 2><<<(n + 7) / 8, 256>>>(tab, idx, n, nvec, out, 1)))
            RUN("gather G=16 1 row/group", (double)n * d * 4, (k_gather<16, 4><<<(n + 15) / 16, 256>>>(tab, idx, n, nvec, out, 1)))
            RUN("gather G=64 4 rows/group", (double)n * d * 4, (k_gather<64, 1><<<(n / 4 + 3) / 4, 256>>>(tab, idx, n, nvec, out, 4)))
            RUN("gather G=64 8 rows/group", (double)n * d * 4, (k_gather<64, 1><<<(n / 8 + 3) / 4, 256>>>(tab, idx, n, nvec, out, 8)))
            if (n <= rows) {
                RUN("rmw    G=64 (p + state)", (double)n * d * 4 * 4, (k_rmw<64, 1><<<(n + 3) / 4, 256>>>(tab, st, idx, n, nvec)))
                RUN("rmw    G=32 (p + state)", (double)n * d * 4 * 4, (k_rmw<32, 2><<<(n + 7) / 8, 256>>>(tab, st, idx, n, nvec)))
            }
            RUN("chain  G=64 4 hops", (double)n * d * 4 * 4, (k_chain<64, 1><<<(n + 3) / 4, 256>>>(tab, idx, n, nvec, rows, 4, out)))
            RUN("chain  G=64 8 hops", (double)n * d * 4 * 8, (k_chain<64, 1><<<(n + 3) / 4, 256>>>(tab, idx, n, nvec, rows, 8, out)))
            RUN("chain  G=32 8 hops", (double)n * d * 4 * 8, (k_chain<32, 2><<<(n + 7) / 8, 256>>>(tab, idx, n, nvec, rows, 8, out)))
        }
        CK(hipFree(tab)); CK(hipFree(st)); CK(hipFree(idx)); CK(hipFree(out));
    }
    // empty-kernel launch cadence on this box
    {
        float* out; CK(hipMalloc(&out, 4096));
        int* idx; CK(hipMalloc(&idx, 4096)); CK(hipMemset(idx, 0, 4096));
        float4* tab; CK(hipMalloc(&tab, 1 << 20)); CK(hipMemset(tab, 0, 1 << 20));
        CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
        for (int r = 0; r < 200; ++r) k_gather<64, 1><<<1, 256>>>(tab, idx, 1, 50, out, 1);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        printf("back-to-back trivial launches: %.2f us each\n", time_us(e0, e1, 200));
    }
    return 0;
}
