// Dev micro-benchmark: fp32 atomic-add scatter of 400-byte rows, agent scope (memory side) vs XCD-local L2
// (workgroup scope into a per-XCC private copy).  hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ int xcc_id() {
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_scatter(float* __restrict__ g, const int* __restrict__ rows, int n, int d,
                                                 long stride_xcc, int* xcc_hist) {
    const int G = 32;
    const int gl = threadIdx.x % G;
    const int x = xcc_id();
    if (threadIdx.x == 0 && xcc_hist) atomicAdd(xcc_hist + x, 1);
    float* base = MODE == 1 ? g + (long)x * stride_xcc : g;
    for (long i = (long)blockIdx.x * (256 / G) + threadIdx.x / G; i < n; i += (long)gridDim.x * (256 / G)) {
        float* row = base + (long)rows[i] * d;
#pragma unroll 4
        for (int e = gl; e < d; e += G) {
            if (MODE == 0) unsafeAtomicAdd(row + e, 1.0f);
            else if (MODE == 1) __hip_atomic_fetch_add(row + e, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (MODE == 2) { if (e * 2 < d) atomicAdd((unsigned long long*)row + e, 0x0000000100000001ull); }  // 2 x int32 per op
            else { if (e * 2 < d) unsafeAtomicAdd((double*)row + e, 1.0); }
        }
    }
}

int main() {
    const int E = 14951 + 1345, d = 100;
    const long numel = (long)E * d;
    for (int n : {32768 * 4, 131072 * 4}) {
        std::vector<int> h(n);
        srand(1);
        for (auto& v : h) v = rand() % E;
        int* rows; CK(hipMalloc(&rows, n * sizeof(int)));
        CK(hipMemcpy(rows, h.data(), n * sizeof(int), hipMemcpyHostToDevice));
        float* g; CK(hipMalloc(&g, numel * 8 * sizeof(float)));
        int* hist; CK(hipMalloc(&hist, 64)); CK(hipMemset(hist, 0, 64));
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        for (int mode = 0; mode < 4; ++mode) {
            CK(hipMemset(g, 0, numel * 8 * sizeof(float)));
            int grid = 2048;
            for (int it = 0; it < 3; ++it) {
                if (mode == 0) k_scatter<0><<<grid, 256>>>(g, rows, n, d, numel, it == 0 ? hist : nullptr);
                else if (mode == 1) k_scatter<1><<<grid, 256>>>(g, rows, n, d, numel, nullptr);
                else if (mode == 2) k_scatter<2><<<grid, 256>>>(g, rows, n, d, numel, nullptr);
                else k_scatter<3><<<grid, 256>>>(g, rows, n, d, numel, nullptr);
            }
            CK(hipDeviceSynchronize());
            CK(hipMemset(g, 0, numel * 8 * sizeof(float)));
            CK(hipEventRecord(a));
            const int iters = 20;
            for (int it = 0; it < iters; ++it) {
                if (mode == 0) k_scatter<0><<<grid, 256>>>(g, rows, n, d, numel, nullptr);
                else if (mode == 1) k_scatter<1><<<grid, 256>>>(g, rows, n, d, numel, nullptr);
                else if (mode == 2) k_scatter<2><<<grid, 256>>>(g, rows, n, d, numel, nullptr);
                else k_scatter<3><<<grid, 256>>>(g, rows, n, d, numel, nullptr);
            }
            CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            std::vector<float> out(numel * 8);
            CK(hipMemcpy(out.data(), g, numel * 8 * sizeof(float), hipMemcpyDeviceToHost));
            double tot = 0; for (float v : out) tot += v;
            printf("n=%d mode=%s: %.1f us/launch, %.2f TB/s payload, sum=%.0f expected=%.0f\n", n,
                   mode == 0 ? "f32 agent " : mode == 1 ? "f32 wg    " : mode == 2 ? "u64 2xi32 " : "f64       ", ms / iters * 1e3, (double)n * d * 4 / (ms / iters * 1e-3) / 1e12, tot,
                   (double)n * d * iters);
        }
        int hh[16]; CK(hipMemcpy(hh, hist, 64, hipMemcpyDeviceToHost));
        printf("xcc histogram:"); for (int i = 0; i < 8; ++i) printf(" %d", hh[i]); printf("\n");
    }
    return 0;
}
