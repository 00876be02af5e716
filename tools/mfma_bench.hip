// Dev micro-benchmark: f32-input MFMA issue throughput on MI355X (v_mfma_f32_32x32x2_f32, v_mfma_f32_16x16x4_f32) with NACC
// independent accumulators per wave at 1 / 2 / 4 waves per SIMD, operands in registers -- the roof of k_eval_gemm.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ITERS = 4096;

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma32(float* out, const float* in) {
    f32x16 acc[NACC];
    const float a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x16{0};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float r = 0.f;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) r += acc[i][j];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = r;
}
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma16(float* out, const float* in) {
    f32x4 acc[NACC];
    const float a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float r = 0.f;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 4; ++j) r += acc[i][j];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = r;
}

template <class F>
void timeit(const char* name, F launch, double flop_per_wave) {
    printf("%-34s", name);
    for (int wps : {1, 2, 4}) {
        const int grid = 256 * wps;
        launch(grid);
        CK(hipDeviceSynchronize());
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        CK(hipEventRecord(a));
        for (int r = 0; r < 5; ++r) launch(grid);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        ms /= 5;
        printf(" | %d waves/SIMD: %.1f TFLOP/s", wps, flop_per_wave * 4.0 * grid / (ms * 1e-3) / 1e12);
    }
    printf("\n");
}

int main() {
    float *out, *in;
    CK(hipMalloc(&out, 256 * 4 * 256 * sizeof(float)));
    CK(hipMalloc(&in, 128 * sizeof(float)));
    float h[128];
    for (int i = 0; i < 128; ++i) h[i] = 1e-3f * (i + 1);
    CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
    timeit("mfma_f32_32x32x2f32, 1 acc", [&](int g) { k_mfma32<1><<<g, 256>>>(out, in); }, (double)ITERS * 1 * 4096);
    timeit("mfma_f32_32x32x2f32, 2 acc", [&](int g) { k_mfma32<2><<<g, 256>>>(out, in); }, (double)ITERS * 2 * 4096);
    timeit("mfma_f32_32x32x2f32, 4 acc", [&](int g) { k_mfma32<4><<<g, 256>>>(out, in); }, (double)ITERS * 4 * 4096);
    timeit("mfma_f32_16x16x4f32, 4 acc", [&](int g) { k_mfma16<4><<<g, 256>>>(out, in); }, (double)ITERS * 4 * 2048);
    timeit("mfma_f32_16x16x4f32, 8 acc", [&](int g) { k_mfma16<8><<<g, 256>>>(out, in); }, (double)ITERS * 8 * 2048);
    // sustained rate: the same 4-accumulator kernel back to back for ~0.2 s (clock / power management settles), reported per
    // block of 50 launches -- what a long GEMM sweep can actually count on
    printf("sustained, mfma_f32_32x32x2f32 4 acc, 4 waves/SIMD, blocks of 50 launches:");
    for (int blk = 0; blk < 8; ++blk) {
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        CK(hipEventRecord(a));
        for (int r = 0; r < 50; ++r) k_mfma32<4><<<1024, 256>>>(out, in);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf(" %.1f", (double)ITERS * 4 * 4096 * 4.0 * 1024 * 50 / (ms * 1e-3) / 1e12);
    }
    printf(" TFLOP/s\n");
    return 0;
}
