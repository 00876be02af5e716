// Dev micro-benchmark: VALU issue throughput on MI355X for the instruction forms the rank sweep (kge_eval.hip) is made of.
// Answers "what is the VALU issue roof of k_eval_sweep<L1>": cycles per wave64 instruction per SIMD for
//   v_add_f32 (VOP2)            v_add_f32 |x| (VOP3, abs modifier)       v_add_f32 with an SGPR source
//   v_fma_f32                   v_pk_add_f32                             v_pk_fma_f32
//   the L1 mix of the sweep (1 x v_pk_add_f32 with SGPR pair + 2 x v_add_f32 |x|  per two elements)
// at 1 / 2 / 4 / 8 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 tools/valu_bench.hip -o tools/valu_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int NACC = 16;     // independent accumulators (dependent-issue latency is 4 cycles: 16 chains hide it)
constexpr int ITERS = 2048;

template <int MODE>
__global__ __launch_bounds__(256) void k_valu(float* out, const float* in, unsigned long long* cycles) {
    float a[NACC];
    f32x2 p[NACC];
    const float x = in[threadIdx.x & 63];
    // wave-uniform operands in SGPRs
    const float s0 = __builtin_amdgcn_readfirstlane(in[64]), s1 = __builtin_amdgcn_readfirstlane(in[65]);
    f32x2 sq; sq.x = s0; sq.y = s1;
#pragma unroll
    for (int i = 0; i < NACC; ++i) { a[i] = x + i; p[i].x = x + i; p[i].y = x - i; }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if constexpr (MODE == 0) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
            else if constexpr (MODE == 1) asm volatile("v_add_f32 %0, |%1|, %0" : "+v"(a[i]) : "v"(x));
            else if constexpr (MODE == 2) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "s"(s0));
            else if constexpr (MODE == 3) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(a[i]) : "v"(x));
            else if constexpr (MODE == 4) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[i]) : "v"(p[(i + 1) % NACC]));
            else if constexpr (MODE == 5) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p[i]) : "v"(p[(i + 1) % NACC]));
            else if constexpr (MODE == 6) {
                // the sweep's L1 step for two elements of one (query, candidate) pair
                f32x2 d;
                asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(p[i]), "s"(sq));
                asm volatile("v_add_f32 %0, |%1|, %0" : "+v"(a[i]) : "v"(d.x));
                asm volatile("v_add_f32 %0, |%1|, %0" : "+v"(a[i]) : "v"(d.y));
            } else if constexpr (MODE == 7) {
                // all-scalar alternative: 2 x (v_sub_f32 with SGPR + v_add_f32 |x|) per two elements
                float d0, d1;
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d0) : "v"(p[i].x), "s"(s0));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d1) : "v"(p[i].y), "s"(s1));
                asm volatile("v_add_f32 %0, |%1|, %0" : "+v"(a[i]) : "v"(d0));
                asm volatile("v_add_f32 %0, |%1|, %0" : "+v"(a[i]) : "v"(d1));
            } else if constexpr (MODE == 9) {
                // dot form (DistMult / ComplEx sweep): one v_pk_fma_f32 with an SGPR pair per two elements
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(p[(i + 1) % NACC]), "s"(sq));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) r += a[i] + p[i].x + p[i].y;
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cycles[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
void run(const char* name, int instr_per_acc, double elems_per_acc, float* out, float* in, unsigned long long* cyc) {
    printf("%-58s", name);
    for (int wps : {1, 2, 4, 8}) {            // waves per SIMD: blocks of 4 waves (one per SIMD) x wps per CU
        const int grid = 256 * wps;
        k_valu<MODE><<<grid, 256>>>(out, in, cyc);
        CK(hipDeviceSynchronize());
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        CK(hipEventRecord(a));
        const int reps = 5;
        for (int r = 0; r < reps; ++r) k_valu<MODE><<<grid, 256>>>(out, in, cyc);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        ms /= reps;
        unsigned long long h[16];
        CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
        const double winstr = (double)ITERS * NACC * instr_per_acc;          // wave-instructions per wave
        const double wave_cyc = (double)h[0] / winstr;                       // cycles per instruction seen by ONE wave
        const double simd_rate = winstr * wps / (ms * 1e-3);                 // wave-instr/s per SIMD
        const double elems = (double)ITERS * NACC * elems_per_acc * 64.0 * 4 * grid / (ms * 1e-3);
        printf(" | w%d: %.2f cyc/instr/wave, %.3f Ginstr/s/SIMD (=%.2f cyc@2.4GHz), %.1f Telem/s", wps, wave_cyc,
               simd_rate / 1e9, 2.4e9 / simd_rate, elems / 1e12);
    }
    printf("\n");
}

int main() {
    float *out, *in; unsigned long long* cyc;
    CK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float)));
    CK(hipMalloc(&in, 128 * sizeof(float)));
    CK(hipMalloc(&cyc, 256 * 8 * 4 * sizeof(unsigned long long)));
    float h[128];
    for (int i = 0; i < 128; ++i) h[i] = 1e-3f * (i + 1);
    CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
    run<0>("v_add_f32 v,v,v (VOP2)", 1, 1, out, in, cyc);
    run<1>("v_add_f32 v,|v|,v (VOP3 abs)", 1, 1, out, in, cyc);
    run<2>("v_add_f32 v,s,v (SGPR source)", 1, 1, out, in, cyc);
    run<3>("v_fma_f32", 1, 1, out, in, cyc);
    run<4>("v_pk_add_f32", 1, 2, out, in, cyc);
    run<5>("v_pk_fma_f32", 1, 2, out, in, cyc);
    run<6>("L1 mix: v_pk_add(sgpr pair) + 2 v_add|x| per 2 elems", 3, 2, out, in, cyc);
    run<7>("L1 scalar: 2 v_sub(sgpr) + 2 v_add|x| per 2 elems", 4, 2, out, in, cyc);
    run<9>("dot: v_pk_fma_f32 (sgpr pair) per 2 elems", 1, 2, out, in, cyc);
    return 0;
}
