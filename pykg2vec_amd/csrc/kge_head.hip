// kge_head.hip -- the 1-N scoring head of the projection models:
//     preds[B, E] = sigmoid(x[B, d] @ ent[E, d]^T + bias[E])
// ConvE / TuckER / InteractE / HypER / AcrE end their forward with exactly these lines
// (pykg2vec/models/projection.py:100-102, 335-336, 444-447, 606-609, 734-737) and train it with
// Criterion.multi_class_bce (utils/criterion.py:41-49): BCEWithLogitsLoss applied to the SIGMOID OUTPUTS against the
// (label-smoothed) multi-hot rows hr_t / tr_h, mean over B*E.
//
// GEMM-shaped, so it runs on the matrix cores: v_mfma_f32_32x32x2_f32 (exact fp32, 157 TF peak), 64x64 output tile per
// workgroup (4 waves x one 32x32 tile), K staged through LDS in 32-wide slabs.  Operand maps as in kge_dense.hip.
//   k_head_fwd        Z = X Ent^T (+bias) -> sigmoid            [B,d]x[d,E]
//   k_head_bce        the same GEMM with the loss fused into the epilogue: every element is first treated as a negative
//                     (label y0); loss += sum softplus(p) - p*y0 and dz = (sigmoid(p) - y0) p (1-p) / (B*E) is written
//   k_head_bce_pos    one 32-lane group per positive (b, e) of the CSR label lists: recomputes p and applies the
//                     (y1 - y0) correction to the loss and to dz[b, e]  (a positive is touched by exactly one group)
//   k_head_dx         dX = dZ Ent            [B,E]x[E,d]   split over E, atomics into dX
//   k_head_dent       g_ent += dZ^T X        [E,B]x[B,d] ;  g_bias += column sums of dZ
// For the autograd form dZ = dpreds * p * (1 - p) is formed while the operand tile is staged.
#include "kge_internal.h"

namespace kge {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int HT = 64;   // output tile edge per workgroup
constexpr int HK = 32;   // K slab staged in LDS
constexpr int HS = HK + 1;

__device__ __forceinline__ int head_row(int reg, int lk) { return (reg & 3) + 8 * (reg >> 2) + 4 * lk; }
__device__ __forceinline__ float sigmoidf_(float z) { return 1.0f / (1.0f + expf(-z)); }
__device__ __forceinline__ float softplusf_(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }

// one 32-wide K slab: acc += A[32 x 32] B[32 x 32] with A rows / B columns taken from LDS tiles stored [row][k]
__device__ __forceinline__ void slab_mfma(const float* __restrict__ sa, const float* __restrict__ sb, int li, int lk, f32x16& acc) {
#pragma unroll
    for (int kk = 0; kk < HK; kk += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sa[li * HS + kk + lk], sb[li * HS + kk + lk], acc, 0, 0, 0);
}

// Pipelined K loop shared by the three GEMMs: every thread stages HT*HK/256 = 8 elements of each operand tile per slab;
// the NEXT slab's global loads are issued into registers before the current slab's MFMAs, so their latency overlaps the
// matrix work.  fa / fb(slab, j, pos) return the j-th element this thread stages and its LDS position.
constexpr int HPT = HT * HK / 256;

struct NoSlabHook { __device__ __forceinline__ void operator()() const {} };

template <class FA, class FB, class FH = NoSlabHook>
__device__ __forceinline__ void gemm_pipeline(int nslabs, FA fa, FB fb, float* __restrict__ sA, float* __restrict__ sB,
                                              int wr, int wc, int li, int lk, f32x16& acc, FH hook = FH()) {
    float ra[HPT], rb[HPT];
    int pa[HPT], pb[HPT];
#pragma unroll
    for (int j = 0; j < HPT; ++j) { ra[j] = fa(0, j, pa[j]); rb[j] = fb(0, j, pb[j]); }
    for (int sl = 0; sl < nslabs; ++sl) {
#pragma unroll
        for (int j = 0; j < HPT; ++j) { sA[pa[j]] = ra[j]; sB[pb[j]] = rb[j]; }
        __syncthreads();
        if (sl + 1 < nslabs) {
#pragma unroll
            for (int j = 0; j < HPT; ++j) { ra[j] = fa(sl + 1, j, pa[j]); rb[j] = fb(sl + 1, j, pb[j]); }
        }
        hook();  // sees the staged slab (k_head_dent: bias-gradient column sums)
        slab_mfma(sA + wr * 32 * HS, sB + wc * 32 * HS, li, lk, acc);
        __syncthreads();
    }
}

struct HeadArgs {
    const float* x; const float* ent; const float* bias;
    int64_t B, E; int d;
    float* preds;                 // fwd: sigmoid outputs [B,E]
    float* dz;                    // bce: gradient wrt the logits [B,E]
    float y0, inv_count;          // bce: negative label, 1/(B*E)
    float* loss;                  // striped accumulators
};

// MODE 0: preds.  MODE 1: fused multi_class_bce with every label = y0 (positives are corrected by k_head_bce_pos).
template <int MODE>
__global__ __launch_bounds__(256) void k_head_gemm(HeadArgs a) {
    __shared__ float sA[HT * HS], sB[HT * HS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int64_t e0 = (int64_t)blockIdx.x * HT, b0 = (int64_t)blockIdx.y * HT;
    const int wr = wave >> 1, wc = wave & 1;  // wave's 32x32 sub-tile: rows (batch) wr, columns (entities) wc
    f32x16 acc = {0};
    auto fa = [&](int sl, int j, int& pos) -> float {  // x tile [b][k], coalesced along k
        const int idx = threadIdx.x + 256 * j, r = idx / HK, k = idx - r * HK, kg = sl * HK + k;
        pos = r * HS + k;
        return (kg < a.d && b0 + r < a.B) ? a.x[(b0 + r) * a.d + kg] : 0.f;
    };
    auto fb = [&](int sl, int j, int& pos) -> float {  // entity tile [e][k]
        const int idx = threadIdx.x + 256 * j, r = idx / HK, k = idx - r * HK, kg = sl * HK + k;
        pos = r * HS + k;
        return (kg < a.d && e0 + r < a.E) ? a.ent[(e0 + r) * a.d + kg] : 0.f;
    };
    gemm_pipeline((a.d + HK - 1) / HK, fa, fb, sA, sB, wr, wc, li, lk, acc);
    const int64_t e = e0 + wc * 32 + li;
    const float bias = (a.bias && e < a.E) ? a.bias[e] : 0.f;
    float lsum = 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int64_t b = b0 + wr * 32 + head_row(reg, lk);
        if (b < a.B && e < a.E) {
            const float p = sigmoidf_(acc[reg] + bias);
            if constexpr (MODE == 0) {
                a.preds[b * a.E + e] = p;
            } else {
                lsum += softplusf_(p) - p * a.y0;                                   // BCEWithLogits(p, y0)
                a.dz[b * a.E + e] = (sigmoidf_(p) - a.y0) * p * (1.f - p) * a.inv_count;
            }
        }
    }
    if constexpr (MODE == 1) block_accumulate_loss<1>(lsum * a.inv_count, 0, a.loss);
}

// ---- the same GEMM at a 128 x 128 macro-tile (large batches): 4 waves x (2 x 2) accumulators of 32 x 32, i.e. 64 FMAs per
// operand float fetched from LDS instead of 16; both operand tiles are row-major in memory (k contiguous), fetched with one
// 16-byte load per thread and 16-deep K slab and written k-major into LDS ([k][128 + pad]: conflict-free MFMA operand reads);
// slabs double-buffered in LDS with the next slab's global loads in flight during the current slab's 32 MFMAs per wave (one
// barrier per slab).  The accumulation order over k is the old kernel's (one MFMA chain), so the logits are bit-identical.
constexpr int HT2 = 128, HK2 = 16, HLD2 = HT2 + 4;

template <int MODE>
__global__ __launch_bounds__(256, 3) void k_head_gemm128(HeadArgs a) {
    __shared__ float sA[2][HK2][HLD2], sB[2][HK2][HLD2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int64_t e0 = (int64_t)blockIdx.x * HT2, b0 = (int64_t)blockIdx.y * HT2;
    const int wr = wave >> 1, wc = wave & 1;   // wave's 64 x 64 sub-tile: batch rows wr, entity columns wc
    // staging role: float4 number t + 256 j of a slab = (row = idx / 4, k = 4 * (idx % 4) .. + 3)
    const int srow = threadIdx.x >> 2, sk4 = (threadIdx.x & 3) * 4;
    const int nslab = (a.d + HK2 - 1) / HK2;
    float4 ra[2], rb[2];
    auto load_slab = [&](int sl) {
        const int k = sl * HK2 + sk4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t b = b0 + srow + 64 * j, e = e0 + srow + 64 * j;
            const bool live = k < a.d;   // d % 4 == 0: a float4 is inside the row or past its end
            ra[j] = (live && b < a.B) ? *reinterpret_cast<const float4*>(a.x + b * a.d + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            rb[j] = (live && e < a.E) ? *reinterpret_cast<const float4*>(a.ent + e * a.d + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = f32x16{0};
    load_slab(0);
    int buf = 0;
    for (int sl = 0; sl < nslab; ++sl) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = srow + 64 * j;
            sA[buf][sk4 + 0][r] = ra[j].x; sA[buf][sk4 + 1][r] = ra[j].y; sA[buf][sk4 + 2][r] = ra[j].z; sA[buf][sk4 + 3][r] = ra[j].w;
            sB[buf][sk4 + 0][r] = rb[j].x; sB[buf][sk4 + 1][r] = rb[j].y; sB[buf][sk4 + 2][r] = rb[j].z; sB[buf][sk4 + 3][r] = rb[j].w;
        }
        __syncthreads();   // slab sl is in LDS; everybody finished reading the buffer that is written next
        if (sl + 1 < nslab) load_slab(sl + 1);
        // operands of step kk + 2 are read from LDS before the MFMAs of step kk are issued (register double buffer): the LDS
        // latency hides under 4 x 64 cycles of matrix work instead of stalling the wave in front of every quadruple
        float na0 = sA[buf][lk][wr * 64 + li], na1 = sA[buf][lk][wr * 64 + 32 + li];
        float nc0 = sB[buf][lk][wc * 64 + li], nc1 = sB[buf][lk][wc * 64 + 32 + li];
#pragma unroll
        for (int kk = 0; kk < HK2; kk += 2) {
            const float a0 = na0, a1 = na1, c0 = nc0, c1 = nc1;
            if (kk + 2 < HK2) {
                na0 = sA[buf][kk + 2 + lk][wr * 64 + li]; na1 = sA[buf][kk + 2 + lk][wr * 64 + 32 + li];
                nc0 = sB[buf][kk + 2 + lk][wc * 64 + li]; nc1 = sB[buf][kk + 2 + lk][wc * 64 + 32 + li];
            }
            KGE_KEEP_READS_AHEAD();
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, c0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, c1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, c0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, c1, acc[1][1], 0, 0, 0);
        }
        buf ^= 1;
    }
    float lsum = 0.f;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int64_t e = e0 + wc * 64 + ni * 32 + li;
        const float bias = (a.bias && e < a.E) ? a.bias[e] : 0.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int64_t b = b0 + wr * 64 + mi * 32 + head_row(reg, lk);
                if (b < a.B && e < a.E) {
                    const float p = sigmoidf_(acc[mi][ni][reg] + bias);
                    if constexpr (MODE == 0) {
                        a.preds[b * a.E + e] = p;
                    } else {
                        lsum += softplusf_(p) - p * a.y0;
                        a.dz[b * a.E + e] = (sigmoidf_(p) - a.y0) * p * (1.f - p) * a.inv_count;
                    }
                }
            }
    }
    if constexpr (MODE == 1) block_accumulate_loss<1>(lsum * a.inv_count, 0, a.loss);
}

// ---- bf16 option of the forward ("bf16/f32 MFMA GEMM"): the operands are rounded to bfloat16 (round to
// nearest even) while they are staged into LDS, products are exact in fp32 and accumulate in fp32 on
// v_mfma_f32_32x32x16_bf16 (gfx950: 16 k per instruction, 16x the rate of the f32-input form).  Inputs and outputs stay fp32 in
// HBM, so the [B, E] output write (not the matrix cores) bounds it.  NOT the default: the reference computes this head in fp32
// (projection.py:100-102); a logit differs from the fp32 one by ~2^-8 relative per operand (tolerance in the tests).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
constexpr int HSB = HK + 8;   // bf16 elements per LDS row: 80 bytes, 16-byte aligned operand reads

// two floats -> two bfloat16 (round to nearest even) in one VALU instruction: v_cvt_pk_bf16_f32 (gfx950)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f) { return (unsigned short)(pack_bf16x2(f, 0.f) & 0xFFFFu); }

__device__ __forceinline__ float sigmoid_fast(float z) { return __builtin_amdgcn_rcpf(1.0f + __expf(-z)); }   // 1-2 ulp: bf16 option only

__global__ __launch_bounds__(256) void k_head_gemm_bf16(HeadArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned short sA[HT * HSB], sB[HT * HSB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int64_t e0 = (int64_t)blockIdx.x * HT, b0 = (int64_t)blockIdx.y * HT;
    const int wr = wave >> 1, wc = wave & 1;
    f32x16 acc = {0};
    const int nslabs = (a.d + HK - 1) / HK;
    float ra[HPT], rb[HPT];
    auto load = [&](int sl) {
#pragma unroll
        for (int j = 0; j < HPT; ++j) {
            const int idx = threadIdx.x + 256 * j, r = idx / HK, k = idx - r * HK, kg = sl * HK + k;
            ra[j] = (kg < a.d && b0 + r < a.B) ? a.x[(b0 + r) * a.d + kg] : 0.f;
            rb[j] = (kg < a.d && e0 + r < a.E) ? a.ent[(e0 + r) * a.d + kg] : 0.f;
        }
    };
    load(0);
    for (int sl = 0; sl < nslabs; ++sl) {
#pragma unroll
        for (int j = 0; j < HPT; ++j) {
            const int idx = threadIdx.x + 256 * j, r = idx / HK, k = idx - r * HK;
            sA[r * HSB + k] = f32_to_bf16_rne(ra[j]);
            sB[r * HSB + k] = f32_to_bf16_rne(rb[j]);
        }
        __syncthreads();
        if (sl + 1 < nslabs) load(sl + 1);      // next slab's global loads in flight under the MFMAs
#pragma unroll
        for (int kk = 0; kk < HK; kk += 16) {    // lane (li, lk): A row wr*32 + li / B column wc*32 + li, k = kk + 8 lk .. + 7
            const s16x8 va = *reinterpret_cast<const s16x8*>(sA + (wr * 32 + li) * HSB + kk + 8 * lk);
            const s16x8 vb = *reinterpret_cast<const s16x8*>(sB + (wc * 32 + li) * HSB + kk + 8 * lk);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, va), __builtin_bit_cast(bf16x8, vb), acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const int64_t e = e0 + wc * 32 + li;
    const float bias = (a.bias && e < a.E) ? a.bias[e] : 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int64_t b = b0 + wr * 32 + head_row(reg, lk);
        if (b < a.B && e < a.E) a.preds[b * a.E + e] = sigmoid_fast(acc[reg] + bias);
    }
}

// the bf16 forward at a 128 x 128 macro-tile: 4 waves x (2 x 2) accumulators of 32 x 32; operands fetched as float4 (8 threads
// cover 128 contiguous bytes of a row), rounded to bf16 and stored 8 bytes at a time into a ROW-major LDS image ([row][k], 80-byte
// rows) from which a lane reads its 8 consecutive k of an MFMA operand with one 16-byte read; slabs of 32 k double-buffered in LDS,
// the next slab's global loads in flight during the current slab's 8 MFMAs per wave (one barrier per slab).
constexpr int HKB = 32;
// (a variant with the operand loads running TWO slabs ahead through two register sets -- 236 VGPRs, two waves per SIMD -- measured
//  slower: 229 vs 162 us at B = 4096; what bounds this kernel is its epilogue, 61 M sigmoids and a [B, E] store, not its operands)
__global__ __launch_bounds__(256, 2) void k_head_gemm128_bf16(HeadArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned short sA[2][HT2][HSB], sB[2][HT2][HSB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int64_t e0 = (int64_t)blockIdx.x * HT2, b0 = (int64_t)blockIdx.y * HT2;
    const int wr = wave >> 1, wc = wave & 1;
    const int srow = threadIdx.x >> 3, sk4 = (threadIdx.x & 7) * 4;   // float4 number t + 256 j of a slab: row srow + 32 j, k = sk4 .. + 3
    const int nslab = (a.d + HKB - 1) / HKB;
    float4 ra[4], rb[4];
    auto load_slab = [&](int sl) {
        const int k = sl * HKB + sk4;
        const bool live = k < a.d;   // d % 4 == 0: a float4 is inside the row or past its end
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t b = b0 + srow + 32 * j, e = e0 + srow + 32 * j;
            ra[j] = (live && b < a.B) ? *reinterpret_cast<const float4*>(a.x + b * a.d + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            rb[j] = (live && e < a.E) ? *reinterpret_cast<const float4*>(a.ent + e * a.d + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = f32x16{0};
    load_slab(0);
    int buf = 0;
    for (int sl = 0; sl < nslab; ++sl) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<uint2*>(&sA[buf][srow + 32 * j][sk4]) = make_uint2(pack_bf16x2(ra[j].x, ra[j].y), pack_bf16x2(ra[j].z, ra[j].w));
            *reinterpret_cast<uint2*>(&sB[buf][srow + 32 * j][sk4]) = make_uint2(pack_bf16x2(rb[j].x, rb[j].y), pack_bf16x2(rb[j].z, rb[j].w));
        }
        __syncthreads();   // slab sl is in LDS; everybody finished reading the buffer that is written next
        if (sl + 1 < nslab) load_slab(sl + 1);
#pragma unroll
        for (int kk = 0; kk < HKB; kk += 16) {
            const bf16x8 a0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const s16x8*>(&sA[buf][wr * 64 + li][kk + 8 * lk]));
            const bf16x8 a1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const s16x8*>(&sA[buf][wr * 64 + 32 + li][kk + 8 * lk]));
            const bf16x8 c0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const s16x8*>(&sB[buf][wc * 64 + li][kk + 8 * lk]));
            const bf16x8 c1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const s16x8*>(&sB[buf][wc * 64 + 32 + li][kk + 8 * lk]));
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, c0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, c1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, c0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, c1, acc[1][1], 0, 0, 0);
        }
        buf ^= 1;
    }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int64_t e = e0 + wc * 64 + ni * 32 + li;
        const float bias = (a.bias && e < a.E) ? a.bias[e] : 0.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int64_t b = b0 + wr * 64 + mi * 32 + head_row(reg, lk);
                // (plain stores: non-temporal ones measured slower, 149.6 -> 204.4 us at B = 4 096 -- the 128-byte row segments of an
                //  odd-E output are partial lines that need the L2 to merge them; profiles/r05_experiments.md section 8)
                if (b < a.B && e < a.E) a.preds[b * a.E + e] = sigmoid_fast(acc[mi][ni][reg] + bias);
            }
    }
}

// the large tile pays once the grid fills the chip with it and the rows can be fetched 16 bytes at a time
static bool head_use_128(const HeadArgs& a) {
    const int force = switch_value("HEAD_TILE");
    if (force >= 0) return force == 1 && a.d % 4 == 0;
    const int64_t tiles = ((a.E + HT2 - 1) / HT2) * ((a.B + HT2 - 1) / HT2);
    return a.d % 4 == 0 && (((uintptr_t)a.x | (uintptr_t)a.ent) & 15) == 0 && tiles >= 512;
}

template <int MODE>
static void launch_head_gemm(const HeadArgs& a, hipStream_t s) {
    if (head_use_128(a))
        hipLaunchKernelGGL(k_head_gemm128<MODE>, dim3((unsigned)((a.E + HT2 - 1) / HT2), (unsigned)((a.B + HT2 - 1) / HT2)), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(k_head_gemm<MODE>, dim3((unsigned)((a.E + HT - 1) / HT), (unsigned)((a.B + HT - 1) / HT)), dim3(256), 0, s, a);
}

// positives: group of 32 lanes per (b, e) entry of the CSR label lists
__global__ __launch_bounds__(256) void k_head_bce_pos(HeadArgs a, const int64_t* __restrict__ lab_off, const int32_t* __restrict__ lab_ids,
                                                      const int32_t* __restrict__ row_of, int64_t n_pos, float dy) {
    const int gl = threadIdx.x & 31;
    float acc = 0.f;
    for (int64_t j = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5); j < n_pos; j += (int64_t)gridDim.x * 8) {
        const int64_t b = row_of[j], e = lab_ids[j];
        float z = 0.f;
        for (int k = gl; k < a.d; k += 32) z = fmaf(a.x[b * a.d + k], a.ent[e * a.d + k], z);
        z = gsum<32>(z) + (a.bias ? a.bias[e] : 0.f);
        const float p = sigmoidf_(z);
        if (gl == 0) a.dz[b * a.E + e] -= dy * p * (1.f - p) * a.inv_count;  // label y1 instead of y0: d(-p*y)/dp
        acc -= p * dy * a.inv_count;
    }
    (void)lab_off;
    block_accumulate_loss<32>(acc, gl, a.loss);
}

// per-element gradient wrt the logits: dz[b,e] (fused form) or dpreds[b,e] * p (1 - p) (autograd form)
struct DzSrc {
    const float* dz; const float* dpreds; const float* preds;
    __device__ __forceinline__ float at(int64_t i) const {
        if (dz) return dz[i];
        const float p = preds[i];
        return dpreds[i] * p * (1.f - p);
    }
};

// dX[b, k] += sum_{e in this split} dz[b, e] ent[e, k]     grid: (d tiles, B tiles, E splits)
__global__ __launch_bounds__(256) void k_head_dx(DzSrc g, const float* __restrict__ ent, int64_t B, int64_t E, int d,
                                                 int64_t e_per_split, float* __restrict__ dx) {
    __shared__ float sA[HT * HS], sB[HT * HS];  // sA[b][e-slab], sB[k][e-slab]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int64_t k0 = (int64_t)blockIdx.x * HT, b0 = (int64_t)blockIdx.y * HT;
    const int64_t es = (int64_t)blockIdx.z * e_per_split, ee = min(E, es + e_per_split);
    const int wr = wave >> 1, wc = wave & 1;
    f32x16 acc = {0};
    auto fa = [&](int sl, int j, int& pos) -> float {  // dz tile [b][e-slab], coalesced along e
        const int idx = threadIdx.x + 256 * j, r = idx / HK, c = idx - r * HK;
        const int64_t e = es + (int64_t)sl * HK + c;
        pos = r * HS + c;
        return (e < ee && b0 + r < B) ? g.at((b0 + r) * E + e) : 0.f;
    };
    auto fb = [&](int sl, int j, int& pos) -> float {  // entity tile stored [k][e-slab], read coalesced along k
        const int idx = threadIdx.x + 256 * j, c = idx / HT, r = idx - c * HT;
        const int64_t e = es + (int64_t)sl * HK + c;
        pos = r * HS + c;
        return (e < ee && k0 + r < d) ? ent[e * d + k0 + r] : 0.f;
    };
    gemm_pipeline((int)((ee - es + HK - 1) / HK), fa, fb, sA, sB, wr, wc, li, lk, acc);
    const int64_t k = k0 + wc * 32 + li;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int64_t b = b0 + wr * 32 + head_row(reg, lk);
        if (b < B && k < d && acc[reg] != 0.f) unsafeAtomicAdd(dx + b * d + k, acc[reg]);
    }
}

// g_ent[e, k] += sum_{b in this split} dz[b, e] x[b, k] ;  g_bias[e] += sum_b dz[b, e]    grid: (d tiles, E tiles, B splits)
__global__ __launch_bounds__(256) void k_head_dent(DzSrc g, const float* __restrict__ x, int64_t B, int64_t E, int d,
                                                   int64_t b_per_split, float* __restrict__ g_ent, float* __restrict__ g_bias) {
    __shared__ float sA[HT * HS], sB[HT * HS];  // sA[e][b-slab], sB[k][b-slab]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int64_t k0 = (int64_t)blockIdx.x * HT, e0 = (int64_t)blockIdx.y * HT;
    const int64_t bs = (int64_t)blockIdx.z * b_per_split, be = min(B, bs + b_per_split);
    const int wr = wave >> 1, wc = wave & 1;
    f32x16 acc = {0};
    float colsum = 0.f;  // threads 0..63 of the k-tile-0 workgroups: bias gradient of entity e0 + threadIdx.x
    const bool do_bias = g_bias != nullptr && blockIdx.x == 0 && threadIdx.x < HT;
    auto fa = [&](int sl, int j, int& pos) -> float {  // dz tile stored [e][b-slab], read coalesced along e
        const int idx = threadIdx.x + 256 * j, c = idx / HT, r = idx - c * HT;
        const int64_t b = bs + (int64_t)sl * HK + c;
        pos = r * HS + c;
        return (b < be && e0 + r < E) ? g.at(b * E + e0 + r) : 0.f;
    };
    auto fb = [&](int sl, int j, int& pos) -> float {  // x tile stored [k][b-slab], read coalesced along k
        const int idx = threadIdx.x + 256 * j, c = idx / HT, r = idx - c * HT;
        const int64_t b = bs + (int64_t)sl * HK + c;
        pos = r * HS + c;
        return (b < be && k0 + r < d) ? x[b * d + k0 + r] : 0.f;
    };
    auto hook = [&]() {
        if (do_bias) {
#pragma unroll 8
            for (int c = 0; c < HK; ++c) colsum += sA[threadIdx.x * HS + c];
        }
    };
    gemm_pipeline((int)((be - bs + HK - 1) / HK), fa, fb, sA, sB, wr, wc, li, lk, acc, hook);
    const int64_t k = k0 + wc * 32 + li;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int64_t e = e0 + wr * 32 + head_row(reg, lk);
        if (e < E && k < d && acc[reg] != 0.f) unsafeAtomicAdd(g_ent + e * d + k, acc[reg]);
    }
    if (do_bias && e0 + threadIdx.x < E && colsum != 0.f) unsafeAtomicAdd(g_bias + e0 + threadIdx.x, colsum);
}

__global__ void k_head_rows_of(const int64_t* __restrict__ lab_off, int64_t B, int32_t* __restrict__ row_of) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    for (int64_t j = lab_off[b]; j < lab_off[b + 1]; ++j) row_of[j] = (int32_t)b;
}

static int head_check(const char* who, const float* x, const float* ent, int64_t B, int64_t E, int d) {
    if (!x || !ent || B <= 0 || E <= 0 || d <= 0) { set_error("%s: bad arguments", who); return -1; }
    if (B > (int64_t)65535 * HT) { set_error("%s: batch too large", who); return -1; }
    return 0;
}


int launch_head_forward(const float* x, int64_t B, int d, const float* ent, int64_t E, const float* bias, float* preds,
                        int bf16, hipStream_t s) {
    if (head_check("kge_head_1n_forward", x, ent, B, E, d) || !preds) { if (!preds) set_error("kge_head_1n_forward: null output"); return -1; }
    HeadArgs a{};
    a.x = x; a.ent = ent; a.bias = bias; a.B = B; a.E = E; a.d = d; a.preds = preds;
    if (bf16) {
        // (the same accumulation order over k in both tile shapes: 16 k per MFMA, slabs in order -- identical logits)
        if (head_use_128(a))
            hipLaunchKernelGGL(k_head_gemm128_bf16, dim3((unsigned)((E + HT2 - 1) / HT2), (unsigned)((B + HT2 - 1) / HT2)), dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL(k_head_gemm_bf16, dim3((unsigned)((E + HT - 1) / HT), (unsigned)((B + HT - 1) / HT)), dim3(256), 0, s, a);
        return check_launch("k_head_gemm_bf16");
    }
    launch_head_gemm<0>(a, s);
    return check_launch("k_head_gemm<0>");
}

static int head_backward_gemms(const DzSrc& g, const float* x, int64_t B, int d, const float* ent, int64_t E, float* dx,
                               float* g_ent, float* g_bias, hipStream_t s) {
    if (dx) {
        hipError_t e = hipMemsetAsync(dx, 0, (size_t)B * d * sizeof(float), s);
        if (e != hipSuccess) { set_error("kge_head: memset: %s", hipGetErrorString(e)); return -2; }
        // split the long E reduction so that ~8 workgroups per CU exist; every split a multiple of the slab
        const int64_t tiles = ((B + HT - 1) / HT) * ((d + HT - 1) / HT);
        int64_t splits = (2048 + tiles - 1) / tiles;
        const int64_t max_splits = (E + HK - 1) / HK;
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
        int64_t per = ((E + splits - 1) / splits + HK - 1) / HK * HK;
        splits = (E + per - 1) / per;
        hipLaunchKernelGGL(k_head_dx, dim3((unsigned)((d + HT - 1) / HT), (unsigned)((B + HT - 1) / HT), (unsigned)splits),
                           dim3(256), 0, s, g, ent, B, E, d, per, dx);
    }
    if (g_ent) {
        const int64_t tiles = ((E + HT - 1) / HT) * ((d + HT - 1) / HT);
        int64_t splits = (2048 + tiles - 1) / tiles;
        const int64_t max_splits = (B + HK - 1) / HK;
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
        int64_t per = ((B + splits - 1) / splits + HK - 1) / HK * HK;
        splits = (B + per - 1) / per;
        hipLaunchKernelGGL(k_head_dent, dim3((unsigned)((d + HT - 1) / HT), (unsigned)((E + HT - 1) / HT), (unsigned)splits),
                           dim3(256), 0, s, g, x, B, E, d, per, g_ent, g_bias);
    }
    return check_launch("k_head_dx / k_head_dent");
}

int launch_head_backward(const float* x, int64_t B, int d, const float* ent, int64_t E, const float* preds,
                         const float* dpreds, float* dx, float* g_ent, float* g_bias, hipStream_t s) {
    if (head_check("kge_head_1n_backward", x, ent, B, E, d)) return -1;
    if (!preds || !dpreds) { set_error("kge_head_1n_backward: preds / dpreds are null"); return -1; }
    DzSrc g{nullptr, dpreds, preds};
    return head_backward_gemms(g, x, B, d, ent, E, dx, g_ent, g_bias, s);
}

size_t head_bce_workspace_bytes(int64_t B, int64_t E, int64_t n_pos) {
    return ((size_t)B * E * sizeof(float) + 255) / 256 * 256 + ((size_t)n_pos * sizeof(int32_t) + 255) / 256 * 256;
}

int launch_head_bce(const float* x, int64_t B, int d, const float* ent, int64_t E, const float* bias,
                    const int64_t* lab_off, const int32_t* lab_ids, int64_t n_pos, float label_smoothing, void* ws,
                    size_t ws_bytes, float* loss, float* dx, float* g_ent, float* g_bias, hipStream_t s) {
    if (head_check("kge_head_1n_bce", x, ent, B, E, d)) return -1;
    if (!lab_off || (n_pos > 0 && !lab_ids) || !loss || n_pos < 0) { set_error("kge_head_1n_bce: bad arguments"); return -1; }
    if (!ws || ws_bytes < head_bce_workspace_bytes(B, E, n_pos)) {
        set_error("kge_head_1n_bce: workspace too small (need kge_head_1n_bce_workspace_bytes)");
        return -1;
    }
    float* dz = (float*)ws;
    int32_t* row_of = (int32_t*)((char*)ws + ((size_t)B * E * sizeof(float) + 255) / 256 * 256);
    // Criterion.multi_class_bce (utils/criterion.py:42-45): y <- y (1 - ls) + 1/E when label smoothing is given
    const bool smooth = label_smoothing >= 0.f;
    const float y0 = smooth ? 1.0f / (float)E : 0.f;
    const float y1 = smooth ? (1.0f - label_smoothing) + 1.0f / (float)E : 1.f;
    HeadArgs a{};
    a.x = x; a.ent = ent; a.bias = bias; a.B = B; a.E = E; a.d = d; a.dz = dz; a.loss = loss;
    a.y0 = y0; a.inv_count = 1.0f / ((float)B * (float)E);
    launch_head_gemm<1>(a, s);
    if (n_pos > 0) {
        hipLaunchKernelGGL(k_head_rows_of, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, lab_off, B, row_of);
        int64_t blocks = (n_pos + 7) / 8;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(k_head_bce_pos, dim3((unsigned)blocks), dim3(256), 0, s, a, lab_off, lab_ids, row_of, n_pos, y1 - y0);
    }
    DzSrc g{dz, nullptr, nullptr};
    return head_backward_gemms(g, x, B, d, ent, E, dx, g_ent, g_bias, s);
}

}  // namespace kge
