// kge_head.hip -- the 1-N scoring head of the projection models:
//     preds[B, E] = sigmoid(x[B, d] @ ent[E, d]^T + bias[E])
// ConvE / TuckER / InteractE / HypER / AcrE end their forward with exactly these lines
// (pykg2vec/models/projection.py:100-102, 335-336, 444-447, 606-609, 734-737) and train it with
// Criterion.multi_class_bce (utils/criterion.py:41-49): BCEWithLogitsLoss applied to the SIGMOID OUTPUTS against the
// (label-smoothed) multi-hot rows hr_t / tr_h, mean over B*E.
//
// GEMM-shaped, so it runs on the f32 matrix cores (exact fp32 products, 157 TF peak), in two tile shapes:
//   small batches (the reference's B = 128)   64 x 64 outputs per workgroup on v_mfma_f32_32x32x2_f32, K in 32-wide LDS slabs
//       k_head_gemm<MODE>, k_head_dx, k_head_dent (split-K partial tiles combined with float atomics)
//   large batches                             128 x 128 per workgroup on v_mfma_f32_16x16x4_f32, one K loop (gemm128x_core)
//       k_head_gemm128x<MODE>, k_head_dx128, k_head_dent128 (+ k_head_sum_parts: split-K partial tiles added in split order --
//       no atomics, reproducible gradients)
// MODE 0  Z = X Ent^T (+bias) -> sigmoid                        [B,d]x[d,E]
// MODE 1  the same GEMM with the loss fused into the epilogue: every element is first treated as a negative (label y0);
//         loss += sum softplus(p) - p*y0 and dz = (sigmoid(p) - y0) p (1-p) / (B*E) is written (never the predictions)
//   k_head_bce_pos    one 32-lane group per positive (b, e) of the CSR label lists: recomputes p and applies the
//                     (y1 - y0) correction to the loss and to dz[b, e]  (a positive is touched by exactly one group)
//   dX = dZ Ent            [B,E]x[E,d]   split over E
//   g_ent += dZ^T X        [E,B]x[B,d]   split over B;  g_bias += column sums of dZ
// For the autograd form dZ = dpreds * p * (1 - p) is formed while the operand tile is staged.
// bf16 option of the forward: k_head_gemm_bf16 / k_head_gemm128_bf16 (v_mfma_f32_32x32x16_bf16).
#include "kge_internal.h"
#include <type_traits>

namespace kge {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int HT = 64;   // output tile edge per workgroup
constexpr int HK = 32;   // K slab staged in LDS
constexpr int HS = HK + 1;

__device__ __forceinline__ int head_row(int reg, int lk) { return (reg & 3) + 8 * (reg >> 2) + 4 * lk; }
__device__ __forceinline__ float sigmoidf_(float z) { return 1.0f / (1.0f + expf(-z)); }
// Criterion.multi_class_bce on a NEGATIVE label y0 (utils/criterion.py:41-49: BCEWithLogitsLoss applied to the sigmoid OUTPUT p):
// loss term softplus(p) - p y0 and its derivative through the sigmoid, (sigmoid(p) - y0) p (1 - p).  With t = exp(-p), p in (0, 1):
// softplus(p) = p + log(1 + t), sigmoid(p) = 1 / (1 + t), and 1 + t is in (1.36, 2] -- no cancellation, so the hardware exp / log /
// rcp (1-2 ulp each) serve: these terms enter sums over B E elements.  p itself, the model output, keeps the exact sigmoidf_.
__device__ __forceinline__ float bce_neg_terms(float p, float y0, float& loss_term) {
    const float u = 1.0f + __expf(-p);
    loss_term = p + __logf(u) - p * y0;
    return (__builtin_amdgcn_rcpf(u) - y0) * p * (1.f - p);
}

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
    float* dz;                    // bce: gradient wrt the logits [B, ldz]: rows padded to a multiple of four floats, so that the
    int64_t ldz;                  //      backward products fetch them with aligned 16-byte loads (E is odd in every dataset)
    float y0, inv_count;          // bce: negative label, 1/(B*E)
    float* loss;                  // striped accumulators
    int xcd_runs;                 // tile numbering: 1 = one contiguous run of tiles per XCD (head_tile)
};

// Workgroup -> output tile.  Consecutive workgroup ids go round robin over the 8 XCDs, each with its own L2.  E is odd in every
// dataset of the reference (14 951, 40 943), so the 128-byte row segments a wave stores straddle cache lines and the tiles either
// side of a column boundary share a line: with the plain (x = entity tile, y = batch tile) grid those two tiles sit on DIFFERENT
// XCDs and the line leaves both L2s as a partial write.  Tiles are therefore numbered so that every XCD owns one contiguous run of
// tiles (entity tile fastest): neighbours along a row are written from one L2, which merges them, and share the x rows.
__device__ __forceinline__ void head_tile(const HeadArgs& a, int tile, int64_t& e0, int64_t& b0) {
    const int64_t nte = (a.E + tile - 1) / tile, ntb = (a.B + tile - 1) / tile, T = nte * ntb;
    int64_t t = blockIdx.x;
    if (a.xcd_runs) {
        const int64_t x = t & 7, slot = t >> 3, lo = T >> 3, rem = T & 7;
        t = x * lo + (x < rem ? x : rem) + slot;
    }
    const int64_t bt = t / nte;
    e0 = (t - bt * nte) * tile; b0 = bt * tile;
}

// MODE 0: preds.  MODE 1: fused multi_class_bce with every label = y0 (positives are corrected by k_head_bce_pos).
template <int MODE>
__global__ __launch_bounds__(256) void k_head_gemm(HeadArgs a) {
    __shared__ float sA[HT * HS], sB[HT * HS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lk = lane >> 5;
    int64_t e0, b0;
    head_tile(a, HT, e0, b0);
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
                float lt;
                a.dz[b * a.ldz + e] = bce_neg_terms(p, a.y0, lt) * a.inv_count;
                lsum += lt;
            }
        }
    }
    if constexpr (MODE == 1) block_accumulate_loss<1>(lsum * a.inv_count, 0, a.loss);
}

constexpr int HT2 = 128, HK2 = 16;   // the large tile: 128 x 128 outputs per workgroup, K slabs of 16

// ---- the 128 x 128 macro-tile on v_mfma_f32_16x16x4_f32 (the form kge_eval.hip::k_eval_gemm settled on: 40-cycle dependent latency
// instead of 64, holds its issue rate with four waves per SIMD -- tools/mfma_bench.hip 154 vs 129 TF -- and 64 accumulator registers
// as 4 x 4 blocks of 16 x 16 leave room for four workgroups per CU).  Row block mi of a wave holds batch rows 4 r + mi, column block
// ni entity columns 4 c + ni (r, c = row / column inside the MFMA block): a lane's four A operands (and four B operands) of a k-step
// are 16 consecutive bytes of the k-major slab, and its four results of one batch row are four CONSECUTIVE entities -- one 16-byte
// store (4-byte aligned: E is odd, global_store_dwordx4 takes it), 256 contiguous bytes of a row per 16 lanes.  One MFMA chain over
// k per element, like the other forms: bit-identical logits.
// workgroups per CU the register allocation must allow: four for the forward, three for the backward products (168 VGPRs: at four
// the loaders' 64-bit indices spill, and a spill reload inside the K loop waits for every load in flight)
#define HEAD_OCC 4
#define HEAD_BWD_OCC 3
constexpr int HLD3 = HT2 + 16;   // 16 lanes read 16 consecutive floats of row k, the next 16 lanes row k + 1: rows 16 banks apart
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) float4u { float x, y, z, w; };

// Loads of the 128-wide kernels are UNCONDITIONAL on clamped addresses, and what they fetched is masked one slab LATER, where it is
// stored into LDS (the "fix" functions of gemm128x_core): a load under a lane mask, inside a branch, or with a select on its result
// makes the compiler wait for it on the spot, and the point of the K loop is that a slab's loads stay in flight for a whole slab
// of matrix work.
__device__ __forceinline__ float4 load4_clamped(const float* __restrict__ p, int64_t i, bool ok) {
    return *reinterpret_cast<const float4*>(p + (ok ? i : 0));
}
__device__ __forceinline__ float4 keep_if(float4 v, bool ok) {
    v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
    return v;
}
__device__ __forceinline__ float4 first_n(float4 v, int nv) {
    v.x = nv > 0 ? v.x : 0.f; v.y = nv > 1 ? v.y : 0.f; v.z = nv > 2 ? v.z : 0.f; v.w = nv > 3 ? v.w : 0.f;
    return v;
}
// One operand float4 of a 16-deep slab goes to the k-major LDS image [k][128 + pad].  Two memory layouts feed it:
//   RK  "row-major, k contiguous" (x[b][k], ent[e][k], dz[b][e] as the A operand of dX): thread t stages float4 number t + 256 j =
//       (row t / 4 + 64 j, k = 4 (t % 4) .. + 3) -- four scalar LDS stores (the transposition);
//   KM  "k-major" (ent[e][k] as the B operand of dX, dz[b][e] and x[b][k] in the entity-gradient product): float4 number t + 256 j =
//       (k = idx / 32, rows 4 (idx % 32) .. + 3) -- one 16-byte LDS store.
template <bool KM>
__device__ __forceinline__ void stage_store(float (*sX)[HLD3], int j, const float4& v) {
    if constexpr (KM) {
        const int idx = threadIdx.x + 256 * j;
        *reinterpret_cast<float4*>(&sX[idx >> 5][(idx & 31) * 4]) = v;
    } else {
        const int r = (threadIdx.x >> 2) + 64 * j, k = (threadIdx.x & 3) * 4;
        sX[k + 0][r] = v.x; sX[k + 1][r] = v.y; sX[k + 2][r] = v.z; sX[k + 3][r] = v.w;
    }
}
struct NoSlabHookX { __device__ __forceinline__ void operator()(float (*)[HLD3]) const {} };

// K loop of a 128 x 128 tile: la / lb(slab, j) return the j-th float4 this thread stages of the A / B operand (zeros outside the
// operand); slabs double-buffered in LDS, the next slab's global loads in flight during the current slab's 64 MFMAs per wave (one
// barrier per slab), LDS operands of k-step kk + 4 read before the MFMAs of k-step kk are issued.  acc[mi][ni][reg] = element
// (row 64 wr + 4 (4 lk4 + reg) + mi, column 64 wc + 4 lcol + ni) of the tile.
template <bool A_KM, bool B_KM, class LA, class FA, class LB, class FB, class HOOK = NoSlabHookX>
__device__ __forceinline__ void gemm128x_core(int nslab, LA la, FA fa, LB lb, FB fb, float (&sA)[2][HK2][HLD3],
                                              float (&sB)[2][HK2][HLD3], int wr, int wc, int lcol, int lk4, f32x4 (&acc)[4][4],
                                              HOOK hook = HOOK()) {
    // la / lb(slab, j): the raw fetch of the j-th float4 this thread stages; fa / fb(slab, j, raw): its value (see load4_clamped)
    decltype(la(0, 0)) ra[2];
    decltype(lb(0, 0)) rb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { ra[j] = la(0, j); rb[j] = lb(0, j); }
#pragma unroll
    for (int j = 0; j < 2; ++j) { stage_store<A_KM>(sA[0], j, fa(0, j, ra[j])); stage_store<B_KM>(sB[0], j, fb(0, j, rb[j])); }
    if (1 < nslab) {
#pragma unroll
        for (int j = 0; j < 2; ++j) { ra[j] = la(1, j); rb[j] = lb(1, j); }
    }
    __syncthreads();
    int buf = 0;
    for (int sl = 0; sl < nslab; ++sl) {
        // LDS buffer `buf` holds slab sl; the registers hold slab sl + 1 (requested one slab ago)
        hook(sA[buf]);
        float na[4], nb[4];
#define KGE_HEAD_READ(K)                                                                                                   \
    {                                                                                                                      \
        const float4 va = *reinterpret_cast<const float4*>(&sA[buf][(K) + lk4][wr * 64 + 4 * lcol]);                       \
        const float4 vb = *reinterpret_cast<const float4*>(&sB[buf][(K) + lk4][wc * 64 + 4 * lcol]);                       \
        na[0] = va.x; na[1] = va.y; na[2] = va.z; na[3] = va.w;                                                            \
        nb[0] = vb.x; nb[1] = vb.y; nb[2] = vb.z; nb[3] = vb.w;                                                            \
    }
        KGE_HEAD_READ(0)
#pragma unroll
        for (int kk = 0; kk < HK2; kk += 4) {
            float a4[4], b4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a4[i] = na[i]; b4[i] = nb[i]; }
            if (kk + 4 < HK2) KGE_HEAD_READ(kk + 4)
            KGE_KEEP_READS_AHEAD();
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[mi], b4[ni], acc[mi][ni], 0, 0, 0);
            if (kk == 0) {
                // behind the first quarter of this slab's MFMAs (so that the matrix pipe has work queued while this wave stores):
                // slab sl + 1 goes from the registers into the OTHER buffer -- free since the barrier that ended slab sl - 1 -- and
                // slab sl + 2 is requested.  Workgroups of one CU run in step; a store phase in front of the barrier (the round-4
                // form of this loop) left the matrix pipe idle in all of them at once.
                KGE_KEEP_READS_AHEAD();
                if (sl + 1 < nslab) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) { stage_store<A_KM>(sA[buf ^ 1], j, fa(sl + 1, j, ra[j])); stage_store<B_KM>(sB[buf ^ 1], j, fb(sl + 1, j, rb[j])); }
                    if (sl + 2 < nslab) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) { ra[j] = la(sl + 2, j); rb[j] = lb(sl + 2, j); }
                    }
                }
                KGE_KEEP_READS_AHEAD();
            }
        }
#undef KGE_HEAD_READ
        __syncthreads();   // slab sl + 1 is in LDS; everybody finished reading slab sl
        buf ^= 1;
    }
}

template <int MODE>
__global__ __launch_bounds__(256, HEAD_OCC) void k_head_gemm128x(HeadArgs a) {
    __shared__ __attribute__((aligned(16))) float sA[2][HK2][HLD3], sB[2][HK2][HLD3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lcol = lane & 15, lk4 = lane >> 4;
    int64_t e0, b0;
    head_tile(a, HT2, e0, b0);
    const int wr = wave >> 1, wc = wave & 1;   // wave's 64 x 64 sub-tile: batch rows wr, entity columns wc
    const int srow = threadIdx.x >> 2, sk4 = (threadIdx.x & 3) * 4;
    // (d % 4 == 0: a float4 is inside the row or past its end)
    auto oka = [&](int sl, int j) { return sl * HK2 + sk4 < a.d && b0 + srow + 64 * j < a.B; };
    auto okb = [&](int sl, int j) { return sl * HK2 + sk4 < a.d && e0 + srow + 64 * j < a.E; };
    auto la = [&](int sl, int j) { return load4_clamped(a.x, (b0 + srow + 64 * j) * a.d + sl * HK2 + sk4, oka(sl, j)); };
    auto lb = [&](int sl, int j) { return load4_clamped(a.ent, (e0 + srow + 64 * j) * a.d + sl * HK2 + sk4, okb(sl, j)); };
    auto fa = [&](int sl, int j, const float4& v) { return keep_if(v, oka(sl, j)); };
    auto fb = [&](int sl, int j, const float4& v) { return keep_if(v, okb(sl, j)); };
    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    gemm128x_core<false, false>((a.d + HK2 - 1) / HK2, la, fa, lb, fb, sA, sB, wr, wc, lcol, lk4, acc);
    // the lane owns entity columns e .. e + 3 (column blocks ni = 0 .. 3) of batch rows 4 (4 lk4 + reg) + mi
    const int64_t e = e0 + wc * 64 + 4 * lcol;
    float bias[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) bias[ni] = (a.bias && e + ni < a.E) ? a.bias[e + ni] : 0.f;
    const bool whole = e + 3 < a.E;
    float lsum = 0.f;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int64_t b = b0 + wr * 64 + 4 * (4 * lk4 + reg) + mi;
            if (b >= a.B || e >= a.E) continue;
            float o[4];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const float p = sigmoidf_(acc[mi][ni][reg] + bias[ni]);
                if constexpr (MODE == 0) {
                    o[ni] = p;
                } else {
                    float lt;
                    o[ni] = bce_neg_terms(p, a.y0, lt) * a.inv_count;
                    if (e + ni < a.E) lsum += lt;
                }
            }
            float* const dst = MODE == 0 ? a.preds + b * a.E + e : a.dz + b * a.ldz + e;
            if (whole) {
                *reinterpret_cast<float4u*>(dst) = float4u{o[0], o[1], o[2], o[3]};
            } else {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    if (e + ni < a.E) dst[ni] = o[ni];
            }
        }
    if constexpr (MODE == 1) block_accumulate_loss<1>(lsum * a.inv_count, 0, a.loss);
}

// ---- the same loop at a 64 x 64 tile for small batches (the reference's B = 128 at E = 14 951: 2 x 234 tiles): 4 waves x (2 x 2
// blocks of 16 x 16), K slabs of 32 (two float4 per thread and operand), rows 2 r + mi / columns 2 c + ni per block so that a lane's
// two A (two B) operands of a k-step are one 8-byte LDS read and its two results of a row are consecutive entities.  The 32x32x2
// kernel it replaces (k_head_gemm, kept for hidden sizes that are no multiple of 4) fetched its slabs with scalar loads under a
// select: synchronous, 21.5 us at B = 128 against 11 us here.
constexpr int HK4 = 32, HLD4 = HT + 32;   // 96-float rows: the four k rows a wave reads per k-step sit 32 banks apart
struct __attribute__((packed, aligned(4))) float2u { float x, y; };

template <int MODE>
__global__ __launch_bounds__(256, 3) void k_head_gemm64x(HeadArgs a) {
    __shared__ __attribute__((aligned(16))) float sA[2][HK4][HLD4], sB[2][HK4][HLD4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lcol = lane & 15, lk4 = lane >> 4;
    int64_t e0, b0;
    head_tile(a, HT, e0, b0);
    const int wr = wave >> 1, wc = wave & 1;   // wave's 32 x 32 sub-tile: batch rows wr, entity columns wc
    const int srow = threadIdx.x >> 2, sk4 = (threadIdx.x & 3) * 4;   // float4 j of a slab: row srow, k = sk4 + 16 j .. + 3
    const int nslab = (a.d + HK4 - 1) / HK4;
    const bool rowa = b0 + srow < a.B, rowb = e0 + srow < a.E;
    const float* const pa = a.x + (rowa ? (b0 + srow) * a.d : 0);
    const float* const pb = a.ent + (rowb ? (e0 + srow) * a.d : 0);
    // THREE slabs of operands in registers (slab s in set s % 3): one being stored, two in flight -- the matrix work of a slab (32
    // MFMAs per wave) is far shorter than a round trip, and this tile leaves the registers for it (64 VGPRs)
    float4 ra[3][2], rb[3][2];
    // (d % 4 == 0: a float4 is inside the row or past its end; past it the row's first float4 is fetched and dropped at the store)
    auto load_slab = [&](int sl, auto set) {
        constexpr int P = decltype(set)::value;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = sl * HK4 + sk4 + 16 * j;
            ra[P][j] = *reinterpret_cast<const float4*>(pa + (k < a.d ? k : 0));
            rb[P][j] = *reinterpret_cast<const float4*>(pb + (k < a.d ? k : 0));
        }
    };
    auto store_slab = [&](int sl, int buf, auto set) {
        constexpr int P = decltype(set)::value;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kk = sk4 + 16 * j;
            const bool live = sl * HK4 + kk < a.d;
            const float4 va = keep_if(ra[P][j], live && rowa), vb = keep_if(rb[P][j], live && rowb);
            sA[buf][kk + 0][srow] = va.x; sA[buf][kk + 1][srow] = va.y; sA[buf][kk + 2][srow] = va.z; sA[buf][kk + 3][srow] = va.w;
            sB[buf][kk + 0][srow] = vb.x; sB[buf][kk + 1][srow] = vb.y; sB[buf][kk + 2][srow] = vb.z; sB[buf][kk + 3][srow] = vb.w;
        }
    };
    using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>; using S2 = std::integral_constant<int, 2>;
    f32x4 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    load_slab(0, S0());
    if (1 < nslab) load_slab(1, S1());
    if (2 < nslab) load_slab(2, S2());
    store_slab(0, 0, S0());
    __syncthreads();
    int buf = 0;
    auto step = [&](int sl, auto set, auto next) {   // set = sl % 3, next = (sl + 1) % 3
        float2 na, nb;
#define KGE_HEAD_READ2(K)                                                                                                  \
    {                                                                                                                      \
        na = *reinterpret_cast<const float2*>(&sA[buf][(K) + lk4][wr * 32 + 2 * lcol]);                                    \
        nb = *reinterpret_cast<const float2*>(&sB[buf][(K) + lk4][wc * 32 + 2 * lcol]);                                    \
    }
        KGE_HEAD_READ2(0)
#pragma unroll
        for (int kk = 0; kk < HK4; kk += 4) {
            const float2 a2 = na, b2 = nb;
            if (kk + 4 < HK4) KGE_HEAD_READ2(kk + 4)
            KGE_KEEP_READS_AHEAD();
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.x, b2.x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.x, b2.y, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.y, b2.x, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.y, b2.y, acc[1][1], 0, 0, 0);
            if (kk == 0) {   // (as in gemm128x_core: the next slab is stored behind the first MFMAs of this one)
                KGE_KEEP_READS_AHEAD();
                if (sl + 1 < nslab) {
                    store_slab(sl + 1, buf ^ 1, next);
                    if (sl + 3 < nslab) load_slab(sl + 3, set);   // (this set's slab went to LDS one step ago)
                }
                KGE_KEEP_READS_AHEAD();
            }
        }
#undef KGE_HEAD_READ2
        __syncthreads();
        buf ^= 1;
    };
    for (int sl = 0; sl < nslab; sl += 3) {
        step(sl, S0(), S1());
        if (sl + 1 < nslab) step(sl + 1, S1(), S2());
        if (sl + 2 < nslab) step(sl + 2, S2(), S0());
    }
    // the lane owns entity columns e, e + 1 (column blocks ni = 0, 1) of batch rows 2 (4 lk4 + reg) + mi
    const int64_t e = e0 + wc * 32 + 2 * lcol;
    float bias[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) bias[ni] = (a.bias && e + ni < a.E) ? a.bias[e + ni] : 0.f;
    const bool whole = e + 1 < a.E;
    float lsum = 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int64_t b = b0 + wr * 32 + 2 * (4 * lk4 + reg) + mi;
            if (b >= a.B || e >= a.E) continue;
            float o[2];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const float p = sigmoidf_(acc[mi][ni][reg] + bias[ni]);
                if constexpr (MODE == 0) {
                    o[ni] = p;
                } else {
                    float lt;
                    o[ni] = bce_neg_terms(p, a.y0, lt) * a.inv_count;
                    if (e + ni < a.E) lsum += lt;
                }
            }
            float* const dst = MODE == 0 ? a.preds + b * a.E + e : a.dz + b * a.ldz + e;
            if (whole) *reinterpret_cast<float2u*>(dst) = float2u{o[0], o[1]};
            else dst[0] = o[0];
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
    int64_t e0, b0;
    head_tile(a, HT, e0, b0);
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
    int64_t e0, b0;
    head_tile(a, HT2, e0, b0);
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

static dim3 head_grid(const HeadArgs& a, int tile) { return dim3((unsigned)(((a.E + tile - 1) / tile) * ((a.B + tile - 1) / tile))); }

// the large tile pays once its grid covers most of the chip (measured from B = 256 at E = 14 951: 234 tiles) and the rows can be
// fetched 16 bytes at a time
static bool head_use_128(const HeadArgs& a) {
    const int force = switch_value("HEAD_TILE");
    if (force >= 0) return force == 1 && a.d % 4 == 0;
    const int64_t tiles = ((a.E + HT2 - 1) / HT2) * ((a.B + HT2 - 1) / HT2);
    return a.d % 4 == 0 && (((uintptr_t)a.x | (uintptr_t)a.ent) & 15) == 0 && tiles >= 200;
}

template <int MODE>
static void launch_head_gemm(const HeadArgs& a, hipStream_t s) {
    if (head_use_128(a)) {
        hipLaunchKernelGGL(k_head_gemm128x<MODE>, head_grid(a, HT2), dim3(256), 0, s, a);
    }
    else if (a.d % 4 == 0 && (((uintptr_t)a.x | (uintptr_t)a.ent) & 15) == 0 && switch_value("HEAD_SMALL_X") != 0)
        hipLaunchKernelGGL(k_head_gemm64x<MODE>, head_grid(a, HT), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(k_head_gemm<MODE>, head_grid(a, HT), dim3(256), 0, s, a);
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
        if (gl == 0) a.dz[b * a.ldz + e] -= dy * p * (1.f - p) * a.inv_count;  // label y1 instead of y0: d(-p*y)/dp
        acc -= p * dy * a.inv_count;
    }
    (void)lab_off;
    block_accumulate_loss<32>(acc, gl, a.loss);
}

// per-element gradient wrt the logits: dz[b,e] (fused form) or dpreds[b,e] * p (1 - p) (autograd form)
struct DzSrc {
    const float* dz; const float* dpreds; const float* preds;
    int64_t ld;   // row stride of whichever source is set (dz: padded workspace rows; preds / dpreds: E)
    __device__ __forceinline__ float at(int64_t i) const {
        if (dz) return dz[i];
        const float p = preds[i];
        return dpreds[i] * p * (1.f - p);
    }
};

// four consecutive elements of a gradient row for the 128-wide products.  FUSED (the workspace of kge_head_1n_bce): rows are padded
// to a multiple of four floats and i is a multiple of four, so this is ONE aligned 16-byte load with no control flow around it (the
// load stays in flight across the slab it is issued in); elements past the operand -- pad columns are never written -- are masked.
// four consecutive elements of a gradient row, the first nv of them inside the operand, in two steps: the raw fetch (dz_fetch4)
// and, at the LDS store, the value (dz_value4).  FUSED (the workspace of kge_head_1n_bce): rows are padded to a multiple of four
// floats and i is a multiple of four -- one aligned 16-byte load; pad columns are never written, hence the mask.  Otherwise
// (autograd form) dpreds * p (1 - p) from two 4-byte aligned loads.
struct DzRaw { float4 a, b; };
template <bool FUSED>
__device__ __forceinline__ DzRaw dz_fetch4(const DzSrc& g, int64_t i, int nv, int64_t total) {
    DzRaw r;
    const int64_t ii = nv > 0 ? i : 0;
    if constexpr (FUSED) {
        r.a = *reinterpret_cast<const float4*>(g.dz + ii);
        r.b = r.a;
    } else {
        const int64_t ic = ii < total - 4 ? ii : total - 4;   // (the tail of the last row: fetched from the last in-bounds position)
        const float4u p = *reinterpret_cast<const float4u*>(g.preds + ic), q = *reinterpret_cast<const float4u*>(g.dpreds + ic);
        r.a = make_float4(p.x, p.y, p.z, p.w); r.b = make_float4(q.x, q.y, q.z, q.w);
    }
    return r;
}
__device__ __forceinline__ float4 shift4(float4 v, int sh) {
    float4 r;
    r.x = sh == 0 ? v.x : (sh == 1 ? v.y : (sh == 2 ? v.z : v.w));
    r.y = sh == 0 ? v.y : (sh == 1 ? v.z : v.w);
    r.z = sh == 0 ? v.z : v.w;
    r.w = v.w;
    return r;
}
template <bool FUSED>
__device__ __forceinline__ float4 dz_value4(const DzRaw& w, int64_t i, int nv, int64_t total) {
    if constexpr (FUSED) {
        return first_n(w.a, nv);
    } else {
        const int64_t ii = nv > 0 ? i : 0;
        const int sh = ii < total - 4 ? 0 : (int)(ii - (total - 4));
        const float4 p = shift4(w.a, sh), q = shift4(w.b, sh);
        float4 r;
        r.x = q.x * p.x * (1.f - p.x); r.y = q.y * p.y * (1.f - p.y); r.z = q.z * p.z * (1.f - p.z); r.w = q.w * p.w * (1.f - p.w);
        return first_n(r, nv);
    }
}

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
        return (e < ee && b0 + r < B) ? g.at((b0 + r) * g.ld + e) : 0.f;
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
        return (b < be && e0 + r < E) ? g.at(b * g.ld + e0 + r) : 0.f;
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

// What a split-K workgroup of the backward products does with its 128 x 128 tile:
//   kOutDirect  the product is not split: the tile IS the result -- plain 16-byte stores (dX) or read-modify-writes (g_ent +=);
//   kOutParts   the tile goes to its slot of the partials buffer (workgroup number x 64 KB) and k_head_sum_parts adds the splits in
//               split order: no atomics, bit-reproducible gradients (the fused entry point, which owns a workspace);
//   kOutAtomic  float atomics into the result (kge_head_1n_backward called without its optional workspace).
constexpr int kOutDirect = 0, kOutParts = 1, kOutAtomic = 2;
constexpr int64_t kHeadPartSlots = 768;   // split-K workgroups per product (three per CU)

__device__ __forceinline__ void tile_out(const f32x4 (&acc)[4][4], int emode, float* __restrict__ part, float* __restrict__ out,
                                         int64_t m0, int64_t M, int64_t n0, int N, int wr, int wc, int lcol, int lk4, bool accumulate) {
    const int col = wc * 64 + 4 * lcol;
    const int64_t n = n0 + col;
    if (emode == kOutParts) {
        float* const pt = part + ((((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) << 14);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = wr * 64 + 4 * (4 * lk4 + reg) + mi;
                *reinterpret_cast<float4*>(pt + row * HT2 + col) = make_float4(acc[mi][0][reg], acc[mi][1][reg], acc[mi][2][reg], acc[mi][3][reg]);
            }
        return;
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int64_t m = m0 + wr * 64 + 4 * (4 * lk4 + reg) + mi;
            if (m >= M || n >= N) continue;   // (N % 4 == 0: the lane's four columns are inside the row or past its end)
            float* const o = out + m * N + n;
            if (emode == kOutDirect) {
                float4 v = make_float4(acc[mi][0][reg], acc[mi][1][reg], acc[mi][2][reg], acc[mi][3][reg]);
                if (accumulate) { const float4 w = *reinterpret_cast<const float4*>(o); v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
                *reinterpret_cast<float4*>(o) = v;
            } else {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    if (acc[mi][ni][reg] != 0.f) unsafeAtomicAdd(o + ni, acc[mi][ni][reg]);
            }
        }
}

// ---- the two backward products at the 128 x 128 tile of k_head_gemm128x (large batches; d % 4 == 0)
// dX[b, k] += sum_{e in this split} dz[b, e] ent[e, k]     grid: (d tiles, B tiles, E splits); A = dz rows (RK), B = ent (KM)
template <bool FUSED>
__global__ __launch_bounds__(256, HEAD_BWD_OCC) void k_head_dx128(DzSrc g, const float* __restrict__ ent, int64_t B, int64_t E, int d,
                                                       int64_t e_per_split, float* __restrict__ dx, int emode, float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) float sA[2][HK2][HLD3], sB[2][HK2][HLD3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lcol = lane & 15, lk4 = lane >> 4;
    const int64_t k0 = (int64_t)blockIdx.x * HT2, b0 = (int64_t)blockIdx.y * HT2;
    const int64_t es = (int64_t)blockIdx.z * e_per_split, ee = min(E, es + e_per_split);
    const int wr = wave >> 1, wc = wave & 1;
    const int srow = threadIdx.x >> 2, sk4 = (threadIdx.x & 3) * 4;
    const int64_t total = B * E;
    auto ia = [&](int sl, int j, int& nv) {   // element index of this thread's dz float4 and how many of its elements count
        const int64_t b = b0 + srow + 64 * j, e = es + (int64_t)sl * HK2 + sk4;
        const int64_t n = b < B ? ee - e : 0;
        nv = n > 4 ? 4 : (int)n;
        return b * g.ld + e;
    };
    auto ib = [&](int sl, int j, bool& ok) {
        const int idx = threadIdx.x + 256 * j;
        const int64_t e = es + (int64_t)sl * HK2 + (idx >> 5), k = k0 + (idx & 31) * 4;
        ok = e < ee && k < d;
        return e * d + k;
    };
    auto la = [&](int sl, int j) { int nv; const int64_t i = ia(sl, j, nv); return dz_fetch4<FUSED>(g, i, nv, total); };
    auto fa = [&](int sl, int j, const DzRaw& w) { int nv; const int64_t i = ia(sl, j, nv); return dz_value4<FUSED>(w, i, nv, total); };
    auto lb = [&](int sl, int j) { bool ok; const int64_t i = ib(sl, j, ok); return load4_clamped(ent, i, ok); };
    auto fb = [&](int sl, int j, const float4& v) { bool ok; ib(sl, j, ok); return keep_if(v, ok); };
    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    gemm128x_core<false, true>((int)((ee - es + HK2 - 1) / HK2), la, fa, lb, fb, sA, sB, wr, wc, lcol, lk4, acc);
    tile_out(acc, emode, part, dx, b0, B, k0, d, wr, wc, lcol, lk4, /* accumulate = */ false);
}

// g_ent[e, k] += sum_{b in this split} dz[b, e] x[b, k] ;  g_bias[e] += sum_b dz[b, e]    grid: (d tiles, E tiles, B splits);
// both operands are k-major in memory (k = the batch row): no transposition on the way into LDS
template <bool FUSED>
__global__ __launch_bounds__(256, HEAD_BWD_OCC) void k_head_dent128(DzSrc g, const float* __restrict__ x, int64_t B, int64_t E, int d,
                                                         int64_t b_per_split, float* __restrict__ g_ent, float* __restrict__ g_bias,
                                                         int emode, float* __restrict__ part, float* __restrict__ bpart) {
    __shared__ __attribute__((aligned(16))) float sA[2][HK2][HLD3], sB[2][HK2][HLD3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lcol = lane & 15, lk4 = lane >> 4;
    const int64_t k0 = (int64_t)blockIdx.x * HT2, e0 = (int64_t)blockIdx.y * HT2;
    const int64_t bs = (int64_t)blockIdx.z * b_per_split, be = min(B, bs + b_per_split);
    const int wr = wave >> 1, wc = wave & 1;
    const int64_t total = B * E;
    auto ia = [&](int sl, int j, int& nv) {
        const int idx = threadIdx.x + 256 * j;
        const int64_t b = bs + (int64_t)sl * HK2 + (idx >> 5), e = e0 + (idx & 31) * 4;
        const int64_t n = b < be ? E - e : 0;
        nv = n > 4 ? 4 : (int)n;
        return b * g.ld + e;
    };
    auto ib = [&](int sl, int j, bool& ok) {
        const int idx = threadIdx.x + 256 * j;
        const int64_t b = bs + (int64_t)sl * HK2 + (idx >> 5), k = k0 + (idx & 31) * 4;
        ok = b < be && k < d;
        return b * d + k;
    };
    auto la = [&](int sl, int j) { int nv; const int64_t i = ia(sl, j, nv); return dz_fetch4<FUSED>(g, i, nv, total); };
    auto fa = [&](int sl, int j, const DzRaw& w) { int nv; const int64_t i = ia(sl, j, nv); return dz_value4<FUSED>(w, i, nv, total); };
    auto lb = [&](int sl, int j) { bool ok; const int64_t i = ib(sl, j, ok); return load4_clamped(x, i, ok); };
    auto fb = [&](int sl, int j, const float4& v) { bool ok; ib(sl, j, ok); return keep_if(v, ok); };
    float colsum = 0.f;   // threads 0 .. 127 of the k-tile-0 workgroups: bias gradient of entity e0 + threadIdx.x
    const bool do_bias = g_bias != nullptr && blockIdx.x == 0 && threadIdx.x < HT2;
    auto hook = [&](float (*sa)[HLD3]) {
        if (do_bias) {
#pragma unroll
            for (int c = 0; c < HK2; ++c) colsum += sa[c][threadIdx.x];
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    gemm128x_core<true, true>((int)((be - bs + HK2 - 1) / HK2), la, fa, lb, fb, sA, sB, wr, wc, lcol, lk4, acc, hook);
    tile_out(acc, emode, part, g_ent, e0, E, k0, d, wr, wc, lcol, lk4, /* accumulate = */ true);
    if (do_bias) {
        if (emode == kOutParts) bpart[((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * HT2 + threadIdx.x] = colsum;
        else if (e0 + threadIdx.x < E && colsum != 0.f) {
            if (emode == kOutDirect) g_bias[e0 + threadIdx.x] += colsum;
            else unsafeAtomicAdd(g_bias + e0 + threadIdx.x, colsum);
        }
    }
}

// out[m, n] (+)= the sum over the splits of the partial tiles the workgroups of a split-K product left (kOutParts), in a FIXED
// order: L lanes per float4 of the tile space, lane j adds splits j, j + L, ... in ascending order, then the L lane sums are added
// by a butterfly (L = 1: plain split order).  bpart: the bias-gradient partials of the entity-gradient product (column tile 0).
template <int L>
__global__ __launch_bounds__(256) void k_head_sum_parts(const float* __restrict__ part, int splits, int MT, int NT, int64_t M, int N,
                                                        float* __restrict__ out, int accumulate, const float* __restrict__ bpart,
                                                        float* __restrict__ g_bias) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) / L;
    const int lane = threadIdx.x % L;
    const int c4 = (int)(i & 31), r = (int)((i >> 5) & 127);
    const int64_t t = i >> 12;
    if (t >= (int64_t)MT * NT) return;
    const int nt = (int)(t % NT), mt = (int)(t / NT);
    const int64_t m = (int64_t)mt * HT2 + r;
    const int n = nt * HT2 + 4 * c4;
    if (m >= M) return;
    if (n < N) {   // (uniform over the L lanes of a group)
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int z = lane; z < splits; z += L) {
            const float4 v = *reinterpret_cast<const float4*>(part + ((((int64_t)z * MT + mt) * NT + nt) << 14) + r * HT2 + 4 * c4);
            sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        }
#pragma unroll
        for (int o = L / 2; o > 0; o >>= 1) {
            sum.x += __shfl_xor(sum.x, o, 64); sum.y += __shfl_xor(sum.y, o, 64);
            sum.z += __shfl_xor(sum.z, o, 64); sum.w += __shfl_xor(sum.w, o, 64);
        }
        if (lane == 0) {
            float4* const o4 = reinterpret_cast<float4*>(out + m * N + n);
            if (accumulate) { const float4 w = *o4; sum.x += w.x; sum.y += w.y; sum.z += w.z; sum.w += w.w; }
            *o4 = sum;
        }
    }
    if (bpart != nullptr && nt == 0 && c4 == 0) {
        float bs = 0.f;
        for (int z = lane; z < splits; z += L) bs += bpart[((int64_t)z * MT + mt) * HT2 + r];
#pragma unroll
        for (int o = L / 2; o > 0; o >>= 1) bs += __shfl_xor(bs, o, 64);
        if (lane == 0) g_bias[m] += bs;
    }
}

static void launch_head_sum_parts(const float* part, int64_t splits, int64_t MT, int64_t NT, int64_t M, int N, float* out, int accumulate,
                                  const float* bpart, float* g_bias, hipStream_t s) {
    if (splits >= 16)
        hipLaunchKernelGGL(k_head_sum_parts<8>, dim3((unsigned)(MT * NT * 16 * 8)), dim3(256), 0, s, part, (int)splits, (int)MT, (int)NT, M, N,
                           out, accumulate, bpart, g_bias);
    else
        hipLaunchKernelGGL(k_head_sum_parts<1>, dim3((unsigned)(MT * NT * 16)), dim3(256), 0, s, part, (int)splits, (int)MT, (int)NT, M, N, out,
                           accumulate, bpart, g_bias);
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
    a.xcd_runs = switch_value("HEAD_XCD") != 0;   // KGE_HEAD_XCD=0: plain tile numbering (A/B)
    a.x = x; a.ent = ent; a.bias = bias; a.B = B; a.E = E; a.d = d; a.preds = preds;
    if (bf16) {
        // (the same accumulation order over k in both tile shapes: 16 k per MFMA, slabs in order -- identical logits)
        if (head_use_128(a))
            hipLaunchKernelGGL(k_head_gemm128_bf16, head_grid(a, HT2), dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL(k_head_gemm_bf16, head_grid(a, HT), dim3(256), 0, s, a);
        return check_launch("k_head_gemm_bf16");
    }
    launch_head_gemm<0>(a, s);
    return check_launch("k_head_gemm<0>");
}

// parts (may be NULL): head_parts_bytes() of workspace for the split-K partial tiles of the wide products
static size_t head_parts_bytes() { return (size_t)kHeadPartSlots * HT2 * HT2 * sizeof(float) + (size_t)kHeadPartSlots * HT2 * sizeof(float); }

static int head_backward_gemms(const DzSrc& g, const float* x, int64_t B, int d, const float* ent, int64_t E, float* dx,
                               float* g_ent, float* g_bias, float* parts, hipStream_t s) {
    // the 128-wide forms pay once the batch fills their tiles (the same switch as the forward: KGE_HEAD_TILE = 0 / 1 forces)
    const int force = switch_value("HEAD_TILE");
    const bool aligned = d % 4 == 0 && (((uintptr_t)x | (uintptr_t)ent | (uintptr_t)dx | (uintptr_t)g_ent) & 15) == 0;
    // with a partials buffer the wide forms win from the reference's own batch (B = 128: dX 40 -> 12 us, g_ent 31 -> 17 us); without
    // one their split-K tiles meet in float atomics, which only pays from B ~ 2 560
    const int min_b = switch_value("HEAD_WIDE_B") > 0 ? switch_value("HEAD_WIDE_B") : (parts ? 1 : 2560);
    // (the autograd form's 16-byte fetches are clamped to the last in-bounds position of the [B, E] arrays: they need four elements)
    const bool wide = aligned && (g.dz != nullptr || B * E >= 4) && (force >= 0 ? force == 1 : B >= min_b);
    const int T = wide ? HT2 : HT, SK = wide ? HK2 : HK;
    // split-K: ~8 small-tile workgroups per CU; wide: at most three per CU, all resident in ONE round (rounding the split count up
    // left a second, mostly empty round), never more than the partials buffer has slots, and no split shorter than kMinK of K
    // (every split pays a prologue, an epilogue and a partial tile that somebody has to add).  Every split a multiple of the slab.
    const int min_k = switch_value("HEAD_MINK") > 0 ? switch_value("HEAD_MINK") : 128;
    auto split = [&](int64_t tiles, int64_t K, int64_t* per) {
        int64_t splits = wide ? kHeadPartSlots / tiles : (2048 + tiles - 1) / tiles;
        const int64_t max_splits = wide ? (K + min_k - 1) / min_k : (K + SK - 1) / SK;
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
        *per = ((K + splits - 1) / splits + SK - 1) / SK * SK;
        return (K + *per - 1) / *per;
    };
    float* const bparts = parts ? parts + kHeadPartSlots * HT2 * HT2 : nullptr;
    if (dx) {
        int64_t per;
        const int64_t MT = (B + T - 1) / T, NT = (d + T - 1) / T;
        const int64_t splits = split(MT * NT, E, &per);
        const int emode = !wide ? kOutAtomic : (splits == 1 ? kOutDirect : (parts ? kOutParts : kOutAtomic));
        if (emode == kOutAtomic) {
            hipError_t e = hipMemsetAsync(dx, 0, (size_t)B * d * sizeof(float), s);
            if (e != hipSuccess) { set_error("kge_head: memset: %s", hipGetErrorString(e)); return -2; }
        }
        const dim3 grid((unsigned)NT, (unsigned)MT, (unsigned)splits);
        if (wide && g.dz) hipLaunchKernelGGL(k_head_dx128<true>, grid, dim3(256), 0, s, g, ent, B, E, d, per, dx, emode, parts);
        else if (wide) hipLaunchKernelGGL(k_head_dx128<false>, grid, dim3(256), 0, s, g, ent, B, E, d, per, dx, emode, parts);
        else hipLaunchKernelGGL(k_head_dx, grid, dim3(256), 0, s, g, ent, B, E, d, per, dx);
        if (emode == kOutParts) launch_head_sum_parts(parts, splits, MT, NT, B, d, dx, 0, nullptr, nullptr, s);
    }
    if (g_ent) {
        int64_t per;
        const int64_t MT = (E + T - 1) / T, NT = (d + T - 1) / T;
        const int64_t splits = split(MT * NT, B, &per);
        const int emode = !wide ? kOutAtomic : (splits == 1 ? kOutDirect : (parts ? kOutParts : kOutAtomic));
        const dim3 grid((unsigned)NT, (unsigned)MT, (unsigned)splits);
        if (wide && g.dz) hipLaunchKernelGGL(k_head_dent128<true>, grid, dim3(256), 0, s, g, x, B, E, d, per, g_ent, g_bias, emode, parts, bparts);
        else if (wide) hipLaunchKernelGGL(k_head_dent128<false>, grid, dim3(256), 0, s, g, x, B, E, d, per, g_ent, g_bias, emode, parts, bparts);
        else hipLaunchKernelGGL(k_head_dent, grid, dim3(256), 0, s, g, x, B, E, d, per, g_ent, g_bias);
        if (emode == kOutParts) launch_head_sum_parts(parts, splits, MT, NT, E, d, g_ent, 1, g_bias ? bparts : nullptr, g_bias, s);
    }
    return check_launch("k_head_dx / k_head_dent");
}

size_t head_backward_workspace_bytes() { return head_parts_bytes(); }

int launch_head_backward(const float* x, int64_t B, int d, const float* ent, int64_t E, const float* preds,
                         const float* dpreds, float* dx, float* g_ent, float* g_bias, void* ws, size_t ws_bytes, hipStream_t s) {
    if (head_check("kge_head_1n_backward", x, ent, B, E, d)) return -1;
    if (!preds || !dpreds) { set_error("kge_head_1n_backward: preds / dpreds are null"); return -1; }
    if (ws && (ws_bytes < head_parts_bytes() || ((uintptr_t)ws & 15))) {
        set_error("kge_head_1n_backward: workspace too small or misaligned (need kge_head_1n_backward_workspace_bytes, 16-byte aligned)");
        return -1;
    }
    DzSrc g{nullptr, dpreds, preds, E};
    return head_backward_gemms(g, x, B, d, ent, E, dx, g_ent, g_bias, (float*)ws, s);
}

static int64_t head_ldz(int64_t E) { return (E + 3) / 4 * 4; }

size_t head_bce_workspace_bytes(int64_t B, int64_t E, int64_t n_pos) {
    return ((size_t)B * head_ldz(E) * sizeof(float) + 255) / 256 * 256 + ((size_t)n_pos * sizeof(int32_t) + 255) / 256 * 256 + head_parts_bytes();
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
    int32_t* row_of = (int32_t*)((char*)ws + ((size_t)B * head_ldz(E) * sizeof(float) + 255) / 256 * 256);
    // Criterion.multi_class_bce (utils/criterion.py:42-45): y <- y (1 - ls) + 1/E when label smoothing is given
    const bool smooth = label_smoothing >= 0.f;
    const float y0 = smooth ? 1.0f / (float)E : 0.f;
    const float y1 = smooth ? (1.0f - label_smoothing) + 1.0f / (float)E : 1.f;
    HeadArgs a{};
    a.xcd_runs = switch_value("HEAD_XCD") != 0;   // KGE_HEAD_XCD=0: plain tile numbering (A/B)
    a.x = x; a.ent = ent; a.bias = bias; a.B = B; a.E = E; a.d = d; a.dz = dz; a.ldz = head_ldz(E); a.loss = loss;
    a.y0 = y0; a.inv_count = 1.0f / ((float)B * (float)E);
    launch_head_gemm<1>(a, s);
    if (n_pos > 0) {
        hipLaunchKernelGGL(k_head_rows_of, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, lab_off, B, row_of);
        int64_t blocks = (n_pos + 7) / 8;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(k_head_bce_pos, dim3((unsigned)blocks), dim3(256), 0, s, a, lab_off, lab_ids, row_of, n_pos, y1 - y0);
    }
    DzSrc g{dz, nullptr, nullptr, a.ldz};
    float* const parts = (float*)((char*)row_of + ((size_t)n_pos * sizeof(int32_t) + 255) / 256 * 256);
    return head_backward_gemms(g, x, B, d, ent, E, dx, g_ent, g_bias, parts, s);
}

}  // namespace kge
