// kge_transr_rows.hip -- TransR (pykg2vec/models/pairwise.py:367-470), the pairwise hinge step of LARGE batches in two launches.
//
// kge_transr.hip's tile kernels (one 32-triple tile per workgroup, forward / hinge / backward as three launches, the forward run
// twice, M_r operands straight from global memory, d_e x d_r float atomics per tile for the matrix gradient) stop at 14 TFLOP/s
// on the FB15k shape at B = 32 768.  A negative keeps its positive's relation (every sampler of the reference corrupts heads or
// tails: the caller says so by passing the SAME buffer for nr and pr, as for RESCAL), so PAIRS are grouped by relation and:
//
//   k_transr_rows<NB>   a workgroup takes 32 pairs of one relation (two consecutive 16-pair tiles), a wave 8 pairs = one 16-row
//                       block with a pair's positive and negative in adjacent rows.  The normalised rows H^ and T^ are the A
//                       operands (registers) of ONE pass over M_r (resident in LDS, v_mfma_f32_16x16x4_f32, two accumulator sets), the
//                       TransE tail (three normalisations, distance, hinge, and back) is register arithmetic in accumulator
//                       layout, GA = dL/d(h^ M) and GC = dL/d(t^ M) turn into A operands through a wave-private LDS transpose,
//                       a second pass over M_r^T gives GH^ / GT^, which go back through the entity normalisation and leave as
//                       row-wise float atomics (16 consecutive floats per lane group; the uncorrupted side of a pair: one
//                       merged atomic).  GA / GC rows and the rows' inverse norms are left in the workspace.
//   k_transr_g<NBI,NBJ> the relation-matrix gradient G_r = H^^T GA + T^^T GC as a GEMM over gathered rows (K = 4 rows per pair of
//                       the relation), in the style of kge_dense.hip's k_rescal_g: a run of 8 tiles per workgroup, one half of the
//                       output columns, ids resolved two slabs ahead and rows one slab ahead; plain read-modify-write where the
//                       relation has a single run, float atomics otherwise.
//
// MFMA operand maps: kge_mfma_blocks.h.  fp32 throughout (exact products, fp32 accumulation).
#include "kge_internal.h"
#include "kge_relgroup.h"
#include "kge_mfma_blocks.h"
#define KGE_TS_UNIT transr
#include "kge_ts_debug.h"   // (slot 0 = k_transr_g2; no-ops in the product build)

namespace kge {

constexpr int kTrPairTile = 16;      // pairs per grouping tile (a workgroup of k_transr_rows takes two)
constexpr int kTrGRun = 8;           // tiles per run of k_transr_g

struct TransRRowsArgs {
    const float* ent; const float* rel; const float* mat;
    float* g_ent; float* g_rel; float* g_mat;
    const int64_t* ph; const int64_t* pt; const int64_t* nh; const int64_t* nt;
    const int* offsets; const int* tile_off; const int* tile_rel; const int* perm;
    int R, de, dr, l1;
    float margin;
    float* loss;
    float* invs;   // [4 n]  grouped pair g: 1 / max(|row|, eps) of (pos h, neg h, pos t, neg t); 0 = the pair has no gradient
    int tiles;     // tiles of the grouping (the grid is rounded up to a multiple of the kernel's tile stride)
    float* gws;    // [4 n][dr]  grouped pair g: GA of the positive, GA of the negative, GC of the positive, GC of the negative
};

__device__ __forceinline__ float grp16_sum(float v) { return gsum<16>(v); }   // DPP steps inside a 16-lane row: no LDS crossbar

// k_transr_rows: M_r is RESIDENT in LDS for the workgroup's lifetime (d_e x d_r floats: 40 KB at 100 / 100, two workgroups per CU), so
// the two passes run without slab staging, barriers or global loads in their loops.  M_r is kept in natural [k][c] layout with
// pitch 16 NB + 4 (= 4 x odd mod 32):
//   pass 1 (X M, K = d_e):   lane (l, lk) takes k = 16 kb + 4 lk + kk (its A elements are 16 consecutive bytes of the entity row); the
//                            B operands of FOUR column blocks are one 16-byte read of row k -- block cb, lane l is column
//                            BlkMap<NB>::at(cb, l), a permutation the tail and the transpose carry along;
//   pass 2 (G M^T, K = d_r): output column c = 16 cb + l (natural: the entity-gradient atomics want 16 consecutive floats per lane
//                            group) reads M[c][16 kb + 4 lk .. + 3] as one 16-byte read = the operands of its four kk steps; the A
//                            operands come out of the transpose buffer the same way.
// Rows k >= d_e (pass 1) and c >= d_e (pass 2) are redirected to a zero row.  GA / GC go to the workspace from A layout (64-byte
// segments per row).  VEC: d_e and d_r multiples of 4 and 16-byte aligned tables (16-byte global accesses).
// Every global load sits on a clamped, always valid address and is masked by a select afterwards: a load inside a divergent branch
// costs a branch and an s_waitcnt vmcnt(0) at the join (the first build of this kernel had 1 194 loads in 2 623 branches, 250 KB of code).
template <int NB, bool VEC, bool L1>
__global__ __launch_bounds__(256, 2) void k_transr_rows(TransRRowsArgs A) {
    constexpr int DP = 16 * NB, PITCH = DP + 4, NK = 4 * NB;
    using BM = BlkMap<NB>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int rel, tin;
    const int tile = strided_tile<2>(A.tiles);
    if (tile >= A.tiles || !locate_tile(A.tile_off, A.tile_rel, A.R, tile, rel, tin)) return;
    if (tin & 1) return;                       // (the workgroup of an even tile takes the odd one after it as well)
    const int de = A.de, dr = A.dr;
    float* sM = smem;                                   // [de + 1][PITCH]   M_r, row de = zeros
    float* sXall = sM + (de + 1) * PITCH;               // [4][16][PITCH]    the waves' transpose buffers
    float* sInv = sXall + 4 * 16 * PITCH;               // [4][2][16]        per wave, side (h, t), row: 1 / max(|x|, eps)
    float* sFlg = sInv + 128;                           // [4][2][16]        |x| > eps
    float* sC = sFlg + 128;                             // [4][16]           hinge coefficient of the row's pair
    float* sGR = sC + 64;                               // [DP]              the workgroup's gradient with respect to r^
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l = lane & 15, lk = lane >> 4;
    float* sX = sXall + wave * 16 * PITCH;
    const float* __restrict__ M = A.mat + (int64_t)rel * de * dr;
    const int g_lo = A.offsets[rel] + tin * kTrPairTile, g_hi = min(A.offsets[rel + 1], g_lo + 2 * kTrPairTile);
    const bool wave_live = g_lo + 8 * wave < g_hi;     // wave-uniform
    // A-operand row of this lane: row l of the wave = pair 8 wave + l / 2, side l & 1 (0 positive, 1 negative)
    const int a_g = g_lo + 8 * wave + (l >> 1);
    const bool a_on = a_g < g_hi;
    const int a_pair = A.perm[min(a_g, g_hi - 1)];
    // accumulator rows of this lane: 4 lk + q = pairs 2 lk (q = 0 positive, 1 negative) and 2 lk + 1 (q = 2, 3)
    int c_g[2], c_pair[2];
    bool c_on[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        c_g[j] = g_lo + 8 * wave + 2 * lk + j;
        c_on[j] = c_g[j] < g_hi;
        c_pair[j] = A.perm[min(c_g[j], g_hi - 1)];
    }
    // ---- M_r -> LDS (runs under the pair -> id -> row chain)
    if constexpr (VEC) {
        const int c4 = 4 * (lane & 31);
        const int cl = min(c4, dr - 4);
        for (int r0 = 2 * wave + (lane >> 5); r0 < de; r0 += 32) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(M + (int64_t)min(r0 + 8 * u, de - 1) * dr + cl);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = r0 + 8 * u;
                if (r < de && c4 < PITCH) *reinterpret_cast<float4*>(sM + r * PITCH + c4) = c4 < dr ? v[u] : float4{0.f, 0.f, 0.f, 0.f};
            }
        }
    } else {
        for (int r0 = wave; r0 < de; r0 += 16) {
            float v[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int w = 0; w < 3; ++w) v[u][w] = M[(int64_t)min(r0 + 4 * u, de - 1) * dr + min(lane + 64 * w, dr - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int w = 0; w < 3; ++w) {
                    const int r = r0 + 4 * u, c = lane + 64 * w;
                    if (r < de && c < PITCH) sM[r * PITCH + c] = c < dr ? v[u][w] : 0.f;
                }
        }
    }
    if (threadIdx.x < PITCH) sM[de * PITCH + threadIdx.x] = 0.f;
    if (threadIdx.x < DP) sGR[threadIdx.x] = 0.f;
    const int a_h = (int)((l & 1) ? A.nh[a_pair] : A.ph[a_pair]), a_t = (int)((l & 1) ? A.nt[a_pair] : A.pt[a_pair]);
    int c_h[4], c_t[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        c_h[2 * j] = (int)A.ph[c_pair[j]]; c_h[2 * j + 1] = (int)A.nh[c_pair[j]];
        c_t[2 * j] = (int)A.pt[c_pair[j]]; c_t[2 * j + 1] = (int)A.nt[c_pair[j]];
    }

    // ---- 0. rows in A layout (element 4 kb + kk of the array = k index 16 kb + 4 lk + kk), normalised
    float a0[NK], a1[NK];
    {
        const float* __restrict__ hr = A.ent + (int64_t)a_h * de;
        const float* __restrict__ tr = A.ent + (int64_t)a_t * de;
        if constexpr (VEC) {
            unroll_seq([&](auto kbc) __attribute__((always_inline)) {
                constexpr int kb = decltype(kbc)::value;
                const int k0 = 16 * kb + 4 * lk, kc = min(k0, de - 4);
                const float4 vh = *reinterpret_cast<const float4*>(hr + kc), vt = *reinterpret_cast<const float4*>(tr + kc);
                a0[4 * kb] = vh.x; a0[4 * kb + 1] = vh.y; a0[4 * kb + 2] = vh.z; a0[4 * kb + 3] = vh.w;
                a1[4 * kb] = vt.x; a1[4 * kb + 1] = vt.y; a1[4 * kb + 2] = vt.z; a1[4 * kb + 3] = vt.w;
            }, std::make_integer_sequence<int, NB>{});
        } else {
            unroll_seq([&](auto ksc) __attribute__((always_inline)) {
                constexpr int ks = decltype(ksc)::value;
                const int k = min(16 * (ks >> 2) + 4 * lk + (ks & 3), de - 1);
                a0[ks] = hr[k]; a1[ks] = tr[k];
            }, std::make_integer_sequence<int, NK>{});
        }
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const bool in = a_on && 16 * (ks >> 2) + 4 * lk + (ks & 3) < de;
            a0[ks] = in ? a0[ks] : 0.f; a1[ks] = in ? a1[ks] : 0.f;
            s0 = fmaf(a0[ks], a0[ks], s0); s1 = fmaf(a1[ks], a1[ks], s1);
        }
        s0 += swz_xor16(s0); s0 += __shfl_xor(s0, 32, 64);
        s1 += swz_xor16(s1); s1 += __shfl_xor(s1, 32, 64);
        const float n0 = sqrtf(s0), n1 = sqrtf(s1);
        const float i0 = 1.0f / fmaxf(n0, kEpsNormalize), i1 = 1.0f / fmaxf(n1, kEpsNormalize);
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) { a0[ks] *= i0; a1[ks] *= i1; }
        if (lk == 0) {
            sInv[(wave * 2 + 0) * 16 + l] = i0; sInv[(wave * 2 + 1) * 16 + l] = i1;
            sFlg[(wave * 2 + 0) * 16 + l] = n0 > kEpsNormalize ? 1.f : 0.f; sFlg[(wave * 2 + 1) * 16 + l] = n1 > kEpsNormalize ? 1.f : 0.f;
        }
    }
    // the relation row in accumulator layout (block cb, lane l = column colp[cb]); used after pass 1, requested now
    int colp[NB];
    float rr[NB];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) { colp[cb] = BM::at(cb, l); rr[cb] = A.rel[(int64_t)rel * dr + min(colp[cb], dr - 1)]; }
    __syncthreads();   // M_r, the zero row, sGR, sInv / sFlg are in LDS

    // ---- 1. HP = H^ M, TP = T^ M
    f32x4v acc0[NB], acc1[NB];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) { acc0[cb] = f32x4v{0.f, 0.f, 0.f, 0.f}; acc1[cb] = f32x4v{0.f, 0.f, 0.f, 0.f}; }
    if (wave_live) {
        unroll_seq([&](auto kbc) __attribute__((always_inline)) {
            constexpr int kb = decltype(kbc)::value;   // (no skipping of slabs past d_e: the compiler threads such a chain of tests
            float b[2][NB];                            //  into 1 + 2 + ... + NB copies of the slab body)
            auto rowp = [&](int kk) __attribute__((always_inline)) { const int k = 16 * kb + 4 * lk + kk; return sM + (k < de ? k : de) * PITCH; };
            read_blocks<NB>(rowp(0), l, b[0]);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                if (kk + 1 < 4) read_blocks<NB>(rowp(kk + 1), l, b[(kk + 1) & 1]);
                KGE_KEEP_READS_AHEAD();
#pragma unroll
                for (int cb = 0; cb < NB; ++cb) {
                    acc0[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[4 * kb + kk], b[kk & 1][cb], acc0[cb], 0, 0, 0);
                    acc1[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[4 * kb + kk], b[kk & 1][cb], acc1[cb], 0, 0, 0);
                }
            }
        }, std::make_integer_sequence<int, NB>{});
    }

    // ---- 2. the TransE tail per row (pairwise.py:459-470) in accumulator layout: row 4 lk + q, column colp[cb]
    float invr, flgr, ib, flgb;
    {
        float ss = 0.f;
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) { rr[cb] = colp[cb] < dr ? rr[cb] : 0.f; ss = fmaf(rr[cb], rr[cb], ss); }
        const float n1 = sqrtf(grp16_sum(ss));
        invr = 1.0f / fmaxf(n1, kEpsNormalize);
        flgr = n1 > kEpsNormalize ? 1.f : 0.f;
        ss = 0.f;
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) { rr[cb] *= invr; ss = fmaf(rr[cb], rr[cb], ss); }   // r^ (embed); forward normalises once more
        const float nb = sqrtf(grp16_sum(ss));
        ib = 1.0f / fmaxf(nb, kEpsNormalize);
        flgb = nb > kEpsNormalize ? 1.f : 0.f;
    }
    constexpr bool l1 = L1;   // (a template parameter: as a kernel argument every l1 ? : below became a scalar branch)
    float ia[4], ic[4], sc[4];
    bool fa[4], fc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float sa = 0.f, s2 = 0.f;
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) { sa = fmaf(acc0[cb][q], acc0[cb][q], sa); s2 = fmaf(acc1[cb][q], acc1[cb][q], s2); }
        const float na = sqrtf(grp16_sum(sa)), nc = sqrtf(grp16_sum(s2));
        ia[q] = 1.0f / fmaxf(na, kEpsNormalize); ic[q] = 1.0f / fmaxf(nc, kEpsNormalize);
        fa[q] = na > kEpsNormalize; fc[q] = nc > kEpsNormalize;
        float p = 0.f;
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
            const float u = acc0[cb][q] * ia[q] + rr[cb] * ib - acc1[cb][q] * ic[q];
            p += l1 ? fabsf(u) : u * u;
        }
        p = grp16_sum(p);
        sc[q] = l1 ? p : sqrtf(p);
    }
    float c[2], hl = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float v = sc[2 * j] + A.margin - sc[2 * j + 1];
        c[j] = c_on[j] ? (v > 0.f ? 1.f : (v == 0.f ? 0.5f : 0.f)) : 0.f;
        hl += c_on[j] ? fmaxf(v, 0.f) : 0.f;
    }
    {
        const float tot = wave_sum(l == 0 ? hl : 0.f);
        if (lane == 0 && tot != 0.f) unsafeAtomicAdd(A.loss + (blockIdx.x % kLossSlots) * kLossStride, tot);
    }
    if (l == 0) {      // what k_transr_g needs of a row besides GA / GC: its inverse norm, 0 where the pair has no gradient
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = q >> 1;
            sC[wave * 16 + 4 * lk + q] = c[j];
            if (c_on[j]) {
                A.invs[4 * (int64_t)c_g[j] + (q & 1)] = c[j] != 0.f ? sInv[(wave * 2 + 0) * 16 + 4 * lk + q] : 0.f;
                A.invs[4 * (int64_t)c_g[j] + 2 + (q & 1)] = c[j] != 0.f ? sInv[(wave * 2 + 1) * 16 + 4 * lk + q] : 0.f;
            }
        }
    }
    if (!__syncthreads_or(c[0] != 0.f || c[1] != 0.f)) return;   // every pair of the workgroup inside the margin: no gradient
    float gb[NB];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) gb[cb] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float ds = (q & 1) ? -c[q >> 1] : c[q >> 1];
        const float invs = (!l1 && sc[q] > 0.f) ? ds / sc[q] : 0.f;
        float da = 0.f, db = 0.f, dc = 0.f;
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
            const float u = acc0[cb][q] * ia[q] + rr[cb] * ib - acc1[cb][q] * ic[q];
            const float g = l1 ? (u > 0.f ? ds : (u < 0.f ? -ds : 0.f)) : u * invs;
            da = fmaf(acc0[cb][q], g, da); db = fmaf(rr[cb], g, db); dc = fmaf(acc1[cb][q], g, dc);
        }
        da = grp16_sum(da) * ia[q]; db = grp16_sum(db) * ib; dc = grp16_sum(dc) * ic[q];
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
            const float av = acc0[cb][q], cv = acc1[cb][q];
            const float u = av * ia[q] + rr[cb] * ib - cv * ic[q];
            const float g = l1 ? (u > 0.f ? ds : (u < 0.f ? -ds : 0.f)) : u * invs;
            acc0[cb][q] = fa[q] ? (g - (av * ia[q]) * da) * ia[q] : g * ia[q];
            acc1[cb][q] = -(fc[q] ? (g - (cv * ic[q]) * dc) * ic[q] : g * ic[q]);
            gb[cb] += flgb != 0.f ? (g - (rr[cb] * ib) * db) * ib : g * ib;
        }
    }
    // the workgroup's gradient with respect to r^: rows of the lane -> rows of the wave -> LDS
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
        float v = gb[cb];
        v += swz_xor16(v); v += __shfl_xor(v, 32, 64);
        if (lk == 0 && v != 0.f) atomicAdd(&sGR[colp[cb]], v);
    }

    // ---- 3. GA / GC through the wave's transpose buffer into A layout (k index 16 kb + 4 lk + kk), and from there to the workspace
    const bool row_grad = a_on && sC[wave * 16 + l] != 0.f;
    auto to_a_layout = [&](f32x4v (&acc)[NB], float (&a)[NK], int x0) __attribute__((always_inline)) {
#pragma unroll
        for (int cb = 0; cb < NB; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) sX[(4 * lk + q) * PITCH + colp[cb]] = acc[cb][q];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float* __restrict__ out = A.gws + (4 * (int64_t)a_g + x0 + (l & 1)) * dr;
        unroll_seq([&](auto kbc) __attribute__((always_inline)) {
            constexpr int kb = decltype(kbc)::value;
            const float4 v = *reinterpret_cast<const float4*>(sX + l * PITCH + 16 * kb + 4 * lk);
            a[4 * kb] = v.x; a[4 * kb + 1] = v.y; a[4 * kb + 2] = v.z; a[4 * kb + 3] = v.w;
            const int k0 = 16 * kb + 4 * lk;
            if (row_grad) {
                if constexpr (VEC) { if (k0 < dr) *reinterpret_cast<float4*>(out + k0) = v; }
                else {
                    if (k0 < dr) out[k0] = v.x;
                    if (k0 + 1 < dr) out[k0 + 1] = v.y;
                    if (k0 + 2 < dr) out[k0 + 2] = v.z;
                    if (k0 + 3 < dr) out[k0 + 3] = v.w;
                }
            }
        }, std::make_integer_sequence<int, NB>{});
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    to_a_layout(acc0, a0, 0);
    to_a_layout(acc1, a1, 2);

    // ---- 4. GH^ = GA M^T, GT^ = GC M^T; accumulator block cb, lane l = column 16 cb + l
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) { acc0[cb] = f32x4v{0.f, 0.f, 0.f, 0.f}; acc1[cb] = f32x4v{0.f, 0.f, 0.f, 0.f}; }
    if (wave_live) {
        const float* rowc[NB];
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) { const int cc = 16 * cb + l; rowc[cb] = sM + (cc < de ? cc : de) * PITCH + 4 * lk; }
        unroll_seq([&](auto kbc) __attribute__((always_inline)) {
            constexpr int kb = decltype(kbc)::value;
            {
                float4 b[NB];
#pragma unroll
                for (int cb = 0; cb < NB; ++cb) b[cb] = *reinterpret_cast<const float4*>(rowc[cb] + 16 * kb);
#pragma unroll
                for (int cb = 0; cb < NB; ++cb) {
                    acc0[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[4 * kb], b[cb].x, acc0[cb], 0, 0, 0);
                    acc1[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[4 * kb], b[cb].x, acc1[cb], 0, 0, 0);
                }
#pragma unroll
                for (int cb = 0; cb < NB; ++cb) {
                    acc0[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[4 * kb + 1], b[cb].y, acc0[cb], 0, 0, 0);
                    acc1[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[4 * kb + 1], b[cb].y, acc1[cb], 0, 0, 0);
                }
#pragma unroll
                for (int cb = 0; cb < NB; ++cb) {
                    acc0[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[4 * kb + 2], b[cb].z, acc0[cb], 0, 0, 0);
                    acc1[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[4 * kb + 2], b[cb].z, acc1[cb], 0, 0, 0);
                }
#pragma unroll
                for (int cb = 0; cb < NB; ++cb) {
                    acc0[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[4 * kb + 3], b[cb].w, acc0[cb], 0, 0, 0);
                    acc1[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[4 * kb + 3], b[cb].w, acc1[cb], 0, 0, 0);
                }
            }
        }, std::make_integer_sequence<int, NB>{});
    }

    // ---- 5. back through the entity normalisation, row-wise atomics (16 consecutive floats per lane group)
    auto scatter = [&](f32x4v (&acc)[NB], const int (&ids)[4], int side) __attribute__((always_inline)) {
        float x[4][NB];
#pragma unroll
        for (int q = 0; q < 4; ++q)     // (all 4 NB row elements requested before the first is used)
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) x[q][cb] = A.ent[(int64_t)ids[q] * de + min(16 * cb + l, de - 1)];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float inv = sInv[(wave * 2 + side) * 16 + 4 * lk + q];
            const bool f = sFlg[(wave * 2 + side) * 16 + 4 * lk + q] != 0.f;
            float dot = 0.f;
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) {
                x[q][cb] = 16 * cb + l < de ? x[q][cb] * inv : 0.f;
                dot = fmaf(x[q][cb], acc[cb][q], dot);
            }
            dot = grp16_sum(dot);
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) acc[cb][q] = f ? (acc[cb][q] - x[q][cb] * dot) * inv : acc[cb][q] * inv;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (c[j] == 0.f) continue;
            const bool same = ids[2 * j] == ids[2 * j + 1];
            float* __restrict__ op = A.g_ent + (int64_t)ids[2 * j] * de + l;
            float* __restrict__ on = A.g_ent + (int64_t)ids[2 * j + 1] * de + l;
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) {
                if (16 * cb + l < de) {
                    const float xp = acc[cb][2 * j], xn = acc[cb][2 * j + 1];
                    unsafeAtomicAdd(op + 16 * cb, same ? xp + xn : xp);
                    if (!same) unsafeAtomicAdd(on + 16 * cb, xn);
                }
            }
        }
    };
    if (wave_live) {
        scatter(acc0, c_h, 0);
        scatter(acc1, c_t, 1);
    }
    // ---- 6. the relation row: back through embed's normalisation of rel_embeddings
    __syncthreads();   // sGR is complete
    if (wave == 0) {
        float g[NB], dot = 0.f;
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) { g[cb] = sGR[colp[cb]]; dot = fmaf(rr[cb], g[cb], dot); }
        dot = grp16_sum(dot);
        if (lk == 0) {
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) {
                const float v = flgr != 0.f ? (g[cb] - rr[cb] * dot) * invr : g[cb] * invr;
                if (colp[cb] < dr && v != 0.f) unsafeAtomicAdd(A.g_rel + (int64_t)rel * dr + colp[cb], v);
            }
        }
    }
}

static size_t transr_rows2_lds_bytes(int nb, int de) {
    const int pitch = 16 * nb + 4;
    return (size_t)((de + 1) * pitch + 4 * 16 * pitch + 128 + 128 + 64 + 16 * nb) * sizeof(float);
}

// G_r = sum over the relation's rows of x^ (x) G  (x^ = normalised head row with GA, normalised tail row with GC): see the header.
template <int NBI, int NBJ>
__global__ __launch_bounds__(256, 2) void k_transr_g(TransRRowsArgs A) {
    constexpr int DPI = 16 * NBI, DPJ = 16 * NBJ, PA = DPI + 4, PB = DPJ + 4;
    constexpr int RBW = (NBI + 3) / 4;   // row blocks per wave
    __shared__ __attribute__((aligned(16))) float sA[2][16][PA], sB[2][16][PB];
    const int jbase = blockIdx.y * DPJ;   // the workgroup's half of the output columns
    int rel, tin;
    const int tile = strided_tile<kTrGRun>(A.tiles);
    if (tile >= A.tiles || !locate_tile(A.tile_off, A.tile_rel, A.R, tile, rel, tin)) return;
    if (tin % kTrGRun) return;
    const int de = A.de, dr = A.dr;
    if (jbase >= dr) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l = lane & 15, lk = lane >> 4;
    const int r0 = A.offsets[rel], r1 = A.offsets[rel + 1];
    const int g_lo = r0 + tin * kTrPairTile, g_hi = min(r1, g_lo + kTrGRun * kTrPairTile);
    const bool shared_rel = (r1 - r0) > kTrGRun * kTrPairTile;
    const int nslab = (g_hi - g_lo + 3) / 4;   // 4 pairs = 16 rows per slab
    f32x4v acc[RBW][NBJ];
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
        for (int cb = 0; cb < NBJ; ++cb) acc[rb][cb] = f32x4v{0.f, 0.f, 0.f, 0.f};
    // staging roles: slab row kq = tid / 16 = pair kq / 4, kind kq & 3 (pos h, neg h, pos t, neg t); columns c0 + 16 u
    const int kq = threadIdx.x >> 4, c0 = threadIdx.x & 15;
    int64_t id_e = 0, id_w = 0;
    float id_s = 0.f;
    auto resolve = [&](int sl) __attribute__((always_inline)) {   // entity id, workspace row and inverse norm of this thread's row of slab sl
        const int g = g_lo + 4 * sl + (kq >> 2);
        id_s = 0.f; id_e = 0; id_w = 0;
        if (sl < nslab && g < g_hi) {
            const int x = kq & 3;
            id_w = 4 * (int64_t)g + x;
            id_s = A.invs[id_w];
            if (id_s != 0.f) {
                const int pair = A.perm[g];
                id_e = x == 0 ? A.ph[pair] : x == 1 ? A.nh[pair] : x == 2 ? A.pt[pair] : A.nt[pair];
            }
        }
    };
    float sta[NBI], stb[NBJ], st_s = 0.f, nx_s = 0.f;
    auto fetch = [&]() __attribute__((always_inline)) {
        nx_s = id_s;
        const bool live = id_s != 0.f;
        const float* __restrict__ er = A.ent + id_e * de + c0;
        const float* __restrict__ gr = A.gws + id_w * dr + jbase + c0;
#pragma unroll
        for (int u = 0; u < NBI; ++u) sta[u] = (live && c0 + 16 * u < de) ? er[16 * u] : 0.f;
#pragma unroll
        for (int u = 0; u < NBJ; ++u) stb[u] = (live && jbase + c0 + 16 * u < dr) ? gr[16 * u] : 0.f;
    };
    int buf = 0;
    resolve(0);
    fetch();
    st_s = nx_s;
    resolve(1);
    for (int sl = 0; sl < nslab; ++sl) {
#pragma unroll
        for (int u = 0; u < NBI; ++u) sA[buf][kq][c0 + 16 * u] = sta[u] * st_s;
#pragma unroll
        for (int u = 0; u < NBJ; ++u) sB[buf][kq][BlkMapNat<NBJ>::pos(u, c0)] = stb[u];
        __syncthreads();
        if (sl + 1 < nslab) { fetch(); st_s = nx_s; }   // rows of slab sl + 1 (ids resolved one iteration ago)
        resolve(sl + 2);
        float b[2][NBJ], av[2][RBW];
        auto operands = [&](int kk, int slot) __attribute__((always_inline)) {
            read_blocks<NBJ>(&sB[buf][4 * kk + lk][0], l, b[slot]);
#pragma unroll
            for (int rb = 0; rb < RBW; ++rb) av[slot][rb] = wave + 4 * rb < NBI ? sA[buf][4 * kk + lk][16 * (wave + 4 * rb) + l] : 0.f;
        };
        operands(0, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk + 1 < 4) operands(kk + 1, (kk + 1) & 1);
            KGE_KEEP_READS_AHEAD();
#pragma unroll
            for (int rb = 0; rb < RBW; ++rb) {
                if (wave + 4 * rb < NBI) {   // wave-uniform
#pragma unroll
                    for (int cb = 0; cb < NBJ; ++cb)
                        acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk & 1][rb], b[kk & 1][cb], acc[rb][cb], 0, 0, 0);
                }
            }
        }
        buf ^= 1;
    }
    float* __restrict__ gM = A.g_mat + (int64_t)rel * de * dr;
    if (!shared_rel) {   // sole writer of these elements: the old values are requested together, on clamped addresses, before the first store
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb) {
            float old[NBJ][4];
#pragma unroll
            for (int cb = 0; cb < NBJ; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = min(16 * (wave + 4 * rb) + 4 * lk + q, de - 1), j = min(jbase + 16 * cb + l, dr - 1);
                    old[cb][q] = gM[(int64_t)i * dr + j];
                }
#pragma unroll
            for (int cb = 0; cb < NBJ; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[rb][cb][q] += old[cb][q];
        }
    }
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb) {
        if (wave + 4 * rb >= NBI) continue;
#pragma unroll
        for (int cb = 0; cb < NBJ; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = 16 * (wave + 4 * rb) + 4 * lk + q, j = jbase + 16 * cb + l;
                const float v = acc[rb][cb][q];
                if (i < de && j < dr) {
                    float* o = gM + (int64_t)i * dr + j;
                    if (shared_rel) { if (v != 0.f) unsafeAtomicAdd(o, v); } else *o = v;
                }
            }
    }
}

// k_transr_g, second form (d_e and d_r multiples of 4, 16-byte aligned tables): ONE workgroup per run with all output columns (the
// entity rows are gathered once, not once per column half), rows fetched as 16-byte pieces (a thread: two pieces of its row's
// entity half and two of its GA / GC half per slab instead of 11 dwords), and a ring of D slabs in flight in registers -- a run of
// the FB15k shape (24 pairs = 6 slabs) is otherwise one exposed gather latency per slab.
template <int NB, int D>
__global__ __launch_bounds__(256, 2) void k_transr_g2(TransRRowsArgs A) {
    constexpr int DP = 16 * NB, PA = DP + 4, PB = DP + 4;
    constexpr int RBW = (NB + 3) / 4;   // row blocks per wave
    constexpr int NV = (NB + 3) / 4;    // 16-byte pieces per thread and row half (16 threads per row: pieces f, f + 16)
    __shared__ __attribute__((aligned(16))) float sA[2][16][PA], sB[2][16][PB];
    KGE_TS_BEGIN(0)
    int rel, tin;
    const int tile = strided_tile<kTrGRun>(A.tiles);
    if (tile >= A.tiles || !locate_tile(A.tile_off, A.tile_rel, A.R, tile, rel, tin)) return;
    if (tin % kTrGRun) return;
    const int de = A.de, dr = A.dr;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l = lane & 15, lk = lane >> 4;
    const int r0 = A.offsets[rel], r1 = A.offsets[rel + 1];
    const int g_lo = r0 + tin * kTrPairTile, g_hi = min(r1, g_lo + kTrGRun * kTrPairTile);
    const bool shared_rel = (r1 - r0) > kTrGRun * kTrPairTile;
    const int nslab = (g_hi - g_lo + 3) / 4;   // 4 pairs = 16 rows per slab
    f32x4v acc[RBW][NB];
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) acc[rb][cb] = f32x4v{0.f, 0.f, 0.f, 0.f};
    // staging roles: slab row kq = tid / 16 = pair kq / 4, kind kq & 3 (pos h, neg h, pos t, neg t); pieces f and f + 16 of the row
    const int kq = threadIdx.x >> 4, f = threadIdx.x & 15;
    float4 ra[D][NV], rb_[D][NV];
    float rs[D];
    // Every load below is UNCONDITIONAL on a clamped (always valid) address, dead rows / columns are zeroed by selects afterwards: a
    // load inside a divergent branch ends in s_waitcnt vmcnt(0) at the join, which drains the row requests in flight.
    const int64_t* __restrict__ idcol = (kq & 3) == 0 ? A.ph : (kq & 3) == 1 ? A.nh : (kq & 3) == 2 ? A.pt : A.nt;
    auto resolve1 = [&](int sl, int64_t& idw, float& inv, int& pair) __attribute__((always_inline)) {
        const int g = g_lo + 4 * sl + (kq >> 2);
        const int gq = min(g, g_hi - 1);
        idw = 4 * (int64_t)gq + (kq & 3);
        const float v = A.invs[idw];
        pair = A.perm[gq];
        inv = g < g_hi ? v : 0.f;
    };
    auto resolve2 = [&](float inv, int pair) __attribute__((always_inline)) -> int64_t { return idcol[pair]; };
    auto fetch = [&](auto slot, int64_t ide, int64_t idw, float inv) __attribute__((always_inline)) {
        constexpr int S = decltype(slot)::value;
        rs[S] = inv;
        const float* __restrict__ er = A.ent + ide * de;
        const float* __restrict__ gr = A.gws + idw * dr;
#pragma unroll
        for (int v = 0; v < NV; ++v) {   // (raw: the selects wait until the LDS store -- a select here would wait for the load)
            const int c = 4 * (f + 16 * v);
            ra[S][v] = *reinterpret_cast<const float4*>(er + min(c, de - 4));
            rb_[S][v] = *reinterpret_cast<const float4*>(gr + min(c, dr - 4));
        }
    };
    // prologue: the first D slabs' chains side by side (inverse norm + pair -> id -> rows), then the ids of slab D
    int64_t p_idw[D + 2], p_ide[D + 1];
    float p_inv[D + 2];
    int p_pair[D + 2];
#pragma unroll
    for (int s = 0; s <= D + 1; ++s) resolve1(s, p_idw[s], p_inv[s], p_pair[s]);
#pragma unroll
    for (int s = 0; s <= D; ++s) p_ide[s] = resolve2(p_inv[s], p_pair[s]);
    unroll_seq([&](auto sc) __attribute__((always_inline)) { fetch(sc, p_ide[decltype(sc)::value], p_idw[decltype(sc)::value], p_inv[decltype(sc)::value]); },
               std::make_integer_sequence<int, D>{});
    // in the loop every load of an iteration depends only on values loaded in EARLIER iterations (the memory counter is in order: a
    // dependent chain inside an iteration would drain the row requests issued before it): at slab sl the ids of slab sl + D + 1
    // (from the pair resolved one iteration ago), inverse norm + pair of slab sl + D + 2, then the rows of slab sl + D
    int64_t n_ide = p_ide[D], n_idw = p_idw[D];          // slab sl + D: id known
    float n_inv = p_inv[D];
    int64_t m_idw = p_idw[D + 1];                        // slab sl + D + 1: inverse norm and pair known
    float m_inv = p_inv[D + 1];
    int m_pair = p_pair[D + 1];
    int buf = 0;
    auto step = [&](auto slot, int sl) __attribute__((always_inline)) {
        constexpr int S = decltype(slot)::value;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = 4 * (f + 16 * v);
            if (c < DP) {
                const float sc = c < de ? rs[S] : 0.f;                 // the row's inverse norm; 0 for a dead row, past d_e
                const bool lb = rs[S] != 0.f && c < dr;                // (a dead row of the workspace was never written)
                *reinterpret_cast<float4*>(&sA[buf][kq][c]) = float4{ra[S][v].x * sc, ra[S][v].y * sc, ra[S][v].z * sc, ra[S][v].w * sc};
                const int u = c >> 4, c0 = c & 15;   // columns c .. c + 3 of block u: positions 4 apart
                sB[buf][kq][BlkMapNat<NB>::pos(u, c0)] = lb ? rb_[S][v].x : 0.f;
                sB[buf][kq][BlkMapNat<NB>::pos(u, c0 + 1)] = lb ? rb_[S][v].y : 0.f;
                sB[buf][kq][BlkMapNat<NB>::pos(u, c0 + 2)] = lb ? rb_[S][v].z : 0.f;
                sB[buf][kq][BlkMapNat<NB>::pos(u, c0 + 3)] = lb ? rb_[S][v].w : 0.f;
            }
        }
        __syncthreads();
        {
            const int64_t ide1 = resolve2(m_inv, m_pair);
            int64_t idw2; float inv2; int pair2;
            resolve1(sl + D + 2, idw2, inv2, pair2);
            fetch(slot, n_ide, n_idw, n_inv);   // rows of slab sl + D into the registers just emptied (past the run: a dead row)
            n_ide = ide1; n_idw = m_idw; n_inv = m_inv;
            m_idw = idw2; m_inv = inv2; m_pair = pair2;
        }
        float b[2][NB], av[2][RBW];
        auto operands = [&](int kk, int s2) __attribute__((always_inline)) {
            read_blocks<NB>(&sB[buf][4 * kk + lk][0], l, b[s2]);
#pragma unroll
            for (int rb = 0; rb < RBW; ++rb) av[s2][rb] = wave + 4 * rb < NB ? sA[buf][4 * kk + lk][16 * (wave + 4 * rb) + l] : 0.f;
        };
        operands(0, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk + 1 < 4) operands(kk + 1, (kk + 1) & 1);
            KGE_KEEP_READS_AHEAD();
#pragma unroll
            for (int rb = 0; rb < RBW; ++rb) {
                if (wave + 4 * rb < NB) {   // wave-uniform
#pragma unroll
                    for (int cb = 0; cb < NB; ++cb)
                        acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk & 1][rb], b[kk & 1][cb], acc[rb][cb], 0, 0, 0);
                }
            }
        }
        buf ^= 1;
    };
    int sl = 0;
    for (; sl + D <= nslab; sl += D)   // whole groups of D steps without a branch between them: the memory counter waits stay partial
        unroll_seq([&](auto sc) __attribute__((always_inline)) { step(sc, sl + decltype(sc)::value); }, std::make_integer_sequence<int, D>{});
    unroll_seq([&](auto sc) __attribute__((always_inline)) {
        if (sl + decltype(sc)::value < nslab) step(sc, sl + decltype(sc)::value);   // workgroup-uniform
    }, std::make_integer_sequence<int, D>{});
    float* __restrict__ gM = A.g_mat + (int64_t)rel * de * dr;
    if (!shared_rel) {   // sole writer of these elements: the old values of a row block are requested together, on clamped addresses
#pragma unroll      //   (a load inside a divergent branch would be waited for one by one), before the first store
        for (int rb = 0; rb < RBW; ++rb) {
            float old[NB][4];
#pragma unroll
            for (int cb = 0; cb < NB; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = min(16 * (wave + 4 * rb) + 4 * lk + q, de - 1), j = min(16 * cb + l, dr - 1);
                    old[cb][q] = gM[(int64_t)i * dr + j];
                }
#pragma unroll
            for (int cb = 0; cb < NB; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[rb][cb][q] += old[cb][q];
        }
    }
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb) {
        if (wave + 4 * rb >= NB) continue;
#pragma unroll
        for (int cb = 0; cb < NB; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = 16 * (wave + 4 * rb) + 4 * lk + q, j = 16 * cb + l;
                const float v = acc[rb][cb][q];
                if (i < de && j < dr) {
                    float* o = gM + (int64_t)i * dr + j;
                    if (shared_rel) { if (v != 0.f) unsafeAtomicAdd(o, v); } else *o = v;
                }
            }
    }
    KGE_TS_END(0, 1 + nslab)
}

// workspace of the two-launch step: the grouping of n pairs in 16-pair tiles, then invs [4 n] and gws [4 n][dr]
static size_t transr_rows_group_ints(int64_t R, int64_t n) { return (size_t)(4 * (R + 1) + n + (n / kTrPairTile + R + 1) + 8); }
size_t transr_rows_ws_bytes(const kge_model_desc* m, int64_t n) {
    const size_t gi = (transr_rows_group_ints(m->tot_relation, n) * sizeof(int) + 255) & ~(size_t)255;
    return gi + (size_t)4 * n * (m->rel_dim + 1) * sizeof(float);
}

bool transr_rows_ok(const kge_model_desc* m, int64_t n, size_t ws_bytes) {
    if (!(m->dim >= 1 && m->rel_dim >= 1 && m->dim <= 128 && m->rel_dim <= 128 && n >= 1 && n < (1ll << 29) &&
          ws_bytes >= transr_rows_ws_bytes(m, n)))
        return false;
    // the tile's dynamic LDS (about 101 KB at d = 128) must fit what THIS device grants a workgroup on opt-in; a shape that does
    // not fit falls back to the tile kernels of kge_transr.hip instead of failing at launch
    const int nb = (max(m->dim, m->rel_dim) + 15) / 16;
    int dev = 0, optin = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&optin, hipDeviceAttributeSharedMemPerBlockOptin, dev) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return transr_rows2_lds_bytes(nb, m->dim) <= (size_t)optin;
}

template <int NB>
static bool launch_transr_rows_nb(const TransRRowsArgs& a, unsigned tiles, hipStream_t s) {
    const bool vec = ((a.de | a.dr) & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(a.ent) | reinterpret_cast<uintptr_t>(a.mat) | reinterpret_cast<uintptr_t>(a.gws)) & 15) == 0;
    const size_t lds = transr_rows2_lds_bytes(NB, a.de);
    const bool l1 = a.l1 != 0;
    // set on every launch (a host-side table write, ~0.1 us): a process-wide "done" flag would leave the attribute unset on a second
    // device or under a concurrent first launch, and the > 64 KB dynamic-LDS launch would then fail instead of running
    auto go = [&](auto kern) -> bool {
        if (lds > 64 * 1024 &&
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            set_error("TransR pair step: the device refused %zu bytes of dynamic LDS", lds);
            return false;
        }
        hipLaunchKernelGGL(kern, dim3((tiles + 1) / 2 * 2), dim3(256), lds, s, a);
        return true;
    };
    const bool launched = vec ? (l1 ? go(k_transr_rows<NB, true, true>) : go(k_transr_rows<NB, true, false>))
                              : (l1 ? go(k_transr_rows<NB, false, true>) : go(k_transr_rows<NB, false, false>));
    if (!launched) return false;
    if (vec && switch_value("TRANSR_G") != 0) {   // (KGE_TRANSR_G=0: the dword-gather form, A/B)
        hipLaunchKernelGGL((k_transr_g2<NB, 3>), dim3((tiles + kTrGRun - 1) / kTrGRun * kTrGRun), dim3(256), 0, s, a);
        return true;
    }
    constexpr int JA = (NB + 1) / 2;   // column blocks per half (an odd NB leaves one masked block in the second half)
    hipLaunchKernelGGL((k_transr_g<NB, JA>), dim3((tiles + kTrGRun - 1) / kTrGRun * kTrGRun, NB > 1 ? 2 : 1), dim3(256), 0, s, a);
    return true;
}

// negatives share pr (the caller passed nr == pr); ws: the pairwise step's scorer workspace
int launch_transr_pair_step(const kge_model_desc* m, const int64_t* ph, const int64_t* pr, const int64_t* pt, const int64_t* nh,
                            const int64_t* nt, int64_t n, float margin, float* loss, void* ws, size_t ws_bytes, hipStream_t s) {
    if (!transr_rows_ok(m, n, ws_bytes)) { set_error("TransR pair step: unsupported shape or workspace"); return -1; }
    const int64_t R = m->tot_relation;
    const GroupWs g = carve_group_ws(ws, R, n);       // (tile_rel, the last array, holds n / 16 + R + 1 entries here)
    int rc = group_by_relation_split(id_whole(pr, n), n, R, g, s, nullptr, 0, kTrPairTile);
    if (rc) return rc;
    const size_t gi = (transr_rows_group_ints(R, n) * sizeof(int) + 255) & ~(size_t)255;
    TransRRowsArgs a;
    a.ent = m->tables[0]; a.rel = m->tables[1]; a.mat = m->tables[2];
    a.g_ent = m->grads[0]; a.g_rel = m->grads[1]; a.g_mat = m->grads[2];
    a.ph = ph; a.pt = pt; a.nh = nh; a.nt = nt;
    a.offsets = g.offsets; a.tile_off = g.tile_off; a.tile_rel = g.tile_rel; a.perm = g.perm;
    a.R = (int)R; a.de = m->dim; a.dr = m->rel_dim; a.l1 = (m->flags & KGE_FLAG_L1) ? 1 : 0;
    a.margin = margin; a.loss = loss;
    a.invs = (float*)((char*)ws + gi);
    a.gws = a.invs + 4 * n;
    const unsigned tiles = (unsigned)(n / kTrPairTile + R + 1);
    a.tiles = (int)tiles;
    const int nb = (max(m->dim, m->rel_dim) + 15) / 16;
    bool ok;
    switch (nb) {
        case 1: ok = launch_transr_rows_nb<1>(a, tiles, s); break;
        case 2: ok = launch_transr_rows_nb<2>(a, tiles, s); break;
        case 3: ok = launch_transr_rows_nb<3>(a, tiles, s); break;
        case 4: ok = launch_transr_rows_nb<4>(a, tiles, s); break;
        case 5: ok = launch_transr_rows_nb<5>(a, tiles, s); break;
        case 6: ok = launch_transr_rows_nb<6>(a, tiles, s); break;
        case 7: ok = launch_transr_rows_nb<7>(a, tiles, s); break;
        default: ok = launch_transr_rows_nb<8>(a, tiles, s); break;
    }
    if (!ok) return -1;
    return check_launch("k_transr_rows / k_transr_g");
}

}  // namespace kge
