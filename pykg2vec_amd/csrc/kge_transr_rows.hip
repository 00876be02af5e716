// kge_transr_rows.hip -- TransR (pykg2vec/models/pairwise.py:367-470), the pairwise hinge step of LARGE batches in two launches.
//
// kge_transr.hip's tile kernels (one 32-triple tile per workgroup, forward / hinge / backward as three launches, the forward run
// twice, M_r operands straight from global memory, d_e x d_r float atomics per tile for the matrix gradient) stop at 14 TFLOP/s
// on the FB15k shape at B = 32 768.  A negative keeps its positive's relation (every sampler of the reference corrupts heads or
// tails: the caller says so by passing the SAME buffer for nr and pr, as for RESCAL), so PAIRS are grouped by relation and:
//
//   k_transr_rows<NB>   a workgroup takes 32 pairs of one relation (two consecutive 16-pair tiles), a wave 8 pairs = one 16-row
//                       block with a pair's positive and negative in adjacent rows.  The normalised rows H^ and T^ are the A
//                       operands (registers) of ONE pass over M_r (LDS slabs, v_mfma_f32_16x16x4_f32, two accumulator sets), the
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
    float* gws;    // [4 n][dr]  grouped pair g: GA of the positive, GA of the negative, GC of the positive, GC of the negative
};

__device__ __forceinline__ float grp16_sum(float v) {
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
    return v;
}

template <int NB>
__global__ __launch_bounds__(256, 2) void k_transr_rows(TransRRowsArgs A) {
    constexpr int DP = 16 * NB, PITCH = DP + 4, NK = 4 * NB;
    // one region, two uses: slab double buffer of the passes ([2][16][PITCH]) / the waves' transpose buffers ([4][16][PITCH])
    __shared__ __attribute__((aligned(16))) float sBuf[4 * 16 * PITCH];
    __shared__ float sInv[4][2][16];     // per wave, side (h, t), row: 1 / max(|x|, eps)
    __shared__ float sFlg[4][2][16];     // |x| > eps
    __shared__ float sGR[DP];            // the workgroup's gradient with respect to r^
    int rel, tin;
    if (!locate_tile(A.tile_off, A.tile_rel, A.R, blockIdx.x, rel, tin)) return;
    if (tin & 1) return;                       // (the workgroup of an even tile takes the odd one after it as well)
    const int de = A.de, dr = A.dr;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l = lane & 15, lk = lane >> 4;
    const int g_lo = A.offsets[rel] + tin * kTrPairTile, g_hi = min(A.offsets[rel + 1], g_lo + 2 * kTrPairTile);
    const bool wave_live = g_lo + 8 * wave < g_hi;     // wave-uniform: a wave without pairs only helps staging
    float (*sW)[16][PITCH] = reinterpret_cast<float (*)[16][PITCH]>(sBuf);
    float (*sX)[PITCH] = reinterpret_cast<float (*)[PITCH]>(sBuf + wave * 16 * PITCH);
    if (threadIdx.x < DP) sGR[threadIdx.x] = 0.f;

    // A-operand row of this lane: row l of the wave = pair 8 wave + l / 2, side l & 1 (0 positive, 1 negative)
    int a_h, a_t;
    bool a_on;
    {
        const int gp = g_lo + 8 * wave + (l >> 1);
        a_on = gp < g_hi;
        const int pair = a_on ? A.perm[gp] : 0;
        a_h = a_on ? (int)((l & 1) ? A.nh[pair] : A.ph[pair]) : 0;
        a_t = a_on ? (int)((l & 1) ? A.nt[pair] : A.pt[pair]) : 0;
    }
    // accumulator rows of this lane: 4 lk + q = pairs 2 lk (q = 0 positive, 1 negative) and 2 lk + 1 (q = 2, 3)
    int c_h[4], c_t[4], c_g[2];
    bool c_on[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        c_g[j] = g_lo + 8 * wave + 2 * lk + j;
        c_on[j] = c_g[j] < g_hi;
        const int pair = c_on[j] ? A.perm[c_g[j]] : 0;
        c_h[2 * j] = c_on[j] ? (int)A.ph[pair] : 0; c_h[2 * j + 1] = c_on[j] ? (int)A.nh[pair] : 0;
        c_t[2 * j] = c_on[j] ? (int)A.pt[pair] : 0; c_t[2 * j + 1] = c_on[j] ? (int)A.nt[pair] : 0;
    }

    // ---- 0. rows in A layout, normalised (embed: F.normalize(p=2, dim=-1), eps 1e-12)
    float a0[NK], a1[NK];
    unroll_seq([&](auto ksc) __attribute__((always_inline)) {
        constexpr int ks = decltype(ksc)::value;
        const int kk = 4 * ks + lk;
        a0[ks] = (a_on && kk < de) ? A.ent[(int64_t)a_h * de + kk] : 0.f;
        a1[ks] = (a_on && kk < de) ? A.ent[(int64_t)a_t * de + kk] : 0.f;
    }, std::make_integer_sequence<int, NK>{});
    {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) { s0 = fmaf(a0[ks], a0[ks], s0); s1 = fmaf(a1[ks], a1[ks], s1); }
        s0 += __shfl_xor(s0, 16, 64); s0 += __shfl_xor(s0, 32, 64);
        s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
        const float n0 = sqrtf(s0), n1 = sqrtf(s1);
        const float i0 = 1.0f / fmaxf(n0, kEpsNormalize), i1 = 1.0f / fmaxf(n1, kEpsNormalize);
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) { a0[ks] *= i0; a1[ks] *= i1; }
        if (lk == 0) {
            sInv[wave][0][l] = i0; sInv[wave][1][l] = i1;
            sFlg[wave][0][l] = n0 > kEpsNormalize ? 1.f : 0.f; sFlg[wave][1][l] = n1 > kEpsNormalize ? 1.f : 0.f;
        }
    }

    f32x4v acc0[NB], acc1[NB];
    const float* __restrict__ M = A.mat + (int64_t)rel * de * dr;
    float st[NB];
    int buf = 0;
    // one pass over M_r with both A operand sets.  TR = false: B[kq][c] = M[16 kb + kq][c] (K = d_e, columns d_r: X M);
    // TR = true: B[kq][c] = M[c][16 kb + kq] (K = d_r, columns d_e: G M^T).  Slabs past K are skipped (workgroup-uniform).
    auto pass = [&](auto tr_tag) __attribute__((always_inline)) {
        constexpr bool TR = decltype(tr_tag)::value;
        const int Kd = TR ? dr : de, Cd = TR ? de : dr;
        const int kq = TR ? (threadIdx.x & 15) : (threadIdx.x >> 4), c0 = TR ? (threadIdx.x >> 4) : (threadIdx.x & 15);
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) { acc0[cb] = f32x4v{0.f, 0.f, 0.f, 0.f}; acc1[cb] = f32x4v{0.f, 0.f, 0.f, 0.f}; }
        auto fetch = [&](int kb) __attribute__((always_inline)) {
            const int kr = 16 * kb + kq;
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int c = c0 + 16 * u;
                st[u] = (kr < Kd && c < Cd) ? (TR ? M[(int64_t)c * dr + kr] : M[(int64_t)kr * dr + c]) : 0.f;
            }
        };
        fetch(0);
        unroll_seq([&](auto kbc) __attribute__((always_inline)) {
            constexpr int kb = decltype(kbc)::value;
            if (16 * kb < Kd) {
#pragma unroll
                for (int u = 0; u < NB; ++u) sW[buf][kq][BlkMapNat<NB>::pos(u, c0)] = st[u];   // natural accumulator columns
                __syncthreads();   // slab kb is in LDS; everybody finished reading the buffer that is written next
                if (kb + 1 < NB && 16 * (kb + 1) < Kd) fetch(kb + 1);
                if (wave_live) {
                    float b[2][NB];
                    read_blocks<NB>(&sW[buf][lk][0], l, b[0]);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        if (kk + 1 < 4) read_blocks<NB>(&sW[buf][4 * (kk + 1) + lk][0], l, b[(kk + 1) & 1]);
                        KGE_KEEP_READS_AHEAD();
#pragma unroll
                        for (int cb = 0; cb < NB; ++cb) {
                            acc0[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[4 * kb + kk], b[kk & 1][cb], acc0[cb], 0, 0, 0);
                            acc1[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[4 * kb + kk], b[kk & 1][cb], acc1[cb], 0, 0, 0);
                        }
                    }
                }
                buf ^= 1;
            }
        }, std::make_integer_sequence<int, NB>{});
    };
    // ---- 1. HP = H^ M, TP = T^ M
    pass(std::false_type{});

    // ---- 2. the TransE tail per row (pairwise.py:459-470) in accumulator layout: row 4 lk + q, column 16 cb + l
    float rr[NB], invr, flgr, ib, flgb;
    {
        float ss = 0.f;
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
            const int col = 16 * cb + l;
            rr[cb] = col < dr ? A.rel[(int64_t)rel * dr + col] : 0.f;
            ss = fmaf(rr[cb], rr[cb], ss);
        }
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
    const bool l1 = A.l1 != 0;
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
            p = l1 ? p + fabsf(u) : fmaf(u, u, p);
        }
        p = grp16_sum(p);
        sc[q] = l1 ? p : sqrtf(p);
    }
    float c[2], hl = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float v = sc[2 * j] + A.margin - sc[2 * j + 1];
        c[j] = c_on[j] ? (v > 0.f ? 1.f : (v == 0.f ? 0.5f : 0.f)) : 0.f;
        if (c_on[j]) hl += fmaxf(v, 0.f);
    }
    {
        const float tot = wave_sum(l == 0 ? hl : 0.f);
        if (lane == 0 && tot != 0.f) unsafeAtomicAdd(A.loss + (blockIdx.x % kLossSlots) * kLossStride, tot);
    }
    __syncthreads();   // (sInv / sFlg / sGR visible; every wave is done with the pass's last slab: sBuf becomes the transpose buffers)
    if (l == 0) {      // what k_transr_g needs of a row besides GA / GC: its inverse norm, 0 where the pair has no gradient
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = q >> 1;
            if (c_on[j]) {
                A.invs[4 * (int64_t)c_g[j] + (q & 1)] = c[j] != 0.f ? sInv[wave][0][4 * lk + q] : 0.f;
                A.invs[4 * (int64_t)c_g[j] + 2 + (q & 1)] = c[j] != 0.f ? sInv[wave][1][4 * lk + q] : 0.f;
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
        v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
        if (lk == 0 && v != 0.f) atomicAdd(&sGR[16 * cb + l], v);
    }

    // ---- 3. GA / GC: to the workspace (rows of pairs with a gradient) and, through the wave's transpose buffer, into A layout
    auto to_a_layout = [&](f32x4v (&acc)[NB], float (&a)[NK], int x0) __attribute__((always_inline)) {
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
            const int col = 16 * cb + l;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                sX[4 * lk + q][col] = acc[cb][q];
                if (c[q >> 1] != 0.f && col < dr) A.gws[(4 * (int64_t)c_g[q >> 1] + x0 + (q & 1)) * dr + col] = acc[cb][q];
            }
        }
        __syncthreads();
        unroll_seq([&](auto ksc) __attribute__((always_inline)) {
            constexpr int ks = decltype(ksc)::value;
            a[ks] = sX[l][4 * ks + lk];
        }, std::make_integer_sequence<int, NK>{});
        __syncthreads();
    };
    to_a_layout(acc0, a0, 0);
    to_a_layout(acc1, a1, 2);

    // ---- 4. GH^ = GA M^T, GT^ = GC M^T
    pass(std::true_type{});

    // ---- 5. back through the entity normalisation, row-wise atomics (16 consecutive floats per lane group)
    auto scatter = [&](f32x4v (&acc)[NB], const int (&ids)[4], int side) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool on = c[q >> 1] != 0.f;
            const float inv = sInv[wave][side][4 * lk + q];
            const bool f = sFlg[wave][side][4 * lk + q] != 0.f;
            float x[NB], dot = 0.f;
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) {
                const int col = 16 * cb + l;
                x[cb] = (on && col < de) ? A.ent[(int64_t)ids[q] * de + col] * inv : 0.f;
                dot = fmaf(x[cb], acc[cb][q], dot);
            }
            dot = grp16_sum(dot);
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) acc[cb][q] = f ? (acc[cb][q] - x[cb] * dot) * inv : acc[cb][q] * inv;
        }
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
            const int col = 16 * cb + l;
            if (col < de) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (c[j] == 0.f) continue;
                    const float xp = acc[cb][2 * j], xn = acc[cb][2 * j + 1];
                    if (ids[2 * j] == ids[2 * j + 1]) unsafeAtomicAdd(A.g_ent + (int64_t)ids[2 * j] * de + col, xp + xn);
                    else {
                        unsafeAtomicAdd(A.g_ent + (int64_t)ids[2 * j] * de + col, xp);
                        unsafeAtomicAdd(A.g_ent + (int64_t)ids[2 * j + 1] * de + col, xn);
                    }
                }
            }
        }
    };
    if (wave_live) {
        scatter(acc0, c_h, 0);
        scatter(acc1, c_t, 1);
    }
    // ---- 6. the relation row: back through embed's normalisation of rel_embeddings (sGR is complete: barriers of step 3 / 4)
    if (wave == 0) {
        float g[NB], dot = 0.f;
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) { g[cb] = sGR[16 * cb + l]; dot = fmaf(rr[cb], g[cb], dot); }
        dot = grp16_sum(dot);
        if (lk == 0) {
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) {
                const int col = 16 * cb + l;
                const float v = flgr != 0.f ? (g[cb] - rr[cb] * dot) * invr : g[cb] * invr;
                if (col < dr && v != 0.f) unsafeAtomicAdd(A.g_rel + (int64_t)rel * dr + col, v);
            }
        }
    }
}

// G_r = sum over the relation's rows of x^ (x) G  (x^ = normalised head row with GA, normalised tail row with GC): see the header.
template <int NBI, int NBJ>
__global__ __launch_bounds__(256, 2) void k_transr_g(TransRRowsArgs A) {
    constexpr int DPI = 16 * NBI, DPJ = 16 * NBJ, PA = DPI + 4, PB = DPJ + 4;
    constexpr int RBW = (NBI + 3) / 4;   // row blocks per wave
    __shared__ __attribute__((aligned(16))) float sA[2][16][PA], sB[2][16][PB];
    const int jbase = blockIdx.y * DPJ;   // the workgroup's half of the output columns
    int rel, tin;
    if (!locate_tile(A.tile_off, A.tile_rel, A.R, blockIdx.x, rel, tin)) return;
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
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb) {
        if (wave + 4 * rb >= NBI) continue;
#pragma unroll
        for (int cb = 0; cb < NBJ; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = 16 * (wave + 4 * rb) + 4 * lk + q, j = jbase + 16 * cb + l;
                const float v = acc[rb][cb][q];
                if (i < de && j < dr && v != 0.f) {
                    float* o = gM + (int64_t)i * dr + j;
                    if (shared_rel) unsafeAtomicAdd(o, v); else *o += v;
                }
            }
    }
}

// workspace of the two-launch step: the grouping of n pairs in 16-pair tiles, then invs [4 n] and gws [4 n][dr]
static size_t transr_rows_group_ints(int64_t R, int64_t n) { return (size_t)(4 * (R + 1) + n + (n / kTrPairTile + R + 1) + 8); }
size_t transr_rows_ws_bytes(const kge_model_desc* m, int64_t n) {
    const size_t gi = (transr_rows_group_ints(m->tot_relation, n) * sizeof(int) + 255) & ~(size_t)255;
    return gi + (size_t)4 * n * (m->rel_dim + 1) * sizeof(float);
}

bool transr_rows_ok(const kge_model_desc* m, int64_t n, size_t ws_bytes) {
    return m->dim >= 1 && m->rel_dim >= 1 && m->dim <= 128 && m->rel_dim <= 128 && n >= 1 && n < (1ll << 29) &&
           ws_bytes >= transr_rows_ws_bytes(m, n);
}

template <int NB>
static void launch_transr_rows_nb(const TransRRowsArgs& a, unsigned tiles, hipStream_t s) {
    hipLaunchKernelGGL(k_transr_rows<NB>, dim3(tiles), dim3(256), 0, s, a);
    constexpr int JA = (NB + 1) / 2;   // column blocks per half (an odd NB leaves one masked block in the second half)
    hipLaunchKernelGGL((k_transr_g<NB, JA>), dim3(tiles, NB > 1 ? 2 : 1), dim3(256), 0, s, a);
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
    const int nb = (max(m->dim, m->rel_dim) + 15) / 16;
    switch (nb) {
        case 1: launch_transr_rows_nb<1>(a, tiles, s); break;
        case 2: launch_transr_rows_nb<2>(a, tiles, s); break;
        case 3: launch_transr_rows_nb<3>(a, tiles, s); break;
        case 4: launch_transr_rows_nb<4>(a, tiles, s); break;
        case 5: launch_transr_rows_nb<5>(a, tiles, s); break;
        case 6: launch_transr_rows_nb<6>(a, tiles, s); break;
        case 7: launch_transr_rows_nb<7>(a, tiles, s); break;
        default: launch_transr_rows_nb<8>(a, tiles, s); break;
    }
    return check_launch("k_transr_rows / k_transr_g");
}

}  // namespace kge
