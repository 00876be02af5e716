// kge_transr.hip -- TransR (pykg2vec/models/pairwise.py:367-470): entities are L2-normalised, projected into the
// relation space by a per-relation matrix M_r = rel_matrix[r].view(d_e, d_r), and scored with the TransE tail
//     s = || n(h^ M_r) + n(r^) - n(t^ M_r) ||_{1|2},     x^ = n(x) = x / max(||x||, 1e-12).
// The reference gathers a [B, d_e, d_r] tensor per forward and its autograd scatters B dense outer products back.
// Here the batch is grouped by relation on the device (kge_relgroup.h) and every (relation, 32-triple tile) is one
// workgroup that keeps the normalised rows in LDS and runs its small GEMMs on v_mfma_f32_32x32x2_f32 (exact fp32):
//     HP = H^ M_r , TP = T^ M_r            [32,d_e]x[d_e,d_r]     forward
//     G_M = H^^T GA + T^^T GC              [d_e,32]x[32,d_r]      grad of M_r: one atomic per element per TILE
//     GH = GA M_r^T , GT = GC M_r^T        [32,d_r]x[d_r,d_e]     grads of the normalised rows
// with the tail (three normalisations, distance, their backward) done per triple by one wave in between.
// MFMA operand maps as in kge_dense.hip.  Bound: matrix cores / L2 (M_r is read once per tile, not once per triple).
//
// Evaluation (utils/evaluator.py:249-273 over TransR.forward): the candidate table must be projected by M_r, so one
// kge_eval_ranks call serves test triples of ONE relation (the host groups them): k_transr_project writes
// n(e^ M_r) for every entity straight into the sweep layout, k_transr_queries the two query vectors, and the plain
// L1 / L2 sweep of kge_eval.hip does the rest.
#include "kge_internal.h"
#include "kge_relgroup.h"

namespace kge {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TR_MAXD = 128;  // rows are held two elements per lane in the per-triple passes

struct TransRArgs {
    const float* ent; const float* rel; const float* mat;
    float* g_ent; float* g_rel; float* g_mat;
    IdSplit h, t;
    const int* offsets; const int* tile_off; const int* tile_rel; const int* perm;
    int R, de, dr, l1;
    const float* dscore; float* scores;
};

__device__ __forceinline__ int mfma_row(int reg, int lk) { return (reg & 3) + 8 * (reg >> 2) + 4 * lk; }

// MODE 0: scores.  MODE 1: gradients (needs dscore).
template <int MODE>
__global__ __launch_bounds__(256) void k_transr(TransRArgs A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int rel, tin;
    if (!locate_tile(A.tile_off, A.tile_rel, A.R, blockIdx.x, rel, tin)) return;
    const int de = A.de, dr = A.dr;
    const int Se = (de + 1) | 1, Sr = (dr + 1) | 1;  // odd LDS row strides: conflict-free column reads
    float* sH = smem;                   // [32][Se]  h^
    float* sT = sH + TILE * Se;         // [32][Se]  t^
    float* sHP = sT + TILE * Se;        // [32][Sr]  h^ M   (MODE 1: overwritten by GA = d/d(h^ M))
    float* sTP = sHP + TILE * Sr;       // [32][Sr]  t^ M   (MODE 1: overwritten by GC)
    float* sGH = sTP + TILE * Sr;       // [32][Se]  MODE 1: GA M^T
    float* sGT = sGH + (MODE == 1 ? TILE * Se : 0);
    float* sR = sGT + (MODE == 1 ? TILE * Se : 0);  // [dr] r^
    float* sGR = sR + TR_MAXD;          // [dr] sum of the tile's gradients wrt r^
    float* sIh = sGR + TR_MAXD;         // [32] 1/max(|h|,eps)
    float* sIt = sIh + TILE;
    float* sFh = sIt + TILE;            // [32] |h| > eps
    float* sFt = sFh + TILE;
    float* sDs = sFt + TILE;            // [32]
    float* sRs = sDs + TILE;            // [2] 1/max(|r|,eps), |r| > eps
    int* sRow = (int*)(sRs + 2);        // [32]
    long long* sHid = (long long*)(sRow + TILE);  // byte offset from smem is a multiple of 8 (all counts above are even)
    long long* sTid = sHid + TILE;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int g0 = A.offsets[rel] + tin * TILE;
    const int cnt = min(TILE, A.offsets[rel + 1] - g0);
    if (threadIdx.x < TILE) {
        const int row = threadIdx.x < cnt ? A.perm[g0 + threadIdx.x] : -1;
        sRow[threadIdx.x] = row;
        sHid[threadIdx.x] = row >= 0 ? A.h.at(row) : 0;
        sTid[threadIdx.x] = row >= 0 ? A.t.at(row) : 0;
        sDs[threadIdx.x] = (MODE == 1 && row >= 0) ? A.dscore[row] : 0.f;
    }
    if (threadIdx.x < TR_MAXD) sGR[threadIdx.x] = 0.f;
    __syncthreads();
    const int c0 = lane, c1 = lane + 64;

    // ---- 1. gather + normalise the 2 x 32 entity rows: row q = (triple q>>1, side q&1) goes to wave q%4, so a short tile
    // still spreads over all four waves; four rows' loads are issued before the first reduction; padding rows are zeros
    for (int jb = 0; jb < 16; jb += 4) {
        float v0[4], v1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = wave + 4 * (jb + u), i = q >> 1, side = q & 1;
            const float* row = A.ent + (side ? sTid[i] : sHid[i]) * (int64_t)de;
            v0[u] = (i < cnt && c0 < de) ? row[c0] : 0.f;
            v1[u] = (i < cnt && c1 < de) ? row[c1] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = wave + 4 * (jb + u), i = q >> 1, side = q & 1;
            float* dst = (side ? sT : sH) + i * Se;
            if (i >= cnt) {  // wave-uniform
                if (c0 < de) dst[c0] = 0.f;
                if (c1 < de) dst[c1] = 0.f;
                continue;
            }
            const float nrm = sqrtf(wave_sum(fmaf(v1[u], v1[u], v0[u] * v0[u])));
            const float inv = 1.0f / fmaxf(nrm, kEpsNormalize);
            if (c0 < de) dst[c0] = v0[u] * inv;
            if (c1 < de) dst[c1] = v1[u] * inv;
            if (lane == 0) { (side ? sIt : sIh)[i] = inv; (side ? sFt : sFh)[i] = nrm > kEpsNormalize ? 1.f : 0.f; }
        }
    }
    if (wave == 0) {
        const float* row = A.rel + (int64_t)rel * dr;
        const float v0 = c0 < dr ? row[c0] : 0.f, v1 = c1 < dr ? row[c1] : 0.f;
        const float nrm = sqrtf(wave_sum(fmaf(v1, v1, v0 * v0)));
        const float inv = 1.0f / fmaxf(nrm, kEpsNormalize);
        if (c0 < TR_MAXD) sR[c0] = c0 < dr ? v0 * inv : 0.f;
        if (c1 < TR_MAXD) sR[c1] = c1 < dr ? v1 * inv : 0.f;
        if (lane == 0) { sRs[0] = inv; sRs[1] = nrm > kEpsNormalize ? 1.f : 0.f; }
    }
    __syncthreads();

    const float* M = A.mat + (int64_t)rel * de * dr;
    const int nte = (de + 31) / 32, ntr = (dr + 31) / 32;

    // ---- 2. HP = H^ M, TP = T^ M: one 32x32 output tile per work unit
    for (int u = wave; u < 2 * ntr; u += 4) {
        const int which = u / ntr, ct = u - which * ntr;
        const float* sX = which ? sT : sH;
        float* sXP = which ? sTP : sHP;
        const int j = ct * 32 + li;
        f32x16 acc = {0};
        for (int k0 = 0; k0 < de; k0 += 16) {  // fixed-trip inner loop: 8 operand loads in flight
            float av[8], bv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int k = k0 + 2 * q + lk;
                av[q] = k < de ? sX[li * Se + k] : 0.f;
                bv[q] = (k < de && j < dr) ? M[(int64_t)k * dr + j] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc, 0, 0, 0);
        }
        if (j < dr) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) sXP[mfma_row(reg, lk) * Sr + j] = acc[reg];
        }
    }
    __syncthreads();

    // ---- 3. the TransE tail per triple (pairwise.py:459-470), forward and -- MODE 1 -- backward in one pass
    float accR0 = 0.f, accR1 = 0.f;
    const bool l1 = A.l1 != 0;
    for (int i = wave; i < cnt; i += 4) {  // padding rows keep HP = TP = 0, i.e. GA = GC = 0 for step 4
        const float a0 = c0 < dr ? sHP[i * Sr + c0] : 0.f, a1 = c1 < dr ? sHP[i * Sr + c1] : 0.f;
        const float b0 = sR[c0 < TR_MAXD ? c0 : 0], b1 = c1 < TR_MAXD ? sR[c1] : 0.f;
        const float q0 = c0 < dr ? sTP[i * Sr + c0] : 0.f, q1 = c1 < dr ? sTP[i * Sr + c1] : 0.f;
        float na = fmaf(a1, a1, a0 * a0), nb = fmaf(b1, b1, b0 * b0), nc = fmaf(q1, q1, q0 * q0);
        na = sqrtf(wave_sum(na)); nb = sqrtf(wave_sum(nb)); nc = sqrtf(wave_sum(nc));
        const float ia = 1.0f / fmaxf(na, kEpsNormalize), ib = 1.0f / fmaxf(nb, kEpsNormalize),
                    ic = 1.0f / fmaxf(nc, kEpsNormalize);
        const float u0 = a0 * ia + b0 * ib - q0 * ic, u1 = a1 * ia + b1 * ib - q1 * ic;
        float p = l1 ? fabsf(u0) + fabsf(u1) : fmaf(u1, u1, u0 * u0);
        p = wave_sum(p);
        const float s = l1 ? p : sqrtf(p);
        if constexpr (MODE == 0) {
            if (lane == 0) A.scores[sRow[i]] = s;
        } else {
            const float ds = sDs[i];
            const float invs = (!l1 && s > 0.f) ? ds / s : 0.f;
            const float g0 = l1 ? (u0 > 0.f ? ds : (u0 < 0.f ? -ds : 0.f)) : u0 * invs;
            const float g1 = l1 ? (u1 > 0.f ? ds : (u1 < 0.f ? -ds : 0.f)) : u1 * invs;
            float da = fmaf(a1, g1, a0 * g0), db = fmaf(b1, g1, b0 * g0), dc = fmaf(q1, g1, q0 * g0);
            da = wave_sum(da) * ia; db = wave_sum(db) * ib; dc = wave_sum(dc) * ic;
            const bool fa = na > kEpsNormalize, fb = nb > kEpsNormalize, fc = nc > kEpsNormalize;
            const float ga0 = fa ? (g0 - (a0 * ia) * da) * ia : g0 * ia, ga1 = fa ? (g1 - (a1 * ia) * da) * ia : g1 * ia;
            const float gb0 = fb ? (g0 - (b0 * ib) * db) * ib : g0 * ib, gb1 = fb ? (g1 - (b1 * ib) * db) * ib : g1 * ib;
            const float gc0 = -(fc ? (g0 - (q0 * ic) * dc) * ic : g0 * ic), gc1 = -(fc ? (g1 - (q1 * ic) * dc) * ic : g1 * ic);
            if (c0 < dr) { sHP[i * Sr + c0] = ga0; sTP[i * Sr + c0] = gc0; }
            if (c1 < dr) { sHP[i * Sr + c1] = ga1; sTP[i * Sr + c1] = gc1; }
            accR0 += gb0; accR1 += gb1;
        }
    }
    if constexpr (MODE == 0) return;
    if (c0 < dr) atomicAdd(&sGR[c0], accR0);
    if (c1 < dr) atomicAdd(&sGR[c1], accR1);
    __syncthreads();

    // ---- 4. G_M = H^^T GA + T^^T GC (atomics into grad M_r) and GH = GA M^T, GT = GC M^T (into LDS)
    const int nab = nte * ntr;
    float* gM = A.g_mat + (int64_t)rel * de * dr;
    for (int u = wave; u < nab + 2 * nte; u += 4) {
        if (u < nab) {
            const int at = u / ntr, bt = u - at * ntr;
            const int a_in = at * 32 + li, b_in = bt * 32 + li;
            f32x16 acc = {0};
#pragma unroll 4
            for (int kk = 0; kk < TILE; kk += 2) {
                const int i = kk + lk;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_in < de ? sH[i * Se + a_in] : 0.f,
                                                           b_in < dr ? sHP[i * Sr + b_in] : 0.f, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_in < de ? sT[i * Se + a_in] : 0.f,
                                                           b_in < dr ? sTP[i * Sr + b_in] : 0.f, acc, 0, 0, 0);
            }
            if (b_in < dr) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int a = at * 32 + mfma_row(reg, lk);
                    if (a < de && acc[reg] != 0.f) unsafeAtomicAdd(gM + (int64_t)a * dr + b_in, acc[reg]);
                }
            }
        } else {
            const int v = u - nab;
            const int which = v / nte, at = v - which * nte;
            const float* sG = which ? sTP : sHP;
            float* sOut = which ? sGT : sGH;
            const int a = at * 32 + li;
            f32x16 acc = {0};
            for (int k0 = 0; k0 < dr; k0 += 16) {
                float av[8], bv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int b = k0 + 2 * q + lk;
                    av[q] = b < dr ? sG[li * Sr + b] : 0.f;
                    bv[q] = (a < de && b < dr) ? M[(int64_t)a * dr + b] : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc, 0, 0, 0);
            }
            if (a < de) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) sOut[mfma_row(reg, lk) * Se + a] = acc[reg];
            }
        }
    }
    __syncthreads();

    // ---- 5. back through the entity / relation normalisations, one coalesced row scatter each
    for (int i = wave; i < cnt; i += 4) {
        if (sDs[i] == 0.f) continue;  // wave-uniform
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const float* xh = (side ? sT : sH) + i * Se;
            const float* gx = (side ? sGT : sGH) + i * Se;
            const float x0 = c0 < de ? xh[c0] : 0.f, x1 = c1 < de ? xh[c1] : 0.f;
            const float g0 = c0 < de ? gx[c0] : 0.f, g1 = c1 < de ? gx[c1] : 0.f;
            const float dot = wave_sum(fmaf(x1, g1, x0 * g0));
            const float inv = (side ? sIt : sIh)[i];
            const bool f = (side ? sFt : sFh)[i] != 0.f;
            float* dst = A.g_ent + (side ? sTid[i] : sHid[i]) * (int64_t)de;
            if (c0 < de) unsafeAtomicAdd(dst + c0, f ? (g0 - x0 * dot) * inv : g0 * inv);
            if (c1 < de) unsafeAtomicAdd(dst + c1, f ? (g1 - x1 * dot) * inv : g1 * inv);
        }
    }
    if (wave == 0) {
        const float x0 = c0 < dr ? sR[c0] : 0.f, x1 = c1 < dr ? sR[c1] : 0.f;
        const float g0 = c0 < dr ? sGR[c0] : 0.f, g1 = c1 < dr ? sGR[c1] : 0.f;
        const float dot = wave_sum(fmaf(x1, g1, x0 * g0));
        const float inv = sRs[0];
        const bool f = sRs[1] != 0.f;
        float* dst = A.g_rel + (int64_t)rel * dr;
        if (c0 < dr) unsafeAtomicAdd(dst + c0, f ? (g0 - x0 * dot) * inv : g0 * inv);
        if (c1 < dr) unsafeAtomicAdd(dst + c1, f ? (g1 - x1 * dot) * inv : g1 * inv);
    }
}

static size_t transr_lds_bytes(int mode, int de, int dr) {
    const int Se = (de + 1) | 1, Sr = (dr + 1) | 1;
    size_t f = (size_t)2 * TILE * Se + 2 * TILE * Sr + (mode == 1 ? 2 * TILE * Se : 0) + 2 * TR_MAXD + 5 * TILE + 2;
    return f * sizeof(float) + (size_t)TILE * sizeof(int) + (size_t)2 * TILE * sizeof(long long) + 16;
}

// per side (the pairwise step gets two of these): the grouping of the tile kernels, or half of what the two-launch large-batch
// step keeps between its kernels (kge_transr_rows.hip), whichever is larger
size_t transr_workspace_bytes(const kge_model_desc* m, int64_t n) {
    const size_t a = group_ws_bytes(m->tot_relation, n), b = (transr_rows_ws_bytes(m, n) + 1) / 2;
    return a > b ? a : b;
}

static int transr_check(const kge_model_desc* m, int64_t n) {
    if (m->dim > TR_MAXD || m->rel_dim > TR_MAXD || m->rel_dim <= 0) {
        set_error("TransR: hidden sizes %d/%d exceed the LDS-resident tile kernel (max %d)", m->dim, m->rel_dim, TR_MAXD);
        return -1;
    }
    if (n >= (1ll << 31)) { set_error("TransR: batch too large"); return -1; }
    return 0;
}

static int transr_run(int mode, const kge_model_desc* m, IdSplit h, IdSplit r, IdSplit t, int64_t n,
                      const float* dscore, float* scores, void* ws, size_t ws_bytes, bool grouped, hipStream_t s) {
    if (transr_check(m, n)) return -1;
    const int64_t R = m->tot_relation;
    if (!ws || ws_bytes < group_ws_bytes(R, n)) {
        set_error("TransR needs a workspace of %zu bytes (kge_workspace_bytes)", group_ws_bytes(R, n));
        return -1;
    }
    const GroupWs g = carve_group_ws(ws, R, n);
    if (!grouped) {  // the fused train step's backward reuses the grouping its forward left in this workspace
        int rc = group_by_relation_split(r, n, R, g, s);
        if (rc) return rc;
    }
    TransRArgs a;
    a.ent = m->tables[0]; a.rel = m->tables[1]; a.mat = m->tables[2];
    a.g_ent = m->grads[0]; a.g_rel = m->grads[1]; a.g_mat = m->grads[2];
    a.h = h; a.t = t; a.offsets = g.offsets; a.tile_off = g.tile_off; a.tile_rel = g.tile_rel; a.perm = g.perm;
    a.R = (int)R; a.de = m->dim; a.dr = m->rel_dim; a.l1 = (m->flags & KGE_FLAG_L1) ? 1 : 0;
    a.dscore = dscore; a.scores = scores;
    const unsigned max_tiles = (unsigned)group_max_tiles(R, n);  // surplus blocks exit
    const size_t lds = transr_lds_bytes(mode, m->dim, m->rel_dim);
    if (mode == 0) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k_transr<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_transr<0>, dim3(max_tiles), dim3(256), lds, s, a);
    } else {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k_transr<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_transr<1>, dim3(max_tiles), dim3(256), lds, s, a);
    }
    return check_launch("k_transr");
}

int launch_transr_forward(const kge_model_desc* m, const int64_t* h, const int64_t* r, const int64_t* t, int64_t n,
                          float* scores, void* ws, size_t ws_bytes, hipStream_t s) {
    return transr_run(0, m, id_whole(h, n), id_whole(r, n), id_whole(t, n), n, nullptr, scores, ws, ws_bytes, false, s);
}
int launch_transr_backward(const kge_model_desc* m, const int64_t* h, const int64_t* r, const int64_t* t, int64_t n,
                           const float* dscore, void* ws, size_t ws_bytes, bool grouped, hipStream_t s) {
    return transr_run(1, m, id_whole(h, n), id_whole(r, n), id_whole(t, n), n, dscore, nullptr, ws, ws_bytes, grouped, s);
}

// positives and negatives of the fused pairwise step as ONE grouped batch of 2n triples (see launch_rescal_pair_forward)
int launch_transr_pair_forward(const kge_model_desc* m, const int64_t* ph, const int64_t* pr, const int64_t* pt,
                               const int64_t* nh, const int64_t* nr, const int64_t* nt, int64_t n, float* scores2, void* ws,
                               size_t ws_bytes, hipStream_t s) {
    return transr_run(0, m, IdSplit{ph, nh, n}, IdSplit{pr, nr, n}, IdSplit{pt, nt, n}, 2 * n, nullptr, scores2, ws, ws_bytes, false, s);
}
int launch_transr_pair_backward(const kge_model_desc* m, const int64_t* ph, const int64_t* pr, const int64_t* pt,
                                const int64_t* nh, const int64_t* nr, const int64_t* nt, int64_t n, const float* dscore2,
                                void* ws, size_t ws_bytes, hipStream_t s) {
    return transr_run(1, m, IdSplit{ph, nh, n}, IdSplit{pr, nr, n}, IdSplit{pt, nt, n}, 2 * n, dscore2, nullptr, ws, ws_bytes, true, s);
}

// ------------------------------------------------------------------ evaluation helpers (one relation per call)
// candidate tile -> sweep layout cand[tile][k][64] with cand[e][:] = n(n(ent[e]) M_r), r = triples[1]
__global__ __launch_bounds__(256) void k_transr_project(const float* __restrict__ ent, const float* __restrict__ mat,
                                                        const int64_t* __restrict__ triples,
                                                        const int64_t* __restrict__ group_rel, int64_t table_stride,
                                                        int64_t E, int de, int dr, int Kpad, float* __restrict__ cand) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Se = (de + 1) | 1, Sr = (dr + 1) | 1;
    float* sE = smem;            // [64][Se] normalised entity rows
    float* sP = sE + 64 * Se;    // [64][Sr] projected rows
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tile = blockIdx.x, e0 = tile * 64;
    const int64_t rel = group_rel ? group_rel[blockIdx.y] : triples[1];  // blockIdx.y = relation group = candidate table
    cand += blockIdx.y * table_stride;
    const float* M = mat + rel * (int64_t)de * dr;
    const int c0 = lane, c1 = lane + 64;
    {   // the wave's 16 rows are gathered before the first reduction (one memory round trip, not sixteen)
        float v0[16], v1[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int64_t e = e0 + wave * 16 + j;
            v0[j] = (e < E && c0 < de) ? ent[e * de + c0] : 0.f;
            v1[j] = (e < E && c1 < de) ? ent[e * de + c1] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int row = wave * 16 + j;
            const float inv = 1.0f / fmaxf(sqrtf(wave_sum(fmaf(v1[j], v1[j], v0[j] * v0[j]))), kEpsNormalize);
            if (c0 < de) sE[row * Se + c0] = v0[j] * inv;
            if (c1 < de) sE[row * Se + c1] = v1[j] * inv;
        }
    }
    __syncthreads();
    // P[64, dr] = E^[64, de] M_r on the matrix cores: 32x32 output tiles dealt to the four waves
    {
        const int li = lane & 31, lk = lane >> 5;
        const int nct = (dr + 31) / 32;
        for (int u = wave; u < 2 * nct; u += 4) {
            const int rt = u / nct, ct = u - rt * nct;
            const int j = ct * 32 + li;
            f32x16 acc = {0};
            for (int k0 = 0; k0 < de; k0 += 16) {
                float av[8], bv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int k = k0 + 2 * q + lk;
                    av[q] = k < de ? sE[(rt * 32 + li) * Se + k] : 0.f;
                    bv[q] = (k < de && j < dr) ? M[(int64_t)k * dr + j] : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc, 0, 0, 0);
            }
            if (j < dr) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) sP[(rt * 32 + mfma_row(reg, lk)) * Sr + j] = acc[reg];
            }
        }
    }
    __syncthreads();
    float n2 = 0.f;
    for (int j = 0; j < dr; ++j) n2 = fmaf(sP[lane * Sr + j], sP[lane * Sr + j], n2);
    const float inv = 1.0f / fmaxf(sqrtf(n2), kEpsNormalize);
    const bool valid = e0 + lane < E;
    for (int j = wave; j < Kpad; j += 4)
        cand[(tile * Kpad + j) * 64 + lane] = (valid && j < dr) ? sP[lane * Sr + j] * inv : 0.f;
}

// one wave per test triple: qt = n(h^ M) + n(r^), qh = n(t^ M) - n(r^)   (score = || q - candidate ||)
__global__ __launch_bounds__(256) void k_transr_queries(const float* __restrict__ ent, const float* __restrict__ relt,
                                                        const float* __restrict__ mat, const int64_t* __restrict__ triples,
                                                        int64_t n, int de, int dr, int Kpad, float* __restrict__ qvec,
                                                        float* __restrict__ qscale) {
    __shared__ float s_row[4][2][TR_MAXD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    if (i >= n) return;
    const int64_t h = triples[3 * i], r = triples[3 * i + 1], t = triples[3 * i + 2];
    const int c0 = lane, c1 = lane + 64;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const float* row = ent + (side ? t : h) * (int64_t)de;
        const float v0 = c0 < de ? row[c0] : 0.f, v1 = c1 < de ? row[c1] : 0.f;
        const float inv = 1.0f / fmaxf(sqrtf(wave_sum(fmaf(v1, v1, v0 * v0))), kEpsNormalize);
        if (c0 < de) s_row[wave][side][c0] = v0 * inv;
        if (c1 < de) s_row[wave][side][c1] = v1 * inv;
    }
    __builtin_amdgcn_wave_barrier();
    const float* M = mat + r * (int64_t)de * dr;
    float hp0 = 0.f, hp1 = 0.f, tp0 = 0.f, tp1 = 0.f;
    for (int a = 0; a < de; ++a) {  // lanes = output columns: M rows are read coalesced
        const float m0 = c0 < dr ? M[(int64_t)a * dr + c0] : 0.f, m1 = c1 < dr ? M[(int64_t)a * dr + c1] : 0.f;
        const float xh = s_row[wave][0][a], xt = s_row[wave][1][a];
        hp0 = fmaf(xh, m0, hp0); hp1 = fmaf(xh, m1, hp1);
        tp0 = fmaf(xt, m0, tp0); tp1 = fmaf(xt, m1, tp1);
    }
    const float* rr = relt + r * (int64_t)dr;
    float r0 = c0 < dr ? rr[c0] : 0.f, r1 = c1 < dr ? rr[c1] : 0.f;
    const float ir = 1.0f / fmaxf(sqrtf(wave_sum(fmaf(r1, r1, r0 * r0))), kEpsNormalize);
    r0 *= ir; r1 *= ir;  // r^ (embed), normalised once more by forward (pairwise.py:463)
    const float ia = 1.0f / fmaxf(sqrtf(wave_sum(fmaf(hp1, hp1, hp0 * hp0))), kEpsNormalize);
    const float ib = 1.0f / fmaxf(sqrtf(wave_sum(fmaf(r1, r1, r0 * r0))), kEpsNormalize);
    const float ic = 1.0f / fmaxf(sqrtf(wave_sum(fmaf(tp1, tp1, tp0 * tp0))), kEpsNormalize);
    float* qt = qvec + (2 * i) * (int64_t)Kpad;
    float* qh = qvec + (2 * i + 1) * (int64_t)Kpad;
    for (int k = lane; k < Kpad; k += 64) {
        const bool in = k < dr;
        const float hp = k < 64 ? hp0 : hp1, tp = k < 64 ? tp0 : tp1, rv = k < 64 ? r0 : r1;
        qt[k] = in ? hp * ia + rv * ib : 0.f;
        qh[k] = in ? tp * ic - rv * ib : 0.f;
    }
    if (lane == 0) { qscale[2 * i] = 1.0f; qscale[2 * i + 1] = 1.0f; }
}

int launch_transr_eval_prepare(const kge_model_desc* m, const int64_t* triples, int64_t n, const int64_t* group_rel,
                               int64_t n_groups, int Kpad, int64_t ntiles, float* cand, float* qvec, float* qscale,
                               hipStream_t s) {
    if (transr_check(m, n)) return -1;
    const int de = m->dim, dr = m->rel_dim;
    const size_t lds = (size_t)64 * (((de + 1) | 1) + ((dr + 1) | 1)) * sizeof(float);
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k_transr_project, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (n_groups < 1 || n_groups > 65535) { set_error("TransR evaluation: %lld relation groups per call (1..65535)", (long long)n_groups); return -1; }
    hipLaunchKernelGGL(k_transr_project, dim3((unsigned)ntiles, (unsigned)n_groups), dim3(256), lds, s, m->tables[0],
                       m->tables[2], triples, group_rel, ntiles * (int64_t)Kpad * 64, m->tot_entity, de, dr, Kpad, cand);
    hipLaunchKernelGGL(k_transr_queries, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, m->tables[0], m->tables[1],
                       m->tables[2], triples, n, de, dr, Kpad, qvec, qscale);
    return check_launch("k_transr_project / k_transr_queries");
}

}  // namespace kge
