// kge_pullx.hip -- TransH / TransD training gradients without float atomics, in the two-launch owner-computes form of
// kge_pull.hip (k_pull_eval + k_pull_step<DIR>):
//     Generator (data/generator.py:42-97) + Trainer.train_step_pairwise (utils/trainer.py:147-157) + Criterion.pairwise_hinge
//     (utils/criterion.py:25-29) + loss.backward() for TransH (models/pairwise.py:143-182) and TransD (:229-278);
//     the dense optimiser then runs as kge_optimizer_step over the flat buffers, exactly as after the atomic kernels.
//
//   phase 1  k_transx_eval   one lane group per (positive, negative) pair: the forward and the hand-derived backward of
//            kge_score.hip::k_transx_pair_sampled, once per pair, but every gradient row the pair produces is STORED to the
//            pair's own slots of a staging buffer (plain coalesced stores) instead of being scattered with atomics:
//              slot 0 / 1 / 2   gradient wrt the entity rows H / T / C (through normalisation and projection)
//              slot 3           gradient wrt the NORMALISED relation row r^  (the normalisation backward is linear: applied once
//              slot 4           TransH: gradient wrt w^ ; TransD: gradient wrt r_m            per row by its owner)
//              slot 5 / 6 / 7   TransD: gradient wrt the mapping rows h_m / t_m / c_m
//            plus one float per pair: the hinge coefficient (0 = inside the margin: nothing stored, nothing read).
//   phase 2  k_transx_own    one owner group per parameter row with incidences (the incidence index, bucket lists and ride-along
//            sampler of kge_pull.hip): sums the staged rows of its incidences in ascending (role, pair) order -- each staged row
//            is read exactly once, by its one owner --, relation owners finish with the normalisation backward of r (and w), and
//            the row is written ONCE into the dense gradient table.  Rows cut into several work items combine through LDS / a
//            finishing launch as in k_pull_step.  Deterministic: no float atomics anywhere.
#include "kge_row_kernels.h"
#include "kge_sampler_device.h"
#include "kge_pull_device.h"

namespace kge {

struct XArgs {
    const float* ent; const float* rel; const float* p3; const float* entmap;   // TransH: p3 = w.  TransD: p3 = rel mappings, entmap = ent mappings
    float* g_ent; float* g_rel; float* g_p3; float* g_entmap;
    const int4* pairs;
    PullLists lists;
    const int4* items; const int32_t* inc; float* partials; const int4* multi;
    int64_t n_items, n_multi, n_pairs;
    const uint32_t* listed;    // bitmap of the rows `items` lists (compact index); entities that were only DRAWN are owned by the
                               // first pair that drew them (pc bit 27), enumerated after the items
    int sample_blocks;
    int E, d, l1, reset_lists;
    float margin;
    float* stage;              // [n_pairs][NS][4 * G * NV]
    float* recs;               // [n_pairs] hinge coefficient
};

template <int M> constexpr int xslots() { return M == KGE_TRANSH ? 5 : 8; }

// lane gl of a G-lane group holds elements 4 * (v * G + gl) .. + 3, v < NV, as NE = 4 * NV floats
template <int G, int NV>
__device__ __forceinline__ void xload(float (&x)[4 * NV], const float* __restrict__ row, int nvec, int gl) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int i = v * G + gl;
        const float4 q = i < nvec ? reinterpret_cast<const float4*>(row)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        x[4 * v] = q.x; x[4 * v + 1] = q.y; x[4 * v + 2] = q.z; x[4 * v + 3] = q.w;
    }
}
template <int G, int NV>
__device__ __forceinline__ void xstore(float* __restrict__ row, const float (&x)[4 * NV], int nvec, int gl) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int i = v * G + gl;
        if (i < nvec) reinterpret_cast<float4*>(row)[i] = make_float4(x[4 * v], x[4 * v + 1], x[4 * v + 2], x[4 * v + 3]);
    }
}

// ------------------------------------------------------------------ phase 1
template <int M, int G, int NV>
__global__ __launch_bounds__(kBlock) void k_transx_eval(XArgs a, float* __restrict__ loss) {
    static_assert(M == KGE_TRANSH || M == KGE_TRANSD, "TransH / TransD");
    constexpr int GPB = kBlock / G, NE = 4 * NV, NS = xslots<M>();
    constexpr int RS = 4 * G * NV;
    const int gl = threadIdx.x % G;
    const int d = a.d, nvec = a.d >> 2;
    const bool l1 = a.l1 != 0;
    const int64_t i = (int64_t)blockIdx.x * GPB + threadIdx.x / G;
    float acc = 0.f;
    if (i < a.n_pairs) {
        const int4 p = a.pairs[i];
        const int w = a.lists.pc[i];
        const bool tail = ((w >> 24) & 1) != 0;
        const int64_t h = p.x, r = p.y, t = p.z, c = w & 0xFFFFFF;
        float H[NE], T[NE], C[NE], HM[NE], TM[NE], CM[NE], R[NE], P[NE];
        xload<G, NV>(H, a.ent + h * d, nvec, gl);
        xload<G, NV>(T, a.ent + t * d, nvec, gl);
        xload<G, NV>(C, a.ent + c * d, nvec, gl);
        if constexpr (M == KGE_TRANSD) {
            xload<G, NV>(HM, a.entmap + h * d, nvec, gl);
            xload<G, NV>(TM, a.entmap + t * d, nvec, gl);
            xload<G, NV>(CM, a.entmap + c * d, nvec, gl);
        }
        xload<G, NV>(R, a.rel + r * d, nvec, gl);
        xload<G, NV>(P, a.p3 + r * d, nvec, gl);
        // (from here on: the arithmetic of k_transx_pair_sampled for one pair, in its order)
        float nR = 0.f, nW = 0.f;
#pragma unroll
        for (int k = 0; k < NE; ++k) { nR = fmaf(R[k], R[k], nR); nW = fmaf(P[k], P[k], nW); }
        gsum2<G>(nR, nW);
        nR = sqrtf(nR);
        const float iR = 1.0f / fmaxf(nR, kEpsNormalize);
        if constexpr (M == KGE_TRANSH) {
            nW = sqrtf(nW);
            const float iW = 1.0f / fmaxf(nW, kEpsNormalize);
#pragma unroll
            for (int k = 0; k < NE; ++k) P[k] *= iW;  // P = w^
        }
        float pH = 0.f, pT = 0.f, pC = 0.f;
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            if constexpr (M == KGE_TRANSH) { pH = fmaf(H[k], P[k], pH); pT = fmaf(T[k], P[k], pT); pC = fmaf(C[k], P[k], pC); }
            else { pH = fmaf(H[k], HM[k], pH); pT = fmaf(T[k], TM[k], pT); pC = fmaf(C[k], CM[k], pC); }
        }
        gsum3<G>(pH, pT, pC);
        float aH[NE], aT[NE], aC[NE];
        float nH = 0.f, nT = 0.f, nC = 0.f;
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            if constexpr (M == KGE_TRANSH) { aH[k] = H[k] - pH * P[k]; aT[k] = T[k] - pT * P[k]; aC[k] = C[k] - pC * P[k]; }
            else { aH[k] = H[k] + pH * P[k]; aT[k] = T[k] + pT * P[k]; aC[k] = C[k] + pC * P[k]; }
            nH = fmaf(aH[k], aH[k], nH); nT = fmaf(aT[k], aT[k], nT); nC = fmaf(aC[k], aC[k], nC);
        }
        gsum3<G>(nH, nT, nC);
        nH = sqrtf(nH); nT = sqrtf(nT); nC = sqrtf(nC);
        const bool fH = nH > kEpsNormalize, fT = nT > kEpsNormalize, fC = nC > kEpsNormalize;
        const float iH = 1.0f / fmaxf(nH, kEpsNormalize), iT = 1.0f / fmaxf(nT, kEpsNormalize);
        const float iC = 1.0f / fmaxf(nC, kEpsNormalize);
        float up[NE], un[NE];
        float sp = 0.f, sn = 0.f;
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const float hh = aH[k] * iH, rr = R[k] * iR, tt = aT[k] * iT, cc = aC[k] * iC;
            up[k] = hh + rr - tt;
            un[k] = tail ? (hh + rr - cc) : (cc + rr - tt);
            sp = l1 ? sp + fabsf(up[k]) : fmaf(up[k], up[k], sp);
            sn = l1 ? sn + fabsf(un[k]) : fmaf(un[k], un[k], sn);
        }
        gsum2<G>(sp, sn);
        if (!l1) { sp = sqrtf(sp); sn = sqrtf(sn); }
        const float v = sp + a.margin - sn;
        acc = fmaxf(v, 0.f);
        const float coef = v > 0.f ? 1.f : (v == 0.f ? 0.5f : 0.f);
        if (gl == 0) a.recs[i] = coef;
        if (coef != 0.f) {
            const float ip = (!l1 && sp > 0.f) ? coef / sp : 0.f, in = (!l1 && sn > 0.f) ? -coef / sn : 0.f;
            float gH[NE], gT[NE], gC[NE], gRh[NE], gP[NE];
            float dH = 0.f, dT = 0.f, dC = 0.f;
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                const float gp = l1 ? (up[k] > 0.f ? coef : (up[k] < 0.f ? -coef : 0.f)) : up[k] * ip;
                const float gn = l1 ? (un[k] > 0.f ? -coef : (un[k] < 0.f ? coef : 0.f)) : un[k] * in;
                gRh[k] = gp + gn;
                gH[k] = tail ? gp + gn : gp;
                gT[k] = tail ? -gp : -(gp + gn);
                gC[k] = tail ? -gn : gn;
                dH = fmaf(aH[k], gH[k], dH); dT = fmaf(aT[k], gT[k], dT); dC = fmaf(aC[k], gC[k], dC);
            }
            gsum3<G>(dH, dT, dC);
            dH *= iH; dT *= iT; dC *= iC;
            float qH = 0.f, qT = 0.f, qC = 0.f;
#pragma unroll
            for (int k = 0; k < NE; ++k) {  // gradient wrt a_X, and its dot with the projection direction
                gH[k] = fH ? (gH[k] - (aH[k] * iH) * dH) * iH : gH[k] * iH;
                gT[k] = fT ? (gT[k] - (aT[k] * iT) * dT) * iT : gT[k] * iT;
                gC[k] = fC ? (gC[k] - (aC[k] * iC) * dC) * iC : gC[k] * iC;
                qH = fmaf(gH[k], P[k], qH); qT = fmaf(gT[k], P[k], qT); qC = fmaf(gC[k], P[k], qC);
            }
            gsum3<G>(qH, qT, qC);
            float* st = a.stage + i * (int64_t)(NS * RS);
            if constexpr (M == KGE_TRANSH) {
#pragma unroll
                for (int k = 0; k < NE; ++k) {
                    gP[k] = -(pH * gH[k] + qH * H[k] + pT * gT[k] + qT * T[k] + pC * gC[k] + qC * C[k]);
                    gH[k] -= qH * P[k]; gT[k] -= qT * P[k]; gC[k] -= qC * P[k];
                }
            } else {
                float gHM[NE], gTM[NE], gCM[NE];
#pragma unroll
                for (int k = 0; k < NE; ++k) {
                    gP[k] = pH * gH[k] + pT * gT[k] + pC * gC[k];
                    gHM[k] = qH * H[k]; gTM[k] = qT * T[k]; gCM[k] = qC * C[k];
                    gH[k] += qH * HM[k]; gT[k] += qT * TM[k]; gC[k] += qC * CM[k];
                }
                xstore<G, NV>(st + 5 * RS, gHM, G * NV, gl);
                xstore<G, NV>(st + 6 * RS, gTM, G * NV, gl);
                xstore<G, NV>(st + 7 * RS, gCM, G * NV, gl);
            }
            xstore<G, NV>(st + 0 * RS, gH, G * NV, gl);
            xstore<G, NV>(st + 1 * RS, gT, G * NV, gl);
            xstore<G, NV>(st + 2 * RS, gC, G * NV, gl);
            xstore<G, NV>(st + 3 * RS, gRh, G * NV, gl);
            xstore<G, NV>(st + 4 * RS, gP, G * NV, gl);
        }
    }
    block_accumulate_loss<G>(acc, gl, loss);
}

// ------------------------------------------------------------------ phase 2
// the row's two accumulated gradient rows -> the dense gradient tables.  Entity rows are final as staged; a relation owner
// applies the normalisation backward of r (and, TransH, of w) once to the sums (it is linear in them)
template <int M, int G, int NV>
__device__ __forceinline__ void xfinish(const XArgs& a, int g, const float (&A0)[4 * NV], const float (&A1)[4 * NV], int gl) {
    constexpr int NE = 4 * NV;
    const int d = a.d, nvec = a.d >> 2;
    if (g < a.E) {
        xstore<G, NV>(a.g_ent + (int64_t)g * d, A0, nvec, gl);
        if constexpr (M == KGE_TRANSD) xstore<G, NV>(a.g_entmap + (int64_t)g * d, A1, nvec, gl);
        return;
    }
    const int64_t rr = g - a.E;
    float R[NE], P[NE];
    xload<G, NV>(R, a.rel + rr * d, nvec, gl);
    xload<G, NV>(P, a.p3 + rr * d, nvec, gl);
    float nR = 0.f, nW = 0.f;
#pragma unroll
    for (int k = 0; k < NE; ++k) { nR = fmaf(R[k], R[k], nR); nW = fmaf(P[k], P[k], nW); }
    gsum2<G>(nR, nW);
    nR = sqrtf(nR);
    const bool fR = nR > kEpsNormalize;
    const float iR = 1.0f / fmaxf(nR, kEpsNormalize);
    bool fW = false;
    float iW = 1.f;
    if constexpr (M == KGE_TRANSH) {
        nW = sqrtf(nW);
        fW = nW > kEpsNormalize;
        iW = 1.0f / fmaxf(nW, kEpsNormalize);
#pragma unroll
        for (int k = 0; k < NE; ++k) P[k] *= iW;
    }
    float dR = 0.f, dW = 0.f;
#pragma unroll
    for (int k = 0; k < NE; ++k) { dR = fmaf(R[k], A0[k], dR); dW = fmaf(P[k], A1[k], dW); }
    gsum2<G>(dR, dW);
    dR *= iR;
    float gR[NE], gW[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        gR[k] = fR ? (A0[k] - (R[k] * iR) * dR) * iR : A0[k] * iR;
        if constexpr (M == KGE_TRANSH) gW[k] = fW ? (A1[k] - P[k] * dW) * iW : A1[k] * iW;  // through F.normalize(w)
        else gW[k] = A1[k];
    }
    xstore<G, NV>(a.g_rel + rr * d, gR, nvec, gl);
    xstore<G, NV>(a.g_p3 + rr * d, gW, nvec, gl);
}

template <int M, int G, int NV>
__global__ __launch_bounds__(kBlock) void k_transx_own(XArgs a, PullSampleArgs sa) {
    constexpr int GPB = kBlock / G, NE = 4 * NV, NS = xslots<M>();
    constexpr int RS = 4 * G * NV;
    if ((int)blockIdx.x < a.sample_blocks) {   // leading blocks: the sampler of the NEXT batch rides along (other list set)
        const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
        if (i < sa.n) pull_sample_one(sa, i);
        return;
    }
    const int gl = threadIdx.x % G;
    const int gbase = (threadIdx.x & 63) / G * G;
    const int64_t item = (int64_t)((int)blockIdx.x - a.sample_blocks) * GPB + threadIdx.x / G;
    __shared__ float4 s_part[GPB][2][NV * G];
    __shared__ int s_vis[GPB][G];
    int4 it = make_int4(-1, 0, 0, 0);
    if (item < a.n_items) it = a.items[item];
    else if (a.listed != nullptr && item - a.n_items < a.n_pairs) {   // pair j stands in as the owner of the entity it drew first
        const int w = a.lists.pc[item - a.n_items];
        const int c = w & 0xFFFFFF;
        if (((w >> kPcFirstBit) & 1) && !((a.listed[c >> 5] >> (c & 31)) & 1u)) it = make_int4(c, 0, 0, 0);
    }
    const int g = it.x;
    const int kind = it.w & 3;
    float A0[NE], A1[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) { A0[k] = 0.f; A1[k] = 0.f; }
    if (g >= 0) {
        const bool is_rel = g >= a.E;
        int cnt = 0;
        bool fast_c = true;
        const bool walks_c = !is_rel && (kind == 0 || kind == 1 || (kind == 3 && ((it.w >> 2) & 15) == 0));
        const int nvis = own_visit_list_dir<G>(a.lists, a.inc, it, g, walks_c, gl, gbase, s_vis[threadIdx.x / G], &cnt, &fast_c);
        const int* __restrict__ vis = s_vis[threadIdx.x / G];
        // one visit: the staged row(s) of (pair, role) added to the accumulators; nothing to do for a pair inside its margin
        auto rows_of = [&](int e, float4 (&r0)[NV], float4 (&r1)[NV], float& coef) {
            const int pair = e >> 2, role = e & 3;
            coef = a.recs[pair];
            const float4* st = reinterpret_cast<const float4*>(a.stage + pair * (int64_t)(NS * RS));
            const int s0 = role == kRoleR ? 3 : (role == kRoleC ? 2 : role);          // H 0, T 1, C 2, R 3
            const int s1 = role == kRoleR ? 4 : (M == KGE_TRANSD ? 5 + s0 : -1);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                r0[v] = st[s0 * (G * NV) + v * G + gl];
                r1[v] = s1 >= 0 ? st[s1 * (G * NV) + v * G + gl] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        auto add = [&](const float4 (&r0)[NV], const float4 (&r1)[NV], float coef) {
            if (coef == 0.f) return;     // (the slots of an inactive pair hold stale data)
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                A0[4 * v] += r0[v].x; A0[4 * v + 1] += r0[v].y; A0[4 * v + 2] += r0[v].z; A0[4 * v + 3] += r0[v].w;
                A1[4 * v] += r1[v].x; A1[4 * v + 1] += r1[v].y; A1[4 * v + 2] += r1[v].z; A1[4 * v + 3] += r1[v].w;
            }
        };
        constexpr int kBatch = 4;     // visits whose rows are requested before the first is added
        for (int v0 = 0; v0 < nvis; v0 += kBatch) {
            float4 r0[kBatch][NV], r1[kBatch][NV];
            float cf[kBatch];
#pragma unroll
            for (int q = 0; q < kBatch; ++q) {
                cf[q] = 0.f;
                if (v0 + q < nvis) rows_of(vis[v0 + q], r0[q], r1[q], cf[q]);
            }
#pragma unroll
            for (int q = 0; q < kBatch; ++q)
                if (v0 + q < nvis) add(r0[q], r1[q], cf[q]);
        }
        if (cnt > 0 && !fast_c) {   // more drawers than the bucket / lane group holds: pair-ordered walk over bucket + chain
            const int nb = cnt < kPullCap ? cnt : kPullCap;
            int last = -1;
            for (;;) {
                int best = 0x7FFFFFFF;
                for (int m = 0; m < nb; ++m) { const int j = a.lists.bucket[(int64_t)g * kPullCap + m]; if (j > last && j < best) best = j; }
                for (int j = a.lists.head[g]; j >= 0; j = a.lists.next[j]) if (j > last && j < best) best = j;
                if (best == 0x7FFFFFFF) break;
                float4 r0[NV], r1[NV];
                float cf;
                rows_of((best << 2) | kRoleC, r0, r1, cf);
                add(r0, r1, cf);
                last = best;
            }
        }
        if (cnt > 0 && a.reset_lists && gl == 0) {
            a.lists.count[g] = 0;
            if (cnt > kPullCap) a.lists.head[g] = -1;
        }
        if (kind == 0) {
            // (a row nobody touched this step keeps the zero gradient the optimiser left behind: nothing to write)
            if (nvis > 0 || cnt > 0) xfinish<M, G, NV>(a, g, A0, A1, gl);
        } else if (kind == 3) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                s_part[threadIdx.x / G][0][v * G + gl] = make_float4(A0[4 * v], A0[4 * v + 1], A0[4 * v + 2], A0[4 * v + 3]);
                s_part[threadIdx.x / G][1][v * G + gl] = make_float4(A1[4 * v], A1[4 * v + 1], A1[4 * v + 2], A1[4 * v + 3]);
            }
        } else {
            float4* out = reinterpret_cast<float4*>(a.partials) + (int64_t)(it.w >> 2) * (2 * G * NV);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                out[v * G + gl] = make_float4(A0[4 * v], A0[4 * v + 1], A0[4 * v + 2], A0[4 * v + 3]);
                out[G * NV + v * G + gl] = make_float4(A1[4 * v], A1[4 * v + 1], A1[4 * v + 2], A1[4 * v + 3]);
            }
        }
    }
    __syncthreads();
    if (g >= 0 && kind == 3 && ((it.w >> 2) & 15) == 0) {   // first item of a workgroup-local row: add the others' sums in segment order
        const int nseg = it.w >> 6, me = threadIdx.x / G;
        for (int m = 1; m < nseg; ++m) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const float4 p0 = s_part[me + m][0][v * G + gl], p1 = s_part[me + m][1][v * G + gl];
                A0[4 * v] += p0.x; A0[4 * v + 1] += p0.y; A0[4 * v + 2] += p0.z; A0[4 * v + 3] += p0.w;
                A1[4 * v] += p1.x; A1[4 * v + 1] += p1.y; A1[4 * v + 2] += p1.z; A1[4 * v + 3] += p1.w;
            }
        }
        xfinish<M, G, NV>(a, g, A0, A1, gl);
    }
}

// rows cut into several items across workgroups: add their partial sums in slot order, then finish the row
template <int M, int G, int NV>
__global__ __launch_bounds__(kBlock) void k_transx_finish(XArgs a) {
    constexpr int GPB = kBlock / G, NE = 4 * NV;
    const int gl = threadIdx.x % G;
    const int64_t m = (int64_t)blockIdx.x * GPB + threadIdx.x / G;
    if (m >= a.n_multi) return;
    const int4 row = a.multi[m];
    float A0[NE], A1[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) { A0[k] = 0.f; A1[k] = 0.f; }
    for (int s = 0; s < row.z; ++s) {
        const float4* in = reinterpret_cast<const float4*>(a.partials) + (int64_t)(row.y + s) * (2 * G * NV);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float4 p0 = in[v * G + gl], p1 = in[G * NV + v * G + gl];
            A0[4 * v] += p0.x; A0[4 * v + 1] += p0.y; A0[4 * v + 2] += p0.z; A0[4 * v + 3] += p0.w;
            A1[4 * v] += p1.x; A1[4 * v + 1] += p1.y; A1[4 * v + 2] += p1.z; A1[4 * v + 3] += p1.w;
        }
    }
    xfinish<M, G, NV>(a, row.x, A0, A1, gl);
}

// ------------------------------------------------------------------ host side
struct XGeo { int G, NV; };
static XGeo xgeo(int dim) {    // rows of up to 512 floats: 32-lane groups (the eval kernel holds ~20 rows of NV float4 per lane)
    XGeo g{0, 0};
    if (dim <= 0 || (dim & 3) || dim > 512) return g;
    g.G = 32;
    const int nvec = dim >> 2;
    g.NV = nvec <= 32 ? 1 : (nvec <= 64 ? 2 : 4);
    return g;
}

int transx_groups_per_block(int dim) { const XGeo g = xgeo(dim); return g.G ? kBlock / g.G : 0; }
int transx_partial_stride(int dim) { const XGeo g = xgeo(dim); return 2 * 4 * g.G * g.NV; }
void transx_scratch_bytes(int model, int dim, int64_t n, size_t* stage, size_t* recs) {
    const XGeo g = xgeo(dim);
    *stage = g.G ? (size_t)n * (model == KGE_TRANSH ? 5 : 8) * 4 * g.G * g.NV * sizeof(float) : 0;
    *recs = (size_t)n * sizeof(float);
}

template <int M, int G, int NV>
static int launch_transx_geo(XArgs& a, const PullSampleArgs& sa, float* loss, hipStream_t s) {
    constexpr int GPB = kBlock / G;
    hipLaunchKernelGGL((k_transx_eval<M, G, NV>), dim3((unsigned)((a.n_pairs + GPB - 1) / GPB)), dim3(kBlock), 0, s, a, loss);
    const int64_t units = a.n_items + (a.listed ? a.n_pairs : 0);
    a.sample_blocks = sa.n > 0 ? (int)((sa.n + kBlock - 1) / kBlock) : 0;
    hipLaunchKernelGGL((k_transx_own<M, G, NV>), dim3((unsigned)((units + GPB - 1) / GPB + a.sample_blocks)), dim3(kBlock), 0, s, a, sa);
    int rc = check_launch("k_transx_own");
    if (rc || a.n_multi == 0) return rc;
    hipLaunchKernelGGL((k_transx_finish<M, G, NV>), dim3((unsigned)((a.n_multi + GPB - 1) / GPB)), dim3(kBlock), 0, s, a);
    return check_launch("k_transx_finish");
}

int launch_transx_grad_step(const kge_model_desc* m, const int32_t* pairs, int64_t n_pairs, const kge_pull_lists* lists,
                            const int32_t* items, int64_t n_items, const uint32_t* listed, const int32_t* inc, float* partials,
                            const int32_t* multi, int64_t n_multi, float margin, float* stage, float* recs, int reset_lists,
                            const int32_t* next_pairs, const int32_t* next_inv, int64_t next_n, const float* bern,
                            const uint64_t* slots, int64_t n_slots, uint64_t seed, uint64_t next_offset,
                            const kge_pull_lists* next_lists, float* loss, hipStream_t s) {
    const XGeo geo = xgeo(m->dim);
    if (!geo.G) { set_error("kge_transx_grad_step: hidden size %d must be a multiple of 4 and at most 512", m->dim); return -1; }
    const bool th = m->model == KGE_TRANSH;
    XArgs a;
    a.ent = m->tables[0]; a.rel = m->tables[1];
    a.p3 = th ? m->tables[2] : m->tables[3];
    a.entmap = th ? nullptr : m->tables[2];
    a.g_ent = m->grads[0]; a.g_rel = m->grads[1];
    a.g_p3 = th ? m->grads[2] : m->grads[3];
    a.g_entmap = th ? nullptr : m->grads[2];
    a.pairs = (const int4*)pairs; a.lists = to_lists(lists);
    a.items = (const int4*)items; a.inc = inc; a.partials = partials; a.multi = (const int4*)multi;
    a.n_items = n_items; a.n_multi = n_multi; a.n_pairs = n_pairs;
    a.listed = listed; a.sample_blocks = 0;
    a.E = (int)m->tot_entity; a.d = m->dim; a.l1 = (m->flags & KGE_FLAG_L1) ? 1 : 0; a.reset_lists = reset_lists;
    a.margin = margin; a.stage = stage; a.recs = recs;
    PullSampleArgs sa = make_sample_args(next_pairs, next_inv, next_pairs && next_lists ? next_n : 0, m->tot_entity, bern, slots,
                                         n_slots, seed, next_offset, nullptr, next_lists);
    sa.no_desc = 1;
#define KGE_XG(NV_)                                                                                   \
    if (geo.NV == NV_)                                                                                \
        return th ? launch_transx_geo<KGE_TRANSH, 32, NV_>(a, sa, loss, s) : launch_transx_geo<KGE_TRANSD, 32, NV_>(a, sa, loss, s);
    KGE_XG(1) KGE_XG(2) KGE_XG(4)
#undef KGE_XG
    return -1;
}

}  // namespace kge
