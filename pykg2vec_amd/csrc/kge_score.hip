// kge_score.hip -- the shared-row, sampler-fused training kernels of the gather-type scorers (TransE / TransM,
// TransH / TransD, RotatE); the model-generic forward / backward / fused steps are in kge_row_kernels.h, instantiated
// in kge_score_generic.hip and kge_score_ext.hip.
//
// Reference sites replaced (paths relative to the reference tree):
//   forward           pykg2vec/models/pairwise.py:56-76,166-174,270-278,786-791 ; pointwise.py:97-104,185-188,444-446
//   backward          autograd of the above (utils/trainer.py:298), dense nn.Embedding grads
//   pairwise hinge    utils/trainer.py:147-157 + utils/criterion.py:25-29
//   pointwise         utils/trainer.py:176-180 + utils/criterion.py:31-34 + get_reg (pointwise.py:106-119,190-202,224-238,448-458)
//   self-adversarial  utils/criterion.py:13-23
//
// Roofline: HBM/L2-bound gather + atomic scatter (<= 3 flop/byte).  One G-lane group per triple keeps all of
// its rows in registers from the gather to the gradient scatter; the only HBM traffic is the row gather, the
// 24-byte id read and the float atomics into the dense gradient tables.
#include "kge_row_kernels.h"

namespace kge {

// ---- TransE, sampler fused, shared rows loaded ONCE.  A sampled negative is the positive with its head OR its tail
// replaced (data/generator.py:77-95), so a pair touches 4 distinct rows: H, R, T and the corrupting entity C -- not 6.
// 4 row gathers, 4 norms, 2 energies, 4 normalisation dot products (three batched butterflies), 4 scatters; the
// gradients wrt the normalised vectors add linearly before the normalisation backward, which is exactly the sum of
// the two per-triple backward passes the reference's autograd performs.
template <int G>
__device__ __forceinline__ void gsum4(float& a, float& b, float& c, float& d) {
    a = gsum<G>(a); b = gsum<G>(b); c = gsum<G>(c); d = gsum<G>(d);
}

// Batches arrive SORTED BY RELATION (generator: each batch slice of the permutation is sorted once at start-up -- a
// batch is a set, its order changes nothing but fp summation order), and a group walks CH consecutive pairs: the
// relation row, its norm and its gradient stay in registers across pairs of the same relation and are scattered once
// per run instead of once per pair.
// SCALED = TransM (pairwise.py:341-347): both energies are multiplied by the fixed per-relation weight theta_r, which
// travels with the relation row.
template <int G, int NCH, int CH, bool SCALED>
__global__ __launch_bounds__(kBlock) void k_transe_pair_sampled(DeviceModel m, int64_t n, float margin,
                                                                float* __restrict__ loss, FusedSampler fs) {
    constexpr int GPB = kBlock / G;
    const int gl = threadIdx.x % G;
    const int d = m.dim;
    const bool l1 = m.l1;
    float acc = 0.f;
    const int64_t s_start = fs.cursor ? fs.start + fs.cursor[0] : fs.start;
    const unsigned long long s_off = fs.cursor ? fs.offset + (unsigned long long)fs.cursor[1] : fs.offset;
    const int64_t nchunks = (n + CH - 1) / CH;
    for (int64_t ck = (int64_t)blockIdx.x * GPB + threadIdx.x / G; ck < nchunks; ck += (int64_t)gridDim.x * GPB) {
        int64_t r_cur = -1;
        float R[NCH], gRh[NCH];  // relation row and the running gradient wrt its NORMALISED form
        float iR = 0.f, theta = 1.f;
        bool fR = false, r_dirty = false;
        auto flush_r = [&]() {
            if (!r_dirty) return;
            float dR = 0.f;
#pragma unroll
            for (int k = 0; k < NCH; ++k) dR = fmaf(R[k], gRh[k], dR);
            dR = gsum<G>(dR) * iR;
            float gR[NCH];
#pragma unroll
            for (int k = 0; k < NCH; ++k) gR[k] = fR ? (gRh[k] - (R[k] * iR) * dR) * iR : gRh[k] * iR;
            atomic_add_row<G, NCH>(m.grad[1] + r_cur * (int64_t)d, gR, d, gl);
            r_dirty = false;
        };
        // sampling, lane-parallel: lane j of the group draws pair j of the chunk (perm -> triple -> Philox -> hash-set
        // probe: four dependent memory round trips and ~250 scalar-style instructions, done once for the CH pairs
        // side by side instead of redundantly in every lane, pair after pair); ids travel by shuffle
        const int64_t i0 = ck * CH;
        const int64_t i_end = min(n, i0 + CH);
        int my_h = 0, my_r = 0, my_t = 0, my_c = 0, my_tail = 0;
        if (gl < CH && i0 + gl < n) {
            const int64_t row = fs.perm[s_start + i0 + gl];
            const int64_t sh = fs.triples[3 * row], sr = fs.triples[3 * row + 1], st = fs.triples[3 * row + 2];
            int64_t nh, nt;
            corrupt_one(sh, sr, st, fs.E, fs.bern, fs.slots, fs.mask, fs.seed, s_off + (unsigned long long)(i0 + gl), nh, nt);
            my_h = (int)sh; my_r = (int)sr; my_t = (int)st;
            my_tail = nh == sh;
            my_c = (int)(my_tail ? nt : nh);
        }
        const int gbase = (threadIdx.x & 63) / G * G;  // first lane of this group inside its wave
        for (int64_t i = i0; i < i_end; ++i) {
            const int src = gbase + (int)(i - i0);
            const int64_t h = __shfl(my_h, src, 64), r = __shfl(my_r, src, 64), t = __shfl(my_t, src, 64);
            const int64_t c = __shfl(my_c, src, 64);
            const bool tail = __shfl(my_tail, src, 64) != 0;   // tail corrupted (else head corrupted)
            float H[NCH], T[NCH], C[NCH];
            load_row<G, NCH>(H, m.tab[0] + h * (int64_t)d, d, gl);
            load_row<G, NCH>(T, m.tab[0] + t * (int64_t)d, d, gl);
            load_row<G, NCH>(C, m.tab[0] + c * (int64_t)d, d, gl);
            float nH = 0.f, nR = 0.f, nT = 0.f, nC = 0.f;
            const bool new_r = r != r_cur;  // group-uniform
            if (new_r) {
                flush_r();
                r_cur = r;
                load_row<G, NCH>(R, m.tab[1] + r * (int64_t)d, d, gl);
                if constexpr (SCALED) theta = m.tab[2][r];
#pragma unroll
                for (int k = 0; k < NCH; ++k) { nR = fmaf(R[k], R[k], nR); gRh[k] = 0.f; }
            }
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                nH = fmaf(H[k], H[k], nH); nT = fmaf(T[k], T[k], nT); nC = fmaf(C[k], C[k], nC);
            }
            gsum4<G>(nH, nR, nT, nC);
            if (new_r) {
                nR = sqrtf(nR);
                fR = nR > kEpsNormalize;
                iR = 1.0f / fmaxf(nR, kEpsNormalize);
            }
            nH = sqrtf(nH); nT = sqrtf(nT); nC = sqrtf(nC);
            const bool fH = nH > kEpsNormalize, fT = nT > kEpsNormalize, fC = nC > kEpsNormalize;
            const float iH = 1.0f / fmaxf(nH, kEpsNormalize), iT = 1.0f / fmaxf(nT, kEpsNormalize);
            const float iC = 1.0f / fmaxf(nC, kEpsNormalize);
            float up[NCH], un[NCH];
            float sp = 0.f, sn = 0.f;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const float hh = H[k] * iH, rr = R[k] * iR, tt = T[k] * iT, cc = C[k] * iC;
                up[k] = hh + rr - tt;
                un[k] = tail ? (hh + rr - cc) : (cc + rr - tt);
                sp = l1 ? sp + fabsf(up[k]) : fmaf(up[k], up[k], sp);
                sn = l1 ? sn + fabsf(un[k]) : fmaf(un[k], un[k], sn);
            }
            gsum2<G>(sp, sn);
            if (!l1) { sp = sqrtf(sp); sn = sqrtf(sn); }
            const float v = SCALED ? theta * sp + margin - theta * sn : sp + margin - sn;
            acc += fmaxf(v, 0.f);
            const float c01 = v > 0.f ? 1.f : (v == 0.f ? 0.5f : 0.f);
            if (c01 == 0.f) continue;
            const float coef = SCALED ? c01 * theta : c01;  // d loss / d (unscaled distance)
            const float ip = (!l1 && sp > 0.f) ? coef / sp : 0.f, in = (!l1 && sn > 0.f) ? -coef / sn : 0.f;
            // gradients wrt the normalised vectors (gp: positive with ds=+coef, gn: negative with ds=-coef)
            float gH[NCH], gT[NCH], gC[NCH];
            float dH = 0.f, dT = 0.f, dC = 0.f;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const float gp = l1 ? (up[k] > 0.f ? coef : (up[k] < 0.f ? -coef : 0.f)) : up[k] * ip;
                const float gn = l1 ? (un[k] > 0.f ? -coef : (un[k] < 0.f ? coef : 0.f)) : un[k] * in;
                gRh[k] += gp + gn;
                gH[k] = tail ? gp + gn : gp;
                gT[k] = tail ? -gp : -(gp + gn);
                gC[k] = tail ? -gn : gn;
                dH = fmaf(H[k], gH[k], dH); dT = fmaf(T[k], gT[k], dT); dC = fmaf(C[k], gC[k], dC);
            }
            r_dirty = true;
            gsum3<G>(dH, dT, dC);
            dH *= iH; dT *= iT; dC *= iC;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                gH[k] = fH ? (gH[k] - (H[k] * iH) * dH) * iH : gH[k] * iH;
                gT[k] = fT ? (gT[k] - (T[k] * iT) * dT) * iT : gT[k] * iT;
                gC[k] = fC ? (gC[k] - (C[k] * iC) * dC) * iC : gC[k] * iC;
            }
            atomic_add_row<G, NCH>(m.grad[0] + h * (int64_t)d, gH, d, gl);
            atomic_add_row<G, NCH>(m.grad[0] + t * (int64_t)d, gT, d, gl);
            atomic_add_row<G, NCH>(m.grad[0] + c * (int64_t)d, gC, d, gl);
        }
        flush_r();
    }
    block_accumulate_loss<G>(acc, gl, loss);
}

// ---- TransH / TransD, sampler fused, shared rows loaded once (the TransE pair kernel plus the entity projections).
//   TransH (pairwise.py:143-182): a_X = X - (X.w^) w^          w^ = F.normalize(w_r)
//   TransD (pairwise.py:229-278): a_X = X + (X.X_m) r_m
// for X in {H, T, C}; the pair is then TransE on (a_H, r, a_T) / (a_H, r, a_C) or (a_C, r, a_T).  The relation-side rows
// (r, and w_r or r_m) and their gradients are carried across the pairs of a relation run and scattered once per run;
// each pair gathers 3 entity rows (TransD: + their 3 mapping rows) instead of 8 (12).
template <int M, int G, int NCH, int CH>
__global__ __launch_bounds__(kBlock) void k_transx_pair_sampled(DeviceModel m, int64_t n, float margin,
                                                                float* __restrict__ loss, FusedSampler fs) {
    static_assert(M == KGE_TRANSH || M == KGE_TRANSD, "TransH / TransD");
    constexpr int GPB = kBlock / G;
    const int gl = threadIdx.x % G;
    const int d = m.dim;
    const bool l1 = m.l1;
    float acc = 0.f;
    const int64_t s_start = fs.cursor ? fs.start + fs.cursor[0] : fs.start;
    const unsigned long long s_off = fs.cursor ? fs.offset + (unsigned long long)fs.cursor[1] : fs.offset;
    const int64_t nchunks = (n + CH - 1) / CH;
    const int gbase = (threadIdx.x & 63) / G * G;
    const float* entmap = M == KGE_TRANSD ? m.tab[2] : nullptr;
    float* g_entmap = M == KGE_TRANSD ? m.grad[2] : nullptr;
    for (int64_t ck = (int64_t)blockIdx.x * GPB + threadIdx.x / G; ck < nchunks; ck += (int64_t)gridDim.x * GPB) {
        int64_t r_cur = -1;
        float R[NCH], gRh[NCH];   // relation row, gradient wrt its normalised form
        float P[NCH], gP[NCH];    // TransH: w^ and gradient wrt w^ ; TransD: r_m and its gradient
        float iR = 0.f, iW = 0.f;
        bool fR = false, fW = false, r_dirty = false;
        auto flush_r = [&]() {
            if (!r_dirty) return;
            float dR = 0.f, dW = 0.f;
#pragma unroll
            for (int k = 0; k < NCH; ++k) { dR = fmaf(R[k], gRh[k], dR); dW = fmaf(P[k], gP[k], dW); }
            gsum2<G>(dR, dW);
            dR *= iR;
            float gR[NCH], gW[NCH];
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                gR[k] = fR ? (gRh[k] - (R[k] * iR) * dR) * iR : gRh[k] * iR;
                if constexpr (M == KGE_TRANSH) gW[k] = fW ? (gP[k] - P[k] * dW) * iW : gP[k] * iW;  // through F.normalize(w)
                else gW[k] = gP[k];
            }
            atomic_add_row<G, NCH>(m.grad[1] + r_cur * (int64_t)d, gR, d, gl);
            atomic_add_row<G, NCH>(m.grad[M == KGE_TRANSH ? 2 : 3] + r_cur * (int64_t)d, gW, d, gl);
            r_dirty = false;
        };
        const int64_t i0 = ck * CH;
        const int64_t i_end = min(n, i0 + CH);
        int my_h = 0, my_r = 0, my_t = 0, my_c = 0, my_tail = 0;
        if (gl < CH && i0 + gl < n) {  // lane-parallel sampling of the chunk's pairs
            const int64_t row = fs.perm[s_start + i0 + gl];
            const int64_t sh = fs.triples[3 * row], sr = fs.triples[3 * row + 1], st = fs.triples[3 * row + 2];
            int64_t nh, nt;
            corrupt_one(sh, sr, st, fs.E, fs.bern, fs.slots, fs.mask, fs.seed, s_off + (unsigned long long)(i0 + gl), nh, nt);
            my_h = (int)sh; my_r = (int)sr; my_t = (int)st;
            my_tail = nh == sh;
            my_c = (int)(my_tail ? nt : nh);
        }
        for (int64_t i = i0; i < i_end; ++i) {
            const int src = gbase + (int)(i - i0);
            const int64_t h = __shfl(my_h, src, 64), r = __shfl(my_r, src, 64), t = __shfl(my_t, src, 64);
            const int64_t c = __shfl(my_c, src, 64);
            const bool tail = __shfl(my_tail, src, 64) != 0;
            float H[NCH], T[NCH], C[NCH], HM[NCH], TM[NCH], CM[NCH];
            load_row<G, NCH>(H, m.tab[0] + h * (int64_t)d, d, gl);
            load_row<G, NCH>(T, m.tab[0] + t * (int64_t)d, d, gl);
            load_row<G, NCH>(C, m.tab[0] + c * (int64_t)d, d, gl);
            if constexpr (M == KGE_TRANSD) {
                load_row<G, NCH>(HM, entmap + h * (int64_t)d, d, gl);
                load_row<G, NCH>(TM, entmap + t * (int64_t)d, d, gl);
                load_row<G, NCH>(CM, entmap + c * (int64_t)d, d, gl);
            }
            const bool new_r = r != r_cur;  // group-uniform
            float nR = 0.f;
            if (new_r) {
                flush_r();
                r_cur = r;
                load_row<G, NCH>(R, m.tab[1] + r * (int64_t)d, d, gl);
                load_row<G, NCH>(P, m.tab[M == KGE_TRANSH ? 2 : 3] + r * (int64_t)d, d, gl);
                float nW = 0.f;
#pragma unroll
                for (int k = 0; k < NCH; ++k) { nR = fmaf(R[k], R[k], nR); nW = fmaf(P[k], P[k], nW); gRh[k] = 0.f; gP[k] = 0.f; }
                gsum2<G>(nR, nW);
                nR = sqrtf(nR);
                fR = nR > kEpsNormalize;
                iR = 1.0f / fmaxf(nR, kEpsNormalize);
                if constexpr (M == KGE_TRANSH) {
                    nW = sqrtf(nW);
                    fW = nW > kEpsNormalize;
                    iW = 1.0f / fmaxf(nW, kEpsNormalize);
#pragma unroll
                    for (int k = 0; k < NCH; ++k) P[k] *= iW;  // P = w^
                }
            }
            // projections
            float pH = 0.f, pT = 0.f, pC = 0.f;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                if constexpr (M == KGE_TRANSH) { pH = fmaf(H[k], P[k], pH); pT = fmaf(T[k], P[k], pT); pC = fmaf(C[k], P[k], pC); }
                else { pH = fmaf(H[k], HM[k], pH); pT = fmaf(T[k], TM[k], pT); pC = fmaf(C[k], CM[k], pC); }
            }
            gsum3<G>(pH, pT, pC);
            float aH[NCH], aT[NCH], aC[NCH];
            float nH = 0.f, nT = 0.f, nC = 0.f;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                if constexpr (M == KGE_TRANSH) { aH[k] = H[k] - pH * P[k]; aT[k] = T[k] - pT * P[k]; aC[k] = C[k] - pC * P[k]; }
                else { aH[k] = H[k] + pH * P[k]; aT[k] = T[k] + pT * P[k]; aC[k] = C[k] + pC * P[k]; }
                nH = fmaf(aH[k], aH[k], nH); nT = fmaf(aT[k], aT[k], nT); nC = fmaf(aC[k], aC[k], nC);
            }
            gsum3<G>(nH, nT, nC);
            nH = sqrtf(nH); nT = sqrtf(nT); nC = sqrtf(nC);
            const bool fH = nH > kEpsNormalize, fT = nT > kEpsNormalize, fC = nC > kEpsNormalize;
            const float iH = 1.0f / fmaxf(nH, kEpsNormalize), iT = 1.0f / fmaxf(nT, kEpsNormalize);
            const float iC = 1.0f / fmaxf(nC, kEpsNormalize);
            float up[NCH], un[NCH];
            float sp = 0.f, sn = 0.f;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const float hh = aH[k] * iH, rr = R[k] * iR, tt = aT[k] * iT, cc = aC[k] * iC;
                up[k] = hh + rr - tt;
                un[k] = tail ? (hh + rr - cc) : (cc + rr - tt);
                sp = l1 ? sp + fabsf(up[k]) : fmaf(up[k], up[k], sp);
                sn = l1 ? sn + fabsf(un[k]) : fmaf(un[k], un[k], sn);
            }
            gsum2<G>(sp, sn);
            if (!l1) { sp = sqrtf(sp); sn = sqrtf(sn); }
            const float v = sp + margin - sn;
            acc += fmaxf(v, 0.f);
            const float coef = v > 0.f ? 1.f : (v == 0.f ? 0.5f : 0.f);
            if (coef == 0.f) continue;
            const float ip = (!l1 && sp > 0.f) ? coef / sp : 0.f, in = (!l1 && sn > 0.f) ? -coef / sn : 0.f;
            float gH[NCH], gT[NCH], gC[NCH];  // first: wrt the normalised projections; then wrt a_X; then wrt X
            float dH = 0.f, dT = 0.f, dC = 0.f;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const float gp = l1 ? (up[k] > 0.f ? coef : (up[k] < 0.f ? -coef : 0.f)) : up[k] * ip;
                const float gn = l1 ? (un[k] > 0.f ? -coef : (un[k] < 0.f ? coef : 0.f)) : un[k] * in;
                gRh[k] += gp + gn;
                gH[k] = tail ? gp + gn : gp;
                gT[k] = tail ? -gp : -(gp + gn);
                gC[k] = tail ? -gn : gn;
                dH = fmaf(aH[k], gH[k], dH); dT = fmaf(aT[k], gT[k], dT); dC = fmaf(aC[k], gC[k], dC);
            }
            r_dirty = true;
            gsum3<G>(dH, dT, dC);
            dH *= iH; dT *= iT; dC *= iC;
            float qH = 0.f, qT = 0.f, qC = 0.f;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {  // gradient wrt a_X, and its dot with the projection direction
                gH[k] = fH ? (gH[k] - (aH[k] * iH) * dH) * iH : gH[k] * iH;
                gT[k] = fT ? (gT[k] - (aT[k] * iT) * dT) * iT : gT[k] * iT;
                gC[k] = fC ? (gC[k] - (aC[k] * iC) * dC) * iC : gC[k] * iC;
                qH = fmaf(gH[k], P[k], qH); qT = fmaf(gT[k], P[k], qT); qC = fmaf(gC[k], P[k], qC);
            }
            gsum3<G>(qH, qT, qC);
            if constexpr (M == KGE_TRANSH) {
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    gP[k] -= pH * gH[k] + qH * H[k] + pT * gT[k] + qT * T[k] + pC * gC[k] + qC * C[k];
                    gH[k] -= qH * P[k]; gT[k] -= qT * P[k]; gC[k] -= qC * P[k];
                }
            } else {
                float gHM[NCH], gTM[NCH], gCM[NCH];
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    gP[k] += pH * gH[k] + pT * gT[k] + pC * gC[k];
                    gHM[k] = qH * H[k]; gTM[k] = qT * T[k]; gCM[k] = qC * C[k];
                    gH[k] += qH * HM[k]; gT[k] += qT * TM[k]; gC[k] += qC * CM[k];
                }
                atomic_add_row<G, NCH>(g_entmap + h * (int64_t)d, gHM, d, gl);
                atomic_add_row<G, NCH>(g_entmap + t * (int64_t)d, gTM, d, gl);
                atomic_add_row<G, NCH>(g_entmap + c * (int64_t)d, gCM, d, gl);
            }
            atomic_add_row<G, NCH>(m.grad[0] + h * (int64_t)d, gH, d, gl);
            atomic_add_row<G, NCH>(m.grad[0] + t * (int64_t)d, gT, d, gl);
            atomic_add_row<G, NCH>(m.grad[0] + c * (int64_t)d, gC, d, gl);
        }
        flush_r();
    }
    block_accumulate_loss<G>(acc, gl, loss);
}

// ---- RotatE, sampler fused, shared rows loaded ONCE per bundle (config C3: d=1000, neg_rate 16).
// A bundle = a positive (h,r,t) and its neg_rate corruptions; every negative differs from the positive in ONE entity,
// so the five rows of the positive (h_re, h_im, rel, t_re, t_im), the sin/cos of the relation phases and the rotated
// head h o r are computed once and stay in registers; per negative only the corrupting entity's two rows are gathered
// (2 rows instead of 5, no sincos).  Lane j of the group draws negative j (Philox, hash-set probe) in parallel.
// Pass 1 scores, group softmax gives the detached self-adversarial weights (criterion.py:13-23), pass 2 re-gathers the
// two rows per negative and back-propagates; gradients of the positive's five rows accumulate in registers and are
// scattered once per bundle.
template <int G, int NCH>
__global__ __launch_bounds__(kBlock) void k_rotate_bundle_sampled(DeviceModel m, int64_t n_pos, int neg_rate, float alpha,
                                                                  float* __restrict__ loss, FusedSampler fs) {
    constexpr int GPB = kBlock / G;
    const int gl = threadIdx.x % G;
    const int d = m.dim;
    const float inv_b = 1.0f / (float)n_pos;
    const int64_t s_start = fs.cursor ? fs.start + fs.cursor[0] : fs.start;
    const unsigned long long s_off = fs.cursor ? fs.offset + (unsigned long long)fs.cursor[1] : fs.offset;
    const int gbase = (threadIdx.x & 63) / G * G;
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * GPB + threadIdx.x / G; i < n_pos; i += (int64_t)gridDim.x * GPB) {
        const int64_t row = fs.perm[s_start + i];
        const int64_t h = fs.triples[3 * row], r = fs.triples[3 * row + 1], t = fs.triples[3 * row + 2];
        int my_c = 0, my_tail = 0;  // lane j: corrupting entity and side of negative j
        if (gl < neg_rate) {
            int64_t nh, nt;
            corrupt_one(h, r, t, fs.E, fs.bern, fs.slots, fs.mask, fs.seed, s_off + (unsigned long long)(i * neg_rate + gl), nh, nt);
            my_tail = nh == h;
            my_c = (int)(my_tail ? nt : nh);
        }
        float HR[NCH], HI[NCH], TR[NCH], TI[NCH], CS[NCH], SN[NCH];
        {
            float RL[NCH];
            load_row<G, NCH>(HR, m.tab[0] + h * (int64_t)d, d, gl);
            load_row<G, NCH>(HI, m.tab[1] + h * (int64_t)d, d, gl);
            load_row<G, NCH>(RL, m.tab[2] + r * (int64_t)d, d, gl);
            load_row<G, NCH>(TR, m.tab[0] + t * (int64_t)d, d, gl);
            load_row<G, NCH>(TI, m.tab[1] + t * (int64_t)d, d, gl);
#pragma unroll
            for (int k = 0; k < NCH; ++k) sincosf(RL[k] / m.phase_div, &SN[k], &CS[k]);
        }
        // energy of (head rows A, tail rows B):  sum |A o r - B|^2 - margin
        auto energy = [&](const float (&AR)[NCH], const float (&AI)[NCH], const float (&BR)[NCH], const float (&BI)[NCH]) {
            float p = 0.f;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const float re = AR[k] * CS[k] - AI[k] * SN[k] - BR[k];
                const float im = AR[k] * SN[k] + AI[k] * CS[k] - BI[k];
                p += re * re + im * im;
            }
            return -(m.margin - gsum<G>(p));
        };
        const float s_pos = energy(HR, HI, TR, TI);
        float s_mine = 0.f;
        for (int j = 0; j < neg_rate; ++j) {
            const int64_t c = __shfl(my_c, gbase + j, 64);
            const bool tail = __shfl(my_tail, gbase + j, 64) != 0;
            float CR[NCH], CI[NCH];
            load_row<G, NCH>(CR, m.tab[0] + c * (int64_t)d, d, gl);
            load_row<G, NCH>(CI, m.tab[1] + c * (int64_t)d, d, gl);
            const float sj = tail ? energy(HR, HI, CR, CI) : energy(CR, CI, TR, TI);
            if (gl == j) s_mine = sj;
        }
        // loss and coefficients (as k_selfadv_bundle)
        const bool live = gl < neg_rate;
        const float nj = -s_mine;
        float mx = live ? nj * alpha : -INFINITY;
#pragma unroll
        for (int o = G / 2; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        const float ex = live ? expf(nj * alpha - mx) : 0.f;
        const float den = gsum<G>(ex);
        const float wj = ex / den;
        const float term = gsum<G>(live ? wj * logsigmoid_t(-nj) : 0.f);
        acc += (-term - logsigmoid_t(-s_pos)) * inv_b;
        const float c_mine = live ? -(wj * sigmoid_t(nj)) * inv_b : 0.f;
        const float c_pos = sigmoid_t(s_pos) * inv_b;
        // pass 2: gradient accumulators of the positive's rows; GP = gradient wrt the phase
        float gHR[NCH], gHI[NCH], gTR[NCH], gTI[NCH], GP[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const float re = HR[k] * CS[k] - HI[k] * SN[k] - TR[k], im = HR[k] * SN[k] + HI[k] * CS[k] - TI[k];
            const float Rr = 2.f * c_pos * re, Ii = 2.f * c_pos * im;
            gHR[k] = Rr * CS[k] + Ii * SN[k];
            gHI[k] = -Rr * SN[k] + Ii * CS[k];
            gTR[k] = -Rr; gTI[k] = -Ii;
            GP[k] = Rr * (-HR[k] * SN[k] - HI[k] * CS[k]) + Ii * (HR[k] * CS[k] - HI[k] * SN[k]);
        }
        for (int j = 0; j < neg_rate; ++j) {
            const float cj = __shfl(c_mine, gbase + j, 64);
            if (cj == 0.f) continue;
            const int64_t c = __shfl(my_c, gbase + j, 64);
            const bool tail = __shfl(my_tail, gbase + j, 64) != 0;
            float CR[NCH], CI[NCH], gCR[NCH], gCI[NCH];
            load_row<G, NCH>(CR, m.tab[0] + c * (int64_t)d, d, gl);
            load_row<G, NCH>(CI, m.tab[1] + c * (int64_t)d, d, gl);
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                if (tail) {  // (h, r, c): head rows are the positive's, tail rows are C
                    const float re = HR[k] * CS[k] - HI[k] * SN[k] - CR[k], im = HR[k] * SN[k] + HI[k] * CS[k] - CI[k];
                    const float Rr = 2.f * cj * re, Ii = 2.f * cj * im;
                    gCR[k] = -Rr; gCI[k] = -Ii;
                    gHR[k] += Rr * CS[k] + Ii * SN[k];
                    gHI[k] += -Rr * SN[k] + Ii * CS[k];
                    GP[k] += Rr * (-HR[k] * SN[k] - HI[k] * CS[k]) + Ii * (HR[k] * CS[k] - HI[k] * SN[k]);
                } else {     // (c, r, t): head rows are C, tail rows are the positive's
                    const float re = CR[k] * CS[k] - CI[k] * SN[k] - TR[k], im = CR[k] * SN[k] + CI[k] * CS[k] - TI[k];
                    const float Rr = 2.f * cj * re, Ii = 2.f * cj * im;
                    gCR[k] = Rr * CS[k] + Ii * SN[k];
                    gCI[k] = -Rr * SN[k] + Ii * CS[k];
                    gTR[k] -= Rr; gTI[k] -= Ii;
                    GP[k] += Rr * (-CR[k] * SN[k] - CI[k] * CS[k]) + Ii * (CR[k] * CS[k] - CI[k] * SN[k]);
                }
            }
            atomic_add_row<G, NCH>(m.grad[0] + c * (int64_t)d, gCR, d, gl);
            atomic_add_row<G, NCH>(m.grad[1] + c * (int64_t)d, gCI, d, gl);
        }
#pragma unroll
        for (int k = 0; k < NCH; ++k) GP[k] = GP[k] / m.phase_div;
        atomic_add_row<G, NCH>(m.grad[0] + h * (int64_t)d, gHR, d, gl);
        atomic_add_row<G, NCH>(m.grad[1] + h * (int64_t)d, gHI, d, gl);
        atomic_add_row<G, NCH>(m.grad[2] + r * (int64_t)d, GP, d, gl);
        atomic_add_row<G, NCH>(m.grad[0] + t * (int64_t)d, gTR, d, gl);
        atomic_add_row<G, NCH>(m.grad[1] + t * (int64_t)d, gTI, d, gl);
    }
    block_accumulate_loss<G>(acc, gl, loss);
}

// ---- the staged (atomic-free) form of the same step, SINGLE PASS over the negatives.  The two-pass kernel gathers every
// negative's two rows twice (energies first, gradients once the softmax weights are known); at d = 1000 those 4 KB rows come
// from the Infinity Cache / HBM and the gathers ARE the kernel (283 MB per C3 step at ~3.1 TB/s).  Here the self-adversarial
// softmax is accumulated online (running maximum M, running sums rescaled by exp(M_old - M_new) whenever the maximum moves):
// negative j contributes with the unnormalised weight e_j = exp(alpha n_j - M_j) the moment its rows are in registers --
// its own two gradient rows are staged scaled by e_j sigmoid(n_j) together with the per-pair factor
// exp(M_j - M_final) * (-1 / (B * sum_j e_j)), which kge_optimizer_step_staged applies when it sums the slot; the positive's
// five rows accumulate in registers and are scaled once at the end.  Same mathematics, one gather per row.
template <int G, int NCH>
__global__ __launch_bounds__(kBlock) void k_rotate_bundle_staged(DeviceModel m, int64_t n_pos, int neg_rate, float alpha,
                                                                 float* __restrict__ loss, FusedSampler fs, StageSink sink,
                                                                 float* __restrict__ pair_scale) {
    constexpr int GPB = kBlock / G;
    const int gl = threadIdx.x % G;
    const int d = m.dim;
    const float inv_b = 1.0f / (float)n_pos;
    const int gbase = (threadIdx.x & 63) / G * G;
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * GPB + threadIdx.x / G; i < n_pos; i += (int64_t)gridDim.x * GPB) {
        const int64_t row = fs.perm[fs.start + i];
        const int64_t h = fs.triples[3 * row], r = fs.triples[3 * row + 1], t = fs.triples[3 * row + 2];
        int my_c = 0, my_tail = 0;  // lane j: corrupting entity and side of negative j
        if (gl < neg_rate) {
            int64_t nh, nt;
            corrupt_one(h, r, t, fs.E, fs.bern, fs.slots, fs.mask, fs.seed, fs.offset + (unsigned long long)(i * neg_rate + gl), nh, nt);
            my_tail = nh == h;
            my_c = (int)(my_tail ? nt : nh);
        }
        float HR[NCH], HI[NCH], TR[NCH], TI[NCH], CS[NCH], SN[NCH];
        {
            float RL[NCH];
            load_row<G, NCH>(HR, m.tab[0] + h * (int64_t)d, d, gl);
            load_row<G, NCH>(HI, m.tab[1] + h * (int64_t)d, d, gl);
            load_row<G, NCH>(RL, m.tab[2] + r * (int64_t)d, d, gl);
            load_row<G, NCH>(TR, m.tab[0] + t * (int64_t)d, d, gl);
            load_row<G, NCH>(TI, m.tab[1] + t * (int64_t)d, d, gl);
#pragma unroll
            for (int k = 0; k < NCH; ++k) sincosf(RL[k] / m.phase_div, &SN[k], &CS[k]);
        }
        // unnormalised, running-maximum-relative sums over the negatives seen so far
        float aHR[NCH], aHI[NCH], aTR[NCH], aTI[NCH], aP[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) { aHR[k] = aHI[k] = aTR[k] = aTI[k] = aP[k] = 0.f; }
        float M = -INFINITY, D = 0.f, Lw = 0.f;
        float m_mine = 0.f;   // lane j: the running maximum at the time negative j was staged
        // the rows of negative j + 1 are requested before negative j is evaluated (one wave per SIMD: nothing else would
        // cover the round trip to the Infinity Cache / HBM)
        float NR[NCH], NI[NCH];
        {
            const int64_t c0 = __shfl(my_c, gbase, 64);
            load_row<G, NCH>(NR, m.tab[0] + c0 * (int64_t)d, d, gl);
            load_row<G, NCH>(NI, m.tab[1] + c0 * (int64_t)d, d, gl);
        }
        for (int j = 0; j < neg_rate; ++j) {
            const int64_t c = __shfl(my_c, gbase + j, 64);
            const bool tail = __shfl(my_tail, gbase + j, 64) != 0;
            float CR[NCH], CI[NCH];
#pragma unroll
            for (int k = 0; k < NCH; ++k) { CR[k] = NR[k]; CI[k] = NI[k]; }
            if (j + 1 < neg_rate) {
                const int64_t cn1 = __shfl(my_c, gbase + j + 1, 64);
                load_row<G, NCH>(NR, m.tab[0] + cn1 * (int64_t)d, d, gl);
                load_row<G, NCH>(NI, m.tab[1] + cn1 * (int64_t)d, d, gl);
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the requests in front of this negative's arithmetic
            // residual of the negative triple: (h, r, c) when the tail was corrupted, (c, r, t) otherwise
            float re[NCH], im[NCH], p = 0.f;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const float ar = tail ? HR[k] : CR[k], ai = tail ? HI[k] : CI[k];
                const float br = tail ? CR[k] : TR[k], bi = tail ? CI[k] : TI[k];
                re[k] = ar * CS[k] - ai * SN[k] - br;
                im[k] = ar * SN[k] + ai * CS[k] - bi;
                p += re[k] * re[k] + im[k] * im[k];
            }
            const float nj = m.margin - gsum<G>(p);          // = -(energy of negative j)
            const float a = nj * alpha;
            const float Mn = fmaxf(M, a);
            const float f = expf(M - Mn);                     // 0 for the first negative (M = -inf)
            const float e = expf(a - Mn);
            const float wt = e * sigmoid_t(nj);               // unnormalised coefficient weight
            D = D * f + e;
            Lw = Lw * f + e * logsigmoid_t(-nj);
            M = Mn;
            if (gl == j) m_mine = Mn;
            float gCR[NCH], gCI[NCH];
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const float Rr = 2.f * wt * re[k], Ii = 2.f * wt * im[k];
                aHR[k] *= f; aHI[k] *= f; aTR[k] *= f; aTI[k] *= f; aP[k] *= f;
                if (tail) {
                    gCR[k] = -Rr; gCI[k] = -Ii;
                    aHR[k] += Rr * CS[k] + Ii * SN[k];
                    aHI[k] += -Rr * SN[k] + Ii * CS[k];
                    aP[k] += Rr * (-HR[k] * SN[k] - HI[k] * CS[k]) + Ii * (HR[k] * CS[k] - HI[k] * SN[k]);
                } else {
                    gCR[k] = Rr * CS[k] + Ii * SN[k];
                    gCI[k] = -Rr * SN[k] + Ii * CS[k];
                    aTR[k] -= Rr; aTI[k] -= Ii;
                    aP[k] += Rr * (-CR[k] * SN[k] - CI[k] * CS[k]) + Ii * (CR[k] * CS[k] - CI[k] * SN[k]);
                }
            }
            const int64_t pair = i * neg_rate + j;
            float* slot = sink.stage + (n_pos * sink.ns + pair * sink.nd) * sink.stride;
            store_row<G, NCH>(slot, gCR, d, gl);
            store_row<G, NCH>(slot + sink.stride, gCI, d, gl);
            if (gl == 0) stage_register(sink, (int)c, (int)pair);
        }
        // normalisation: coefficient of negative j = -(e_j sigmoid(n_j) / D) / B with e_j relative to the FINAL maximum
        const float cn = -inv_b / D;
        if (gl < neg_rate) pair_scale[i * neg_rate + gl] = expf(m_mine - M) * cn;
        // the positive triple itself
        float p0 = 0.f, re0[NCH], im0[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            re0[k] = HR[k] * CS[k] - HI[k] * SN[k] - TR[k];
            im0[k] = HR[k] * SN[k] + HI[k] * CS[k] - TI[k];
            p0 += re0[k] * re0[k] + im0[k] * im0[k];
        }
        const float s_pos = -(m.margin - gsum<G>(p0));
        acc += (-(Lw / D) - logsigmoid_t(-s_pos)) * inv_b;
        const float c_pos = sigmoid_t(s_pos) * inv_b;
        float gHR[NCH], gHI[NCH], gTR[NCH], gTI[NCH], GP[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const float Rr = 2.f * c_pos * re0[k], Ii = 2.f * c_pos * im0[k];
            gHR[k] = Rr * CS[k] + Ii * SN[k] + cn * aHR[k];
            gHI[k] = -Rr * SN[k] + Ii * CS[k] + cn * aHI[k];
            gTR[k] = -Rr + cn * aTR[k];
            gTI[k] = -Ii + cn * aTI[k];
            GP[k] = (Rr * (-HR[k] * SN[k] - HI[k] * CS[k]) + Ii * (HR[k] * CS[k] - HI[k] * SN[k]) + cn * aP[k]) / m.phase_div;
        }
        float* slot = sink.stage + i * sink.ns * sink.stride;   // static slots of positive i: h_re, h_im, r, t_re, t_im
        store_row<G, NCH>(slot, gHR, d, gl);
        store_row<G, NCH>(slot + sink.stride, gHI, d, gl);
        store_row<G, NCH>(slot + 2 * sink.stride, GP, d, gl);
        store_row<G, NCH>(slot + 3 * sink.stride, gTR, d, gl);
        store_row<G, NCH>(slot + 4 * sink.stride, gTI, d, gl);
    }
    block_accumulate_loss<G>(acc, gl, loss);
}

// ---- the same single-pass staged step with a bundle's ROWS SPLIT OVER TWO WAVES (64-lane groups only: rows of more than 512
// floats).  One wave per bundle holds eleven 1024-float rows in registers (256 VGPRs + 244 AGPRs): one wave per SIMD, and C3's
// 1024 bundles walk their 16 negatives as a chain of dependent (gather -> evaluate) rounds with nothing to hide the round trips
// behind.  Here each of the two waves of a bundle owns one half of every row (NCH/2 chunks: ~180 registers, two waves per SIMD);
// the only coupling is the energy of a triple, whose two partial sums meet in LDS (one workgroup barrier per negative, double
// buffered).  Everything derived from the energy (softmax state, weights) is computed redundantly by both waves.  Same
// mathematics; the energy is the sum of two half-row sums instead of one 64-lane butterfly over the whole row.
// (Splitting the NEGATIVES over two waves instead keeps all eleven rows per wave: 626 spilled registers at two waves per SIMD.)
// (waves per SIMD the register budget is sized for: a wave owning 8 chunks of every row needs ~256 registers whatever the split)
template <int NCH, int SPLIT, bool PAD>
__global__ __launch_bounds__(kBlock, NCH >= 8 ? 2 : SPLIT) void k_rotate_bundle_staged_split(DeviceModel m, int64_t n_pos, int neg_rate, float alpha,
                                                                          float* __restrict__ loss, FusedSampler fs, StageSink sink,
                                                                          float* __restrict__ pair_scale) {
    constexpr int G = 64;
    constexpr int BPB = kBlock / (64 * SPLIT);  // bundles per workgroup (SPLIT waves each)
    constexpr int HALF = G * NCH;              // floats of a row one wave owns
    __shared__ float s_p[2][BPB][SPLIT];       // [parity][bundle][part] partial energies
    const int gl = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6, half = wave % SPLIT, slot_b = wave / SPLIT;   // half: which part of every row
    const int d = m.dim;
    const int dh = min(max(d - half * HALF, 0), HALF);     // this wave's share of a row
    const int64_t ho = (int64_t)half * HALF;
    const float inv_b = 1.0f / (float)n_pos;
    float acc = 0.f;
    int par = 0;
    // stage rows as wide as the SPLIT waves cover, ns + nd spare rows behind the n_pos * (ns + nd * neg_rate) used ones (StagedPlan)
    constexpr bool padded = PAD;     // (chosen at launch: sink.stride >= SPLIT * HALF and sink.spare)
    float* const spare = sink.stage + (n_pos * sink.ns + n_pos * (int64_t)neg_rate * sink.nd) * sink.stride + ho;
    for (int64_t i0 = (int64_t)blockIdx.x * BPB; i0 < n_pos; i0 += (int64_t)gridDim.x * BPB) {
        const int64_t i = i0 + slot_b;
        const bool valid = i < n_pos;
        int64_t h = 0, r = 0, t = 0;
        int my_c = 0, my_tail = 0;
        if (valid) {
            const int64_t row = fs.perm[fs.start + i];
            h = fs.triples[3 * row]; r = fs.triples[3 * row + 1]; t = fs.triples[3 * row + 2];
            if (gl < neg_rate) {
                int64_t nh, nt;
                corrupt_one(h, r, t, fs.E, fs.bern, fs.slots, fs.mask, fs.seed, fs.offset + (unsigned long long)(i * neg_rate + gl), nh, nt);
                my_tail = nh == h;
                my_c = (int)(my_tail ? nt : nh);
            }
        }
        const int dv = valid ? dh : 0;         // (an invalid bundle loads and stores nothing but keeps the barriers)
        // Round 6: where the stage rows are padded to the width the bundle's waves cover (`padded`) every gradient store is
        // UNCONDITIONAL (lanes beyond the row write the row's padding; an invalid bundle writes the spare rows behind the stage) and the
        // negatives register behind the loop.  With every store in its own predicated region (store_row) the loop's memory operations
        // could not be counted, and the wait in front of the prefetched negative was an s_waitcnt vmcnt(0): it drained the previous
        // iteration's 16 stores as well.  Same lanes, same elements, same arithmetic: bit-identical results.  (Unconditional clamped
        // LOADS as well were measured: 256 VGPRs + spills, C3 step 217 -> 275 us.)
        auto ld = [&](float (&x)[NCH], const float* __restrict__ row) __attribute__((always_inline)) { load_row<G, NCH>(x, row, dv, gl); };
        auto st = [&](float* __restrict__ row, const float (&g)[NCH]) __attribute__((always_inline)) {
            if constexpr (padded) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) row[c * G + gl] = g[c];
            } else {
                store_row<G, NCH>(row, g, dv, gl);
            }
        };
        float HR[NCH], HI[NCH], TR[NCH], TI[NCH], CS[NCH], SN[NCH];
        {
            float RL[NCH];
            ld(HR, m.tab[0] + h * (int64_t)d + ho);
            ld(HI, m.tab[1] + h * (int64_t)d + ho);
            ld(RL, m.tab[2] + r * (int64_t)d + ho);
            ld(TR, m.tab[0] + t * (int64_t)d + ho);
            ld(TI, m.tab[1] + t * (int64_t)d + ho);
#pragma unroll
            for (int k = 0; k < NCH; ++k) sincosf(RL[k] / m.phase_div, &SN[k], &CS[k]);
        }
        float aHR[NCH], aHI[NCH], aTR[NCH], aTI[NCH], aP[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) { aHR[k] = aHI[k] = aTR[k] = aTI[k] = aP[k] = 0.f; }
        float M = -INFINITY, D = 0.f, Lw = 0.f, m_mine = 0.f;
        float NR[NCH], NI[NCH];
        {
            const int64_t c0 = __shfl(my_c, 0, 64);
            ld(NR, m.tab[0] + c0 * (int64_t)d + ho);
            ld(NI, m.tab[1] + c0 * (int64_t)d + ho);
        }
        for (int j = 0; j < neg_rate; ++j) {
            const int64_t c = __shfl(my_c, j, 64);
            const bool tail = __shfl(my_tail, j, 64) != 0;
            float CR[NCH], CI[NCH];
#pragma unroll
            for (int k = 0; k < NCH; ++k) { CR[k] = NR[k]; CI[k] = NI[k]; }
            if (j + 1 < neg_rate) {
                const int64_t cn1 = __shfl(my_c, j + 1, 64);
                ld(NR, m.tab[0] + cn1 * (int64_t)d + ho);
                ld(NI, m.tab[1] + cn1 * (int64_t)d + ho);
            }
            __builtin_amdgcn_sched_barrier(0);
            float re[NCH], im[NCH], p = 0.f;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const float ar = tail ? HR[k] : CR[k], ai = tail ? HI[k] : CI[k];
                const float br = tail ? CR[k] : TR[k], bi = tail ? CI[k] : TI[k];
                re[k] = ar * CS[k] - ai * SN[k] - br;
                im[k] = ar * SN[k] + ai * CS[k] - bi;
                p += re[k] * re[k] + im[k] * im[k];
            }
            p = gsum<G>(p);
            if (gl == 0) s_p[par][slot_b][half] = p;
            __syncthreads();
            float en = 0.f;
#pragma unroll
            for (int q = 0; q < SPLIT; ++q) en += s_p[par][slot_b][q];
            const float nj = m.margin - en;   // = -(energy of negative j)
            par ^= 1;
            const float a = nj * alpha;
            const float Mn = fmaxf(M, a);
            const float f = expf(M - Mn);
            const float e = expf(a - Mn);
            const float wt = e * sigmoid_t(nj);
            D = D * f + e;
            Lw = Lw * f + e * logsigmoid_t(-nj);
            M = Mn;
            if (gl == j) m_mine = Mn;
            float gCR[NCH], gCI[NCH];
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const float Rr = 2.f * wt * re[k], Ii = 2.f * wt * im[k];
                aHR[k] *= f; aHI[k] *= f; aTR[k] *= f; aTI[k] *= f; aP[k] *= f;
                if (tail) {
                    gCR[k] = -Rr; gCI[k] = -Ii;
                    aHR[k] += Rr * CS[k] + Ii * SN[k];
                    aHI[k] += -Rr * SN[k] + Ii * CS[k];
                    aP[k] += Rr * (-HR[k] * SN[k] - HI[k] * CS[k]) + Ii * (HR[k] * CS[k] - HI[k] * SN[k]);
                } else {
                    gCR[k] = Rr * CS[k] + Ii * SN[k];
                    gCI[k] = -Rr * SN[k] + Ii * CS[k];
                    aTR[k] -= Rr; aTI[k] -= Ii;
                    aP[k] += Rr * (-CR[k] * SN[k] - CI[k] * CS[k]) + Ii * (CR[k] * CS[k] - CI[k] * SN[k]);
                }
            }
            const int64_t pair = i * neg_rate + j;
            float* slot = (valid || !padded) ? sink.stage + (n_pos * sink.ns + pair * sink.nd) * sink.stride + ho : spare + sink.ns * sink.stride;
            st(slot, gCR);
            st(slot + sink.stride, gCI);
            if constexpr (!padded) { if (valid && half == 0 && gl == 0) stage_register(sink, (int)c, (int)pair); }
        }
        // padded form: the negatives' rows register with their entities behind the loop, one lane per negative (inside it the returning
        // atomics made the loop's memory-operation count path dependent; the owners sum in ascending slot order whatever order these
        // arrive in).  The tight-row form keeps the registration where it was: there it measured faster (217 vs 245 us per C3 step).
        if constexpr (padded) { if (valid && half == 0 && gl < neg_rate) stage_register(sink, my_c, (int)(i * neg_rate + gl)); }
        const float cn = -inv_b / D;
        if (valid && half == 0 && gl < neg_rate) pair_scale[i * neg_rate + gl] = expf(m_mine - M) * cn;
        float p0 = 0.f, re0[NCH], im0[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            re0[k] = HR[k] * CS[k] - HI[k] * SN[k] - TR[k];
            im0[k] = HR[k] * SN[k] + HI[k] * CS[k] - TI[k];
            p0 += re0[k] * re0[k] + im0[k] * im0[k];
        }
        p0 = gsum<G>(p0);
        if (gl == 0) s_p[par][slot_b][half] = p0;
        __syncthreads();
        float e0 = 0.f;
#pragma unroll
        for (int q = 0; q < SPLIT; ++q) e0 += s_p[par][slot_b][q];
        const float s_pos = -(m.margin - e0);
        par ^= 1;
        if (valid && half == 0) acc += (-(Lw / D) - logsigmoid_t(-s_pos)) * inv_b;
        const float c_pos = sigmoid_t(s_pos) * inv_b;
        float gHR[NCH], gHI[NCH], gTR[NCH], gTI[NCH], GP[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const float Rr = 2.f * c_pos * re0[k], Ii = 2.f * c_pos * im0[k];
            gHR[k] = Rr * CS[k] + Ii * SN[k] + cn * aHR[k];
            gHI[k] = -Rr * SN[k] + Ii * CS[k] + cn * aHI[k];
            gTR[k] = -Rr + cn * aTR[k];
            gTI[k] = -Ii + cn * aTI[k];
            GP[k] = (Rr * (-HR[k] * SN[k] - HI[k] * CS[k]) + Ii * (HR[k] * CS[k] - HI[k] * SN[k]) + cn * aP[k]) / m.phase_div;
        }
        float* slot = (valid || !padded) ? sink.stage + i * sink.ns * sink.stride + ho : spare;
        st(slot, gHR);
        st(slot + sink.stride, gHI);
        st(slot + 2 * sink.stride, GP);
        st(slot + 3 * sink.stride, gTR);
        st(slot + 4 * sink.stride, gTI);
    }
    block_accumulate_loss<G>(acc, gl, loss);
}

// ---- self-adversarial loss coefficients (criterion.py:13-23).  In: energies.  Out (in place): dL/d energy.
__global__ __launch_bounds__(kBlock) void k_selfadv_coeffs(float* __restrict__ pos, float* __restrict__ neg,
                                                           int64_t n_pos, int neg_rate, float alpha,
                                                           float* __restrict__ loss) {
    const float inv_b = 1.0f / (float)n_pos;
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pos; i += (int64_t)gridDim.x * blockDim.x) {
        float* ng = neg + i * neg_rate;
        float mx = -INFINITY;
        for (int j = 0; j < neg_rate; ++j) mx = fmaxf(mx, -ng[j] * alpha);
        float den = 0.f;
        for (int j = 0; j < neg_rate; ++j) den += expf(-ng[j] * alpha - mx);
        float term = 0.f;
        for (int j = 0; j < neg_rate; ++j) {
            const float nj = -ng[j];
            const float w = expf(nj * alpha - mx) / den;
            term += w * logsigmoid_t(-nj);
            ng[j] = -(w * sigmoid_t(nj)) * inv_b;
        }
        const float p = -pos[i];
        acc += (-term - logsigmoid_t(p)) * inv_b;
        pos[i] = sigmoid_t(-p) * inv_b;
    }
    block_accumulate_loss<1>(acc, 0, loss);
}

static bool geometry_for(const kge_model_desc* m, Geometry* geo) {
    if (!pick_geometry(m->dim, geo)) {
        set_error("hidden size %d exceeds the register-resident row kernels (max 2048)", m->dim);
        return false;
    }
    return true;
}

int launch_pairwise_hinge_sampled(const kge_model_desc* m, const int64_t* triples, const int64_t* perm, int64_t start,
                                  int64_t n, const float* bern, const uint64_t* slots, int64_t n_slots, uint64_t seed,
                                  uint64_t offset, const int64_t* cursor, float margin, float* loss, hipStream_t s) {
    Geometry geo;
    if (!geometry_for(m, &geo)) return -1;
    if (m->tot_entity > (1 << 24)) { set_error("fused sampler: more than 2^24 entities not supported by the packed key"); return -1; }
    const DeviceModel dm = to_device_model(m);
    FusedSampler fs;
    fs.triples = triples; fs.perm = perm; fs.start = start; fs.E = m->tot_entity; fs.bern = bern;
    fs.slots = (const unsigned long long*)slots; fs.mask = (unsigned long long)(slots ? n_slots - 1 : 0);
    fs.seed = seed; fs.offset = offset; fs.cursor = cursor;
    if (m->model == KGE_TRANSE || m->model == KGE_TRANSM) {  // shared-row specialisation: 4 row gathers / scatters per pair, not 6
#define KGE_TE(G_, NCH_, SC)                                                                                               \
    if (geo.G == G_ && geo.NCH == NCH_ && (m->model == KGE_TRANSM) == SC) {                                                 \
        if (n >= 16384)  /* big batch: 4 pairs per group share the relation row; small batch: one pair per group */ \
            k_transe_pair_sampled<G_, NCH_, 4, SC><<<dim3(Launch<KGE_TRANSE, G_, NCH_>::grid((n + 3) / 4)), dim3(kBlock), 0, s>>>(dm, n, margin, loss, fs); \
        else                                                                                                                \
            k_transe_pair_sampled<G_, NCH_, 1, SC><<<dim3(Launch<KGE_TRANSE, G_, NCH_>::grid(n)), dim3(kBlock), 0, s>>>(dm, n, margin, loss, fs); \
        return check_launch("k_transe_pair_sampled");                                                                       \
    }
#define KGE_TE_ALL(SC) KGE_TE(32, 1, SC) KGE_TE(32, 2, SC) KGE_TE(32, 4, SC) KGE_TE(32, 8, SC) KGE_TE(64, 8, SC) KGE_TE(64, 16, SC)
        KGE_TE_ALL(false) KGE_TE_ALL(true)
#undef KGE_TE_ALL
#undef KGE_TE
    }
    if (m->model == KGE_TRANSH || m->model == KGE_TRANSD) {  // shared-row specialisations with the projections
#define KGE_TX(MID, G_, NCH_)                                                                                               \
    if (m->model == MID && geo.G == G_ && geo.NCH == NCH_) {                                                                 \
        k_transx_pair_sampled<MID, G_, NCH_, 4><<<dim3(Launch<MID, G_, NCH_>::grid((n + 3) / 4)), dim3(kBlock), 0, s>>>(dm, n, margin, loss, fs); \
        return check_launch("k_transx_pair_sampled");                                                                        \
    }
#define KGE_TX_ALL(MID) KGE_TX(MID, 32, 1) KGE_TX(MID, 32, 2) KGE_TX(MID, 32, 4) KGE_TX(MID, 32, 8) KGE_TX(MID, 64, 8) KGE_TX(MID, 64, 16)
        KGE_TX_ALL(KGE_TRANSH) KGE_TX_ALL(KGE_TRANSD)
#undef KGE_TX_ALL
#undef KGE_TX
    }
    return launch_pairwise_hinge_sampled_generic(m, geo, fs, n, margin, loss, s);
}

int launch_rotate_bundle_sampled(const kge_model_desc* m, const int64_t* triples, const int64_t* perm, int64_t start,
                                 int64_t n_pos, int neg_rate, float alpha, const float* bern, const uint64_t* slots,
                                 int64_t n_slots, uint64_t seed, uint64_t offset, const int64_t* cursor, float* loss,
                                 const StageSink* sink, float* pair_scale, hipStream_t s) {
    Geometry geo;
    if (!geometry_for(m, &geo)) return -1;
    if (m->model != KGE_ROTATE) { set_error("kge_train_pairwise_selfadv_sampled: RotatE only"); return -1; }
    if (neg_rate > geo.G) { set_error("kge_train_pairwise_selfadv_sampled: neg_rate %d exceeds the lane group (%d)", neg_rate, geo.G); return -1; }
    if (m->tot_entity > (1 << 24)) { set_error("fused sampler: more than 2^24 entities not supported by the packed key"); return -1; }
    const DeviceModel dm = to_device_model(m);
    FusedSampler fs;
    fs.triples = triples; fs.perm = perm; fs.start = start; fs.E = m->tot_entity; fs.bern = bern;
    fs.slots = (const unsigned long long*)slots; fs.mask = (unsigned long long)(slots ? n_slots - 1 : 0);
    fs.seed = seed; fs.offset = offset; fs.cursor = cursor;
    const int split_sw = switch_value("ROTATE_SPLIT");
    const int split = split_sw >= 0 ? split_sw : 2;   // A/B switch (0 / 2 / 4); measured 83 / 73 / 79 us at C3
    // rows of 1025..2048 floats (geometry {64, 32}) exist only as four waves of 512 floats each: every other form would hold
    // 1024 floats of eleven rows per wave (spills) or, worse, cover only the first 1024 floats of a row
    const bool wide = geo.G == 64 && geo.NCH == 32;
    if (wide && !sink) {
        set_error("kge_train_pairwise_selfadv_sampled: hidden size %d > 1024 needs the staged form (a stage sink); "
                  "use the explicit-id step (kge_train_pairwise_selfadv)", m->dim);
        return -1;
    }
    if (sink && geo.G == 64 && (wide || split == 2 || split == 4)) {   // rows of more than 512 floats: a bundle's rows over two or four waves
        const int sp = wide ? 4 : split;
        int64_t b = (n_pos * sp + 3) / 4;
        if (b > kMaxBlocks) b = kMaxBlocks;
#define KGE_RS(NCH_, SP_)                                                                                                                  \
        {                                                                                                                                  \
            if (sink->spare && sink->stride >= (int64_t)64 * NCH_ * SP_)                                                                   \
                k_rotate_bundle_staged_split<NCH_, SP_, true><<<dim3((unsigned)b), dim3(kBlock), 0, s>>>(dm, n_pos, neg_rate, alpha, loss, fs, *sink, pair_scale);  \
            else                                                                                                                           \
                k_rotate_bundle_staged_split<NCH_, SP_, false><<<dim3((unsigned)b), dim3(kBlock), 0, s>>>(dm, n_pos, neg_rate, alpha, loss, fs, *sink, pair_scale); \
        }
        if (wide) KGE_RS(8, 4)
        else if (geo.NCH == 8 && split == 2) KGE_RS(4, 2)
        else if (geo.NCH == 8) KGE_RS(2, 4)
        else if (split == 2) KGE_RS(8, 2)
        else KGE_RS(4, 4)
#undef KGE_RS
        return check_launch("k_rotate_bundle_staged_split");
    }
#define KGE_RB(G_, NCH_)                                                                                                      \
    if (geo.G == G_ && geo.NCH == NCH_) {                                                                                      \
        if (sink)                                                                                                              \
            k_rotate_bundle_staged<G_, NCH_><<<dim3(Launch<KGE_ROTATE, G_, NCH_>::grid(n_pos)), dim3(kBlock), 0, s>>>(dm, n_pos, neg_rate, alpha, loss, fs, *sink, pair_scale); \
        else                                                                                                                   \
            k_rotate_bundle_sampled<G_, NCH_><<<dim3(Launch<KGE_ROTATE, G_, NCH_>::grid(n_pos)), dim3(kBlock), 0, s>>>(dm, n_pos, neg_rate, alpha, loss, fs); \
        return check_launch("k_rotate_bundle_sampled");                                                                        \
    }
    KGE_RB(32, 1) KGE_RB(32, 2) KGE_RB(32, 4) KGE_RB(32, 8) KGE_RB(64, 8) KGE_RB(64, 16)
#undef KGE_RB
    set_error("kge_train_pairwise_selfadv_sampled: no kernel for the row geometry {%d, %d} (hidden size %d)", geo.G, geo.NCH, m->dim);
    return -1;
}

int launch_selfadv_coeffs(float* pos_scores, float* neg_scores, int64_t n_pos, int neg_rate, float alpha,
                          float* loss, hipStream_t s) {
    int64_t b = (n_pos + kBlock - 1) / kBlock;
    if (b > kMaxBlocks) b = kMaxBlocks;
    hipLaunchKernelGGL(k_selfadv_coeffs, dim3((int)b), dim3(kBlock), 0, s, pos_scores, neg_scores, n_pos, neg_rate,
                       alpha, loss);
    return check_launch("k_selfadv_coeffs");
}

}  // namespace kge
