// kge_device.h -- device-side building blocks shared by the gfx950 KGE kernels.
//
// Geometry.  A *group* of G lanes (G in {16,32,64}, a sub-division of the 64-lane wavefront) owns one
// triple.  Row element e of a gathered embedding row lives in lane (e % G), register (e / G): every
// global load / atomic of a group touches G consecutive floats, so a wave instruction covers 64/G rows
// with fully used 64..256-byte segments, any row length, no alignment requirement.  Reductions over a
// row are butterflies inside the group.  Rows stay in registers between the forward and backward
// halves of a fused kernel: nothing [B,d]-shaped is ever written to HBM.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/kge_hip.h"

namespace kge {

constexpr float kEpsNormalize = 1e-12f;  // F.normalize eps
constexpr int kLossSlots = 32;           // loss accumulators are spread over 32 cache lines ...
constexpr int kLossStride = 32;          // ... 32 floats (128 B) apart: loss = sum_k loss[k*32]

struct DeviceModel {  // by-value kernel argument
    const float* tab[KGE_MAX_TABLES];
    float* grad[KGE_MAX_TABLES];
    int dim;       // entity dim
    int rel_dim;
    int l1;
    float margin;
    float phase_div;  // RotatE: embedding_range / pi (the divisor the reference uses, pairwise.py:781)
};

// ------------------------------------------------------------------ group reductions
// All-reduce (sum) over the G lanes of a group, every lane gets the total.  Steps inside a 16-lane row are DPP
// modifiers folded into the v_add_f32 (quad_perm xor-1, xor-2, then row_half_mirror / row_mirror, valid because the
// lanes being mirrored already hold equal partial sums): no LDS traffic, no address VGPR, one VALU op per step.  Only
// the cross-row steps (lane ^ 16, lane ^ 32) go through the LDS crossbar (ds_swizzle / ds_bpermute).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float swz_xor16(float v) {  // BitMode swizzle: and 0x1F, or 0, xor 0x10
    return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));
}
template <int G>
__device__ __forceinline__ float gsum(float v) {
    static_assert(G == 1 || G == 16 || G == 32 || G == 64, "group width");
    if constexpr (G >= 16) {
        v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
        v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
        v += dpp_mov<0x141>(v);  // row_half_mirror
        v += dpp_mov<0x140>(v);  // row_mirror
    }
    if constexpr (G >= 32) v += swz_xor16(v);
    if constexpr (G >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}
template <int G>
__device__ __forceinline__ void gsum3(float& a, float& b, float& c) {
    a = gsum<G>(a); b = gsum<G>(b); c = gsum<G>(c);  // independent chains; the scheduler interleaves them
}
template <int G>
__device__ __forceinline__ void gsum2(float& a, float& b) {
    a = gsum<G>(a); b = gsum<G>(b);
}
__device__ __forceinline__ float wave_sum(float v) { return gsum<64>(v); }

// ------------------------------------------------------------------ row <-> register movement
template <int G, int NCH>
__device__ __forceinline__ void load_row(float (&x)[NCH], const float* __restrict__ row, int dim, int gl) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        int e = c * G + gl;
        x[c] = (e < dim) ? row[e] : 0.0f;
    }
}
template <int G, int NCH>
__device__ __forceinline__ void atomic_add_row(float* __restrict__ row, const float (&g)[NCH], int dim, int gl) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        int e = c * G + gl;
        if (e < dim) unsafeAtomicAdd(row + e, g[c]);  // global_atomic_add_f32, no return
    }
}

template <int G, int NCH>
__device__ __forceinline__ void store_row(float* __restrict__ row, const float (&g)[NCH], int dim, int gl) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        int e = c * G + gl;
        if (e < dim) row[e] = g[c];
    }
}

// ------------------------------------------------------------------ roles: which table / which id a row comes from
// role r of model M is row  tab[role_tab(M,r)][ id[role_sel(M,r)] ],  id = {h, r, t}
__host__ __device__ constexpr int role_count(int M) {
    return M == KGE_TRANSE ? 3 : M == KGE_TRANSH ? 4 : M == KGE_TRANSD ? 6 : M == KGE_ROTATE ? 5
         : M == KGE_DISTMULT ? 3 : M == KGE_COMPLEX ? 6 : M == KGE_ANALOGY ? 9
         : M == KGE_TRANSM ? 4 : M == KGE_CP ? 3 : (M == KGE_SIMPLE || M == KGE_SIMPLE_IGNR) ? 6 : M == KGE_QUATE ? 12 : 0;
}
__host__ __device__ constexpr int role_tab(int M, int r) {
    switch (M) {
        case KGE_TRANSE: case KGE_DISTMULT: { constexpr int t[3] = {0, 1, 0}; return t[r]; }
        case KGE_TRANSH: { constexpr int t[4] = {0, 1, 0, 2}; return t[r]; }            // eh, er, et, w[r]
        case KGE_TRANSD: { constexpr int t[6] = {0, 1, 0, 2, 3, 2}; return t[r]; }      // eh, er, et, hm, rm, tm
        case KGE_ROTATE: { constexpr int t[5] = {0, 1, 2, 0, 1}; return t[r]; }         // hr, hi, rel, tr, ti
        case KGE_COMPLEX: { constexpr int t[6] = {0, 1, 2, 3, 0, 1}; return t[r]; }     // hr, hi, rr, ri, tr, ti
        case KGE_ANALOGY: { constexpr int t[9] = {0, 1, 0, 2, 3, 4, 5, 2, 3}; return t[r]; }  // eh, er, et | hr,hi,rr,ri,tr,ti
        case KGE_TRANSM: { constexpr int t[4] = {0, 1, 0, 2}; return t[r]; }            // eh, er, et, theta[r] (1 float)
        case KGE_CP: { constexpr int t[3] = {0, 1, 2}; return t[r]; }                   // sub[h], rel[r], obj[t]
        case KGE_SIMPLE: case KGE_SIMPLE_IGNR: { constexpr int t[6] = {0, 0, 2, 3, 1, 1}; return t[r]; }  // h1,h2,r1,r2,t1,t2
        case KGE_QUATE: { constexpr int t[12] = {0, 1, 2, 3, 0, 1, 2, 3, 4, 5, 6, 7}; return t[r]; }      // h sxyz, t sxyz, r sxyz
    }
    return 0;
}
__host__ __device__ constexpr int role_sel(int M, int r) {
    switch (M) {
        case KGE_TRANSE: case KGE_DISTMULT: { constexpr int s[3] = {0, 1, 2}; return s[r]; }
        case KGE_TRANSH: { constexpr int s[4] = {0, 1, 2, 1}; return s[r]; }
        case KGE_TRANSD: { constexpr int s[6] = {0, 1, 2, 0, 1, 2}; return s[r]; }
        case KGE_ROTATE: { constexpr int s[5] = {0, 0, 1, 2, 2}; return s[r]; }
        case KGE_COMPLEX: { constexpr int s[6] = {0, 0, 1, 1, 2, 2}; return s[r]; }
        case KGE_ANALOGY: { constexpr int s[9] = {0, 1, 2, 0, 0, 1, 1, 2, 2}; return s[r]; }
        case KGE_TRANSM: { constexpr int s[4] = {0, 1, 2, 1}; return s[r]; }
        case KGE_CP: { constexpr int s[3] = {0, 1, 2}; return s[r]; }
        case KGE_SIMPLE: case KGE_SIMPLE_IGNR: { constexpr int s[6] = {0, 2, 1, 1, 2, 0}; return s[r]; }  // h2 = head[t], t2 = tail[h]
        case KGE_QUATE: { constexpr int s[12] = {0, 0, 0, 0, 2, 2, 2, 2, 1, 1, 1, 1}; return s[r]; }
    }
    return 0;
}
// roles whose table is a fixed input (no gradient buffer, never scattered)
__host__ __device__ constexpr bool role_trainable(int M, int r) { return !(M == KGE_TRANSM && r == 3); }
template <int M>
__device__ __forceinline__ int role_dim(const DeviceModel& m, int r) {
    if (M == KGE_ANALOGY && r >= 3) return m.dim / 2;
    if (M == KGE_TRANSM && r == 3) return 1;
    return m.dim;
}

template <int M, int NCH>
struct Rows {
    float x[role_count(M)][NCH];
};

template <int M, int G, int NCH>
__device__ __forceinline__ void load_rows(Rows<M, NCH>& R, const DeviceModel& m, const int64_t (&id)[3], int gl) {
#pragma unroll
    for (int r = 0; r < role_count(M); ++r) {
        const int d = role_dim<M>(m, r);
        load_row<G, NCH>(R.x[r], m.tab[role_tab(M, r)] + id[role_sel(M, r)] * (int64_t)d, d, gl);
    }
}
template <int M, int G, int NCH>
__device__ __forceinline__ void scatter_rows(const Rows<M, NCH>& Gr, const DeviceModel& m, const int64_t (&id)[3], int gl) {
#pragma unroll
    for (int r = 0; r < role_count(M); ++r) {
        if (!role_trainable(M, r)) continue;
        const int d = role_dim<M>(m, r);
        atomic_add_row<G, NCH>(m.grad[role_tab(M, r)] + id[role_sel(M, r)] * (int64_t)d, Gr.x[r], d, gl);
    }
}

// ------------------------------------------------------------------ TransE / TransH / TransD distance tail
// s = || a/max(|a|,eps) + b/max(|b|,eps) - c/max(|c|,eps) ||_{1 or 2}       (pairwise.py:69-76)
template <int NCH>
struct TailSaved {
    float ia, ib, ic;  // 1 / max(norm, eps)
    bool fa, fb, fc;   // norm > eps (gradient flows through the norm)
    float u[NCH];
    float s;
};

template <int G, int NCH>
__device__ __forceinline__ float tail_fwd(const float (&a)[NCH], const float (&b)[NCH], const float (&c)[NCH],
                                          bool l1, TailSaved<NCH>& sv) {
    float na = 0.f, nb = 0.f, nc = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        na = fmaf(a[i], a[i], na);
        nb = fmaf(b[i], b[i], nb);
        nc = fmaf(c[i], c[i], nc);
    }
    gsum3<G>(na, nb, nc);
    na = sqrtf(na); nb = sqrtf(nb); nc = sqrtf(nc);
    sv.fa = na > kEpsNormalize; sv.fb = nb > kEpsNormalize; sv.fc = nc > kEpsNormalize;
    sv.ia = 1.0f / fmaxf(na, kEpsNormalize);
    sv.ib = 1.0f / fmaxf(nb, kEpsNormalize);
    sv.ic = 1.0f / fmaxf(nc, kEpsNormalize);
    float p = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        float u = a[i] * sv.ia + b[i] * sv.ib - c[i] * sv.ic;
        sv.u[i] = u;
        p = l1 ? p + fabsf(u) : fmaf(u, u, p);
    }
    p = gsum<G>(p);
    sv.s = l1 ? p : sqrtf(p);
    return sv.s;
}

// gradient of ds*s wrt a, b, c
template <int G, int NCH>
__device__ __forceinline__ void tail_bwd(const float (&a)[NCH], const float (&b)[NCH], const float (&c)[NCH],
                                         bool l1, const TailSaved<NCH>& sv, float ds,
                                         float (&ga)[NCH], float (&gb)[NCH], float (&gc)[NCH]) {
    float g[NCH];
    const float inv = (!l1 && sv.s > 0.f) ? ds / sv.s : 0.f;
    float da = 0.f, db = 0.f, dc = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const float u = sv.u[i];
        g[i] = l1 ? ((u > 0.f) ? ds : (u < 0.f) ? -ds : 0.f) : u * inv;
        da = fmaf(a[i], g[i], da);
        db = fmaf(b[i], g[i], db);
        dc = fmaf(c[i], g[i], dc);
    }
    gsum3<G>(da, db, dc);
    da *= sv.ia; db *= sv.ib; dc *= sv.ic;  // a_hat . g
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        ga[i] = sv.fa ? (g[i] - (a[i] * sv.ia) * da) * sv.ia : g[i] * sv.ia;
        gb[i] = sv.fb ? (g[i] - (b[i] * sv.ib) * db) * sv.ib : g[i] * sv.ib;
        gc[i] = -(sv.fc ? (g[i] - (c[i] * sv.ic) * dc) * sv.ic : g[i] * sv.ic);
    }
}

// ------------------------------------------------------------------ per-model forward (+ saved state) and backward
template <int M, int NCH>
struct Saved {
    TailSaved<NCH> tail;
    float a[NCH], c[NCH];  // projected head / tail (TransH, TransD)
    float wh[NCH];         // TransH: normalised hyperplane normal; RotatE: cos(phase)
    float aux[NCH];        // RotatE: sin(phase)
    float ph, pt, iw;      // projections h.w^ / t.w^ (TransH), h.hm / t.tm (TransD); 1/max(|w|,eps); TransM: theta_r
    bool fw;               // TransH: |w| > eps; SimplE: the clamp passes the gradient
};

template <int M, int G, int NCH>
__device__ __forceinline__ float model_fwd(const Rows<M, NCH>& R, const DeviceModel& m, Saved<M, NCH>& sv) {
    if constexpr (M == KGE_TRANSE) {
        return tail_fwd<G, NCH>(R.x[0], R.x[1], R.x[2], m.l1, sv.tail);
    } else if constexpr (M == KGE_TRANSH) {  // pairwise.py:143-182
        float nw = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) nw = fmaf(R.x[3][i], R.x[3][i], nw);
        nw = sqrtf(gsum<G>(nw));
        sv.fw = nw > kEpsNormalize;
        sv.iw = 1.0f / fmaxf(nw, kEpsNormalize);
        float ph = 0.f, pt = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            sv.wh[i] = R.x[3][i] * sv.iw;
            ph = fmaf(R.x[0][i], sv.wh[i], ph);
            pt = fmaf(R.x[2][i], sv.wh[i], pt);
        }
        gsum2<G>(ph, pt);
        sv.ph = ph; sv.pt = pt;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            sv.a[i] = R.x[0][i] - ph * sv.wh[i];
            sv.c[i] = R.x[2][i] - pt * sv.wh[i];
        }
        return tail_fwd<G, NCH>(sv.a, R.x[1], sv.c, m.l1, sv.tail);
    } else if constexpr (M == KGE_TRANSD) {  // pairwise.py:229-278
        float ph = 0.f, pt = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            ph = fmaf(R.x[0][i], R.x[3][i], ph);
            pt = fmaf(R.x[2][i], R.x[5][i], pt);
        }
        gsum2<G>(ph, pt);
        sv.ph = ph; sv.pt = pt;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            sv.a[i] = R.x[0][i] + ph * R.x[4][i];
            sv.c[i] = R.x[2][i] + pt * R.x[4][i];
        }
        return tail_fwd<G, NCH>(sv.a, R.x[1], sv.c, m.l1, sv.tail);
    } else if constexpr (M == KGE_ROTATE) {  // pairwise.py:765-791
        float p = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const float phase = R.x[2][i] / m.phase_div;
            float sn, cs;
            sincosf(phase, &sn, &cs);
            sv.wh[i] = cs; sv.aux[i] = sn;
            const float re = R.x[0][i] * cs - R.x[1][i] * sn - R.x[3][i];
            const float im = R.x[0][i] * sn + R.x[1][i] * cs - R.x[4][i];
            sv.a[i] = re; sv.c[i] = im;
            p += re * re + im * im;
        }
        return -(m.margin - gsum<G>(p));
    } else if constexpr (M == KGE_DISTMULT) {  // pointwise.py:444-446
        float p = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) p += R.x[0][i] * R.x[1][i] * R.x[2][i];
        return -gsum<G>(p);
    } else if constexpr (M == KGE_COMPLEX) {  // pointwise.py:185-188
        float p = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const float hr = R.x[0][i], hi = R.x[1][i], rr = R.x[2][i], ri = R.x[3][i], tr = R.x[4][i], ti = R.x[5][i];
            p += hr * tr * rr + hi * ti * rr + hr * ti * ri - hi * tr * ri;
        }
        return -gsum<G>(p);
    } else if constexpr (M == KGE_ANALOGY) {  // pointwise.py:97-104
        float p = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const float hr = R.x[3][i], hi = R.x[4][i], rr = R.x[5][i], ri = R.x[6][i], tr = R.x[7][i], ti = R.x[8][i];
            p += hr * tr * rr + hi * ti * rr + hr * ti * ri - hi * tr * ri;
            q += R.x[0][i] * R.x[1][i] * R.x[2][i];
        }
        gsum2<G>(p, q);
        return (-p) + (-q);
    } else if constexpr (M == KGE_TRANSM) {  // pairwise.py:325-347: theta_r * TransE distance
        sv.ph = gsum<G>(R.x[3][0]);          // theta_r sits in lane 0 of the group, zeros elsewhere
        return sv.ph * tail_fwd<G, NCH>(R.x[0], R.x[1], R.x[2], m.l1, sv.tail);
    } else if constexpr (M == KGE_CP) {      // pointwise.py:374-376
        float p = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) p += R.x[0][i] * R.x[1][i] * R.x[2][i];
        return -gsum<G>(p);
    } else if constexpr (M == KGE_SIMPLE || M == KGE_SIMPLE_IGNR) {  // pointwise.py:522-526, 581-585
        float p = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            p += R.x[0][i] * R.x[2][i] * R.x[4][i];   // <head[h], rel[r], tail[t]>
            q += R.x[1][i] * R.x[3][i] * R.x[5][i];   // <head[t], rel_inv[r], tail[h]>
        }
        gsum2<G>(p, q);
        const float init = (M == KGE_SIMPLE) ? p + q / 2.0f : p + q;
        sv.fw = init >= -20.f && init <= 20.f;        // torch.clamp backward: gradient inside the closed interval
        return -fminf(fmaxf(init, -20.f), 20.f);
    } else if constexpr (M == KGE_QUATE) {  // pointwise.py:683-700: (h (x) r/|r|) . t with elementwise quaternions
        float p = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const float hs = R.x[0][i], hx = R.x[1][i], hy = R.x[2][i], hz = R.x[3][i];
            const float ts = R.x[4][i], tx = R.x[5][i], ty = R.x[6][i], tz = R.x[7][i];
            const float rs = R.x[8][i], rx = R.x[9][i], ry = R.x[10][i], rz = R.x[11][i];
            const float den = sqrtf(rs * rs + rx * rx + ry * ry + rz * rz);
            // padding lanes (all-zero rows) would give 0/0: they contribute nothing, keep them at zero
            const float inv = den > 0.f ? 1.0f / den : 0.f;
            const float ps = rs * inv, px = rx * inv, py = ry * inv, pz = rz * inv;
            sv.a[i] = ps; sv.c[i] = px; sv.wh[i] = py; sv.aux[i] = pz;
            const float a = hs * ps - hx * px - hy * py - hz * pz;
            const float b = hs * px + ps * hx + hy * pz - py * hz;
            const float c = hs * py + ps * hy + hz * px - pz * hx;
            const float d = hs * pz + ps * hz + hx * py - px * hy;
            p += a * ts + b * tx + c * ty + d * tz;
        }
        return -gsum<G>(p);
    }
    return 0.f;
}

// Gr = d(ds * score)/d rows
template <int M, int G, int NCH>
__device__ __forceinline__ void model_bwd(const Rows<M, NCH>& R, const DeviceModel& m, const Saved<M, NCH>& sv,
                                          float ds, Rows<M, NCH>& Gr) {
    if constexpr (M == KGE_TRANSE) {
        tail_bwd<G, NCH>(R.x[0], R.x[1], R.x[2], m.l1, sv.tail, ds, Gr.x[0], Gr.x[1], Gr.x[2]);
    } else if constexpr (M == KGE_TRANSH) {
        float ga[NCH], gc[NCH];
        tail_bwd<G, NCH>(sv.a, R.x[1], sv.c, m.l1, sv.tail, ds, ga, Gr.x[1], gc);
        float gaw = 0.f, gcw = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            gaw = fmaf(ga[i], sv.wh[i], gaw);
            gcw = fmaf(gc[i], sv.wh[i], gcw);
        }
        gsum2<G>(gaw, gcw);
        float gwh[NCH];
        float dw = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            Gr.x[0][i] = ga[i] - gaw * sv.wh[i];
            Gr.x[2][i] = gc[i] - gcw * sv.wh[i];
            gwh[i] = -(sv.ph * ga[i] + gaw * R.x[0][i]) - (sv.pt * gc[i] + gcw * R.x[2][i]);
            dw = fmaf(sv.wh[i], gwh[i], dw);
        }
        dw = gsum<G>(dw);
#pragma unroll
        for (int i = 0; i < NCH; ++i)
            Gr.x[3][i] = sv.fw ? (gwh[i] - sv.wh[i] * dw) * sv.iw : gwh[i] * sv.iw;
    } else if constexpr (M == KGE_TRANSD) {
        float ga[NCH], gc[NCH];
        tail_bwd<G, NCH>(sv.a, R.x[1], sv.c, m.l1, sv.tail, ds, ga, Gr.x[1], gc);
        float gar = 0.f, gcr = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            gar = fmaf(ga[i], R.x[4][i], gar);
            gcr = fmaf(gc[i], R.x[4][i], gcr);
        }
        gsum2<G>(gar, gcr);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            Gr.x[0][i] = ga[i] + gar * R.x[3][i];
            Gr.x[3][i] = gar * R.x[0][i];
            Gr.x[2][i] = gc[i] + gcr * R.x[5][i];
            Gr.x[5][i] = gcr * R.x[2][i];
            Gr.x[4][i] = sv.ph * ga[i] + sv.pt * gc[i];
        }
    } else if constexpr (M == KGE_ROTATE) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const float cs = sv.wh[i], sn = sv.aux[i];
            const float Rr = 2.f * ds * sv.a[i], Ii = 2.f * ds * sv.c[i];
            const float hr = R.x[0][i], hi = R.x[1][i];
            Gr.x[0][i] = Rr * cs + Ii * sn;
            Gr.x[1][i] = -Rr * sn + Ii * cs;
            Gr.x[3][i] = -Rr;
            Gr.x[4][i] = -Ii;
            Gr.x[2][i] = (Rr * (-hr * sn - hi * cs) + Ii * (hr * cs - hi * sn)) / m.phase_div;
        }
    } else if constexpr (M == KGE_DISTMULT) {
        const float nds = -ds;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            Gr.x[0][i] = R.x[1][i] * R.x[2][i] * nds;
            Gr.x[1][i] = R.x[0][i] * R.x[2][i] * nds;
            Gr.x[2][i] = R.x[0][i] * R.x[1][i] * nds;
        }
    } else if constexpr (M == KGE_COMPLEX || M == KGE_ANALOGY) {
        constexpr int o = (M == KGE_ANALOGY) ? 3 : 0;
        const float nds = -ds;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const float hr = R.x[o + 0][i], hi = R.x[o + 1][i], rr = R.x[o + 2][i], ri = R.x[o + 3][i],
                        tr = R.x[o + 4][i], ti = R.x[o + 5][i];
            Gr.x[o + 0][i] = (tr * rr + ti * ri) * nds;
            Gr.x[o + 1][i] = (ti * rr - tr * ri) * nds;
            Gr.x[o + 2][i] = (hr * tr + hi * ti) * nds;
            Gr.x[o + 3][i] = (hr * ti - hi * tr) * nds;
            Gr.x[o + 4][i] = (hr * rr - hi * ri) * nds;
            Gr.x[o + 5][i] = (hi * rr + hr * ri) * nds;
            if constexpr (M == KGE_ANALOGY) {
                Gr.x[0][i] = R.x[1][i] * R.x[2][i] * nds;
                Gr.x[1][i] = R.x[0][i] * R.x[2][i] * nds;
                Gr.x[2][i] = R.x[0][i] * R.x[1][i] * nds;
            }
        }
    } else if constexpr (M == KGE_TRANSM) {
        tail_bwd<G, NCH>(R.x[0], R.x[1], R.x[2], m.l1, sv.tail, ds * sv.ph, Gr.x[0], Gr.x[1], Gr.x[2]);
#pragma unroll
        for (int i = 0; i < NCH; ++i) Gr.x[3][i] = 0.f;  // theta is a fixed input
    } else if constexpr (M == KGE_CP) {
        const float nds = -ds;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            Gr.x[0][i] = R.x[1][i] * R.x[2][i] * nds;
            Gr.x[1][i] = R.x[0][i] * R.x[2][i] * nds;
            Gr.x[2][i] = R.x[0][i] * R.x[1][i] * nds;
        }
    } else if constexpr (M == KGE_SIMPLE || M == KGE_SIMPLE_IGNR) {
        const float g1 = sv.fw ? -ds : 0.f;
        const float g2 = (M == KGE_SIMPLE) ? g1 / 2.0f : g1;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            Gr.x[0][i] = R.x[2][i] * R.x[4][i] * g1;
            Gr.x[2][i] = R.x[0][i] * R.x[4][i] * g1;
            Gr.x[4][i] = R.x[0][i] * R.x[2][i] * g1;
            Gr.x[1][i] = R.x[3][i] * R.x[5][i] * g2;
            Gr.x[3][i] = R.x[1][i] * R.x[5][i] * g2;
            Gr.x[5][i] = R.x[1][i] * R.x[3][i] * g2;
        }
    } else if constexpr (M == KGE_QUATE) {
        const float g = -ds;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const float hs = R.x[0][i], hx = R.x[1][i], hy = R.x[2][i], hz = R.x[3][i];
            const float ts = R.x[4][i], tx = R.x[5][i], ty = R.x[6][i], tz = R.x[7][i];
            const float rs = R.x[8][i], rx = R.x[9][i], ry = R.x[10][i], rz = R.x[11][i];
            const float ps = sv.a[i], px = sv.c[i], py = sv.wh[i], pz = sv.aux[i];
            // d/d t = Hamilton product h (x) r^
            Gr.x[4][i] = (hs * ps - hx * px - hy * py - hz * pz) * g;
            Gr.x[5][i] = (hs * px + ps * hx + hy * pz - py * hz) * g;
            Gr.x[6][i] = (hs * py + ps * hy + hz * px - pz * hx) * g;
            Gr.x[7][i] = (hs * pz + ps * hz + hx * py - px * hy) * g;
            // d/d h
            Gr.x[0][i] = (ps * ts + px * tx + py * ty + pz * tz) * g;
            Gr.x[1][i] = (-px * ts + ps * tx - pz * ty + py * tz) * g;
            Gr.x[2][i] = (-py * ts + pz * tx + ps * ty - px * tz) * g;
            Gr.x[3][i] = (-pz * ts - py * tx + px * ty + ps * tz) * g;
            // d/d r^ , then through the per-element normalisation r^ = r / |r|
            const float gs = (hs * ts + hx * tx + hy * ty + hz * tz) * g;
            const float gx = (-hx * ts + hs * tx + hz * ty - hy * tz) * g;
            const float gy = (-hy * ts - hz * tx + hs * ty + hx * tz) * g;
            const float gz = (-hz * ts + hy * tx - hx * ty + hs * tz) * g;
            const float den = sqrtf(rs * rs + rx * rx + ry * ry + rz * rz);
            const float inv = den > 0.f ? 1.0f / den : 0.f;
            const float dot = ps * gs + px * gx + py * gy + pz * gz;
            Gr.x[8][i] = (gs - ps * dot) * inv;
            Gr.x[9][i] = (gx - px * dot) * inv;
            Gr.x[10][i] = (gy - py * dot) * inv;
            Gr.x[11][i] = (gz - pz * dot) * inv;
        }
    }
}

// block-level accumulation of a per-group scalar (value valid in every lane of the group; counted once per group)
template <int G>
__device__ __forceinline__ void block_accumulate_loss(float v_group, int gl, float* __restrict__ loss) {
    __shared__ float s_part[4];  // 256 threads = 4 waves
    float v = (gl == 0) ? v_group : 0.f;
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) s_part[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        if (tot != 0.f) unsafeAtomicAdd(loss + (blockIdx.x % kLossSlots) * kLossStride, tot);
    }
}

}  // namespace kge
