// kge_own.hip -- the owner-computes training step of the POINTWISE models (DistMult, ComplEx / ComplexN3), two phases, no float
// atomics, no dense gradient sweep, bit-reproducible.  One step =
//     Generator (data/generator.py:99-158, neg_rate 1) + Trainer.train_step_pointwise (utils/trainer.py:176-180) +
//     Criterion.pointwise_logistic (utils/criterion.py:31-34) + get_reg (pointwise.py:190-202,224-238,448-458) +
//     loss.backward() + optimizer.step() (utils/trainer.py:298-299, 112-131)
// in two launches:
//   phase 1  k_own_step   every touched parameter row has ONE owner group which re-evaluates the bundles the row takes part in
//                         -- a bundle = the positive (h, r, t) and its sampled corruption (h, r, c) or (c, r, t) -- gathers only the
//                         partner rows it needs, and sums the row's gradient in registers in a fixed order; the gradient ROW is
//                         stored once (plain stores) into a buffer of the tables' shape.  Entity owners hold the re and im rows
//                         of their entity together (one incidence list serves both tables), relation owners likewise.  The
//                         sampler of the NEXT batch rides in the leading blocks, exactly as in kge_pull.hip.
//   phase 2  k_own_apply  the dense-semantics optimiser on the rows that have a gradient: SGD / Adagrad leave every other row
//                         untouched (a zero dense gradient changes nothing), Adam / RMSprop visit every row.
// Why two phases when TransE's owner-computes step (kge_pull.hip) needs one: that step writes every row of a double-buffered
// table each step, which the dense Adam of the headline config does anyway; the pointwise presets train with Adagrad on tables
// of 65 MB (config C2) of which a step touches ~15 %, so rows are updated IN PLACE and no owner may read a partner row that
// another owner has already moved -- hence gradients first, updates second.
// The incidence index, the per-step sampler lists and the work-item kinds are those of kge_pull.hip (kge_pull_index_build).
#include "kge_pull_device.h"
#include "kge_opt_device.h"
#include <stdlib.h>

namespace kge {

struct OwnArgs {
    const float* ent[2]; const float* rel[2];     // parameter tables (NT per class: ComplEx re, im)
    float* g_ent[2]; float* g_rel[2];             // gradient rows out, same row layout
    float* s1_ent[2]; float* s1_rel[2];           // optimiser state (phase 2)
    float* s2_ent[2]; float* s2_rel[2];
    const int4* pairs; PullLists lists;
    const int4* items; const int32_t* inc; float* partials; const int4* multi;
    int64_t n_items, n_multi;
    const uint32_t* listed;    // bitmap of the rows with explicit items (compact index); NULL: every row has an item
    int n_rows, n_pairs;
    int dense;                 // 1: every unlisted row is an implicit owner (Adam / RMSprop); 0: only the entities drawn this step
    int sample_blocks, E, d, reset_lists;
    float inv_n, lmbda; int reg_type;
    OptArgs opt;
};

// the rows of one visit that the owner does not hold itself
template <int NT, int NV>
struct OwnRows {
    float4 hh[NT][NV], tt[NT][NV], cc[NT][NV], rr[NT][NV];
    int w;   // corrupting entity | tail << 24 | role << 25
};

template <int NT, int G, int NV>
__device__ __forceinline__ void load_rows_nt(float4 (&x)[NT][NV], const float* const (&tab)[2], int64_t row, int d, int nvec, int gl) {
#pragma unroll
    for (int t = 0; t < NT; ++t) load_row4<G, NV>(x[t], tab[t] + row * d, nvec, gl);
}

// -(score) partial of one lane: sum over its elements of Re(<a, r, conj b>) (ComplEx) or a*r*b (DistMult)
template <int NT, int NV>
__device__ __forceinline__ float dot3(const float4 (&a)[NT][NV], const float4 (&r)[NT][NV], const float4 (&b)[NT][NV]) {
    float p = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
#define KGE_D3(c)                                                                                                       \
        if constexpr (NT == 1) p = fmaf(a[0][v].c * r[0][v].c, b[0][v].c, p);                                             \
        else {                                                                                                            \
            const float ar = a[0][v].c, ai = a[1][v].c, rr_ = r[0][v].c, ri = r[1][v].c, br = b[0][v].c, bi = b[1][v].c;  \
            p += ar * br * rr_ + ai * bi * rr_ + ar * bi * ri - ai * br * ri;                                             \
        }
        KGE_D3(x) KGE_D3(y) KGE_D3(z) KGE_D3(w)
#undef KGE_D3
    }
    return p;
}

// g += k * d(-score)/d(position) ; position 0 = head (uses r, b), 1 = tail (uses a, r), 2 = relation (uses a, b)
template <int NT, int NV>
__device__ __forceinline__ void add_grad(float4 (&g)[NT][NV], int pos, float k, const float4 (&a)[NT][NV], const float4 (&r)[NT][NV],
                                         const float4 (&b)[NT][NV]) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
#define KGE_AG(c)                                                                                                       \
        if constexpr (NT == 1) {                                                                                          \
            const float x1 = pos == 0 ? r[0][v].c : a[0][v].c, x2 = pos == 1 ? r[0][v].c : b[0][v].c;                      \
            g[0][v].c = fmaf(k, x1 * x2, g[0][v].c);                                                                      \
        } else {                                                                                                          \
            const float ar = a[0][v].c, ai = a[1][v].c, rr_ = r[0][v].c, ri = r[1][v].c, br = b[0][v].c, bi = b[1][v].c;  \
            float gre, gim;                                                                                               \
            if (pos == 0) { gre = br * rr_ + bi * ri; gim = bi * rr_ - br * ri; }                                         \
            else if (pos == 1) { gre = ar * rr_ - ai * ri; gim = ai * rr_ + ar * ri; }                                    \
            else { gre = ar * br + ai * bi; gim = ar * bi - ai * br; }                                                    \
            g[0][v].c = fmaf(k, gre, g[0][v].c); g[1][v].c = fmaf(k, gim, g[1][v].c);                                     \
        }
        KGE_AG(x) KGE_AG(y) KGE_AG(z) KGE_AG(w)
#undef KGE_AG
    }
}

// regulariser of one row set: value partial (sum of x^2 | x^3 | |x|^3 over this lane's elements)
template <int NT, int NV>
__device__ __forceinline__ float reg_value(const float4 (&x)[NT][NV], int reg_type) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int v = 0; v < NV; ++v) {
#define KGE_RV(c) { const float q = x[t][v].c; s += reg_type == KGE_REG_F2 ? q * q : (reg_type == KGE_REG_N3 ? q * q * q : fabsf(q) * q * q); }
            KGE_RV(x) KGE_RV(y) KGE_RV(z) KGE_RV(w)
#undef KGE_RV
        }
    return s;
}
// g += k * d reg / d x  (k already carries lmbda / n and the number of occurrences)
template <int NT, int NV>
__device__ __forceinline__ void reg_grad(float4 (&g)[NT][NV], const float4 (&x)[NT][NV], float k, int reg_type) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int v = 0; v < NV; ++v) {
#define KGE_RG(c) { const float q = x[t][v].c;                                                                         \
                    const float dq = reg_type == KGE_REG_F2 ? 2.f * q : (reg_type == KGE_REG_N3 ? 3.f * q * q : 3.f * q * fabsf(q)); \
                    g[t][v].c = fmaf(k, dq, g[t][v].c); }
            KGE_RG(x) KGE_RG(y) KGE_RG(z) KGE_RG(w)
#undef KGE_RG
        }
}

template <int NT, int G, int NV>
__global__ __launch_bounds__(kBlock) void k_own_step(OwnArgs a, PullSampleArgs sa, float* __restrict__ loss) {
    constexpr int GPB = kBlock / G;
    if ((int)blockIdx.x < a.sample_blocks) {   // leading blocks: the sampler of the NEXT batch rides along (other list set)
        const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
        if (i < sa.n) pull_sample_one(sa, i);
        return;
    }
    const int gl = threadIdx.x % G;
    const int grp = threadIdx.x / G;
    const int gbase = (threadIdx.x & 63) / G * G;
    const int d = a.d, nvec = a.d >> 2;
    const int64_t item = (int64_t)((int)blockIdx.x - a.sample_blocks) * GPB + grp;
    __shared__ float4 s_part[GPB][NT * NV * G];
    __shared__ int4 s_desc[GPB][G];
    float acc = 0.f;
    int4 it = make_int4(-1, 0, 0, 0);
    if (item < a.n_items) it = a.items[item];
    else if (a.listed != nullptr) {
        const int64_t j = item - a.n_items;
        if (a.dense) {            // every row without an explicit item: its corrupting-entity draws, or a zero gradient row
            if (j < a.n_rows && !((a.listed[j >> 5] >> (j & 31)) & 1u)) it = make_int4((int)j, 0, 0, 0);
        } else if (j < a.n_pairs) {   // sparse: pair j stands in as the owner of the entity it drew, if it was the first to draw it
            const int w = a.lists.pc[j];
            const int c = w & 0xFFFFFF;
            if (((w >> kPcFirstBit) & 1) && !((a.listed[c >> 5] >> (c & 31)) & 1u)) it = make_int4(c, 0, 0, 0);
        }
    }
    const int g = it.x;
    const int kind = it.w & 3;
    float4 X[NT][NV], gs[NT][NV];
    if (g >= 0) {
        const bool is_rel = g >= a.E;
        const int64_t own = is_rel ? g - a.E : g;
        const int n_static = it.z - it.y;
        int cnt = 0, nvis = 0;
        int vi = -1, vrole = 0, slot = gl;
        if (gl < n_static) { const int e = a.inc[it.y + gl]; vi = e >> 2; vrole = e & 3; }
        const bool walks_c = !is_rel && (kind == 0 || kind == 1 || (kind == 3 && ((it.w >> 2) & 15) == 0));
        if (walks_c) cnt = a.lists.count[g];
        const bool fast_c = cnt <= kPullCap && n_static + cnt <= G;
        nvis = n_static;
        if (cnt > 0 && fast_c) {
            const int q = gl - n_static;
            if (q >= 0 && q < cnt) { vi = a.lists.bucket[(int64_t)g * kPullCap + q]; vrole = kRoleC; }
            if (cnt > 1) {   // arrival order is arbitrary: rank the entries by pair index, visit by rank
                int rank = 0;
                for (int m = 0; m < cnt; ++m) rank += __shfl(vi, gbase + n_static + m, 64) < vi ? 1 : 0;
                if (q >= 0 && q < cnt) slot = n_static + rank;
            }
            nvis += cnt;
        }
        if (vi >= 0) {
            int4 pr = a.pairs[vi];
            pr.w = (a.lists.pc[vi] & 0x1FFFFFF) | (vrole << 25);
            s_desc[grp][slot] = pr;
        }
        if (is_rel) load_rows_nt<NT, G, NV>(X, a.rel, own, d, nvec, gl);
        else load_rows_nt<NT, G, NV>(X, a.ent, own, d, nvec, gl);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int v = 0; v < NV; ++v) gs[t][v] = make_float4(0.f, 0.f, 0.f, 0.f);

        // which of the four row sets a visit needs besides the owner's own one.  P = (h, r, t); N = (h, r, c) when the tail was
        // corrupted, (c, r, t) otherwise.  An entity owner only needs the partners of the triples it occurs in.
        auto fetch = [&](int h, int r, int t, int w, OwnRows<NT, NV>& b) {
            b.w = w;
            const int role = (w >> 25) & 3;
            const bool tail = ((w >> 24) & 1) != 0;
            const int c = w & 0xFFFFFF;
            const bool needH = role == kRoleR || role == kRoleT || (role == kRoleC && tail);
            const bool needT = role == kRoleR || role == kRoleH || (role == kRoleC && !tail);
            const bool needC = role == kRoleR || (role == kRoleH && tail) || (role == kRoleT && !tail);
            const bool needR = role != kRoleR;
            if (needH) load_rows_nt<NT, G, NV>(b.hh, a.ent, h, d, nvec, gl);
            if (needT) load_rows_nt<NT, G, NV>(b.tt, a.ent, t, d, nvec, gl);
            if (needC) load_rows_nt<NT, G, NV>(b.cc, a.ent, c, d, nvec, gl);
            if (needR) load_rows_nt<NT, G, NV>(b.rr, a.rel, r, d, nvec, gl);
        };
        auto fetch_visit = [&](int v, OwnRows<NT, NV>& b) {
            const int4 ds = s_desc[grp][v];
            fetch(ds.x, ds.y, ds.z, ds.w, b);
        };
        auto compute = [&](OwnRows<NT, NV>& b) {
            const int role = (b.w >> 25) & 3;
            const bool tail = ((b.w >> 24) & 1) != 0;
            // the owner's rows take their place among the four
            if (role == kRoleH) { for (int t = 0; t < NT; ++t) for (int v = 0; v < NV; ++v) b.hh[t][v] = X[t][v]; }
            else if (role == kRoleT) { for (int t = 0; t < NT; ++t) for (int v = 0; v < NV; ++v) b.tt[t][v] = X[t][v]; }
            else if (role == kRoleC) { for (int t = 0; t < NT; ++t) for (int v = 0; v < NV; ++v) b.cc[t][v] = X[t][v]; }
            else { for (int t = 0; t < NT; ++t) for (int v = 0; v < NV; ++v) b.rr[t][v] = X[t][v]; }
            const bool inP = role != kRoleC;
            const bool inN = role == kRoleR || role == kRoleC || (role == kRoleH && tail) || (role == kRoleT && !tail);
            float pP = 0.f, pN = 0.f, rs = 0.f;
            if (inP) pP = dot3<NT, NV>(b.hh, b.rr, b.tt);
            if (inN) pN = tail ? dot3<NT, NV>(b.hh, b.rr, b.cc) : dot3<NT, NV>(b.cc, b.rr, b.tt);
            const bool reg_on = a.reg_type != KGE_REG_NONE;
            if (role == kRoleR && reg_on)   // the relation owner sees every row of both triples: it accounts for the loss terms
                rs = reg_value<NT, NV>(b.hh, a.reg_type) + reg_value<NT, NV>(b.tt, a.reg_type) + reg_value<NT, NV>(b.cc, a.reg_type) +
                     2.f * reg_value<NT, NV>(b.rr, a.reg_type) + (tail ? reg_value<NT, NV>(b.hh, a.reg_type) : reg_value<NT, NV>(b.tt, a.reg_type));
            gsum3<G>(pP, pN, rs);
            const float sP = -pP, sN = -pN;                      // energies
            // loss = mean softplus(y s): y = +1 for P, -1 for N (utils/criterion.py:31-34, utils/trainer.py:178)
            const float dP = inP ? sigmoid_t(sP) * a.inv_n : 0.f;          // d loss / d sP
            const float dN = inN ? -sigmoid_t(-sN) * a.inv_n : 0.f;        // d loss / d sN
            if (role == kRoleR) acc += (softplus_t(sP) + softplus_t(-sN)) * a.inv_n + a.lmbda * a.inv_n * rs;
            // own position in P and in N (0 head, 1 tail, 2 relation); d s / d row = -(d dot3 / d row)
            const int posP = role == kRoleR ? 2 : role;            // H -> 0, T -> 1 (C: not in P)
            const int posN = role == kRoleR ? 2 : (role == kRoleC ? (tail ? 1 : 0) : role);
            if (inP) add_grad<NT, NV>(gs, posP, -dP, b.hh, b.rr, b.tt);
            if (inN) { if (tail) add_grad<NT, NV>(gs, posN, -dN, b.hh, b.rr, b.cc); else add_grad<NT, NV>(gs, posN, -dN, b.cc, b.rr, b.tt); }
            if (reg_on) reg_grad<NT, NV>(gs, X, a.lmbda * a.inv_n * (float)((inP ? 1 : 0) + (inN ? 1 : 0)), a.reg_type);
        };
        if (nvis > 0) {   // software pipeline: the gathers of visit v+1 are in flight while visit v is evaluated
            OwnRows<NT, NV> ba, bb;
            fetch_visit(0, ba);
            for (int v = 0; v < nvis; v += 2) {
                const bool more = v + 1 < nvis;
                if (more) fetch_visit(v + 1, bb);
                compute(ba);
                if (more) {
                    if (v + 2 < nvis) fetch_visit(v + 2, ba);
                    compute(bb);
                }
            }
        }
        if (cnt > 0 && !fast_c) {   // more drawers than the bucket / the lane group holds: ascending pair order, one at a time
            const int nb = cnt < kPullCap ? cnt : kPullCap;
            int last = -1;
            for (;;) {
                int best = 0x7FFFFFFF;
                for (int m = 0; m < nb; ++m) { const int j = a.lists.bucket[(int64_t)g * kPullCap + m]; if (j > last && j < best) best = j; }
                for (int j = a.lists.head[g]; j >= 0; j = a.lists.next[j]) if (j > last && j < best) best = j;
                if (best == 0x7FFFFFFF) break;
                const int4 p2 = a.pairs[best];
                OwnRows<NT, NV> b;
                fetch(p2.x, p2.y, p2.z, (a.lists.pc[best] & 0x1FFFFFF) | (kRoleC << 25), b);
                compute(b);
                last = best;
            }
        }
        if (cnt > 0 && a.reset_lists && gl == 0) {
            a.lists.count[g] = 0;
            if (cnt > kPullCap) a.lists.head[g] = -1;
        }
        if (kind == 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t) store_row4<G, NV>((is_rel ? a.g_rel[t] : a.g_ent[t]) + own * d, gs[t], nvec, gl);
        } else if (kind == 3) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int v = 0; v < NV; ++v) s_part[grp][(t * NV + v) * G + gl] = gs[t][v];
        } else {
            float4* out = reinterpret_cast<float4*>(a.partials) + (int64_t)(it.w >> 2) * (NT * NV * G);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int v = 0; v < NV; ++v) out[(t * NV + v) * G + gl] = gs[t][v];
        }
    }
    __syncthreads();
    if (g >= 0 && kind == 3 && ((it.w >> 2) & 15) == 0) {   // first item of a workgroup-local row: add the others in segment order
        const int nseg = it.w >> 6;
        const bool is_rel = g >= a.E;
        const int64_t own = is_rel ? g - a.E : g;
        for (int m = 1; m < nseg; ++m) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const float4 p = s_part[grp + m][(t * NV + v) * G + gl];
                    gs[t][v].x += p.x; gs[t][v].y += p.y; gs[t][v].z += p.z; gs[t][v].w += p.w;
                }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) store_row4<G, NV>((is_rel ? a.g_rel[t] : a.g_ent[t]) + own * d, gs[t], nvec, gl);
    }
    block_accumulate_loss<G>(acc, gl, loss);
}

// phase 2: the optimiser on every row that has a gradient row (or a list of partial sums)
template <int OPT, int NT, int G, int NV>
__global__ __launch_bounds__(kBlock) void k_own_apply(OwnArgs a) {
    constexpr int GPB = kBlock / G;
    const int gl = threadIdx.x % G;
    const int64_t unit = (int64_t)blockIdx.x * GPB + threadIdx.x / G;
    const int d = a.d, nvec = a.d >> 2;
    int g = -1, slot0 = 0, nslots = 0;
    if (unit < a.n_items) {
        const int4 it = a.items[unit];
        const int kind = it.w & 3;
        if (it.x >= 0 && (kind == 0 || (kind == 3 && ((it.w >> 2) & 15) == 0))) g = it.x;
    } else if (unit < a.n_items + a.n_multi) {
        const int4 row = a.multi[unit - a.n_items];
        g = row.x; slot0 = row.y; nslots = row.z;
    } else if (a.listed != nullptr) {
        const int64_t j = unit - a.n_items - a.n_multi;
        if (a.dense) {
            if (j < a.n_rows && !((a.listed[j >> 5] >> (j & 31)) & 1u)) g = (int)j;
        } else if (j < a.n_pairs) {
            const int w = a.lists.pc[j];
            const int c = w & 0xFFFFFF;
            if (((w >> kPcFirstBit) & 1) && !((a.listed[c >> 5] >> (c & 31)) & 1u)) g = c;
        }
    }
    if (g < 0) return;
    const bool is_rel = g >= a.E;
    const int64_t off = (int64_t)(is_rel ? g - a.E : g) * d;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float* const p = const_cast<float*>(is_rel ? a.rel[t] : a.ent[t]) + off;
        float* const s1 = (is_rel ? a.s1_rel[t] : a.s1_ent[t]);
        float* const s2 = (is_rel ? a.s2_rel[t] : a.s2_ent[t]);
        float4 gv[NV], P[NV], M1[NV], M2[NV];
        if (nslots > 0) {   // rows cut into several items: partial sums added in segment order
#pragma unroll
            for (int v = 0; v < NV; ++v) gv[v] = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int s = 0; s < nslots; ++s) {
                const float4* in = reinterpret_cast<const float4*>(a.partials) + (int64_t)(slot0 + s) * (NT * NV * G);
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const float4 q = in[(t * NV + v) * G + gl];
                    gv[v].x += q.x; gv[v].y += q.y; gv[v].z += q.z; gv[v].w += q.w;
                }
            }
        } else {
            load_row4<G, NV>(gv, (is_rel ? a.g_rel[t] : a.g_ent[t]) + off, nvec, gl);
        }
        load_row4<G, NV>(P, p, nvec, gl);
        if constexpr (OPT != KGE_OPT_SGD) load_row4<G, NV>(M1, s1 + off, nvec, gl);
        if constexpr (OPT == KGE_OPT_ADAM) load_row4<G, NV>(M2, s2 + off, nvec, gl);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
#define KGE_UP(c) { float m1 = 0.f, m2 = 0.f;                                                        \
                    if constexpr (OPT != KGE_OPT_SGD) m1 = M1[v].c;                                    \
                    if constexpr (OPT == KGE_OPT_ADAM) m2 = M2[v].c;                                   \
                    opt_update<OPT>(P[v].c, gv[v].c, m1, m2, a.opt);                                   \
                    if constexpr (OPT != KGE_OPT_SGD) M1[v].c = m1;                                    \
                    if constexpr (OPT == KGE_OPT_ADAM) M2[v].c = m2; }
            KGE_UP(x) KGE_UP(y) KGE_UP(z) KGE_UP(w)
#undef KGE_UP
        }
        store_row4<G, NV>(p, P, nvec, gl);
        if constexpr (OPT != KGE_OPT_SGD) store_row4<G, NV>(s1 + off, M1, nvec, gl);
        if constexpr (OPT == KGE_OPT_ADAM) store_row4<G, NV>(s2 + off, M2, nvec, gl);
    }
}

// ------------------------------------------------------------------ host side
struct OwnGeo { int NT, G, NV; };
static OwnGeo own_geo(int model, int dim) {
    OwnGeo g{0, 0, 0};
    if (dim <= 0 || (dim & 3) || dim > 512) return g;
    g.NT = model == KGE_COMPLEX ? 2 : (model == KGE_DISTMULT ? 1 : 0);
    if (!g.NT) return g;
    const int nvec = dim >> 2;
    if (nvec <= 32) { g.G = 32; g.NV = 1; }
    else if (nvec <= 64) { g.G = 64; g.NV = 1; }
    else { g.G = 64; g.NV = 2; }
    return g;
}

static int fill_own_args(const kge_model_desc* m, float* const* state1, float* const* state2, OwnArgs* a, OwnGeo* geo, const char* who) {
    *geo = own_geo(m->model, m->dim);
    if (!geo->G) { set_error("%s: DistMult / ComplEx with a hidden size that is a multiple of 4 and at most 512 (model %d, dim %d)", who, m->model, m->dim); return -1; }
    const int NT = geo->NT;
    for (int t = 0; t < 2; ++t) {
        const bool on = t < NT;
        a->ent[t] = on ? m->tables[t] : nullptr;           a->rel[t] = on ? m->tables[NT + t] : nullptr;
        a->g_ent[t] = on ? m->grads[t] : nullptr;          a->g_rel[t] = on ? m->grads[NT + t] : nullptr;
        a->s1_ent[t] = (on && state1) ? state1[t] : nullptr; a->s1_rel[t] = (on && state1) ? state1[NT + t] : nullptr;
        a->s2_ent[t] = (on && state2) ? state2[t] : nullptr; a->s2_rel[t] = (on && state2) ? state2[NT + t] : nullptr;
        if (on && (!a->ent[t] || !a->rel[t] || !a->g_ent[t] || !a->g_rel[t])) { set_error("%s: tables / gradient row buffers missing", who); return -1; }
        if (on && ((((uintptr_t)a->ent[t] | (uintptr_t)a->rel[t] | (uintptr_t)a->g_ent[t] | (uintptr_t)a->g_rel[t]) & 15))) {
            set_error("%s: tables and gradient buffers must be 16-byte aligned", who); return -1;
        }
    }
    a->E = (int)m->tot_entity; a->d = m->dim; a->n_rows = (int)(m->tot_entity + m->tot_relation);
    return 0;
}

#define KGE_OWN_GEO(BODY)                                                                          \
    if (geo.NT == 1 && geo.G == 32) { constexpr int NT = 1, G = 32, NV = 1; BODY }                  \
    else if (geo.NT == 1 && geo.NV == 1) { constexpr int NT = 1, G = 64, NV = 1; BODY }            \
    else if (geo.NT == 1) { constexpr int NT = 1, G = 64, NV = 2; BODY }                           \
    else if (geo.G == 32) { constexpr int NT = 2, G = 32, NV = 1; BODY }                           \
    else if (geo.NV == 1) { constexpr int NT = 2, G = 64, NV = 1; BODY }                           \
    else { constexpr int NT = 2, G = 64, NV = 2; BODY }

static int64_t own_extra_units(const OwnArgs& a) { return a.listed ? (a.dense ? a.n_rows : a.n_pairs) : 0; }

int launch_own_step(const kge_model_desc* m, const int32_t* pairs, int64_t n_pairs, const kge_pull_lists* lists, const int32_t* items,
                    int64_t n_items, const uint32_t* listed, const int32_t* inc, float* partials, int dense, float lmbda, int reg_type,
                    int reset_lists, const int32_t* next_pairs, int64_t next_n, const float* bern, const uint64_t* slots,
                    int64_t n_slots, uint64_t seed, uint64_t next_offset, const kge_pull_lists* next_lists, float* loss, hipStream_t s) {
    OwnArgs a;
    OwnGeo geo;
    if (fill_own_args(m, nullptr, nullptr, &a, &geo, "kge_own_step")) return -1;
    a.pairs = (const int4*)pairs; a.lists = to_lists(lists); a.items = (const int4*)items; a.inc = inc; a.partials = partials;
    a.multi = nullptr; a.n_items = n_items; a.n_multi = 0; a.listed = listed; a.n_pairs = (int)n_pairs; a.dense = dense ? 1 : 0;
    a.reset_lists = reset_lists; a.inv_n = 1.0f / (float)(2 * n_pairs); a.lmbda = lmbda; a.reg_type = reg_type;
    a.opt = make_opt_args(0.f, 1);
    const PullSampleArgs sa = make_sample_args(next_pairs, next_pairs && next_lists ? next_n : 0, m->tot_entity, bern, slots, n_slots,
                                               seed, next_offset, nullptr, next_lists);
    a.sample_blocks = sa.n > 0 ? (int)((sa.n + kBlock - 1) / kBlock) : 0;
    const int64_t units = n_items + own_extra_units(a);
    KGE_OWN_GEO({
        const int64_t blocks = (units + kBlock / G - 1) / (kBlock / G) + a.sample_blocks;
        hipLaunchKernelGGL((k_own_step<NT, G, NV>), dim3((unsigned)blocks), dim3(kBlock), 0, s, a, sa, loss);
    })
    return check_launch("k_own_step");
}

template <int OPT>
static int launch_own_apply_opt(OwnArgs& a, OwnGeo geo, hipStream_t s) {
    const int64_t units = a.n_items + a.n_multi + own_extra_units(a);
    KGE_OWN_GEO({
        const int64_t blocks = (units + kBlock / G - 1) / (kBlock / G);
        hipLaunchKernelGGL((k_own_apply<OPT, NT, G, NV>), dim3((unsigned)blocks), dim3(kBlock), 0, s, a);
    })
    return check_launch("k_own_apply");
}

int launch_own_apply(const kge_model_desc* m, float* const* state1, float* const* state2, const int32_t* pairs, int64_t n_pairs,
                     const kge_pull_lists* lists, const int32_t* items, int64_t n_items, const uint32_t* listed, const int32_t* multi,
                     int64_t n_multi, float* partials, int dense, int optimizer, float lr, int64_t step, hipStream_t s) {
    OwnArgs a;
    OwnGeo geo;
    if (fill_own_args(m, state1, state2, &a, &geo, "kge_own_apply")) return -1;
    const int NT = geo.NT;
    for (int t = 0; t < NT; ++t) {
        if (optimizer != KGE_OPT_SGD && (!a.s1_ent[t] || !a.s1_rel[t])) { set_error("kge_own_apply: optimizer state missing"); return -1; }
        if (optimizer == KGE_OPT_ADAM && (!a.s2_ent[t] || !a.s2_rel[t])) { set_error("kge_own_apply: adam needs two state buffers"); return -1; }
    }
    a.pairs = (const int4*)pairs; a.lists = to_lists(lists); a.items = (const int4*)items; a.inc = nullptr; a.partials = partials;
    a.multi = (const int4*)multi; a.n_items = n_items; a.n_multi = n_multi; a.listed = listed; a.n_pairs = (int)n_pairs;
    a.dense = dense ? 1 : 0; a.reset_lists = 0; a.inv_n = 0.f; a.lmbda = 0.f; a.reg_type = 0; a.sample_blocks = 0;
    a.opt = make_opt_args(lr, step < 1 ? 1 : step);
    switch (optimizer) {
        case KGE_OPT_SGD: return launch_own_apply_opt<KGE_OPT_SGD>(a, geo, s);
        case KGE_OPT_ADAM: return launch_own_apply_opt<KGE_OPT_ADAM>(a, geo, s);
        case KGE_OPT_ADAGRAD: return launch_own_apply_opt<KGE_OPT_ADAGRAD>(a, geo, s);
        case KGE_OPT_RMSPROP: return launch_own_apply_opt<KGE_OPT_RMSPROP>(a, geo, s);
    }
    set_error("kge_own_apply: unknown optimizer %d", optimizer);
    return -1;
}

int own_groups_per_block(int model, int dim) { const OwnGeo g = own_geo(model, dim); return g.G ? kBlock / g.G : 0; }
int own_partial_stride(int model, int dim) { const OwnGeo g = own_geo(model, dim); return g.G ? 4 * g.NT * g.NV * g.G : 0; }

}  // namespace kge
