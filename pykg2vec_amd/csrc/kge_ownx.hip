// kge_ownx.hip -- the staged owner-computes training step (kge_own.hip, staged form) for the remaining pointwise gather models:
// ANALOGY (pointwise.py:241-317), CP (:320-387), SimplE / SimplE_ignr (:461-592), QuatE (:595-768).  Same contract as kge_own_run:
//     Generator (data/generator.py:99-158, neg_rate 1) + Trainer.train_step_pointwise (utils/trainer.py:176-180) +
//     Criterion.pointwise_logistic (utils/criterion.py:31-34) + get_reg + loss.backward() + optimizer.step(), no float atomics.
// These models differ from DistMult / ComplEx only in which tables a triple's rows come from (the role tables of kge_device.h) and in
// their forward / backward (model_fwd / model_bwd, shared with the atomic kernels), so the step is written once over those:
//   k_ownx_eval   one lane group per TRIPLE of the batch (positive i = triple 2i, its sampled corruption = triple 2i + 1): gather the
//                 model's rows, forward, logistic coefficient, regulariser, backward -- and store the gradient row of every ROLE in the
//                 triple's own slots of `stage` (plain stores; the atomic kernels scatter exactly these rows).
//   k_ownx_step   one owner group per entity / relation with incidences (index, bucket lists, ride-along sampler of kge_pull.hip): for
//                 each incidence (pair, role) it adds the staged rows of the roles that are sourced from ITS id -- head-sourced roles of
//                 the positive (and of the negative when the tail was corrupted), and so on -- into one accumulator per table it owns,
//                 in ascending (role, pair) order, and applies the optimiser to its rows in place (a staged owner reads nobody else's
//                 parameters).  Rows cut into several items combine through LDS, or through partial sums + k_ownx_finish.
#include "kge_row_kernels.h"
#include "kge_sampler_device.h"
#include "kge_pull_device.h"
#include "kge_opt_device.h"
#include <type_traits>

namespace kge {

// ---- which tables an entity / a relation owns, per model (the tables its roles with source h|t / r live in)
__host__ __device__ constexpr int ownx_count(int M, bool rel) {
    return M == KGE_ANALOGY ? 3 : M == KGE_CP ? (rel ? 1 : 2) : (M == KGE_SIMPLE || M == KGE_SIMPLE_IGNR) ? 2 : M == KGE_QUATE ? 4 : 0;
}
__host__ __device__ constexpr int ownx_tab(int M, bool rel, int slot) {
    switch (M) {
        case KGE_ANALOGY: { constexpr int e[3] = {0, 2, 3}, r[3] = {1, 4, 5}; return rel ? r[slot] : e[slot]; }
        case KGE_CP: { constexpr int e[2] = {0, 2}; return rel ? 1 : e[slot]; }
        case KGE_SIMPLE: case KGE_SIMPLE_IGNR: return rel ? 2 + slot : slot;
        case KGE_QUATE: return rel ? 4 + slot : slot;
    }
    return 0;
}
__host__ __device__ constexpr int ownx_slot(int M, int tab) {   // position of `tab` in its class's list
    switch (M) {
        case KGE_ANALOGY: { constexpr int s[6] = {0, 0, 1, 2, 1, 2}; return s[tab]; }
        case KGE_CP: { constexpr int s[3] = {0, 0, 1}; return s[tab]; }
        case KGE_SIMPLE: case KGE_SIMPLE_IGNR: { constexpr int s[4] = {0, 1, 0, 1}; return s[tab]; }
        case KGE_QUATE: return tab & 3;
    }
    return 0;
}
__host__ __device__ constexpr int ownx_nacc(int M) { return ownx_count(M, false) > ownx_count(M, true) ? ownx_count(M, false) : ownx_count(M, true); }
__host__ __device__ inline int ownx_tab_dim(int M, int tab, int dim) { return (M == KGE_ANALOGY && tab >= 2) ? dim / 2 : dim; }

struct OwnXArgs {
    DeviceModel m;                        // tables (updated in place by the owners); grads unused
    float* s1[KGE_MAX_TABLES]; float* s2[KGE_MAX_TABLES];
    const int4* pairs; PullLists lists;
    const int4* items; const int32_t* inc; float* partials; const int4* multi;
    int64_t n_items, n_multi;
    const uint32_t* listed;
    int n_rows, n_pairs, dense, sample_blocks, E, reset_lists;
    float inv_n, lmbda; int reg_type;
    OptArgs opt;
    float* stage;                         // [2 n_pairs][roles][G * NCH]
};

// ------------------------------------------------------------------ phase 1
template <int M, int G, int NCH>
__global__ __launch_bounds__(kBlock) void k_ownx_eval(OwnXArgs a, float* __restrict__ loss) {
    constexpr int GPB = kBlock / G;
    constexpr int NR = role_count(M);
    constexpr int RS = G * NCH;
    const int gl = threadIdx.x % G;
    const int64_t j = (int64_t)blockIdx.x * GPB + threadIdx.x / G;     // triple: 2 * pair + (1 = the corruption)
    float acc = 0.f;
    if (j < 2 * (int64_t)a.n_pairs) {
        const int64_t i = j >> 1;
        const bool neg = (j & 1) != 0;
        const int4 p = a.pairs[i];
        int64_t id[3] = {p.x, p.y, p.z};
        if (neg) {
            const int w = a.lists.pc[i];
            if ((w >> 24) & 1) id[2] = w & 0xFFFFFF; else id[0] = w & 0xFFFFFF;
        }
        const float yy = neg ? -1.f : 1.f;
        Rows<M, NCH> R;
        load_rows<M, G, NCH>(R, a.m, id, gl);
        Saved<M, NCH> sv;
        const float s = model_fwd<M, G, NCH>(R, a.m, sv);
        const float x = yy * s;
        acc += softplus_t(x) * a.inv_n;
        const float ds = yy * sigmoid_t(x) * a.inv_n;
        Rows<M, NCH> Gr;
        model_bwd<M, G, NCH>(R, a.m, sv, ds, Gr);
        if (a.reg_type >= KGE_REG_ID_F2) {        // (SimplE: a function of the ids, no parameter gradient -- see enum kge_reg)
            acc += id_reg_term(id, a.lmbda, a.reg_type);
        } else if (a.reg_type != KGE_REG_NONE) {  // (as k_pointwise_logistic)
            float rs = 0.f;
            const float c2 = 2.f * a.lmbda * a.inv_n, c3 = 3.f * a.lmbda * a.inv_n;
#pragma unroll
            for (int q = 0; q < NR; ++q) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const float v = R.x[q][c];
                    if (a.reg_type == KGE_REG_F2) { rs = fmaf(v, v, rs); Gr.x[q][c] += c2 * v; }
                    else if (a.reg_type == KGE_REG_N3) { rs += v * v * v; Gr.x[q][c] += c3 * v * v; }
                    else { const float b = fabsf(v); rs += b * b * b; Gr.x[q][c] += c3 * v * b; }
                }
            }
            acc += a.lmbda * a.inv_n * gsum<G>(rs);
        }
        float* st = a.stage + j * (int64_t)(NR * RS);
#pragma unroll
        for (int q = 0; q < NR; ++q) store_row<G, NCH>(st + q * RS, Gr.x[q], role_dim<M>(a.m, q), gl);
    }
    block_accumulate_loss<G>(acc, gl, loss);
}

// ------------------------------------------------------------------ phase 2
// A += the staged rows of one triple whose role is sourced from id position SEL; compile-time recursion over the roles so that every
// accumulator index is a constant (a loop variable -- even fully unrolled -- in the index left the accumulators in scratch)
template <int M, int SEL, int G, int NCH, int Q = 0>
__device__ __forceinline__ void ownx_add_roles(float (&A)[ownx_nacc(M)][NCH], const float* __restrict__ st, int dim, int gl) {
    if constexpr (Q < role_count(M)) {
        if constexpr (role_sel(M, Q) == SEL) {
            constexpr int SL = ownx_slot(M, role_tab(M, Q));
            float r[NCH];
            load_row<G, NCH>(r, st + Q * (G * NCH), ownx_tab_dim(M, role_tab(M, Q), dim), gl);
#pragma unroll
            for (int c = 0; c < NCH; ++c) A[SL][c] += r[c];
        }
        ownx_add_roles<M, SEL, G, NCH, Q + 1>(A, st, dim, gl);
    }
}

// the optimiser on the tables this owner holds, in place
template <int M, int OPT, int G, int NCH, int SL = 0>
__device__ __forceinline__ void ownx_apply(const OwnXArgs& a, int g, float (&A)[ownx_nacc(M)][NCH], int gl) {
    if constexpr (SL < ownx_nacc(M)) {
        const bool is_rel = g >= a.E;
        const int64_t own = is_rel ? g - a.E : g;
        constexpr bool has_e = SL < ownx_count(M, false), has_r = SL < ownx_count(M, true);
        constexpr int te = ownx_tab(M, false, has_e ? SL : 0), tr = ownx_tab(M, true, has_r ? SL : 0);
        if ((is_rel && has_r) || (!is_rel && has_e)) {
            // (pointer selects with constant table indices: a runtime index into the kernel-argument arrays would put them in scratch)
            const int dt = is_rel ? ownx_tab_dim(M, tr, a.m.dim) : ownx_tab_dim(M, te, a.m.dim);
            float* const P = const_cast<float*>(is_rel ? a.m.tab[tr] : a.m.tab[te]) + own * dt;
            float* const S1 = (is_rel ? a.s1[tr] : a.s1[te]);
            float* const S2 = (is_rel ? a.s2[tr] : a.s2[te]);
            float p[NCH], m1[NCH], m2[NCH];
            load_row<G, NCH>(p, P, dt, gl);
            if constexpr (OPT != KGE_OPT_SGD) load_row<G, NCH>(m1, S1 + own * dt, dt, gl);
            if constexpr (OPT == KGE_OPT_ADAM) load_row<G, NCH>(m2, S2 + own * dt, dt, gl);
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                float x1 = 0.f, x2 = 0.f;
                if constexpr (OPT != KGE_OPT_SGD) x1 = m1[c];
                if constexpr (OPT == KGE_OPT_ADAM) x2 = m2[c];
                opt_update<OPT>(p[c], A[SL][c], x1, x2, a.opt);
                if constexpr (OPT != KGE_OPT_SGD) m1[c] = x1;
                if constexpr (OPT == KGE_OPT_ADAM) m2[c] = x2;
            }
            store_row<G, NCH>(P, p, dt, gl);
            if constexpr (OPT != KGE_OPT_SGD) store_row<G, NCH>(S1 + own * dt, m1, dt, gl);
            if constexpr (OPT == KGE_OPT_ADAM) store_row<G, NCH>(S2 + own * dt, m2, dt, gl);
        }
        ownx_apply<M, OPT, G, NCH, SL + 1>(a, g, A, gl);
    }
}

template <int M, int OPT, int G, int NCH>
__global__ __launch_bounds__(kBlock) void k_ownx_step(OwnXArgs a, PullSampleArgs sa) {
    constexpr int GPB = kBlock / G;
    constexpr int NR = role_count(M);
    constexpr int RS = G * NCH;
    constexpr int NA = ownx_nacc(M);
    if ((int)blockIdx.x < a.sample_blocks) {   // leading blocks: the sampler of the NEXT batch rides along (other list set)
        const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
        if (i < sa.n) pull_sample_one(sa, i);
        return;
    }
    const int gl = threadIdx.x % G;
    const int grp = threadIdx.x / G;
    const int gbase = (threadIdx.x & 63) / G * G;
    extern __shared__ float s_part[];          // [GPB][NA * NCH * G] partial sums of rows cut into items of this workgroup
    __shared__ int s_vis[GPB][G];
    const int64_t item = (int64_t)((int)blockIdx.x - a.sample_blocks) * GPB + grp;
    int4 it = make_int4(-1, 0, 0, 0);
    if (item < a.n_items) it = a.items[item];
    else if (a.listed != nullptr) {
        const int64_t j = item - a.n_items;
        if (a.dense) {            // every row without an explicit item: its corrupting-entity draws, or a zero gradient
            if (j < a.n_rows && !((a.listed[j >> 5] >> (j & 31)) & 1u)) it = make_int4((int)j, 0, 0, 0);
        } else if (j < a.n_pairs) {   // sparse: pair j stands in as the owner of the entity it drew, if it was the first to draw it
            const int w = a.lists.pc[j];
            const int c = w & 0xFFFFFF;
            if (((w >> kPcFirstBit) & 1) && !((a.listed[c >> 5] >> (c & 31)) & 1u)) it = make_int4(c, 0, 0, 0);
        }
    }
    const int g = it.x;
    const int kind = it.w & 3;
    float A[NA][NCH];
#pragma unroll
    for (int sl = 0; sl < NA; ++sl)
#pragma unroll
        for (int c = 0; c < NCH; ++c) A[sl][c] = 0.f;
    if (g >= 0) {
        const bool is_rel = g >= a.E;
        int cnt = 0;
        bool fast_c = true;
        const bool walks_c = !is_rel && (kind == 0 || kind == 1 || (kind == 3 && ((it.w >> 2) & 15) == 0));
        const int nvis = own_visit_list_dir<G>(a.lists, a.inc, it, g, walks_c, gl, gbase, s_vis[grp], &cnt, &fast_c);
        // (locals, not a.xxx, inside the lambdas: a lambda that captures the kernel-argument struct by reference and hands a member
        // to a function taking a reference leaves the whole struct in scratch -- 720 bytes per lane here)
        const float* __restrict__ stage = a.stage;
        const int32_t* __restrict__ pcs = a.lists.pc;
        const int dim = a.m.dim;
        // add the staged rows of triple `tri` whose role is sourced from id position SEL (0 head, 1 relation, 2 tail)
        // (always_inline: an out-of-line lambda that captures the accumulators by reference keeps them in scratch)
        auto add_sel = [&](int64_t tri, auto sel_tag) __attribute__((always_inline)) {
            constexpr int SEL = decltype(sel_tag)::value;
            ownx_add_roles<M, SEL, G, NCH>(A, stage + tri * (int64_t)(NR * RS), dim, gl);
        };
        auto visit = [&](int e) __attribute__((always_inline)) {
            const int64_t pair = e >> 2;
            const int role = e & 3;
            if (role == kRoleR) {
                add_sel(2 * pair, std::integral_constant<int, 1>{});
                add_sel(2 * pair + 1, std::integral_constant<int, 1>{});
                return;
            }
            const bool tail = ((pcs[pair] >> 24) & 1) != 0;
            if (role == kRoleH) {
                add_sel(2 * pair, std::integral_constant<int, 0>{});
                if (tail) add_sel(2 * pair + 1, std::integral_constant<int, 0>{});
            } else if (role == kRoleT) {
                add_sel(2 * pair, std::integral_constant<int, 2>{});
                if (!tail) add_sel(2 * pair + 1, std::integral_constant<int, 2>{});
            } else {   // drawn as the corrupting entity: the negative's tail-sourced (tail corrupted) or head-sourced roles
                if (tail) add_sel(2 * pair + 1, std::integral_constant<int, 2>{});
                else add_sel(2 * pair + 1, std::integral_constant<int, 0>{});
            }
        };
        for (int v = 0; v < nvis; ++v) visit(s_vis[grp][v]);
        if (cnt > 0 && !fast_c) {   // more drawers than the bucket / the lane group holds: ascending pair order, one at a time
            const int nb = cnt < kPullCap ? cnt : kPullCap;
            int last = -1;
            for (;;) {
                int best = 0x7FFFFFFF;
                for (int m = 0; m < nb; ++m) { const int j = a.lists.bucket[(int64_t)g * kPullCap + m]; if (j > last && j < best) best = j; }
                for (int j = a.lists.head[g]; j >= 0; j = a.lists.next[j]) if (j > last && j < best) best = j;
                if (best == 0x7FFFFFFF) break;
                visit((best << 2) | kRoleC);
                last = best;
            }
        }
        if (cnt > 0 && a.reset_lists && gl == 0) {
            a.lists.count[g] = 0;
            if (cnt > kPullCap) a.lists.head[g] = -1;
        }
        if (kind == 0) {
            ownx_apply<M, OPT, G, NCH>(a, g, A, gl);
        } else if (kind == 3) {
#pragma unroll
            for (int sl = 0; sl < NA; ++sl)
#pragma unroll
                for (int c = 0; c < NCH; ++c) s_part[(grp * NA + sl) * RS + c * G + gl] = A[sl][c];
        } else {
            float* out = a.partials + (int64_t)(it.w >> 2) * (NA * RS);
#pragma unroll
            for (int sl = 0; sl < NA; ++sl)
#pragma unroll
                for (int c = 0; c < NCH; ++c) out[sl * RS + c * G + gl] = A[sl][c];
        }
    }
    __syncthreads();
    if (g >= 0 && kind == 3 && ((it.w >> 2) & 15) == 0) {   // first item of a workgroup-local row: add the others in segment order
        const int nseg = it.w >> 6;
        for (int m = 1; m < nseg; ++m) {
#pragma unroll
            for (int sl = 0; sl < NA; ++sl)
#pragma unroll
                for (int c = 0; c < NCH; ++c) A[sl][c] += s_part[((grp + m) * NA + sl) * RS + c * G + gl];
        }
        ownx_apply<M, OPT, G, NCH>(a, g, A, gl);
    }
}

// rows cut into items across workgroups: their partial sums added in slot order, then the optimiser
template <int M, int OPT, int G, int NCH>
__global__ __launch_bounds__(kBlock) void k_ownx_finish(OwnXArgs a) {
    constexpr int GPB = kBlock / G;
    constexpr int RS = G * NCH;
    constexpr int NA = ownx_nacc(M);
    const int gl = threadIdx.x % G;
    const int64_t mi = (int64_t)blockIdx.x * GPB + threadIdx.x / G;
    if (mi >= a.n_multi) return;
    const int4 row = a.multi[mi];
    float A[NA][NCH];
#pragma unroll
    for (int sl = 0; sl < NA; ++sl)
#pragma unroll
        for (int c = 0; c < NCH; ++c) A[sl][c] = 0.f;
    for (int s = 0; s < row.z; ++s) {
        const float* in = a.partials + (int64_t)(row.y + s) * (NA * RS);
#pragma unroll
        for (int sl = 0; sl < NA; ++sl)
#pragma unroll
            for (int c = 0; c < NCH; ++c) A[sl][c] += in[sl * RS + c * G + gl];
    }
    ownx_apply<M, OPT, G, NCH>(a, row.x, A, gl);
}

// ------------------------------------------------------------------ host side
bool ownx_model(int model) {
    return model == KGE_ANALOGY || model == KGE_CP || model == KGE_SIMPLE || model == KGE_SIMPLE_IGNR || model == KGE_QUATE;
}
static bool ownx_geo(int model, int dim, Geometry* geo) {
    if (!ownx_model(model) || dim <= 0 || dim > 256 || (model == KGE_ANALOGY && (dim & 1))) return false;   // 32-lane groups only
    return pick_geometry(dim, geo);
}
int ownx_groups_per_block(int model, int dim) { Geometry g; return ownx_geo(model, dim, &g) ? kBlock / g.G : 0; }
int ownx_partial_stride(int model, int dim) { Geometry g; return ownx_geo(model, dim, &g) ? ownx_nacc(model) * g.G * g.NCH : 0; }
size_t ownx_stage_floats(int model, int dim, int64_t n_pairs) {
    Geometry g;
    return ownx_geo(model, dim, &g) ? (size_t)2 * n_pairs * role_count(model) * g.G * g.NCH : 0;
}

template <int M, int OPT>
static int launch_ownx_model(OwnXArgs& a, const PullSampleArgs& sa, Geometry geo, float* loss, hipStream_t s) {
    const int64_t extra = a.listed ? (a.dense ? a.n_rows : a.n_pairs) : 0;
    const int64_t units = a.n_items + extra;
#define KGE_OX(G_, NCH_)                                                                                                      \
    if (geo.G == G_ && geo.NCH == NCH_) {                                                                                      \
        constexpr int GPB = kBlock / G_;                                                                                       \
        const int save = a.sample_blocks;                                                                                      \
        a.sample_blocks = 0;                                                                                                   \
        hipLaunchKernelGGL((k_ownx_eval<M, G_, NCH_>), dim3((unsigned)((2 * (int64_t)a.n_pairs + GPB - 1) / GPB)), dim3(kBlock), 0, s, a, loss); \
        a.sample_blocks = save;                                                                                                \
        const size_t lds = (size_t)GPB * ownx_nacc(M) * G_ * NCH_ * sizeof(float);                                             \
        hipLaunchKernelGGL((k_ownx_step<M, OPT, G_, NCH_>), dim3((unsigned)((units + GPB - 1) / GPB + a.sample_blocks)), dim3(kBlock), lds, s, a, sa); \
        if (a.n_multi > 0)                                                                                                     \
            hipLaunchKernelGGL((k_ownx_finish<M, OPT, G_, NCH_>), dim3((unsigned)((a.n_multi + GPB - 1) / GPB)), dim3(kBlock), 0, s, a); \
        return check_launch("k_ownx_step");                                                                                    \
    }
    KGE_OX(32, 1) KGE_OX(32, 2) KGE_OX(32, 4) KGE_OX(32, 8)
#undef KGE_OX
    return -1;
}

template <int OPT>
static int launch_ownx_opt(OwnXArgs& a, const PullSampleArgs& sa, int model, Geometry geo, float* loss, hipStream_t s) {
    switch (model) {
        case KGE_ANALOGY: return launch_ownx_model<KGE_ANALOGY, OPT>(a, sa, geo, loss, s);
        case KGE_CP: return launch_ownx_model<KGE_CP, OPT>(a, sa, geo, loss, s);
        case KGE_SIMPLE: return launch_ownx_model<KGE_SIMPLE, OPT>(a, sa, geo, loss, s);
        case KGE_SIMPLE_IGNR: return launch_ownx_model<KGE_SIMPLE_IGNR, OPT>(a, sa, geo, loss, s);
        case KGE_QUATE: return launch_ownx_model<KGE_QUATE, OPT>(a, sa, geo, loss, s);
    }
    return -1;
}

// one step: evaluate every triple once, owners add + apply, finish the rows cut across workgroups
int launch_ownx_step(const kge_model_desc* m, float* const* state1, float* const* state2, const int32_t* pairs, int64_t n_pairs,
                     const kge_pull_lists* lists, const int32_t* items, int64_t n_items, const uint32_t* listed, const int32_t* inc,
                     float* partials, const int32_t* multi, int64_t n_multi, int dense, float lmbda, int reg_type, int optimizer, float lr,
                     int64_t step, const int32_t* next_pairs, const int32_t* next_inv, int64_t next_n, const float* bern,
                     const uint64_t* slots, int64_t n_slots, uint64_t seed, uint64_t next_offset, const kge_pull_lists* next_lists,
                     float* loss, float* stage, hipStream_t s) {
    Geometry geo;
    if (!ownx_geo(m->model, m->dim, &geo)) { set_error("kge_own_run: model %d / hidden size %d not supported by the staged step", m->model, m->dim); return -1; }
    if (!stage) { set_error("kge_own_run: this model needs the staging buffer (kge_own_stage_bytes)"); return -1; }
    OwnXArgs a;
    a.m = to_device_model(m);
    const int nt = ownx_count(m->model, false) + ownx_count(m->model, true);
    for (int t = 0; t < KGE_MAX_TABLES; ++t) { a.s1[t] = state1 ? state1[t] : nullptr; a.s2[t] = state2 ? state2[t] : nullptr; }
    for (int sl = 0; sl < ownx_count(m->model, false) + ownx_count(m->model, true); ++sl) {
        const bool rel = sl >= ownx_count(m->model, false);
        const int t = ownx_tab(m->model, rel, rel ? sl - ownx_count(m->model, false) : sl);
        if (!m->tables[t]) { set_error("kge_own_run: table %d missing", t); return -1; }
        if (optimizer != KGE_OPT_SGD && !a.s1[t]) { set_error("kge_own_run: optimizer state of table %d missing", t); return -1; }
        if (optimizer == KGE_OPT_ADAM && !a.s2[t]) { set_error("kge_own_run: adam needs two state buffers (table %d)", t); return -1; }
    }
    (void)nt;
    a.pairs = (const int4*)pairs; a.lists = to_lists(lists); a.items = (const int4*)items; a.inc = inc; a.partials = partials;
    a.multi = (const int4*)multi; a.n_items = n_items; a.n_multi = n_multi; a.listed = listed;
    a.n_rows = (int)(m->tot_entity + m->tot_relation); a.n_pairs = (int)n_pairs; a.dense = dense ? 1 : 0;
    a.E = (int)m->tot_entity; a.reset_lists = 1;
    a.inv_n = 1.0f / (float)(2 * n_pairs); a.lmbda = lmbda; a.reg_type = reg_type;
    a.opt = make_opt_args(lr, step < 1 ? 1 : step);
    a.stage = stage;
    PullSampleArgs sa = make_sample_args(next_pairs, next_inv, next_pairs && next_lists ? next_n : 0, m->tot_entity, bern, slots,
                                         n_slots, seed, next_offset, nullptr, next_lists);
    sa.no_desc = 1;
    a.sample_blocks = sa.n > 0 ? (int)((sa.n + kBlock - 1) / kBlock) : 0;
    switch (optimizer) {
        case KGE_OPT_SGD: return launch_ownx_opt<KGE_OPT_SGD>(a, sa, m->model, geo, loss, s);
        case KGE_OPT_ADAM: return launch_ownx_opt<KGE_OPT_ADAM>(a, sa, m->model, geo, loss, s);
        case KGE_OPT_ADAGRAD: return launch_ownx_opt<KGE_OPT_ADAGRAD>(a, sa, m->model, geo, loss, s);
        case KGE_OPT_RMSPROP: return launch_ownx_opt<KGE_OPT_RMSPROP>(a, sa, m->model, geo, loss, s);
    }
    set_error("kge_own_run: unknown optimizer %d", optimizer);
    return -1;
}

}  // namespace kge
