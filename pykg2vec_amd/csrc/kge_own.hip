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
#include <type_traits>

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
    int multi_only;            // k_own_apply: only the rows cut into many items (the staged step applied the others itself)
    float* stage;              // staged form: [n_pairs][4 roles: H T R C][NT][VEC * NV * G] gradient rows left by k_own_eval; NULL: owners re-evaluate
};

// Rows in registers: lane gl of a G-lane group holds NE = VEC * NV elements of a row.  VEC = 4 (hidden size % 4 == 0): NV float4
// per lane, one 16-byte load per float4 (element 4 * (v * G + gl) + c); VEC = 1 (any hidden size): NV dwords per lane (element
// v * G + gl).  All arithmetic below is per element and does not care.
template <int VEC, int G, int NV>
__device__ __forceinline__ void load_row_e(float (&x)[VEC * NV], const float* __restrict__ row, int d, int gl) {
    if constexpr (VEC == 4) {
        const int nvec = d >> 2;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int i = v * G + gl;
            const float4 q = i < nvec ? reinterpret_cast<const float4*>(row)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            x[4 * v] = q.x; x[4 * v + 1] = q.y; x[4 * v + 2] = q.z; x[4 * v + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int v = 0; v < NV; ++v) { const int e = v * G + gl; x[v] = e < d ? row[e] : 0.f; }
    }
}
template <int VEC, int G, int NV>
__device__ __forceinline__ void store_row_e(float* __restrict__ row, const float (&x)[VEC * NV], int d, int gl) {
    if constexpr (VEC == 4) {
        const int nvec = d >> 2;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int i = v * G + gl;
            if (i < nvec) reinterpret_cast<float4*>(row)[i] = make_float4(x[4 * v], x[4 * v + 1], x[4 * v + 2], x[4 * v + 3]);
        }
    } else {
#pragma unroll
        for (int v = 0; v < NV; ++v) { const int e = v * G + gl; if (e < d) row[e] = x[v]; }
    }
}

// (table pointers by value: a reference to a kernel-argument array would put the argument struct in scratch; d_eff = 0 turns the
// loads of a row set that is not needed into zero fills)
template <int NT, int VEC, int G, int NV>
__device__ __forceinline__ void load_rows_nt(float (&x)[NT][VEC * NV], const float* t0, const float* t1, int64_t row, int d, int d_eff, int gl) {
    load_row_e<VEC, G, NV>(x[0], t0 + row * d, d_eff, gl);
    if constexpr (NT == 2) load_row_e<VEC, G, NV>(x[1], t1 + row * d, d_eff, gl);
}

// -(score) partial of one lane: sum over its elements of Re(<a, r, conj b>) (ComplEx) or a*r*b (DistMult)
template <int NT, int NE>
__device__ __forceinline__ float dot3(const float (&a)[NT][NE], const float (&r)[NT][NE], const float (&b)[NT][NE]) {
    float p = 0.f;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        if constexpr (NT == 1) p = fmaf(a[0][e] * r[0][e], b[0][e], p);
        else {
            const float ar = a[0][e], ai = a[1][e], rr_ = r[0][e], ri = r[1][e], br = b[0][e], bi = b[1][e];
            p += ar * br * rr_ + ai * bi * rr_ + ar * bi * ri - ai * br * ri;
        }
    }
    return p;
}

// g += k * d(-score)/d(position) ; position 0 = head (uses r, b), 1 = tail (uses a, r), 2 = relation (uses a, b).  The position
// is uniform over the owner group: one branch per visit, not a select per element
template <int NT, int NE>
__device__ __forceinline__ void add_grad(float (&g)[NT][NE], int pos, float k, const float (&a)[NT][NE], const float (&r)[NT][NE],
                                         const float (&b)[NT][NE]) {
    if (pos == 0) {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            if constexpr (NT == 1) g[0][e] = fmaf(k, r[0][e] * b[0][e], g[0][e]);
            else {
                g[0][e] = fmaf(k, b[0][e] * r[0][e] + b[1][e] * r[1][e], g[0][e]);
                g[1][e] = fmaf(k, b[1][e] * r[0][e] - b[0][e] * r[1][e], g[1][e]);
            }
        }
    } else if (pos == 1) {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            if constexpr (NT == 1) g[0][e] = fmaf(k, a[0][e] * r[0][e], g[0][e]);
            else {
                g[0][e] = fmaf(k, a[0][e] * r[0][e] - a[1][e] * r[1][e], g[0][e]);
                g[1][e] = fmaf(k, a[1][e] * r[0][e] + a[0][e] * r[1][e], g[1][e]);
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            if constexpr (NT == 1) g[0][e] = fmaf(k, a[0][e] * b[0][e], g[0][e]);
            else {
                g[0][e] = fmaf(k, a[0][e] * b[0][e] + a[1][e] * b[1][e], g[0][e]);
                g[1][e] = fmaf(k, a[0][e] * b[1][e] - a[1][e] * b[0][e], g[1][e]);
            }
        }
    }
}

// regulariser of one row set: value partial (sum of x^2 | x^3 | |x|^3 over this lane's elements)
template <int NT, int NE>
__device__ __forceinline__ float reg_value(const float (&x)[NT][NE], int reg_type) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const float q = x[t][e];
            s += reg_type == KGE_REG_F2 ? q * q : (reg_type == KGE_REG_N3 ? q * q * q : fabsf(q) * q * q);
        }
    return s;
}
// g += k * d reg / d x  (k already carries lmbda / n and the number of occurrences)
template <int NT, int NE>
__device__ __forceinline__ void reg_grad(float (&g)[NT][NE], const float (&x)[NT][NE], float k, int reg_type) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const float q = x[t][e];
            const float dq = reg_type == KGE_REG_F2 ? 2.f * q : (reg_type == KGE_REG_N3 ? 3.f * q * q : 3.f * q * fabsf(q));
            g[t][e] = fmaf(k, dq, g[t][e]);
        }
}

// the dense-semantics optimiser on ONE row set, in place: parameter, state (and, unless the caller holds it, the gradient row) are
// requested together
template <int OPT, int NT, int VEC, int G, int NV>
__device__ __forceinline__ void own_apply_row(const OwnArgs& a, int g, float (&gv)[NT][VEC * NV], bool have_g, int gl) {
    constexpr int NE = VEC * NV;
    const int d = a.d;
    const bool is_rel = g >= a.E;
    const int64_t off = (int64_t)(is_rel ? g - a.E : g) * d;
    float P[NT][NE], M1[NT][NE], M2[NT][NE];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        load_row_e<VEC, G, NV>(P[t], (is_rel ? a.rel[t] : a.ent[t]) + off, d, gl);
        if constexpr (OPT != KGE_OPT_SGD) load_row_e<VEC, G, NV>(M1[t], (is_rel ? a.s1_rel[t] : a.s1_ent[t]) + off, d, gl);
        if constexpr (OPT == KGE_OPT_ADAM) load_row_e<VEC, G, NV>(M2[t], (is_rel ? a.s2_rel[t] : a.s2_ent[t]) + off, d, gl);
        if (!have_g) load_row_e<VEC, G, NV>(gv[t], (is_rel ? a.g_rel[t] : a.g_ent[t]) + off, d, gl);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            float m1 = 0.f, m2 = 0.f;
            if constexpr (OPT != KGE_OPT_SGD) m1 = M1[t][e];
            if constexpr (OPT == KGE_OPT_ADAM) m2 = M2[t][e];
            opt_update<OPT>(P[t][e], gv[t][e], m1, m2, a.opt);
            if constexpr (OPT != KGE_OPT_SGD) M1[t][e] = m1;
            if constexpr (OPT == KGE_OPT_ADAM) M2[t][e] = m2;
        }
        store_row_e<VEC, G, NV>(const_cast<float*>(is_rel ? a.rel[t] : a.ent[t]) + off, P[t], d, gl);
        if constexpr (OPT != KGE_OPT_SGD) store_row_e<VEC, G, NV>((is_rel ? a.s1_rel[t] : a.s1_ent[t]) + off, M1[t], d, gl);
        if constexpr (OPT == KGE_OPT_ADAM) store_row_e<VEC, G, NV>((is_rel ? a.s2_rel[t] : a.s2_ent[t]) + off, M2[t], d, gl);
    }
}

// ---- staged form, phase 0: every bundle (positive i and the negative the sampler drew for it) is evaluated ONCE by one lane
// group and leaves the gradient rows it produces in its own four slots of `stage` -- for the head entity (the negative's share
// merged in when the tail was corrupted), the tail entity (likewise), the relation (both triples) and the drawn entity -- with plain
// coalesced stores; the owners of k_own_step<..., STAGED> then only ADD the rows of their incidences (each staged row has exactly
// one reader) instead of re-evaluating every bundle they occur in (3.25 evaluations per bundle -> 1, and the 127-VGPR evaluation
// leaves the kernel whose occupancy decides the step).  Loss and regulariser value are accounted here.
template <int NT, int VEC, int G, int NV>
__global__ __launch_bounds__(kBlock, NT == 2 ? 5 : 4) void k_own_eval(OwnArgs a, float* __restrict__ loss) {
    constexpr int GPB = kBlock / G;
    constexpr int NE = VEC * NV;
    constexpr int RSE = NE * G;
    const int gl = threadIdx.x % G;
    const int d = a.d;
    const int64_t i = (int64_t)blockIdx.x * GPB + threadIdx.x / G;
    float acc = 0.f;
    if (i < a.n_pairs) {
        const int4 p = a.pairs[i];
        const int w = a.lists.pc[i];
        const bool tail = ((w >> 24) & 1) != 0;
        const int c = w & 0xFFFFFF;
        float A[NT][NE], RL[NT][NE], B[NT][NE], C[NT][NE];
        load_rows_nt<NT, VEC, G, NV>(A, a.ent[0], a.ent[1], p.x, d, d, gl);
        load_rows_nt<NT, VEC, G, NV>(RL, a.rel[0], a.rel[1], p.y, d, d, gl);
        load_rows_nt<NT, VEC, G, NV>(B, a.ent[0], a.ent[1], p.z, d, d, gl);
        load_rows_nt<NT, VEC, G, NV>(C, a.ent[0], a.ent[1], c, d, d, gl);
        const bool reg_on = a.reg_type != KGE_REG_NONE;
        const float reg_k = a.lmbda * a.inv_n;
        // both energies first (two butterflies), then ONE output row set at a time, stored as soon as it is complete: four live
        // row sets of inputs + one of output instead of eight (146 -> ~70 VGPRs: the kernel is a latency chain, occupancy decides it)
        float pp = dot3<NT, NE>(A, RL, B), rsp = 0.f, pn, rsn = 0.f;
        if (tail) pn = dot3<NT, NE>(A, RL, C); else pn = dot3<NT, NE>(C, RL, B);
        if (reg_on) {
            const float ra = reg_value<NT, NE>(A, a.reg_type), rr = reg_value<NT, NE>(RL, a.reg_type), rb = reg_value<NT, NE>(B, a.reg_type);
            rsp = ra + rr + rb;
            rsn = (tail ? ra : rb) + rr + reg_value<NT, NE>(C, a.reg_type);
        }
        gsum2<G>(pp, rsp);
        gsum2<G>(pn, rsn);
        const float xp = -pp, xn = pn;                         // y * energy, energy = -p; labels +1 / -1
        const float dsp = sigmoid_t(xp) * a.inv_n, dsn = -sigmoid_t(xn) * a.inv_n;
        acc += (softplus_t(xp) * a.inv_n + reg_k * rsp) + (softplus_t(xn) * a.inv_n + reg_k * rsn);
        float* st = a.stage + i * (int64_t)(4 * NT * RSE);
        auto zero = [](float (&g)[NT][NE]) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int e = 0; e < NE; ++e) g[t][e] = 0.f;
        };
        auto put = [&](int role, const float (&g)[NT][NE]) {
#pragma unroll
            for (int t = 0; t < NT; ++t) store_row_e<VEC, G, NV>(st + (role * NT + t) * RSE, g[t], d, gl);
        };
        {   // head entity: the positive, and the negative when it kept the head
            float g[NT][NE]; zero(g);
            add_grad<NT, NE>(g, 0, -dsp, A, RL, B);
            if (reg_on) reg_grad<NT, NE>(g, A, reg_k, a.reg_type);
            if (tail) { add_grad<NT, NE>(g, 0, -dsn, A, RL, C); if (reg_on) reg_grad<NT, NE>(g, A, reg_k, a.reg_type); }
            put(kRoleH, g);
        }
        __builtin_amdgcn_sched_barrier(0);   // (one output row set at a time: keeps the four computations from being interleaved)
        {   // tail entity
            float g[NT][NE]; zero(g);
            add_grad<NT, NE>(g, 1, -dsp, A, RL, B);
            if (reg_on) reg_grad<NT, NE>(g, B, reg_k, a.reg_type);
            if (!tail) { add_grad<NT, NE>(g, 1, -dsn, C, RL, B); if (reg_on) reg_grad<NT, NE>(g, B, reg_k, a.reg_type); }
            put(kRoleT, g);
        }
        __builtin_amdgcn_sched_barrier(0);   // (one output row set at a time: keeps the four computations from being interleaved)
        {   // relation: both triples
            float g[NT][NE]; zero(g);
            add_grad<NT, NE>(g, 2, -dsp, A, RL, B);
            if (reg_on) reg_grad<NT, NE>(g, RL, reg_k, a.reg_type);
            if (tail) add_grad<NT, NE>(g, 2, -dsn, A, RL, C); else add_grad<NT, NE>(g, 2, -dsn, C, RL, B);
            if (reg_on) reg_grad<NT, NE>(g, RL, reg_k, a.reg_type);
            put(kRoleR, g);
        }
        __builtin_amdgcn_sched_barrier(0);   // (one output row set at a time: keeps the four computations from being interleaved)
        {   // the drawn entity: the negative only, as its tail (tail corrupted) or head
            float g[NT][NE]; zero(g);
            if (tail) add_grad<NT, NE>(g, 1, -dsn, A, RL, C); else add_grad<NT, NE>(g, 0, -dsn, C, RL, B);
            if (reg_on) reg_grad<NT, NE>(g, C, reg_k, a.reg_type);
            put(kRoleC, g);
        }
    }
    block_accumulate_loss<G>(acc, gl, loss);
}

// resident waves per SIMD the register allocation must allow (A/B builds: -DKGE_OWN_WAVES=n)
#ifndef KGE_OWN_WAVES
#define KGE_OWN_WAVES 4
#endif
// OPT >= 0 (staged form only): the owner applies the optimiser to its row set in place as soon as the row's gradient is complete
// -- safe because a staged owner reads nobody else's parameters -- instead of storing a gradient row for k_own_apply to pick up.
template <int NT, int VEC, int G, int NV, bool STAGED = false, int OPT = -1>
__global__ __launch_bounds__(kBlock, KGE_OWN_WAVES) void k_own_step(OwnArgs a, PullSampleArgs sa, float* __restrict__ loss) {
    static_assert(OPT < 0 || STAGED, "the fused apply needs the staged form");
    constexpr int GPB = kBlock / G;
    constexpr int NE = VEC * NV;
    if ((int)blockIdx.x < a.sample_blocks) {   // leading blocks: the sampler of the NEXT batch rides along (other list set)
        const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
        if (i < sa.n) pull_sample_one(sa, i);
        return;
    }
    const int gl = threadIdx.x % G;
    const int grp = threadIdx.x / G;
    const int gbase = (threadIdx.x & 63) / G * G;
    const int d = a.d;
    const int64_t item = (int64_t)((int)blockIdx.x - a.sample_blocks) * GPB + grp;
    __shared__ float s_part[GPB][NT * NE * G];
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
    float X[NT][NE], gs[NT][NE];
    if (g >= 0) {
        const bool is_rel = g >= a.E;
        const int64_t own = is_rel ? g - a.E : g;
        // the owner's rows: requested first, they depend on nothing but the item (the staged form adds rows and never evaluates)
        if constexpr (!STAGED) {
            if (is_rel) load_rows_nt<NT, VEC, G, NV>(X, a.rel[0], a.rel[1], own, d, d, gl);
            else load_rows_nt<NT, VEC, G, NV>(X, a.ent[0], a.ent[1], own, d, d, gl);
        }
        int cnt = 0;
        bool fast_c = true;
        const bool walks_c = !is_rel && (kind == 0 || kind == 1 || (kind == 3 && ((it.w >> 2) & 15) == 0));
        int nvis;
        if constexpr (STAGED) nvis = own_visit_list_dir<G>(a.lists, a.inc, it, g, walks_c, gl, gbase, reinterpret_cast<int*>(s_desc[grp]), &cnt, &fast_c);
        else nvis = own_visit_list<G>(a.lists, it, g, walks_c, gl, gbase, s_desc[grp], &cnt, &fast_c);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < NE; ++e) gs[t][e] = 0.f;
        if constexpr (STAGED) {
            // a visit = (pair << 2 | role): add the NT rows k_own_eval staged for that role of that bundle (slot = role)
            constexpr int RSE = NE * G;
            const int* __restrict__ vis = reinterpret_cast<const int*>(s_desc[grp]);
            auto rows_of = [&](int e, float (&r)[NT][NE]) {
                const float* st = a.stage + ((int64_t)(e >> 2) * 4 + (e & 3)) * (NT * RSE);
#pragma unroll
                for (int t = 0; t < NT; ++t) load_row_e<VEC, G, NV>(r[t], st + t * RSE, d, gl);
            };
#ifndef KGE_OWN_BATCH
#define KGE_OWN_BATCH 4
#endif
            constexpr int kBatch = KGE_OWN_BATCH;     // visits whose rows are requested before the first is added
            for (int v0 = 0; v0 < nvis; v0 += kBatch) {
                float r[kBatch][NT][NE];
#pragma unroll
                for (int q = 0; q < kBatch; ++q)
                    if (v0 + q < nvis) rows_of(vis[v0 + q], r[q]);
#pragma unroll
                for (int q = 0; q < kBatch; ++q)
                    if (v0 + q < nvis) {
#pragma unroll
                        for (int t = 0; t < NT; ++t)
#pragma unroll
                            for (int e = 0; e < NE; ++e) gs[t][e] += r[q][t][e];
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
                    float r[NT][NE];
                    rows_of((best << 2) | kRoleC, r);
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int e = 0; e < NE; ++e) gs[t][e] += r[t][e];
                    last = best;
                }
            }
        } else {

        // One visit = the bundle (P = (h, r, t), N = (h, r, c) | (c, r, t)) seen from the owner's row.  The two triples are
        // independent loss terms, so they are evaluated one after the other with a working set of three row sets (head, relation,
        // tail -- the owner's own set among them is X itself): 40 row registers instead of the 72 a whole-bundle evaluation holds,
        // which is what decides how many owners a CU keeps in flight (the kernel is latency x occupancy bound).
        const bool reg_on = a.reg_type != KGE_REG_NONE;
        const float reg_k = a.lmbda * a.inv_n;
        // triple (hd, rl, tl) with label y, the owner at position POS (0 head, 1 tail, 2 relation); RL = the relation rows when
        // the owner is an entity (loaded once per visit)
        // the arithmetic of one triple once its rows are in registers
        auto triple_core = [&](auto pos_tag, float y, const float (&A)[NT][NE], const float (&RL)[NT][NE], const float (&B)[NT][NE]) {
            constexpr int POS = decltype(pos_tag)::value;
            float p, rs = 0.f;
            if constexpr (POS == 0) p = dot3<NT, NE>(X, RL, B);
            else if constexpr (POS == 1) p = dot3<NT, NE>(A, RL, X);
            else {
                p = dot3<NT, NE>(A, X, B);
                if (reg_on) rs = reg_value<NT, NE>(A, a.reg_type) + reg_value<NT, NE>(X, a.reg_type) + reg_value<NT, NE>(B, a.reg_type);
            }
            if constexpr (POS == 2) gsum2<G>(p, rs); else p = gsum<G>(p);
            const float x = -y * p;                                   // y * energy, energy = -p
            const float ds = y * sigmoid_t(x) * a.inv_n;              // d loss / d energy   (loss = mean softplus(y * energy))
            if constexpr (POS == 2) acc += softplus_t(x) * a.inv_n + reg_k * rs;   // the relation owner accounts for the loss terms
            if constexpr (POS == 0) add_grad<NT, NE>(gs, 0, -ds, X, RL, B);
            else if constexpr (POS == 1) add_grad<NT, NE>(gs, 1, -ds, A, RL, X);
            else add_grad<NT, NE>(gs, 2, -ds, A, X, B);
            if (reg_on) reg_grad<NT, NE>(gs, X, reg_k, a.reg_type);
        };
        auto triple = [&](auto pos_tag, int hd, int tl, float y, const float (&RL)[NT][NE]) {
            constexpr int POS = decltype(pos_tag)::value;
            float A[NT][NE], B[NT][NE];
            if constexpr (POS != 0) load_rows_nt<NT, VEC, G, NV>(A, a.ent[0], a.ent[1], hd, d, d, gl);
            if constexpr (POS != 1) load_rows_nt<NT, VEC, G, NV>(B, a.ent[0], a.ent[1], tl, d, d, gl);
            triple_core(pos_tag, y, A, RL, B);
        };
        auto visit = [&](int h, int r, int t, int w) {
            const int role = (w >> 25) & 3;
            const bool tail = ((w >> 24) & 1) != 0;
            const int c = w & 0xFFFFFF;
            const int nhd = tail ? h : c, ntl = tail ? c : t;          // the corrupted triple
            if (role == kRoleR) {
                triple(std::integral_constant<int, 2>{}, h, t, 1.f, X);
                triple(std::integral_constant<int, 2>{}, nhd, ntl, -1.f, X);
                return;
            }
            float RL[NT][NE];
            load_rows_nt<NT, VEC, G, NV>(RL, a.rel[0], a.rel[1], r, d, d, gl);
            if (role == kRoleH) {
                triple(std::integral_constant<int, 0>{}, h, t, 1.f, RL);
                if (tail) triple(std::integral_constant<int, 0>{}, nhd, ntl, -1.f, RL);
            } else if (role == kRoleT) {
                triple(std::integral_constant<int, 1>{}, h, t, 1.f, RL);
                if (!tail) triple(std::integral_constant<int, 1>{}, nhd, ntl, -1.f, RL);
            } else {   // drawn as the corrupting entity: only the corrupted triple, as its tail (tail corrupted) or head
                if (tail) triple(std::integral_constant<int, 1>{}, nhd, ntl, -1.f, RL);
                else triple(std::integral_constant<int, 0>{}, nhd, ntl, -1.f, RL);
            }
        };
        for (int v = 0; v < nvis; ++v) {
            const int4 ds = s_desc[grp][v];   // same address for the whole group: a broadcast read
            visit(ds.x, ds.y, ds.z, ds.w);
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
                visit(p2.x, p2.y, p2.z, (a.lists.pc[best] & 0x1FFFFFF) | (kRoleC << 25));
                last = best;
            }
        }
        }   // (owners that re-evaluate)
        if (cnt > 0 && a.reset_lists && gl == 0) {
            a.lists.count[g] = 0;
            if (cnt > kPullCap) a.lists.head[g] = -1;
        }
        if (kind == 0) {
            if constexpr (OPT >= 0) own_apply_row<OPT, NT, VEC, G, NV>(a, g, gs, true, gl);
            else {
#pragma unroll
                for (int t = 0; t < NT; ++t) store_row_e<VEC, G, NV>((is_rel ? a.g_rel[t] : a.g_ent[t]) + own * d, gs[t], d, gl);
            }
        } else if (kind == 3) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int e = 0; e < NE; ++e) s_part[grp][(t * NE + e) * G + gl] = gs[t][e];
        } else {
            float* out = a.partials + (int64_t)(it.w >> 2) * (NT * NE * G);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int e = 0; e < NE; ++e) out[(t * NE + e) * G + gl] = gs[t][e];
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
                for (int e = 0; e < NE; ++e) gs[t][e] += s_part[grp + m][(t * NE + e) * G + gl];
        }
        if constexpr (OPT >= 0) own_apply_row<OPT, NT, VEC, G, NV>(a, g, gs, true, gl);
        else {
#pragma unroll
            for (int t = 0; t < NT; ++t) store_row_e<VEC, G, NV>((is_rel ? a.g_rel[t] : a.g_ent[t]) + own * d, gs[t], d, gl);
        }
    }
    block_accumulate_loss<G>(acc, gl, loss);
}

// phase 2: the optimiser on every row that has a gradient row (or a list of partial sums).  Blocks [0, n_multi): one WORKGROUP
// per row that was cut into many items (its partial list is the long chain of this launch: the owner groups of the workgroup
// each add a contiguous share of the slots, the shares are then added in order); the other blocks: one owner group per row.
template <int OPT, int NT, int VEC, int G, int NV>
__global__ __launch_bounds__(kBlock) void k_own_apply(OwnArgs a) {
    constexpr int GPB = kBlock / G;
    constexpr int NE = VEC * NV;
    const int gl = threadIdx.x % G;
    const int grp = threadIdx.x / G;
    __shared__ float s_sum[GPB][NT * NE * G];
    int g = -1;
    float gv[NT][NE];
    bool have_g = false;
    if ((int64_t)blockIdx.x < a.n_multi) {
        const int4 row = a.multi[blockIdx.x];
        const int nslots = row.z;
        const int per = (nslots + GPB - 1) / GPB;
        const int s0 = grp * per, s1 = min(nslots, s0 + per);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < NE; ++e) gv[t][e] = 0.f;
        const float* base = a.partials + (int64_t)row.y * (NT * NE * G);
        int s = s0;
        for (; s + 4 <= s1; s += 4) {   // four slots in flight, added in slot order
            float q[4][NT][NE];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int e = 0; e < NE; ++e) q[u][t][e] = base[(int64_t)(s + u) * (NT * NE * G) + (t * NE + e) * G + gl];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int e = 0; e < NE; ++e) gv[t][e] += q[u][t][e];
        }
        for (; s < s1; ++s)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int e = 0; e < NE; ++e) gv[t][e] += base[(int64_t)s * (NT * NE * G) + (t * NE + e) * G + gl];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < NE; ++e) s_sum[grp][(t * NE + e) * G + gl] = gv[t][e];
        __syncthreads();
        if (grp != 0) return;
        for (int m = 1; m < GPB; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int e = 0; e < NE; ++e) gv[t][e] += s_sum[m][(t * NE + e) * G + gl];
        g = row.x;
        have_g = true;
    } else {
        if (a.multi_only) return;
        const int64_t unit = ((int64_t)blockIdx.x - a.n_multi) * GPB + grp;
        if (unit < a.n_items) {
            const int4 it = a.items[unit];
            const int kind = it.w & 3;
            if (it.x >= 0 && (kind == 0 || (kind == 3 && ((it.w >> 2) & 15) == 0))) g = it.x;
        } else if (a.listed != nullptr) {
            const int64_t j = unit - a.n_items;
            if (a.dense) {
                if (j < a.n_rows && !((a.listed[j >> 5] >> (j & 31)) & 1u)) g = (int)j;
            } else if (j < a.n_pairs) {
                const int w = a.lists.pc[j];
                const int c = w & 0xFFFFFF;
                if (((w >> kPcFirstBit) & 1) && !((a.listed[c >> 5] >> (c & 31)) & 1u)) g = c;
            }
        }
    }
    if (g < 0) return;
    own_apply_row<OPT, NT, VEC, G, NV>(a, g, gv, have_g, gl);
}

// ------------------------------------------------------------------ host side
struct OwnGeo { int NT, VEC, G, NV; };
static OwnGeo own_geo(int model, int dim) {
    OwnGeo g{0, 0, 0, 0};
    if (dim <= 0 || dim > 512) return g;
    g.NT = model == KGE_COMPLEX ? 2 : (model == KGE_DISTMULT ? 1 : 0);
    if (!g.NT) return g;
    if ((dim & 3) == 0) {   // rows move as float4
        const int nvec = dim >> 2;
        g.VEC = 4;
        if (nvec <= 32) { g.G = 32; g.NV = 1; }
        else if (nvec <= 64) { g.G = 64; g.NV = 1; }
        else { g.G = 64; g.NV = 2; }
    } else {                // any other hidden size: one dword per element
        g.VEC = 1;
        if (dim <= 128) { g.G = 32; g.NV = 4; }
        else if (dim <= 256) { g.G = 64; g.NV = 4; }
        else { g.G = 64; g.NV = 8; }
    }
    return g;
}

static int fill_own_args(const kge_model_desc* m, float* const* state1, float* const* state2, OwnArgs* a, OwnGeo* geo, const char* who) {
    *geo = own_geo(m->model, m->dim);
    if (!geo->G) { set_error("%s: DistMult / ComplEx with a hidden size of at most 512 (model %d, dim %d)", who, m->model, m->dim); return -1; }
    const int NT = geo->NT;
    for (int t = 0; t < 2; ++t) {
        const bool on = t < NT;
        a->ent[t] = on ? m->tables[t] : nullptr;           a->rel[t] = on ? m->tables[NT + t] : nullptr;
        a->g_ent[t] = on ? m->grads[t] : nullptr;          a->g_rel[t] = on ? m->grads[NT + t] : nullptr;
        a->s1_ent[t] = (on && state1) ? state1[t] : nullptr; a->s1_rel[t] = (on && state1) ? state1[NT + t] : nullptr;
        a->s2_ent[t] = (on && state2) ? state2[t] : nullptr; a->s2_rel[t] = (on && state2) ? state2[NT + t] : nullptr;
        if (on && (!a->ent[t] || !a->rel[t] || !a->g_ent[t] || !a->g_rel[t])) { set_error("%s: tables / gradient row buffers missing", who); return -1; }
        if (on && geo->VEC == 4 && ((((uintptr_t)a->ent[t] | (uintptr_t)a->rel[t] | (uintptr_t)a->g_ent[t] | (uintptr_t)a->g_rel[t]) & 15))) {
            set_error("%s: tables and gradient buffers must be 16-byte aligned", who); return -1;
        }
    }
    a->E = (int)m->tot_entity; a->d = m->dim; a->n_rows = (int)(m->tot_entity + m->tot_relation);
    return 0;
}

#define KGE_OWN_CASE(NT_, VEC_, G_, NV_, ...)                                                                      \
    if (geo.NT == NT_ && geo.VEC == VEC_ && geo.G == G_ && geo.NV == NV_) { constexpr int NT = NT_, VEC = VEC_, G = G_, NV = NV_; __VA_ARGS__ }
#define KGE_OWN_GEO(...)                                                                                            \
    KGE_OWN_CASE(1, 4, 32, 1, __VA_ARGS__) KGE_OWN_CASE(1, 4, 64, 1, __VA_ARGS__) KGE_OWN_CASE(1, 4, 64, 2, __VA_ARGS__)  \
    KGE_OWN_CASE(2, 4, 32, 1, __VA_ARGS__) KGE_OWN_CASE(2, 4, 64, 1, __VA_ARGS__) KGE_OWN_CASE(2, 4, 64, 2, __VA_ARGS__)  \
    KGE_OWN_CASE(1, 1, 32, 4, __VA_ARGS__) KGE_OWN_CASE(1, 1, 64, 4, __VA_ARGS__) KGE_OWN_CASE(1, 1, 64, 8, __VA_ARGS__)  \
    KGE_OWN_CASE(2, 1, 32, 4, __VA_ARGS__) KGE_OWN_CASE(2, 1, 64, 4, __VA_ARGS__) KGE_OWN_CASE(2, 1, 64, 8, __VA_ARGS__)

static int64_t own_extra_units(const OwnArgs& a) { return a.listed ? (a.dense ? a.n_rows : a.n_pairs) : 0; }

int launch_own_step(const kge_model_desc* m, const int32_t* pairs, int64_t n_pairs, const kge_pull_lists* lists, const int32_t* items,
                    int64_t n_items, const uint32_t* listed, const int32_t* inc, float* partials, int dense, float lmbda, int reg_type,
                    int reset_lists, const int32_t* next_pairs, const int32_t* next_inv, int64_t next_n, const float* bern,
                    const uint64_t* slots, int64_t n_slots, uint64_t seed, uint64_t next_offset, const kge_pull_lists* next_lists,
                    float* loss, float* stage, hipStream_t s) {
    OwnArgs a;
    OwnGeo geo;
    if (fill_own_args(m, nullptr, nullptr, &a, &geo, "kge_own_step")) return -1;
    a.stage = stage; a.multi_only = 0;
    a.pairs = (const int4*)pairs; a.lists = to_lists(lists); a.items = (const int4*)items; a.inc = inc; a.partials = partials;
    a.multi = nullptr; a.n_items = n_items; a.n_multi = 0; a.listed = listed; a.n_pairs = (int)n_pairs; a.dense = dense ? 1 : 0;
    a.reset_lists = reset_lists; a.inv_n = 1.0f / (float)(2 * n_pairs); a.lmbda = lmbda; a.reg_type = reg_type;
    a.opt = make_opt_args(0.f, 1);
    PullSampleArgs sa = make_sample_args(next_pairs, next_inv, next_pairs && next_lists ? next_n : 0, m->tot_entity, bern, slots,
                                         n_slots, seed, next_offset, nullptr, next_lists);
    a.sample_blocks = sa.n > 0 ? (int)((sa.n + kBlock - 1) / kBlock) : 0;
    const int64_t units = n_items + own_extra_units(a);
    if (stage) {   // staged form: evaluate every bundle once, then the owners add the staged rows
        KGE_OWN_GEO({
            const int save = a.sample_blocks;
            a.sample_blocks = 0;
            hipLaunchKernelGGL((k_own_eval<NT, VEC, G, NV>), dim3((unsigned)((n_pairs + kBlock / G - 1) / (kBlock / G))), dim3(kBlock), 0, s, a, loss);
            a.sample_blocks = save;
            const int64_t blocks = (units + kBlock / G - 1) / (kBlock / G) + a.sample_blocks;
            hipLaunchKernelGGL((k_own_step<NT, VEC, G, NV, true>), dim3((unsigned)blocks), dim3(kBlock), 0, s, a, sa, loss);
        })
        return check_launch("k_own_step (staged)");
    }
    KGE_OWN_GEO({
        const int64_t blocks = (units + kBlock / G - 1) / (kBlock / G) + a.sample_blocks;
        hipLaunchKernelGGL((k_own_step<NT, VEC, G, NV>), dim3((unsigned)blocks), dim3(kBlock), 0, s, a, sa, loss);
    })
    return check_launch("k_own_step");
}

// the staged step with the optimiser fused into the owners (kge_own_run): evaluate every bundle once, then every owner adds its
// staged rows and updates its row set in place; only rows cut across workgroups are left to k_own_apply (multi_only)
template <int OPT>
static int launch_own_fused_opt(OwnArgs& a, OwnGeo geo, PullSampleArgs& sa, int64_t n_pairs, float* loss, hipStream_t s) {
    const int64_t units = a.n_items + own_extra_units(a);
    KGE_OWN_GEO({
        const int save = a.sample_blocks;
        a.sample_blocks = 0;
        hipLaunchKernelGGL((k_own_eval<NT, VEC, G, NV>), dim3((unsigned)((n_pairs + kBlock / G - 1) / (kBlock / G))), dim3(kBlock), 0, s, a, loss);
        a.sample_blocks = save;
        const int64_t blocks = (units + kBlock / G - 1) / (kBlock / G) + a.sample_blocks;
        hipLaunchKernelGGL((k_own_step<NT, VEC, G, NV, true, OPT>), dim3((unsigned)blocks), dim3(kBlock), 0, s, a, sa, loss);
    })
    return check_launch("k_own_step (staged, fused optimiser)");
}

int launch_own_step_fused(const kge_model_desc* m, float* const* state1, float* const* state2, const int32_t* pairs, int64_t n_pairs,
                          const kge_pull_lists* lists, const int32_t* items, int64_t n_items, const uint32_t* listed, const int32_t* inc,
                          float* partials, int dense, float lmbda, int reg_type, int optimizer, float lr, int64_t step,
                          const int32_t* next_pairs, const int32_t* next_inv, int64_t next_n, const float* bern, const uint64_t* slots,
                          int64_t n_slots, uint64_t seed, uint64_t next_offset, const kge_pull_lists* next_lists, float* loss,
                          float* stage, hipStream_t s) {
    OwnArgs a;
    OwnGeo geo;
    if (fill_own_args(m, state1, state2, &a, &geo, "kge_own_run")) return -1;
    for (int t = 0; t < geo.NT; ++t) {
        if (optimizer != KGE_OPT_SGD && (!a.s1_ent[t] || !a.s1_rel[t])) { set_error("kge_own_run: optimizer state missing"); return -1; }
        if (optimizer == KGE_OPT_ADAM && (!a.s2_ent[t] || !a.s2_rel[t])) { set_error("kge_own_run: adam needs two state buffers"); return -1; }
    }
    a.stage = stage; a.multi_only = 0;
    a.pairs = (const int4*)pairs; a.lists = to_lists(lists); a.items = (const int4*)items; a.inc = inc; a.partials = partials;
    a.multi = nullptr; a.n_items = n_items; a.n_multi = 0; a.listed = listed; a.n_pairs = (int)n_pairs; a.dense = dense ? 1 : 0;
    a.reset_lists = 1; a.inv_n = 1.0f / (float)(2 * n_pairs); a.lmbda = lmbda; a.reg_type = reg_type;
    a.opt = make_opt_args(lr, step < 1 ? 1 : step);
    PullSampleArgs sa = make_sample_args(next_pairs, next_inv, next_pairs && next_lists ? next_n : 0, m->tot_entity, bern, slots,
                                         n_slots, seed, next_offset, nullptr, next_lists);
    a.sample_blocks = sa.n > 0 ? (int)((sa.n + kBlock - 1) / kBlock) : 0;
    switch (optimizer) {
        case KGE_OPT_SGD: return launch_own_fused_opt<KGE_OPT_SGD>(a, geo, sa, n_pairs, loss, s);
        case KGE_OPT_ADAM: return launch_own_fused_opt<KGE_OPT_ADAM>(a, geo, sa, n_pairs, loss, s);
        case KGE_OPT_ADAGRAD: return launch_own_fused_opt<KGE_OPT_ADAGRAD>(a, geo, sa, n_pairs, loss, s);
        case KGE_OPT_RMSPROP: return launch_own_fused_opt<KGE_OPT_RMSPROP>(a, geo, sa, n_pairs, loss, s);
    }
    set_error("kge_own_run: unknown optimizer %d", optimizer);
    return -1;
}

template <int OPT>
static int launch_own_apply_opt(OwnArgs& a, OwnGeo geo, hipStream_t s) {
    const int64_t units = a.multi_only ? 0 : a.n_items + own_extra_units(a);
    KGE_OWN_GEO({
        const int64_t blocks = a.n_multi + (units + kBlock / G - 1) / (kBlock / G);
        if (blocks > 0) hipLaunchKernelGGL((k_own_apply<OPT, NT, VEC, G, NV>), dim3((unsigned)blocks), dim3(kBlock), 0, s, a);
    })
    return check_launch("k_own_apply");
}

int launch_own_apply(const kge_model_desc* m, float* const* state1, float* const* state2, const int32_t* pairs, int64_t n_pairs,
                     const kge_pull_lists* lists, const int32_t* items, int64_t n_items, const uint32_t* listed, const int32_t* multi,
                     int64_t n_multi, float* partials, int dense, int optimizer, float lr, int64_t step, int multi_only, hipStream_t s) {
    OwnArgs a;
    OwnGeo geo;
    if (fill_own_args(m, state1, state2, &a, &geo, "kge_own_apply")) return -1;
    a.multi_only = multi_only;
    const int NT = geo.NT;
    for (int t = 0; t < NT; ++t) {
        if (optimizer != KGE_OPT_SGD && (!a.s1_ent[t] || !a.s1_rel[t])) { set_error("kge_own_apply: optimizer state missing"); return -1; }
        if (optimizer == KGE_OPT_ADAM && (!a.s2_ent[t] || !a.s2_rel[t])) { set_error("kge_own_apply: adam needs two state buffers"); return -1; }
    }
    a.pairs = (const int4*)pairs; a.lists = to_lists(lists); a.items = (const int4*)items; a.inc = nullptr; a.partials = partials;
    a.multi = (const int4*)multi; a.n_items = n_items; a.n_multi = n_multi; a.listed = listed; a.n_pairs = (int)n_pairs;
    a.dense = dense ? 1 : 0; a.reset_lists = 0; a.inv_n = 0.f; a.lmbda = 0.f; a.reg_type = 0; a.sample_blocks = 0; a.stage = nullptr;
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
int own_partial_stride(int model, int dim) { const OwnGeo g = own_geo(model, dim); return g.G ? g.NT * g.VEC * g.NV * g.G : 0; }

}  // namespace kge
