// kge_pull.hip -- the owner-computes ("pull") training step for TransE: the whole of
//     Generator (data/generator.py:42-97) + Trainer.train_step_pairwise (utils/trainer.py:147-157) +
//     Criterion.pairwise_hinge (utils/criterion.py:25-29) + loss.backward() + optimizer.step() (utils/trainer.py:298-299)
// with NO atomics on parameters or gradients, no gradient buffer, bit-reproducible results.
//
// Why.  The push-style step (kge_score.hip::k_transe_pair_sampled) computes a pair once and scatters 3.25 gradient rows
// per pair with global_atomic_add_f32.  Those atomics execute memory-side at ~0.7-0.9 TB/s of payload (tools/atomic_bench.hip)
// and are what the kernel waits for (profiles/r01*): at B = 32768 pairs it moves 40 MB of atomic payload per step while
// the VALUs idle ~90 % of the time and the 6.5 MB of tables sit in L2 / Infinity Cache.  This file trades the scarce
// resource for the idle ones: every parameter ROW has one owner group per step which
//   * walks the row's incidence list -- the (pair, role) occurrences of the row in this batch: as head, as tail, as
//     relation (static per batch, indexed once when the generator is built: a batch is a fixed slice of the permutation,
//     data/generator.py:23-35) and as corrupting entity (per step: the sampler threads its draws into per-entity linked
//     lists, consumed in pair order),
//   * RE-computes each incident pair's forward (three L2-resident row gathers + pre-computed row norms: one butterfly
//     reduction per incidence) and keeps only the gradient of ITS OWN row, summed in registers in a fixed order,
//   * applies the normalisation backward once, then the dense optimiser update of that row (torch.optim semantics, every
//     row every step), writes the new row into the other half of a double-buffered table, and leaves the new row norm
//     for the next step.
// Each pair is evaluated ~3.25 times instead of once (cheap: VALU + L2 reads), nothing is scattered, the optimiser pass
// and the gradient zeroing pass disappear.  Rows with long lists (frequent relations, hub entities) are cut into
// segments of at most kPullSegment incidences: each segment's owner writes a partial sum, a small second kernel adds the
// partials in segment order and finishes the row -- still deterministic.
#include "kge_row_kernels.h"
#include "kge_opt_device.h"
#include "kge_sampler_device.h"

namespace kge {

constexpr int kRoleH = 0, kRoleT = 1, kRoleR = 2, kRoleC = 3;

struct PullArgs {
    const float* tab_in[2];    // entity / relation table read by this step
    float* tab_out[2];         // the tables the step writes (other half of the double buffer)
    const float* norm_in;      // [E + R] L2 norms of the rows of tab_in (entities first)
    float* norm_out;           // norms of the rows of tab_out
    float* s1[2];              // optimiser state, same row layout as the tables (NULL where the optimiser has none)
    float* s2[2];
    const int4* pairs;         // this batch: (h, r, t, -)
    const int32_t* pc;         // per pair: corrupting entity | (tail corrupted) << 24
    int32_t* head;             // per entity: most recent pair that drew it as corrupting entity, -1 = none
    const int32_t* next;       // per pair: the previous pair that drew the same entity, -1 = none
    const int4* items;         // work items: (row g, first incidence, end incidence, kind | slot << 2)
    const int32_t* inc;        // static incidences of the batch sorted by (row, pair, role): pair << 2 | role
    float* partials;           // [slots][G * NCH] partial gradient sums of multi-segment rows
    const int4* multi;         // rows with several segments: (row g, first slot, number of slots, -)
    int64_t n_items, n_multi;
    int E, d, l1, reset_lists;
    float margin;
    OptArgs opt;
    const float* dev_hyper;    // optional device-resident {lr, step_size, bc2_sqrt}
};

// gradient wrt the NORMALISED own row, summed over incidences -> normalisation backward -> optimiser -> new row + norm
template <int OPT, int G, int NCH>
__device__ __forceinline__ void pull_finish_row(const PullArgs& a, int g, const float (&X)[NCH], float nX, const float (&gs)[NCH],
                                                int gl) {
    const int d = a.d;
    const bool is_rel = g >= a.E;
    const int tb = is_rel ? 1 : 0;
    const int64_t off = (int64_t)(is_rel ? g - a.E : g) * d;
    const bool fX = nX > kEpsNormalize;
    const float iX = 1.0f / fmaxf(nX, kEpsNormalize);
    float dX = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) dX = fmaf(X[k], gs[k], dX);
    dX = gsum<G>(dX) * iX;
    OptArgs o = a.opt;
    if (a.dev_hyper) { o.lr = a.dev_hyper[0]; o.step_size = a.dev_hyper[1]; o.bc2_sqrt = a.dev_hyper[2]; }
    float n2 = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int e = k * G + gl;
        const bool live = e < d;
        const float graw = fX ? (gs[k] - (X[k] * iX) * dX) * iX : gs[k] * iX;
        float p = X[k], m1 = 0.f, m2 = 0.f;
        if (live) {
            if constexpr (OPT != KGE_OPT_SGD) m1 = a.s1[tb][off + e];
            if constexpr (OPT == KGE_OPT_ADAM) m2 = a.s2[tb][off + e];
            opt_update<OPT>(p, graw, m1, m2, o);
            a.tab_out[tb][off + e] = p;
            if constexpr (OPT != KGE_OPT_SGD) a.s1[tb][off + e] = m1;
            if constexpr (OPT == KGE_OPT_ADAM) a.s2[tb][off + e] = m2;
            n2 = fmaf(p, p, n2);
        }
    }
    n2 = gsum<G>(n2);
    if (gl == 0) a.norm_out[g] = sqrtf(n2);
}

template <int OPT, int G, int NCH>
__global__ __launch_bounds__(kBlock) void k_pull_step(PullArgs a, float* __restrict__ loss) {
    constexpr int GPB = kBlock / G;
    const int gl = threadIdx.x % G;
    const int d = a.d;
    const bool l1 = a.l1 != 0;
    const int64_t item = (int64_t)blockIdx.x * GPB + threadIdx.x / G;
    float acc = 0.f;
    if (item < a.n_items) {
        const int4 it = a.items[item];
        const int g = it.x;
        const int kind = it.w & 3;
        const bool is_rel = g >= a.E;
        float X[NCH], Xh[NCH], gs[NCH];
        load_row<G, NCH>(X, a.tab_in[is_rel ? 1 : 0] + (int64_t)(is_rel ? g - a.E : g) * d, d, gl);
        const float nX = a.norm_in[g];
        {
            const float iX = 1.0f / fmaxf(nX, kEpsNormalize);
#pragma unroll
            for (int k = 0; k < NCH; ++k) { Xh[k] = X[k] * iX; gs[k] = 0.f; }
        }
        // one incidence: pair i seen from role `role` (group-uniform)
        auto visit = [&](int i, int role) {
            const int4 pr = a.pairs[i];
            const int pcv = a.pc[i];
            const int c = pcv & 0xFFFFFF;
            const bool tail = (pcv >> 24) != 0;
            float hh[NCH], rr[NCH], tt[NCH], cc[NCH];
            // the three rows that are not the owner's: gather + scale by the stored norm (own row: registers)
            if (role != kRoleH) {
                load_row<G, NCH>(hh, a.tab_in[0] + (int64_t)pr.x * d, d, gl);
                const float s = 1.0f / fmaxf(a.norm_in[pr.x], kEpsNormalize);
#pragma unroll
                for (int k = 0; k < NCH; ++k) hh[k] *= s;
            } else {
#pragma unroll
                for (int k = 0; k < NCH; ++k) hh[k] = Xh[k];
            }
            if (role != kRoleR) {
                load_row<G, NCH>(rr, a.tab_in[1] + (int64_t)pr.y * d, d, gl);
                const float s = 1.0f / fmaxf(a.norm_in[a.E + pr.y], kEpsNormalize);
#pragma unroll
                for (int k = 0; k < NCH; ++k) rr[k] *= s;
            } else {
#pragma unroll
                for (int k = 0; k < NCH; ++k) rr[k] = Xh[k];
            }
            if (role != kRoleT) {
                load_row<G, NCH>(tt, a.tab_in[0] + (int64_t)pr.z * d, d, gl);
                const float s = 1.0f / fmaxf(a.norm_in[pr.z], kEpsNormalize);
#pragma unroll
                for (int k = 0; k < NCH; ++k) tt[k] *= s;
            } else {
#pragma unroll
                for (int k = 0; k < NCH; ++k) tt[k] = Xh[k];
            }
            if (role != kRoleC) {
                load_row<G, NCH>(cc, a.tab_in[0] + (int64_t)c * d, d, gl);
                const float s = 1.0f / fmaxf(a.norm_in[c], kEpsNormalize);
#pragma unroll
                for (int k = 0; k < NCH; ++k) cc[k] *= s;
            } else {
#pragma unroll
                for (int k = 0; k < NCH; ++k) cc[k] = Xh[k];
            }
            float up[NCH], un[NCH];
            float sp = 0.f, sn = 0.f;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                up[k] = hh[k] + rr[k] - tt[k];
                un[k] = tail ? (hh[k] + rr[k] - cc[k]) : (cc[k] + rr[k] - tt[k]);
                sp = l1 ? sp + fabsf(up[k]) : fmaf(up[k], up[k], sp);
                sn = l1 ? sn + fabsf(un[k]) : fmaf(un[k], un[k], sn);
            }
            gsum2<G>(sp, sn);
            if (!l1) { sp = sqrtf(sp); sn = sqrtf(sn); }
            const float v = sp + a.margin - sn;
            if (role == kRoleH) acc += fmaxf(v, 0.f);   // every pair has exactly one head incidence: the loss is counted there
            const float coef = v > 0.f ? 1.f : (v == 0.f ? 0.5f : 0.f);   // torch.max splits the subgradient at equality
            if (coef == 0.f) return;
            const float ip = (!l1 && sp > 0.f) ? coef / sp : 0.f, in = (!l1 && sn > 0.f) ? -coef / sn : 0.f;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const float gp = l1 ? (up[k] > 0.f ? coef : (up[k] < 0.f ? -coef : 0.f)) : up[k] * ip;
                const float gn = l1 ? (un[k] > 0.f ? -coef : (un[k] < 0.f ? coef : 0.f)) : un[k] * in;
                float gx;
                if (role == kRoleH) gx = tail ? gp + gn : gp;
                else if (role == kRoleT) gx = tail ? -gp : -(gp + gn);
                else if (role == kRoleR) gx = gp + gn;
                else gx = tail ? -gn : gn;
                gs[k] += gx;
            }
        };
        for (int p = it.y; p < it.z; ++p) {
            const int e = a.inc[p];
            visit(e >> 2, e & 3);
        }
        if (!is_rel && kind != 2) {
            // pairs whose sampler drew this entity: a linked list in arrival (i.e. arbitrary) order; visit in pair order
            int last = -1;
            for (;;) {
                int best = 0x7FFFFFFF;
                for (int j = a.head[g]; j >= 0; j = a.next[j])
                    if (j > last && j < best) best = j;
                if (best == 0x7FFFFFFF) break;
                visit(best, kRoleC);
                last = best;
            }
            if (a.reset_lists && gl == 0) a.head[g] = -1;
        }
        if (kind == 0) {
            pull_finish_row<OPT, G, NCH>(a, g, X, nX, gs, gl);
        } else {
            float* out = a.partials + (int64_t)(it.w >> 2) * (G * NCH);
#pragma unroll
            for (int k = 0; k < NCH; ++k) out[k * G + gl] = gs[k];
        }
    }
    block_accumulate_loss<G>(acc, gl, loss);
}

// rows cut into several segments: add the segments' partial sums in segment order, then finish the row
template <int OPT, int G, int NCH>
__global__ __launch_bounds__(kBlock) void k_pull_finish(PullArgs a) {
    constexpr int GPB = kBlock / G;
    const int gl = threadIdx.x % G;
    const int64_t m = (int64_t)blockIdx.x * GPB + threadIdx.x / G;
    if (m >= a.n_multi) return;
    const int4 row = a.multi[m];
    const int g = row.x;
    const bool is_rel = g >= a.E;
    float X[NCH], gs[NCH];
    load_row<G, NCH>(X, a.tab_in[is_rel ? 1 : 0] + (int64_t)(is_rel ? g - a.E : g) * a.d, a.d, gl);
#pragma unroll
    for (int k = 0; k < NCH; ++k) gs[k] = 0.f;
    for (int s = 0; s < row.z; ++s) {
        const float* in = a.partials + (int64_t)(row.y + s) * (G * NCH);
#pragma unroll
        for (int k = 0; k < NCH; ++k) gs[k] += in[k * G + gl];
    }
    pull_finish_row<OPT, G, NCH>(a, g, X, a.norm_in[g], gs, gl);
}

// L2 norms of the rows of a table, in the lane layout / operation order pull_finish_row uses
template <int G, int NCH>
__global__ __launch_bounds__(kBlock) void k_row_norms(const float* __restrict__ tab, int64_t rows, int d, float* __restrict__ out) {
    constexpr int GPB = kBlock / G;
    const int gl = threadIdx.x % G;
    const int64_t r = (int64_t)blockIdx.x * GPB + threadIdx.x / G;
    if (r >= rows) return;
    float X[NCH];
    load_row<G, NCH>(X, tab + r * d, d, gl);
    float n2 = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) n2 = fmaf(X[k], X[k], n2);
    n2 = gsum<G>(n2);
    if (gl == 0) out[r] = sqrtf(n2);
}

// per pair: draw the corruption (same Philox counters as kge_sample_batch / the fused push kernels: offset + pair index)
// and thread the pair into the corrupting entity's list
__global__ __launch_bounds__(256) void k_pull_sample(const int4* __restrict__ pairs, int64_t n, int64_t E,
                                                     const float* __restrict__ bern, const unsigned long long* __restrict__ slots,
                                                     unsigned long long mask, unsigned long long seed, unsigned long long offset,
                                                     const int64_t* __restrict__ cursor, int32_t* __restrict__ pc,
                                                     int32_t* __restrict__ head, int32_t* __restrict__ next) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long off = cursor ? offset + (unsigned long long)cursor[1] : offset;
    const int4 p = pairs[i];
    int64_t nh, nt;
    corrupt_one(p.x, p.y, p.z, E, bern, slots, mask, seed, off + (unsigned long long)i, nh, nt);
    const bool tail = nh == p.x;
    const int c = (int)(tail ? nt : nh);
    pc[i] = c | ((int)tail << 24);
    next[i] = atomicExch(head + c, (int)i);
}

// the same lists from explicit negatives (parity tests drive the step with the reference's golden batches)
__global__ __launch_bounds__(256) void k_pull_lists_explicit(const int4* __restrict__ pairs, const int64_t* __restrict__ nh,
                                                             const int64_t* __restrict__ nt, int64_t n, int32_t* __restrict__ pc,
                                                             int32_t* __restrict__ head, int32_t* __restrict__ next) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool tail = nh[i] == pairs[i].x;   // the sampler's own rule (kge_score.hip: my_tail = nh == sh)
    const int c = (int)(tail ? nt[i] : nh[i]);
    pc[i] = c | ((int)tail << 24);
    next[i] = atomicExch(head + c, (int)i);
}

// ------------------------------------------------------------------ host side
template <int OPT, int G, int NCH>
static int launch_pull_geo(const PullArgs& a, float* loss, hipStream_t s) {
    constexpr int GPB = kBlock / G;
    hipLaunchKernelGGL((k_pull_step<OPT, G, NCH>), dim3((unsigned)((a.n_items + GPB - 1) / GPB)), dim3(kBlock), 0, s, a, loss);
    int rc = check_launch("k_pull_step");
    if (rc || a.n_multi == 0) return rc;
    hipLaunchKernelGGL((k_pull_finish<OPT, G, NCH>), dim3((unsigned)((a.n_multi + GPB - 1) / GPB)), dim3(kBlock), 0, s, a);
    return check_launch("k_pull_finish");
}

template <int OPT>
static int launch_pull_opt(const PullArgs& a, Geometry geo, float* loss, hipStream_t s) {
#define KGE_PULL(G_, NCH_) if (geo.G == G_ && geo.NCH == NCH_) return launch_pull_geo<OPT, G_, NCH_>(a, loss, s);
    KGE_PULL(32, 1) KGE_PULL(32, 2) KGE_PULL(32, 4) KGE_PULL(32, 8) KGE_PULL(64, 8) KGE_PULL(64, 16)
#undef KGE_PULL
    return -1;
}

int pull_partial_stride(int dim) {
    Geometry geo;
    if (!pick_geometry(dim, &geo)) return 0;
    return geo.G * geo.NCH;
}

int launch_pull_step(const kge_model_desc* m, float* const tables_out[2], const float* norm_in, float* norm_out,
                     float* const state1[2], float* const state2[2], const int32_t* pairs, const int32_t* pc, int32_t* head,
                     const int32_t* next, const int32_t* items, int64_t n_items, const int32_t* inc, float* partials,
                     const int32_t* multi, int64_t n_multi, float margin, int optimizer, float lr, int64_t step,
                     const float* dev_hyper, int reset_lists, float* loss, hipStream_t s) {
    Geometry geo;
    if (!pick_geometry(m->dim, &geo)) { set_error("kge_pull_step: hidden size %d exceeds the register-resident rows", m->dim); return -1; }
    PullArgs a;
    for (int i = 0; i < 2; ++i) {
        a.tab_in[i] = m->tables[i]; a.tab_out[i] = tables_out[i];
        a.s1[i] = state1 ? state1[i] : nullptr; a.s2[i] = state2 ? state2[i] : nullptr;
    }
    a.norm_in = norm_in; a.norm_out = norm_out;
    a.pairs = (const int4*)pairs; a.pc = pc; a.head = head; a.next = next;
    a.items = (const int4*)items; a.inc = inc; a.partials = partials; a.multi = (const int4*)multi;
    a.n_items = n_items; a.n_multi = n_multi;
    a.E = (int)m->tot_entity; a.d = m->dim; a.l1 = (m->flags & KGE_FLAG_L1) ? 1 : 0; a.reset_lists = reset_lists;
    a.margin = margin;
    a.opt = make_opt_args(lr, step < 1 ? 1 : step);
    a.dev_hyper = dev_hyper;
    switch (optimizer) {
        case KGE_OPT_SGD: return launch_pull_opt<KGE_OPT_SGD>(a, geo, loss, s);
        case KGE_OPT_ADAM: return launch_pull_opt<KGE_OPT_ADAM>(a, geo, loss, s);
        case KGE_OPT_ADAGRAD: return launch_pull_opt<KGE_OPT_ADAGRAD>(a, geo, loss, s);
        case KGE_OPT_RMSPROP: return launch_pull_opt<KGE_OPT_RMSPROP>(a, geo, loss, s);
    }
    set_error("kge_pull_step: unknown optimizer %d", optimizer);
    return -1;
}

int launch_row_norms(const float* table, int64_t rows, int dim, float* out, hipStream_t s) {
    Geometry geo;
    if (!pick_geometry(dim, &geo)) { set_error("kge_row_norms: row length %d too long", dim); return -1; }
    if (rows == 0) return 0;
#define KGE_RN(G_, NCH_)                                                                                                   \
    if (geo.G == G_ && geo.NCH == NCH_) {                                                                                   \
        hipLaunchKernelGGL((k_row_norms<G_, NCH_>), dim3((unsigned)((rows + kBlock / G_ - 1) / (kBlock / G_))), dim3(kBlock), 0, s, \
                           table, rows, dim, out);                                                                          \
        return check_launch("k_row_norms");                                                                                 \
    }
    KGE_RN(32, 1) KGE_RN(32, 2) KGE_RN(32, 4) KGE_RN(32, 8) KGE_RN(64, 8) KGE_RN(64, 16)
#undef KGE_RN
    return -1;
}

int launch_pull_sample(const int32_t* pairs, int64_t n, int64_t E, const float* bern, const uint64_t* slots, int64_t n_slots,
                       uint64_t seed, uint64_t offset, const int64_t* cursor, int32_t* pc, int32_t* head, int32_t* next,
                       hipStream_t s) {
    hipLaunchKernelGGL(k_pull_sample, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const int4*)pairs, n, E, bern,
                       (const unsigned long long*)slots, (unsigned long long)(slots ? n_slots - 1 : 0), seed, offset, cursor, pc,
                       head, next);
    return check_launch("k_pull_sample");
}

int launch_pull_lists_explicit(const int32_t* pairs, const int64_t* nh, const int64_t* nt, int64_t n, int32_t* pc,
                               int32_t* head, int32_t* next, hipStream_t s) {
    hipLaunchKernelGGL(k_pull_lists_explicit, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const int4*)pairs, nh, nt, n,
                       pc, head, next);
    return check_launch("k_pull_lists_explicit");
}

}  // namespace kge
