// kge_row_kernels.h -- the model-generic row kernels (one G-lane group per triple, roles table in kge_device.h):
// forward, backward, fused pairwise hinge (optionally with the sampler in front), fused pointwise logistic (plain and
// bundled) and the self-adversarial bundle.  Included by the translation units that instantiate them for their
// model lists (kge_score.hip, kge_score_ext.hip).
#pragma once
#include "kge_internal.h"
#include "kge_sampler_device.h"

namespace kge {

constexpr int kBlock = 256;
constexpr int kMaxBlocks = 256 * 8;  // 256 CUs x 8 resident 256-thread blocks; rest is grid-stride

__device__ __forceinline__ float softplus_t(float x) { return x > 20.f ? x : log1pf(expf(x)); }  // F.softplus, threshold 20
__device__ __forceinline__ float sigmoid_t(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float logsigmoid_t(float x) { return fminf(x, 0.f) - log1pf(expf(-fabsf(x))); }

template <int M, int G, int NCH>
__global__ __launch_bounds__(kBlock) void k_score_fwd(DeviceModel m, const int64_t* __restrict__ h,
                                                      const int64_t* __restrict__ r, const int64_t* __restrict__ t,
                                                      int64_t n, float* __restrict__ out) {
    constexpr int GPB = kBlock / G;
    const int gl = threadIdx.x % G;
    for (int64_t i = (int64_t)blockIdx.x * GPB + threadIdx.x / G; i < n; i += (int64_t)gridDim.x * GPB) {
        const int64_t id[3] = {h[i], r[i], t[i]};
        Rows<M, NCH> R;
        load_rows<M, G, NCH>(R, m, id, gl);
        Saved<M, NCH> sv;
        const float s = model_fwd<M, G, NCH>(R, m, sv);
        if (gl == 0) out[i] = s;
    }
}

template <int M, int G, int NCH>
__global__ __launch_bounds__(kBlock) void k_score_bwd(DeviceModel m, const int64_t* __restrict__ h,
                                                      const int64_t* __restrict__ r, const int64_t* __restrict__ t,
                                                      int64_t n, const float* __restrict__ dscore) {
    constexpr int GPB = kBlock / G;
    const int gl = threadIdx.x % G;
    for (int64_t i = (int64_t)blockIdx.x * GPB + threadIdx.x / G; i < n; i += (int64_t)gridDim.x * GPB) {
        const float ds = dscore[i];
        if (ds == 0.f) continue;  // group-uniform
        const int64_t id[3] = {h[i], r[i], t[i]};
        Rows<M, NCH> R;
        load_rows<M, G, NCH>(R, m, id, gl);
        Saved<M, NCH> sv;
        model_fwd<M, G, NCH>(R, m, sv);  // recompute: cheaper than spilling [B,d] intermediates to HBM
        Rows<M, NCH> Gr;
        model_bwd<M, G, NCH>(R, m, sv, ds, Gr);
        scatter_rows<M, G, NCH>(Gr, m, id, gl);
    }
}

// ---- fused pairwise hinge step: score(+) , score(-), max(0, s+ + margin - s-), both backward passes.
// A negative produced by the reference sampler shares the relation and one entity with its positive
// (data/generator.py:71-95); rows with equal ids get ONE combined atomic scatter (4 row scatters, not 6).
// SAMPLED = true fuses the negative sampler in front (north_star: corruption + both scores + margin ranking + backward
// in ONE kernel): the positive is triples[perm[start+i]] and its negative is drawn here by corrupt_one() with the same
// Philox counters as the stand-alone sampler, so kge_sample_batch + kge_train_pairwise_hinge and this kernel see
// identical batches.  Every lane of a group runs the (scalar) draw redundantly: no shuffle, no divergence.
struct FusedSampler {
    const int64_t* triples; const int64_t* perm; int64_t start; int64_t E;
    const float* bern; const unsigned long long* slots; unsigned long long mask; unsigned long long seed, offset;
    const int64_t* cursor;
};

template <int M, int G, int NCH, bool SAMPLED>
__global__ __launch_bounds__(kBlock) void k_pairwise_hinge(DeviceModel m, const int64_t* __restrict__ ph,
                                                           const int64_t* __restrict__ pr, const int64_t* __restrict__ pt,
                                                           const int64_t* __restrict__ nh, const int64_t* __restrict__ nr,
                                                           const int64_t* __restrict__ nt, int64_t n, float margin,
                                                           float* __restrict__ loss, FusedSampler fs) {
    constexpr int GPB = kBlock / G;
    constexpr int NR = role_count(M);
    const int gl = threadIdx.x % G;
    float acc = 0.f;
    int64_t s_start = 0;
    unsigned long long s_off = 0;
    if constexpr (SAMPLED) {
        s_start = fs.cursor ? fs.start + fs.cursor[0] : fs.start;
        s_off = fs.cursor ? fs.offset + (unsigned long long)fs.cursor[1] : fs.offset;
    }
    for (int64_t i = (int64_t)blockIdx.x * GPB + threadIdx.x / G; i < n; i += (int64_t)gridDim.x * GPB) {
        int64_t idp[3], idn[3];
        if constexpr (SAMPLED) {
            const int64_t row = fs.perm[s_start + i];
            idp[0] = fs.triples[3 * row]; idp[1] = fs.triples[3 * row + 1]; idp[2] = fs.triples[3 * row + 2];
            idn[1] = idp[1];
            corrupt_one(idp[0], idp[1], idp[2], fs.E, fs.bern, fs.slots, fs.mask, fs.seed, s_off + (unsigned long long)i,
                        idn[0], idn[2]);
        } else {
            idp[0] = ph[i]; idp[1] = pr[i]; idp[2] = pt[i];
            idn[0] = nh[i]; idn[1] = nr[i]; idn[2] = nt[i];
        }
        Rows<M, NCH> Rp, Rn;
        load_rows<M, G, NCH>(Rp, m, idp, gl);
        load_rows<M, G, NCH>(Rn, m, idn, gl);
        Saved<M, NCH> svp, svn;
        const float sp = model_fwd<M, G, NCH>(Rp, m, svp);
        const float sn = model_fwd<M, G, NCH>(Rn, m, svn);
        const float v = sp + margin - sn;
        acc += fmaxf(v, 0.f);
        const float coef = v > 0.f ? 1.f : (v == 0.f ? 0.5f : 0.f);  // ATen max(a, 0) backward splits ties
        if (coef != 0.f) {
            Rows<M, NCH> Gp, Gn;
            model_bwd<M, G, NCH>(Rp, m, svp, coef, Gp);
            model_bwd<M, G, NCH>(Rn, m, svn, -coef, Gn);
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                if (!role_trainable(M, q)) continue;
                const int sel = role_sel(M, q);
                const int d = role_dim<M>(m, q);
                float* gt = m.grad[role_tab(M, q)];
                if (idp[sel] == idn[sel]) {
#pragma unroll
                    for (int c = 0; c < NCH; ++c) Gp.x[q][c] += Gn.x[q][c];
                    atomic_add_row<G, NCH>(gt + idp[sel] * (int64_t)d, Gp.x[q], d, gl);
                } else {
                    atomic_add_row<G, NCH>(gt + idp[sel] * (int64_t)d, Gp.x[q], d, gl);
                    atomic_add_row<G, NCH>(gt + idn[sel] * (int64_t)d, Gn.x[q], d, gl);
                }
            }
        }
    }
    block_accumulate_loss<G>(acc, gl, loss);
}


// SimplE's regulariser as the reference executes it: a function of the ids (see enum kge_reg)
__device__ __forceinline__ float id_reg_term(const int64_t (&id)[3], float lmbda, int reg_type) {
    const float a = (float)id[0], b = (float)id[1], c = (float)id[2];
    return reg_type == KGE_REG_ID_F2 ? lmbda * (a * a + b * b + c * c) : lmbda * (a * a * a + b * b * b + c * c * c);
}

// ---- fused pointwise step: mean(softplus(y*s)) + lmbda * mean_i(sum of squares/cubes of the rows of row i)
template <int M, int G, int NCH>
__global__ __launch_bounds__(kBlock) void k_pointwise_logistic(DeviceModel m, const int64_t* __restrict__ h,
                                                               const int64_t* __restrict__ r, const int64_t* __restrict__ t,
                                                               const int64_t* __restrict__ y, int64_t n, float lmbda,
                                                               int reg_type, float* __restrict__ loss) {
    constexpr int GPB = kBlock / G;
    constexpr int NR = role_count(M);
    const int gl = threadIdx.x % G;
    const float inv_n = 1.0f / (float)n;
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * GPB + threadIdx.x / G; i < n; i += (int64_t)gridDim.x * GPB) {
        const int64_t id[3] = {h[i], r[i], t[i]};
        const float yy = (float)y[i];
        Rows<M, NCH> R;
        load_rows<M, G, NCH>(R, m, id, gl);
        Saved<M, NCH> sv;
        const float s = model_fwd<M, G, NCH>(R, m, sv);
        const float x = yy * s;
        acc += softplus_t(x) * inv_n;
        const float ds = yy * sigmoid_t(x) * inv_n;
        Rows<M, NCH> Gr;
        model_bwd<M, G, NCH>(R, m, sv, ds, Gr);
        if (reg_type >= KGE_REG_ID_F2) {
            acc += id_reg_term(id, lmbda, reg_type);
        } else if (reg_type != KGE_REG_NONE) {
            float rs = 0.f;
            const float c2 = 2.f * lmbda * inv_n, c3 = 3.f * lmbda * inv_n;
#pragma unroll
            for (int q = 0; q < NR; ++q) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const float v = R.x[q][c];
                    if (reg_type == KGE_REG_F2) { rs = fmaf(v, v, rs); Gr.x[q][c] += c2 * v; }
                    else if (reg_type == KGE_REG_N3) { rs += v * v * v; Gr.x[q][c] += c3 * v * v; }
                    else { const float a = fabsf(v); rs += a * a * a; Gr.x[q][c] += c3 * v * a; }
                }
            }
            acc += lmbda * inv_n * gsum<G>(rs);
        }
        scatter_rows<M, G, NCH>(Gr, m, id, gl);
    }
    block_accumulate_loss<G>(acc, gl, loss);
}

// ---- pointwise step over BUNDLES: the reference sampler emits each positive followed by its neg_rate corruptions
// (data/generator.py:125-156), which share the relation row(s) and one entity's row(s) with it.  One group walks the
// `bundle` consecutive rows, accumulating in registers every gradient row whose id equals the bundle's first row's id
// for that role, and scatters those once: 1+neg_rate rows cost NR + neg_rate*(rows of one entity) scatters instead of
// (1+neg_rate)*NR.  Pure id-equality test, so it is exact for arbitrary input rows too.
// Batches are sorted by relation (generator), so a group also walks CHB consecutive bundles and carries the RELATION
// rows' gradients across them while the relation id does not change (few-relation graphs such as WN18RR otherwise
// funnel thousands of atomic row-adds per step into a dozen rows, i.e. into a handful of memory channels).
constexpr int kMaxChunkBundles = 8;
static inline int chunk_bundles(int64_t nb) {  // enough groups to fill the chip first, then amortise relation scatters
    int64_t c = nb / 2048;
    return (int)(c < 1 ? 1 : (c > kMaxChunkBundles ? kMaxChunkBundles : c));
}

// floats of relation-side gradient per relation id (sum of the row lengths of the roles selected by r)
template <int M>
__host__ __device__ inline int rel_span(int dim) {
    int s = 0;
    for (int q = 0; q < role_count(M); ++q)
        if (role_sel(M, q) == 1 && role_trainable(M, q)) s += (M == KGE_ANALOGY && q >= 3) ? dim / 2 : dim;
    return s;
}

static inline int rel_span_host(int model, int dim) {
    switch (model) {
        case KGE_DISTMULT: return rel_span<KGE_DISTMULT>(dim);
        case KGE_COMPLEX: return rel_span<KGE_COMPLEX>(dim);
        case KGE_ANALOGY: return rel_span<KGE_ANALOGY>(dim);
        case KGE_CP: return rel_span<KGE_CP>(dim);
        case KGE_SIMPLE: return rel_span<KGE_SIMPLE>(dim);
        case KGE_SIMPLE_IGNR: return rel_span<KGE_SIMPLE_IGNR>(dim);
        case KGE_QUATE: return rel_span<KGE_QUATE>(dim);
        case KGE_TRANSE: return rel_span<KGE_TRANSE>(dim);
        case KGE_TRANSH: return rel_span<KGE_TRANSH>(dim);
        case KGE_TRANSD: return rel_span<KGE_TRANSD>(dim);
        case KGE_TRANSM: return rel_span<KGE_TRANSM>(dim);
        case KGE_ROTATE: return rel_span<KGE_ROTATE>(dim);
        default: return 1 << 28;
    }
}

// LDSREL: graphs with a handful of relations (WN18RR: 11, YAGO3-10: 37) funnel every bundle's relation-row gradient into
// the same few rows -- thousands of same-address float atomics that the memory side serialises.  When R * rel_span floats
// fit in LDS the workgroup accumulates all relation rows there (ds_add_f32) and flushes the non-zero entries once at
// the end: each global relation-gradient address then receives one atomic per workgroup instead of one per bundle.
// position of role q among the roles that take their id from the same triple slot (h / r / t): the dynamic staging site
__host__ __device__ constexpr int role_rank_in_sel(int M, int q) {
    int k = 0;
    for (int j = 0; j < q; ++j) k += role_sel(M, j) == role_sel(M, q) ? 1 : 0;
    return k;
}

// STAGED (sampler-fused, one bundle per lane group, no LDS relation accumulation): instead of atomics into dense gradient
// tables every gradient row goes to its own staging slot with plain stores (kge_internal.h: StageSink) -- the bundle's
// anchor rows (all roles of the positive, with the negatives' contributions to shared rows merged in registers) to static
// slots b * NR + role, the rows of a negative's corrupted entity to dynamic slots, registered with that entity.
template <int M, int G, int NCH, bool LDSREL, bool STAGED = false>
__global__ __launch_bounds__(kBlock) void k_pointwise_bundle(DeviceModel m, const int64_t* __restrict__ h,
                                                             const int64_t* __restrict__ r, const int64_t* __restrict__ t,
                                                             const int64_t* __restrict__ y, int64_t n, int bundle, int CHB,
                                                             float lmbda, int reg_type, float* __restrict__ loss,
                                                             int64_t tot_relation, FusedSampler fs, StageSink sink = StageSink{}) {
    static_assert(!(STAGED && LDSREL), "staged output keeps relation rows per bundle");
    constexpr int GPB = kBlock / G;
    constexpr int NR = role_count(M);
    const int gl = threadIdx.x % G;
    // fs.triples != NULL: the sampler is fused in front (kge_train_pointwise_logistic_sampled): bundle b is the positive
    // triples[perm[start + b]] (y = +1) followed by bundle-1 corruptions (y = -1), negative j drawn by lane j of the group
    // with the Philox counter kge_sample_batch uses for it, so both paths see identical rows
    const bool sampled = fs.triples != nullptr;
    const int64_t s_start = (sampled && fs.cursor) ? fs.start + fs.cursor[0] : fs.start;
    const unsigned long long s_off = (sampled && fs.cursor) ? fs.offset + (unsigned long long)fs.cursor[1] : fs.offset;
    const int gbase = (threadIdx.x & 63) / G * G;
    extern __shared__ float s_rel[];  // LDSREL: [tot_relation][rel_span]
    const int span = LDSREL ? rel_span<M>(m.dim) : 0;
    if constexpr (LDSREL) {
        for (int64_t i = threadIdx.x; i < tot_relation * span; i += kBlock) s_rel[i] = 0.f;
        __syncthreads();
    }
    const float inv_n = 1.0f / (float)n;
    const float c2 = 2.f * lmbda * inv_n, c3 = 3.f * lmbda * inv_n;
    const int64_t nb = (n + bundle - 1) / bundle;
    const int64_t nchunks = (nb + CHB - 1) / CHB;
    float acc = 0.f;
    for (int64_t ck = (int64_t)blockIdx.x * GPB + threadIdx.x / G; ck < nchunks; ck += (int64_t)gridDim.x * GPB) {
        Rows<M, NCH> A;  // anchors: entity roles are reset per bundle, relation roles persist while the relation id holds
        int64_t ida[3] = {-1, -1, -1};
        auto flush_role = [&](int q) {
            const int sel = role_sel(M, q);
            if (ida[sel] < 0 || !role_trainable(M, q) || (LDSREL && sel == 1)) return;
            const int d = role_dim<M>(m, q);
            if constexpr (STAGED) store_row<G, NCH>(sink.stage + (ck * NR + q) * sink.stride, A.x[q], d, gl);   // CHB == 1: bundle ck
            else atomic_add_row<G, NCH>(m.grad[role_tab(M, q)] + ida[sel] * (int64_t)d, A.x[q], d, gl);
        };
        const int64_t b1 = min(nb, (ck + 1) * CHB);
        for (int64_t b = ck * CHB; b < b1; ++b) {
            const int64_t i0 = b * bundle;
            int64_t idb[3];
            int my_nh = 0, my_nt = 0;
            if (sampled) {
                const int64_t row = fs.perm[s_start + b];
                idb[0] = fs.triples[3 * row]; idb[1] = fs.triples[3 * row + 1]; idb[2] = fs.triples[3 * row + 2];
                if (gl < bundle - 1) {
                    int64_t nh, nt;
                    corrupt_one(idb[0], idb[1], idb[2], fs.E, fs.bern, fs.slots, fs.mask, fs.seed,
                                s_off + (unsigned long long)(b * (bundle - 1) + gl), nh, nt);
                    my_nh = (int)nh; my_nt = (int)nt;
                }
            } else {
                idb[0] = h[i0]; idb[1] = r[i0]; idb[2] = t[i0];
            }
            const bool same_rel = idb[1] == ida[1];  // group-uniform
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const int sel = role_sel(M, q);
                if (sel == 1 && same_rel) continue;  // keep accumulating this relation's rows
                flush_role(q);
#pragma unroll
                for (int c = 0; c < NCH; ++c) A.x[q][c] = 0.f;
            }
            ida[0] = idb[0]; ida[1] = idb[1]; ida[2] = idb[2];
            const int64_t i1 = min(n, i0 + bundle);
            for (int64_t i = i0; i < i1; ++i) {
                int64_t id[3];
                float yy;
                if (sampled) {
                    const int k = (int)(i - i0);
                    const int src = gbase + (k > 0 ? k - 1 : 0);
                    const int64_t nh = __shfl(my_nh, src, 64), nt = __shfl(my_nt, src, 64);
                    id[0] = k == 0 ? idb[0] : nh; id[1] = idb[1]; id[2] = k == 0 ? idb[2] : nt;
                    yy = k == 0 ? 1.f : -1.f;
                } else {
                    id[0] = h[i]; id[1] = r[i]; id[2] = t[i];
                    yy = (float)y[i];
                }
                Rows<M, NCH> R, Gr;
                Saved<M, NCH> sv;
                load_rows<M, G, NCH>(R, m, id, gl);
                const float s = model_fwd<M, G, NCH>(R, m, sv);
                const float x = yy * s;
                acc += softplus_t(x) * inv_n;
                model_bwd<M, G, NCH>(R, m, sv, yy * sigmoid_t(x) * inv_n, Gr);
                if (reg_type >= KGE_REG_ID_F2) {
                    acc += id_reg_term(id, lmbda, reg_type);
                } else if (reg_type != KGE_REG_NONE) {
                    float rs = 0.f;
#pragma unroll
                    for (int q = 0; q < NR; ++q) {
#pragma unroll
                        for (int c = 0; c < NCH; ++c) {
                            const float v = R.x[q][c];
                            if (reg_type == KGE_REG_F2) { rs = fmaf(v, v, rs); Gr.x[q][c] += c2 * v; }
                            else if (reg_type == KGE_REG_N3) { rs += v * v * v; Gr.x[q][c] += c3 * v * v; }
                            else { const float a = fabsf(v); rs += a * a * a; Gr.x[q][c] += c3 * v * a; }
                        }
                    }
                    acc += lmbda * inv_n * gsum<G>(rs);
                }
                int rel_off = 0;
#pragma unroll
                for (int q = 0; q < NR; ++q) {
                    const int sel = role_sel(M, q);
                    if (LDSREL && sel == 1 && role_trainable(M, q)) {
                        const int d = role_dim<M>(m, q);
                        float* dst = s_rel + id[1] * span + rel_off;
#pragma unroll
                        for (int c = 0; c < NCH; ++c) {
                            const int e = c * G + gl;
                            if (e < d) atomicAdd(dst + e, Gr.x[q][c]);  // ds_add_f32
                        }
                        rel_off += d;
                    } else if (id[sel] == ida[sel]) {
#pragma unroll
                        for (int c = 0; c < NCH; ++c) A.x[q][c] += Gr.x[q][c];
                    } else if (role_trainable(M, q)) {
                        const int d = role_dim<M>(m, q);
                        if constexpr (STAGED) {   // a negative's corrupted entity (sampled bundles: row k > 0 of bundle b)
                            const int64_t pair = b * (bundle - 1) + (i - i0 - 1);
                            constexpr int NRc = role_count(M);
                            store_row<G, NCH>(sink.stage + (nb * NRc + pair * sink.nd + role_rank_in_sel(M, q)) * sink.stride, Gr.x[q], d, gl);
                            if (role_rank_in_sel(M, q) == 0 && gl == 0) stage_register(sink, (int)id[sel], (int)pair);
                        } else {
                            atomic_add_row<G, NCH>(m.grad[role_tab(M, q)] + id[sel] * (int64_t)d, Gr.x[q], d, gl);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NR; ++q) flush_role(q);
    }
    if constexpr (LDSREL) {
        __syncthreads();
        int rel_off = 0;
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            if (role_sel(M, q) != 1 || !role_trainable(M, q)) continue;
            const int d = role_dim<M>(m, q);
            float* gt = m.grad[role_tab(M, q)];
            for (int64_t i = threadIdx.x; i < tot_relation * d; i += kBlock) {
                const int64_t rel = i / d;
                const int e = (int)(i - rel * d);
                const float v = s_rel[rel * span + rel_off + e];
                if (v != 0.f) unsafeAtomicAdd(gt + rel * d + e, v);
            }
            rel_off += d;
        }
    }
    block_accumulate_loss<G>(acc, gl, loss);
}

// ---- fused self-adversarial step (criterion.py:13-23 + trainer.py:147-157): one group owns a positive AND its
// neg_rate negatives (rows [i*neg_rate, (i+1)*neg_rate), data/generator.py:71-95).  Pass 1 scores the 1+neg_rate
// triples (lane j of the group keeps the energy of negative j), the group computes the detached softmax weights and
// the loss; pass 2 re-gathers each triple (L2 hits), back-propagates, and accumulates every gradient row whose id
// equals the positive's id for that role in REGISTERS -- a sampled negative shares its relation and one entity with
// its positive, so 1+neg_rate triples scatter ~(NR + neg_rate*(rows of one entity)) rows instead of (1+neg_rate)*NR.
template <int M, int G, int NCH>
__global__ __launch_bounds__(kBlock) void k_selfadv_bundle(DeviceModel m, const int64_t* __restrict__ ph,
                                                           const int64_t* __restrict__ pr, const int64_t* __restrict__ pt,
                                                           const int64_t* __restrict__ nh, const int64_t* __restrict__ nr,
                                                           const int64_t* __restrict__ nt, int64_t n_pos, int neg_rate,
                                                           float alpha, float* __restrict__ loss) {
    constexpr int GPB = kBlock / G;
    constexpr int NR = role_count(M);
    const int gl = threadIdx.x % G;
    const float inv_b = 1.0f / (float)n_pos;
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * GPB + threadIdx.x / G; i < n_pos; i += (int64_t)gridDim.x * GPB) {
        const int64_t idp[3] = {ph[i], pr[i], pt[i]};
        // ---- pass 1: energies
        float s_mine = 0.f;  // lane j: energy of negative j
        float s_pos;
        {
            Rows<M, NCH> R;
            Saved<M, NCH> sv;
            load_rows<M, G, NCH>(R, m, idp, gl);
            s_pos = model_fwd<M, G, NCH>(R, m, sv);
            for (int j = 0; j < neg_rate; ++j) {
                const int64_t q = i * neg_rate + j;
                const int64_t idn[3] = {nh[q], nr[q], nt[q]};
                load_rows<M, G, NCH>(R, m, idn, gl);
                const float sj = model_fwd<M, G, NCH>(R, m, sv);
                if (gl == j) s_mine = sj;
            }
        }
        // ---- loss and coefficients: n_j = -s_j, w = softmax(alpha n), L_i = -sum w logsig(-n) - logsig(-s_pos)
        const bool live = gl < neg_rate;
        const float nj = -s_mine;
        float mx = live ? nj * alpha : -INFINITY;
#pragma unroll
        for (int o = G / 2; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));  // max: rare, keep the shuffle form
        const float ex = live ? expf(nj * alpha - mx) : 0.f;
        const float den = gsum<G>(ex);
        const float wj = ex / den;
        const float term = gsum<G>(live ? wj * logsigmoid_t(-nj) : 0.f);
        acc += (-term - logsigmoid_t(-s_pos)) * inv_b;
        const float c_mine = live ? -(wj * sigmoid_t(nj)) * inv_b : 0.f;  // dL/d s_j
        const float c_pos = sigmoid_t(s_pos) * inv_b;                      // dL/d s_pos
        // ---- pass 2: backward with anchor accumulation
        Rows<M, NCH> A;  // gradient rows of the positive's ids
        {
            Rows<M, NCH> R;
            Saved<M, NCH> sv;
            load_rows<M, G, NCH>(R, m, idp, gl);
            model_fwd<M, G, NCH>(R, m, sv);
            model_bwd<M, G, NCH>(R, m, sv, c_pos, A);
        }
        for (int j = 0; j < neg_rate; ++j) {
            const float cj = __shfl(c_mine, (threadIdx.x / G) * G % 64 + j, 64);
            if (cj == 0.f) continue;
            const int64_t q = i * neg_rate + j;
            const int64_t idn[3] = {nh[q], nr[q], nt[q]};
            Rows<M, NCH> R, Gn;
            Saved<M, NCH> sv;
            load_rows<M, G, NCH>(R, m, idn, gl);
            model_fwd<M, G, NCH>(R, m, sv);
            model_bwd<M, G, NCH>(R, m, sv, cj, Gn);
#pragma unroll
            for (int q2 = 0; q2 < NR; ++q2) {
                const int sel = role_sel(M, q2);
                if (idn[sel] == idp[sel]) {
#pragma unroll
                    for (int c = 0; c < NCH; ++c) A.x[q2][c] += Gn.x[q2][c];
                } else if (role_trainable(M, q2)) {
                    const int d = role_dim<M>(m, q2);
                    atomic_add_row<G, NCH>(m.grad[role_tab(M, q2)] + idn[sel] * (int64_t)d, Gn.x[q2], d, gl);
                }
            }
        }
        scatter_rows<M, G, NCH>(A, m, idp, gl);
    }
    block_accumulate_loss<G>(acc, gl, loss);
}

// ------------------------------------------------------------------ dispatch
template <int M, int G, int NCH>
struct Launch {
    static int grid(int64_t n) {
        int64_t b = (n + (kBlock / G) - 1) / (kBlock / G);
        return (int)(b < 1 ? 1 : (b > kMaxBlocks ? kMaxBlocks : b));
    }
};

#define KGE_FOR_GEOMETRY(MID, G_, NCH_, BODY)                   \
    if (geo.G == G_ && geo.NCH == NCH_) {                       \
        constexpr int M = MID;                                  \
        constexpr int G = G_; constexpr int NCH = NCH_;         \
        BODY;                                                   \
        return check_launch(#MID);                              \
    }
#define KGE_FOR_MODEL(MID, BODY)                                \
    case MID: {                                                 \
        KGE_FOR_GEOMETRY(MID, 32, 1, BODY)                      \
        KGE_FOR_GEOMETRY(MID, 32, 2, BODY)                      \
        KGE_FOR_GEOMETRY(MID, 32, 4, BODY)                      \
        KGE_FOR_GEOMETRY(MID, 32, 8, BODY)                      \
        KGE_FOR_GEOMETRY(MID, 64, 8, BODY)                      \
        KGE_FOR_GEOMETRY(MID, 64, 16, BODY)                     \
        KGE_FOR_GEOMETRY(MID, 64, 32, BODY)                     \
        break;                                                  \
    }

}  // namespace kge
