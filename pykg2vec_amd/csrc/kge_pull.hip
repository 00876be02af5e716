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
//     data/generator.py:23-35) and as corrupting entity (per step: the sampler registers each draw in the drawn entity's
//     bucket, consumed in pair order; the sampler of step k+1 rides in the launch of step k),
//   * RE-computes each incident pair's forward (three L2-resident row gathers + pre-computed row norms: one butterfly
//     reduction per incidence) and keeps only the gradient of ITS OWN row, summed in registers in a fixed order,
//   * applies the normalisation backward once, then the dense optimiser update of that row (torch.optim semantics, every
//     row every step), writes the new row into the other half of a double-buffered table, and leaves the new row norm
//     for the next step.
// Each pair is evaluated ~3.25 times instead of once (cheap: VALU + L2 reads), nothing is scattered, the optimiser pass
// and the gradient zeroing pass disappear.  Rows with long lists (frequent relations, hub entities) are cut into
// segments of at most PullIndex.SEGMENT (host) incidences: each segment's owner writes a partial sum, a small second kernel adds the
// partials in segment order and finishes the row -- still deterministic.
#include "kge_row_kernels.h"
#include "kge_opt_device.h"
#include "kge_sampler_device.h"
#include "kge_pull_device.h"
#define KGE_TS_UNIT pull
#include "kge_ts_debug.h"   // (slots: 0 = k_pull_step, 1 = k_pull_eval; no-ops in the product build)
#include <stdlib.h>
#include <type_traits>

namespace kge {

struct PullArgs {
    const float* tab_in[2];    // entity / relation table read by this step
    float* tab_out[2];         // the tables the step writes (other half of the double buffer)
    const float* hat_in[2];    // row-normalised copies x / max(||x||, eps) of tab_in: what the other owners gather.  Rows
                               // are padded with zeros to 4 * G * NV floats, so a gather is one unconditional 16-byte load
    float* hat_out[2];         // the same for tab_out, written by each row's owner
    unsigned hat_row_bytes;    // bytes between consecutive hat rows: 16 * G * NV (padded: every lane loads) or 4 * d (compact, round 6:
                               // lanes beyond the row skip their load; 6.5 instead of 8.3 MB of normalised rows for the L2s to hold at C1)
    const float* norm_in;      // [E + R] L2 norms of the rows of tab_in (entities first)
    float* norm_out;           // norms of the rows of tab_out
    float* s1[2];              // optimiser state, same row layout as the tables (NULL where the optimiser has none)
    float* s2[2];
    const int4* pairs;         // this batch: (h, r, t, -)
    PullLists lists;           // this batch's sampler output
    const int4* items;         // work items: (row g, first incidence, end incidence, kind | slot << 2)
    const int32_t* inc;        // static incidences of the batch sorted by (row, pair, role): pair << 2 | role
    float* partials;           // [slots][4 * G * NV] partial gradient sums of multi-segment rows
    const int4* multi;         // rows with several segments: (row g, first slot, number of slots, -)
    int64_t n_items, n_multi;
    const uint32_t* dense_skip;   // optional bitmap: rows with explicit items; all other rows are visited implicitly after them
    int n_rows;                   // E + R
    int sample_blocks;         // blocks [0, sample_blocks) sample the NEXT batch; the others own work items
    int E, d, l1, reset_lists;
    float margin;
    OptArgs opt;
    const float* dev_hyper;    // optional device-resident {lr, step_size, bc2_sqrt}
    const float* theta;        // TransM (pairwise.py:341-347): fixed per-relation weight of both energies; NULL = TransE
    // two-phase ("staged direction") form: k_pull_eval leaves one record per pair, the owners of k_pull_step<..., DIR> sum them
    float4* recs;              // [n_pairs] (coef * theta, energy(+), energy(-), tail as 0 / 1)
    void* codes;               // [n_pairs][2][G * NV] one byte per lane = the signs of its four residual elements (2 bits each)
    int64_t n_pairs;
};

#define KGE_F4_EACH(expr_x, expr_y, expr_z, expr_w) expr_x; expr_y; expr_z; expr_w;

// a * b + c on 24-bit signed operands as ONE VALU instruction (the compiler turns __mul24(a, b) + c into v_mul_i32_i24 + v_add3_u32)
__device__ __forceinline__ int mad_i24(int a, int b, int c) {
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// gradient wrt the NORMALISED own row, summed over incidences -> normalisation backward -> optimiser -> new row, its
// normalised copy and its norm
template <int OPT, int G, int NV, bool PRE = false>
__device__ __forceinline__ void pull_finish_row(const PullArgs& a, int g, const float4 (&X)[NV], float nX, const float4 (&gs)[NV],
                                                int gl, const float4* pre1 = nullptr, const float4* pre2 = nullptr) {
    const int nvec = a.d >> 2;
    const bool is_rel = g >= a.E;
    // (pointer selects, not a[tb]: a runtime index into a kernel-argument array would put the array in scratch)
    float* const t_out = is_rel ? a.tab_out[1] : a.tab_out[0];
    float* const h_out = is_rel ? a.hat_out[1] : a.hat_out[0];
    float* const st1 = is_rel ? a.s1[1] : a.s1[0];
    float* const st2 = is_rel ? a.s2[1] : a.s2[0];
    const int64_t off = (int64_t)(is_rel ? g - a.E : g) * a.d;
    const bool fX = nX > kEpsNormalize;
    const float iX = 1.0f / fmaxf(nX, kEpsNormalize);
    float dX = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        dX = fmaf(X[v].x, gs[v].x, dX); dX = fmaf(X[v].y, gs[v].y, dX);
        dX = fmaf(X[v].z, gs[v].z, dX); dX = fmaf(X[v].w, gs[v].w, dX);
    }
    dX = gsum<G>(dX) * iX;
    if constexpr (OPT == KGE_OPT_GRADIENT) {   // data-parallel ranks: the dense gradient row itself (reduced across ranks later)
        float4 Gw[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            Gw[v].x = fX ? (gs[v].x - (X[v].x * iX) * dX) * iX : gs[v].x * iX;
            Gw[v].y = fX ? (gs[v].y - (X[v].y * iX) * dX) * iX : gs[v].y * iX;
            Gw[v].z = fX ? (gs[v].z - (X[v].z * iX) * dX) * iX : gs[v].z * iX;
            Gw[v].w = fX ? (gs[v].w - (X[v].w * iX) * dX) * iX : gs[v].w * iX;
        }
        store_row4<G, NV>(t_out + off, Gw, nvec, gl);
        return;
    }
    OptArgs o = a.opt;
    if (a.dev_hyper) { o.lr = a.dev_hyper[0]; o.step_size = a.dev_hyper[1]; o.bc2_sqrt = a.dev_hyper[2]; }
    float4 P[NV], M1[NV], M2[NV];
    if constexpr (PRE) {   // (two-phase form: the state rows were requested together with the row itself, before the visits)
#pragma unroll
        for (int v = 0; v < NV; ++v) { M1[v] = pre1[v]; M2[v] = pre2[v]; }
    } else {
        if constexpr (OPT != KGE_OPT_SGD && OPT != KGE_OPT_GRADIENT) load_row4<G, NV>(M1, st1 + off, nvec, gl);
        if constexpr (OPT == KGE_OPT_ADAM) load_row4<G, NV>(M2, st2 + off, nvec, gl);
    }
    float n2 = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        P[v] = X[v];
#define KGE_UPD(c)                                                                                           \
        {                                                                                                    \
            const float graw = fX ? (gs[v].c - (X[v].c * iX) * dX) * iX : gs[v].c * iX;                      \
            float m1 = 0.f, m2 = 0.f;                                                                        \
            if constexpr (OPT != KGE_OPT_SGD) m1 = M1[v].c;                                                  \
            if constexpr (OPT == KGE_OPT_ADAM) m2 = M2[v].c;                                                 \
            opt_update<OPT>(P[v].c, graw, m1, m2, o);                                                        \
            if constexpr (OPT != KGE_OPT_SGD) M1[v].c = m1;                                                  \
            if constexpr (OPT == KGE_OPT_ADAM) M2[v].c = m2;                                                 \
            n2 = fmaf(P[v].c, P[v].c, n2);                                                                   \
        }
        KGE_UPD(x) KGE_UPD(y) KGE_UPD(z) KGE_UPD(w)
#undef KGE_UPD
    }
    // (lanes beyond the row hold zeros: X = 0, gs = 0, state loaded as 0 -> every optimiser leaves p = 0, n2 unchanged)
    n2 = gsum<G>(n2);
    const float nn = sqrtf(n2);
    const float inn = 1.0f / fmaxf(nn, kEpsNormalize);
    store_row4<G, NV>(t_out + off, P, nvec, gl);
    if constexpr (OPT != KGE_OPT_SGD) store_row4<G, NV>(st1 + off, M1, nvec, gl);
    if constexpr (OPT == KGE_OPT_ADAM) store_row4<G, NV>(st2 + off, M2, nvec, gl);
    float4* const hrow = reinterpret_cast<float4*>(reinterpret_cast<char*>(h_out) + (int64_t)(is_rel ? g - a.E : g) * a.hat_row_bytes);
    const int hvec = (int)(a.hat_row_bytes >> 4);   // float4 slots of a hat row: G * NV (padded: lanes beyond the row store their zeros) or d / 4
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        P[v].x *= inn; P[v].y *= inn; P[v].z *= inn; P[v].w *= inn;
        if (v * G + gl < hvec) hrow[v * G + gl] = P[v];
    }
    if (gl == 0) a.norm_out[g] = nn;
}

// the four NORMALISED rows of one incident pair as gathered for one owner
template <int NV>
struct PullRows {
    float4 hh[NV], rr[NV], tt[NV], cc[NV];
    int w;   // corrupting entity | tail << 24 | role << 25
    float th;   // TransM weight of the pair's relation (1 for TransE)
};

// ---- phase 1 of the two-phase form: every pair of the batch is evaluated ONCE by one lane group -- the same four gathers, the
// same arithmetic in the same order as a visit of k_pull_step -- and leaves a record: the hinge coefficient, and the signed
// direction of both residuals (two bits per element).  L1 only (see launch_pull_step).  No sampling here: the draw was
// registered by the sampler riding in the previous step's launch.
// kEvalPP pairs per lane group, their descriptors and then their four rows requested together (independent chains): half the
// workgroups -- 2 048 at B = 32 768, ONE residency round of 8 x 256 workgroups instead of two -- and half the workgroup launches
// (per-workgroup timestamps: the dispatcher needs 1.8 us to start 2 048 workgroups; profiles/r04_experiments.md section 7).
constexpr int kEvalPP = 2;
constexpr bool kHatCompactDefault = true;   // hat rows: padded to 4 * G * NV floats (rounds 2-5) or compact (KGE_HAT_COMPACT=1)
template <bool L1, int G, int NV>
__global__ __launch_bounds__(kBlock) void k_pull_eval(PullArgs a, float* __restrict__ loss) {
    constexpr int GPB = kBlock / G, PP = kEvalPP;
    KGE_TS_BEGIN(1)
    const int gl = threadIdx.x % G;
    float acc = 0.f;
    const char* __restrict__ hat_e = reinterpret_cast<const char*>(a.hat_in[0]);
    const char* __restrict__ hat_r = reinterpret_cast<const char*>(a.hat_in[1]);
    const unsigned kRowBytes = a.hat_row_bytes;
    const unsigned lane_off = 16u * gl;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    int64_t idx[PP];
    bool on[PP];
    int4 p[PP];
    int w[PP];
#pragma unroll
    for (int q = 0; q < PP; ++q) {   // (clamped: the loads stay unconditional, a dead pair is masked at its stores)
        idx[q] = ((int64_t)blockIdx.x * PP + q) * GPB + threadIdx.x / G;
        on[q] = idx[q] < a.n_pairs;
        const int64_t ic = on[q] ? idx[q] : a.n_pairs - 1;
        p[q] = a.pairs[ic];
        w[q] = a.lists.pc[ic];
    }
    float4 hh[PP][NV], rr[PP][NV], tt[PP][NV], cc[PP][NV];
    float th[PP];
    unsigned oh[PP], orr[PP], ot[PP], oc[PP];
#pragma unroll
    for (int q = 0; q < PP; ++q) {
        oh[q] = (unsigned)p[q].x * kRowBytes + lane_off; orr[q] = (unsigned)p[q].y * kRowBytes + lane_off;
        ot[q] = (unsigned)p[q].z * kRowBytes + lane_off; oc[q] = (unsigned)(w[q] & 0xFFFFFF) * kRowBytes + lane_off;
        th[q] = a.theta ? a.theta[p[q].y] : 1.0f;
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {   // ONE predicated region per chunk for all PP x 4 row gathers (padded rows: every lane is inside)
#pragma unroll
        for (int q = 0; q < PP; ++q) { hh[q][v] = zero4; rr[q][v] = zero4; tt[q][v] = zero4; cc[q][v] = zero4; }
        if (lane_off + 16u * G * v < kRowBytes) {
#pragma unroll
            for (int q = 0; q < PP; ++q) {
                hh[q][v] = *reinterpret_cast<const float4*>(hat_e + (oh[q] + 16u * G * v));
                rr[q][v] = *reinterpret_cast<const float4*>(hat_r + (orr[q] + 16u * G * v));
                tt[q][v] = *reinterpret_cast<const float4*>(hat_e + (ot[q] + 16u * G * v));
                cc[q][v] = *reinterpret_cast<const float4*>(hat_e + (oc[q] + 16u * G * v));
            }
        }
    }
#pragma unroll
    for (int q = 0; q < PP; ++q) {
        const int64_t i = idx[q];
        const bool tail = ((w[q] >> 24) & 1) != 0;
        float4 up[NV], un[NV];
        float sp = 0.f, sn = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
#define KGE_FWD(c)                                                                                            \
            up[v].c = hh[q][v].c + rr[q][v].c - tt[q][v].c;                                                   \
            un[v].c = (tail ? hh[q][v].c : cc[q][v].c) + rr[q][v].c - (tail ? cc[q][v].c : tt[q][v].c);       \
            sp = L1 ? sp + fabsf(up[v].c) : fmaf(up[v].c, up[v].c, sp);                                       \
            sn = L1 ? sn + fabsf(un[v].c) : fmaf(un[v].c, un[v].c, sn);
            KGE_FWD(x) KGE_FWD(y) KGE_FWD(z) KGE_FWD(w)
#undef KGE_FWD
        }
        gsum2<G>(sp, sn);
        if constexpr (!L1) { sp = sqrtf(sp); sn = sqrtf(sn); }
        const float vv = th[q] * sp + a.margin - th[q] * sn;
        if (!on[q]) continue;   // (group-uniform)
        acc += fmaxf(vv, 0.f);
        const float coef = (vv > 0.f ? 1.f : (vv == 0.f ? 0.5f : 0.f)) * th[q];
        if (gl == 0) a.recs[i] = make_float4(tail ? -coef : coef, sp, sn, 0.f);   // (coef >= 0: its sign carries `tail`)
        if (coef != 0.f) {
            if constexpr (L1) {
                unsigned char* cp = reinterpret_cast<unsigned char*>(a.codes) + i * (int64_t)(2 * G * NV);
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    // the sign k_pull_step uses: clamp(x * 2^100, -1, 1); code 0 = zero, 1 = positive, 3 = negative: the two bits ARE the
                    // sign as a two's-complement number (the owner decodes with one v_bfe_i32 + one v_cvt_f32_i32 per element)
                    unsigned bp = 0, bn = 0;
#define KGE_CODE(c, sh)                                                                                       \
                    { const float s1 = __builtin_amdgcn_fmed3f(up[v].c * 0x1p100f, -1.f, 1.f);                 \
                      const float s2 = __builtin_amdgcn_fmed3f(un[v].c * 0x1p100f, -1.f, 1.f);                 \
                      bp |= (s1 > 0.f ? 1u : (s1 < 0.f ? 3u : 0u)) << sh;                                      \
                      bn |= (s2 > 0.f ? 1u : (s2 < 0.f ? 3u : 0u)) << sh; }
                    KGE_CODE(x, 0) KGE_CODE(y, 2) KGE_CODE(z, 4) KGE_CODE(w, 6)
#undef KGE_CODE
                    cp[v * G + gl] = (unsigned char)bp;
                    cp[G * NV + v * G + gl] = (unsigned char)bn;
                }
            }
        }
    }
    block_accumulate_loss<G>(acc, gl, loss);
    KGE_TS_END(1, 1)
}

template <int OPT, bool L1, int G, int NV, bool DIR = false>
__global__ __launch_bounds__(kBlock) void k_pull_step(PullArgs a, PullSampleArgs sa, float* __restrict__ loss) {
    static_assert(!DIR || L1, "the two-phase form exists for L1 only");
    constexpr int GPB = kBlock / G;
    KGE_TS_BEGIN(0)
    if ((int)blockIdx.x < a.sample_blocks) {   // leading blocks: the sampler of the NEXT batch rides along (writes the other list set)
        const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
        if (i < sa.n) pull_sample_one(sa, i);
        KGE_TS_END(0, 2)
        return;
    }
    const int gl = threadIdx.x % G;
    const int gbase = (threadIdx.x & 63) / G * G;   // first lane of this group inside its wave
    const int d = a.d, nvec = a.d >> 2;
    // partial sums of rows cut into 2..GPB items: their owners sit in consecutive groups of this workgroup
    __shared__ float4 s_part[GPB][NV * G];
    // visit descriptors (h, r, t, c | tail << 24 | role << 25) of each owner group, in visit order: one broadcast
    // ds_read_b128 per visit instead of four cross-lane shuffles
    __shared__ int4 s_desc[GPB][G];
    float acc = 0.f;
    const int64_t item = (int64_t)((int)blockIdx.x - a.sample_blocks) * GPB + threadIdx.x / G;
    int4 it = make_int4(-1, 0, 0, 0);
    if (item < a.n_items) it = a.items[item];
    else if (a.dense_skip != nullptr && item - a.n_items < a.n_rows) {   // implicit part: a row without static incidences
        const int row = (int)(item - a.n_items);
        if (!((a.dense_skip[row >> 5] >> (row & 31)) & 1u)) it = make_int4(row, 0, 0, 0);
    }
    const int g = it.x;
    const int kind = it.w & 3;
    float4 X[NV], gs[NV];
    float4 S1[NV], S2[NV];     // (two-phase form only: prefetched optimiser state)
#pragma unroll
    for (int v = 0; v < NV; ++v) { S1[v] = make_float4(0.f, 0.f, 0.f, 0.f); S2[v] = S1[v]; }
    float nX = 0.f;
    if (g >= 0) {
        const bool is_rel = g >= a.E;
        // the owner's row and norm depend on nothing but the item: requested first
        load_row4<G, NV>(X, (is_rel ? a.tab_in[1] : a.tab_in[0]) + (int64_t)(is_rel ? g - a.E : g) * d, nvec, gl);
        nX = a.norm_in[g];
        if constexpr (DIR && OPT != KGE_OPT_GRADIENT) {
            // the visits of the two-phase form need few registers: the optimiser state of a row this item will finish is requested
            // now instead of after the last visit (one dependent round trip less at the end of every owner)
            if (kind == 0 || (kind == 3 && ((it.w >> 2) & 15) == 0)) {
                const int64_t off0 = (int64_t)(is_rel ? g - a.E : g) * d;
                if constexpr (OPT != KGE_OPT_SGD) load_row4<G, NV>(S1, (is_rel ? a.s1[1] : a.s1[0]) + off0, nvec, gl);
                if constexpr (OPT == KGE_OPT_ADAM) load_row4<G, NV>(S2, (is_rel ? a.s2[1] : a.s2[0]) + off0, nvec, gl);
            }
        }
        int cnt = 0;
        bool fast_c = true;
        // the row's corrupting-entity draws are walked by its first (or only) item
        const bool walks_c = !is_rel && (kind == 0 || kind == 1 || (kind == 3 && ((it.w >> 2) & 15) == 0));
        int nvis;
        if constexpr (DIR) nvis = own_visit_list_dir<G>(a.lists, a.inc, it, g, walks_c, gl, gbase, reinterpret_cast<int*>(s_desc[threadIdx.x / G]), &cnt, &fast_c,
                                                        /* ranked = */ !(L1 && a.theta == nullptr));
        else nvis = own_visit_list<G>(a.lists, it, g, walks_c, gl, gbase, s_desc[threadIdx.x / G], &cnt, &fast_c);
#pragma unroll
        for (int v = 0; v < NV; ++v) gs[v] = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (DIR) {
            // two-phase form: a visit is the pair's record + this lane's share of its two direction codes -- no row gathers,
            // no reductions.  Same coefficients, same fused multiply-adds in the same order as the one-phase visit below.
            const int* __restrict__ vis = reinterpret_cast<const int*>(s_desc[threadIdx.x / G]);
            using RecT = float;                                 // a visit needs only the signed coefficient of the pair's record
            auto rec_x = [](const RecT& r) { return r; };
            // The owner kernel is VALU-bound (SQ_ACTIVE_INST_VALU: 0.75 of its SIMD cycles, ~1 000 VALU instructions per wave, most of
            // them in the visits), so a visit is kept short: the own row's coefficients in up / un come out of two 16-bit tables of
            // 2-bit two's-complement entries indexed by (role, tail), the direction codes decode with one v_bfe_i32 each, and TransE
            // (theta == NULL: every coefficient is 1 or 1/2) sums in INTEGER half units -- v_mad_i32_i24 instead of v_cvt + v_fma;
            // those float sums were exact, so the result is bit-identical (profiles/r04_experiments.md section 7).
            //   su: H +1, T -1, R +1, C 0;   sv: H (tail ? -1 : 0), T (tail ? 0 : +1), R -1, C (tail ? +1 : -1);   index 2 role + tail
            const bool unit = L1 && a.theta == nullptr;   // kernel-uniform
            int4 gi[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) gi[v] = make_int4(0, 0, 0, 0);
            auto visit_dir = [&](int e, const RecT rec, const auto& cpv, const auto& cnv) {
                const int role = e & 3;
                const unsigned rbits = __float_as_uint(rec_x(rec));
                const float coef = fabsf(rec_x(rec));
                if (coef == 0.f) return;
                const unsigned sh2 = (unsigned)(((role << 1) | (int)(rbits >> 31)) << 1);
                const int lu = __builtin_amdgcn_sbfe(0x05F5, sh2, 2u), lv = __builtin_amdgcn_sbfe(0x7F1C, sh2, 2u);
                float su = (float)lu * coef, sv = (float)lv * coef;
                if constexpr (L1) {
                    if (unit) {
                        const int c2 = coef == 1.f ? 2 : 1;
                        const int iu = __mul24(lu, c2), iv = __mul24(lv, c2);
#pragma unroll
                        for (int v2 = 0; v2 < NV; ++v2) {
                            const unsigned bp = cpv[v2], bn = cnv[v2];
#define KGE_DECI(b, sh) (((int)((unsigned)(b) << (30 - (sh)))) >> 30)
                            gi[v2].x = mad_i24(iv, KGE_DECI(bn, 0), mad_i24(iu, KGE_DECI(bp, 0), gi[v2].x));
                            gi[v2].y = mad_i24(iv, KGE_DECI(bn, 2), mad_i24(iu, KGE_DECI(bp, 2), gi[v2].y));
                            gi[v2].z = mad_i24(iv, KGE_DECI(bn, 4), mad_i24(iu, KGE_DECI(bp, 4), gi[v2].z));
                            gi[v2].w = mad_i24(iv, KGE_DECI(bn, 6), mad_i24(iu, KGE_DECI(bp, 6), gi[v2].w));
#undef KGE_DECI
                        }
                        return;
                    }
#pragma unroll
                    for (int v2 = 0; v2 < NV; ++v2) {
                        const unsigned bp = cpv[v2], bn = cnv[v2];
#define KGE_DEC(b, sh) ((float)(((int)((b) << (30 - (sh)))) >> 30))   /* 2-bit two's complement: 01 -> +1, 11 -> -1, 00 -> 0 */
                        gs[v2].x = fmaf(sv, KGE_DEC(bn, 0), fmaf(su, KGE_DEC(bp, 0), gs[v2].x));
                        gs[v2].y = fmaf(sv, KGE_DEC(bn, 2), fmaf(su, KGE_DEC(bp, 2), gs[v2].y));
                        gs[v2].z = fmaf(sv, KGE_DEC(bn, 4), fmaf(su, KGE_DEC(bp, 4), gs[v2].z));
                        gs[v2].w = fmaf(sv, KGE_DEC(bn, 6), fmaf(su, KGE_DEC(bp, 6), gs[v2].w));
#undef KGE_DEC
                    }
                }
            };
            using CodeT = unsigned;
            auto load_codes = [&](int pair, CodeT (&cpv)[NV], CodeT (&cnv)[NV]) {
                if constexpr (L1) {
                    // (32-bit byte offsets from the uniform base: one shift-add per visit instead of 64-bit address arithmetic per load;
                    //  n_pairs * 2 G NV bytes < 4 GiB is checked where the buffers are sized)
                    const unsigned char* __restrict__ cbase = reinterpret_cast<const unsigned char*>(a.codes);
                    const unsigned off = (unsigned)pair * (unsigned)(2 * G * NV) + (unsigned)gl;
#pragma unroll
                    for (int v2 = 0; v2 < NV; ++v2) { cpv[v2] = cbase[off + (unsigned)(v2 * G)]; cnv[v2] = cbase[off + (unsigned)(G * NV + v2 * G)]; }
                }
            };
#ifndef KGE_DIR_BATCH
#define KGE_DIR_BATCH 4
#endif
            constexpr int kDirBatch = KGE_DIR_BATCH;     // visits whose records and codes are requested before the first is summed
#ifdef KGE_DIR_NOVISIT   /* timing experiment only: how long is phase 2 without its visits? */
            nvis = 0;
#endif
            for (int v0 = 0; v0 < nvis; v0 += kDirBatch) {
                int e[kDirBatch];
                RecT rec[kDirBatch];
                CodeT cpv[kDirBatch][NV], cnv[kDirBatch][NV];
#pragma unroll
                for (int q = 0; q < kDirBatch; ++q) {
                    e[q] = v0 + q < nvis ? vis[v0 + q] : -1;
                    const int pair = e[q] >= 0 ? (e[q] >> 2) : 0;
                    rec[q] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.recs) + (unsigned)pair * 16u);
                    load_codes(pair, cpv[q], cnv[q]);
                }
#pragma unroll
                for (int q = 0; q < kDirBatch; ++q)
                    if (e[q] >= 0) visit_dir(e[q], rec[q], cpv[q], cnv[q]);
            }
            if (cnt > 0 && !fast_c) {   // more drawers than the bucket / lane group holds: pair-ordered walk over bucket + chain
                const int nb = cnt < kPullCap ? cnt : kPullCap;
                int last = -1;
                for (;;) {
                    int best = 0x7FFFFFFF;
                    for (int m = 0; m < nb; ++m) { const int j = a.lists.bucket[(int64_t)g * kPullCap + m]; if (j > last && j < best) best = j; }
                    for (int j = a.lists.head[g]; j >= 0; j = a.lists.next[j]) if (j > last && j < best) best = j;
                    if (best == 0x7FFFFFFF) break;
                    CodeT c1[NV], c2[NV];
                    load_codes(best, c1, c2);
                    RecT rb;
                    rb = reinterpret_cast<const float*>(a.recs)[4 * (int64_t)best];
                    visit_dir((best << 2) | kRoleC, rb, c1, c2);
                    last = best;
                }
            }
            if (unit) {   // integer half units -> float (exact)
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    gs[v].x = (float)gi[v].x * 0.5f; gs[v].y = (float)gi[v].y * 0.5f; gs[v].z = (float)gi[v].z * 0.5f; gs[v].w = (float)gi[v].w * 0.5f;
                }
            }
        } else {

        // gather the four normalised rows of one incident pair (the owner's own row among them: it is L1 / L2 hot, and
        // loading it like the others keeps the gather branch-free and the arithmetic identical for all four owners)
        // (32-bit byte offsets from the uniform table bases: one shift-or per row instead of 64-bit address arithmetic)
        const char* __restrict__ hat_e = reinterpret_cast<const char*>(a.hat_in[0]);
        const char* __restrict__ hat_r = reinterpret_cast<const char*>(a.hat_in[1]);
        const unsigned kRowBytes = a.hat_row_bytes;
        const unsigned lane_off = 16u * gl;
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        auto fetch = [&](int h, int r, int t, int w, PullRows<NV>& b) {
            b.w = w;
            b.th = a.theta ? a.theta[r] : 1.0f;
            const unsigned oh = (unsigned)h * kRowBytes + lane_off, orr = (unsigned)r * kRowBytes + lane_off;
            const unsigned ot = (unsigned)t * kRowBytes + lane_off, oc = (unsigned)(w & 0xFFFFFF) * kRowBytes + lane_off;
#pragma unroll
            for (int v = 0; v < NV; ++v) {   // (padded rows: every lane is inside)
                b.hh[v] = zero4; b.rr[v] = zero4; b.tt[v] = zero4; b.cc[v] = zero4;
                if (lane_off + 16u * G * v < kRowBytes) {
                    b.hh[v] = *reinterpret_cast<const float4*>(hat_e + (oh + 16u * G * v));
                    b.rr[v] = *reinterpret_cast<const float4*>(hat_r + (orr + 16u * G * v));
                    b.tt[v] = *reinterpret_cast<const float4*>(hat_e + (ot + 16u * G * v));
                    b.cc[v] = *reinterpret_cast<const float4*>(hat_e + (oc + 16u * G * v));
                }
            }
        };
        auto fetch_visit = [&](int v, PullRows<NV>& b) {
            const int4 ds = s_desc[threadIdx.x / G][v];   // same address for the whole group: a broadcast read
            fetch(ds.x, ds.y, ds.z, ds.w, b);
        };
        // forward of the pair (identical arithmetic whichever of its four rows the owner holds), hinge, and the owner's
        // share of the backward: gradient wrt its NORMALISED row
        auto compute = [&](const PullRows<NV>& b) {
            const int role = (b.w >> 25) & 3;
            const bool tail = ((b.w >> 24) & 1) != 0;
            float4 up[NV], un[NV];
            float sp = 0.f, sn = 0.f;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
#define KGE_FWD(c)                                                                                            \
                up[v].c = b.hh[v].c + b.rr[v].c - b.tt[v].c;                                                  \
                un[v].c = (tail ? b.hh[v].c : b.cc[v].c) + b.rr[v].c - (tail ? b.cc[v].c : b.tt[v].c);        \
                sp = L1 ? sp + fabsf(up[v].c) : fmaf(up[v].c, up[v].c, sp);                                   \
                sn = L1 ? sn + fabsf(un[v].c) : fmaf(un[v].c, un[v].c, sn);
                KGE_FWD(x) KGE_FWD(y) KGE_FWD(z) KGE_FWD(w)
#undef KGE_FWD
            }
            gsum2<G>(sp, sn);
            if constexpr (!L1) { sp = sqrtf(sp); sn = sqrtf(sn); }
            const float v = b.th * sp + a.margin - b.th * sn;   // (TransE: th = 1, the products are exact)
            if (role == kRoleH) acc += fmaxf(v, 0.f);   // every pair has exactly one head incidence: the loss is counted there
            const float coef = (v > 0.f ? 1.f : (v == 0.f ? 0.5f : 0.f)) * b.th;   // torch.max splits the subgradient at equality
            if (coef == 0.f) return;
            // own row's coefficient in up / un:  H: +1 / (tail ? +1 : 0)   T: -1 / (tail ? 0 : -1)   R: +1 / +1   C: 0 / (tail ? -1 : +1);
            // d loss / d up = +coef * g(up), d loss / d un = -coef * g(un): the signs are folded into su / sv
            float su = role == kRoleT ? -coef : (role == kRoleC ? 0.f : coef);
            float sv = role == kRoleR ? -coef : (role == kRoleH ? (tail ? -coef : 0.f) : role == kRoleT ? (tail ? 0.f : coef) : (tail ? coef : -coef));
            if constexpr (L1) {
#pragma unroll
                for (int v2 = 0; v2 < NV; ++v2) {
                    // sign(x) as clamp(x * 2^100, -1, 1): +-1 for every |x| >= 2^-100, 0 for 0 (sums of normalised components
                    // are either exactly 0 -- padding lanes -- or ~1e-2..1): two plain VALU ops, no VOPC -> SGPR round trip
#define KGE_BWD(c)                                                                                            \
                    gs[v2].c = fmaf(sv, __builtin_amdgcn_fmed3f(un[v2].c * 0x1p100f, -1.f, 1.f),              \
                                    fmaf(su, __builtin_amdgcn_fmed3f(up[v2].c * 0x1p100f, -1.f, 1.f), gs[v2].c));
                    KGE_BWD(x) KGE_BWD(y) KGE_BWD(z) KGE_BWD(w)
#undef KGE_BWD
                }
            } else {
                su = sp > 0.f ? su / sp : 0.f;
                sv = sn > 0.f ? sv / sn : 0.f;
#pragma unroll
                for (int v2 = 0; v2 < NV; ++v2) {
                    gs[v2].x = fmaf(sv, un[v2].x, fmaf(su, up[v2].x, gs[v2].x)); gs[v2].y = fmaf(sv, un[v2].y, fmaf(su, up[v2].y, gs[v2].y));
                    gs[v2].z = fmaf(sv, un[v2].z, fmaf(su, up[v2].z, gs[v2].z)); gs[v2].w = fmaf(sv, un[v2].w, fmaf(su, up[v2].w, gs[v2].w));
                }
            }
        };
        // software pipeline: the gathers of the next visits are in flight while a visit is evaluated (a ring of row buffers
        // indexed at compile time, no copies)
#ifndef KGE_PULL_DEPTH
#define KGE_PULL_DEPTH 2
#endif
#if KGE_PULL_DEPTH == 2
        if (nvis > 0) {
            PullRows<NV> ba, bb;
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
#else
        if (nvis > 0) {
            constexpr int D = KGE_PULL_DEPTH;
            PullRows<NV> buf[D];
#pragma unroll
            for (int q = 0; q < D; ++q) if (q < nvis) fetch_visit(q, buf[q]);
            for (int v = 0; v < nvis; v += D) {
#pragma unroll
                for (int q = 0; q < D; ++q) {
                    if (v + q < nvis) {
                        compute(buf[q]);
                        if (v + q + D < nvis) fetch_visit(v + q + D, buf[q]);
                    }
                }
            }
        }
#endif
        if (cnt > 0 && !fast_c) {
            // more drawers than the lane group or the bucket holds (tiny entity sets only): pair-ordered selection over
            // the bucket and the overflow list, one unpipelined visit at a time
            const int nb = cnt < kPullCap ? cnt : kPullCap;
            int last = -1;
            for (;;) {
                int best = 0x7FFFFFFF;
                for (int m = 0; m < nb; ++m) { const int j = a.lists.bucket[(int64_t)g * kPullCap + m]; if (j > last && j < best) best = j; }
                for (int j = a.lists.head[g]; j >= 0; j = a.lists.next[j]) if (j > last && j < best) best = j;
                if (best == 0x7FFFFFFF) break;
                const int4 p2 = a.pairs[best];
                PullRows<NV> b;
                fetch(p2.x, p2.y, p2.z, a.lists.pc[best] | (kRoleC << 25), b);
                compute(b);
                last = best;
            }
        }
        }   // (one-phase visits)
        if (cnt > 0 && a.reset_lists && gl == 0) {
            a.lists.count[g] = 0;
            if (cnt > kPullCap) a.lists.head[g] = -1;
        }
        if (kind == 0) {
            pull_finish_row<OPT, G, NV, DIR && OPT != KGE_OPT_GRADIENT>(a, g, X, nX, gs, gl, S1, S2);
        } else if (kind == 3) {
#pragma unroll
            for (int v = 0; v < NV; ++v) s_part[threadIdx.x / G][v * G + gl] = gs[v];
        } else {
            float4* out = reinterpret_cast<float4*>(a.partials) + (int64_t)(it.w >> 2) * (G * NV);
#pragma unroll
            for (int v = 0; v < NV; ++v) out[v * G + gl] = gs[v];
        }
    }
    __syncthreads();
    if (g >= 0 && kind == 3 && ((it.w >> 2) & 15) == 0) {   // first item of a workgroup-local row: add the others' sums in segment order
        const int nseg = it.w >> 6, me = threadIdx.x / G;
        for (int m = 1; m < nseg; ++m) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const float4 p = s_part[me + m][v * G + gl];
                gs[v].x += p.x; gs[v].y += p.y; gs[v].z += p.z; gs[v].w += p.w;
            }
        }
        pull_finish_row<OPT, G, NV, DIR && OPT != KGE_OPT_GRADIENT>(a, g, X, nX, gs, gl, S1, S2);
    }
    block_accumulate_loss<G>(acc, gl, loss);
    KGE_TS_END(0, 1)
}

// rows cut into several segments: add the segments' partial sums in segment order, then finish the row
template <int OPT, int G, int NV>
__global__ __launch_bounds__(kBlock) void k_pull_finish(PullArgs a) {
    constexpr int GPB = kBlock / G;
    const int gl = threadIdx.x % G;
    const int64_t m = (int64_t)blockIdx.x * GPB + threadIdx.x / G;
    if (m >= a.n_multi) return;
    const int4 row = a.multi[m];
    const int g = row.x;
    const bool is_rel = g >= a.E;
    float4 X[NV], gs[NV];
    load_row4<G, NV>(X, (is_rel ? a.tab_in[1] : a.tab_in[0]) + (int64_t)(is_rel ? g - a.E : g) * a.d, a.d >> 2, gl);
#pragma unroll
    for (int v = 0; v < NV; ++v) gs[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < row.z; ++s) {
        const float4* in = reinterpret_cast<const float4*>(a.partials) + (int64_t)(row.y + s) * (G * NV);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float4 p = in[v * G + gl];
            gs[v].x += p.x; gs[v].y += p.y; gs[v].z += p.z; gs[v].w += p.w;
        }
    }
    pull_finish_row<OPT, G, NV>(a, g, X, a.norm_in[g], gs, gl);
}

// L2 norms and normalised copies of the rows of a table, in the lane layout / operation order pull_finish_row uses
template <int G, int NV>
__global__ __launch_bounds__(kBlock) void k_row_norms(const float* __restrict__ tab, int64_t rows, int d, float* __restrict__ out,
                                                      float* __restrict__ hat, unsigned hat_row_bytes) {
    constexpr int GPB = kBlock / G;
    const int gl = threadIdx.x % G;
    const int64_t r = (int64_t)blockIdx.x * GPB + threadIdx.x / G;
    if (r >= rows) return;
    float4 X[NV];
    load_row4<G, NV>(X, tab + r * d, d >> 2, gl);
    float n2 = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) { n2 = fmaf(X[v].x, X[v].x, n2); n2 = fmaf(X[v].y, X[v].y, n2); n2 = fmaf(X[v].z, X[v].z, n2); n2 = fmaf(X[v].w, X[v].w, n2); }
    n2 = gsum<G>(n2);
    const float nn = sqrtf(n2);
    if (gl == 0) out[r] = nn;
    if (hat) {   // rows of hat_row_bytes: padded to 4 * G * NV floats, or compact (d floats)
        const float inn = 1.0f / fmaxf(nn, kEpsNormalize);
        float4* const hrow = reinterpret_cast<float4*>(reinterpret_cast<char*>(hat) + r * (int64_t)hat_row_bytes);
        const int hvec = (int)(hat_row_bytes >> 4);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            X[v].x *= inn; X[v].y *= inn; X[v].z *= inn; X[v].w *= inn;
            if (v * G + gl < hvec) hrow[v * G + gl] = X[v];
        }
    }
}

// stand-alone sampler launch (first step of an epoch; later steps' sampling rides in the previous step's launch)
__global__ __launch_bounds__(256) void k_pull_sample(PullSampleArgs sa) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < sa.n) pull_sample_one(sa, i);
}

// the same registration from explicit negatives (parity tests drive the step with the reference's golden batches)
__global__ __launch_bounds__(256) void k_pull_lists_explicit(const int4* __restrict__ pairs, const int32_t* __restrict__ inv,
                                                             const int64_t* __restrict__ nh, const int64_t* __restrict__ nt, int64_t n,
                                                             PullLists out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 p = pairs[i];
    const bool tail = nh[i] == p.x;   // the sampler's own rule (kge_score.hip: my_tail = nh == sh)
    const int c = (int)(tail ? nt[i] : nh[i]);
    const int pos = atomicAdd(out.count + c, 1);
    out.pc[i] = c | ((int)tail << 24) | ((pos == 0 ? 1 : 0) << kPcFirstBit);
    if (pos < kPullCap) {
        out.bucket[(int64_t)c * kPullCap + pos] = (int)i;
        out.dbucket[(int64_t)c * kPullCap + pos] = make_int4(p.x, p.y, p.z, (int)i | ((int)tail << 24) | (kRoleC << 25));
    } else out.next[i] = atomicExch(out.head + c, (int)i);
    const int w = c | ((int)tail << 24);
    for (int role = 0; role < 3; ++role) out.sdesc[inv[3 * i + role]] = make_int4(p.x, p.y, p.z, w | (role << 25));
}

// ------------------------------------------------------------------ host side
// float4-per-lane geometry (row length d, d % 4 == 0, d <= 1024): G-lane owner groups, NV float4s per lane, padded row =
// 4 * G * NV floats.  32-lane groups by default; KGE_PULL_G=16 selects 16-lane groups for rows of up to 128 floats (four
// owners per wave share the per-visit fixed work, but diverge more: measured 38.8 vs 33.5 us per step at FB15k shape).
struct PullGeo { int G, NV; };
static PullGeo pull_geo(int dim) {
    PullGeo g{0, 0};
    if (dim <= 0 || (dim & 3) || dim > 1024) return g;
    const int nvec = dim >> 2;
    if (nvec <= 32 && switch_value("PULL_G") == 16) {
        g.G = 16; g.NV = nvec <= 16 ? 1 : 2;
        return g;
    }
    g.G = 32;
    for (int nv = 1; nv <= 8; nv <<= 1)
        if (nvec <= 32 * nv) { g.NV = nv; return g; }
    g.G = 0;
    return g;
}

template <int OPT, int G, int NV>
static int launch_pull_geo(PullArgs& a, const PullSampleArgs& sa, float* loss, hipStream_t s) {
    constexpr int GPB = kBlock / G;
    const int item_blocks = (int)((a.n_items + (a.dense_skip ? a.n_rows : 0) + GPB - 1) / GPB);
    a.sample_blocks = sa.n > 0 ? (int)((sa.n + kBlock - 1) / kBlock) : 0;
    if (a.recs) {   // two-phase form (L1 only): evaluate every pair once, then let the owners sum the records
        const unsigned eb = (unsigned)((a.n_pairs + GPB * kEvalPP - 1) / (GPB * kEvalPP));
        hipLaunchKernelGGL((k_pull_eval<true, G, NV>), dim3(eb), dim3(kBlock), 0, s, a, loss);
        hipLaunchKernelGGL((k_pull_step<OPT, true, G, NV, true>), dim3((unsigned)(item_blocks + a.sample_blocks)), dim3(kBlock), 0, s, a, sa, loss);
    } else if (a.l1)
        hipLaunchKernelGGL((k_pull_step<OPT, true, G, NV>), dim3((unsigned)(item_blocks + a.sample_blocks)), dim3(kBlock), 0, s, a, sa, loss);
    else
        hipLaunchKernelGGL((k_pull_step<OPT, false, G, NV>), dim3((unsigned)(item_blocks + a.sample_blocks)), dim3(kBlock), 0, s, a, sa, loss);
    int rc = check_launch("k_pull_step");
    if (rc || a.n_multi == 0) return rc;
    hipLaunchKernelGGL((k_pull_finish<OPT, G, NV>), dim3((unsigned)((a.n_multi + GPB - 1) / GPB)), dim3(kBlock), 0, s, a);
    return check_launch("k_pull_finish");
}

template <int OPT>
static int launch_pull_opt(PullArgs& a, const PullSampleArgs& sa, PullGeo g, float* loss, hipStream_t s) {
#define KGE_PG(G_, NV_) if (g.G == G_ && g.NV == NV_) return launch_pull_geo<OPT, G_, NV_>(a, sa, loss, s);
    KGE_PG(16, 1) KGE_PG(16, 2) KGE_PG(32, 1) KGE_PG(32, 2) KGE_PG(32, 4) KGE_PG(32, 8)
#undef KGE_PG
    return -1;
}

// bytes of the two-phase form's scratch for n pairs: direction codes (L1: one byte per lane and residual; L2: a float4) and records
void pull_direction_bytes(int dim, int l1, int64_t n, size_t* codes, size_t* recs) {
    const PullGeo g = pull_geo(dim);
    *codes = g.G ? (l1 ? (size_t)n * 2 * g.G * g.NV : (size_t)16) : 0;   // (L2 has no two-phase form: the buffers only have to exist)
    *recs = (size_t)n * 16;
}
int pull_partial_stride(int dim) { const PullGeo g = pull_geo(dim); return 4 * g.G * g.NV; }
// Bytes between hat rows (kge_pull_hat_stride floats: callers size and slice their hat buffers with it).  Read per launch: every kernel of
// a run -- and the caller's slicing -- must see the same value of the switch.
static unsigned hat_row_bytes(const PullGeo& g, int dim) {
    const int sw = switch_value("HAT_COMPACT");
    return (sw >= 0 ? sw == 1 : kHatCompactDefault) ? 4u * (unsigned)dim : 16u * (unsigned)g.G * (unsigned)g.NV;
}
int pull_hat_stride(int dim) { const PullGeo g = pull_geo(dim); return g.G ? (int)(hat_row_bytes(g, dim) / 4) : 0; }
int pull_groups_per_block(int dim) { const PullGeo g = pull_geo(dim); return g.G ? kBlock / g.G : 0; }

int launch_pull_step(const kge_model_desc* m, float* const tables_out[2], const float* const hat_in[2], float* const hat_out[2],
                     const float* norm_in, float* norm_out, float* const state1[2], float* const state2[2], const int32_t* pairs,
                     const kge_pull_lists* lists, const int32_t* items, int64_t n_items, const uint32_t* dense_skip, const int32_t* inc, float* partials,
                     const int32_t* multi, int64_t n_multi, float margin, int optimizer, float lr, int64_t step,
                     const float* dev_hyper, int reset_lists, const int32_t* next_pairs, const int32_t* next_inv, int64_t next_n, const float* bern,
                     const uint64_t* slots, int64_t n_slots, uint64_t seed, uint64_t next_offset,
                     const kge_pull_lists* next_lists, float* loss, const kge_pull_direction* dir, hipStream_t s) {
    const PullGeo geo = pull_geo(m->dim);
    if (!geo.G) { set_error("kge_pull_step: hidden size %d must be a multiple of 4 and at most 1024", m->dim); return -1; }
    {   // the row gathers use 32-bit byte offsets into the padded normalised tables (16 * G * NV bytes per row)
        const uint64_t row_bytes = 16ull * (uint64_t)geo.G * (uint64_t)geo.NV;
        const uint64_t rows = (uint64_t)(m->tot_entity > m->tot_relation ? m->tot_entity : m->tot_relation);
        if (rows * row_bytes > 0xFFFFFFFFull) {
            set_error("kge_pull_step: %llu rows of %llu bytes exceed the 4 GiB the 32-bit gather offsets address",
                      (unsigned long long)rows, (unsigned long long)row_bytes);
            return -1;
        }
    }
    if (dir && dir->codes && (uint64_t)dir->n_pairs * (uint64_t)(2 * geo.G * geo.NV) > 0xFFFFFFFFull) {   // (L1 codes: 32-bit byte offsets)
        set_error("kge_pull_step: %lld pairs exceed the 4 GiB the 32-bit offsets into the direction codes address", (long long)dir->n_pairs);
        return -1;
    }
    PullArgs a;
    for (int i = 0; i < 2; ++i) {
        a.tab_in[i] = m->tables[i]; a.tab_out[i] = tables_out[i];
        a.hat_in[i] = hat_in[i]; a.hat_out[i] = hat_out[i];
        a.s1[i] = state1 ? state1[i] : nullptr; a.s2[i] = state2 ? state2[i] : nullptr;
    }
    a.hat_row_bytes = hat_row_bytes(geo, m->dim);
    a.norm_in = norm_in; a.norm_out = norm_out;
    a.pairs = (const int4*)pairs; a.lists = to_lists(lists);
    a.items = (const int4*)items; a.inc = inc; a.partials = partials; a.multi = (const int4*)multi;
    a.n_items = n_items; a.n_multi = n_multi; a.sample_blocks = 0;
    a.dense_skip = dense_skip; a.n_rows = (int)(m->tot_entity + m->tot_relation);
    a.E = (int)m->tot_entity; a.d = m->dim; a.l1 = (m->flags & KGE_FLAG_L1) ? 1 : 0; a.reset_lists = reset_lists;
    a.margin = margin;
    a.theta = m->model == KGE_TRANSM ? m->tables[2] : nullptr;
    a.opt = make_opt_args(lr, step < 1 ? 1 : step);
    a.dev_hyper = dev_hyper;
    // The two-phase form exists for L1 only.  For L2 both evaluate-once forms lost to the one-phase step on the same box and were
    // removed: residual rows staged by phase 1 (1 KB per pair: 46.3 -> 53.6 us, round 3) and records of the two scalars with the owners
    // recomputing the residuals from their four gathers (45.9 -> 51.3 us, profiles/r05_l2_mid_ab.txt) -- an L2 visit is bound by its
    // row gathers, not by the reductions an evaluate-once form saves.  A direction passed with an L2 model is ignored.
    const bool two_phase = dir && dir->recs && (m->flags & KGE_FLAG_L1);
    a.recs = two_phase ? reinterpret_cast<float4*>(dir->recs) : nullptr;
    a.codes = two_phase ? dir->codes : nullptr;
    a.n_pairs = two_phase ? dir->n_pairs : 0;
    PullSampleArgs sa = make_sample_args(next_pairs, next_inv, next_pairs && next_lists ? next_n : 0, m->tot_entity, bern, slots,
                                         n_slots, seed, next_offset, nullptr, next_lists);
    sa.no_desc = two_phase && dir->lists_without_descriptors ? 1 : 0;
    switch (optimizer) {
        case KGE_OPT_SGD: return launch_pull_opt<KGE_OPT_SGD>(a, sa, geo, loss, s);
        case KGE_OPT_ADAM: return launch_pull_opt<KGE_OPT_ADAM>(a, sa, geo, loss, s);
        case KGE_OPT_ADAGRAD: return launch_pull_opt<KGE_OPT_ADAGRAD>(a, sa, geo, loss, s);
        case KGE_OPT_RMSPROP: return launch_pull_opt<KGE_OPT_RMSPROP>(a, sa, geo, loss, s);
        case KGE_OPT_GRADIENT: return launch_pull_opt<KGE_OPT_GRADIENT>(a, sa, geo, loss, s);
    }
    set_error("kge_pull_step: unknown optimizer %d", optimizer);
    return -1;
}

int launch_row_norms(const float* table, int64_t rows, int dim, float* out, float* hat, hipStream_t s) {
    const PullGeo g = pull_geo(dim);
    if (!g.G) { set_error("kge_row_norms: row length %d must be a multiple of 4 and at most 1024", dim); return -1; }
    if (rows == 0) return 0;
#define KGE_RN(G_, NV_)                                                                                               \
    if (g.G == G_ && g.NV == NV_) {                                                                                    \
        hipLaunchKernelGGL((k_row_norms<G_, NV_>), dim3((unsigned)((rows + kBlock / G_ - 1) / (kBlock / G_))), dim3(kBlock), 0, s, \
                           table, rows, dim, out, hat, hat_row_bytes(g, dim));                                         \
        return check_launch("k_row_norms");                                                                            \
    }
    KGE_RN(16, 1) KGE_RN(16, 2) KGE_RN(32, 1) KGE_RN(32, 2) KGE_RN(32, 4) KGE_RN(32, 8)
#undef KGE_RN
    return -1;
}

int launch_pull_sample(const int32_t* pairs, const int32_t* inv, int64_t n, int64_t E, const float* bern, const uint64_t* slots, int64_t n_slots,
                       uint64_t seed, uint64_t offset, const int64_t* cursor, const kge_pull_lists* out, hipStream_t s) {
    const PullSampleArgs sa = make_sample_args(pairs, inv, n, E, bern, slots, n_slots, seed, offset, cursor, out);
    hipLaunchKernelGGL(k_pull_sample, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, sa);
    return check_launch("k_pull_sample");
}

int launch_pull_lists_explicit(const int32_t* pairs, const int32_t* inv, const int64_t* nh, const int64_t* nt, int64_t n,
                               const kge_pull_lists* out, hipStream_t s) {
    hipLaunchKernelGGL(k_pull_lists_explicit, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const int4*)pairs, inv, nh, nt, n,
                       to_lists(out));
    return check_launch("k_pull_lists_explicit");
}

}  // namespace kge
