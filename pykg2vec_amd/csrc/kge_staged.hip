// kge_staged.hip -- optimiser sweep that PULLS staged gradient rows (include/kge_hip.h: kge_optimizer_step_staged).
// One wave owns one parameter row: it sums the row's staging slots in ascending slot order (static incidences of the batch
// from a CSR built once per epoch order, dynamic ones -- the negatives that drew this entity -- from a per-entity bucket
// sorted by pair), applies the dense optimiser with torch.optim default semantics (utils/trainer.py:112-131) and writes
// parameter and state rows in place.  Replaces: zero_grad + atomic scatter + k_opt's gradient read / clear streams.
#include "kge_internal.h"
#include "kge_opt_device.h"

namespace kge {

// optimiser state is read once and written once per step: non-temporal accesses (kge_opt_device.h; -DKGE_NO_NT: plain ones,
// for A/B builds)
#ifdef KGE_NO_NT
#define KGE_STREAM_LOAD(ptr) stream_load<false>(ptr)
#define KGE_STREAM_STORE(ptr, v) stream_store<false>((ptr), (v))
#else
#define KGE_STREAM_LOAD(ptr) stream_load<true>(ptr)
#define KGE_STREAM_STORE(ptr, v) stream_store<true>((ptr), (v))
#endif

struct StagedKArgs {
    kge_staged_step st;
    OptArgs opt;
    int64_t rows_total;
    int32_t* clear_count; int32_t* clear_head;   // registration set of the next step (NULL: the caller clears)
    int n_rel_tables;
    int n_tables_cls[2];                         // number of entity-class / relation-class tables
};

template <int NV>
__device__ __forceinline__ void add_slot_scaled(float4 (&g)[NV], const float* __restrict__ slot, float sc, int nvec, int lane) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int i = lane + 64 * v;
        if (i < nvec) {
            const float4 x = *reinterpret_cast<const float4*>(slot + 4 * i);
            g[v].x = fmaf(sc, x.x, g[v].x); g[v].y = fmaf(sc, x.y, g[v].y); g[v].z = fmaf(sc, x.z, g[v].z); g[v].w = fmaf(sc, x.w, g[v].w);
        }
    }
}

template <int NV>
__device__ __forceinline__ void add_slot(float4 (&g)[NV], const float* __restrict__ slot, int nvec, int lane) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int i = lane + 64 * v;
        if (i < nvec) {
            const float4 x = *reinterpret_cast<const float4*>(slot + 4 * i);
            g[v].x += x.x; g[v].y += x.y; g[v].z += x.z; g[v].w += x.w;
        }
    }
}

constexpr int kRelChunk = 16;   // slots per pre-reduced chunk of a relation's list

// one wave per (chunk, relation table): partial[chunk][j] = sum of the chunk's slots in list order
template <int NV>
__global__ __launch_bounds__(256) void k_stage_rel_chunks(StagedKArgs a, int n_rel_tables) {
    const kge_staged_step& st = a.st;
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= (int64_t)st.n_chunks * n_rel_tables) return;
    const int c = (int)(w / n_rel_tables), j = (int)(w - (int64_t)c * n_rel_tables);
    // the j-th relation-class table's static site
    int site = 0, seen = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (k < st.n_tables && st.tables[k].cls == 1) { if (seen == j) site = st.tables[k].site_a; ++seen; }
    const int r = st.chunk_rel[c];
    const int lo = st.rel_off[r] + kRelChunk * (c - st.rel_chunk_off[r]);
    const int hi = min(st.rel_off[r + 1], lo + kRelChunk);
    const int nvec = st.dim >> 2;
    float4 g[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) g[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    int k = lo;
    for (; k + 8 <= hi; k += 8) {   // eight slot rows in flight; additions in list order
        float4 x8[8][NV];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float* slot = st.stage + ((int64_t)st.rel_inc[k + u] * st.static_slots + site) * st.stage_stride;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int i = lane + 64 * v;
                x8[u][v] = i < nvec ? *reinterpret_cast<const float4*>(slot + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int v = 0; v < NV; ++v) { g[v].x += x8[u][v].x; g[v].y += x8[u][v].y; g[v].z += x8[u][v].z; g[v].w += x8[u][v].w; }
    }
    for (; k < hi; ++k) add_slot<NV>(g, st.stage + ((int64_t)st.rel_inc[k] * st.static_slots + site) * st.stage_stride, nvec, lane);
    float* out = st.rel_partials + ((int64_t)c * n_rel_tables + j) * st.stage_stride;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int i = lane + 64 * v;
        if (i < nvec) reinterpret_cast<float4*>(out)[i] = g[v];
    }
}

template <int KIND, int NV>
__global__ __launch_bounds__(256) void k_opt_staged(StagedKArgs a) {
    const kge_staged_step& st = a.st;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    int tb = 0;
    int64_t id = row;
    constexpr bool SPARSE = KIND == KGE_OPT_SGD || KIND == KGE_OPT_ADAGRAD;
    if (SPARSE && st.touched_ent != nullptr) {
        // work list instead of every row: [entities of the positives | entities drawn this step | relations of the positives],
        // each crossed with the tables of its class; a drawn entity that also has static incidences is served by the first part
        const int net = a.n_tables_cls[0], nrt = a.n_tables_cls[1];
        const int64_t nA = (int64_t)st.n_touched_ent * net, nB = st.n_neg * net, nC = (int64_t)st.n_touched_rel * nrt;
        int cls_w, j;
        // (relation rows first: theirs are the long slot lists, so they should start with the first workgroups)
        if (row < nC) { id = st.touched_rel[row / nrt]; j = (int)(row % nrt); cls_w = 1; }
        else if (row < nC + nA) { id = st.touched_ent[(row - nC) / net]; j = (int)((row - nC) % net); cls_w = 0; }
        else if (row < nC + nA + nB) {
            const int64_t k = (row - nC - nA) / net;   // pair k: its entity, if the pair was the entity's first registrant
            id = st.dyn_list[k]; j = (int)((row - nC - nA) % net); cls_w = 0;
            if (id < 0 || st.ent_off[id + 1] > st.ent_off[id]) return;
        } else return;
        int seen = 0;
        tb = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < st.n_tables && st.tables[k].cls == cls_w) { if (seen == j) tb = k; ++seen; }
    } else {
        if (row >= a.rows_total) return;
        while (tb + 1 < st.n_tables && id >= st.tables[tb].rows) { id -= st.tables[tb].rows; ++tb; }
    }
    int cls = 0, site_a = 0, site_b = 0, dsite = -1;
    int64_t flat_off = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k)   // (static indexing of the kernel-argument array: no scratch)
        if (k == tb) { cls = st.tables[k].cls; site_a = st.tables[k].site_a; site_b = st.tables[k].site_b; dsite = st.tables[k].dsite; flat_off = st.tables[k].flat_off; }
    const int d = st.dim, nvec = d >> 2;
    const int64_t stride = st.stage_stride;
    float* p = st.param + flat_off + id * d;
    float* s1 = st.state1 ? st.state1 + flat_off + id * d : nullptr;
    float* s2 = st.state2 ? st.state2 + flat_off + id * d : nullptr;
    constexpr bool DENSE = KIND == KGE_OPT_ADAM || KIND == KGE_OPT_RMSPROP;   // every row moves every step
    // work-list mode with short rows (NV == 1): nearly every visited row has a slot, and a wave's life is a chain of dependent
    // round trips (offsets -> incidences -> slot rows -> bucket -> ...): start the row's own parameter / state loads now
    const bool early = !DENSE && NV == 1 && st.touched_ent != nullptr;
    float4 pe = make_float4(0, 0, 0, 0), ae = make_float4(0, 0, 0, 0);
    if (early && lane < (d >> 2)) {
        pe = KGE_STREAM_LOAD(reinterpret_cast<const float4*>(p) + lane);
        if constexpr (KIND != KGE_OPT_SGD) ae = KGE_STREAM_LOAD(reinterpret_cast<const float4*>(s1) + lane);
    }
    float4 g[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) g[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    bool any = false;
    // ---- relation rows with pre-reduced chunks: sum the partial rows in chunk order
    if (cls == 1 && st.rel_chunk_off != nullptr) {
        int jrel = 0;   // index of this table among the relation-class tables
#pragma unroll
        for (int k = 0; k < 8; ++k) jrel += (k < tb && st.tables[k].cls == 1) ? 1 : 0;
        const int c0 = st.rel_chunk_off[id], c1 = st.rel_chunk_off[id + 1];
        any = c1 > c0;
        int c = c0;
        constexpr int UP = NV == 1 ? 8 : 4;   // partial rows in flight (no more live registers than the slot walk below needs)
        for (; c + UP <= c1; c += UP) {   // additions stay in chunk order
            float4 x8[UP][NV];
#pragma unroll
            for (int u = 0; u < UP; ++u)
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const int i = lane + 64 * v;
                    x8[u][v] = i < nvec ? *reinterpret_cast<const float4*>(st.rel_partials + ((int64_t)(c + u) * a.n_rel_tables + jrel) * stride + 4 * i)
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
            for (int u = 0; u < UP; ++u)
#pragma unroll
                for (int v = 0; v < NV; ++v) { g[v].x += x8[u][v].x; g[v].y += x8[u][v].y; g[v].z += x8[u][v].z; g[v].w += x8[u][v].w; }
        }
        for (; c < c1; ++c)
            add_slot<NV>(g, st.rel_partials + ((int64_t)c * a.n_rel_tables + jrel) * stride, nvec, lane);
    } else
    // ---- static incidences (ascending positive index => ascending slot)
    {
        const int32_t* off = cls == 0 ? st.ent_off : st.rel_off;
        const int32_t* inc = cls == 0 ? st.ent_inc : st.rel_inc;
        const int lo = off[id], hi = off[id + 1];
        any = hi > lo;
        int k = lo;
        for (; k + 4 <= hi; k += 4) {   // four slot rows in flight; the additions keep slot order
            int64_t s4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int x = inc[k + u];
                s4[u] = cls == 0 ? (int64_t)(x >> 1) * st.static_slots + ((x & 1) ? site_b : site_a) : (int64_t)x * st.static_slots + site_a;
            }
            float4 x4[4][NV];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const int i = lane + 64 * v;
                    x4[u][v] = i < nvec ? *reinterpret_cast<const float4*>(st.stage + s4[u] * stride + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < NV; ++v) { g[v].x += x4[u][v].x; g[v].y += x4[u][v].y; g[v].z += x4[u][v].z; g[v].w += x4[u][v].w; }
        }
        for (; k < hi; ++k) {
            const int x = inc[k];
            const int64_t s = cls == 0 ? (int64_t)(x >> 1) * st.static_slots + ((x & 1) ? site_b : site_a) : (int64_t)x * st.static_slots + site_a;
            add_slot<NV>(g, st.stage + s * stride, nvec, lane);
        }
    }
    // ---- dynamic incidences: negatives that drew this entity, in ascending pair order
    if (cls == 0 && dsite >= 0) {
        const int cnt = st.dyn_count[id];
        // the OTHER registration set (used by the previous step, fully consumed) is cleared here for the next step: no
        // memset launches between steps.  One owner per entity does it (the table whose dynamic site is 0).
        if (dsite == 0 && lane == 0 && a.clear_count) { a.clear_count[id] = 0; a.clear_head[id] = 0; }
        if (cnt > 0) {
            any = true;
            const int64_t dyn_base = st.n_pos * st.static_slots;
            if (cnt <= st.dyn_cap && cnt <= 64) {
                const int mine = lane < cnt ? st.dyn_bucket[id * st.dyn_cap + lane] : 0x7FFFFFFF;
                int rank = 0;
                for (int m = 0; m < cnt; ++m) rank += __shfl(mine, m, 64) < mine ? 1 : 0;
                for (int r = 0; r < cnt; ++r) {
                    const unsigned long long who = __ballot(lane < cnt && rank == r);
                    const int src = __ffsll((long long)who) - 1;
                    const int pair = __shfl(mine, src, 64);
                    const float* sl = st.stage + (dyn_base + (int64_t)pair * st.dynamic_slots + dsite) * stride;
                    if (st.dyn_scale) add_slot_scaled<NV>(g, sl, st.dyn_scale[pair], nvec, lane);
                    else add_slot<NV>(g, sl, nvec, lane);
                }
            } else {   // overflowed bucket: selection by ascending pair over bucket + chain
                const int nb = cnt < st.dyn_cap ? cnt : st.dyn_cap;
                int last = -1;
                for (;;) {
                    int best = 0x7FFFFFFF;
                    for (int m = 0; m < nb; ++m) { const int j = st.dyn_bucket[id * st.dyn_cap + m]; if (j > last && j < best) best = j; }
                    for (int j = st.dyn_head[id] - 1; j >= 0; j = st.dyn_next[j]) if (j > last && j < best) best = j;
                    if (best == 0x7FFFFFFF) break;
                    const float* sl = st.stage + (dyn_base + (int64_t)best * st.dynamic_slots + dsite) * stride;
                    if (st.dyn_scale) add_slot_scaled<NV>(g, sl, st.dyn_scale[best], nvec, lane);
                    else add_slot<NV>(g, sl, nvec, lane);
                    last = best;
                }
            }
        }
    }
    // SGD / Adagrad: a zero gradient leaves parameter and state unchanged
    if constexpr (!DENSE) { if (!any) return; }
    // (issuing the parameter / state loads before the slot walk was measured slower: 177 vs 164 us at the C3 shape -- the
    // extra live registers cost more resident waves than the overlap buys)
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int i = lane + 64 * v;
        if (i < nvec) {
            float4 pv, av = make_float4(0, 0, 0, 0), bv = make_float4(0, 0, 0, 0);
            if (early) { pv = pe; av = ae; }
            else {
                pv = KGE_STREAM_LOAD(reinterpret_cast<const float4*>(p) + i);
                if constexpr (KIND != KGE_OPT_SGD) av = KGE_STREAM_LOAD(reinterpret_cast<const float4*>(s1) + i);
            }
            if constexpr (KIND == KGE_OPT_ADAM) bv = KGE_STREAM_LOAD(reinterpret_cast<const float4*>(s2) + i);
            opt_update<KIND>(pv.x, g[v].x, av.x, bv.x, a.opt);
            opt_update<KIND>(pv.y, g[v].y, av.y, bv.y, a.opt);
            opt_update<KIND>(pv.z, g[v].z, av.z, bv.z, a.opt);
            opt_update<KIND>(pv.w, g[v].w, av.w, bv.w, a.opt);
            reinterpret_cast<float4*>(p)[i] = pv;
            if constexpr (KIND != KGE_OPT_SGD) KGE_STREAM_STORE(reinterpret_cast<float4*>(s1) + i, av);
            if constexpr (KIND == KGE_OPT_ADAM) KGE_STREAM_STORE(reinterpret_cast<float4*>(s2) + i, bv);
        }
    }
}

template <int KIND>
static int launch_staged_kind(const StagedKArgs& a, hipStream_t s) {
    const int nvec = a.st.dim >> 2;
    if (a.st.rel_chunk_off != nullptr && a.st.n_chunks > 0) {
        const dim3 cg((unsigned)(((int64_t)a.st.n_chunks * a.n_rel_tables + 3) / 4));
        if (nvec <= 64) hipLaunchKernelGGL((k_stage_rel_chunks<1>), cg, dim3(256), 0, s, a, a.n_rel_tables);
        else if (nvec <= 128) hipLaunchKernelGGL((k_stage_rel_chunks<2>), cg, dim3(256), 0, s, a, a.n_rel_tables);
        else if (nvec <= 256) hipLaunchKernelGGL((k_stage_rel_chunks<4>), cg, dim3(256), 0, s, a, a.n_rel_tables);
        else hipLaunchKernelGGL((k_stage_rel_chunks<8>), cg, dim3(256), 0, s, a, a.n_rel_tables);
    }
    const dim3 grid((unsigned)((a.rows_total + 3) / 4));
    if (nvec <= 64) hipLaunchKernelGGL((k_opt_staged<KIND, 1>), grid, dim3(256), 0, s, a);
    else if (nvec <= 128) hipLaunchKernelGGL((k_opt_staged<KIND, 2>), grid, dim3(256), 0, s, a);
    else if (nvec <= 256) hipLaunchKernelGGL((k_opt_staged<KIND, 4>), grid, dim3(256), 0, s, a);
    else if (nvec <= 512) hipLaunchKernelGGL((k_opt_staged<KIND, 8>), grid, dim3(256), 0, s, a);
    else { set_error("kge_optimizer_step_staged: rows longer than 2048 floats are not supported"); return -1; }
    return check_launch("k_opt_staged");
}

int launch_optimizer_staged(int kind, const kge_staged_step* st, float lr, int64_t step, hipStream_t s) {
    if (!st || !st->param || !st->stage || !st->ent_off || !st->rel_off) { set_error("kge_optimizer_step_staged: null arguments"); return -1; }
    if (st->n_tables < 1 || st->n_tables > 8 || st->dim <= 0 || st->dim % 4 || st->stage_stride % 4 || st->stage_stride < st->dim) {
        set_error("kge_optimizer_step_staged: bad table list / row length (dim and stage_stride must be multiples of 4)");
        return -1;
    }
    if ((((uintptr_t)st->param | (uintptr_t)st->state1 | (uintptr_t)st->state2 | (uintptr_t)st->stage) & 15)) {
        set_error("kge_optimizer_step_staged: buffers must be 16-byte aligned");
        return -1;
    }
    StagedKArgs a;
    a.st = *st;
    a.opt = make_opt_args(lr, step);
    a.clear_count = st->dyn_count_next;
    a.clear_head = st->dyn_head_next;
    a.n_tables_cls[0] = a.n_tables_cls[1] = 0;
    for (int k = 0; k < st->n_tables; ++k) a.n_tables_cls[st->tables[k].cls == 1 ? 1 : 0] += 1;
    const bool sparse_kind = kind == KGE_OPT_SGD || kind == KGE_OPT_ADAGRAD;
    if (st->touched_ent && sparse_kind) {
        if (!st->touched_rel || !st->dyn_list || st->dyn_count_next) {
            set_error("kge_optimizer_step_staged: touched-row lists need touched_rel, dyn_list and a single registration set "
                      "(dyn_count_next == NULL: the sweep no longer visits every entity, so it cannot clear the other set)");
            return -1;
        }
    }
    a.rows_total = 0;
    a.n_rel_tables = 0;
    for (int k = 0; k < st->n_tables; ++k) a.n_rel_tables += st->tables[k].cls == 1 ? 1 : 0;
    if (st->rel_chunk_off && (!st->chunk_rel || !st->rel_partials || st->n_chunks < 0)) {
        set_error("kge_optimizer_step_staged: relation chunking needs chunk_rel, rel_partials and n_chunks");
        return -1;
    }
    for (int k = 0; k < st->n_tables; ++k) {
        if (st->tables[k].flat_off % 4) { set_error("kge_optimizer_step_staged: table offsets must be multiples of 4 floats"); return -1; }
        a.rows_total += st->tables[k].rows;
    }
    if (st->touched_ent && sparse_kind)   // work list: [touched entities | drawn entities (upper bound n_neg) | touched relations]
        a.rows_total = ((int64_t)st->n_touched_ent + st->n_neg) * a.n_tables_cls[0] + (int64_t)st->n_touched_rel * a.n_tables_cls[1];
    switch (kind) {
        case KGE_OPT_SGD: return launch_staged_kind<KGE_OPT_SGD>(a, s);
        case KGE_OPT_ADAM:
            if (!st->state1 || !st->state2) { set_error("adam needs two state buffers"); return -1; }
            return launch_staged_kind<KGE_OPT_ADAM>(a, s);
        case KGE_OPT_ADAGRAD:
            if (!st->state1) { set_error("adagrad needs a state buffer"); return -1; }
            return launch_staged_kind<KGE_OPT_ADAGRAD>(a, s);
        case KGE_OPT_RMSPROP:
            if (!st->state1) { set_error("rmsprop needs a state buffer"); return -1; }
            return launch_staged_kind<KGE_OPT_RMSPROP>(a, s);
    }
    set_error("kge_optimizer_step_staged: unknown optimizer %d", kind);
    return -1;
}

}  // namespace kge
