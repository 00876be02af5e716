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
};

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

template <int KIND, int NV>
__global__ __launch_bounds__(256) void k_opt_staged(StagedKArgs a) {
    const kge_staged_step& st = a.st;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows_total) return;
    // which table
    int tb = 0;
    int64_t id = row;
    while (tb + 1 < st.n_tables && id >= st.tables[tb].rows) { id -= st.tables[tb].rows; ++tb; }
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
    float4 g[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) g[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    bool any = false;
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
        if (dsite == 0 && lane == 0 && a.clear_count) { a.clear_count[id] = 0; a.clear_head[id] = -1; }
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
                    add_slot<NV>(g, st.stage + (dyn_base + (int64_t)pair * st.dynamic_slots + dsite) * stride, nvec, lane);
                }
            } else {   // overflowed bucket: selection by ascending pair over bucket + chain
                const int nb = cnt < st.dyn_cap ? cnt : st.dyn_cap;
                int last = -1;
                for (;;) {
                    int best = 0x7FFFFFFF;
                    for (int m = 0; m < nb; ++m) { const int j = st.dyn_bucket[id * st.dyn_cap + m]; if (j > last && j < best) best = j; }
                    for (int j = st.dyn_head[id]; j >= 0; j = st.dyn_next[j]) if (j > last && j < best) best = j;
                    if (best == 0x7FFFFFFF) break;
                    add_slot<NV>(g, st.stage + (dyn_base + (int64_t)best * st.dynamic_slots + dsite) * stride, nvec, lane);
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
            float4 pv = KGE_STREAM_LOAD(reinterpret_cast<const float4*>(p) + i);
            float4 av = make_float4(0, 0, 0, 0), bv = make_float4(0, 0, 0, 0);
            if constexpr (KIND != KGE_OPT_SGD) av = KGE_STREAM_LOAD(reinterpret_cast<const float4*>(s1) + i);
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
    a.rows_total = 0;
    for (int k = 0; k < st->n_tables; ++k) {
        if (st->tables[k].flat_off % 4) { set_error("kge_optimizer_step_staged: table offsets must be multiples of 4 floats"); return -1; }
        a.rows_total += st->tables[k].rows;
    }
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
