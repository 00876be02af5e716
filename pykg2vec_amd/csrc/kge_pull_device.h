// kge_pull_device.h -- pieces shared by the owner-computes training steps (kge_pull.hip: TransE / TransM with the fused dense
// optimiser; kge_own.hip: the two-phase step of the pointwise models): incidence roles, the per-step sampler lists and the
// sampler that fills them, float4 row movement.
#pragma once
#include "kge_row_kernels.h"
#include "kge_sampler_device.h"

namespace kge {

constexpr int kRoleH = 0, kRoleT = 1, kRoleR = 2, kRoleC = 3;
constexpr int kPullCap = 16;   // per-entity bucket of "drawn as corrupting entity" pairs; overflow goes to a linked list
constexpr int kPcFirstBit = 27;   // pc[i] bit 27: pair i was the FIRST to register with its corrupting entity this step (the entity's
                                  // owner-by-default when the entity has no work item of its own: sparse optimisers, kge_own.hip)

// per-step sampler output: which pairs drew entity e as their corrupting entity
struct PullLists {
    int32_t* pc;       // [B]  per pair: corrupting entity | (tail corrupted) << 24 | (first registrant of the entity) << 27
    int32_t* count;    // [E]  number of pairs that drew e this step (reset to 0 by e's owner)
    int32_t* bucket;   // [E * kPullCap] the first kPullCap of them, in arrival (i.e. arbitrary) order (pair indices: overflow path)
    int32_t* head;     // [E]  overflow list head (-1: none; reset by e's owner)
    int32_t* next;     // [B]  overflow list links
    int4* sdesc;       // [3B] visit descriptors of the static incidences, in `inc` order: (h, r, t, c | tail << 24 | role << 25)
    int4* dbucket;     // [E * kPullCap] visit descriptors of an entity's drawers: (h, r, t, pair | tail << 24 | kRoleC << 25)
};

struct PullSampleArgs {
    const int4* pairs;         // batch to sample: (h, r, t, -)
    const int32_t* inv;        // [3n] position of incidence (pair, role) in the batch's sorted incidence list
    int64_t n, E;
    const float* bern;
    const unsigned long long* slots;
    unsigned long long mask, seed, offset;
    const int64_t* cursor;
    PullLists out;
    int no_desc;               // != 0: skip the visit descriptors (the two-phase step's owners read pair records instead)
};

// one pair of the sampled batch: draw the corruption (same Philox counters as kge_sample_batch / the fused push kernels:
// offset + pair index) and register the pair with the corrupting entity
__device__ __forceinline__ void pull_sample_one(const PullSampleArgs& sa, int64_t i) {
    const unsigned long long off = sa.cursor ? sa.offset + (unsigned long long)sa.cursor[1] : sa.offset;
    const int4 p = sa.pairs[i];
    int64_t nh, nt;
    corrupt_one(p.x, p.y, p.z, sa.E, sa.bern, sa.slots, sa.mask, sa.seed, off + (unsigned long long)i, nh, nt);
    const bool tail = nh == p.x;
    const int c = (int)(tail ? nt : nh);
    const int pos = atomicAdd(sa.out.count + c, 1);
    sa.out.pc[i] = c | ((int)tail << 24) | ((pos == 0 ? 1 : 0) << kPcFirstBit);
    if (pos < kPullCap) {
        sa.out.bucket[(int64_t)c * kPullCap + pos] = (int)i;
        if (!sa.no_desc) sa.out.dbucket[(int64_t)c * kPullCap + pos] = make_int4(p.x, p.y, p.z, (int)i | ((int)tail << 24) | (kRoleC << 25));
    } else sa.out.next[i] = atomicExch(sa.out.head + c, (int)i);
    if (sa.no_desc) return;
    // the three static incidences of the pair, ready for their owners (one dependent load instead of inc -> pairs + pc)
    const int w = c | ((int)tail << 24);
#pragma unroll
    for (int role = 0; role < 3; ++role) sa.out.sdesc[sa.inv[3 * i + role]] = make_int4(p.x, p.y, p.z, w | (role << 25));
}

// The owner's visit list, one descriptor per lane, staged in LDS in visit order: static incidences first (their descriptors were
// filed by the sampler in incidence order), then the pairs that drew this entity as their corrupting entity (bucket descriptors,
// requested speculatively together with the count, ordered by pair index).  Returns the number of visits; *cnt_out = draws of the
// entity, *fast = they all fit the bucket and the lane group (otherwise the caller walks bucket + chain one visit at a time).
template <int G>
__device__ __forceinline__ int own_visit_list(const PullLists& lists, const int4 it, int g, bool walks_c, int gl, int gbase,
                                              int4* __restrict__ s_desc_row, int* cnt_out, bool* fast) {
    const int n_static = it.z - it.y;
    const int q = gl - n_static;
    int4 ds = make_int4(0, 0, 0, -1);
    bool have = false;
    if (gl < n_static) { ds = lists.sdesc[it.y + gl]; have = true; }
    int cnt = 0;
    int4 dd = make_int4(0, 0, 0, -1);
    if (walks_c) {
        cnt = lists.count[g];
        if (q >= 0 && q < kPullCap) dd = lists.dbucket[(int64_t)g * kPullCap + q];   // speculative: valid for q < cnt
    }
    const bool fast_c = cnt <= kPullCap && n_static + cnt <= G;
    int slot = gl, nvis = n_static;
    if (cnt > 0 && fast_c) {
        const bool mine = q >= 0 && q < cnt;
        const int key = mine ? (dd.w & 0xFFFFFF) : 0x7FFFFFFF;   // pair index
        if (cnt > 1) {   // arrival order is arbitrary: rank the entries by pair index, visit by rank
            int rank = 0;
            for (int m = 0; m < cnt; ++m) rank += __shfl(key, gbase + n_static + m, 64) < key ? 1 : 0;
            if (mine) slot = n_static + rank;
        }
        if (mine) { ds = dd; ds.w = (dd.w & ~0xFFFFFF) | g; have = true; }   // the corrupting entity is the owner itself
        nvis += cnt;
    }
    if (have) s_desc_row[slot] = ds;
    *cnt_out = cnt; *fast = fast_c;
    return nvis;
}

// The same list for the two-phase ("staged direction") step: a visit is (pair << 2 | role) -- static incidences straight from
// `inc`, drawers from the entity's bucket of pair indices, ranked by pair index -- because the owner needs nothing but the pair's
// evaluation record, which phase 1 left behind.
// ranked == false: the drawers are visited in arrival order (the caller's sums are exact in any order -- TransE's integer half units --,
// so the 47-instruction ranking loop per bucket entry is skipped: it was a quarter of the owner kernel's VALU instructions)
template <int G>
__device__ __forceinline__ int own_visit_list_dir(const PullLists& lists, const int32_t* __restrict__ inc, const int4 it, int g,
                                                  bool walks_c, int gl, int gbase, int* __restrict__ s_vis_row, int* cnt_out,
                                                  bool* fast, bool ranked = true) {
    const int n_static = it.z - it.y;
    const int q = gl - n_static;
    int e = -1;
    bool have = false;
    if (gl < n_static) { e = inc[it.y + gl]; have = true; }
    int cnt = 0, dd = 0x7FFFFFFF;
    if (walks_c) {
        cnt = lists.count[g];
        if (q >= 0 && q < kPullCap) dd = lists.bucket[(int64_t)g * kPullCap + q];   // speculative: valid for q < cnt
    }
    const bool fast_c = cnt <= kPullCap && n_static + cnt <= G;
    int slot = gl, nvis = n_static;
    if (cnt > 0 && fast_c) {
        const bool mine = q >= 0 && q < cnt;
        const int key = mine ? dd : 0x7FFFFFFF;
        if (cnt > 1 && ranked) {
            int rank = 0;
            for (int m = 0; m < cnt; ++m) rank += __shfl(key, gbase + n_static + m, 64) < key ? 1 : 0;
            if (mine) slot = n_static + rank;
        }
        if (mine) { e = (dd << 2) | kRoleC; have = true; }
        nvis += cnt;
    }
    if (have) s_vis_row[slot] = e;
    *cnt_out = cnt; *fast = fast_c;
    return nvis;
}

// Rows as float4 per lane: lane gl of a G-lane group holds elements 4*(v*G + gl) .. +3 for v < NV (d % 4 == 0): one
// 16-byte load / store instruction per lane moves a whole 100-float row with 25 lanes.
template <int G, int NV>
__device__ __forceinline__ void load_row4(float4 (&x)[NV], const float* __restrict__ row, int nvec, int gl) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int i = v * G + gl;
        x[v] = i < nvec ? reinterpret_cast<const float4*>(row)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
template <int G, int NV>
__device__ __forceinline__ void store_row4(float* __restrict__ row, const float4 (&x)[NV], int nvec, int gl) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int i = v * G + gl;
        if (i < nvec) reinterpret_cast<float4*>(row)[i] = x[v];
    }
}
static inline PullLists to_lists(const kge_pull_lists* l) {
    PullLists o;
    o.pc = l->pc; o.count = l->count; o.bucket = l->bucket; o.head = l->head; o.next = l->next;
    o.sdesc = (int4*)l->sdesc; o.dbucket = (int4*)l->dbucket;
    return o;
}

static inline PullSampleArgs make_sample_args(const int32_t* pairs, const int32_t* inv, int64_t n, int64_t E, const float* bern, const uint64_t* slots,
                                       int64_t n_slots, uint64_t seed, uint64_t offset, const int64_t* cursor,
                                       const kge_pull_lists* out) {
    PullSampleArgs sa;
    sa.pairs = (const int4*)pairs; sa.inv = inv; sa.n = n; sa.E = E; sa.bern = bern;
    sa.slots = (const unsigned long long*)slots; sa.mask = (unsigned long long)(slots ? n_slots - 1 : 0);
    sa.seed = seed; sa.offset = offset; sa.cursor = cursor; sa.no_desc = 0;
    if (out) sa.out = to_lists(out);
    else { sa.out.pc = sa.out.count = sa.out.bucket = sa.out.head = sa.out.next = nullptr; sa.out.sdesc = sa.out.dbucket = nullptr; }
    return sa;
}


}  // namespace kge
