// kge_api.hip -- the extern "C" boundary of libkge_hip.so (declared in include/kge_hip.h): argument validation,
// error strings, dispatch to the launchers.  No device memory is allocated and no stream is synchronised here.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <cstdlib>
#include "kge_internal.h"

namespace kge {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

bool is_vector_model(int model) {
    switch (model) {
        case KGE_TRANSE: case KGE_TRANSH: case KGE_TRANSD: case KGE_ROTATE:
        case KGE_DISTMULT: case KGE_COMPLEX: case KGE_ANALOGY:
        case KGE_TRANSM: case KGE_CP: case KGE_SIMPLE: case KGE_SIMPLE_IGNR: case KGE_QUATE: return true;
        default: return false;
    }
}

bool pick_geometry(int d, Geometry* out) {
    if (d <= 0) return false;
    if (d <= 32) *out = {32, 1};
    else if (d <= 64) *out = {32, 2};
    else if (d <= 128) *out = {32, 4};
    else if (d <= 256) *out = {32, 8};
    else if (d <= 512) *out = {64, 8};
    else if (d <= 1024) *out = {64, 16};
    else if (d <= 2048) *out = {64, 32};   // (32 floats of every row per lane: functional -- the wide rows spill for the many-row models)
    else return false;
    return true;
}

static int table_count(int model) {
    switch (model) {
        case KGE_TRANSE: case KGE_DISTMULT: case KGE_RESCAL: return 2;
        case KGE_TRANSH: case KGE_ROTATE: return 3;
        case KGE_TRANSD: case KGE_COMPLEX: return 4;
        case KGE_NTN: case KGE_ANALOGY: return 6;
        case KGE_TRANSM: case KGE_CP: case KGE_TRANSR: return 3;
        case KGE_SIMPLE: case KGE_SIMPLE_IGNR: return 4;
        case KGE_QUATE: return 8;
    }
    return -1;
}
static int grad_count(int model) { return model == KGE_TRANSM ? 2 : table_count(model); }  // TransM's theta is a fixed input

DeviceModel to_device_model(const kge_model_desc* m) {
    DeviceModel d;
    for (int i = 0; i < KGE_MAX_TABLES; ++i) { d.tab[i] = m->tables[i]; d.grad[i] = m->grads[i]; }
    d.dim = m->dim;
    d.rel_dim = m->rel_dim;
    d.l1 = (m->flags & KGE_FLAG_L1) ? 1 : 0;
    d.margin = m->margin;
    d.phase_div = m->phase_scale != 0.f ? 1.0f / m->phase_scale : 1.0f;
    return d;
}

static int validate(const kge_model_desc* m, bool need_grads, const char* who) {
    if (!m) { set_error("%s: null model descriptor", who); return -1; }
    const int nt = table_count(m->model);
    if (nt < 0) { set_error("%s: unknown model id %d", who, m->model); return -1; }
    if (m->dim <= 0 || m->tot_entity <= 0 || m->tot_relation <= 0) {
        set_error("%s: bad sizes (dim %d, entities %lld, relations %lld)", who, m->dim, (long long)m->tot_entity,
                  (long long)m->tot_relation);
        return -1;
    }
    if (m->model == KGE_TRANSD && m->rel_dim != m->dim) {
        set_error("%s: TransD needs ent_hidden_size == rel_hidden_size (pairwise.py:277-278 broadcasts them)", who);
        return -1;
    }
    if (m->model == KGE_ANALOGY && (m->dim & 1)) { set_error("%s: ANALOGY needs an even hidden size", who); return -1; }
    for (int i = 0; i < nt; ++i) {
        if (!m->tables[i]) { set_error("%s: table %d is null", who, i); return -1; }
        if (need_grads && i < grad_count(m->model) && !m->grads[i]) { set_error("%s: gradient buffer %d is null", who, i); return -1; }
    }
    return 0;
}

// The train-triple hash set packs a triple into one 64-bit key h:24 | r:16 | t:24 (kge_sampler_device.h): beyond these
// ranges keys alias and valid negatives would be rejected silently, so every entry point that probes the set refuses.
static int validate_packed_key(const kge_model_desc* m, const char* who) {
    if (m->tot_entity > (1 << 24) || m->tot_relation > (1 << 16)) {
        set_error("%s: the packed train-triple key holds 2^24 entities and 2^16 relations (got %lld / %lld)", who,
                  (long long)m->tot_entity, (long long)m->tot_relation);
        return -1;
    }
    return 0;
}

}  // namespace kge

using namespace kge;

extern "C" {

int kge_abi_version(void) { return KGE_ABI_VERSION; }
const char* kge_last_error(void) { return g_err; }

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
static const int64_t kTransRRowsMinPairs = 1;      // the two-launch step wins at every batch size measured (128 ... 32 768 pairs: profiles/r04_transr_threshold.txt)

size_t kge_workspace_bytes(const kge_model_desc* m, int64_t n) {
    if (validate(m, false, "kge_workspace_bytes") || n < 0) return 0;
    if (is_vector_model(m->model)) return 0;
    // the pairwise step keeps one scorer workspace per side (positive / negative) plus the two score vectors
    size_t b = 2 * align256(dense_workspace_bytes(m, n)) + align256((size_t)2 * n * sizeof(float));
    if (m->model == KGE_RESCAL) b += rescal_slab_extra_bytes(m, n);   // V rows + energy shares of the slab form of the pairwise step
    return b;
}

int kge_score_forward(const kge_model_desc* m, const int64_t* h, const int64_t* r, const int64_t* t, int64_t n,
                      float* scores, void* workspace, size_t workspace_bytes, void* stream) {
    if (validate(m, false, "kge_score_forward")) return -1;
    if (n == 0) return 0;
    if (n < 0 || !h || !r || !t || !scores) { set_error("kge_score_forward: bad arguments"); return -1; }
    hipStream_t s = (hipStream_t)stream;
    if (int rc = debug_check_hrt("kge_score_forward", m, h, r, t, n, s)) return rc;
    if (m->model == KGE_RESCAL) return launch_rescal_forward(m, h, r, t, n, scores, workspace, workspace_bytes, s);
    if (m->model == KGE_NTN) return launch_ntn_forward(m, h, r, t, n, scores, workspace, workspace_bytes, s);
    if (m->model == KGE_TRANSR) return launch_transr_forward(m, h, r, t, n, scores, workspace, workspace_bytes, s);
    return launch_score_forward(m, h, r, t, n, scores, s);
}

int kge_score_backward(const kge_model_desc* m, const int64_t* h, const int64_t* r, const int64_t* t, int64_t n,
                       const float* dscore, void* workspace, size_t workspace_bytes, void* stream) {
    if (validate(m, true, "kge_score_backward")) return -1;
    if (n == 0) return 0;
    if (n < 0 || !h || !r || !t || !dscore) { set_error("kge_score_backward: bad arguments"); return -1; }
    hipStream_t s = (hipStream_t)stream;
    if (int rc = debug_check_hrt("kge_score_backward", m, h, r, t, n, s)) return rc;
    if (m->model == KGE_RESCAL) return launch_rescal_backward(m, h, r, t, n, dscore, workspace, workspace_bytes, false, s);
    if (m->model == KGE_NTN) return launch_ntn_backward(m, h, r, t, n, dscore, workspace, workspace_bytes, false, s);
    if (m->model == KGE_TRANSR) return launch_transr_backward(m, h, r, t, n, dscore, workspace, workspace_bytes, false, s);
    return launch_score_backward(m, h, r, t, n, dscore, s);
}

int kge_rescal_normalize(float* ent, int64_t tot_entity, float* rel, int64_t tot_relation, int32_t k, void* stream) {
    if (!ent || !rel || k <= 0) { set_error("kge_rescal_normalize: bad arguments"); return -1; }
    return launch_rescal_normalize(ent, tot_entity, rel, tot_relation, k, nullptr, 0, (hipStream_t)stream);
}

size_t kge_rescal_normalize_scratch_bytes(int64_t tot_relation, int32_t k) {
    return (size_t)tot_relation * (size_t)(((int64_t)k * k + 4095) / 4096) * sizeof(float);
}

int kge_rescal_normalize_ws(float* ent, int64_t tot_entity, float* rel, int64_t tot_relation, int32_t k, void* scratch,
                            size_t scratch_bytes, void* stream) {
    if (!rel || k <= 0) { set_error("kge_rescal_normalize_ws: bad arguments"); return -1; }
    return launch_rescal_normalize(ent, tot_entity, rel, tot_relation, k, (float*)scratch, scratch_bytes / sizeof(float),
                                   (hipStream_t)stream);
}

int kge_train_pairwise_hinge(const kge_model_desc* m, const int64_t* ph, const int64_t* pr, const int64_t* pt,
                             const int64_t* nh, const int64_t* nr, const int64_t* nt, int64_t n, float margin,
                             void* workspace, size_t workspace_bytes, float* loss, void* stream) {
    if (validate(m, true, "kge_train_pairwise_hinge")) return -1;
    if (n == 0) return 0;
    if (n < 0 || !ph || !pr || !pt || !nh || !nr || !nt || !loss) { set_error("kge_train_pairwise_hinge: bad arguments"); return -1; }
    hipStream_t s = (hipStream_t)stream;
    if (int rc = debug_check_hrt("kge_train_pairwise_hinge (positives)", m, ph, pr, pt, n, s)) return rc;
    if (int rc = debug_check_hrt("kge_train_pairwise_hinge (negatives)", m, nh, nr, nt, n, s)) return rc;
    if (is_vector_model(m->model)) return launch_pairwise_hinge(m, ph, pr, pt, nh, nr, nt, n, margin, loss, s);
    // dense-contraction models: forward over [positives | negatives], hinge coefficients in place, backward over the same 2n
    const size_t gws = align256(dense_workspace_bytes(m, n));
    if (!workspace || workspace_bytes < 2 * gws + align256((size_t)2 * n * sizeof(float))) {
        set_error("kge_train_pairwise_hinge: workspace too small (need kge_workspace_bytes)");
        return -1;
    }
    void* wsp = workspace;                       // scorer workspace: both sides as one batch of 2n triples (2 gws bytes)
    float* sp = (float*)((char*)workspace + 2 * gws);
    float* sn = sp + n;
    int rc;
    if (m->model == KGE_RESCAL) {
        // nr == pr (the same buffer: the caller's way of saying that negatives keep their positives' relations, as every sampler
        // of the reference does): scores, hinge and gradients of a (relation, 16 pairs) tile in one launch
        const bool unfused = switch_value("RESCAL_UNFUSED") == 1;   // A/B switch (same 0 / 1 meaning as Trainer.switches)
        if (nr == pr && rescal_pair_step_ok(m, n, 2 * gws) && !unfused)
            return launch_rescal_pair_step(m, ph, pr, pt, nh, nt, n, margin, loss, wsp, workspace_bytes, nullptr, nullptr, s);
        // positives and negatives as ONE grouped batch of 2n triples (scores / coefficients contiguous: sp | sn); the
        // region of the two per-side workspaces holds the grouping of 2n triples (group_ws_bytes(R, 2n) <= 2 gws)
        if ((rc = launch_rescal_pair_forward(m, ph, pr, pt, nh, nr, nt, n, sp, wsp, 2 * gws, s))) return rc;
        if ((rc = launch_hinge_coeffs(sp, sn, n, margin, loss, s))) return rc;
        return launch_rescal_pair_backward(m, ph, pr, pt, nh, nr, nt, n, sp, wsp, 2 * gws, s);
    }
    if (m->model == KGE_NTN) {
        if ((rc = launch_ntn_pair_forward(m, ph, pr, pt, nh, nr, nt, n, sp, wsp, 2 * gws, s))) return rc;
        if ((rc = launch_hinge_coeffs(sp, sn, n, margin, loss, s))) return rc;
        return launch_ntn_pair_backward(m, ph, pr, pt, nh, nr, nt, n, sp, wsp, 2 * gws, s);
    }
    if (m->model == KGE_TRANSR) {
        // nr == pr (one buffer): the two-launch step of kge_transr_rows.hip (KGE_TRANSR_ROWS=0: the tile kernels)
        const int rows_sw = switch_value("TRANSR_ROWS");
        if (nr == pr && rows_sw != 0 && (rows_sw >= 1 || n >= kTransRRowsMinPairs) && transr_rows_ok(m, n, 2 * gws))
            return launch_transr_pair_step(m, ph, pr, pt, nh, nt, n, margin, loss, wsp, 2 * gws, s);
        if ((rc = launch_transr_pair_forward(m, ph, pr, pt, nh, nr, nt, n, sp, wsp, 2 * gws, s))) return rc;
        if ((rc = launch_hinge_coeffs(sp, sn, n, margin, loss, s))) return rc;
        return launch_transr_pair_backward(m, ph, pr, pt, nh, nr, nt, n, sp, wsp, 2 * gws, s);
    }
    set_error("kge_train_pairwise_hinge: unsupported model %d", m->model);
    return -1;
}

int kge_train_pairwise_hinge_sampled(const kge_model_desc* m, const int64_t* triples, const int64_t* perm, int64_t start,
                                     int64_t n, const float* bern_prob, const uint64_t* slots, int64_t n_slots,
                                     uint64_t seed, uint64_t offset, const int64_t* dev_cursor, float margin, float* loss,
                                     void* stream) {
    if (validate(m, true, "kge_train_pairwise_hinge_sampled")) return -1;
    if (validate_packed_key(m, "kge_train_pairwise_hinge_sampled")) return -1;
    if (n == 0) return 0;
    if (n < 0 || start < 0 || !triples || !perm || !loss) { set_error("kge_train_pairwise_hinge_sampled: bad arguments"); return -1; }
    if (slots && (n_slots & (n_slots - 1))) { set_error("kge_train_pairwise_hinge_sampled: n_slots must be a power of two"); return -1; }
    if (!is_vector_model(m->model)) {
        set_error("kge_train_pairwise_hinge_sampled: model %d needs kge_sample_batch + kge_train_pairwise_hinge", m->model);
        return -1;
    }
    if (!dev_cursor)
        if (int rc = debug_check_triples("kge_train_pairwise_hinge_sampled", m->tot_entity, m->tot_relation, triples, n, (hipStream_t)stream, perm, start)) return rc;
    return launch_pairwise_hinge_sampled(m, triples, perm, start, n, bern_prob, slots, n_slots, seed, offset, dev_cursor,
                                         margin, loss, (hipStream_t)stream);
}

int kge_train_pairwise_selfadv(const kge_model_desc* m, const int64_t* ph, const int64_t* pr, const int64_t* pt,
                               const int64_t* nh, const int64_t* nr, const int64_t* nt, int64_t n_pos,
                               int32_t neg_rate, float alpha, float* workspace, float* loss, void* stream) {
    if (validate(m, true, "kge_train_pairwise_selfadv")) return -1;
    if (n_pos == 0) return 0;
    if (n_pos < 0 || neg_rate <= 0 || !ph || !pr || !pt || !nh || !nr || !nt || !workspace || !loss) {
        set_error("kge_train_pairwise_selfadv: bad arguments");
        return -1;
    }
    if (!is_vector_model(m->model)) { set_error("kge_train_pairwise_selfadv: unsupported model %d", m->model); return -1; }
    hipStream_t s = (hipStream_t)stream;
    if (int rc = debug_check_hrt("kge_train_pairwise_selfadv (positives)", m, ph, pr, pt, n_pos, s)) return rc;
    if (int rc = debug_check_hrt("kge_train_pairwise_selfadv (negatives)", m, nh, nr, nt, n_pos * neg_rate, s)) return rc;
    {   // fused bundle kernel (one launch) whenever the negatives of a positive fit one lane group
        const int rc1 = launch_selfadv_bundle(m, ph, pr, pt, nh, nr, nt, n_pos, neg_rate, alpha, loss, s);
        if (rc1 <= 0) return rc1;
    }
    float* spos = workspace;
    float* sneg = workspace + n_pos;
    const int64_t n_neg = n_pos * neg_rate;
    int rc;
    if ((rc = launch_score_forward(m, ph, pr, pt, n_pos, spos, s))) return rc;
    if ((rc = launch_score_forward(m, nh, nr, nt, n_neg, sneg, s))) return rc;
    if ((rc = launch_selfadv_coeffs(spos, sneg, n_pos, neg_rate, alpha, loss, s))) return rc;
    if ((rc = launch_score_backward(m, ph, pr, pt, n_pos, spos, s))) return rc;
    return launch_score_backward(m, nh, nr, nt, n_neg, sneg, s);
}

int kge_train_pairwise_selfadv_sampled(const kge_model_desc* m, const int64_t* triples, const int64_t* perm, int64_t start,
                                       int64_t n_pos, int32_t neg_rate, float alpha, const float* bern_prob,
                                       const uint64_t* slots, int64_t n_slots, uint64_t seed, uint64_t offset,
                                       const int64_t* dev_cursor, float* loss, void* stream) {
    if (validate(m, true, "kge_train_pairwise_selfadv_sampled")) return -1;
    if (validate_packed_key(m, "kge_train_pairwise_selfadv_sampled")) return -1;
    if (n_pos == 0) return 0;
    if (n_pos < 0 || start < 0 || neg_rate <= 0 || !triples || !perm || !loss) {
        set_error("kge_train_pairwise_selfadv_sampled: bad arguments");
        return -1;
    }
    if (slots && (n_slots & (n_slots - 1))) { set_error("kge_train_pairwise_selfadv_sampled: n_slots must be a power of two"); return -1; }
    if (!dev_cursor)
        if (int rc = debug_check_triples("kge_train_pairwise_selfadv_sampled", m->tot_entity, m->tot_relation, triples, n_pos, (hipStream_t)stream, perm, start)) return rc;
    return launch_rotate_bundle_sampled(m, triples, perm, start, n_pos, neg_rate, alpha, bern_prob, slots, n_slots, seed,
                                        offset, dev_cursor, loss, nullptr, nullptr, (hipStream_t)stream);
}

static int staged_sink(const kge_model_desc* m, const kge_staged_step* st, int64_t n_pos, int32_t neg_rate, int ns, int nd,
                       const char* who, hipStream_t s, StageSink* sink) {
    if (!st->stage || !st->dyn_count || !st->dyn_bucket || !st->dyn_head || !st->dyn_next || st->dyn_cap <= 0 ||
        st->static_slots != ns || st->dynamic_slots != nd || st->stage_stride < m->dim || st->n_pos != n_pos ||
        st->n_neg != n_pos * neg_rate || st->tot_entity != m->tot_entity) {
        set_error("%s: staging plan does not match the batch (%d static + %d dynamic slots)", who, ns, nd);
        return -1;
    }
    if (!st->dyn_count_next) {
        // single registration set: clear it here (otherwise the previous optimiser sweep did).  An empty set is all zeros
        // (head holds pair + 1), so count | head laid out back to back clear with one memset.
        const size_t E = (size_t)m->tot_entity;
        hipError_t e;
        if (st->dyn_head == st->dyn_count + E) {
            e = hipMemsetAsync(st->dyn_count, 0, 2 * E * sizeof(int32_t), s);
        } else {
            e = hipMemsetAsync(st->dyn_count, 0, E * sizeof(int32_t), s);
            if (e == hipSuccess) e = hipMemsetAsync(st->dyn_head, 0, E * sizeof(int32_t), s);
        }
        if (e != hipSuccess) { set_error("%s: memset: %s", who, hipGetErrorString(e)); return -2; }
    } else if (st->dyn_list) {
        set_error("%s: touched-row lists need a single registration set (dyn_count_next == NULL)", who);
        return -1;
    }
    sink->stage = st->stage; sink->stride = st->stage_stride;
    sink->count = st->dyn_count; sink->bucket = st->dyn_bucket; sink->head = st->dyn_head; sink->next = st->dyn_next;
    sink->cap = st->dyn_cap; sink->ns = st->static_slots; sink->nd = st->dynamic_slots;
    sink->dyn_list = st->dyn_list;
    sink->spare = st->stage_spare;
    return 0;
}

size_t kge_staged_step_bytes(void) { return sizeof(kge_staged_step); }

int kge_train_pairwise_selfadv_sampled_staged(const kge_model_desc* m, const int64_t* triples, const int64_t* perm,
                                              int64_t start, int64_t n_pos, int32_t neg_rate, float alpha,
                                              const float* bern_prob, const uint64_t* slots, int64_t n_slots, uint64_t seed,
                                              uint64_t offset, const kge_staged_step* st, float* loss, void* stream) {
    const char* who = "kge_train_pairwise_selfadv_sampled_staged";
    if (validate(m, false, who)) return -1;
    if (validate_packed_key(m, who)) return -1;
    if (n_pos == 0) return 0;
    if (n_pos < 0 || start < 0 || neg_rate <= 0 || !triples || !perm || !loss || !st) { set_error("%s: bad arguments", who); return -1; }
    if (slots && (n_slots & (n_slots - 1))) { set_error("%s: n_slots must be a power of two", who); return -1; }
    if (m->model != KGE_ROTATE) { set_error("%s: RotatE only", who); return -1; }
    hipStream_t s = (hipStream_t)stream;
    if (int rc = debug_check_triples(who, m->tot_entity, m->tot_relation, triples, n_pos, s, perm, start)) return rc;
    StageSink sink;
    const int rc = staged_sink(m, st, n_pos, neg_rate, 5, 2, who, s, &sink);
    if (rc) return rc;
    if (!st->dyn_scale) { set_error("%s: the single-pass kernel needs dyn_scale (one float per negative pair)", who); return -1; }
    return launch_rotate_bundle_sampled(m, triples, perm, start, n_pos, neg_rate, alpha, bern_prob, slots, n_slots, seed,
                                        offset, nullptr, loss, &sink, st->dyn_scale, s);
}

int kge_train_pointwise_logistic_sampled_staged(const kge_model_desc* m, const int64_t* triples, const int64_t* perm,
                                                int64_t start, int64_t n_pos, int32_t neg_rate, const float* bern_prob,
                                                const uint64_t* slots, int64_t n_slots, uint64_t seed, uint64_t offset,
                                                float lmbda, int32_t reg_type, const kge_staged_step* st, float* loss,
                                                void* stream) {
    const char* who = "kge_train_pointwise_logistic_sampled_staged";
    if (validate(m, false, who)) return -1;
    if (validate_packed_key(m, who)) return -1;
    if (n_pos == 0) return 0;
    if (n_pos < 0 || neg_rate < 1 || start < 0 || !triples || !perm || !loss || !st) { set_error("%s: bad arguments", who); return -1; }
    if (slots && (n_slots & (n_slots - 1))) { set_error("%s: n_slots must be a power of two", who); return -1; }
    if (reg_type < KGE_REG_NONE || reg_type > KGE_REG_ID_N3) { set_error("%s: bad reg_type %d", who, reg_type); return -1; }
    if (m->model != KGE_DISTMULT && m->model != KGE_COMPLEX) { set_error("%s: DistMult / ComplEx only", who); return -1; }
    if (int rc = debug_check_triples(who, m->tot_entity, m->tot_relation, triples, n_pos, (hipStream_t)stream, perm, start)) return rc;
    StageSink sink;
    const int rc = staged_sink(m, st, n_pos, neg_rate, m->model == KGE_COMPLEX ? 6 : 3, m->model == KGE_COMPLEX ? 2 : 1, who,
                               (hipStream_t)stream, &sink);
    if (rc) return rc;
    return launch_pointwise_logistic_sampled_staged(m, triples, perm, start, n_pos, neg_rate, bern_prob, slots, n_slots, seed,
                                                    offset, lmbda, reg_type, loss, sink, (hipStream_t)stream);
}

int kge_optimizer_step_staged(int32_t optimizer, const kge_staged_step* st, float lr, int64_t step, void* stream) {
    if (step < 1) { set_error("kge_optimizer_step_staged: step counts from 1"); return -1; }
    return launch_optimizer_staged(optimizer, st, lr, step, (hipStream_t)stream);
}

int kge_train_pointwise_logistic(const kge_model_desc* m, const int64_t* h, const int64_t* r, const int64_t* t,
                                 const int64_t* y, int64_t n, int32_t bundle, float lmbda, int32_t reg_type, float* loss,
                                 void* stream) {
    if (validate(m, true, "kge_train_pointwise_logistic")) return -1;
    if (n == 0) return 0;
    if (n < 0 || !h || !r || !t || !y || !loss) { set_error("kge_train_pointwise_logistic: bad arguments"); return -1; }
    if (reg_type < KGE_REG_NONE || reg_type > KGE_REG_ID_N3) { set_error("kge_train_pointwise_logistic: bad reg_type %d", reg_type); return -1; }
    if (!is_vector_model(m->model)) { set_error("kge_train_pointwise_logistic: unsupported model %d", m->model); return -1; }
    if (int rc = debug_check_hrt("kge_train_pointwise_logistic", m, h, r, t, n, (hipStream_t)stream)) return rc;
    return launch_pointwise_logistic(m, h, r, t, y, n, bundle, lmbda, reg_type, loss, nullptr, (hipStream_t)stream);
}

int kge_train_pointwise_logistic_sampled(const kge_model_desc* m, const int64_t* triples, const int64_t* perm, int64_t start,
                                         int64_t n_pos, int32_t neg_rate, const float* bern_prob, const uint64_t* slots,
                                         int64_t n_slots, uint64_t seed, uint64_t offset, const int64_t* dev_cursor,
                                         float lmbda, int32_t reg_type, float* loss, void* stream) {
    if (validate(m, true, "kge_train_pointwise_logistic_sampled")) return -1;
    if (validate_packed_key(m, "kge_train_pointwise_logistic_sampled")) return -1;
    if (n_pos == 0) return 0;
    if (n_pos < 0 || neg_rate < 1 || start < 0 || !triples || !perm || !loss) { set_error("kge_train_pointwise_logistic_sampled: bad arguments"); return -1; }
    if (slots && (n_slots & (n_slots - 1))) { set_error("kge_train_pointwise_logistic_sampled: n_slots must be a power of two"); return -1; }
    if (reg_type < KGE_REG_NONE || reg_type > KGE_REG_ID_N3) { set_error("kge_train_pointwise_logistic_sampled: bad reg_type %d", reg_type); return -1; }
    if (!is_vector_model(m->model)) { set_error("kge_train_pointwise_logistic_sampled: unsupported model %d", m->model); return -1; }
    if (!dev_cursor)
        if (int rc = debug_check_triples("kge_train_pointwise_logistic_sampled", m->tot_entity, m->tot_relation, triples, n_pos, (hipStream_t)stream, perm, start)) return rc;
    return launch_pointwise_logistic_sampled(m, triples, perm, start, n_pos, neg_rate, bern_prob, slots, n_slots, seed, offset,
                                             dev_cursor, lmbda, reg_type, loss, (hipStream_t)stream);
}

int kge_optimizer_step(int32_t kind, float* param, float* grad, float* state1, float* state2, int64_t numel, float lr,
                       int64_t step, int32_t zero_grad, const float* dev_hyper, void* stream) {
    if (!param || !grad || numel < 0 || (step < 1 && !dev_hyper)) { set_error("kge_optimizer_step: bad arguments"); return -1; }
    if (numel == 0) return 0;
    return launch_optimizer(kind, param, grad, state1, state2, numel, lr, step < 1 ? 1 : step, zero_grad, dev_hyper,
                            nullptr, nullptr, nullptr, 0, 1, 0, (hipStream_t)stream);
}

int kge_rescal_pair_step(const kge_model_desc* m, const int64_t* ph, const int64_t* pr, const int64_t* pt, const int64_t* nh,
                         const int64_t* nt, int64_t n, float margin, void* workspace, size_t workspace_bytes, float* loss,
                         uint32_t* touched_rows, void* stream) {
    if (validate(m, true, "kge_rescal_pair_step")) return -1;
    if (m->model != KGE_RESCAL) { set_error("kge_rescal_pair_step: not a RESCAL model"); return -1; }
    if (n == 0) return 0;
    if (n < 0 || !ph || !pr || !pt || !nh || !nt || !loss || !workspace) { set_error("kge_rescal_pair_step: bad arguments"); return -1; }
    if (!rescal_pair_step_ok(m, n, workspace_bytes)) {
        set_error("kge_rescal_pair_step: hidden size %d (needs an even size, at most 256) or workspace (kge_workspace_bytes)", m->dim);
        return -1;
    }
    if (int rc = debug_check_hrt("kge_rescal_pair_step (positives)", m, ph, pr, pt, n, (hipStream_t)stream)) return rc;
    if (int rc = debug_check_hrt("kge_rescal_pair_step (negatives)", m, nh, pr, nt, n, (hipStream_t)stream)) return rc;
    return launch_rescal_pair_step(m, ph, pr, pt, nh, nt, n, margin, loss, workspace, workspace_bytes, touched_rows, nullptr, (hipStream_t)stream);
}

int kge_rescal_stage_ok(const kge_model_desc* m, int64_t n) {
    return m && m->model == KGE_RESCAL && n >= 0 && rescal_pair_step_ok(m, n, (size_t)-1) && rescal_stage_ok(m, n, kge_workspace_bytes(m, n)) ? 1 : 0;
}

int kge_rescal_pair_step_staged(const kge_model_desc* m, const int64_t* ph, const int64_t* pr, const int64_t* pt, const int64_t* nh,
                                const int64_t* nt, int64_t n, float margin, void* workspace, size_t workspace_bytes, float* loss,
                                uint32_t* touched_rows, const kge_rescal_stage* stage, void* stream) {
    if (validate(m, true, "kge_rescal_pair_step_staged")) return -1;
    if (m->model != KGE_RESCAL) { set_error("kge_rescal_pair_step_staged: not a RESCAL model"); return -1; }
    if (n == 0) return 0;
    if (n < 0 || !ph || !pr || !pt || !nh || !nt || !loss || !workspace || !touched_rows || !stage || !stage->gstage || !stage->dsv ||
        !stage->count || !stage->bucket || !stage->head || !stage->next || stage->cap < 1 || stage->cap > 32) {
        set_error("kge_rescal_pair_step_staged: bad arguments (every buffer of the stage, 1 <= cap <= 32, and the touched-row bitmap are required)");
        return -1;
    }
    if (!rescal_pair_step_ok(m, n, workspace_bytes)) { set_error("kge_rescal_pair_step_staged: hidden size or workspace (kge_workspace_bytes)"); return -1; }
    if (int rc = debug_check_hrt("kge_rescal_pair_step_staged (positives)", m, ph, pr, pt, n, (hipStream_t)stream)) return rc;
    if (int rc = debug_check_hrt("kge_rescal_pair_step_staged (negatives)", m, nh, pr, nt, n, (hipStream_t)stream)) return rc;
    return launch_rescal_pair_step(m, ph, pr, pt, nh, nt, n, margin, loss, workspace, workspace_bytes, touched_rows, stage, (hipStream_t)stream);
}

int kge_rescal_pair_step_ok(const kge_model_desc* m, int64_t n) {
    return m && m->model == KGE_RESCAL && rescal_pair_step_ok(m, n, (size_t)-1) ? 1 : 0;
}

int kge_optimizer_step_rows(int32_t kind, float* param, float* grad, float* state1, float* state2, int64_t rows, int32_t dim,
                            float lr, int64_t step, int32_t zero_grad, int32_t normalize, const float* dev_hyper,
                            const uint32_t* touched_rows, uint32_t* touched_clear, void* stream) {
    if (!param || !grad || rows < 0 || dim <= 0 || dim > 1024 || (step < 1 && !dev_hyper)) { set_error("kge_optimizer_step_rows: bad arguments (rows of at most 1024 floats)"); return -1; }
    if (rows == 0) return 0;
    if (touched_rows && touched_rows == touched_clear) { set_error("kge_optimizer_step_rows: the bitmap to clear must be the other parity's"); return -1; }
    return launch_optimizer_rows(kind, param, grad, state1, state2, rows, dim, lr, step < 1 ? 1 : step, zero_grad, normalize, dev_hyper,
                                 touched_rows, touched_clear, nullptr, (hipStream_t)stream);
}

int kge_optimizer_step_rows_staged(int32_t kind, float* param, float* grad, float* state1, float* state2, int64_t rows, int32_t dim,
                                   float lr, int64_t step, int32_t normalize, const float* dev_hyper, const uint32_t* touched_rows,
                                   uint32_t* touched_clear, const kge_rescal_stage* stage, void* stream) {
    if (!param || !grad || rows < 0 || dim <= 0 || dim > 1024 || (step < 1 && !dev_hyper) || !stage || !touched_rows) {
        set_error("kge_optimizer_step_rows_staged: bad arguments (rows of at most 1024 floats, the stage and the touched-row bitmap are required)");
        return -1;
    }
    if (rows == 0) return 0;
    if (touched_rows == touched_clear) { set_error("kge_optimizer_step_rows_staged: the bitmap to clear must be the other parity's"); return -1; }
    return launch_optimizer_rows(kind, param, grad, state1, state2, rows, dim, lr, step < 1 ? 1 : step, 0, normalize, dev_hyper,
                                 touched_rows, touched_clear, stage, (hipStream_t)stream);
}

int kge_optimizer_step_rownorm_ok(int64_t rows, int64_t dim) {
    return optimizer_rownorm_ok(rows, dim, (size_t)-1) ? 1 : 0;
}

int kge_optimizer_step_rownorm(int32_t kind, float* param, float* grad, float* state1, float* state2, int64_t rows, int64_t dim, float lr,
                               int64_t step, int32_t zero_grad, const float* dev_hyper, void* scratch, size_t scratch_bytes,
                               const int64_t* dev_cursor, int64_t* next_cursor, float* next_hyper, int64_t batch_stride,
                               int64_t n_batches, int64_t draws_per_batch, void* stream) {
    if (!param || !grad || rows <= 0 || dim <= 0 || !scratch || (step < 1 && !dev_hyper)) { set_error("kge_optimizer_step_rownorm: bad arguments"); return -1; }
    if ((dev_cursor || next_cursor || next_hyper) && (!dev_cursor || !next_cursor || !next_hyper || !dev_hyper || dev_cursor == next_cursor || dev_hyper == next_hyper || n_batches < 1)) {
        set_error("kge_optimizer_step_rownorm: the step-state transition needs dev_hyper, dev_cursor and a different next set");
        return -1;
    }
    return launch_optimizer_rownorm(kind, param, grad, state1, state2, rows, dim, lr, step, zero_grad, dev_hyper, (float*)scratch,
                                    scratch_bytes / sizeof(float), dev_cursor, next_cursor, next_hyper, batch_stride, n_batches, draws_per_batch,
                                    (hipStream_t)stream);
}

int kge_optimizer_step_rows_rownorm(int32_t kind, float* param, float* grad, float* state1, float* state2, int64_t rows, int32_t dim,
                                    float* wparam, float* wgrad, float* wstate1, float* wstate2, int64_t wrows, int64_t wdim,
                                    float lr, int64_t step, int32_t zero_grad, int32_t normalize, const float* dev_hyper,
                                    const uint32_t* touched_rows, uint32_t* touched_clear, const kge_rescal_stage* stage,
                                    void* scratch, size_t scratch_bytes, const int64_t* dev_cursor, int64_t* next_cursor,
                                    float* next_hyper, int64_t batch_stride, int64_t n_batches, int64_t draws_per_batch, void* stream) {
    if (!param || !grad || rows <= 0 || dim <= 0 || dim > 1024 || !wparam || !wgrad || wrows <= 0 || wdim <= 0 || !scratch ||
        (step < 1 && !dev_hyper)) {
        set_error("kge_optimizer_step_rows_rownorm: bad arguments (short rows of at most 1024 floats, both tables and the scratch are required)");
        return -1;
    }
    if (stage && !touched_rows) { set_error("kge_optimizer_step_rows_rownorm: the staged form needs the touched-row bitmap"); return -1; }
    if (touched_rows && touched_rows == touched_clear) { set_error("kge_optimizer_step_rows_rownorm: the bitmap to clear must be the other parity's"); return -1; }
    if ((dev_cursor || next_cursor || next_hyper) && (!dev_cursor || !next_cursor || !next_hyper || !dev_hyper || dev_cursor == next_cursor || dev_hyper == next_hyper || n_batches < 1)) {
        set_error("kge_optimizer_step_rows_rownorm: the step-state transition needs dev_hyper, dev_cursor and a different next set");
        return -1;
    }
    return launch_optimizer_rows_rownorm(kind, param, grad, state1, state2, rows, dim, wparam, wgrad, wstate1, wstate2, wrows, wdim, lr, step,
                                         zero_grad, normalize, dev_hyper, touched_rows, touched_clear, stage, (float*)scratch,
                                         scratch_bytes / sizeof(float), dev_cursor, next_cursor, next_hyper, batch_stride, n_batches,
                                         draws_per_batch, (hipStream_t)stream);
}

int kge_optimizer_step_advance(int32_t kind, float* param, float* grad, float* state1, float* state2, int64_t numel, float lr,
                               int32_t zero_grad, const float* dev_hyper, const int64_t* dev_cursor, int64_t* next_cursor,
                               float* next_hyper, int64_t batch_stride, int64_t n_batches, int64_t draws_per_batch,
                               void* stream) {
    if (!param || !grad || numel <= 0 || !dev_hyper || !dev_cursor || !next_cursor || !next_hyper || n_batches < 1 ||
        dev_cursor == next_cursor || dev_hyper == next_hyper) {
        set_error("kge_optimizer_step_advance: bad arguments (the next-step state must be a different set)");
        return -1;
    }
    return launch_optimizer(kind, param, grad, state1, state2, numel, lr, 1, zero_grad, dev_hyper, dev_cursor, next_cursor,
                            next_hyper, batch_stride, n_batches, draws_per_batch, (hipStream_t)stream);
}

/* ---- owner-computes ("pull") training step, kge_pull.hip */
int kge_pull_partial_stride(int32_t dim) { return pull_partial_stride(dim); }
int kge_pull_hat_stride(int32_t dim) { return pull_hat_stride(dim); }
int kge_pull_groups_per_block(int32_t dim) { return pull_groups_per_block(dim); }

int kge_row_norms(const float* table, int64_t rows, int32_t dim, float* norms, float* normalised, void* stream) {
    if (!table || !norms || rows < 0 || dim <= 0) { set_error("kge_row_norms: bad arguments"); return -1; }
    return launch_row_norms(table, rows, dim, norms, normalised, (hipStream_t)stream);
}

static int lists_ok(const kge_pull_lists* l) { return l && l->pc && l->count && l->bucket && l->head && l->next && l->sdesc && l->dbucket; }

int kge_pull_sample(const int32_t* pairs, const int32_t* inv, int64_t n, int64_t tot_entity, const float* bern_prob,
                    const uint64_t* slots, int64_t n_slots, uint64_t seed, uint64_t offset, const int64_t* dev_cursor,
                    const kge_pull_lists* out, void* stream) {
    if (n == 0) return 0;
    if (n < 0 || !pairs || !inv || !lists_ok(out) || tot_entity <= 0) { set_error("kge_pull_sample: bad arguments"); return -1; }
    if (tot_entity > (1 << 24)) { set_error("kge_pull_sample: more than 2^24 entities not supported by the packed key"); return -1; }
    if (slots && (n_slots & (n_slots - 1))) { set_error("kge_pull_sample: n_slots must be a power of two"); return -1; }
    return launch_pull_sample(pairs, inv, n, tot_entity, bern_prob, slots, n_slots, seed, offset, dev_cursor, out, (hipStream_t)stream);
}

int kge_pull_lists_explicit(const int32_t* pairs, const int32_t* inv, const int64_t* nh, const int64_t* nt, int64_t n,
                            const kge_pull_lists* out, void* stream) {
    if (n == 0) return 0;
    if (n < 0 || !pairs || !inv || !nh || !nt || !lists_ok(out)) { set_error("kge_pull_lists_explicit: bad arguments"); return -1; }
    return launch_pull_lists_explicit(pairs, inv, nh, nt, n, out, (hipStream_t)stream);
}

int kge_pull_step(const kge_model_desc* m, float* const tables_out[2], const float* const hat_in[2], float* const hat_out[2],
                  const float* norm_in, float* norm_out,
                  float* const state1[2], float* const state2[2], const int32_t* pairs, const kge_pull_lists* lists,
                  const int32_t* items, int64_t n_items, const uint32_t* dense_skip, const int32_t* inc, float* partials, const int32_t* multi,
                  int64_t n_multi, float margin, int32_t optimizer, float lr, int64_t step, const float* dev_hyper,
                  int32_t reset_lists, const int32_t* next_pairs, const int32_t* next_inv, int64_t next_n, const float* bern_prob,
                  const uint64_t* slots, int64_t n_slots, uint64_t seed, uint64_t next_offset, const kge_pull_lists* next_lists,
                  float* loss, const kge_pull_direction* direction, void* stream) {
    if (validate(m, false, "kge_pull_step")) return -1;
    if (direction && (!direction->codes != !direction->recs || (direction->codes && direction->n_pairs <= 0))) {
        set_error("kge_pull_step: the two-phase form needs both scratch buffers and the batch's pair count");
        return -1;
    }
    if (direction && !direction->codes) direction = nullptr;
    if (m->model != KGE_TRANSE && m->model != KGE_TRANSM) { set_error("kge_pull_step: TransE / TransM only (model %d)", m->model); return -1; }
    const bool grad_only = optimizer == KGE_OPT_GRADIENT;   // writes gradient rows: no normalised copies / norms / state out
    if (n_items <= 0 || n_multi < 0 || !tables_out || !tables_out[0] || !tables_out[1] || !hat_in || !hat_in[0] || !hat_in[1] ||
        !norm_in || !pairs || !lists_ok(lists) || !items || !inc || !loss || !partials || (n_multi > 0 && !multi) ||
        (!grad_only && (!hat_out || !hat_out[0] || !hat_out[1] || hat_in[0] == hat_out[0] || !norm_out))) {
        set_error("kge_pull_step: bad arguments");
        return -1;
    }
    float* const no_hat[2] = {nullptr, nullptr};
    if (grad_only) { hat_out = no_hat; }
    if (tables_out[0] == m->tables[0] || tables_out[1] == m->tables[1]) {
        set_error("kge_pull_step: the output tables must be the other half of the double buffer (rows are read by other owners)");
        return -1;
    }
    if (optimizer != KGE_OPT_SGD && !grad_only && (!state1 || !state1[0] || !state1[1])) { set_error("kge_pull_step: optimizer state missing"); return -1; }
    if (optimizer == KGE_OPT_ADAM && (!state2 || !state2[0] || !state2[1])) { set_error("kge_pull_step: adam needs two state buffers"); return -1; }
    if (next_pairs) {
        if (next_n < 0 || !next_inv || !lists_ok(next_lists) || next_lists->count == lists->count || (slots && (n_slots & (n_slots - 1)))) {
            set_error("kge_pull_step: the next batch's sampler needs its inverse incidence map and its own list set");
            return -1;
        }
        if (validate_packed_key(m, "kge_pull_step")) return -1;
    }
    return launch_pull_step(m, tables_out, hat_in, hat_out, norm_in, norm_out, state1, state2, pairs, lists, items, n_items, dense_skip, inc, partials, multi,
                            n_multi, margin, optimizer, lr, step, dev_hyper, reset_lists, next_pairs, next_inv, next_n, bern_prob, slots,
                            n_slots, seed, next_offset, next_lists, loss, direction, (hipStream_t)stream);
}

int kge_pull_direction_bytes(int32_t dim, int32_t l1, int64_t n_pairs, size_t* codes_bytes, size_t* recs_bytes) {
    if (!codes_bytes || !recs_bytes || n_pairs < 0) { set_error("kge_pull_direction_bytes: bad arguments"); return -1; }
    pull_direction_bytes(dim, l1, n_pairs, codes_bytes, recs_bytes);
    if (!*codes_bytes && n_pairs) { set_error("kge_pull_direction_bytes: hidden size %d must be a multiple of 4 and at most 1024", dim); return -1; }
    return 0;
}

size_t kge_pull_plan_bytes(void) { return sizeof(kge_pull_plan); }

int kge_pull_run(const kge_pull_plan* p, int64_t first_batch, int64_t n_steps, int32_t src_half, int32_t cur_list,
                 int32_t lists_ready, int64_t first_opt_step, uint64_t first_offset, int32_t sample_after_last, void* stream) {
    if (!p || !p->batches || n_steps < 0 || first_batch < 0 || first_batch + n_steps > p->n_batches || (src_half & ~1) ||
        (cur_list & ~1) || first_opt_step < 1) {
        set_error("kge_pull_run: bad arguments");
        return -1;
    }
    int src = src_half, cl = cur_list;
    uint64_t offset = first_offset;
    for (int64_t k = 0; k < n_steps; ++k) {
        const kge_pull_batch* b = p->batches + first_batch + k;
        int rc;
        if (k == 0 && !lists_ready) {
            rc = kge_pull_sample(b->pairs, b->inv, b->n_pairs, p->model[0].tot_entity, p->bern_prob, p->slots, p->n_slots, p->seed, offset,
                                 nullptr, &p->lists[cl], stream);
            if (rc) return rc;
        }
        // the sampler of the following batch rides in this step's launch: the next step of this run, or -- after the last step --
        // the batch that follows (sample_after_last == 1) or batch 0 of the next epoch (== 2: the same permutation is walked again,
        // the Philox counters just keep counting)
        const bool last = k + 1 == n_steps;
        const bool wrap = last && sample_after_last == 2;
        const bool has_next = wrap || ((!last || sample_after_last == 1) && first_batch + k + 1 < p->n_batches);
        const kge_pull_batch* nb = wrap ? p->batches : (has_next ? b + 1 : nullptr);
        float* const tables_out[2] = {const_cast<float*>(p->model[1 - src].tables[0]), const_cast<float*>(p->model[1 - src].tables[1])};
        const float* const hat_in[2] = {p->hat[src][0], p->hat[src][1]};
        float* const hat_out[2] = {p->hat[1 - src][0], p->hat[1 - src][1]};
        kge_pull_direction dir = p->direction;
        dir.n_pairs = b->n_pairs;
        dir.lists_without_descriptors = 1;
        rc = kge_pull_step(&p->model[src], tables_out, hat_in, hat_out, p->norm[src], p->norm[1 - src], p->state1, p->state2,
                           b->pairs, &p->lists[cl], b->items, b->n_items, b->dense_skip, b->inc, p->partials, b->multi, b->n_multi, p->margin,
                           p->optimizer, p->lr, first_opt_step + k, nullptr, 1, nb ? nb->pairs : nullptr, nb ? nb->inv : nullptr, nb ? nb->n_pairs : 0,
                           p->bern_prob, p->slots, p->n_slots, p->seed, offset + (uint64_t)p->draws_per_batch,
                           nb ? &p->lists[1 - cl] : nullptr, p->loss, dir.codes ? &dir : nullptr, stream);
        if (rc) return rc;
        src ^= 1;
        if (has_next) cl ^= 1;
        offset += (uint64_t)p->draws_per_batch;
    }
    return 0;
}

/* ---- TransH / TransD gradients in the two-launch owner-computes form, kge_pullx.hip */
int kge_transx_groups_per_block(int32_t dim) { return transx_groups_per_block(dim); }
int kge_transx_partial_stride(int32_t dim) { return transx_partial_stride(dim); }
int kge_transx_scratch_bytes(int32_t model, int32_t dim, int64_t n_pairs, size_t* stage_bytes, size_t* recs_bytes) {
    if (!stage_bytes || !recs_bytes || n_pairs < 0 || (model != KGE_TRANSH && model != KGE_TRANSD)) { set_error("kge_transx_scratch_bytes: bad arguments"); return -1; }
    transx_scratch_bytes(model, dim, n_pairs, stage_bytes, recs_bytes);
    if (!*stage_bytes && n_pairs) { set_error("kge_transx_scratch_bytes: hidden size %d must be a multiple of 4 and at most 512", dim); return -1; }
    return 0;
}
int kge_transx_grad_step(const kge_model_desc* m, const int32_t* pairs, int64_t n_pairs, const kge_pull_lists* lists,
                         const int32_t* items, int64_t n_items, const uint32_t* listed, const int32_t* inc, float* partials,
                         const int32_t* multi, int64_t n_multi, float margin, float* stage, float* recs, int32_t reset_lists,
                         const int32_t* next_pairs, const int32_t* next_inv, int64_t next_n, const float* bern_prob,
                         const uint64_t* slots, int64_t n_slots, uint64_t seed, uint64_t next_offset,
                         const kge_pull_lists* next_lists, float* loss, void* stream) {
    if (validate(m, true, "kge_transx_grad_step")) return -1;
    if (m->model != KGE_TRANSH && m->model != KGE_TRANSD) { set_error("kge_transx_grad_step: TransH / TransD only (model %d)", m->model); return -1; }
    const int nt = m->model == KGE_TRANSH ? 3 : 4;
    for (int k = 0; k < nt; ++k)
        if (!m->tables[k] || !m->grads[k]) { set_error("kge_transx_grad_step: table / gradient buffer %d missing", k); return -1; }
    if (n_pairs <= 0 || n_items < 0 || n_multi < 0 || !pairs || !lists_ok(lists) || (n_items > 0 && !items) || !inc || !loss || !partials ||
        !stage || !recs || (n_multi > 0 && !multi) || (n_items == 0 && !listed)) {
        set_error("kge_transx_grad_step: bad arguments");
        return -1;
    }
    if (next_pairs) {
        if (next_n < 0 || !next_inv || !lists_ok(next_lists) || next_lists->count == lists->count || (slots && (n_slots & (n_slots - 1)))) {
            set_error("kge_transx_grad_step: the next batch's sampler needs its inverse incidence map and its own list set");
            return -1;
        }
        if (validate_packed_key(m, "kge_transx_grad_step")) return -1;
    }
    return launch_transx_grad_step(m, pairs, n_pairs, lists, items, n_items, listed, inc, partials, multi, n_multi, margin, stage, recs,
                                   reset_lists, next_pairs, next_inv, next_n, bern_prob, slots, n_slots, seed, next_offset, next_lists, loss,
                                   (hipStream_t)stream);
}

size_t kge_transx_plan_bytes(void) { return sizeof(kge_transx_plan); }

int kge_transx_run(const kge_transx_plan* p, int64_t first_batch, int64_t n_steps, int32_t cur_list, int32_t lists_ready,
                   int64_t first_opt_step, uint64_t first_offset, int32_t sample_after_last, void* stream) {
    if (!p || !p->batches || n_steps < 0 || first_batch < 0 || first_batch + n_steps > p->n_batches || (cur_list & ~1) ||
        first_opt_step < 1 || !p->flat_param || !p->flat_grad || p->flat_numel <= 0) {
        set_error("kge_transx_run: bad arguments");
        return -1;
    }
    int cl = cur_list;
    uint64_t offset = first_offset;
    for (int64_t k = 0; k < n_steps; ++k) {
        const kge_pull_batch* b = p->batches + first_batch + k;
        int rc;
        if (k == 0 && !lists_ready) {
            rc = kge_pull_sample(b->pairs, b->inv, b->n_pairs, p->model.tot_entity, p->bern_prob, p->slots, p->n_slots, p->seed, offset,
                                 nullptr, &p->lists[cl], stream);
            if (rc) return rc;
        }
        const bool last = k + 1 == n_steps;      // (sample_after_last: as kge_pull_run)
        const bool wrap = last && sample_after_last == 2;
        const bool has_next = wrap || ((!last || sample_after_last == 1) && first_batch + k + 1 < p->n_batches);
        const kge_pull_batch* nb = wrap ? p->batches : (has_next ? b + 1 : nullptr);
        rc = kge_transx_grad_step(&p->model, b->pairs, b->n_pairs, &p->lists[cl], b->items, b->n_items, b->dense_skip, b->inc, p->partials,
                                  b->multi, b->n_multi, p->margin, p->stage, p->recs, 1, nb ? nb->pairs : nullptr, nb ? nb->inv : nullptr,
                                  nb ? nb->n_pairs : 0, p->bern_prob, p->slots, p->n_slots, p->seed, offset + (uint64_t)p->draws_per_batch,
                                  nb ? &p->lists[1 - cl] : nullptr, p->loss, stream);
        if (rc) return rc;
        rc = kge_optimizer_step(p->optimizer, p->flat_param, p->flat_grad, p->flat_state1, p->flat_state2, p->flat_numel, p->lr,
                                first_opt_step + k, 1, nullptr, stream);
        if (rc) return rc;
        if (has_next) cl ^= 1;
        offset += (uint64_t)p->draws_per_batch;
    }
    return 0;
}

/* ---- two-phase owner-computes step of the pointwise models, kge_own.hip */
int kge_own_groups_per_block(int32_t model, int32_t dim) { return ownx_model(model) ? ownx_groups_per_block(model, dim) : own_groups_per_block(model, dim); }
int kge_own_partial_stride(int32_t model, int32_t dim) { return ownx_model(model) ? ownx_partial_stride(model, dim) : own_partial_stride(model, dim); }
size_t kge_own_stage_bytes(int32_t model, int32_t dim, int64_t n_pairs) {
    if (n_pairs < 0) return 0;
    if (ownx_model(model)) return ownx_stage_floats(model, dim, n_pairs) * sizeof(float);
    return (size_t)4 * (size_t)n_pairs * (size_t)own_partial_stride(model, dim) * sizeof(float);
}

int kge_own_step(const kge_model_desc* m, const int32_t* pairs, int64_t n_pairs, const kge_pull_lists* lists, const int32_t* items,
                 int64_t n_items, const uint32_t* listed, const int32_t* inc, float* partials, int32_t dense, float lmbda,
                 int32_t reg_type, int32_t reset_lists, const int32_t* next_pairs, const int32_t* next_inv, int64_t next_n,
                 const float* bern_prob, const uint64_t* slots, int64_t n_slots, uint64_t seed, uint64_t next_offset,
                 const kge_pull_lists* next_lists, float* loss, float* stage, void* stream) {
    if (validate(m, true, "kge_own_step")) return -1;
    if (m->model != KGE_DISTMULT && m->model != KGE_COMPLEX) { set_error("kge_own_step: DistMult / ComplEx only (model %d)", m->model); return -1; }
    if (n_pairs <= 0 || n_items <= 0 || !pairs || !lists_ok(lists) || !items || !inc || !partials || !loss ||
        reg_type < KGE_REG_NONE || reg_type > KGE_REG_N3_ABS) {
        set_error("kge_own_step: bad arguments");
        return -1;
    }
    if (next_pairs) {
        if (next_n < 0 || !next_inv || !lists_ok(next_lists) || next_lists->count == lists->count || (slots && (n_slots & (n_slots - 1)))) {
            set_error("kge_own_step: the next batch's sampler needs its inverse incidence map and its own list set");
            return -1;
        }
        if (validate_packed_key(m, "kge_own_step")) return -1;
    }
    return launch_own_step(m, pairs, n_pairs, lists, items, n_items, listed, inc, partials, dense, lmbda, reg_type, reset_lists,
                           next_pairs, next_inv, next_n, bern_prob, slots, n_slots, seed, next_offset, next_lists, loss, stage, (hipStream_t)stream);
}

int kge_own_apply(const kge_model_desc* m, float* const* state1, float* const* state2, const int32_t* pairs, int64_t n_pairs,
                  const kge_pull_lists* lists, const int32_t* items, int64_t n_items, const uint32_t* listed, const int32_t* multi,
                  int64_t n_multi, float* partials, int32_t dense, int32_t optimizer, float lr, int64_t step, void* stream) {
    if (validate(m, true, "kge_own_apply")) return -1;
    if (m->model != KGE_DISTMULT && m->model != KGE_COMPLEX) { set_error("kge_own_apply: DistMult / ComplEx only (model %d)", m->model); return -1; }
    if (n_pairs <= 0 || n_items <= 0 || n_multi < 0 || !pairs || !lists || !lists->pc || !items || (n_multi > 0 && (!multi || !partials)) || step < 1) {
        set_error("kge_own_apply: bad arguments");
        return -1;
    }
    return launch_own_apply(m, state1, state2, pairs, n_pairs, lists, items, n_items, listed, multi, n_multi, partials, dense,
                            optimizer, lr, step, 0, (hipStream_t)stream);
}

size_t kge_own_plan_bytes(void) { return sizeof(kge_own_plan); }

int kge_own_run(const kge_own_plan* p, int64_t first_batch, int64_t n_steps, int32_t cur_list, int32_t lists_ready,
                int64_t first_opt_step, uint64_t first_offset, int32_t sample_after_last, void* stream) {
    if (!p || !p->batches || n_steps < 0 || first_batch < 0 || first_batch + n_steps > p->n_batches || (cur_list & ~1) ||
        first_opt_step < 1) {
        set_error("kge_own_run: bad arguments");
        return -1;
    }
    const int dense = (p->optimizer == KGE_OPT_ADAM || p->optimizer == KGE_OPT_RMSPROP) ? 1 : 0;
    int cl = cur_list;
    uint64_t offset = first_offset;
    for (int64_t k = 0; k < n_steps; ++k) {
        const kge_pull_batch* b = p->batches + first_batch + k;
        int rc;
        if (k == 0 && !lists_ready) {
            rc = kge_pull_sample(b->pairs, b->inv, b->n_pairs, p->model.tot_entity, p->bern_prob, p->slots, p->n_slots, p->seed, offset,
                                 nullptr, &p->lists[cl], stream);
            if (rc) return rc;
        }
        const bool last = k + 1 == n_steps;      // (sample_after_last: as kge_pull_run -- 1 = the following batch, 2 = batch 0 of the next epoch)
        const bool wrap = last && sample_after_last == 2;
        const bool has_next = wrap || ((!last || sample_after_last == 1) && first_batch + k + 1 < p->n_batches);
        const kge_pull_batch* nb = wrap ? p->batches : (has_next ? b + 1 : nullptr);
        if (ownx_model(p->model.model)) {
            // ANALOGY / CP / SimplE / QuatE: the model-generic staged step (kge_ownx.hip)
            rc = launch_ownx_step(&p->model, p->state1, p->state2, b->pairs, b->n_pairs, &p->lists[cl], b->items, b->n_items, b->dense_skip,
                                  b->inc, p->partials, b->multi, b->n_multi, dense, p->lmbda, p->reg_type, p->optimizer, p->lr,
                                  first_opt_step + k, nb ? nb->pairs : nullptr, nb ? nb->inv : nullptr, nb ? nb->n_pairs : 0, p->bern_prob,
                                  p->slots, p->n_slots, p->seed, offset + (uint64_t)p->draws_per_batch, nb ? &p->lists[1 - cl] : nullptr,
                                  p->loss, p->stage, (hipStream_t)stream);
            if (rc) return rc;
        } else if (p->stage) {
            // staged form: the owners apply the optimiser themselves; only rows cut across workgroups go through k_own_apply
            rc = launch_own_step_fused(&p->model, p->state1, p->state2, b->pairs, b->n_pairs, &p->lists[cl], b->items, b->n_items,
                                       b->dense_skip, b->inc, p->partials, dense, p->lmbda, p->reg_type, p->optimizer, p->lr,
                                       first_opt_step + k, nb ? nb->pairs : nullptr, nb ? nb->inv : nullptr, nb ? nb->n_pairs : 0,
                                       p->bern_prob, p->slots, p->n_slots, p->seed, offset + (uint64_t)p->draws_per_batch,
                                       nb ? &p->lists[1 - cl] : nullptr, p->loss, p->stage, (hipStream_t)stream);
            if (rc) return rc;
            if (b->n_multi > 0) {
                rc = launch_own_apply(&p->model, p->state1, p->state2, b->pairs, b->n_pairs, &p->lists[cl], b->items, b->n_items,
                                      b->dense_skip, b->multi, b->n_multi, p->partials, dense, p->optimizer, p->lr, first_opt_step + k, 1,
                                      (hipStream_t)stream);
                if (rc) return rc;
            }
        } else {
            rc = kge_own_step(&p->model, b->pairs, b->n_pairs, &p->lists[cl], b->items, b->n_items, b->dense_skip, b->inc, p->partials,
                              dense, p->lmbda, p->reg_type, 1, nb ? nb->pairs : nullptr, nb ? nb->inv : nullptr, nb ? nb->n_pairs : 0, p->bern_prob, p->slots,
                              p->n_slots, p->seed, offset + (uint64_t)p->draws_per_batch, nb ? &p->lists[1 - cl] : nullptr, p->loss, nullptr, stream);
            if (rc) return rc;
            rc = kge_own_apply(&p->model, p->state1, p->state2, b->pairs, b->n_pairs, &p->lists[cl], b->items, b->n_items, b->dense_skip,
                               b->multi, b->n_multi, p->partials, dense, p->optimizer, p->lr, first_opt_step + k, stream);
            if (rc) return rc;
        }
        if (has_next) cl ^= 1;
        offset += (uint64_t)p->draws_per_batch;
    }
    return 0;
}

int kge_head_1n_forward(const float* x, int64_t batch, int32_t dim, const float* ent, int64_t tot_entity, const float* bias,
                        float* preds, void* stream) {
    return launch_head_forward(x, batch, dim, ent, tot_entity, bias, preds, 0, (hipStream_t)stream);
}

int kge_head_1n_forward_bf16(const float* x, int64_t batch, int32_t dim, const float* ent, int64_t tot_entity, const float* bias,
                             float* preds, void* stream) {
    return launch_head_forward(x, batch, dim, ent, tot_entity, bias, preds, 1, (hipStream_t)stream);
}

size_t kge_head_1n_rank_workspace_bytes(int64_t batch, int32_t dim, int64_t tot_entity, int32_t has_bias) {
    if (batch < 0 || dim <= 0 || tot_entity <= 0) return 0;
    return head_rank_workspace_bytes(batch, dim, tot_entity, has_bias != 0);
}

int kge_head_1n_rank(const float* x, int64_t batch, int32_t dim, const float* ent, int64_t tot_entity, const float* bias,
                     const int64_t* triples, const int64_t* off, const int32_t* ids, void* workspace, size_t workspace_bytes,
                     int32_t* ranks, int32_t* ties, float* energies, void* stream) {
    if (batch == 0) return 0;
    if (batch < 0 || dim <= 0 || tot_entity <= 0 || !x || !ent || !triples || (!ranks && !energies) || ((off == nullptr) != (ids == nullptr))) {
        set_error("kge_head_1n_rank: bad arguments");
        return -1;
    }
    if (tot_entity >= (1ll << 31)) { set_error("kge_head_1n_rank: entity ids must fit 31 bits"); return -1; }
    if (int rc = debug_check_ids("kge_head_1n_rank", "true entity", triples, batch, 3, 2, tot_entity, (hipStream_t)stream)) return rc;
    return launch_head_rank(x, batch, dim, ent, tot_entity, bias, triples, off, ids, workspace, workspace_bytes, ranks, ties, energies,
                            (hipStream_t)stream);
}

size_t kge_head_1n_backward_workspace_bytes(void) { return head_backward_workspace_bytes(); }

int kge_head_1n_backward(const float* x, int64_t batch, int32_t dim, const float* ent, int64_t tot_entity, const float* preds,
                         const float* dpreds, float* dx, float* g_ent, float* g_bias, void* workspace, size_t workspace_bytes,
                         void* stream) {
    return launch_head_backward(x, batch, dim, ent, tot_entity, preds, dpreds, dx, g_ent, g_bias, workspace, workspace_bytes,
                                (hipStream_t)stream);
}

size_t kge_head_1n_bce_workspace_bytes(int64_t batch, int64_t tot_entity, int64_t n_pos) {
    if (batch <= 0 || tot_entity <= 0 || n_pos < 0) return 0;
    return head_bce_workspace_bytes(batch, tot_entity, n_pos);
}

int kge_head_1n_bce(const float* x, int64_t batch, int32_t dim, const float* ent, int64_t tot_entity, const float* bias,
                    const int64_t* label_off, const int32_t* label_ids, int64_t n_pos, float label_smoothing,
                    void* workspace, size_t workspace_bytes, float* loss, float* dx, float* g_ent, float* g_bias,
                    void* stream) {
    return launch_head_bce(x, batch, dim, ent, tot_entity, bias, label_off, label_ids, n_pos, label_smoothing, workspace,
                           workspace_bytes, loss, dx, g_ent, g_bias, (hipStream_t)stream);
}

int kge_l2norm_reg(const float* param, float* grad, int64_t numel, float lmbda, float* scratch, float* loss, void* stream) {
    if (!param || !grad || !scratch || !loss || numel <= 0) { set_error("kge_l2norm_reg: bad arguments"); return -1; }
    return launch_l2norm_reg(param, grad, numel, lmbda, scratch, loss, (hipStream_t)stream);
}

size_t kge_eval_workspace_bytes(const kge_model_desc* m, int64_t n) {
    if (validate(m, false, "kge_eval_workspace_bytes") || n < 0) return 0;
    return eval_workspace_bytes(m, n);
}

int kge_eval_ranks_ties(const kge_model_desc* m, const int64_t* triples, int64_t n, const int64_t* tail_off,
                        const int32_t* tail_ids, const int64_t* head_off, const int32_t* head_ids, void* workspace,
                        size_t workspace_bytes, int32_t* ranks, int32_t* ties, void* stream) {
    if (validate(m, false, "kge_eval_ranks")) return -1;
    if (n == 0) return 0;
    if (n < 0 || !triples || !ranks) { set_error("kge_eval_ranks: bad arguments"); return -1; }
    if ((tail_off && !tail_ids) || (head_off && !head_ids)) { set_error("kge_eval_ranks: CSR offsets without ids"); return -1; }
    if (int rc = debug_check_triples("kge_eval_ranks", m->tot_entity, m->tot_relation, triples, n, (hipStream_t)stream)) return rc;
    return launch_eval_ranks(m, triples, n, tail_off, tail_ids, head_off, head_ids, workspace, workspace_bytes, ranks, ties,
                             (hipStream_t)stream);
}

int kge_eval_ranks(const kge_model_desc* m, const int64_t* triples, int64_t n, const int64_t* tail_off,
                   const int32_t* tail_ids, const int64_t* head_off, const int32_t* head_ids, void* workspace,
                   size_t workspace_bytes, int32_t* ranks, void* stream) {
    return kge_eval_ranks_ties(m, triples, n, tail_off, tail_ids, head_off, head_ids, workspace, workspace_bytes, ranks, nullptr, stream);
}

size_t kge_eval_grouped_workspace_bytes(const kge_model_desc* m, int64_t n, int64_t n_groups) {
    if (validate(m, false, "kge_eval_grouped_workspace_bytes") || n < 0 || n_groups < 1) return 0;
    return eval_workspace_bytes(m, n, n_groups == 1 ? 2 : n_groups);
}

int kge_eval_ranks_grouped_ties(const kge_model_desc* m, const int64_t* triples, int64_t n, const int32_t* group_of_triple,
                                const int64_t* group_rel, int64_t n_groups, const int32_t* qblocks, int64_t n_qblocks,
                                const int64_t* tail_off, const int32_t* tail_ids, const int64_t* head_off,
                                const int32_t* head_ids, void* workspace, size_t workspace_bytes, int32_t* ranks, int32_t* ties,
                                void* stream) {
    if (validate(m, false, "kge_eval_ranks_grouped")) return -1;
    if (n == 0) return 0;
    if (n < 0 || !triples || !ranks || !group_of_triple || !group_rel || !qblocks || n_groups < 1 || n_qblocks < 1) {
        set_error("kge_eval_ranks_grouped: bad arguments");
        return -1;
    }
    if ((tail_off && !tail_ids) || (head_off && !head_ids)) { set_error("kge_eval_ranks_grouped: CSR offsets without ids"); return -1; }
    if (int rc = debug_check_triples("kge_eval_ranks_grouped", m->tot_entity, m->tot_relation, triples, n, (hipStream_t)stream)) return rc;
    return launch_eval_ranks_grouped(m, triples, n, group_of_triple, group_rel, n_groups, qblocks, n_qblocks, tail_off, tail_ids,
                                     head_off, head_ids, workspace, workspace_bytes, ranks, ties, (hipStream_t)stream);
}

int kge_eval_ranks_grouped(const kge_model_desc* m, const int64_t* triples, int64_t n, const int32_t* group_of_triple,
                           const int64_t* group_rel, int64_t n_groups, const int32_t* qblocks, int64_t n_qblocks,
                           const int64_t* tail_off, const int32_t* tail_ids, const int64_t* head_off,
                           const int32_t* head_ids, void* workspace, size_t workspace_bytes, int32_t* ranks, void* stream) {
    return kge_eval_ranks_grouped_ties(m, triples, n, group_of_triple, group_rel, n_groups, qblocks, n_qblocks, tail_off, tail_ids,
                                       head_off, head_ids, workspace, workspace_bytes, ranks, nullptr, stream);
}

int kge_eval_sweep_scores(const kge_model_desc* m, const int64_t* triples, int64_t n, void* workspace,
                          size_t workspace_bytes, float* scores, void* stream) {
    if (validate(m, false, "kge_eval_sweep_scores")) return -1;
    if (n == 0) return 0;
    if (n < 0 || !triples || !scores) { set_error("kge_eval_sweep_scores: bad arguments"); return -1; }
    if (int rc = debug_check_triples("kge_eval_sweep_scores", m->tot_entity, m->tot_relation, triples, n, (hipStream_t)stream)) return rc;
    return launch_eval_sweep_scores(m, triples, n, workspace, workspace_bytes, scores, (hipStream_t)stream);
}

int kge_eval_sweep_scores_side(const kge_model_desc* m, const int64_t* triples, int64_t n, int side, void* workspace,
                               size_t workspace_bytes, float* scores, void* stream) {
    if (validate(m, false, "kge_eval_sweep_scores_side")) return -1;
    if (n == 0) return 0;
    if (n < 0 || !triples || !scores || (side != 0 && side != 1)) { set_error("kge_eval_sweep_scores_side: bad arguments (side is 0 = tail sweep or 1 = head sweep)"); return -1; }
    // (the unused column still passes through the id check: callers put any valid id there)
    if (int rc = debug_check_triples("kge_eval_sweep_scores_side", m->tot_entity, m->tot_relation, triples, n, (hipStream_t)stream)) return rc;
    return launch_eval_sweep_scores(m, triples, n, workspace, workspace_bytes, scores, (hipStream_t)stream, side);
}

int kge_rank_from_scores(const float* scores, int64_t nq, int64_t tot_entity, const int64_t* truth, const int64_t* off,
                         const int32_t* ids, int32_t* rank, int32_t* frank, void* stream) {
    if (nq == 0) return 0;
    if (!scores || !truth || !rank || !frank || nq < 0 || tot_entity <= 0 || (off && !ids)) {
        set_error("kge_rank_from_scores: bad arguments");
        return -1;
    }
    if (int rc = debug_check_ids("kge_rank_from_scores", "true entity", truth, nq, 1, 0, tot_entity, (hipStream_t)stream)) return rc;
    return launch_rank_from_scores(scores, nq, tot_entity, truth, off, ids, rank, frank, (hipStream_t)stream);
}

int kge_triple_set_build(const int64_t* triples, int64_t n, uint64_t* slots, int64_t n_slots, void* stream) {
    if (!triples || !slots || n < 0 || n_slots < 2 * n || (n_slots & (n_slots - 1))) {
        set_error("kge_triple_set_build: n_slots must be a power of two >= 2n");
        return -1;
    }
    // (the packed key h:24 | r:16 | t:24 is the only bound this entry point knows)
    if (int rc = debug_check_triples("kge_triple_set_build", (int64_t)1 << 24, (int64_t)1 << 16, triples, n, (hipStream_t)stream)) return rc;
    return launch_triple_set_build(triples, n, slots, n_slots, (hipStream_t)stream);
}

int kge_corrupt(const int64_t* ph, const int64_t* pr, const int64_t* pt, int64_t n_pos, int32_t neg_rate,
                int64_t tot_entity, const float* bern_prob, const uint64_t* slots, int64_t n_slots, uint64_t seed,
                uint64_t offset, int64_t* nh, int64_t* nr, int64_t* nt, void* stream) {
    if (n_pos == 0) return 0;
    if (!ph || !pr || !pt || !nh || !nr || !nt || n_pos < 0 || neg_rate <= 0 || tot_entity <= 1) {
        set_error("kge_corrupt: bad arguments");
        return -1;
    }
    if (slots && (n_slots & (n_slots - 1))) { set_error("kge_corrupt: n_slots must be a power of two"); return -1; }
    if (int rc = debug_check_ids("kge_corrupt", "head", ph, n_pos, 1, 0, tot_entity, (hipStream_t)stream)) return rc;
    if (int rc = debug_check_ids("kge_corrupt", "tail", pt, n_pos, 1, 0, tot_entity, (hipStream_t)stream)) return rc;
    return launch_corrupt(ph, pr, pt, n_pos, neg_rate, tot_entity, bern_prob, slots, n_slots, seed, offset, nh, nr, nt,
                          (hipStream_t)stream);
}

int kge_sample_batch(const int64_t* triples, const int64_t* perm, int64_t start, int64_t n_pos, int32_t neg_rate,
                     int64_t tot_entity, const float* bern_prob, const uint64_t* slots, int64_t n_slots, uint64_t seed,
                     uint64_t offset, int32_t layout, int64_t* o0, int64_t* o1, int64_t* o2, int64_t* o3, int64_t* o4,
                     int64_t* o5, const int64_t* dev_cursor, void* stream) {
    if (n_pos == 0) return 0;
    if (!triples || !perm || start < 0 || n_pos < 0 || neg_rate <= 0 || tot_entity <= 1 || (layout != 0 && layout != 1) ||
        !o0 || !o1 || !o2 || !o3 || (layout == 0 && (!o4 || !o5))) {
        set_error("kge_sample_batch: bad arguments");
        return -1;
    }
    if (slots && (n_slots & (n_slots - 1))) { set_error("kge_sample_batch: n_slots must be a power of two"); return -1; }
    if (!dev_cursor) {   // (entity columns only: this entry point is not told the relation count)
        if (int rc = debug_check_ids("kge_sample_batch", "head", triples, n_pos, 3, 0, tot_entity, (hipStream_t)stream, perm, start)) return rc;
        if (int rc = debug_check_ids("kge_sample_batch", "tail", triples, n_pos, 3, 2, tot_entity, (hipStream_t)stream, perm, start)) return rc;
    }
    int64_t* const out[6] = {o0, o1, o2, o3, o4, o5};
    return launch_sample_batch(triples, perm, start, n_pos, neg_rate, tot_entity, bern_prob, slots, n_slots, seed, offset,
                               layout, out, dev_cursor, (hipStream_t)stream);
}

int kge_step_advance(int64_t* dev_cursor, float* dev_hyper, int64_t batch_stride, int64_t n_batches,
                     int64_t draws_per_batch, float lr, void* stream) {
    if (!dev_cursor || !dev_hyper || n_batches < 1 || batch_stride < 0 || draws_per_batch < 0) {
        set_error("kge_step_advance: bad arguments");
        return -1;
    }
    return launch_step_advance(dev_cursor, dev_hyper, batch_stride, n_batches, draws_per_batch, lr, (hipStream_t)stream);
}

}  // extern "C"
