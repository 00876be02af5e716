// kge_internal.h -- host-side launcher prototypes shared between the translation units of libkge_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/kge_hip.h"
#include "kge_device.h"

// GEMM inner loops read the LDS operands of step k + 2 before issuing the MFMAs of step k; the scheduling barrier keeps the
// compiler from sinking those reads back in front of their use (-DKGE_NO_LDS_PREFETCH: A/B builds without it)
#ifdef KGE_NO_LDS_PREFETCH
#define KGE_KEEP_READS_AHEAD() ((void)0)
#else
#define KGE_KEEP_READS_AHEAD() __builtin_amdgcn_sched_barrier(0)
#endif

namespace kge {

void set_error(const char* fmt, ...);
bool is_vector_model(int model);  // models handled by the gather/row kernels (everything but RESCAL, NTN)

// (G, NCH) geometry for a row length; returns false when the row is too long for the register-resident kernels
struct Geometry { int G, NCH; };
bool pick_geometry(int max_row_dim, Geometry* out);
DeviceModel to_device_model(const kge_model_desc* m);

// kge_debug.hip -- the A/B and debugging switches of DESIGN.md section 5a, in ONE place with ONE meaning: the value set by
// kge_set_switch(name, v) if any, else the integer in the environment variable KGE_<NAME> ("0" = off, "1" = on, other integers where
// a switch takes them), else -1 = unset (the built-in rule decides).  A getenv per launch costs ~0.1 us; tests flip the variables
// inside one process, so nothing is cached.
int switch_value(const char* name /* without the KGE_ prefix, upper case */);

// kge_debug.hip -- id-range scans of the debug mode (KGE_DEBUG_IDS=1 / kge_set_debug): 0 when the mode is off or every id is in
// range, -3 (+ kge_last_error) on the first offender; they synchronise the stream and are skipped during graph capture
bool debug_ids();
int debug_check_ids(const char* who, const char* what, const int64_t* ids, int64_t n, int stride, int col, int64_t bound, hipStream_t s,
                    const int64_t* perm = nullptr, int64_t start = 0);
int debug_check_ids32(const char* who, const char* what, const int32_t* ids, int64_t n, int stride, int col, int64_t bound, hipStream_t s);
int debug_check_hrt(const char* who, const kge_model_desc* m, const int64_t* h, const int64_t* r, const int64_t* t, int64_t n, hipStream_t s);
int debug_check_triples(const char* who, int64_t tot_entity, int64_t tot_relation, const int64_t* triples, int64_t n, hipStream_t s,
                        const int64_t* perm = nullptr, int64_t start = 0);

// kge_score.hip
int launch_score_forward(const kge_model_desc* m, const int64_t* h, const int64_t* r, const int64_t* t,
                         int64_t n, float* scores, hipStream_t s);
int launch_score_backward(const kge_model_desc* m, const int64_t* h, const int64_t* r, const int64_t* t,
                          int64_t n, const float* dscore, hipStream_t s);
int launch_pairwise_hinge(const kge_model_desc* m, const int64_t* ph, const int64_t* pr, const int64_t* pt,
                          const int64_t* nh, const int64_t* nr, const int64_t* nt, int64_t n, float margin,
                          float* loss, hipStream_t s);
int launch_pairwise_hinge_sampled(const kge_model_desc* m, const int64_t* triples, const int64_t* perm, int64_t start,
                                  int64_t n, const float* bern, const uint64_t* slots, int64_t n_slots, uint64_t seed,
                                  uint64_t offset, const int64_t* cursor, float margin, float* loss, hipStream_t s);
struct FusedSampler;
int launch_pointwise_logistic(const kge_model_desc* m, const int64_t* h, const int64_t* r, const int64_t* t,
                              const int64_t* y, int64_t n, int bundle, float lmbda, int reg_type, float* loss,
                              const FusedSampler* sampler /* NULL: explicit rows */, hipStream_t s);
int launch_pointwise_logistic_sampled(const kge_model_desc* m, const int64_t* triples, const int64_t* perm, int64_t start,
                                      int64_t n_pos, int neg_rate, const float* bern, const uint64_t* slots, int64_t n_slots,
                                      uint64_t seed, uint64_t offset, const int64_t* cursor, float lmbda, int reg_type,
                                      float* loss, hipStream_t s);
int launch_selfadv_bundle(const kge_model_desc* m, const int64_t* ph, const int64_t* pr, const int64_t* pt,
                          const int64_t* nh, const int64_t* nr, const int64_t* nt, int64_t n_pos, int neg_rate,
                          float alpha, float* loss, hipStream_t s);  // returns 1 when neg_rate exceeds the group width
// id array that may come in two pieces (the positives followed by the negatives of a pair batch): element i of [a | b].
// The relation-matrix models score / back-propagate both sides of the fused pairwise step as ONE batch of 2n triples.
struct IdSplit {
    const int64_t* a; const int64_t* b; int64_t na;
    __device__ __forceinline__ int64_t at(int64_t i) const { return i < na ? a[i] : b[i - na]; }
};
inline IdSplit id_whole(const int64_t* a, int64_t n) { return IdSplit{a, nullptr, n}; }

// Staged (atomic-free) gradient output of the bundle kernels: every gradient row a bundle produces goes to its own slot of
// a staging buffer with plain stores; kge_optimizer_step_staged (kge_staged.hip) sums the slots of each parameter row in a
// fixed order inside the optimiser sweep.  Slots: positive i owns static slots [i*ns, (i+1)*ns); negative pair p owns
// dynamic slots n_pos*ns + [p*nd, (p+1)*nd) and registers itself with the entity it drew (count / bucket / overflow chain).
struct StageSink {
    float* stage; int64_t stride;
    int32_t* count; int32_t* bucket; int32_t* head; int32_t* next; int32_t cap;
    int32_t ns, nd;
    int32_t* dyn_list;   // optional [n_neg]: entry p = the entity pair p drew if p was that entity's FIRST registrant, else -1
    int32_t spare;       // != 0: ns + nd spare rows follow the used ones (rows padded to the bundle's width: unconditional stores)
};
__device__ __forceinline__ void stage_register(const StageSink& k, int c, int pair) {
    const int pos = atomicAdd(k.count + c, 1);
    if (pos < k.cap) k.bucket[(int64_t)c * k.cap + pos] = pair;
    else k.next[pair] = atomicExch(k.head + c, pair + 1) - 1;   // head holds pair + 1: an all-zero buffer is an empty set
    if (k.dyn_list) k.dyn_list[pair] = pos == 0 ? c : -1;   // (a plain store per pair: no shared counter to serialise on)
}
int launch_rotate_bundle_sampled(const kge_model_desc* m, const int64_t* triples, const int64_t* perm, int64_t start,
                                 int64_t n_pos, int neg_rate, float alpha, const float* bern, const uint64_t* slots,
                                 int64_t n_slots, uint64_t seed, uint64_t offset, const int64_t* cursor, float* loss,
                                 const StageSink* sink /* NULL: atomic scatter into m->grads */,
                                 float* pair_scale /* staged: per-negative factor the optimiser applies to the dynamic slots */,
                                 hipStream_t s);
int launch_optimizer_staged(int kind, const kge_staged_step* st, float lr, int64_t step, hipStream_t s);
int launch_pointwise_logistic_sampled_staged(const kge_model_desc* m, const int64_t* triples, const int64_t* perm, int64_t start,
                                             int64_t n_pos, int neg_rate, const float* bern, const uint64_t* slots,
                                             int64_t n_slots, uint64_t seed, uint64_t offset, float lmbda, int reg_type,
                                             float* loss, const StageSink& sink, hipStream_t s);
// kge_score_generic.hip: generic (roles-table) tail of the sampler-fused hinge step, after the shared-row specialisations
struct FusedSampler;
int launch_pairwise_hinge_sampled_generic(const kge_model_desc* m, Geometry geo, const FusedSampler& fs, int64_t n,
                                          float margin, float* loss, hipStream_t s);
// kge_score_ext.hip: the same generic kernels instantiated for TransM / CP / SimplE / SimplE_ignr / QuatE
struct FusedSampler;
int launch_score_forward_ext(const kge_model_desc* m, Geometry geo, const int64_t* h, const int64_t* r, const int64_t* t,
                             int64_t n, float* scores, hipStream_t s);
int launch_score_backward_ext(const kge_model_desc* m, Geometry geo, const int64_t* h, const int64_t* r, const int64_t* t,
                              int64_t n, const float* dscore, hipStream_t s);
int launch_pairwise_hinge_ext(const kge_model_desc* m, Geometry geo, const int64_t* ph, const int64_t* pr, const int64_t* pt,
                              const int64_t* nh, const int64_t* nr, const int64_t* nt, int64_t n, float margin,
                              float* loss, const FusedSampler* fs, bool sampled, hipStream_t s);
int launch_pointwise_logistic_ext(const kge_model_desc* m, Geometry geo, const int64_t* h, const int64_t* r, const int64_t* t,
                                  const int64_t* y, int64_t n, int bundle, float lmbda, int reg_type, float* loss,
                                  const FusedSampler* sampler, hipStream_t s);
int launch_selfadv_coeffs(float* pos_scores, float* neg_scores, int64_t n_pos, int neg_rate, float alpha,
                          float* loss, hipStream_t s);

// kge_dense.hip (RESCAL / NTN: f32 MFMA contraction paths) + table normalisation
int launch_rescal_normalize(float* ent, int64_t E, float* rel, int64_t R, int k, float* scratch, size_t scratch_floats,
                            hipStream_t s);
size_t dense_workspace_bytes(const kge_model_desc* m, int64_t n);
size_t ntn_workspace_bytes(const kge_model_desc* m, int64_t n);
int launch_l2norm_reg(const float* param, float* grad, int64_t numel, float lmbda, float* scratch, float* loss, hipStream_t s);
int launch_rescal_forward(const kge_model_desc* m, const int64_t* h, const int64_t* r, const int64_t* t,
                          int64_t n, float* scores, void* ws, size_t ws_bytes, hipStream_t s);
int launch_rescal_backward(const kge_model_desc* m, const int64_t* h, const int64_t* r, const int64_t* t,
                           int64_t n, const float* dscore, void* ws, size_t ws_bytes, bool grouped, hipStream_t s);
int launch_ntn_pair_forward(const kge_model_desc* m, const int64_t* ph, const int64_t* pr, const int64_t* pt, const int64_t* nh,
                            const int64_t* nr, const int64_t* nt, int64_t n, float* scores2, void* ws, size_t ws_bytes,
                            hipStream_t s);
int launch_ntn_pair_backward(const kge_model_desc* m, const int64_t* ph, const int64_t* pr, const int64_t* pt, const int64_t* nh,
                             const int64_t* nr, const int64_t* nt, int64_t n, const float* dscore2, void* ws, size_t ws_bytes,
                             hipStream_t s);
int launch_transr_pair_forward(const kge_model_desc* m, const int64_t* ph, const int64_t* pr, const int64_t* pt,
                               const int64_t* nh, const int64_t* nr, const int64_t* nt, int64_t n, float* scores2, void* ws,
                               size_t ws_bytes, hipStream_t s);
int launch_transr_pair_backward(const kge_model_desc* m, const int64_t* ph, const int64_t* pr, const int64_t* pt,
                                const int64_t* nh, const int64_t* nr, const int64_t* nt, int64_t n, const float* dscore2,
                                void* ws, size_t ws_bytes, hipStream_t s);
int launch_rescal_pair_forward(const kge_model_desc* m, const int64_t* ph, const int64_t* pr, const int64_t* pt,
                               const int64_t* nh, const int64_t* nr, const int64_t* nt, int64_t n, float* scores2, void* ws,
                               size_t ws_bytes, hipStream_t s);
int launch_rescal_pair_backward(const kge_model_desc* m, const int64_t* ph, const int64_t* pr, const int64_t* pt,
                                const int64_t* nh, const int64_t* nr, const int64_t* nt, int64_t n, const float* dscore2,
                                void* ws, size_t ws_bytes, hipStream_t s);
int launch_ntn_forward(const kge_model_desc* m, const int64_t* h, const int64_t* r, const int64_t* t,
                       int64_t n, float* scores, void* ws, size_t ws_bytes, hipStream_t s);
int launch_ntn_backward(const kge_model_desc* m, const int64_t* h, const int64_t* r, const int64_t* t,
                        int64_t n, const float* dscore, void* ws, size_t ws_bytes, bool forward_in_ws, hipStream_t s);
// kge_transr_rows.hip: the pairwise TransR step of large batches (negatives keep their positives' relations) in two launches
size_t transr_rows_ws_bytes(const kge_model_desc* m, int64_t n);
bool transr_rows_ok(const kge_model_desc* m, int64_t n, size_t ws_bytes);
int launch_transr_pair_step(const kge_model_desc* m, const int64_t* ph, const int64_t* pr, const int64_t* pt, const int64_t* nh,
                            const int64_t* nt, int64_t n, float margin, float* loss, void* ws, size_t ws_bytes, hipStream_t s);
// kge_transr.hip
size_t transr_workspace_bytes(const kge_model_desc* m, int64_t n);
int launch_transr_forward(const kge_model_desc* m, const int64_t* h, const int64_t* r, const int64_t* t,
                          int64_t n, float* scores, void* ws, size_t ws_bytes, hipStream_t s);
int launch_transr_backward(const kge_model_desc* m, const int64_t* h, const int64_t* r, const int64_t* t,
                           int64_t n, const float* dscore, void* ws, size_t ws_bytes, bool grouped, hipStream_t s);
int launch_transr_eval_prepare(const kge_model_desc* m, const int64_t* triples, int64_t n, const int64_t* group_rel,
                               int64_t n_groups, int Kpad, int64_t ntiles, float* cand, float* qvec, float* qscale,
                               hipStream_t s);
int launch_hinge_coeffs(float* pos, float* neg, int64_t n, float margin, float* loss, hipStream_t s);
// the whole pairwise RESCAL step in one launch after the grouping (negatives share the positives' relation ids)
bool rescal_pair_step_ok(const kge_model_desc* m, int64_t n, size_t ws_bytes);
size_t rescal_slab_extra_bytes(const kge_model_desc* m, int64_t n);   // workspace behind the standard pairwise layout (kge_rescal_slab.hip)
bool rescal_stage_ok(const kge_model_desc* m, int64_t n, size_t ws_bytes);   // the staged (atomic-free) form can take this shape
int launch_rescal_pair_step(const kge_model_desc* m, const int64_t* ph, const int64_t* pr, const int64_t* pt, const int64_t* nh,
                            const int64_t* nt, int64_t n, float margin, float* loss, void* ws, size_t ws_bytes, unsigned* touched,
                            const kge_rescal_stage* stage /* NULL: entity gradients through float atomics */, hipStream_t s);

// kge_opt.hip
int launch_optimizer(int kind, float* p, float* g, float* s1, float* s2, int64_t numel, float lr, int64_t step,
                     int zero_grad, const float* dev_hyper, const int64_t* cursor_in, int64_t* cursor_out, float* hyper_out,
                     int64_t batch_stride, int64_t n_batches, int64_t draws_per_batch, hipStream_t s);

int launch_optimizer_rows(int kind, float* p, float* g, float* s1, float* s2, int64_t rows, int dim, float lr, int64_t step,
                          int zero_grad, int normalize, const float* dev_hyper, const unsigned* touched, unsigned* touched_clear,
                          const kge_rescal_stage* stage /* NULL: the gradient rows of `g` */, hipStream_t s);

bool optimizer_rownorm_ok(int64_t rows, int64_t dim, size_t scratch_floats);
int launch_optimizer_rownorm(int kind, float* p, float* g, float* s1, float* s2, int64_t rows, int64_t dim, float lr, int64_t step,
                             int zero_grad, const float* dev_hyper, float* scratch, size_t scratch_floats, const int64_t* cursor_in,
                             int64_t* cursor_out, float* hyper_out, int64_t batch_stride, int64_t n_batches, int64_t draws_per_batch,
                             hipStream_t s);

int launch_optimizer_rows_rownorm(int kind, float* p, float* g, float* s1, float* s2, int64_t rows, int dim, float* wp, float* wg,
                                  float* ws1, float* ws2, int64_t wrows, int64_t wdim, float lr, int64_t step, int zero_grad, int normalize,
                                  const float* dev_hyper, const unsigned* touched, unsigned* touched_clear, const kge_rescal_stage* st,
                                  float* scratch, size_t scratch_floats, const int64_t* cursor_in, int64_t* cursor_out, float* hyper_out,
                                  int64_t batch_stride, int64_t n_batches, int64_t draws_per_batch, hipStream_t s);

// kge_pull.hip (owner-computes training step: no atomics, optimiser fused, deterministic)
int pull_partial_stride(int dim);
int pull_hat_stride(int dim);
// the staged owner-computes step of the remaining pointwise gather models (kge_ownx.hip)
bool ownx_model(int model);
int ownx_groups_per_block(int model, int dim);
int ownx_partial_stride(int model, int dim);
size_t ownx_stage_floats(int model, int dim, int64_t n_pairs);
int launch_ownx_step(const kge_model_desc* m, float* const* state1, float* const* state2, const int32_t* pairs, int64_t n_pairs,
                     const kge_pull_lists* lists, const int32_t* items, int64_t n_items, const uint32_t* listed, const int32_t* inc,
                     float* partials, const int32_t* multi, int64_t n_multi, int dense, float lmbda, int reg_type, int optimizer, float lr,
                     int64_t step, const int32_t* next_pairs, const int32_t* next_inv, int64_t next_n, const float* bern,
                     const uint64_t* slots, int64_t n_slots, uint64_t seed, uint64_t next_offset, const kge_pull_lists* next_lists,
                     float* loss, float* stage, hipStream_t s);
// TransH / TransD gradients in the two-launch owner-computes form (kge_pullx.hip)
int transx_groups_per_block(int dim);
int transx_partial_stride(int dim);
void transx_scratch_bytes(int model, int dim, int64_t n, size_t* stage, size_t* recs);
int launch_transx_grad_step(const kge_model_desc* m, const int32_t* pairs, int64_t n_pairs, const kge_pull_lists* lists,
                            const int32_t* items, int64_t n_items, const uint32_t* listed, const int32_t* inc, float* partials,
                            const int32_t* multi, int64_t n_multi, float margin, float* stage, float* recs, int reset_lists,
                            const int32_t* next_pairs, const int32_t* next_inv, int64_t next_n, const float* bern,
                            const uint64_t* slots, int64_t n_slots, uint64_t seed, uint64_t next_offset,
                            const kge_pull_lists* next_lists, float* loss, hipStream_t s);
void pull_direction_bytes(int dim, int l1, int64_t n, size_t* codes, size_t* recs);
int pull_groups_per_block(int dim);
int launch_pull_step(const kge_model_desc* m, float* const tables_out[2], const float* const hat_in[2], float* const hat_out[2],
                     const float* norm_in, float* norm_out, float* const state1[2], float* const state2[2], const int32_t* pairs,
                     const kge_pull_lists* lists, const int32_t* items, int64_t n_items, const uint32_t* dense_skip, const int32_t* inc, float* partials,
                     const int32_t* multi, int64_t n_multi, float margin, int optimizer, float lr, int64_t step,
                     const float* dev_hyper, int reset_lists, const int32_t* next_pairs, const int32_t* next_inv, int64_t next_n, const float* bern,
                     const uint64_t* slots, int64_t n_slots, uint64_t seed, uint64_t next_offset,
                     const kge_pull_lists* next_lists, float* loss, const kge_pull_direction* dir, hipStream_t s);
int launch_row_norms(const float* table, int64_t rows, int dim, float* out, float* hat, hipStream_t s);
int launch_pull_sample(const int32_t* pairs, const int32_t* inv, int64_t n, int64_t E, const float* bern, const uint64_t* slots, int64_t n_slots,
                       uint64_t seed, uint64_t offset, const int64_t* cursor, const kge_pull_lists* out, hipStream_t s);
int launch_pull_lists_explicit(const int32_t* pairs, const int32_t* inv, const int64_t* nh, const int64_t* nt, int64_t n,
                               const kge_pull_lists* out, hipStream_t s);

// kge_own.hip (two-phase owner-computes step of the pointwise models)
int own_groups_per_block(int model, int dim);
int own_partial_stride(int model, int dim);
int launch_own_step(const kge_model_desc* m, const int32_t* pairs, int64_t n_pairs, const kge_pull_lists* lists, const int32_t* items,
                    int64_t n_items, const uint32_t* listed, const int32_t* inc, float* partials, int dense, float lmbda, int reg_type,
                    int reset_lists, const int32_t* next_pairs, const int32_t* next_inv, int64_t next_n, const float* bern,
                    const uint64_t* slots, int64_t n_slots, uint64_t seed, uint64_t next_offset, const kge_pull_lists* next_lists,
                    float* loss, float* stage, hipStream_t s);
int launch_own_apply(const kge_model_desc* m, float* const* state1, float* const* state2, const int32_t* pairs, int64_t n_pairs,
                     const kge_pull_lists* lists, const int32_t* items, int64_t n_items, const uint32_t* listed, const int32_t* multi,
                     int64_t n_multi, float* partials, int dense, int optimizer, float lr, int64_t step, int multi_only, hipStream_t s);
int launch_own_step_fused(const kge_model_desc* m, float* const* state1, float* const* state2, const int32_t* pairs, int64_t n_pairs,
                          const kge_pull_lists* lists, const int32_t* items, int64_t n_items, const uint32_t* listed, const int32_t* inc,
                          float* partials, int dense, float lmbda, int reg_type, int optimizer, float lr, int64_t step,
                          const int32_t* next_pairs, const int32_t* next_inv, int64_t next_n, const float* bern, const uint64_t* slots,
                          int64_t n_slots, uint64_t seed, uint64_t next_offset, const kge_pull_lists* next_lists, float* loss,
                          float* stage, hipStream_t s);

// kge_eval.hip
size_t eval_workspace_bytes(const kge_model_desc* m, int64_t n, int64_t tables = 1);
int launch_eval_ranks_grouped(const kge_model_desc* m, const int64_t* triples, int64_t n, const int32_t* group_of_triple,
                              const int64_t* group_rel, int64_t n_groups, const int32_t* qblocks, int64_t n_qblocks,
                              const int64_t* tail_off, const int32_t* tail_ids, const int64_t* head_off,
                              const int32_t* head_ids, void* ws, size_t ws_bytes, int32_t* ranks, int32_t* ties, hipStream_t s);
int launch_eval_ranks(const kge_model_desc* m, const int64_t* triples, int64_t n, const int64_t* tail_off,
                      const int32_t* tail_ids, const int64_t* head_off, const int32_t* head_ids, void* ws,
                      size_t ws_bytes, int32_t* ranks, int32_t* ties /* [2, n] or NULL */, hipStream_t s);
int launch_eval_sweep_scores(const kge_model_desc* m, const int64_t* triples, int64_t n, void* ws, size_t ws_bytes,
                             float* scores, hipStream_t s, int side = 2);
// The 1-N head's filtered rank through the same sweep pipeline (kge_head_1n_rank): a pseudo-model, internal to the library, whose
// "candidate table" is [ent row | bias] and whose query vectors are the caller's activation rows [x | 1]; energy = -sigmoid(logit).
// tables[0] = ent [E, dim], tables[1] = bias [E] or NULL, tables[2] = x [n, dim].
constexpr int KGE_HEAD_1N_INTERNAL = 100;
size_t head_rank_workspace_bytes(int64_t n, int dim, int64_t E, bool has_bias);
int launch_head_rank(const float* x, int64_t n, int dim, const float* ent, int64_t E, const float* bias, const int64_t* triples,
                     const int64_t* off, const int32_t* ids, void* ws, size_t ws_bytes, int32_t* ranks, int32_t* ties, float* energies,
                     hipStream_t s);

int launch_rank_from_scores(const float* scores, int64_t nq, int64_t E, const int64_t* truth, const int64_t* off,
                            const int32_t* ids, int32_t* rank, int32_t* frank, hipStream_t s);

// kge_ntn_eval.hip
size_t ntn_eval_workspace_bytes(const kge_model_desc* m, int64_t n);
int launch_ntn_eval_ranks(const kge_model_desc* m, const int64_t* triples, int64_t n, const int64_t* tail_off,
                          const int32_t* tail_ids, const int64_t* head_off, const int32_t* head_ids, void* ws,
                          size_t ws_bytes, int32_t* ranks, hipStream_t s);
int launch_ntn_eval_scores(const kge_model_desc* m, const int64_t* triples, int64_t n, void* ws, size_t ws_bytes,
                           float* scores, hipStream_t s);

// kge_head.hip (1-N scoring head of the projection models)
int launch_head_forward(const float* x, int64_t B, int d, const float* ent, int64_t E, const float* bias, float* preds,
                        int bf16, hipStream_t s);
size_t head_backward_workspace_bytes();
int launch_head_backward(const float* x, int64_t B, int d, const float* ent, int64_t E, const float* preds,
                         const float* dpreds, float* dx, float* g_ent, float* g_bias, void* ws, size_t ws_bytes, hipStream_t s);
size_t head_bce_workspace_bytes(int64_t B, int64_t E, int64_t n_pos);
int launch_head_bce(const float* x, int64_t B, int d, const float* ent, int64_t E, const float* bias,
                    const int64_t* lab_off, const int32_t* lab_ids, int64_t n_pos, float label_smoothing, void* ws,
                    size_t ws_bytes, float* loss, float* dx, float* g_ent, float* g_bias, hipStream_t s);

// kge_sampler.hip
int launch_triple_set_build(const int64_t* triples, int64_t n, uint64_t* slots, int64_t n_slots, hipStream_t s);
int launch_corrupt(const int64_t* ph, const int64_t* pr, const int64_t* pt, int64_t n_pos, int neg_rate,
                   int64_t E, const float* bern, const uint64_t* slots, int64_t n_slots, uint64_t seed,
                   uint64_t offset, int64_t* nh, int64_t* nr, int64_t* nt, hipStream_t s);

int launch_sample_batch(const int64_t* triples, const int64_t* perm, int64_t start, int64_t n_pos, int neg_rate,
                        int64_t E, const float* bern, const uint64_t* slots, int64_t n_slots, uint64_t seed,
                        uint64_t offset, int layout, int64_t* const out[6], const int64_t* cursor, hipStream_t s);
int launch_step_advance(int64_t* cursor, float* hyper, int64_t batch_stride, int64_t n_batches, int64_t draws_per_batch,
                        float lr, hipStream_t s);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return -2;
    }
    return 0;
}

}  // namespace kge
