/*
 * kge_hip.h -- C ABI of the MI355X (gfx950) KGE scoring / training / ranking path.
 *
 * This is the drop-in boundary: a plain C shared library (libkge_hip.so), no torch or C++ types in
 * any signature.  The reference (Sujit-O/pykg2vec v0.0.52) is pure Python on PyTorch and has no FFI
 * of its own; each entry point below replaces the chain of stock ATen ops the reference issues at
 * the cited site (paths relative to the reference tree), and INTEGRATION.md shows the ctypes stub a
 * pykg2vec maintainer would add to call it.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; the caller owns all memory,
 *     the library never allocates device memory (workspaces are passed in), so every call is
 *     hipGraph-capturable;
 *   - tables are fp32 row-major [rows, dim] exactly as nn.Embedding stores them (models/Domain.py:8-17),
 *     ids are int64 (torch.LongTensor, utils/trainer.py:288-293), scores fp32;
 *   - `stream` is a hipStream_t passed as void*; calls are asynchronous on it and never synchronise;
 *   - return 0 on success, negative on error; kge_last_error() gives the message (thread-local);
 *   - stateless and re-entrant; one process per GPU for multi-GPU use.
 */
#ifndef KGE_HIP_H
#define KGE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KGE_ABI_VERSION 3
#define KGE_MAX_TABLES 12

/* model ids; tables[] order == the reference's `parameter_list` order */
enum kge_model {
    KGE_TRANSE = 0,   /* pairwise.py:12-93    tables: ent_embeddings, rel_embeddings */
    KGE_TRANSH = 1,   /* pairwise.py:96-182   + w */
    KGE_TRANSD = 2,   /* pairwise.py:185-278  ent_embeddings, rel_embeddings, ent_mappings, rel_mappings */
    KGE_ROTATE = 3,   /* pairwise.py:727-791  ent_embeddings(real), ent_embeddings_imag, rel_embeddings */
    KGE_RESCAL = 4,   /* pairwise.py:794-865  ent_embeddings, rel_matrices[R, k*k] */
    KGE_NTN = 5,      /* pairwise.py:868-963  ent, rel, mr1, mr2, br, mr */
    KGE_DISTMULT = 6, /* pointwise.py:391-458 ent_embeddings, rel_embeddings */
    KGE_COMPLEX = 7,  /* pointwise.py:122-238 ent_real, ent_img, rel_real, rel_img (also ComplexN3) */
    KGE_ANALOGY = 8,  /* pointwise.py:13-119  ent, rel, ent_real, ent_img, rel_real, rel_img */
    KGE_TRANSM = 9,   /* pairwise.py:281-365  ent_embeddings, rel_embeddings, theta[R] (fixed per-relation weight, no grad) */
    KGE_CP = 10,      /* pointwise.py:320-387 sub_embeddings, rel_embeddings, obj_embeddings */
    KGE_SIMPLE = 11,  /* pointwise.py:461-546 ent_head, ent_tail, rel, rel_inv; energy = -clamp(<h,r,t> + <t,r_inv,h>/2, +-20) */
    KGE_SIMPLE_IGNR = 12, /* pointwise.py:549-592 same tables; energy = -clamp(<h,r,t> + <t,r_inv,h>, +-20) */
    KGE_QUATE = 13,   /* pointwise.py:595-768 ent_s, ent_x, ent_y, ent_z, rel_s, rel_x, rel_y, rel_z (rel_w is unused by forward) */
    KGE_TRANSR = 14   /* pairwise.py:367-470  ent_embeddings[E,dim], rel_embeddings[R,rel_dim], rel_matrix[R, dim*rel_dim] */
};

#define KGE_FLAG_L1 1u /* l1_flag of TransE/TransH/TransD (pairwise.py:72-76) */

enum kge_optimizer { KGE_OPT_SGD = 0, KGE_OPT_ADAM = 1, KGE_OPT_ADAGRAD = 2, KGE_OPT_RMSPROP = 3,
                     KGE_OPT_GRADIENT = 4 /* kge_pull_step only: no update -- the row's dense gradient is WRITTEN to tables_out
                                            (every row, zeros included; data-parallel ranks reduce it afterwards) */ };
/* F2 / N3 / N3_ABS: lmbda * mean_i(sum x^2 | x^3 | |x|^3 over the rows gathered for row i) (pointwise.py get_reg's).
 * ID_F2 / ID_N3: SimplE.get_reg as the reference executes it (pointwise.py:528-536) -- it is handed the ID tensors, so
 * the term is the constant lmbda * sum_i(h_i^p + r_i^p + t_i^p) in float32: added to the loss, no gradient. */
enum kge_reg { KGE_REG_NONE = 0, KGE_REG_F2 = 1, KGE_REG_N3 = 2, KGE_REG_N3_ABS = 3, KGE_REG_ID_F2 = 4, KGE_REG_ID_N3 = 5 };

typedef struct kge_model_desc {
    int32_t model;              /* enum kge_model */
    uint32_t flags;             /* KGE_FLAG_* */
    int64_t tot_entity;
    int64_t tot_relation;
    int32_t dim;                /* hidden_size / ent_hidden_size */
    int32_t rel_dim;            /* rel_hidden_size (NTN, TransD); == dim otherwise */
    float margin;               /* RotatE: margin in the score (pairwise.py:791) */
    float phase_scale;          /* RotatE: pi / embedding_range, embedding_range=(margin+2)/dim (pairwise.py:747,781) */
    const float* tables[KGE_MAX_TABLES]; /* parameter tables, parameter_list order */
    float* grads[KGE_MAX_TABLES];        /* dense gradient buffers of the same shapes (may be NULL for forward) */
} kge_model_desc;

int kge_abi_version(void);
const char* kge_last_error(void);

/* Debug mode: range-check every id the caller hands in against its table before anything is launched
 * (0 <= entity id < tot_entity, 0 <= relation id < tot_relation) -- what nn.Embedding does for the reference by raising
 * IndexError (models/Domain.py:8-13).  Off by default (the kernels index with the ids as given); on with kge_set_debug(1) or
 * KGE_DEBUG_IDS=1 in the environment.  A failing call returns -3, kge_last_error() names the entry point, the column, the first
 * offending position and value.  The scan synchronises the stream (skipped while the stream is being captured).
 * Covered: kge_score_forward/backward, kge_train_pairwise_hinge, kge_train_pairwise_selfadv, kge_train_pointwise_logistic,
 * kge_rescal_pair_step, every *_sampled entry point (the batch rows triples[perm[start..start+n)]), kge_sample_batch, kge_corrupt,
 * kge_triple_set_build, kge_pull_index_build (all listed batches: which covers the owner-computes runs built on that index),
 * kge_eval_ranks / _grouped / kge_eval_sweep_scores, kge_rank_from_scores. */
int kge_set_debug(int32_t check_ids);
/* A/B switches of the dispatch rules: RESCAL_UNFUSED, RESCAL_ROWS, RESCAL_G, RESCAL_G2, TRANSR_ROWS, TRANSR_G,
 * EVAL_GEMM, HEAD_TILE, NTN_BIG, OPT_NT, PULL_G, ROTATE_SPLIT.  value >= 0 forces it, -1 hands the decision back to the environment variable KGE_<name> (an
 * integer; "0" off, "1" on) or, if that is unset, to the built-in rule.  Same meaning on both sides of the boundary. */
int kge_set_switch(const char* name, int32_t value);
int kge_get_debug(void);
/* The scan itself, unconditionally: -3 if some ids[i] is outside [0, bound). */
int kge_check_ids(const int64_t* ids, int64_t n, int64_t bound, void* stream);
/* ---- filter lists of the rank sweep, built on the device (kge_index.hip).
 * Replaces the per-query lookups into hr_t[(h, r)] / tr_h[(t, r)] -- dicts of sets over train + valid + test,
 * data/kgcontroller.py:410-428, consulted inside the reference's rank loop utils/evaluator.py:70-123 -- by per-query CSR lists:
 * query i's known tails are tail_ids[tail_off[i] .. tail_off[i+1]) (distinct, ascending), its known heads likewise.
 * known: int64 [n_known, 3] (the three splits concatenated, duplicates allowed), queries: int64 [n_queries, 3].
 * Two calls because the caller owns every buffer: _count sorts the packed keys into `workspace` (kge_filter_csr_workspace_bytes)
 * and writes the offsets [n_queries + 1] and totals[2] = {sum of tail list lengths, sum of head list lengths}; the caller reads
 * totals, sizes the id arrays, and _fill writes them (same workspace, untouched in between). */
size_t kge_filter_csr_workspace_bytes(int64_t n_known, int64_t n_queries);
int kge_filter_csr_count(const int64_t* known, int64_t n_known, const int64_t* queries, int64_t n_queries, int64_t tot_entity,
                         int64_t tot_relation, void* workspace, size_t workspace_bytes, int64_t* tail_off, int64_t* head_off,
                         int64_t* totals, void* stream);
int kge_filter_csr_fill(const int64_t* queries, int64_t n_queries, int64_t n_known, const void* workspace, const int64_t* tail_off,
                        const int64_t* head_off, int32_t* tail_ids, int32_t* head_ids, void* stream);

/* An empty kernel launch of `tag` workgroups of 64 threads ("kge::k_marker"): a boundary that shows up in a rocprofv3 kernel /
 * counter trace.  bench.py's counter passes use it to cut one process's dispatch sequence into per-configuration segments. */
int kge_debug_marker(int32_t tag, void* stream);

/* Scratch bytes the score / train entry points need for a call on n rows (n pairs for the pairwise step).
 * 0 for the gather-type models; RESCAL and TransR group the batch by relation on the device (about 5R + n + n/32 ints per side),
 * TransR's large-batch pairwise step (negatives that keep their positives' relations: nr == pr) keeps
 * 2n*(rel_dim + 1) floats per side between its two launches (dL/d(h^ M), dL/d(t^ M) and the rows' inverse norms),
 * NTN keeps n*(4d + 3k_r + 6) floats of intermediates per side; the hinge step adds 2n floats. */
size_t kge_workspace_bytes(const kge_model_desc* m, int64_t n);

/* Model.forward(h, r, t) -> energies[n]   (pairwise.py:56-76,166-174,270-278,786-791,855-860,955-960;
 * pointwise.py:97-104,185-188,444-446).  RESCAL: call kge_rescal_normalize first (its forward does). */
int kge_score_forward(const kge_model_desc* m, const int64_t* h, const int64_t* r, const int64_t* t,
                      int64_t n, float* scores, void* workspace, size_t workspace_bytes, void* stream);

/* Autograd backward of the above: grads[table] += d(sum_i dscore[i]*score_i)/d table, dense
 * (nn.Embedding sparse=False, models/Domain.py:8-13).  Replaces loss.backward() utils/trainer.py:298. */
int kge_score_backward(const kge_model_desc* m, const int64_t* h, const int64_t* r, const int64_t* t,
                       int64_t n, const float* dscore, void* workspace, size_t workspace_bytes, void* stream);

/* Rescal.embed side effect (pairwise.py:843-844,862-865): W <- W / ||W_row||_2 in place, both tables. */
int kge_rescal_normalize(float* ent, int64_t tot_entity, float* rel, int64_t tot_relation, int32_t k,
                         void* stream);
/* The same with a caller-owned scratch (kge_rescal_normalize_scratch_bytes): long relation-matrix rows are then normalised
 * by many workgroups per row (partial sums of squares per 4096-float chunk, added in chunk order) instead of one. */
size_t kge_rescal_normalize_scratch_bytes(int64_t tot_relation, int32_t k);
int kge_rescal_normalize_ws(float* ent /* NULL: relation matrices only */, int64_t tot_entity, float* rel, int64_t tot_relation,
                            int32_t k, void* scratch, size_t scratch_bytes, void* stream);

/* Fused Trainer.train_step_pairwise (utils/trainer.py:147-157) with Criterion.pairwise_hinge
 * (utils/criterion.py:25-29): scores both triples of each of the n pairs, adds sum(max(0, s+ + margin - s-))
 * to the loss accumulators and the loss gradient to m->grads.  neg_rate must be 1 (the hinge adds [B] to
 * [B*neg_rate]).  `loss`: float[32*32] striped accumulators, total = sum_k loss[32*k].  One kernel for the gather-type
 * models; RESCAL / TransR / NTN run forward over [positives | negatives] as one batch of 2n triples, the hinge
 * coefficients in place, and one backward over the same 2n (RESCAL: normalise-free, call kge_rescal_normalize first). */
int kge_train_pairwise_hinge(const kge_model_desc* m,
                             const int64_t* ph, const int64_t* pr, const int64_t* pt,
                             const int64_t* nh, const int64_t* nr, const int64_t* nt,
                             int64_t n, float margin, void* workspace, size_t workspace_bytes,
                             float* loss, void* stream);

/* The same step with the negative sampler FUSED IN FRONT (data/generator.py:42-97 + utils/trainer.py:147-157 +
 * utils/criterion.py:25-29 in one kernel, neg_rate 1): pair i is triples[perm[start+i]] and its corruption drawn with
 * Philox counter offset+i -- the very batch kge_sample_batch(start, n, 1, ..., seed, offset) would emit.  Gather-type
 * models only.  dev_cursor as in kge_sample_batch. */
int kge_train_pairwise_hinge_sampled(const kge_model_desc* m, const int64_t* triples, const int64_t* perm,
                                     int64_t start, int64_t n, const float* bern_prob, const uint64_t* slots,
                                     int64_t n_slots, uint64_t seed, uint64_t offset, const int64_t* dev_cursor,
                                     float margin, float* loss, void* stream);

/* Fused train_step_pairwise with Criterion.pariwise_logistic (utils/criterion.py:13-23; RotatE):
 * self-adversarial weights softmax(alpha * -s-) over the neg_rate negatives of each positive (detached).
 * Negatives of positive i are rows [i*neg_rate, (i+1)*neg_rate) (data/generator.py:71-95).
 * workspace: at least n_pos*(1+neg_rate) floats. */
int kge_train_pairwise_selfadv(const kge_model_desc* m,
                               const int64_t* ph, const int64_t* pr, const int64_t* pt,
                               const int64_t* nh, const int64_t* nr, const int64_t* nt,
                               int64_t n_pos, int32_t neg_rate, float alpha, float* workspace,
                               float* loss, void* stream);

/* RotatE self-adversarial step with the negative sampler FUSED IN FRONT (one launch): positive i is
 * triples[perm[start+i]], its neg_rate negatives are drawn with Philox counters offset + i*neg_rate + j -- the batch
 * kge_sample_batch(start, n_pos, neg_rate, ..., seed, offset) would emit.  The positive's five rows and the relation's
 * sin/cos stay in registers across the bundle; only the corrupting entity's two rows are gathered per negative.
 * neg_rate <= lane-group width (32 for hidden_size <= 256, else 64). */
int kge_train_pairwise_selfadv_sampled(const kge_model_desc* m, const int64_t* triples, const int64_t* perm,
                                       int64_t start, int64_t n_pos, int32_t neg_rate, float alpha,
                                       const float* bern_prob, const uint64_t* slots, int64_t n_slots,
                                       uint64_t seed, uint64_t offset, const int64_t* dev_cursor,
                                       float* loss, void* stream);

/* Fused Trainer.train_step_pointwise (utils/trainer.py:176-180): Criterion.pointwise_logistic
 * (utils/criterion.py:31-34) mean(softplus(y*s)) + lmbda*get_reg (pointwise.py:106-119,190-202,224-238,448-458). */
int kge_train_pointwise_logistic(const kge_model_desc* m, const int64_t* h, const int64_t* r,
                                 const int64_t* t, const int64_t* y, int64_t n,
                                 int32_t bundle /* rows per sampler bundle = 1+neg_rate (data/generator.py:125-156);
 <=1: none */,
                                 float lmbda, int32_t reg_type, float* loss, void* stream);

/* The same step with the negative sampler FUSED IN FRONT (data/generator.py:99-158 + utils/trainer.py:176-180 +
 * utils/criterion.py:31-34 in one kernel): bundle i = triples[perm[start+i]] with y=+1 followed by its neg_rate
 * corruptions with y=-1, drawn with the Philox counters kge_sample_batch(layout 1) uses -- both paths see identical
 * rows.  neg_rate must not exceed the lane group of the model's row length (32, or 64 above 256 floats). */
int kge_train_pointwise_logistic_sampled(const kge_model_desc* m, const int64_t* triples, const int64_t* perm,
                                         int64_t start, int64_t n_pos, int32_t neg_rate, const float* bern_prob,
                                         const uint64_t* slots, int64_t n_slots, uint64_t seed, uint64_t offset,
                                         const int64_t* dev_cursor, float lmbda, int32_t reg_type, float* loss,
                                         void* stream);

/* Dense optimiser sweep with torch.optim defaults (utils/trainer.py:112-131): SGD, Adam(0.9,0.999,1e-8),
 * Adagrad(eps 1e-10), RMSprop(alpha .99, eps 1e-8).  state1/state2: exp_avg/exp_avg_sq (Adam),
 * sum / square_avg in state1 (Adagrad / RMSprop).  step is 1-based.  zero_grad != 0 also clears grad
 * (optimizer.zero_grad(), utils/trainer.py:272) in the same pass. */
int kge_optimizer_step(int32_t kind, float* param, float* grad, float* state1, float* state2,
                       int64_t numel, float lr, int64_t step, int32_t zero_grad,
                       const float* dev_hyper /* NULL, or device {lr, step_size, bc2_sqrt} from kge_step_advance */,
                       void* stream);

/* The same optimiser with one wave per table ROW, optionally followed by the row renormalisation Rescal.embed applies to its
 * tables at the start of the NEXT forward (pairwise.py:843-844,862-865: W <- W / ||W_row||_2): normalize != 0 stores the
 * renormalised row, so the next step's kge_rescal_normalize pass over this table (a full read + write of it) disappears.  The
 * result equals kge_optimizer_step followed by kge_rescal_normalize up to the fp32 summation order of the row norm.  A caller must leave
 * normalize == 0 on the last step before the tables are observed (the reference leaves them as the optimiser wrote them).
 * rows of at most 1024 floats.
 * touched_rows (may be NULL): one bit per row (uint32 words, bit r & 31 of word r >> 5), set by the step that accumulated
 * gradients (kge_rescal_pair_step); a clear bit promises that the row of `grad` is zero, and the sweep does not read it.
 * touched_clear (may be NULL): the bitmap of the OTHER step parity, reset to zero on the way (two bitmaps alternate so that a
 * step's optimiser never clears bits its own sweep still reads). */
int kge_optimizer_step_rows(int32_t kind, float* param, float* grad, float* state1, float* state2, int64_t rows, int32_t dim,
                            float lr, int64_t step, int32_t zero_grad, int32_t normalize, const float* dev_hyper,
                            const uint32_t* touched_rows, uint32_t* touched_clear, void* stream);

/* The pairwise RESCAL train step of kge_train_pairwise_hinge (Trainer.train_model_epoch -> model.forward on both sides ->
 * Criterion.pairwise_hinge -> backward, utils/trainer.py:262-276 with pairwise.py:838-870) for batches whose negatives keep
 * their positives' relation ids (every sampler of the reference: data/generator.py:143-196 corrupts heads and tails only):
 * pairs are grouped by relation and each (relation, 16 pairs) tile computes scores, hinge and the three gradients in one
 * workgroup.  kge_train_pairwise_hinge takes this path by itself when called with nr == pr (the same buffer).
 * hidden size: even, at most 256 (kge_rescal_pair_step_ok).  workspace: kge_workspace_bytes(m, n) bytes.
 * touched_rows (may be NULL): entity rows that received a gradient get their bit set (see kge_optimizer_step_rows). */
int kge_rescal_pair_step(const kge_model_desc* m, const int64_t* ph, const int64_t* pr, const int64_t* pt, const int64_t* nh,
                         const int64_t* nt, int64_t n, float margin, void* workspace, size_t workspace_bytes, float* loss,
                         uint32_t* touched_rows, void* stream);
int kge_rescal_pair_step_ok(const kge_model_desc* m, int64_t n);

/* ---- Entity gradients of the pairwise RESCAL step WITHOUT float atomics (round 5; the relation-matrix gradient of the slab form has
 * none already).  The grouping launch orders the pairs of every relation of at most 64 pairs by pair index; the forward launch registers
 * slot 4 g + j of grouped pair g (j: 0 positive head, 1 positive tail, 2 negative head, 3 negative tail) with its entity (count / bucket /
 * overflow chain) while marking the entity in touched_rows; the backward launch stores every side's gradient row to gstage[slot] with plain
 * stores and the pair's hinge coefficient to dsv[g]; kge_optimizer_step_rows_staged -- the row-owner optimiser of
 * kge_optimizer_step_rows -- sums the rows registered with a touched entity in ascending slot order (skipping pairs with a zero
 * coefficient), applies the optimiser, and resets the entity's list.  Results are bit-reproducible run to run as long as no relation has
 * more than 64 pairs in the batch (a longer relation spans several chunks, whose shares of the relation-matrix gradient add atomically).
 * Replaces: loss.backward() into dense nn.Embedding gradients + optimizer.step() (utils/trainer.py:298-299) for RESCAL.
 * count / head: int32 [tot_entity], all zero between steps (the optimiser resets what it consumed); bucket: int32 [tot_entity * cap];
 * next: int32 [4 n]; gstage: float [4 n][dim]; dsv: float [n]; 1 <= cap <= 32.  kge_rescal_stage_ok: 1 when the step of n pairs takes
 * the slab form with the workspace kge_workspace_bytes reports, the hidden size is a multiple of 4 and n <= 4096. */
typedef struct kge_rescal_stage {
    float* gstage;
    float* dsv;
    int32_t* count;
    int32_t* bucket;
    int32_t* head;
    int32_t* next;
    int32_t cap;
} kge_rescal_stage;
int kge_rescal_stage_ok(const kge_model_desc* m, int64_t n);
int kge_rescal_pair_step_staged(const kge_model_desc* m, const int64_t* ph, const int64_t* pr, const int64_t* pt, const int64_t* nh,
                                const int64_t* nt, int64_t n, float margin, void* workspace, size_t workspace_bytes, float* loss,
                                uint32_t* touched_rows, const kge_rescal_stage* stage, void* stream);
int kge_optimizer_step_rows_staged(int32_t kind, float* param, float* grad, float* state1, float* state2, int64_t rows, int32_t dim,
                                   float lr, int64_t step, int32_t normalize, const float* dev_hyper, const uint32_t* touched_rows,
                                   uint32_t* touched_clear, const kge_rescal_stage* stage, void* stream);

/* NTN.get_reg (pairwise.py:962-963): loss += lmbda * sqrt(sum_i param[i]^2), grad += lmbda * param / that root, over
 * ONE flat buffer holding every table of the model (pad with zeros).  scratch: 1 float. */
int kge_l2norm_reg(const float* param, float* grad, int64_t numel, float lmbda, float* scratch, float* loss,
                   void* stream);

/* Filtered-rank evaluation: Evaluator.test (utils/evaluator.py:309-334) + MetricCalculator.get_tail_rank /
 * get_head_rank (utils/evaluator.py:70-123) for n test triples, without materialising scores or orderings.
 *   triples        int64 [n,3] (h, r, t)
 *   tail_off/ids   CSR over queries: known tails of (h_i, r_i) = hr_t[(h,r)]  (int64 [n+1], int32 ids)
 *   head_off/ids   CSR over queries: known heads of (t_i, r_i) = tr_h[(t,r)]
 *   ranks          int32 [4,n]: rank_head, rank_tail, filtered_rank_head, filtered_rank_tail (0-based, as the
 *                  reference before settle() adds 1); rank = #{e : s_e < s_true}
 * workspace from kge_eval_workspace_bytes().
 * TransR: candidates are scored in the relation space, so the sweep table is projected by M_r once per call and ALL n
 * triples of a call must carry the same relation id (triples[1] is used); kge_eval_ranks_grouped below takes many
 * relation groups per call. */
size_t kge_eval_workspace_bytes(const kge_model_desc* m, int64_t n);
int kge_eval_ranks(const kge_model_desc* m, const int64_t* triples, int64_t n,
                   const int64_t* tail_off, const int32_t* tail_ids,
                   const int64_t* head_off, const int32_t* head_ids,
                   void* workspace, size_t workspace_bytes, int32_t* ranks, void* stream);

/* Models whose candidate-side transform depends on the relation (TransR: projection by M_r; TransH: hyperplane of w_r;
 * TransD: mapping r_m), several relations per call: one transformed + normalised candidate table per relation group,
 * then the plain L1 / L2 sweep.  For TransH / TransD this is an alternative to kge_eval_ranks (which transforms inside
 * the sweep, per query) that pays off once a relation has a few test triples.  `triples` are sorted by relation; group g = the run of triples with relation
 * group_rel[g] (device int64 [n_groups]); group_of_triple (device int32 [n]) names each triple's group; qblocks
 * (device int32 [n_qblocks,4]) = {group, first query, query count <= 16, 0} partitions every group's query range
 * [2a, 2b) (query 2i = tail sweep of triple i, 2i+1 = head sweep) into sweep workgroups.  One candidate table per group
 * lives in the workspace (kge_eval_grouped_workspace_bytes); ranks as in kge_eval_ranks. */
size_t kge_eval_grouped_workspace_bytes(const kge_model_desc* m, int64_t n, int64_t n_groups);
int kge_eval_ranks_grouped(const kge_model_desc* m, const int64_t* triples, int64_t n,
                           const int32_t* group_of_triple, const int64_t* group_rel, int64_t n_groups,
                           const int32_t* qblocks, int64_t n_qblocks,
                           const int64_t* tail_off, const int32_t* tail_ids,
                           const int64_t* head_off, const int32_t* head_ids,
                           void* workspace, size_t workspace_bytes, int32_t* ranks, void* stream);

/* The same two entry points with a TIE report (round 4).  ranks are count-based, rank = #{e : s_e < s_true}: exact whenever no
 * candidate ties the true one; on exact ties the reference lands somewhere inside the tie group (wherever torch.topk puts it,
 * utils/evaluator.py:70-123) while this count is the OPTIMISTIC end of it.  ties: int32 [2, n] (row 0 head sweeps, row 1 tail
 * sweeps) = the number of OTHER candidates whose energy equals the true candidate's bit for bit, counted in the same sweep:
 * rank <= reference rank <= rank + ties.  0 everywhere for trained distance / dot-product models; large where a scorer saturates
 * (SimplE's clamp, constant outputs).  -1 = not counted (NTN's sweep).  NULL: the plain entry points. */
int kge_eval_ranks_ties(const kge_model_desc* m, const int64_t* triples, int64_t n, const int64_t* tail_off,
                        const int32_t* tail_ids, const int64_t* head_off, const int32_t* head_ids, void* workspace,
                        size_t workspace_bytes, int32_t* ranks, int32_t* ties, void* stream);
int kge_eval_ranks_grouped_ties(const kge_model_desc* m, const int64_t* triples, int64_t n, const int32_t* group_of_triple,
                                const int64_t* group_rel, int64_t n_groups, const int32_t* qblocks, int64_t n_qblocks,
                                const int64_t* tail_off, const int32_t* tail_ids, const int64_t* head_off,
                                const int32_t* head_ids, void* workspace, size_t workspace_bytes, int32_t* ranks, int32_t* ties,
                                void* stream);

/* Evaluator.test_tail_rank / test_head_rank score vectors (utils/evaluator.py:249-273) through the sweep
 * kernels: for each of the n triples, scores[2i][e] = energy of (h_i, r_i, e) and scores[2i+1][e] = energy of
 * (e, r_i, t_i) for every entity e.  scores: float [2n, E].  Used by the predict_tail_rank / predict_head_rank
 * hooks (n = 1) and by parity tests.  Same workspace as kge_eval_ranks. */
int kge_eval_sweep_scores(const kge_model_desc* m, const int64_t* triples, int64_t n,
                          void* workspace, size_t workspace_bytes, float* scores, void* stream);

/* One side of the same sweep: side 0 = Evaluator.test_tail_rank's score vector (utils/evaluator.py:249-260: energies of
 * (h_i, r_i, e) over all e; column 2 of `triples` is not read), side 1 = test_head_rank's (:262-273: energies of (e, r_i, t_i);
 * column 0 not read).  scores: float [n, E].  What the predict_tail_rank / predict_head_rank hooks call (n = 1): the other
 * side's query row is never swept.  TransR (candidates projected per call) and NTN compute both sides anyway: refused here,
 * use kge_eval_sweep_scores.  Same workspace as kge_eval_ranks. */
int kge_eval_sweep_scores_side(const kge_model_desc* m, const int64_t* triples, int64_t n, int side,
                               void* workspace, size_t workspace_bytes, float* scores, void* stream);

/* MetricCalculator.get_tail_rank / get_head_rank (utils/evaluator.py:70-123) from materialised score rows, for models
 * whose sweep is served by kge_score_forward over all candidates (NTN): scores float [nq, E], truth int64 [nq] (the true
 * entity of each row), CSR of known entities per row (may be NULL) -> rank, filtered rank (int32 [nq], 0-based). */
int kge_rank_from_scores(const float* scores, int64_t nq, int64_t tot_entity, const int64_t* truth,
                         const int64_t* off, const int32_t* ids, int32_t* rank, int32_t* frank, void* stream);

/* Negative corruption (data/generator.py:42-97,125-156) on device.
 *   kge_triple_set_build: open-addressing set of packed (h,r,t) train triples; slots = power of two >= 2n.
 *   kge_corrupt: for each positive and each of neg_rate slots draw u (Philox4x32-10 keyed by seed, counter =
 *   global slot index): u > prob[r] (bern) or 0.5 -> replace tail else head; redraw while the corrupted triple
 *   is in the train set. */
int kge_triple_set_build(const int64_t* triples, int64_t n, uint64_t* slots, int64_t n_slots, void* stream);
int kge_corrupt(const int64_t* ph, const int64_t* pr, const int64_t* pt, int64_t n_pos, int32_t neg_rate,
                int64_t tot_entity, const float* bern_prob /* NULL = uniform 0.5 */,
                const uint64_t* slots, int64_t n_slots, uint64_t seed, uint64_t offset,
                int64_t* nh, int64_t* nr, int64_t* nt, void* stream);

/* raw_data_generator + process_function_pairwise / _pointwise (data/generator.py:11-158) as ONE launch: the batch's
 * positives are triples[perm[start + i]], i < n_pos (triples int64 [N,3], perm int64 [N]); negatives as kge_corrupt.
 *   layout 0 (pairwise): out = {ph, pr, pt, nh, nr, nt}; positives [n_pos], negatives [n_pos*neg_rate]
 *   layout 1 (pointwise): out = {h, r, t, y, -, -}, each [n_pos*(1+neg_rate)]: every positive (y=+1) is followed by its
 *                         neg_rate negatives (y=-1) */
int kge_sample_batch(const int64_t* triples, const int64_t* perm, int64_t start, int64_t n_pos, int32_t neg_rate,
                     int64_t tot_entity, const float* bern_prob, const uint64_t* slots, int64_t n_slots,
                     uint64_t seed, uint64_t offset, int32_t layout,
                     int64_t* o0, int64_t* o1, int64_t* o2, int64_t* o3, int64_t* o4, int64_t* o5,
                     const int64_t* dev_cursor /* NULL, or device {start, draw offset}: added to start / offset */,
                     void* stream);

/* Device-resident step state so that a whole training step (advance -> sample -> fused step -> optimiser) is a static
 * launch sequence that can be captured once in a hipGraph and replayed (the launch-bound small-batch regime, default
 * B=128 of the reference).  cursor: int64[8] = {start, draw_offset, opt_step, batch_idx, draws, ...} zero-initialised by
 * the caller; hyper: float[4].  Each call moves to the next batch of the epoch (wrapping at n_batches, like
 * raw_data_generator data/generator.py:28-35), advances the Philox offset and the optimiser step (Adam bias terms). */
int kge_step_advance(int64_t* dev_cursor, float* dev_hyper, int64_t batch_stride, int64_t n_batches,
                     int64_t draws_per_batch, float lr, void* stream);

/* Dense optimiser over a [rows, dim] table of WIDE rows followed by the in-place row renormalisation W <- W / ||row||_2 that
 * Rescal.embed applies to rel_matrices at the next forward (models/pairwise.py:843-844): the optimiser launch leaves the sums of
 * squares of 4 096-float chunks in `scratch`, one rescale launch follows -- two launches and one pass less than kge_optimizer_step +
 * kge_rescal_normalize_ws, bit-identical rows.  rows < 1 024, dim >= 16 384 (kge_optimizer_step_rownorm_ok); scratch: rows *
 * ceil(dim / 4096) floats (kge_rescal_normalize_scratch_bytes).  dev_cursor / next_cursor / next_hyper: all NULL, or the step-state
 * transition of kge_optimizer_step_advance folded into the launch (hipGraph-replayed steps).  Replaces: optimizer.step() +
 * zero_grad() on rel_matrices (utils/trainer.py:272,299) + the rel_matrices half of Rescal.embed's renormalisation. */
int kge_optimizer_step_rownorm_ok(int64_t rows, int64_t dim);
int kge_optimizer_step_rownorm(int32_t kind, float* param, float* grad, float* state1, float* state2, int64_t rows, int64_t dim, float lr,
                               int64_t step, int32_t zero_grad, const float* dev_hyper, void* scratch, size_t scratch_bytes,
                               const int64_t* dev_cursor, int64_t* next_cursor, float* next_hyper, int64_t batch_stride,
                               int64_t n_batches, int64_t draws_per_batch, void* stream);

/* RESCAL's whole optimiser step in two launches (round 6): kge_optimizer_step_rows / _rows_staged on the table of SHORT rows
 * (ent_embeddings: param .. dim, touched_rows / touched_clear / stage as there; normalize must be 1) with the optimiser of the table of
 * WIDE rows (rel_matrices: wparam .. wdim, as kge_optimizer_step_rownorm) riding in the first wrows * ceil(wdim / 4096) workgroups of the
 * same launch, then the rescale launch of the wide rows.  Stored values are bit-identical to the two separate calls; the launch-bound
 * chunk kernel (11.6 us at the YAGO3-10 shape) disappears under the entity table's sweep.  zero_grad applies to the wide-row table and,
 * when stage is NULL, to the short-row table.  Replaces: optimizer.step() + zero_grad() (utils/trainer.py:272,299) + Rescal.embed's
 * renormalisation of both tables (models/pairwise.py:843-844). */
int kge_optimizer_step_rows_rownorm(int32_t kind, float* param, float* grad, float* state1, float* state2, int64_t rows, int32_t dim,
                                    float* wparam, float* wgrad, float* wstate1, float* wstate2, int64_t wrows, int64_t wdim,
                                    float lr, int64_t step, int32_t zero_grad, int32_t normalize, const float* dev_hyper,
                                    const uint32_t* touched_rows, uint32_t* touched_clear, const kge_rescal_stage* stage,
                                    void* scratch, size_t scratch_bytes, const int64_t* dev_cursor, int64_t* next_cursor,
                                    float* next_hyper, int64_t batch_stride, int64_t n_batches, int64_t draws_per_batch, void* stream);

/* kge_optimizer_step for hipGraph-replayed steps, with kge_step_advance for the FOLLOWING step folded into the same
 * launch: the sweep reads its scalars from dev_hyper (this step's set) and thread 0 derives the next step's state
 * next_cursor / next_hyper from dev_cursor.  The two sets must be distinct buffers (the replayed graphs alternate
 * between them), which removes the one-thread advance launch from every step. */
int kge_optimizer_step_advance(int32_t kind, float* param, float* grad, float* state1, float* state2,
                               int64_t numel, float lr, int32_t zero_grad,
                               const float* dev_hyper, const int64_t* dev_cursor,
                               int64_t* next_cursor, float* next_hyper,
                               int64_t batch_stride, int64_t n_batches, int64_t draws_per_batch, void* stream);

/* ---- Owner-computes ("pull") training step for TransE: Generator (data/generator.py:42-97) +
 * Trainer.train_step_pairwise (utils/trainer.py:147-157) + Criterion.pairwise_hinge (utils/criterion.py:25-29) +
 * loss.backward() + optimizer.step() (utils/trainer.py:298-299), neg_rate 1, with no atomics on parameters or gradients,
 * no gradient buffer and bit-reproducible results.  Every parameter row has one owner per step that re-evaluates the
 * pairs the row occurs in and keeps only its own gradient (see csrc/kge_pull.hip).  Host-built, per batch (int32):
 *   pairs  [n, 4]        (h, r, t, 0) of the batch's positives, in batch order (Philox counter of pair i = offset + i)
 *   inc    [3n]          static incidences sorted by (row, pair, role): pair << 2 | role, role 0 = head, 1 = tail,
 *                        2 = relation; rows are numbered entities first, then tot_entity + relation
 *   items  [n_items, 4]  (row, first incidence, end incidence, info), laid out in workgroup slots of
 *                        kge_pull_groups_per_block(dim) items (row -1 = padding).  info & 3 = kind: 0 = the row's only item;
 *                        3 = one of 2..groups-per-block items of a row, all in consecutive slots of ONE workgroup (info =
 *                        3 | segment << 2 | segments << 6; they combine through LDS); 1 / 2 = first / later item of a row
 *                        with more items than that (info = kind | slot << 2: partial sums go to partials[slot] and the
 *                        finishing kernel).  The row's first item also walks the row's corrupting-entity draws.
 *   multi  [n_multi, 4]  (row, first slot, number of slots, 0) of the rows finished by the finishing kernel
 * Per step (device): a kge_pull_lists set, written by kge_pull_sample (same draws as kge_sample_batch with the same
 * seed / offset) or kge_pull_lists_explicit (given negatives) and consumed -- and reset when reset_lists != 0 -- by the
 * step.  next_pairs != NULL: the sampler of the NEXT batch rides in this step's launch and fills next_lists (a second
 * set), so a steady-state step is one launch (+ a small finishing launch when rows are cut into several items).
 * m->tables = the tables read; tables_out = the other half of the double buffer; hat_in / hat_out: the row-normalised
 * copies x / max(||x||, 1e-12) of the tables read / written (what other owners gather), rows of kge_pull_hat_stride(dim)
 * floats (round 6: compact, = dim; the padded layout of rounds 2-5, zeros up to kge_pull_partial_stride(dim), behind
 * KGE_HAT_COMPACT=0); norm_in / norm_out [E + R]: their
 * L2 row norms (kge_row_norms fills both before the first step); state1 / state2: optimiser state per table.
 * dim must be a multiple of 4 (rows move as float4) and at most 1024.  TransM: m->tables[2] = the per-relation weights.
 * partials: kge_pull_partial_stride(dim) floats per slot.
 * dense_skip (optional): bitmap over the E + R rows of the rows `items` lists; every other row is visited implicitly after the
 * listed items (a batch that touches a small part of the tables needs no explicit item per untouched row).  NULL: `items`
 * covers every row.
 * optimizer == KGE_OPT_GRADIENT: no update -- tables_out receive the dense gradient rows (every row, zeros included);
 * hat_out / norm_out / state may be NULL.  This is what data-parallel ranks run before their reduce-scatter. */
#define KGE_PULL_BUCKET 16
typedef struct kge_pull_lists {
    int32_t* pc;      /* [n]  per pair: corrupting entity | (tail corrupted) << 24 | (first pair to register with that entity
                         this step) << 27 */
    int32_t* count;   /* [tot_entity]  pairs that drew the entity this step; all 0 between steps */
    int32_t* bucket;  /* [tot_entity * KGE_PULL_BUCKET]  the first KGE_PULL_BUCKET of them (pair indices; the overflow path) */
    int32_t* head;    /* [tot_entity]  overflow list head, all -1 between steps */
    int32_t* next;    /* [n]  overflow list links */
    /* ready-made visit descriptors, written by the sampler so that an owner reads them with ONE dependent load:
     *   sdesc   [3 n][4]  static incidences, in `inc` order: (h, r, t, corrupting entity | tail << 24 | role << 25)
     *   dbucket [tot_entity * KGE_PULL_BUCKET][4]  the drawers of an entity: (h, r, t, PAIR INDEX | tail << 24 | 3 << 25) -- the
     *           corrupting entity is the bucket's own; the owner orders the entries by pair index */
    int32_t* sdesc;
    int32_t* dbucket;
} kge_pull_lists;
/* Two-phase ("staged direction") form of the step, selected by passing a kge_pull_direction with non-NULL buffers: phase 1
 * (k_pull_eval) evaluates every pair of the batch ONCE -- same gathers and arithmetic as an owner's visit -- and leaves a
 * 16-byte record (hinge coefficient, both energies, which side was corrupted) plus the signed direction of both residuals
 * (2 bits per element; L1 models only -- a direction passed with an L2 model is ignored: both evaluate-once forms of L2 measured
 * slower than its one-phase step and were removed); phase 2 is the owner-computes kernel with visits that read those records
 * instead of re-evaluating the pair (3.25 evaluations per pair -> 1).  Same gradients bit for bit (same coefficients, same fused
 * multiply-adds in the same order); the loss is accumulated by phase 1.  lists_without_descriptors != 0: the sampler riding in
 * this launch skips sdesc / dbucket (the NEXT step must then also be two-phase). */
typedef struct kge_pull_direction {
    void* codes;       /* kge_pull_direction_bytes(...).codes bytes */
    float* recs;       /* 4 floats per pair */
    int64_t n_pairs;   /* pairs in the batch */
    int32_t lists_without_descriptors;
} kge_pull_direction;
int kge_pull_direction_bytes(int32_t dim, int32_t l1, int64_t n_pairs, size_t* codes_bytes, size_t* recs_bytes);
int kge_pull_partial_stride(int32_t dim);
int kge_pull_hat_stride(int32_t dim);         /* floats between consecutive rows of the hat tables (kge_row_norms writes, kge_pull_step reads / writes) */
int kge_pull_groups_per_block(int32_t dim);   /* owner groups per 256-thread workgroup: items are laid out in workgroup slots */
int kge_row_norms(const float* table, int64_t rows, int32_t dim, float* norms, float* normalised, void* stream);
int kge_pull_sample(const int32_t* pairs, const int32_t* inv, int64_t n, int64_t tot_entity, const float* bern_prob,
                    const uint64_t* slots, int64_t n_slots, uint64_t seed, uint64_t offset, const int64_t* dev_cursor,
                    const kge_pull_lists* out, void* stream);
int kge_pull_lists_explicit(const int32_t* pairs, const int32_t* inv, const int64_t* nh, const int64_t* nt, int64_t n,
                            const kge_pull_lists* out, void* stream);
int kge_pull_step(const kge_model_desc* m, float* const tables_out[2], const float* const hat_in[2], float* const hat_out[2],
                  const float* norm_in, float* norm_out,
                  float* const state1[2], float* const state2[2], const int32_t* pairs, const kge_pull_lists* lists,
                  const int32_t* items, int64_t n_items, const uint32_t* dense_skip, const int32_t* inc, float* partials, const int32_t* multi,
                  int64_t n_multi, float margin, int32_t optimizer, float lr, int64_t step, const float* dev_hyper,
                  int32_t reset_lists, const int32_t* next_pairs, const int32_t* next_inv, int64_t next_n, const float* bern_prob,
                  const uint64_t* slots, int64_t n_slots, uint64_t seed, uint64_t next_offset, const kge_pull_lists* next_lists,
                  float* loss, const kge_pull_direction* direction /* NULL: one-phase step */, void* stream);

/* A whole run of consecutive owner-computes steps enqueued by ONE native call (the per-step work is ~35 us of GPU time: a
 * Python-level loop cannot keep the queue full).  The plan holds everything that does not change between steps. */
typedef struct kge_pull_batch {
    const int32_t* pairs;   /* [n_pairs, 4] */
    const int32_t* items;   /* [n_items, 4] */
    int64_t n_items;
    const uint32_t* dense_skip;   /* NULL: `items` covers every parameter row.  Else a bitmap over the E + R rows: `items` lists
                                   only rows with a static incidence (bit set) and every other row is visited implicitly
                                   (dense optimisers move every row every step; its corrupting-entity draws are walked too) */
    const int32_t* inc;     /* [3 * n_pairs] */
    const int32_t* inv;     /* [3 * n_pairs] inverse of inc (the sampler of this batch files its visit descriptors by it) */
    const int32_t* multi;   /* [n_multi, 4] or NULL */
    int64_t n_multi;
    int64_t n_pairs;
} kge_pull_batch;
typedef struct kge_pull_plan {
    kge_model_desc model[2];      /* descriptor over table half 0 / half 1 (tables[0..1]; grads unused) */
    float* hat[2][2];             /* [half][table] row-normalised copies */
    float* norm[2];               /* [half] row norms, E + R */
    float* state1[2];             /* [table] optimiser state (NULL where unused) */
    float* state2[2];
    kge_pull_lists lists[2];      /* the two sampler list sets */
    const kge_pull_batch* batches;  /* HOST array of n_batches entries */
    int64_t n_batches;
    float* partials;
    float margin;
    int32_t optimizer;
    float lr;
    const float* bern_prob;
    const uint64_t* slots;
    int64_t n_slots;
    uint64_t seed;
    int64_t draws_per_batch;      /* Philox counters consumed per batch (= n_pairs * neg_rate) */
    float* loss;
    kge_pull_direction direction; /* codes == NULL: one-phase steps; else the two-phase form (n_pairs is taken from each batch) */
} kge_pull_plan;
/* Steps on batches first_batch .. first_batch + n_steps - 1.  src_half: the table half the first step reads (halves
 * alternate); cur_list: the list set the first step consumes; lists_ready == 0: a stand-alone sampler launch fills it first
 * (it must be cleared).  first_opt_step: optimiser step number of the first step (1-based); first_offset: its Philox
 * offset.  sample_after_last == 1: the last step also carries the sampler of batch first_batch + n_steps; == 2: of batch 0 (the
 * first batch of the next epoch over the same permutation; its Philox offset continues from the last step's). */
size_t kge_pull_plan_bytes(void);   /* sizeof(kge_pull_plan): lets a binding check its struct layout */
int kge_pull_run(const kge_pull_plan* plan, int64_t first_batch, int64_t n_steps, int32_t src_half, int32_t cur_list,
                 int32_t lists_ready, int64_t first_opt_step, uint64_t first_offset, int32_t sample_after_last, void* stream);

/* ---- TransH / TransD gradients without float atomics (csrc/kge_pullx.hip): Generator (data/generator.py:42-97, neg_rate 1) +
 * Trainer.train_step_pairwise (utils/trainer.py:147-157) + Criterion.pairwise_hinge (utils/criterion.py:25-29) + loss.backward()
 * for TransH (models/pairwise.py:143-182; tables ent, rel, w) and TransD (:229-278; tables ent, rel, ent_mappings, rel_mappings),
 * in the two-launch owner-computes form: every (positive, negative) pair is evaluated once and its gradient rows are stored to
 * the pair's slots of `stage`; one owner per parameter row then sums the staged rows of its incidences in a fixed order and
 * writes the row ONCE into m->grads (dense tables of the parameters' shapes; rows without incidence are not written: the
 * buffers must be zero there, which kge_optimizer_step(zero_grad) leaves behind).  The optimiser is the caller's next call.
 * Index: kge_pull_index_build(groups_per_block = kge_transx_groups_per_block(dim)), compact (`listed` = its bitmap: entities that
 * were only drawn are then owned by the first pair that drew them) or not (`listed` = NULL); lists / ride-along sampler as kge_pull_step.  dim % 4 == 0, dim <= 512.  The loss is added to the striped accumulators. */
int kge_transx_groups_per_block(int32_t dim);
int kge_transx_partial_stride(int32_t dim);          /* floats per partial slot */
int kge_transx_scratch_bytes(int32_t model, int32_t dim, int64_t n_pairs, size_t* stage_bytes, size_t* recs_bytes);
int kge_transx_grad_step(const kge_model_desc* m, const int32_t* pairs, int64_t n_pairs, const kge_pull_lists* lists,
                         const int32_t* items, int64_t n_items, const uint32_t* listed, const int32_t* inc, float* partials,
                         const int32_t* multi, int64_t n_multi, float margin, float* stage, float* recs, int32_t reset_lists,
                         const int32_t* next_pairs, const int32_t* next_inv, int64_t next_n, const float* bern_prob,
                         const uint64_t* slots, int64_t n_slots, uint64_t seed, uint64_t next_offset,
                         const kge_pull_lists* next_lists, float* loss, void* stream);

/* A run of consecutive TransH / TransD steps enqueued by ONE native call: per step kge_transx_grad_step (three launches) and the
 * dense optimiser over the flat buffers (kge_optimizer_step with zero_grad: the gradient tables of `model` must be views into
 * flat_grad).  Arguments as kge_pull_run. */
typedef struct kge_transx_plan {
    kge_model_desc model;                 /* tables = the parameters (views into flat_param); grads = views into flat_grad */
    kge_pull_lists lists[2];
    const kge_pull_batch* batches;        /* HOST array of n_batches entries (dense_skip = the `listed` bitmap of the compact index) */
    int64_t n_batches;
    float* partials; float* stage; float* recs;
    float margin;
    float* flat_param; float* flat_grad; float* flat_state1; float* flat_state2; int64_t flat_numel;
    int32_t optimizer; float lr;
    const float* bern_prob; const uint64_t* slots; int64_t n_slots; uint64_t seed;
    int64_t draws_per_batch;
    float* loss;
} kge_transx_plan;
size_t kge_transx_plan_bytes(void);
int kge_transx_run(const kge_transx_plan* plan, int64_t first_batch, int64_t n_steps, int32_t cur_list, int32_t lists_ready,
                   int64_t first_opt_step, uint64_t first_offset, int32_t sample_after_last, void* stream);

/* ---- Owner-computes training step of the POINTWISE models, two phases (csrc/kge_own.hip): Generator (data/generator.py:99-158,
 * neg_rate 1) + Trainer.train_step_pointwise (utils/trainer.py:176-180) + Criterion.pointwise_logistic (utils/criterion.py:31-34)
 * + get_reg (pointwise.py:190-202,224-238,448-458) + loss.backward() + optimizer.step() (utils/trainer.py:298-299,112-131) for
 * DistMult (tables ent, rel) and ComplEx / ComplexN3 (ent_re, ent_im, rel_re, rel_im), with no float atomics and bit-reproducible
 * results.  A bundle = positive i of the batch and its one sampled corruption (the rows kge_sample_batch(layout 1) emits for
 * neg_rate 1 with the same seed / offset); the per-step lists, the incidence index (pairs / inc / items / multi, built by
 * kge_pull_index_build with groups_per_block = kge_own_groups_per_block) and the ride-along sampler are those of kge_pull_step.
 *   kge_own_step   phase 1: one owner group per touched parameter row (an entity owner holds the re and im rows of its entity)
 *                  re-evaluates the bundles the row occurs in and stores the row's gradient ONCE into m->grads (buffers of the
 *                  tables' shapes; only touched rows are written, nothing needs clearing); rows cut into several items leave
 *                  partial sums in `partials` (kge_own_partial_stride floats per slot).  The loss (incl. the regulariser value) is
 *                  added to the striped accumulators.
 *   kge_own_apply  phase 2: the dense-semantics optimiser, in place, on every row that has a gradient row / partial list.
 * `listed`: the compact index's bitmap (kge_pull_batch.dense_skip) or NULL when every row has an item.  dense == 0 (SGD / Adagrad:
 * a zero dense gradient leaves a row unchanged, so only touched rows are visited): an entity that was only DRAWN this step is
 * owned by the first pair that drew it (pc bit 27).  dense != 0 (Adam / RMSprop move every row every step): every unlisted row is
 * an implicit owner.  dim % 4 == 0, dim <= 512.
 * STAGED FORM (stage != NULL, the default of the Python trainer): kge_own_step first evaluates every bundle ONCE (k_own_eval: one
 * lane group per bundle) and stores the gradient rows it produces -- for its head, tail, relation and drawn entity -- in the bundle's
 * four slots of `stage`; the owners then only add the staged rows of their incidences (every staged row has exactly one reader)
 * instead of re-evaluating each bundle they occur in.  Outputs, kge_own_apply and all semantics are unchanged. */
int kge_own_groups_per_block(int32_t model, int32_t dim);
int kge_own_partial_stride(int32_t model, int32_t dim);
size_t kge_own_stage_bytes(int32_t model, int32_t dim, int64_t n_pairs);   /* the staged form's buffer */
int kge_own_step(const kge_model_desc* m, const int32_t* pairs, int64_t n_pairs, const kge_pull_lists* lists, const int32_t* items,
                 int64_t n_items, const uint32_t* listed, const int32_t* inc, float* partials, int32_t dense, float lmbda,
                 int32_t reg_type, int32_t reset_lists, const int32_t* next_pairs, const int32_t* next_inv, int64_t next_n,
                 const float* bern_prob, const uint64_t* slots, int64_t n_slots, uint64_t seed, uint64_t next_offset,
                 const kge_pull_lists* next_lists, float* loss,
                 float* stage /* NULL, or 4 * n_pairs * kge_own_partial_stride floats: the staged form (see below) */, void* stream);
int kge_own_apply(const kge_model_desc* m, float* const* state1, float* const* state2, const int32_t* pairs, int64_t n_pairs,
                  const kge_pull_lists* lists, const int32_t* items, int64_t n_items, const uint32_t* listed, const int32_t* multi,
                  int64_t n_multi, float* partials, int32_t dense, int32_t optimizer, float lr, int64_t step, void* stream);
/* A run of consecutive steps enqueued by ONE native call (two launches per step, no host work in between).
 * kge_own_run also takes ANALOGY, CP, SimplE, SimplE_ignr and QuatE (pointwise.py:241-387,461-768; hidden size <= 256; `stage`
 * required): csrc/kge_ownx.hip evaluates every triple once with the scorer's own forward / backward, stages one gradient row per role,
 * and the owner of an entity / relation adds the rows sourced from its id into one accumulator per table it holds, then applies the
 * optimiser in place. */
typedef struct kge_own_plan {
    kge_model_desc model;                 /* tables = the parameters (updated in place); grads = the gradient row buffers */
    float* state1[KGE_MAX_TABLES];        /* optimiser state per table (NULL where unused) */
    float* state2[KGE_MAX_TABLES];
    kge_pull_lists lists[2];              /* the two sampler list sets */
    const kge_pull_batch* batches;        /* HOST array of n_batches entries (dense_skip = the `listed` bitmap) */
    int64_t n_batches;
    float* partials;
    int32_t optimizer; float lr; float lmbda; int32_t reg_type;
    const float* bern_prob; const uint64_t* slots; int64_t n_slots; uint64_t seed;
    int64_t draws_per_batch;
    float* loss;
    float* stage;                         /* NULL: owners re-evaluate; else the staged form (4 * n_pairs * kge_own_partial_stride floats) */
} kge_own_plan;
size_t kge_own_plan_bytes(void);
int kge_own_run(const kge_own_plan* plan, int64_t first_batch, int64_t n_steps, int32_t cur_list, int32_t lists_ready,
                int64_t first_opt_step, uint64_t first_offset, int32_t sample_after_last, void* stream);

/* ---- The incidence index of kge_pull_step / kge_own_step, built ON THE DEVICE for n_batches batches at once (csrc/kge_index.hip).
 * The reference's per-run set-up of the batch feed is one permutation (data/generator.py:19-35: a batch is a fixed slice of it);
 * the index is this path's own per-run structure over those slices.  Batch b covers the pairs
 * triples[perm[b * batch_stride + slice_lo + i]], i < n_pairs (data-parallel ranks pass their slice of every batch).
 * Outputs, per batch b at fixed strides (device, int32):
 *   pairs  + b * n_pairs * 4          [n_pairs, 4]
 *   inc    + b * n_pairs * 3          [3 n_pairs]
 *   inv    + b * n_pairs * 3          [3 n_pairs]  inverse of inc: inv[3 * pair + role] = position of that incidence in inc
 *   items  + b * item_cap * 4         [item_cap, 4], the first counts[4 b] slots are live (the rest is padding, row -1)
 *   multi  + b * multi_cap * 4        [multi_cap, 4], the first counts[4 b + 1] rows are live
 *   skip   + b * words                bitmap of the listed rows (compact != 0 only: kge_pull_batch.dense_skip)
 *   counts + 4 b                      {item slots, multi rows, partial slots, listed rows}
 * item_cap / multi_cap / words / workspace bytes from kge_pull_index_geometry.  compact != 0 lists only rows with an incidence.
 * The layout rule is the one stated at pykg2vec_amd/generator.py::build_pull_batch (a numpy restatement used by tests and for
 * one-off explicit batches): both produce identical arrays.  Integer work only. */
int kge_pull_index_geometry(int64_t n_batches, int64_t n_pairs, int64_t tot_entity, int64_t tot_relation, int32_t segment,
                            int32_t groups_per_block, int32_t compact, int64_t* item_cap, int64_t* multi_cap, int64_t* words,
                            size_t* workspace_bytes);
int kge_pull_index_build(const int64_t* triples, const int64_t* perm, int64_t batch_stride, int64_t slice_lo, int64_t n_pairs,
                         int64_t n_batches, int64_t tot_entity, int64_t tot_relation, int32_t segment, int32_t groups_per_block,
                         int32_t compact, int32_t* pairs, int32_t* inc, int32_t* inv, int32_t* items, int32_t* multi, uint32_t* skip,
                         int32_t* counts, void* workspace, size_t workspace_bytes, void* stream);

/* ---- Atomic-free ("staged") training step for the long-row bundle kernels (RotatE self-adversarial; replaces the dense
 * gradient buffer + atomics of kge_train_pairwise_selfadv_sampled followed by kge_optimizer_step; same reference lines:
 * data/generator.py:42-97, utils/trainer.py:147-157,298-299, utils/criterion.py:20-23, utils/trainer.py:112-131).
 * The bundle kernel writes every gradient row it produces to its own slot of `stage` with plain stores:
 *   positive i      -> static slots  i*static_slots + site          (RotatE: h_re, h_im, r, t_re, t_im)
 *   negative pair p -> dynamic slots n_pos*static_slots + p*dynamic_slots + site   (RotatE: c_re, c_im), p = i*neg_rate + j,
 * and registers pair p with the entity it drew (dyn_count / dyn_bucket[E][dyn_cap] / overflow chain dyn_head (pair + 1; 0 = empty),
 * dyn_next[p]).
 * kge_optimizer_step_staged then owns one parameter row per wave: it sums the row's slots in ascending slot order --
 * static incidences from the per-batch CSR (ent_off/ent_inc: positive << 1 | side, side 0 head / 1 tail; rel_off/rel_inc:
 * positive), dynamic ones from the entity's bucket sorted by pair -- and applies the dense optimiser in place.  No gradient
 * buffer, no atomics on floats, bit-reproducible.  Rows with no slot get g = 0 (Adam / RMSprop still update them; SGD /
 * Adagrad leave them untouched, as torch.optim does for a zero dense gradient). */
typedef struct kge_staged_table {
    int32_t cls;            /* 0: entity table, 1: relation table */
    int32_t site_a, site_b; /* static slot sites: entity tables head / tail incidence, relation tables site_a */
    int32_t dsite;          /* dynamic slot site of an entity table (-1: the negatives never touch it) */
    int64_t flat_off;       /* float offset of the table in the flat parameter / state buffers */
    int64_t rows;
} kge_staged_table;

typedef struct kge_staged_step {
    float* param; float* state1; float* state2;        /* flat fp32 buffers (state: as kge_optimizer_step) */
    kge_staged_table tables[8];
    int32_t n_tables, dim;                              /* all tables share the row length */
    const int32_t* ent_off; const int32_t* ent_inc;     /* [E+1], [2 n_pos] */
    const int32_t* rel_off; const int32_t* rel_inc;     /* [R+1], [n_pos] */
    /* optional pre-reduction of long relation lists (a graph with a handful of relations funnels thousands of slots into
     * one row): relation r's list is cut into chunks of 16 slots, chunk ids [rel_chunk_off[r], rel_chunk_off[r+1]);
     * chunk_rel[c] = the relation of chunk c; one wave per (chunk, relation table) sums its slots into rel_partials
     * [n_chunks][#relation tables][stage_stride] and the optimiser sums a relation row's partials in chunk order.
     * rel_chunk_off == NULL: relation rows are summed slot by slot like entity rows. */
    const int32_t* rel_chunk_off; const int32_t* chunk_rel; float* rel_partials; int32_t n_chunks;
    int32_t* dyn_count; int32_t* dyn_bucket; int32_t* dyn_head; int32_t* dyn_next; int32_t dyn_cap;
    int32_t* dyn_count_next; int32_t* dyn_head_next;    /* the set the NEXT step registers into: cleared by the optimiser
                                                           sweep of this step (NULL: the train entry point memsets its own) */
    /* optional touched-row lists (SGD / Adagrad leave untouched rows alone, so their sweep only needs the rows with a slot):
     * touched_ent / touched_rel = the entities / relations of the batch's positives (sorted, unique; static per batch);
     * dyn_list[p] = the entity negative pair p drew if p was the first pair to register with it this step, else -1 (written
     * by the train entry point; needs the single-set form, dyn_count_next == NULL, which that entry point clears itself --
     * with one memset when dyn_count | dyn_head are laid out back to back).  touched_ent == NULL: every row of every table
     * is visited (always the case for Adam / RMSprop, which move every row every step). */
    const int32_t* touched_ent; int32_t n_touched_ent; const int32_t* touched_rel; int32_t n_touched_rel;
    int32_t* dyn_list;
    float* dyn_scale;                                   /* optional [n_neg]: factor applied to the dynamic slots of pair p when they are
                                                           summed (RotatE: the single-pass bundle kernel stages them relative to a running
                                                           softmax maximum and writes the normalisation here); NULL = 1 */
    float* stage; int64_t stage_stride;                 /* floats between slots (>= dim, multiple of 4) */
    int32_t static_slots, dynamic_slots;
    int64_t n_pos, n_neg;                               /* positives / negative pairs of the batch */
    int64_t tot_entity, tot_relation;
    int32_t stage_spare;                                /* round 6.  != 0: the caller padded stage_stride to the width the bundle's waves
                                                           cover (1 024 floats for rows of 513..1 024, 2 048 beyond) AND left static_slots +
                                                           dynamic_slots spare rows behind the n_pos * static_slots + n_neg * dynamic_slots
                                                           used ones: the RotatE bundle kernel then stores without predicates (lanes beyond
                                                           a row write its padding, a missing last bundle the spare rows).  0: tight rows */
} kge_staged_step;
size_t kge_staged_step_bytes(void);

/* kge_train_pairwise_selfadv_sampled with staged output (st->stage, st->dyn_*).  dyn_count / dyn_head must be clear on
 * entry: either the previous kge_optimizer_step_staged cleared them (dyn_count_next / dyn_head_next of ITS plan pointed at
 * them: two alternating sets) or, when st->dyn_count_next is NULL, this entry point memsets them itself. */
int kge_train_pairwise_selfadv_sampled_staged(const kge_model_desc* m, const int64_t* triples, const int64_t* perm,
                                              int64_t start, int64_t n_pos, int32_t neg_rate, float alpha,
                                              const float* bern_prob, const uint64_t* slots, int64_t n_slots,
                                              uint64_t seed, uint64_t offset, const kge_staged_step* st, float* loss,
                                              void* stream);
/* kge_train_pointwise_logistic_sampled with staged output, DistMult (3 static + 1 dynamic slots: h, r, t | c) and ComplEx /
 * ComplexN3 (6 + 2: h_re, h_im, r_re, r_im, t_re, t_im | c_re, c_im).  Same clearing contract as above. */
int kge_train_pointwise_logistic_sampled_staged(const kge_model_desc* m, const int64_t* triples, const int64_t* perm,
                                                int64_t start, int64_t n_pos, int32_t neg_rate, const float* bern_prob,
                                                const uint64_t* slots, int64_t n_slots, uint64_t seed, uint64_t offset,
                                                float lmbda, int32_t reg_type, const kge_staged_step* st, float* loss,
                                                void* stream);
int kge_optimizer_step_staged(int32_t optimizer, const kge_staged_step* st, float lr, int64_t step, void* stream);

/* ---- 1-N scoring head of the projection models (ConvE / TuckER / InteractE / HypER / AcrE:
 * projection.py:100-102, 335-336, 444-447, 606-609, 734-737):  preds[B,E] = sigmoid(x[B,dim] @ ent[E,dim]^T + bias[E]).
 * bias may be NULL (TuckER).  fp32 on the matrix cores. */
int kge_head_1n_forward(const float* x, int64_t batch, int32_t dim, const float* ent, int64_t tot_entity,
                        const float* bias, float* preds, void* stream);
/* The same forward with the operands rounded to bfloat16 (round to nearest even) on their way into LDS and the products
 * accumulated in fp32 on v_mfma_f32_32x32x16_bf16 (SURVEY 8(f) rank 4 names a bf16 option).  Inputs and outputs stay fp32.  Opt-in:
 * the reference computes the head in fp32; a logit differs from the fp32 one by about 2^-8 relative per operand. */
int kge_head_1n_forward_bf16(const float* x, int64_t batch, int32_t dim, const float* ent, int64_t tot_entity, const float* bias,
                             float* preds, void* stream);

/* Filtered rank of the head WITHOUT the [B, E] prediction tensor: what the projection models' evaluation does with the head's output
 * (projection.py:119-125: predict_tail_rank / predict_head_rank = topk(-forward(...)), consumed by MetricCalculator.get_tail_rank /
 * get_head_rank, utils/evaluator.py:70-123): for row i, rank = #{e : p_ie > p_i,truth_i} over all E entities and the filtered rank
 * subtracts the entities of the row's known list (CSR off int64 [B+1] / ids int32; NULL = no filter) that outrank the true one.
 * Runs the rank sweep of kge_eval_ranks (candidate tiles x query tiles, count epilogue, matrix cores from 512 rows on) over the
 * candidate rows [ent row | bias] and the query rows [x row | 1]; energy = -sigmoid(logit), the sigmoid being the head's own
 * expression, so ties of saturated predictions are ties here too (ties[i] = candidates whose prediction equals the true one's bit
 * for bit, the true one excluded; may be NULL).  triples: int64 [B, 3], column 2 = the true entity of row i (columns 0 / 1: any
 * valid ids, not read).  ranks: int32 [2, B] = rank, filtered rank (0-based).  energies (optional, tests): float [B, E], the
 * sweep's -p values instead of ranks (ranks / ties are then not written).  workspace: kge_head_1n_rank_workspace_bytes. */
size_t kge_head_1n_rank_workspace_bytes(int64_t batch, int32_t dim, int64_t tot_entity, int32_t has_bias);
int kge_head_1n_rank(const float* x, int64_t batch, int32_t dim, const float* ent, int64_t tot_entity, const float* bias,
                     const int64_t* triples, const int64_t* off, const int32_t* ids, void* workspace, size_t workspace_bytes,
                     int32_t* ranks, int32_t* ties, float* energies, void* stream);

/* Autograd backward of the head given d loss / d preds: dx[B,dim] is overwritten, g_ent[E,dim] and g_bias[E] are
 * accumulated into (any of the three may be NULL).  workspace (kge_head_1n_backward_workspace_bytes(), a constant; may be NULL):
 * room for the partial tiles of the split-K products, which are then added in split order -- bit-reproducible gradients; without it
 * the partial tiles are combined with float atomics. */
size_t kge_head_1n_backward_workspace_bytes(void);
int kge_head_1n_backward(const float* x, int64_t batch, int32_t dim, const float* ent, int64_t tot_entity,
                         const float* preds, const float* dpreds, float* dx, float* g_ent, float* g_bias,
                         void* workspace, size_t workspace_bytes, void* stream);

/* One direction of Criterion.multi_class_bce (utils/criterion.py:41-49) fused with the head and its backward:
 * loss += mean_{B*E} BCEWithLogits(preds, y)  -- the reference applies the logits loss to the SIGMOID OUTPUTS, kept --
 * with y the multi-hot rows given as CSR (label_off int64 [B+1], label_ids int32 [n_pos]: hr_t / tr_h of the batch,
 * data/generator.py:160-213) and, when label_smoothing >= 0, y <- y (1 - ls) + 1/E.  The [B,E] predictions never reach
 * HBM; the [B,E] logit gradient lives in the workspace (kge_head_1n_bce_workspace_bytes).  dx overwritten, g_ent /
 * g_bias accumulated, loss in the striped accumulators used by the other train entry points. */
size_t kge_head_1n_bce_workspace_bytes(int64_t batch, int64_t tot_entity, int64_t n_pos);
int kge_head_1n_bce(const float* x, int64_t batch, int32_t dim, const float* ent, int64_t tot_entity, const float* bias,
                    const int64_t* label_off, const int32_t* label_ids, int64_t n_pos, float label_smoothing,
                    void* workspace, size_t workspace_bytes, float* loss, float* dx, float* g_ent, float* g_bias,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KGE_HIP_H */
