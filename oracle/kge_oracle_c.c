/* kge_oracle_c.c -- multi-threaded C restatement of the BENCH workload only (TransE): one reference train step
 * (utils/trainer.py:147-157 train_step_pairwise + utils/criterion.py:25-29 pairwise_hinge + loss.backward() with dense
 * nn.Embedding gradients + torch.optim.Adam defaults, utils/trainer.py:112-116,298-299) and the filtered-rank evaluation
 * of Evaluator.test / MetricCalculator (utils/evaluator.py:70-123,249-334) in count form.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT: it exists so that bench.py's `cpu_baseline` uses every host core (OpenMP) instead of
 * single-threaded numpy.  It is held to the numpy oracle (oracle/kge_oracle.py, itself pinned to the live reference's
 * golden vectors) by tests/test_oracle_c.py.  Only tests/ and bench.py's cpu_baseline leg load it.
 *
 * Build (done by __graft_entry__.build()):  gcc -O3 -fopenmp -shared -fPIC kge_oracle_c.c -o _build/libkge_oracle_c.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define EPS_NORMALIZE 1e-12f

void kgec_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#endif
}

int kgec_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* x^ = x / max(||x||, eps)  (F.normalize, pairwise.py:69-71); returns the clamped norm */
static float normalize_row(const float* x, int d, float* out) {
    float n2 = 0.f;
    for (int i = 0; i < d; ++i) n2 += x[i] * x[i];
    float den = sqrtf(n2);
    if (den < EPS_NORMALIZE) den = EPS_NORMALIZE;
    for (int i = 0; i < d; ++i) out[i] = x[i] / den;
    return den;
}

/* energy ||h^ + r^ - t^||_{1|2} and, when g* != NULL, ds * d(energy)/d{h,r,t} added into g* (pairwise.py:56-76) */
static float transe_triple(const float* h, const float* r, const float* t, int d, int l1, float ds, float* gh, float* gr,
                           float* gt, float* scratch) {
    float* hh = scratch; float* rh = scratch + d; float* th = scratch + 2 * d; float* g = scratch + 3 * d;
    const float nh = normalize_row(h, d, hh), nr = normalize_row(r, d, rh), nt = normalize_row(t, d, th);
    float s = 0.f;
    for (int i = 0; i < d; ++i) {
        const float u = hh[i] + rh[i] - th[i];
        g[i] = u;
        s += l1 ? fabsf(u) : u * u;
    }
    if (!l1) s = sqrtf(s);
    if (gh) {
        float dh = 0.f, dr = 0.f, dt = 0.f;
        for (int i = 0; i < d; ++i) {
            const float u = g[i];
            const float gi = l1 ? (u > 0.f ? ds : (u < 0.f ? -ds : 0.f)) : (s > 0.f ? ds * u / s : 0.f);
            g[i] = gi;
            dh += hh[i] * gi; dr += rh[i] * gi; dt += th[i] * gi;
        }
        for (int i = 0; i < d; ++i) {
            gh[i] += (g[i] - hh[i] * dh) / nh;
            gr[i] += (g[i] - rh[i] * dr) / nr;
            gt[i] -= (g[i] - th[i] * dt) / nt;
        }
    }
    return s;
}

/* One train step: hinge loss (sum) over n (pos, neg) pairs, dense gradients, dense Adam update of both tables.
 * ent [E,d], rel [R,d]; m_*, v_* Adam moments; g_* dense gradient scratch (zeroed here).  Returns the loss. */
float kgec_transe_adam_step(float* ent, float* rel, int64_t E, int64_t R, int d, int l1, float margin, const int64_t* ph,
                            const int64_t* pr, const int64_t* pt, const int64_t* nh, const int64_t* nr, const int64_t* nt,
                            int64_t n, float lr, int64_t step, float* g_ent, float* g_rel, float* m_ent, float* v_ent,
                            float* m_rel, float* v_rel) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < E * d; ++i) g_ent[i] = 0.f;
    memset(g_rel, 0, (size_t)R * d * sizeof(float));
    double loss = 0.0;
    /* pairs are independent given the tables; gradient rows collide, so each thread owns a slice of ROWS:
       pass 1 computes per-pair coefficients in parallel, pass 2 lets every thread scan the pairs and apply only the
       contributions that land in its row slice (no atomics, deterministic) */
    float* coef = (float*)malloc((size_t)n * sizeof(float));
#pragma omp parallel
    {
        float* scratch = (float*)malloc((size_t)4 * d * sizeof(float));
#pragma omp for reduction(+ : loss) schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            const float sp = transe_triple(ent + ph[i] * d, rel + pr[i] * d, ent + pt[i] * d, d, l1, 0.f, 0, 0, 0, scratch);
            const float sn = transe_triple(ent + nh[i] * d, rel + nr[i] * d, ent + nt[i] * d, d, l1, 0.f, 0, 0, 0, scratch);
            const float v = sp + margin - sn;
            coef[i] = v > 0.f ? 1.f : (v == 0.f ? 0.5f : 0.f);
            if (v > 0.f) loss += v;
        }
#ifdef _OPENMP
        const int tid = omp_get_thread_num(), nth = omp_get_num_threads();
#else
        const int tid = 0, nth = 1;
#endif
        float* th = (float*)malloc((size_t)3 * d * sizeof(float));
        for (int64_t i = 0; i < n; ++i) {
            if (coef[i] == 0.f) continue;
            for (int side = 0; side < 2; ++side) {
                const int64_t h = side ? nh[i] : ph[i], r = side ? nr[i] : pr[i], t = side ? nt[i] : pt[i];
                const int oh = (int)(h % nth) == tid, orr = (int)(r % nth) == tid, ot = (int)(t % nth) == tid;
                if (!(oh || orr || ot)) continue;
                memset(th, 0, (size_t)3 * d * sizeof(float));
                transe_triple(ent + h * d, rel + r * d, ent + t * d, d, l1, side ? -coef[i] : coef[i], th, th + d, th + 2 * d,
                              scratch);
                if (oh) for (int c = 0; c < d; ++c) g_ent[h * d + c] += th[c];
                if (orr) for (int c = 0; c < d; ++c) g_rel[r * d + c] += th[d + c];
                if (ot) for (int c = 0; c < d; ++c) g_ent[t * d + c] += th[2 * d + c];
            }
        }
        free(th);
        free(scratch);
    }
    free(coef);
    /* torch.optim.Adam defaults, dense over every row */
    const double bc1 = 1.0 - pow(0.9, (double)step), bc2 = 1.0 - pow(0.999, (double)step);
    const float step_size = (float)(lr / bc1), bc2s = (float)sqrt(bc2);
    for (int tbl = 0; tbl < 2; ++tbl) {
        float* p = tbl ? rel : ent; float* g = tbl ? g_rel : g_ent; float* m = tbl ? m_rel : m_ent; float* v = tbl ? v_rel : v_ent;
        const int64_t tot = (tbl ? R : E) * d;
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < tot; ++i) {
            m[i] = m[i] + (1.0f - 0.9f) * (g[i] - m[i]);
            v[i] = v[i] * 0.999f + (1.0f - 0.999f) * g[i] * g[i];
            const float denom = sqrtf(v[i]) / bc2s + 1e-8f;
            p[i] = p[i] + (-step_size) * m[i] / denom;
        }
    }
    return (float)loss;
}

/* Filtered ranks of n test triples: for each, the tail sweep (h,r,?) and the head sweep (?,r,t) over all E entities,
 * rank = #{e: s_e < s_true}, filtered rank skips known entities (CSR lists).  ranks: int32 [4,n] = head, tail, fhead, ftail */
void kgec_transe_eval(const float* ent, const float* rel, int64_t E, int d, int l1, const int64_t* triples, int64_t n,
                      const int64_t* tail_off, const int32_t* tail_ids, const int64_t* head_off, const int32_t* head_ids,
                      int32_t* ranks) {
    float* entn = (float*)malloc((size_t)E * d * sizeof(float));
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < E; ++e) normalize_row(ent + e * d, d, entn + e * d);
#pragma omp parallel
    {
        float* q = (float*)malloc((size_t)2 * d * sizeof(float));
        float* sc = (float*)malloc((size_t)E * sizeof(float));
#pragma omp for schedule(dynamic, 4)
        for (int64_t i = 0; i < n; ++i) {
            const int64_t h = triples[3 * i], r = triples[3 * i + 1], t = triples[3 * i + 2];
            float* rn = q + d;
            normalize_row(rel + r * d, d, rn);
            for (int side = 0; side < 2; ++side) {  /* 0: tail sweep, 1: head sweep */
                const float* fixed = entn + (side ? t : h) * d;
                for (int c = 0; c < d; ++c) q[c] = side ? (rn[c] - fixed[c]) : (fixed[c] + rn[c]);
                for (int64_t e = 0; e < E; ++e) {
                    const float* ce = entn + e * d;
                    float s = 0.f;
                    if (side == 0) { for (int c = 0; c < d; ++c) { const float u = q[c] - ce[c]; s += l1 ? fabsf(u) : u * u; } }
                    else { for (int c = 0; c < d; ++c) { const float u = ce[c] + q[c]; s += l1 ? fabsf(u) : u * u; } }
                    sc[e] = l1 ? s : sqrtf(s);
                }
                const int64_t truth = side ? h : t;
                const float st = sc[truth];
                int32_t rank = 0, fc = 0;
                for (int64_t e = 0; e < E; ++e) rank += sc[e] < st;
                const int64_t* off = side ? head_off : tail_off;
                const int32_t* ids = side ? head_ids : tail_ids;
                if (off) for (int64_t j = off[i]; j < off[i + 1]; ++j) fc += (ids[j] != truth && sc[ids[j]] < st);
                ranks[(side ? 0 : 1) * n + i] = rank;
                ranks[(side ? 2 : 3) * n + i] = rank - fc;
            }
        }
        free(q);
        free(sc);
    }
    free(entn);
}
