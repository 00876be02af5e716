// kge_eval.hip -- filtered-rank evaluation as a tiled sweep over the entity table.
//
// Replaces, per test triple, Evaluator.test_tail_rank / test_head_rank (utils/evaluator.py:249-273: build [E]
// id tensors, forward over all E candidates, torch.topk(k=E) i.e. a full sort, D2H copy of E int64) and
// MetricCalculator.get_tail_rank / get_head_rank (utils/evaluator.py:70-123: python scan of the ordering with
// set lookups) by
//     rank(q)          = #{ e : s(q,e) < s(q,true) }
//     filtered rank(q) = rank(q) - #{ e in known(q), e != true : s(q,e) < s(q,true) }
// No score vector, no ordering, no sort ever exists in memory.
//
// Pipeline (all on one stream, no host sync):
//   1. k_eval_prepare   entity table(s) -> *sweep layout* cand[tile][k][64]: candidate e = 64*tile + lane, element
//                        k contiguous over lanes, so lane <-> candidate and every load is a coalesced 256-B line.
//                        Model-specific candidate-side algebra that does not depend on the query is folded in here
//                        (TransE: row normalisation; TransD: the scalar e.e_m; ComplEx/RotatE/ANALOGY: re|im
//                        concatenation), once per evaluation instead of once per (query, candidate).
//   2. k_eval_queries   per test triple the two query vectors: the (h,r) side pre-contracted for the tail sweep,
//                        the (r,t) side for the head sweep, so a sweep is one of 4 pair forms
//                        L1 / L2 / squared-L2-minus-margin distance or negated dot product.
//   3. k_eval_target_filter   s(q,true) and the filtered count over the CSR list of known entities.
//   4. k_eval_sweep     QT queries x 64 candidates per wave: candidate chunk in VGPRs, query elements as
//                        wave-uniform scalar operands (s_load), QT accumulators per lane, ballot+popcount.
//                        Each candidate tile read from L2/HBM is reused by QT queries.
//   5. k_eval_finalize  int32 ranks [4, n].
// Steps 3 and 4 run the SAME per-lane sequential accumulation over k, so s(q,e) is bit-identical in both and
// the integer ranks are exact functions of the fp32 scores.
//
// Roofline: algorithmic bytes = candidate row bytes per scored candidate (400 B for TransE d=100); with QT-fold
// reuse the kernel is VALU-bound (2 lane-ops per element per pair for L1), not HBM-bound.
#include "kge_internal.h"
#include <stdlib.h>
#include <string.h>

namespace kge {

// Summation order of the matrix-core sweep's energies (round 6).  An energy used to be ONE in-order fmaf chain over all of k (what
// back-to-back MFMAs on one accumulator compute): its rounding error grows like K u sigma / sqrt(2), ~15x the error of ATen's blocked
// reduction at K = 400, and on the 512-triple C2 fixture float64 sided with the reference on every rank the HIP path differed in.
// With KGE_GEMM_CHUNK = C > 0 the chain restarts every C elements of k and the finished chunk is added to a running total
// (`tot += acc; acc = 0`: two accumulator sets on the matrix cores whatever the chunk count).  Emulated on the host
// (tools/rank_chain_study.py -> profiles/r06_rank_chain_study.json): 6 -> 0..1 differing ranks of 1 024 at C2, energy error 3x
// smaller.  k_eval_target_filter_chain follows the same chunk boundaries, so ranks stay exact functions of bit-identical energies.
#ifndef KGE_GEMM_CHUNK
#define KGE_GEMM_CHUNK 64
#endif
constexpr int kGemmChunk = KGE_GEMM_CHUNK;   // 0 = one chain over all of k (rounds 2-5); else a multiple of the K slab
constexpr int KC = 8;        // k-chunk held in VGPRs
constexpr int QT_PLAIN = 16; // queries per wave pass, plain forms
constexpr int QT_XF = 8;     // ... candidate-transform forms (TransH / TransD)

enum Form { F_L1 = 0, F_L2 = 1, F_SQM = 2, F_NEGDOT = 3 };
enum XForm { X_NONE = 0, X_TRANSH = 1, X_TRANSD = 2 };
// per-query post-op on the finished energy: TransM multiplies by theta_r (pairwise.py:341-347), SimplE clamps to
// [-20, 20] (pointwise.py:525-526)
enum Post { P_NONE = 0, P_SCALE = 1, P_CLAMP = 2, P_SIGMOID = 3 };   // P_SIGMOID: the 1-N head (energy = -sigmoid(logit), kge_head_1n_rank)

struct EvalPlan {
    int64_t nq;   // query rows the sweep walks: 2n (tail + head sweep per triple), n for a one-sided sweep
    int only;     // 2 = both sides (query row 2i + side); 0 / 1 = one-sided (query row i = triple i's tail / head sweep)
    int64_t E, n, tables, table_stride;  // tables > 1: one projected candidate table per relation group (TransR)
    int K, Kpad, QV, form, xform, post;
    int64_t ntiles;
    float* cand; float* aux; float* qvec; float* qscale; float* st; int32_t* fcount; int32_t* rcount; int32_t* tcount;
    float* qT;   // dot-product forms: the queries as k-major 128-wide tiles (matrix-core sweep)
    size_t bytes;
};

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static int sweep_K(const kge_model_desc* m) {
    switch (m->model) {
        case KGE_COMPLEX: case KGE_ROTATE: case KGE_ANALOGY: return 2 * m->dim;
        case KGE_CP: case KGE_SIMPLE: case KGE_SIMPLE_IGNR: return 2 * m->dim;  // both entity tables side by side
        case KGE_QUATE: return 4 * m->dim;
        case KGE_TRANSR: return m->rel_dim;  // candidates live in the relation space of the call's relation
        case KGE_HEAD_1N_INTERNAL: return m->dim + (m->tables[1] ? 1 : 0);   // [ent row | bias] . [x | 1]
        default: return m->dim;
    }
}

static bool make_plan(const kge_model_desc* m, int64_t n, void* ws, EvalPlan* p, int64_t tables = 1) {
    p->E = m->tot_entity; p->n = n; p->nq = 2 * n; p->only = 2; p->tables = tables;
    const bool per_group_tables = tables > 1 || m->model == KGE_TRANSR;  // candidates already transformed per relation group
    p->K = sweep_K(m);
    p->Kpad = (p->K + KC - 1) / KC * KC;
    p->xform = per_group_tables ? X_NONE : m->model == KGE_TRANSH ? X_TRANSH : m->model == KGE_TRANSD ? X_TRANSD : X_NONE;
    p->QV = p->xform == X_NONE ? 1 : 2;
    switch (m->model) {
        case KGE_TRANSE: case KGE_TRANSH: case KGE_TRANSD: case KGE_TRANSM: case KGE_TRANSR:
            p->form = (m->flags & KGE_FLAG_L1) ? F_L1 : F_L2; break;
        case KGE_ROTATE: p->form = F_SQM; break;
        case KGE_DISTMULT: case KGE_COMPLEX: case KGE_ANALOGY: case KGE_RESCAL:
        case KGE_CP: case KGE_SIMPLE: case KGE_SIMPLE_IGNR: case KGE_QUATE: case KGE_HEAD_1N_INTERNAL: p->form = F_NEGDOT; break;
        default: return false;
    }
    p->post = m->model == KGE_TRANSM ? P_SCALE : (m->model == KGE_SIMPLE || m->model == KGE_SIMPLE_IGNR) ? P_CLAMP
              : m->model == KGE_HEAD_1N_INTERNAL ? P_SIGMOID : P_NONE;
    p->ntiles = (p->E + 63) / 64;
    // the dot-product forms (matrix-core sweep): k padded to whole 16-deep slabs (zeros add nothing to a chain) and one spare 64-candidate
    // tile behind the tables, so that k_eval_gemm's operand loads need neither a k predicate nor an odd-tile-count select (round 6)
    const bool dot_form = (p->form == F_NEGDOT || p->form == F_SQM) && p->xform == X_NONE;
    if (dot_form) p->Kpad = (p->K + 15) / 16 * 16;
    size_t off = 0;
    char* base = (char*)ws;
    auto take = [&](size_t bytes) { char* q = base ? base + off : nullptr; off += align256(bytes); return q; };
    p->table_stride = p->ntiles * p->Kpad * 64;
    p->cand = (float*)take(((size_t)tables * p->table_stride + (dot_form ? (size_t)p->Kpad * 64 : 0)) * sizeof(float));
    p->aux = (float*)take((size_t)p->ntiles * 64 * sizeof(float));
    p->qvec = (float*)take((size_t)2 * n * p->QV * p->Kpad * sizeof(float));
    p->qscale = (float*)take((size_t)2 * n * sizeof(float));
    p->st = (float*)take((size_t)2 * n * sizeof(float));
    p->fcount = (int32_t*)take((size_t)2 * n * sizeof(int32_t));
    p->rcount = (int32_t*)take((size_t)4 * n * sizeof(int32_t));   // [2n] candidates strictly below the target, then
    p->tcount = p->rcount ? p->rcount + 2 * n : nullptr;           // [2n] candidates whose energy EQUALS the target's (itself included)
    p->qT = nullptr;
    if ((p->form == F_NEGDOT || p->form == F_SQM) && p->xform == X_NONE)
        p->qT = (float*)take((size_t)((2 * n + 127) / 128) * p->Kpad * 128 * sizeof(float));
    p->bytes = off;
    return true;
}

size_t eval_workspace_bytes(const kge_model_desc* m, int64_t n, int64_t tables) {
    if (m->model == KGE_NTN) return ntn_eval_workspace_bytes(m, n);
    EvalPlan p;
    if (!make_plan(m, n, nullptr, &p, tables)) return 0;
    return p.bytes;
}

// ------------------------------------------------------------------ 1. candidate sweep layout
struct PrepArgs {
    const float* seg[4]; int seg_dim[4]; int nseg;  // candidate row = concatenation of table rows
    const float* dot_tab;                            // TransD: aux[e] = ent[e] . ent_mappings[e]
    int normalize;                                   // TransE: divide by max(||row||, eps)
    int want_n2;                                     // matrix-core squared-distance sweep: aux[e] = sum_k c_k^2
    int64_t E; int K, Kpad;
    int kper;                                        // K range of one workgroup (multiple of 64; gridDim.y of them per tile)
};

__device__ __forceinline__ float prep_elem(const PrepArgs& a, int64_t e, int k) {
    int kk = k;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (s < a.nseg) {
            if (kk < a.seg_dim[s]) return a.seg[s][e * a.seg_dim[s] + kk];
            kk -= a.seg_dim[s];
        }
    }
    return 0.f;
}

// address of element k of candidate row e (nullptr beyond the last table), without control flow: many of these are issued
// back to back by k_eval_target_filter_chain
__device__ __forceinline__ const float* prep_addr(const PrepArgs& a, int64_t e, int k) {
    const float* ptr = nullptr;
    int kk = k;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (s < a.nseg) {
            const bool in = kk >= 0 && kk < a.seg_dim[s];
            ptr = in ? a.seg[s] + e * a.seg_dim[s] + kk : ptr;
            kk -= a.seg_dim[s];
        }
    }
    return ptr;
}

// one block = one tile of 64 candidates; 4 waves; each wave owns 16 candidates for the row reductions
__global__ __launch_bounds__(256) void k_eval_prepare(PrepArgs a, float* __restrict__ cand, float* __restrict__ aux) {
    __shared__ float s_scale[64];
    __shared__ float s_tile[64][65];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tile = blockIdx.x;
    const int64_t e0 = tile * 64;
    // Row statistics.  Only the TransE normalisation must be known BEFORE the rows are written (it scales them): that
    // case takes a pass of its own over the (short) rows.  Squared norms / the TransD dot product are by-products: they
    // accumulate per lane during the transposing pass below -- one read of the table, not two (RotatE d=1000: 116 MB).
    if (a.normalize) {
        for (int j = 0; j < 16; ++j) {
            const int row = wave * 16 + j;
            const int64_t e = e0 + row;
            float n2 = 0.f;
            if (e < a.E)
                for (int k = lane; k < a.K; k += 64) { const float v = prep_elem(a, e, k); n2 = fmaf(v, v, n2); }
            n2 = wave_sum(n2);
            if (lane == 0) s_scale[row] = 1.0f / fmaxf(sqrtf(n2), kEpsNormalize);
        }
    } else if (threadIdx.x < 64) {
        s_scale[threadIdx.x] = 1.0f;
    }
    __syncthreads();
    float n2acc[16], dtacc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { n2acc[j] = 0.f; dtacc[j] = 0.f; }
    // long rows, few tiles (RotatE d=1000 on 14 541 entities: 228 tiles x 32 chunks each): the K range is split over
    // gridDim.y workgroups so that the chip is full; the row statistics then come from k_eval_cnorm (fixed summation order)
    const int k_lo = blockIdx.y * a.kper, k_hi = min(a.Kpad, k_lo + a.kper);
    for (int k0 = k_lo; k0 < k_hi; k0 += 64) {  // transpose 64x64 through LDS: coalesced reads AND writes
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int row = wave * 16 + j;
            const int64_t e = e0 + row;
            const int k = k0 + lane;
            float v = 0.f;
            if (e < a.E && k < a.K) {
                const float raw = prep_elem(a, e, k);
                if (a.want_n2) n2acc[j] = fmaf(raw, raw, n2acc[j]);
                if (a.dot_tab) dtacc[j] = fmaf(raw, a.dot_tab[e * a.K + k], dtacc[j]);
                v = raw * s_scale[row];
            }
            s_tile[row][lane] = v;
        }
        __syncthreads();
        for (int j = 0; j < 16; ++j) {
            const int kk = wave * 16 + j;
            if (k0 + kk < a.Kpad) cand[(tile * a.Kpad + k0 + kk) * 64 + lane] = s_tile[lane][kk];
        }
        __syncthreads();
    }
    if ((a.want_n2 || a.dot_tab) && gridDim.y == 1) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float tot = wave_sum(a.dot_tab ? dtacc[j] : n2acc[j]);
            if (lane == 0) aux[e0 + wave * 16 + j] = (e0 + wave * 16 + j < a.E) ? tot : 0.f;
        }
    }
}

// |c|^2 of every candidate row when the transposing pass is split over K: one wave per candidate, lanes stride k
__global__ __launch_bounds__(256) void k_eval_cnorm(PrepArgs a, float* __restrict__ aux) {
    const int lane = threadIdx.x & 63;
    const int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // grid covers the padded rows of the last tile too
    if (e >= a.E) { if (lane == 0) aux[e] = 0.f; return; }
    float n2 = 0.f;
    for (int k = lane; k < a.K; k += 64) { const float v = prep_elem(a, e, k); n2 = fmaf(v, v, n2); }
    n2 = wave_sum(n2);
    if (lane == 0) aux[e] = n2;
}

// The plain re-layout (no normalisation, no TransD dot product, every table width a multiple of 4) with 16-byte accesses:
// a workgroup moves 64 candidates x 128 k per step -- 512 contiguous bytes per row and half wave on the way in, float4 of
// four candidates per k on the way out -- and grid.y splits K so that the chip is full whatever the tile count is.  |c|^2
// (squared-distance sweep) is a by-product: per (row, K split) partial sums, added up in split order by k_eval_cnorm_parts
// (part == nullptr with one split: aux is written here), so the table is read once.
__device__ __forceinline__ float4 prep_elem4(const PrepArgs& a, int64_t e, int k) {
    int kk = k;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (s < a.nseg) {
            if (kk < a.seg_dim[s]) return *reinterpret_cast<const float4*>(a.seg[s] + e * a.seg_dim[s] + kk);
            kk -= a.seg_dim[s];
        }
    }
    return make_float4(0.f, 0.f, 0.f, 0.f);
}

__global__ __launch_bounds__(256) void k_eval_prepare4(PrepArgs a, float* __restrict__ cand, float* __restrict__ aux,
                                                       float* __restrict__ part) {
    __shared__ float s_tile[64][129];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, kq = lane & 31;
    const int64_t tile = blockIdx.x, e0 = tile * 64;
    const int k_lo = blockIdx.y * a.kper, k_hi = min(a.Kpad, k_lo + a.kper);
    float n2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) n2[j] = 0.f;
    for (int k0 = k_lo; k0 < k_hi; k0 += 128) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t e = e0 + wave * 16 + 2 * j + half;
            const int k = k0 + 4 * kq;
            v[j] = (e < a.E && k < a.K) ? prep_elem4(a, e, k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float* row = s_tile[wave * 16 + 2 * j + half] + 4 * kq;
            row[0] = v[j].x; row[1] = v[j].y; row[2] = v[j].z; row[3] = v[j].w;
            n2[j] = fmaf(v[j].x, v[j].x, n2[j]); n2[j] = fmaf(v[j].y, v[j].y, n2[j]);
            n2[j] = fmaf(v[j].z, v[j].z, n2[j]); n2[j] = fmaf(v[j].w, v[j].w, n2[j]);
        }
        __syncthreads();
        const int c4 = threadIdx.x & 15;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int kk = it * 16 + (threadIdx.x >> 4);
            if (k0 + kk < k_hi) {
                const float4 o = make_float4(s_tile[4 * c4][kk], s_tile[4 * c4 + 1][kk], s_tile[4 * c4 + 2][kk], s_tile[4 * c4 + 3][kk]);
                *reinterpret_cast<float4*>(cand + (tile * a.Kpad + k0 + kk) * 64 + 4 * c4) = o;
            }
        }
        __syncthreads();
    }
    if (a.want_n2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float t = n2[j];
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) t += __shfl_xor(t, o, 64);   // the 32 lanes of the row's half wave, fixed order
            if (kq == 0) {
                const int64_t e = e0 + wave * 16 + 2 * j + half;
                if (part) part[e * gridDim.y + blockIdx.y] = t;
                else aux[e] = t;
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_eval_cnorm_parts(const float* __restrict__ part, int nky, int64_t rows, float* __restrict__ aux) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= rows) return;
    float t = 0.f;
    for (int y = 0; y < nky; ++y) t += part[e * nky + y];
    aux[e] = t;
}

// ---- per-relation candidate tables for TransH / TransD (grouped evaluation): the candidate-side transform depends on the
// query only through its relation (hyperplane normal w_r / mapping r_m), so for a group of queries sharing r it is applied
// ONCE per candidate -- project, normalise, write the sweep layout -- and the queries then run the plain L1 / L2 sweep
// (1.5 VALU issues per element pair instead of ~7 with the in-sweep transform).  grid: (tiles, groups).
struct XfPrepArgs {
    const float* ent; const float* vec_tab; const float* ent_map; const int64_t* group_rel;
    int xform; int64_t E; int K, Kpad; int64_t table_stride;
};

__global__ __launch_bounds__(256) void k_eval_prepare_xf(XfPrepArgs a, float* __restrict__ cand) {
    __shared__ float s_p[64], s_scale[64];
    __shared__ float s_tile[64][65];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tile = blockIdx.x, e0 = tile * 64;
    const float* vec = a.vec_tab + a.group_rel[blockIdx.y] * (int64_t)a.K;
    cand += blockIdx.y * a.table_stride;
    float iw = 1.f;
    if (a.xform == X_TRANSH) {  // w^ = w / max(|w|, eps)   (pairwise.py:176-180)
        float nw = 0.f;
        for (int k = lane; k < a.K; k += 64) nw = fmaf(vec[k], vec[k], nw);
        iw = 1.0f / fmaxf(sqrtf(wave_sum(nw)), kEpsNormalize);
    }
    if (a.K <= 256) {
        // rows of up to 256 floats: the wave's 16 rows live in registers (4 floats per lane each) from the gather to the
        // transposed store -- one read of the entity table per relation group
        float vh[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { const int k = lane + 64 * c; vh[c] = k < a.K ? vec[k] * iw : 0.f; }
        float v[16][4];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int64_t e = e0 + wave * 16 + j;
#pragma unroll
            for (int c = 0; c < 4; ++c) { const int k = lane + 64 * c; v[j][c] = (e < a.E && k < a.K) ? a.ent[e * a.K + k] : 0.f; }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int64_t e = e0 + wave * 16 + j;
            float dt = 0.f;
            if (a.xform == X_TRANSH) {
#pragma unroll
                for (int c = 0; c < 4; ++c) dt = fmaf(v[j][c], vh[c], dt);
                dt = -wave_sum(dt);                                  // e - (e . w^) w^
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) { const int k = lane + 64 * c; dt = fmaf(v[j][c], (e < a.E && k < a.K) ? a.ent_map[e * a.K + k] : 0.f, dt); }
                dt = wave_sum(dt);                                   // e + (e . e_m) r_m   (pairwise.py:229-251)
            }
            float n2 = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) { v[j][c] = fmaf(dt, vh[c], v[j][c]); n2 = fmaf(v[j][c], v[j][c], n2); }
            const float sc = 1.0f / fmaxf(sqrtf(wave_sum(n2)), kEpsNormalize);
#pragma unroll
            for (int c = 0; c < 4; ++c) v[j][c] *= sc;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k0 = 64 * c;
            if (k0 >= a.Kpad) break;  // block-uniform
#pragma unroll
            for (int j = 0; j < 16; ++j) s_tile[wave * 16 + j][lane] = v[j][c];
            __syncthreads();
            for (int j = 0; j < 16; ++j) {
                const int kk = wave * 16 + j;
                if (k0 + kk < a.Kpad) cand[(tile * a.Kpad + k0 + kk) * 64 + lane] = s_tile[lane][kk];
            }
            __syncthreads();
        }
        return;
    }
    for (int j = 0; j < 16; ++j) {  // long rows: statistics first (projection coefficient, norm of the transformed row) ...
        const int row = wave * 16 + j;
        const int64_t e = e0 + row;
        float p = 0.f, n2 = 0.f;
        if (e < a.E) {
            const float* er = a.ent + e * a.K;
            float dt = 0.f;
            if (a.xform == X_TRANSH) {
                for (int k = lane; k < a.K; k += 64) dt = fmaf(er[k], vec[k] * iw, dt);
                p = -wave_sum(dt);
            } else {
                const float* em = a.ent_map + e * a.K;
                for (int k = lane; k < a.K; k += 64) dt = fmaf(er[k], em[k], dt);
                p = wave_sum(dt);
            }
            for (int k = lane; k < a.K; k += 64) { const float v = fmaf(p, vec[k] * iw, er[k]); n2 = fmaf(v, v, n2); }
            n2 = wave_sum(n2);
        }
        if (lane == 0) { s_p[row] = p; s_scale[row] = 1.0f / fmaxf(sqrtf(n2), kEpsNormalize); }
    }
    __syncthreads();
    for (int k0 = 0; k0 < a.Kpad; k0 += 64) {  // ... then transpose 64x64 through LDS: coalesced reads AND writes
        for (int j = 0; j < 16; ++j) {
            const int row = wave * 16 + j;
            const int64_t e = e0 + row;
            const int k = k0 + lane;
            float v = 0.f;
            if (e < a.E && k < a.K) v = fmaf(s_p[row], vec[k] * iw, a.ent[e * a.K + k]) * s_scale[row];
            s_tile[row][lane] = v;
        }
        __syncthreads();
        for (int j = 0; j < 16; ++j) {
            const int kk = wave * 16 + j;
            if (k0 + kk < a.Kpad) cand[(tile * a.Kpad + k0 + kk) * 64 + lane] = s_tile[lane][kk];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ 2. query vectors
// one wave (translation family) or one workgroup per test triple; writes qvec[(2i+side)*QV*Kpad ...], side 0 = tail sweep (h,r,?), 1 = head sweep (?,r,t).
// `only` = 0 / 1 (one-sided score sweeps, kge_eval_sweep_scores_side): the wanted side's vector of triple i is row i, the other
// side's goes to row n + i, which nothing reads (the sweep then walks n query rows).  `only` = 2: both, interleaved.
template <int M>
__global__ __launch_bounds__(256) void k_eval_queries(DeviceModel m, const int64_t* __restrict__ triples, int64_t n,
                                                      int K, int Kpad, int QV, float* __restrict__ qvec,
                                                      float* __restrict__ qscale, int only) {
    // the translation family needs wave-wide row reductions: one wave per triple; everything else is element-wise over k
    // (or independent per output column, RESCAL): the whole workgroup takes one triple, four times the loads in flight per row
    constexpr int TPT = (M == KGE_TRANSE || M == KGE_TRANSH || M == KGE_TRANSD || M == KGE_TRANSM) ? 64 : 256;
    const int lane = threadIdx.x % TPT;
    const int64_t i = (int64_t)blockIdx.x * (256 / TPT) + threadIdx.x / TPT;
    if (i >= n) return;
    const int64_t h = triples[3 * i], r = triples[3 * i + 1], t = triples[3 * i + 2];
    const int64_t row_t = only == 2 ? 2 * i : only == 0 ? i : n + i;
    const int64_t row_h = only == 2 ? 2 * i + 1 : only == 1 ? i : n + i;
    if (lane == 0) {
        const float sc = (M == KGE_TRANSM) ? m.tab[2][r] : 1.0f;
        qscale[row_t] = sc; qscale[row_h] = sc;
    }
    float* qt = qvec + row_t * (int64_t)QV * Kpad;
    float* qh = qvec + row_h * (int64_t)QV * Kpad;
    const int d = m.dim;
    for (int k = K + lane; k < Kpad; k += TPT) {
        qt[k] = 0.f; qh[k] = 0.f;
        if (QV == 2) { qt[Kpad + k] = 0.f; qh[Kpad + k] = 0.f; }
    }
    if constexpr (M == KGE_TRANSE || M == KGE_TRANSH || M == KGE_TRANSD || M == KGE_TRANSM) {
        const float* eh = m.tab[0] + h * d; const float* er = m.tab[1] + r * d; const float* et = m.tab[0] + t * d;
        // projected head / tail  (a, c) and the relation-side vector the sweep needs
        float ph = 0.f, pt = 0.f, iw = 1.f;
        const float* w = nullptr; const float* rm = nullptr;
        if constexpr (M == KGE_TRANSH) {
            w = m.tab[2] + r * d;
            float nw = 0.f;
            for (int k = lane; k < d; k += 64) nw = fmaf(w[k], w[k], nw);
            iw = 1.0f / fmaxf(sqrtf(wave_sum(nw)), kEpsNormalize);
            for (int k = lane; k < d; k += 64) { ph = fmaf(eh[k], w[k] * iw, ph); pt = fmaf(et[k], w[k] * iw, pt); }
            ph = wave_sum(ph); pt = wave_sum(pt);
        } else if constexpr (M == KGE_TRANSD) {
            const float* hm = m.tab[2] + h * d; const float* tm = m.tab[2] + t * d;
            rm = m.tab[3] + r * d;
            for (int k = lane; k < d; k += 64) { ph = fmaf(eh[k], hm[k], ph); pt = fmaf(et[k], tm[k], pt); }
            ph = wave_sum(ph); pt = wave_sum(pt);
        }
        auto proj_h = [&](int k) -> float {
            if constexpr (M == KGE_TRANSH) return eh[k] - ph * (w[k] * iw);
            else if constexpr (M == KGE_TRANSD) return eh[k] + ph * rm[k];
            else return eh[k];
        };
        auto proj_t = [&](int k) -> float {
            if constexpr (M == KGE_TRANSH) return et[k] - pt * (w[k] * iw);
            else if constexpr (M == KGE_TRANSD) return et[k] + pt * rm[k];
            else return et[k];
        };
        float na = 0.f, nb = 0.f, nc = 0.f;
        for (int k = lane; k < d; k += 64) {
            const float a = proj_h(k), b = er[k], c = proj_t(k);
            na = fmaf(a, a, na); nb = fmaf(b, b, nb); nc = fmaf(c, c, nc);
        }
        const float ia = 1.0f / fmaxf(sqrtf(wave_sum(na)), kEpsNormalize);
        const float ib = 1.0f / fmaxf(sqrtf(wave_sum(nb)), kEpsNormalize);
        const float ic = 1.0f / fmaxf(sqrtf(wave_sum(nc)), kEpsNormalize);
        for (int k = lane; k < d; k += 64) {
            qt[k] = proj_h(k) * ia + er[k] * ib;   // score = || q - c^ ||
            qh[k] = proj_t(k) * ic - er[k] * ib;   // score = || c^ + r^ - t^ || = || c^ - q ||
            if (QV == 2) {  // the in-sweep candidate transform needs the relation-side vector next to the query
                if constexpr (M == KGE_TRANSH) { qt[Kpad + k] = w[k] * iw; qh[Kpad + k] = w[k] * iw; }
                if constexpr (M == KGE_TRANSD) { qt[Kpad + k] = rm[k]; qh[Kpad + k] = rm[k]; }
            }
        }
    } else if constexpr (M == KGE_HEAD_1N_INTERNAL) {
        // the 1-N head: the query IS the caller's activation row (one-sided: row i of x), extended by the 1 that multiplies the bias
        const float* x = m.tab[2] + i * d;
        for (int k = lane; k < d; k += TPT) { qt[k] = x[k]; qh[k] = 0.f; }
        if (lane == 0 && m.tab[1] != nullptr) { qt[d] = 1.0f; qh[d] = 0.f; }
    } else if constexpr (M == KGE_DISTMULT) {
        const float* eh = m.tab[0] + h * d; const float* er = m.tab[1] + r * d; const float* et = m.tab[0] + t * d;
        for (int k = lane; k < d; k += TPT) { qt[k] = eh[k] * er[k]; qh[k] = er[k] * et[k]; }
    } else if constexpr (M == KGE_COMPLEX || M == KGE_ANALOGY) {
        constexpr int o = (M == KGE_ANALOGY) ? 2 : 0;
        const int dc = (M == KGE_ANALOGY) ? d / 2 : d;
        const int base = (M == KGE_ANALOGY) ? d : 0;
        const float* hr = m.tab[o + 0] + h * dc; const float* hi = m.tab[o + 1] + h * dc;
        const float* rr = m.tab[o + 2] + r * dc; const float* ri = m.tab[o + 3] + r * dc;
        const float* tr = m.tab[o + 0] + t * dc; const float* ti = m.tab[o + 1] + t * dc;
        for (int k = lane; k < dc; k += TPT) {
            qt[base + k] = hr[k] * rr[k] - hi[k] * ri[k];
            qt[base + dc + k] = hi[k] * rr[k] + hr[k] * ri[k];
            qh[base + k] = tr[k] * rr[k] + ti[k] * ri[k];
            qh[base + dc + k] = ti[k] * rr[k] - tr[k] * ri[k];
        }
        if constexpr (M == KGE_ANALOGY) {
            const float* eh = m.tab[0] + h * d; const float* er = m.tab[1] + r * d; const float* et = m.tab[0] + t * d;
            for (int k = lane; k < d; k += TPT) { qt[k] = eh[k] * er[k]; qh[k] = er[k] * et[k]; }
        }
    } else if constexpr (M == KGE_ROTATE) {
        const float* hr = m.tab[0] + h * d; const float* hi = m.tab[1] + h * d; const float* rl = m.tab[2] + r * d;
        const float* tr = m.tab[0] + t * d; const float* ti = m.tab[1] + t * d;
        for (int k = lane; k < d; k += TPT) {
            float sn, cs;
            sincosf(rl[k] / m.phase_div, &sn, &cs);
            qt[k] = hr[k] * cs - hi[k] * sn;       // h o r ;  score = |h o r - t|^2 - margin
            qt[d + k] = hr[k] * sn + hi[k] * cs;
            qh[k] = tr[k] * cs + ti[k] * sn;       // t o conj(r) ; |h o r - t| = |h - t o conj(r)| for |r| = 1
            qh[d + k] = ti[k] * cs - tr[k] * sn;
        }
    } else if constexpr (M == KGE_CP) {  // candidate row [sub[e] | obj[e]]
        const float* sh = m.tab[0] + h * d; const float* er = m.tab[1] + r * d; const float* ot = m.tab[2] + t * d;
        for (int k = lane; k < d; k += TPT) {
            qt[k] = 0.f; qt[d + k] = sh[k] * er[k];     // tail sweep: <sub[h] o rel[r], obj[e]>
            qh[k] = er[k] * ot[k]; qh[d + k] = 0.f;     // head sweep: <sub[e], rel[r] o obj[t]>
        }
    } else if constexpr (M == KGE_SIMPLE || M == KGE_SIMPLE_IGNR) {  // candidate row [tail[e] | head[e]]
        const float w2 = (M == KGE_SIMPLE) ? 0.5f : 1.0f;  // pointwise.py:525: only the inverse term is halved
        const float* hh = m.tab[0] + h * d; const float* ht = m.tab[0] + t * d;   // head-role rows of h and t
        const float* th = m.tab[1] + h * d; const float* tt = m.tab[1] + t * d;   // tail-role rows of h and t
        const float* r1 = m.tab[2] + r * d; const float* r2 = m.tab[3] + r * d;
        for (int k = lane; k < d; k += TPT) {
            qt[k] = hh[k] * r1[k];              // <head[h], rel, tail[e]>
            qt[d + k] = w2 * (r2[k] * th[k]);   // <head[e], rel_inv, tail[h]>
            qh[k] = w2 * (ht[k] * r2[k]);       // <head[t], rel_inv, tail[e]>
            qh[d + k] = r1[k] * tt[k];          // <head[e], rel, tail[t]>
        }
    } else if constexpr (M == KGE_QUATE) {  // candidate row [s | x | y | z][e]
        const float* H[4]; const float* T[4]; const float* Rq[4];
        for (int c = 0; c < 4; ++c) { H[c] = m.tab[c] + h * d; T[c] = m.tab[c] + t * d; Rq[c] = m.tab[4 + c] + r * d; }
        for (int k = lane; k < d; k += TPT) {
            const float rs = Rq[0][k], rx = Rq[1][k], ry = Rq[2][k], rz = Rq[3][k];
            const float den = sqrtf(rs * rs + rx * rx + ry * ry + rz * rz);
            const float inv = den > 0.f ? 1.0f / den : 0.f;
            const float ps = rs * inv, px = rx * inv, py = ry * inv, pz = rz * inv;
            const float hs = H[0][k], hx = H[1][k], hy = H[2][k], hz = H[3][k];
            const float ts = T[0][k], tx = T[1][k], ty = T[2][k], tz = T[3][k];
            qt[k] = hs * ps - hx * px - hy * py - hz * pz;          // h (x) r^, dotted with the candidate tail
            qt[d + k] = hs * px + ps * hx + hy * pz - py * hz;
            qt[2 * d + k] = hs * py + ps * hy + hz * px - pz * hx;
            qt[3 * d + k] = hs * pz + ps * hz + hx * py - px * hy;
            qh[k] = ps * ts + px * tx + py * ty + pz * tz;          // coefficients of the candidate head's s, x, y, z
            qh[d + k] = -px * ts + ps * tx - pz * ty + py * tz;
            qh[2 * d + k] = -py * ts + pz * tx + ps * ty - px * tz;
            qh[3 * d + k] = -pz * ts - py * tx + px * ty + ps * tz;
        }
    } else if constexpr (M == KGE_RESCAL) {
        const float* eh = m.tab[0] + h * d; const float* et = m.tab[0] + t * d;
        const float* Mr = m.tab[1] + r * (int64_t)d * d;
        for (int j = lane; j < d; j += TPT) {
            float a = 0.f, b = 0.f;
            int i2 = 0;
            for (; i2 + KC <= d; i2 += KC) {  // operands of KC steps in flight before the dependent fma chains
                float m1[KC], m2[KC], hv[KC], tv[KC];
#pragma unroll
                for (int u = 0; u < KC; ++u) {
                    m1[u] = Mr[(int64_t)(i2 + u) * d + j]; m2[u] = Mr[(int64_t)j * d + i2 + u];
                    hv[u] = eh[i2 + u]; tv[u] = et[i2 + u];
                }
#pragma unroll
                for (int u = 0; u < KC; ++u) { a = fmaf(hv[u], m1[u], a); b = fmaf(m2[u], tv[u], b); }
            }
            for (; i2 < d; ++i2) {
                a = fmaf(eh[i2], Mr[(int64_t)i2 * d + j], a);   // (h^T M)_j
                b = fmaf(Mr[(int64_t)j * d + i2], et[i2], b);   // (M t)_j
            }
            qt[j] = a; qh[j] = b;
        }
    }
}

// ------------------------------------------------------------------ pair arithmetic shared by steps 3 and 4
// L1 step: d = c - q, acc + |d| -- two plain (non-packed) VALU ops per element, abs folded in as a source modifier.
// tools/valu_bench.hip (profiles/r02_valu_bench.txt): a plain VOP2/VOP3 f32 op issues every ~2.4 cycles per SIMD,
// v_pk_add_f32 every ~4.3, and [v_sub, v_sub, v_add |x|, v_add |x|] per two elements sustains 32.9 T elements/s on the
// chip against 24.7 T for [v_pk_add, v_add |x|, v_add |x|]: fewer instructions is not faster here.  This translation
// unit is compiled with -fno-slp-vectorize (Makefile) so that the compiler neither re-packs the subtractions into
// v_pk_add_f32 nor turns pairs of |x| adds into v_and_b32 x2 + v_pk_add_f32, and is free to interleave the queries'
// independent chains (inline asm would pin the order and cost an s_nop per dependent pair).
__device__ __forceinline__ float add_abs(float acc, float d) { return acc + fabsf(d); }

// acc + d*d as ONE v_fma_f32, hidden from the SLP vectoriser for the same reason (it otherwise pairs the two tiles'
// accumulators into v_pk_fma_f32 and pays ~1.2 v_mov per element to shuffle operands); same IEEE fma either way.
__device__ __forceinline__ float fma_sq(float acc, float d) {
    float r;
    asm("v_fma_f32 %0, %1, %1, %2" : "=v"(r) : "v"(d), "v"(acc));
    return r;
}

template <int FORM>
__device__ __forceinline__ float pair_step(float acc, float c, float q) {
    if constexpr (FORM == F_L1) return add_abs(acc, c - q);
    else if constexpr (FORM == F_NEGDOT) return fmaf(c, q, acc);
    else { const float dlt = c - q; return fma_sq(acc, dlt); }
}
template <int FORM>
__device__ __forceinline__ float pair_finish(float acc, float margin) {
    if constexpr (FORM == F_L1) return acc;
    else if constexpr (FORM == F_L2) return sqrtf(acc);
    else if constexpr (FORM == F_SQM) return -(margin - acc);
    else return -acc;
}

template <int POST>
__device__ __forceinline__ float pair_post(float s, float scale) {
    if constexpr (POST == P_SCALE) return scale * s;
    else if constexpr (POST == P_CLAMP) return fminf(fmaxf(s, -20.f), 20.f);
    else if constexpr (POST == P_SIGMOID) return -(1.0f / (1.0f + expf(s)));   // s = -logit: the head's own sigmoid expression (kge_head.hip)
    else return s;
}

// squared L2 distance from the dot product and the two squared norms (matrix-core sweep and its target / filter twin)
__device__ __forceinline__ float sqm_from_dot(float dot, float qn, float cn) { return fmaf(-2.0f, dot, qn + cn); }

// full sequential score of one (query, candidate) pair by ONE lane (target / filter path of the VALU sweep)
template <int FORM, int XFORM, int POST>
__device__ __forceinline__ float pair_score_lane(const float* __restrict__ cand, const float* __restrict__ aux,
                                                 const float* __restrict__ q, int64_t e, int Kpad, float margin,
                                                 float scale) {
    const float* c = cand + ((e >> 6) * Kpad) * 64 + (e & 63);
    float acc = 0.f;
    // candidate / query elements are fetched KC at a time BEFORE the dependent accumulation chain (one memory round trip
    // per KC elements instead of one per element); the accumulation order is unchanged
    if constexpr (XFORM == X_NONE && FORM != F_L1) {
        // the sweep accumulates even and odd k in the two halves of one packed register (v_pk_fma_f32) and adds the
        // halves at the end; the same order here keeps scores bit-identical between the two kernels
        float a0 = 0.f, a1 = 0.f;
        for (int k0 = 0; k0 < Kpad; k0 += KC) {
            float cv[KC], qv[KC];
#pragma unroll
            for (int j = 0; j < KC; ++j) { cv[j] = c[(int64_t)(k0 + j) * 64]; qv[j] = q[k0 + j]; }
#pragma unroll
            for (int j = 0; j < KC; j += 2) {
                if constexpr (FORM == F_NEGDOT) {
                    a0 = fmaf(cv[j], qv[j], a0);
                    a1 = fmaf(cv[j + 1], qv[j + 1], a1);
                } else {
                    const float d0 = cv[j] - qv[j], d1 = cv[j + 1] - qv[j + 1];
                    a0 = fmaf(d0, d0, a0);
                    a1 = fmaf(d1, d1, a1);
                }
            }
        }
        acc = a0 + a1;
    } else if constexpr (XFORM == X_NONE) {
        // (32 operands per dependent round trip, then the tail 8 at a time: Kpad is a multiple of 8; order unchanged)
        constexpr int KL = 32;
        int k0 = 0;
        for (; k0 + KL <= Kpad; k0 += KL) {
            float cv[KL], qv[KL];
#pragma unroll
            for (int j = 0; j < KL; ++j) { cv[j] = c[(int64_t)(k0 + j) * 64]; qv[j] = q[k0 + j]; }
#pragma unroll
            for (int j = 0; j < KL; ++j) acc = pair_step<FORM>(acc, cv[j], qv[j]);
        }
        for (; k0 < Kpad; k0 += KC) {
            float cv[KC], qv[KC];
#pragma unroll
            for (int j = 0; j < KC; ++j) { cv[j] = c[(int64_t)(k0 + j) * 64]; qv[j] = q[k0 + j]; }
#pragma unroll
            for (int j = 0; j < KC; ++j) acc = pair_step<FORM>(acc, cv[j], qv[j]);
        }
    } else {
        const float* w = q + Kpad;
        float p;
        if constexpr (XFORM == X_TRANSH) {
            p = 0.f;
            for (int k0 = 0; k0 < Kpad; k0 += KC) {
                float cv[KC], wv[KC];
#pragma unroll
                for (int j = 0; j < KC; ++j) { cv[j] = c[(int64_t)(k0 + j) * 64]; wv[j] = w[k0 + j]; }
#pragma unroll
                for (int j = 0; j < KC; ++j) p = fmaf(cv[j], wv[j], p);
            }
            p = -p;
        } else {
            p = aux[e];
        }
        float n2 = 0.f;
        for (int k0 = 0; k0 < Kpad; k0 += KC) {
            float cv[KC], wv[KC];
#pragma unroll
            for (int j = 0; j < KC; ++j) { cv[j] = c[(int64_t)(k0 + j) * 64]; wv[j] = w[k0 + j]; }
#pragma unroll
            for (int j = 0; j < KC; ++j) { const float v = fmaf(p, wv[j], cv[j]); n2 = fmaf(v, v, n2); }
        }
        const float inv = 1.0f / fmaxf(sqrtf(n2), kEpsNormalize);
        for (int k0 = 0; k0 < Kpad; k0 += KC) {
            float cv[KC], wv[KC], qv[KC];
#pragma unroll
            for (int j = 0; j < KC; ++j) { cv[j] = c[(int64_t)(k0 + j) * 64]; wv[j] = w[k0 + j]; qv[j] = q[k0 + j]; }
#pragma unroll
            for (int j = 0; j < KC; ++j) {
                const float v = fmaf(p, wv[j], cv[j]) * inv;
                acc = pair_step<FORM>(acc, v, qv[j]);
            }
        }
    }
    return pair_post<POST>(pair_finish<FORM>(acc, margin), scale);
}

// ------------------------------------------------------------------ 3. target score + filtered count
template <int FORM, int XFORM, int POST>
__global__ __launch_bounds__(256) void k_eval_target_filter(const float* __restrict__ cand, const float* __restrict__ aux,
                                                            const float* __restrict__ qvec, const float* __restrict__ qscale,
                                                            const int64_t* __restrict__ triples,
                                                            int64_t n, int Kpad, int QV, float margin,
                                                            const int64_t* __restrict__ tail_off, const int32_t* __restrict__ tail_ids,
                                                            const int64_t* __restrict__ head_off, const int32_t* __restrict__ head_ids,
                                                            float* __restrict__ st, int32_t* __restrict__ fcount,
                                                            const int32_t* __restrict__ group_of_triple, int64_t table_stride, int only) {
    const int lane = threadIdx.x & 63;
    const int64_t qi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= (only == 2 ? 2 * n : n)) return;
    const int64_t i = only == 2 ? qi >> 1 : qi;        // one-sided sweeps (only = 0 / 1): query row i IS triple i's tail / head sweep
    if (group_of_triple) cand += group_of_triple[i] * table_stride;  // grouped evaluation: this query's candidate table
    const int side = only == 2 ? (int)(qi & 1) : only;
    const int64_t truth = side == 0 ? triples[3 * i + 2] : triples[3 * i];
    const float* q = qvec + qi * (int64_t)QV * Kpad;
    const float scale = POST == P_SCALE ? qscale[qi] : 1.0f;
    float s_true = 0.f;
    if (lane == 0) s_true = pair_score_lane<FORM, XFORM, POST>(cand, aux, q, truth, Kpad, margin, scale);
    s_true = __shfl(s_true, 0, 64);
    const int64_t* off = side == 0 ? tail_off : head_off;
    const int32_t* ids = side == 0 ? tail_ids : head_ids;
    int cnt = 0;
    if (off != nullptr) {
        const int64_t b = off[i], e_ = off[i + 1];
        for (int64_t j = b + lane; j < e_; j += 64) {
            const int64_t e = ids[j];
            if (e != truth) {
                const float s = pair_score_lane<FORM, XFORM, POST>(cand, aux, q, e, Kpad, margin, scale);
                cnt += (s < s_true) ? 1 : 0;
            }
        }
    }
    cnt = (int)wave_sum((float)cnt);  // < 2^24 known entities per query
    if (lane == 0) { st[qi] = s_true; fcount[qi] = cnt; }
}

// ---- the same step for the matrix-core sweep (k_eval_gemm): every energy is ONE fmaf chain over k, the order the f32 MFMA
// accumulates in, so ranks stay exact functions of bit-identical energies.  A chain is sequential, its operands need not be:
// one wave per query takes the true candidate and the query's known entities as a list of (query, candidate) pairs, fetches a
// K chunk of up to PG candidate rows -- from the row-major tables the sweep layout was copied from (PrepArgs segments), 256
// coalesced bytes per row and instruction, all in flight together -- into LDS, and PG lanes run their chains out of LDS.
// PG x chunk = 1 024 elements: few pairs (the common case) take long chunks and few round trips, long lists 64 chains at once.
constexpr int kChainElems = 1024;
template <int FORM, int POST>
__global__ __launch_bounds__(256, 4) void k_eval_target_filter_chain(PrepArgs a, const float* __restrict__ aux,
                                                                  const float* __restrict__ qvec, const float* __restrict__ qnorm,
                                                                  const int64_t* __restrict__ triples, int64_t n, int Kpad, float margin,
                                                                  const int64_t* __restrict__ tail_off, const int32_t* __restrict__ tail_ids,
                                                                  const int64_t* __restrict__ head_off, const int32_t* __restrict__ head_ids,
                                                                  float* __restrict__ st, int32_t* __restrict__ fcount, int only) {
    static_assert(FORM == F_NEGDOT || FORM == F_SQM, "chain order: plain dot-product based forms");
    __shared__ __attribute__((aligned(16))) float s_c[4][kChainElems + 4 * 64];   // [pair][chunk + 4]: rows 4 banks apart
    __shared__ __attribute__((aligned(16))) float s_q[4][kChainElems / 4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t qi = (int64_t)blockIdx.x * 4 + wv;
    if (qi >= (only == 2 ? 2 * n : n)) return;
    const int64_t i = only == 2 ? qi >> 1 : qi;
    const int side = only == 2 ? (int)(qi & 1) : only;
    const int truth = (int)(side == 0 ? triples[3 * i + 2] : triples[3 * i]);
    const float* q = qvec + qi * (int64_t)Kpad;
    const float qn = FORM == F_SQM ? qnorm[qi] : 0.f;   // |q|^2
    const int64_t* off = side == 0 ? tail_off : head_off;
    const int32_t* ids = side == 0 ? tail_ids : head_ids;
    const int64_t b = off ? off[i] : 0;
    const int np = 1 + (off ? (int)(off[i + 1] - b) : 0);   // pair 0: the true candidate
    float* sc = s_c[wv];
    float* sq = s_q[wv];
    float s_true = 0.f;
    int cnt = 0;
    for (int base = 0; base < np;) {
        const int rem = np - base;
        const int lg = rem > 16 ? 6 : rem > 4 ? 4 : 2;   // log2 of the pairs of this group
        const int pg = min(1 << lg, rem);
        const int lc = 10 - lg, ch = 1 << lc;             // chunk length: 256 / 64 / 16
        const int pitch = ch + 4;
        const int idx = base + lane;
        int e = truth;
        if (lane < pg && idx > 0) e = ids[b + idx - 1];
        float acc = 0.f, tot = 0.f;   // tot: the finished chunks of kGemmChunk elements (see KGE_GEMM_CHUNK)
        const int total = pg << lc;
        // a chunk's operands travel global -> registers -> LDS; the NEXT chunk's loads are issued before this chunk's chains
        // run, so their round trip hides behind the (sequential) arithmetic
        float v[16], qv[4];
        const int total_u = __builtin_amdgcn_readfirstlane(total), lc_u = __builtin_amdgcn_readfirstlane(lc);
        auto fetch = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                v[u] = 0.f;
                if (64 * u < total_u) {   // (wave-uniform: a short list skips the address arithmetic of the unused slots)
                    const int t = 64 * u + lane;
                    const int kk = k0 + (t & (ch - 1));
                    // chunks of >= 64 elements: the 64 lanes of one load read ONE pair's row -- scalar row base
                    int ee = lc_u >= 6 ? __builtin_amdgcn_readlane(e, (64 * u) >> lc_u) : __shfl(e, (t >> lc_u) & 63, 64);
                    asm volatile("" : "+v"(ee));   // (keeps 16 row addresses from being hoisted out of the chunk loop: 32 live registers)
                    const float* src = prep_addr(a, ee, kk);
                    if (t < total && src) v[u] = *src;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kk = 64 * u + lane;
                qv[u] = (kk < ch && k0 + kk < Kpad) ? q[k0 + kk] : 0.f;
            }
        };
        fetch(0);
        for (int k0 = 0; k0 < Kpad; k0 += ch) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int t = 64 * u + lane;
                if (64 * u < total_u && t < total) sc[(t >> lc) * pitch + (t & (ch - 1))] = v[u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kk = 64 * u + lane;
                if (kk < ch) sq[kk] = qv[u];
            }
            if (k0 + ch < Kpad) fetch(k0 + ch);
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0) only: this wave's LDS writes have landed; the prefetch stays in flight
            if (lane < pg) {
                const float* cr = sc + lane * pitch;
                const int nk = min(ch, Kpad - k0);   // Kpad is a multiple of 8
                // (16 operands per LDS round trip: the reads of a block are issued together, then its 16 dependent fmas)
                int kk = 0;
                for (; kk + 16 <= nk; kk += 16) {
                    float4 c4[4], q4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        c4[u] = *reinterpret_cast<const float4*>(cr + kk + 4 * u);
                        q4[u] = *reinterpret_cast<const float4*>(sq + kk + 4 * u);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        acc = fmaf(c4[u].x, q4[u].x, acc); acc = fmaf(c4[u].y, q4[u].y, acc);
                        acc = fmaf(c4[u].z, q4[u].z, acc); acc = fmaf(c4[u].w, q4[u].w, acc);
                    }
                    if (kGemmChunk && ((k0 + kk + 16) % (kGemmChunk ? kGemmChunk : 1)) == 0) { tot += acc; acc = 0.f; }
                }
                for (; kk < nk; kk += 8) {
                    const float4 c0 = *reinterpret_cast<const float4*>(cr + kk), c1 = *reinterpret_cast<const float4*>(cr + kk + 4);
                    const float4 q0 = *reinterpret_cast<const float4*>(sq + kk), q1 = *reinterpret_cast<const float4*>(sq + kk + 4);
                    acc = fmaf(c0.x, q0.x, acc); acc = fmaf(c0.y, q0.y, acc); acc = fmaf(c0.z, q0.z, acc); acc = fmaf(c0.w, q0.w, acc);
                    acc = fmaf(c1.x, q1.x, acc); acc = fmaf(c1.y, q1.y, acc); acc = fmaf(c1.z, q1.z, acc); acc = fmaf(c1.w, q1.w, acc);
                    if (kGemmChunk && ((k0 + kk + 8) % (kGemmChunk ? kGemmChunk : 1)) == 0) { tot += acc; acc = 0.f; }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        // the last chunk (k_eval_gemm folds after its last K slab whatever the slab count; elements past Kpad are zeros there: the
        // chain passes them unchanged).  A chunk that ended exactly at Kpad left acc = +0: tot + 0 = tot.
        if (kGemmChunk) acc = tot + acc;
        // squared distance through the expansion |q|^2 + |c|^2 - 2 <q, c> the matrix-core sweep uses (aux[e] = |c|^2): same
        // stored norms, same operation order => same bits as k_eval_gemm
        if constexpr (FORM == F_SQM) acc = sqm_from_dot(acc, qn, lane < pg ? aux[e] : 0.f);
        const float sco = pair_post<POST>(pair_finish<FORM>(acc, margin), 1.0f);
        if (base == 0) s_true = __shfl(sco, 0, 64);
        if (lane < pg && idx > 0 && e != truth) cnt += (sco < s_true) ? 1 : 0;
        base += pg;
    }
    cnt = (int)wave_sum((float)cnt);  // < 2^24 known entities per query
    if (lane == 0) { st[qi] = s_true; fcount[qi] = cnt; }
}

// ------------------------------------------------------------------ 4. the sweep
// 1-D grid decoded as (CU slot, tile split, query block): block b runs on XCD b%8 / CU slot b%256 (observed
// round-robin placement, used for locality only), so  slot = b % 256,  m = b / 256,  ts = m % S,  qb = (m / S)*256 + slot
// keeps one CU on ONE block of QT queries for S consecutive workgroups: its query rows stay in the scalar cache while
// all CUs walk the candidate tiles in step (L2-friendly).  Each wave owns tiles ts*4+wave, +4S, +8S, ...
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int FORM>
__device__ __forceinline__ void pair_step2(float& acc, f32x2 c, f32x2 q) {
    // two consecutive k for one (query, candidate) pair: the subtraction is one packed op, the accumulation
    // stays sequential in k (bit-identical to pair_step applied twice)
    if constexpr (FORM == F_NEGDOT) {
        acc = fmaf(c.x, q.x, acc);
        acc = fmaf(c.y, q.y, acc);
    } else if constexpr (FORM == F_L1) {
        acc = add_abs(acc, c.x - q.x);
        acc = add_abs(acc, c.y - q.y);
    } else {
        const f32x2 d = c - q;
        acc = fma_sq(acc, d.x); acc = fma_sq(acc, d.y);
    }
}

// One KC-chunk of every query of the block, in query order: body(q, v) sees v[s][j] = qrow[q][off[s] + j] (scalar loads: the rows are
// wave-uniform).  The chunk of query q + 1 is requested BEFORE the arithmetic of query q.  Scalar loads return out of order, so the only
// wait there is waits for all of them: requested right in front of its use -- what the compiler does by itself -- that wait stands
// between every two queries with nothing to hide it (36 VALU instructions per query and chunk in the L1 sweep).  The empty asm is a
// use of the current chunk in front of the next request: it pins the wait THERE (profiles/r05_experiments.md section 11).
template <int QT, int NS, class F>
__device__ __forceinline__ void for_query_chunks(const float* const (&qrow)[QT], const int (&off)[NS], F body) {
    float nx[NS][KC];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int j = 0; j < KC; ++j) nx[s][j] = qrow[0][off[s] + j];
#pragma unroll
    for (int q = 0; q < QT; ++q) {
        float cur[NS][KC];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int j = 0; j < KC; ++j) cur[s][j] = nx[s][j];
#pragma unroll
        for (int s = 0; s < NS; ++s) asm volatile("" ::"s"(cur[s][0]), "s"(cur[s][KC - 1]));
        if (q + 1 < QT) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int j = 0; j < KC; ++j) nx[s][j] = qrow[q + 1][off[s] + j];
        }
        __builtin_amdgcn_sched_barrier(0);
        body(q, cur);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int FORM, int XFORM, int QT, bool WRITE, int POST>
__global__ __launch_bounds__(256) void k_eval_sweep(const float* __restrict__ cand, const float* __restrict__ aux,
                                                    const float* __restrict__ qvec, const float* __restrict__ qscale,
                                                    const float* __restrict__ st,
                                                    int64_t nq, int64_t E, int64_t ntiles, int Kpad, int QV, float margin,
                                                    int S, int qblocks, int32_t* __restrict__ rcount,
                                                    int32_t* __restrict__ tcount, float* __restrict__ scores_out,
                                                    const int32_t* __restrict__ qdesc, int64_t table_stride) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int slot = blockIdx.x & 255;
    const int mm = blockIdx.x >> 8;
    const int ts = mm % S;
    const int qb = (mm / S) * 256 + slot;
    if (qb >= qblocks) return;
    int64_t q0 = (int64_t)qb * QT;
    if (qdesc) {  // grouped evaluation: block -> {candidate table, first query, query count <= QT}
        cand += qdesc[4 * qb] * table_stride;
        q0 = qdesc[4 * qb + 1];
        nq = q0 + qdesc[4 * qb + 2];
    }
    const int64_t qstride = (int64_t)QV * Kpad;
    // wave-uniform query row pointers (clamped; masked at the end)
    const float* qrow[QT];
    float sthr[QT];
    float qsc[QT];
    int cnt[QT];
    int tie[QT];     // candidates whose energy equals the target's bit for bit (the target itself is one of them): wave-uniform, like cnt
#pragma unroll
    for (int q = 0; q < QT; ++q) {
        const int64_t qi = (q0 + q < nq) ? q0 + q : nq - 1;
        qrow[q] = qvec + qi * qstride;
        sthr[q] = WRITE ? 0.f : st[qi];
        qsc[q] = POST == P_SCALE ? qscale[qi] : 1.0f;
        cnt[q] = 0;
        tie[q] = 0;
    }
    if constexpr (XFORM == X_NONE) {
        // plain forms: TWO candidate tiles per wave pass -- every scalar query operand feeds two VALU streams
        for (int64_t tile = ((int64_t)ts * 4 + wave) * 2; tile < ntiles; tile += (int64_t)S * 8) {
            const bool has_b = tile + 1 < ntiles;
            const float* ca = cand + (tile * Kpad) * 64 + lane;
            const float* cb = has_b ? ca + (int64_t)Kpad * 64 : ca;
            float acca[QT], accb[QT];
            f32x2 pa[QT], pb[QT];  // NEGDOT: packed even/odd-k accumulators
#pragma unroll
            for (int q = 0; q < QT; ++q) { acca[q] = 0.f; accb[q] = 0.f; pa[q] = f32x2{0.f, 0.f}; pb[q] = f32x2{0.f, 0.f}; }
            for (int k0 = 0; k0 < Kpad; k0 += KC) {
                f32x2 va[KC / 2], vb[KC / 2];
#pragma unroll
                for (int j = 0; j < KC / 2; ++j) {
                    va[j].x = ca[(int64_t)(k0 + 2 * j) * 64]; va[j].y = ca[(int64_t)(k0 + 2 * j + 1) * 64];
                    vb[j].x = cb[(int64_t)(k0 + 2 * j) * 64]; vb[j].y = cb[(int64_t)(k0 + 2 * j + 1) * 64];
                }
                const int off1[1] = {k0};
                for_query_chunks<QT, 1>(qrow, off1, [&](int q, const float (&qv)[1][KC]) {
#pragma unroll
                    for (int j = 0; j < KC / 2; ++j) {
                        f32x2 qq;
                        qq.x = qv[0][2 * j];
                        qq.y = qv[0][2 * j + 1];
                        if constexpr (FORM == F_NEGDOT) {  // one v_pk_fma_f32 per k-pair: SGPR pair x VGPR pair
                            pa[q] = __builtin_elementwise_fma(va[j], qq, pa[q]);
                            pb[q] = __builtin_elementwise_fma(vb[j], qq, pb[q]);
                        } else if constexpr (FORM == F_L2 || FORM == F_SQM) {  // v_pk_add (sub) + v_pk_fma per k-pair
                            const f32x2 da = va[j] - qq, db = vb[j] - qq;
                            pa[q] = __builtin_elementwise_fma(da, da, pa[q]);
                            pb[q] = __builtin_elementwise_fma(db, db, pb[q]);
                        } else {
                            pair_step2<FORM>(acca[q], va[j], qq);
                            pair_step2<FORM>(accb[q], vb[j], qq);
                        }
                    }
                });
            }
            if constexpr (FORM != F_L1) {
#pragma unroll
                for (int q = 0; q < QT; ++q) { acca[q] = pa[q].x + pa[q].y; accb[q] = pb[q].x + pb[q].y; }
            }
            const int64_t ea = tile * 64 + lane, eb = ea + 64;
            const bool valid_a = ea < E, valid_b = has_b && eb < E;
#pragma unroll
            for (int q = 0; q < QT; ++q) {
                const float sa = pair_post<POST>(pair_finish<FORM>(acca[q], margin), qsc[q]);
                const float sb = pair_post<POST>(pair_finish<FORM>(accb[q], margin), qsc[q]);
                if constexpr (WRITE) {
                    if (q0 + q < nq) {
                        if (valid_a) scores_out[(q0 + q) * E + ea] = sa;
                        if (valid_b) scores_out[(q0 + q) * E + eb] = sb;
                    }
                } else {
                    cnt[q] += __popcll(__ballot(valid_a && sa < sthr[q])) + __popcll(__ballot(valid_b && sb < sthr[q]));
                    tie[q] += __popcll(__ballot(valid_a && sa == sthr[q])) + __popcll(__ballot(valid_b && sb == sthr[q]));
                }
            }
        }
    } else {
    for (int64_t tile = (int64_t)ts * 4 + wave; tile < ntiles; tile += (int64_t)S * 4) {
        const float* c = cand + (tile * Kpad) * 64 + lane;
        float acc[QT];
#pragma unroll
        for (int q = 0; q < QT; ++q) acc[q] = 0.f;
        {
            float p[QT], inv[QT];
            if constexpr (XFORM == X_TRANSH) {
#pragma unroll
                for (int q = 0; q < QT; ++q) p[q] = 0.f;
                for (int k0 = 0; k0 < Kpad; k0 += KC) {
                    float cv[KC];
#pragma unroll
                    for (int j = 0; j < KC; ++j) cv[j] = c[(int64_t)(k0 + j) * 64];
                    const int off1[1] = {Kpad + k0};
                    for_query_chunks<QT, 1>(qrow, off1, [&](int q, const float (&qv)[1][KC]) {
#pragma unroll
                        for (int j = 0; j < KC; ++j) p[q] = fmaf(cv[j], qv[0][j], p[q]);
                    });
                }
#pragma unroll
                for (int q = 0; q < QT; ++q) p[q] = -p[q];
            } else {
                const float a_e = aux[tile * 64 + lane];
#pragma unroll
                for (int q = 0; q < QT; ++q) p[q] = a_e;
            }
#pragma unroll
            for (int q = 0; q < QT; ++q) inv[q] = 0.f;
            for (int k0 = 0; k0 < Kpad; k0 += KC) {
                float cv[KC];
#pragma unroll
                for (int j = 0; j < KC; ++j) cv[j] = c[(int64_t)(k0 + j) * 64];
                const int off1[1] = {Kpad + k0};
                for_query_chunks<QT, 1>(qrow, off1, [&](int q, const float (&qv)[1][KC]) {
#pragma unroll
                    for (int j = 0; j < KC; ++j) {
                        const float v = fmaf(p[q], qv[0][j], cv[j]);
                        inv[q] = fmaf(v, v, inv[q]);
                    }
                });
            }
#pragma unroll
            for (int q = 0; q < QT; ++q) inv[q] = 1.0f / fmaxf(sqrtf(inv[q]), kEpsNormalize);
            for (int k0 = 0; k0 < Kpad; k0 += KC) {
                float cv[KC];
#pragma unroll
                for (int j = 0; j < KC; ++j) cv[j] = c[(int64_t)(k0 + j) * 64];
                const int off2[2] = {Kpad + k0, k0};
                for_query_chunks<QT, 2>(qrow, off2, [&](int q, const float (&qv)[2][KC]) {
#pragma unroll
                    for (int j = 0; j < KC; ++j) {
                        const float v = fmaf(p[q], qv[0][j], cv[j]) * inv[q];
                        acc[q] = pair_step<FORM>(acc[q], v, qv[1][j]);
                    }
                });
            }
        }
        const int64_t e = tile * 64 + lane;
        const bool valid = e < E;
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            const float s = pair_post<POST>(pair_finish<FORM>(acc[q], margin), qsc[q]);
            if constexpr (WRITE) {
                if (valid && q0 + q < nq) scores_out[(q0 + q) * E + e] = s;
            } else {
                cnt[q] += __popcll(__ballot(valid && s < sthr[q]));
                tie[q] += __popcll(__ballot(valid && s == sthr[q]));
            }
        }
    }
    }
    if constexpr (!WRITE) {
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < QT; ++q) {
                if (q0 + q < nq && cnt[q] != 0) atomicAdd(rcount + q0 + q, cnt[q]);
                if (tcount && q0 + q < nq && tie[q] != 0) atomicAdd(tcount + q0 + q, tie[q]);
            }
        }
    }
}

// ------------------------------------------------------------------ 4b. the dot-product sweep on the matrix cores
// DistMult / ComplEx / ANALOGY / RESCAL / CP / SimplE / QuatE rank by s(q, e) = -<q, c_e>: a [candidates x K] x [K x queries]
// GEMM.  The VALU sweep feeds every packed FMA one scalar operand pair and re-reads each candidate chunk once per 16
// queries; with K = 400 .. 2000 its operand streams (candidates AND queries) come from the Infinity Cache and it stalls at
// ~0.58 of its issue roof (profiles/r02_experiments.md).  Here a workgroup holds a 128-query x 128-candidate tile of
// accumulators on the f32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, 157 TF peak), both operands staged through
// LDS in 16-deep K slabs (one 16-byte load per operand float4, issued a slab ahead, double-buffered, one barrier per
// slab): 64 FMAs per operand float fetched instead of 16.  Candidates are the M side, so every lane owns ONE query column
// and the count epilogue needs one threshold and one counter per lane and column block.  The fp32 result of an MFMA chain
// is bit-identical to one fmaf chain over k, which is the order k_eval_target_filter_chain uses for s(q, true): integer
// ranks stay exact functions of the fp32 energies.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int GT = 128;      // tile edge (queries and candidates) per workgroup
#ifndef KGE_GEMM_KS
#define KGE_GEMM_KS 16
#endif
constexpr int GKS = KGE_GEMM_KS;   // K slab (16; 32 = experiment: half the barriers, twice the LDS -- free at two workgroups per CU)
constexpr int GNJ = GKS / 8;        // float4s of each operand slab a thread stages
static_assert(kGemmChunk % GKS == 0, "the running total is folded at K slab boundaries");
#if defined(KGE_GEMM_32X32) && KGE_GEMM_CHUNK
#error "the 32x32x2 experiment form of k_eval_gemm keeps the single chain: build it with -DKGE_GEMM_CHUNK=0"
#endif
// MFMA form of the sweep: v_mfma_f32_16x16x4_f32 (default) or v_mfma_f32_32x32x2_f32 (-DKGE_GEMM_32X32, the round-2 form).  Both run
// at the same peak; the 16x16x4 form has a 40-cycle dependent latency instead of 64 and holds its issue rate with four waves per
// SIMD (tools/mfma_bench.hip: 154 vs 129 TF), which is how this kernel runs (3-4 workgroups per CU).
#ifdef KGE_GEMM_32X32
constexpr int GLD = GT + 4;  // LDS row length (k-major tiles: [k][GT + pad])
#else
constexpr int GLD = GT + 16; // 16 lanes read 16 consecutive floats of row k, the next 16 lanes row k + 1: rows 16 banks apart
#endif

// queries [nq][Kpad] -> k-major tiles qT[tile][k][GT] (zero rows beyond nq): the layout the candidate table already has
__global__ __launch_bounds__(256) void k_eval_qt(const float* __restrict__ qvec, int64_t nq, int Kpad, float* __restrict__ qT) {
    __shared__ float s_t[64][65];
    const int64_t q0 = (int64_t)blockIdx.x * 64;   // 64 queries x 64 k per block
    const int k0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int64_t q = q0 + r;
        s_t[r][tx] = (q < nq && k0 + tx < Kpad) ? qvec[q * Kpad + k0 + tx] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int k = k0 + r;
        const int64_t q = q0 + tx;
        if (k < Kpad) qT[((q / GT) * Kpad + k) * GT + (q % GT)] = s_t[tx][r];
    }
}

// |q|^2 per query row (squared-distance form), one wave per query
__global__ __launch_bounds__(256) void k_eval_qnorm(const float* __restrict__ qvec, int64_t nq, int Kpad, float* __restrict__ qn) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    float n2 = 0.f;
    for (int k = lane; k < Kpad; k += 64) { const float v = qvec[q * Kpad + k]; n2 = fmaf(v, v, n2); }
    n2 = wave_sum(n2);
    if (lane == 0) qn[q] = n2;
}

// XCD-aware split of the (query tile, candidate tile) grid of k_eval_gemm: per XCD `a` whole query tiles x S candidate splits, plus
// the left-over query tiles in m parts per XCD; xcd = 0: the flat numbering of rounds 2-5 (query tile = block % qtiles)
struct GemmSplit { int xcd, a, S, m; };
static GemmSplit gemm_split(int qtiles, int64_t ctiles, int slots_per_xcd, unsigned* grid, int S_flat) {
    GemmSplit g{0, 0, 0, 0};
    const int sw = switch_value("EVAL_GEMM_XCD");
    if (sw == 0) { *grid = (unsigned)(qtiles * S_flat); return g; }
    const int a = qtiles / 8, r = qtiles % 8;
    const int smax = (int)(ctiles < slots_per_xcd ? ctiles : slots_per_xcd);
    for (int S = smax; S >= 1; --S) {
        int m = 0;
        if (r) {
            m = a ? (S + 7) / 8 : (int)((ctiles + 7) / 8 < slots_per_xcd / r ? (ctiles + 7) / 8 : slots_per_xcd / r);
            if (m < 1) m = 1;
        }
        if (a * S + r * m <= slots_per_xcd || S == 1) {
            g.xcd = 1; g.a = a; g.S = a ? S : 0; g.m = m;
            *grid = (unsigned)(8 * (g.a * g.S + r * m));
            return g;
        }
    }
    *grid = (unsigned)(qtiles * S_flat);
    return g;
}

// (four workgroups per CU for the dot form; the squared-distance epilogue needs ~20 more registers: three per CU, no spills)
// (with the running totals of KGE_GEMM_CHUNK: 64 more registers per lane, two workgroups per CU)
template <bool WRITE, int POST, bool SQM>
#ifndef KGE_GEMM_OCC
#define KGE_GEMM_OCC (kGemmChunk > 0 ? 2 : SQM ? 3 : 4)
#endif
__global__ __launch_bounds__(256, KGE_GEMM_OCC) void k_eval_gemm(const float* __restrict__ cand, const float* __restrict__ qT,
                                                   const float* __restrict__ st, int64_t nq, int64_t E, int64_t ntiles64,
                                                   int Kpad, int qtiles, int S, GemmSplit gs, int32_t* __restrict__ rcount,
                                                   int32_t* __restrict__ tcount, float* __restrict__ scores_out, const float* __restrict__ qn,
                                                   const float* __restrict__ cn, float margin) {
    __shared__ float sA[2][GKS][GLD], sB[2][GKS][GLD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;   // wave's 64 x 64 sub-tile: candidate rows wr, query columns wc
    const int64_t ctiles = (ntiles64 + 1) / 2;   // 128-candidate tiles
    // workgroup -> (query tile, candidate tiles ct0, ct0 + cstride, ...).  Round 6: XCD-aware (GemmSplit): consecutive block ids go round
    // the 8 XCDs, each with its own 4 MB L2, so XCD x takes the query tiles x, x + 8, ... -- a handful of tiles that STAY in its L2 --
    // and its workgroups of one candidate split sweep the same candidate tile at the same time: a candidate tile is fetched once per
    // XCD instead of once per (XCD, few query tiles).  The qtiles % 8 left-over query tiles are cut across all XCDs by candidate range.
    int qt, ct0, cstride;
    if (gs.xcd) {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        if (j < gs.a * gs.S) { qt = x + 8 * (j % gs.a); ct0 = j / gs.a; cstride = gs.S; }
        else { const int j2 = j - gs.a * gs.S; qt = 8 * gs.a + j2 / gs.m; ct0 = x * gs.m + j2 % gs.m; cstride = 8 * gs.m; }
    } else { qt = blockIdx.x % qtiles; ct0 = blockIdx.x / qtiles; cstride = S; }
    const int sp = ct0;
    // staging role: float4 number t + 256 j of a slab = (k = idx / 32, 4 columns at 4 * (idx % 32))
    int sk[GNJ], sc4[GNJ];
#pragma unroll
    for (int j = 0; j < GNJ; ++j) { const int idx = threadIdx.x + 256 * j; sk[j] = idx >> 5; sc4[j] = idx & 31; }
    const float* qsrc = qT + (int64_t)qt * Kpad * GT;
#ifdef KGE_GEMM_32X32
    constexpr int NB = 2;            // column blocks per wave (32 wide)
    const int li = lane & 31, lk = lane >> 5;
#else
    constexpr int NB = 4;            // (16 wide)
    const int lcol = lane & 15;
    const int lk4 = lane >> 4;       // k index of this lane's operands inside a 16x16x4 step
#endif
    float thr[NB], qn2[NB];
    int cnt[NB], tcnt[NB];
    // query column (inside the workgroup's 128) that column block ni holds for this lane
#ifdef KGE_GEMM_32X32
    auto qcol = [&](int ni) { return wc * 64 + ni * 32 + li; };
#else
    auto qcol = [&](int ni) { return wc * 64 + 4 * lcol + ni; };   // four consecutive columns per lane: one 16-byte LDS read
#endif
#pragma unroll
    for (int ni = 0; ni < NB; ++ni) {
        const int64_t q = (int64_t)qt * GT + qcol(ni);
        cnt[ni] = 0;
        tcnt[ni] = 0;
        thr[ni] = (!WRITE && q < nq) ? st[q] : 0.f;
        qn2[ni] = (SQM && q < nq) ? qn[q] : 0.f;
    }
    // energy of one accumulator element: -dot, or the squared distance -(margin - (|q|^2 + |c|^2 - 2 dot))
    auto energy = [&](float dot, int ni, float cn2) {
        if constexpr (SQM) return pair_post<POST>(pair_finish<F_SQM>(sqm_from_dot(dot, qn2[ni], cn2), margin), 1.0f);
        else return pair_post<POST>(-dot, 1.0f);
    };
    const int nslab = Kpad / GKS + ((Kpad % GKS) ? 1 : 0);
    // ONE software pipeline over all (candidate tile, K slab) steps of this workgroup: the loads of step g + 1 -- which may
    // be the first slab of the NEXT candidate tile -- are in flight while step g runs on the matrix cores, so there is no
    // fill / drain bubble at tile boundaries; LDS buffers alternate across the whole sequence (one barrier per step)
    const int64_t my_tiles = sp < ctiles ? (ctiles - sp + cstride - 1) / cstride : 0;
    const int64_t nsteps = my_tiles * nslab;
    typedef float gvec4 __attribute__((ext_vector_type(4)));   // (a native vector: a float4 STRUCT copy from global memory is a memcpy the
    gvec4 ra[GNJ], rb[GNJ];                                      //  optimiser kept in scratch once the loads became unconditional)
    int64_t ld_ct = sp, cu_ct = sp;   // candidate tile / slab of the next load step and of the current compute step (steps are
    int ld_sl = 0, cu_sl = 0;         // visited in order: counters instead of a 64-bit division per step)
    // a lane's two float4 of a slab sit at FIXED offsets from a base that is the same for the whole workgroup: scalar base
    // (advanced by one slab, or re-pointed at the next candidate tile pair) + 32-bit lane offset, no per-step address arithmetic
    int la[GNJ], lq[GNJ];
#pragma unroll
    for (int j = 0; j < GNJ; ++j) {
        la[j] = (sc4[j] >> 4) * Kpad * 64 + sk[j] * 64 + (sc4[j] & 15) * 4;   // two 64-candidate tiles of the sweep layout side by side
        lq[j] = sk[j] * GT + sc4[j] * 4;
    }
    const float* ld_c = cand + ld_ct * 2 * Kpad * 64;
    const float* ld_q = qsrc;
    // (round 6: the counters put as many plain VALU instructions as MFMAs into this loop -- zero fills, selects and 64-bit address
    //  arithmetic of predicated loads -- and every one of them costs the in-order wave issue slots between its MFMAs.  Kpad is a whole
    //  number of slabs and a spare tile follows the tables (make_plan), so a step's loads are unconditional: scalar base + fixed lane
    //  offset.  An odd tile count leaves garbage in the rows of the last pair's second half: rows e >= E, masked in the epilogue.)
    auto load_step = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < GNJ; ++j) {
            ra[j] = *reinterpret_cast<const gvec4*>(ld_c + (unsigned)la[j]);
            rb[j] = *reinterpret_cast<const gvec4*>(ld_q + (unsigned)lq[j]);
        }
        if (++ld_sl == nslab) { ld_sl = 0; ld_ct += cstride; ld_c = cand + ld_ct * 2 * Kpad * 64; ld_q = qsrc; }
        else { ld_c += GKS * 64; ld_q += GKS * GT; }
    };
#ifdef KGE_GEMM_32X32
    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = f32x16{0};
    if (nsteps > 0) load_step();
    int buf = 0;
    for (int64_t g = 0; g < nsteps; ++g) {
#pragma unroll
        for (int j = 0; j < GNJ; ++j) {
            *reinterpret_cast<gvec4*>(&sA[buf][sk[j]][sc4[j] * 4]) = ra[j];
            *reinterpret_cast<gvec4*>(&sB[buf][sk[j]][sc4[j] * 4]) = rb[j];
        }
        __syncthreads();   // step g is in LDS; everybody finished reading the buffer that is written next
        if (g + 1 < nsteps) load_step();
        // operands of step kk + 2 are read from LDS before the MFMAs of step kk are issued (register double buffer; the
        // scheduling barrier keeps the compiler from sinking the reads back in front of their use)
        float na0 = sA[buf][lk][wr * 64 + li], na1 = sA[buf][lk][wr * 64 + 32 + li];
        float nb0 = sB[buf][lk][wc * 64 + li], nb1 = sB[buf][lk][wc * 64 + 32 + li];
#pragma unroll
        for (int kk = 0; kk < GKS; kk += 2) {
            const float a0 = na0, a1 = na1, b0 = nb0, b1 = nb1;
            if (kk + 2 < GKS) {
                na0 = sA[buf][kk + 2 + lk][wr * 64 + li]; na1 = sA[buf][kk + 2 + lk][wr * 64 + 32 + li];
                nb0 = sB[buf][kk + 2 + lk][wc * 64 + li]; nb1 = sB[buf][kk + 2 + lk][wc * 64 + 32 + li];
            }
            KGE_KEEP_READS_AHEAD();
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        buf ^= 1;
        if (++cu_sl != nslab) continue;
        cu_sl = 0;
        // ---- last slab of a candidate tile: epilogue.  energy = -dot (+ post-op); the lane owns query column (ni, li),
        // its 16 registers per block are candidate rows
        const int64_t ct = cu_ct;
        cu_ct += cstride;
        const int e_base = (int)(ct * GT) + wr * 64 + 4 * lk;   // candidate ids fit 31 bits (packed keys: < 2^24)
        const int e_lim = (int)E;
        const bool full = ct * GT + GT <= E && (ct * 2 + 1 < ntiles64);
        // (candidate rows outermost: in the squared-distance form |c|^2 of a row is fetched once and used for both query
        // column blocks, without holding 32 of them in registers)
        const int e_pad = (int)(ntiles64 * 64);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int e = e_base + mi * 32 + (reg & 3) + 8 * (reg >> 2);
                float cn2 = 0.f;
                if constexpr (SQM) cn2 = e < e_pad ? cn[e] : 0.f;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const float sc = energy(acc[mi][ni][reg], ni, cn2);
                    if constexpr (WRITE) {
                        const int64_t q = (int64_t)qt * GT + wc * 64 + ni * 32 + li;
                        if (q < nq && e < e_lim) scores_out[q * E + e] = sc;
                    } else {
                        cnt[ni] += (sc < thr[ni] && (full || e < e_lim)) ? 1 : 0;
                        tcnt[ni] += (sc == thr[ni] && (full || e < e_lim)) ? 1 : 0;
                    }
                }
            }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = f32x16{0};
    }
    if constexpr (!WRITE) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int c2 = cnt[ni] + __shfl_xor(cnt[ni], 32, 64);   // the two lanes that own the same query column
            const int t2 = tcnt[ni] + __shfl_xor(tcnt[ni], 32, 64);
            const int64_t q = (int64_t)qt * GT + wc * 64 + ni * 32 + li;
            if (lk == 0 && q < nq && c2 != 0) atomicAdd(rcount + q, c2);
            if (tcount && lk == 0 && q < nq && t2 != 0) atomicAdd(tcount + q, t2);
        }
    }
}
#else
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 acc[4][4];   // 64 x 64 sub-tile of the wave as 4 x 4 blocks of 16 x 16: candidate rows mi, query columns ni
    f32x4 tot[4][4];   // KGE_GEMM_CHUNK: the finished chunks of every energy
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) { acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f}; tot[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    if (nsteps > 0) load_step();
    // (the LDS buffer of a step is a compile-time constant: its addresses are one lane base + immediates.  A register ring of two slabs
    //  in flight was measured on the way -- no gain, profiles/r06_experiments.md section 9)
    auto body = [&](auto buf_, int64_t g) __attribute__((always_inline)) {
        constexpr int buf = decltype(buf_)::value;
#pragma unroll
        for (int j = 0; j < GNJ; ++j) {
            *reinterpret_cast<gvec4*>(&sA[buf][sk[j]][sc4[j] * 4]) = ra[j];
            *reinterpret_cast<gvec4*>(&sB[buf][sk[j]][sc4[j] * 4]) = rb[j];
        }
        __syncthreads();   // step g is in LDS; everybody finished reading the buffer that is written next
        if (g + 1 < nsteps) load_step();
        // operands of k-step kk + 4 are read from LDS before the MFMAs of k-step kk are issued (register double buffer)
        float na[4], nb[4];
        // row block mi of the wave holds candidate rows 4 r + mi (r = row inside the MFMA block), column block ni query columns
        // 4 c + ni: a lane's four A operands (and four B operands) of a k-step are 16 consecutive bytes of the k-major slab
#define KGE_GEMM_READ(K)                                                                                                   \
    {                                                                                                                      \
        const float4 va = *reinterpret_cast<const float4*>(&sA[buf][(K) + lk4][wr * 64 + 4 * lcol]);                       \
        const float4 vb = *reinterpret_cast<const float4*>(&sB[buf][(K) + lk4][wc * 64 + 4 * lcol]);                       \
        na[0] = va.x; na[1] = va.y; na[2] = va.z; na[3] = va.w;                                                            \
        nb[0] = vb.x; nb[1] = vb.y; nb[2] = vb.z; nb[3] = vb.w;                                                            \
    }
        KGE_GEMM_READ(0)
        // KGE_GEMM_CHUNK: does this slab begin / complete a chunk of k (the last slab of a tile completes one whatever its index)?
        constexpr int SPC = kGemmChunk > 0 ? kGemmChunk / GKS : 1;   // slabs per chunk
        const bool c_first = kGemmChunk > 0 && cu_sl % SPC == 0;
        const bool c_last = kGemmChunk > 0 && ((cu_sl + 1) % SPC == 0 || cu_sl + 1 == nslab);
#pragma unroll
        for (int kk = 0; kk < GKS; kk += 4) {
            float a4[4], b4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a4[i] = na[i]; b4[i] = nb[i]; }
            if (kk + 4 < GKS) KGE_GEMM_READ(kk + 4)
            KGE_KEEP_READS_AHEAD();
            if (kGemmChunk > 0 && kk == 0 && c_first) {
                // a chunk's chain starts from the constant 0 (the C operand of the MFMA): the accumulators are never cleared
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[mi], b4[ni], zero, 0, 0, 0);
            } else {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[mi], b4[ni], acc[mi][ni], 0, 0, 0);
            }
        }
        if (c_last) {   // the chunk is complete: its chains join the running totals (32 packed adds per lane)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) tot[mi][ni] += acc[mi][ni];
        }
        if (++cu_sl != nslab) return;
        cu_sl = 0;
        // ---- last slab of a candidate tile: epilogue.  energy = -dot (+ post-op); the lane owns query column (ni, lcol), its 4
        // registers per block are candidate rows 4 * lk4 + reg of block mi
        const int64_t ct = cu_ct;
        cu_ct += cstride;
        // candidate ids fit 31 bits (packed keys: < 2^24); MFMA block row 4 lk4 + reg of block mi = candidate row 4 (4 lk4 + reg) + mi
        const int e_base = (int)(ct * GT) + wr * 64 + 16 * lk4;
        const int e_lim = (int)E;
        const bool full = ct * GT + GT <= E && (ct * 2 + 1 < ntiles64);
        const int e_pad = (int)(ntiles64 * 64);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int e = e_base + 4 * reg + mi;
                float cn2 = 0.f;
                if constexpr (SQM) cn2 = e < e_pad ? cn[e] : 0.f;
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const float sc = energy(kGemmChunk > 0 ? tot[mi][ni][reg] : acc[mi][ni][reg], ni, cn2);
                    if constexpr (WRITE) {
                        const int64_t q = (int64_t)qt * GT + qcol(ni);
                        if (q < nq && e < e_lim) scores_out[q * E + e] = sc;
                    } else {
                        cnt[ni] += (sc < thr[ni] && (full || e < e_lim)) ? 1 : 0;
                        tcnt[ni] += (sc == thr[ni] && (full || e < e_lim)) ? 1 : 0;
                    }
                }
            }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                if constexpr (kGemmChunk > 0) tot[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};   // (acc restarts from the constant 0)
                else acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
    };
    for (int64_t g = 0; g < nsteps; g += 2) {
        body(std::integral_constant<int, 0>{}, g);
        if (g + 1 < nsteps) body(std::integral_constant<int, 1>{}, g + 1);
    }
    if constexpr (!WRITE) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            int c2 = cnt[ni] + __shfl_xor(cnt[ni], 16, 64);   // the four lanes that own the same query column
            c2 += __shfl_xor(c2, 32, 64);
            int t2 = tcnt[ni] + __shfl_xor(tcnt[ni], 16, 64);
            t2 += __shfl_xor(t2, 32, 64);
            const int64_t q = (int64_t)qt * GT + qcol(ni);
            if (lk4 == 0 && q < nq && c2 != 0) atomicAdd(rcount + q, c2);
            if (tcount && lk4 == 0 && q < nq && t2 != 0) atomicAdd(tcount + q, t2);
        }
    }
}
#undef KGE_GEMM_READ

#endif

// ties (may be NULL): int32 [2, n] = per head sweep / tail sweep the number of OTHER candidates whose energy equals the true one's
__global__ void k_eval_finalize(const int32_t* __restrict__ rcount, const int32_t* __restrict__ fcount, int64_t n,
                                int32_t* __restrict__ ranks, const int32_t* __restrict__ tcount, int32_t* __restrict__ ties) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (ties) {
        ties[i] = max(0, tcount[2 * i + 1] - 1);
        ties[n + i] = max(0, tcount[2 * i] - 1);
    }
    const int32_t rt = rcount[2 * i], rh = rcount[2 * i + 1];
    ranks[i] = rh;                               // rank_head
    ranks[n + i] = rt;                           // rank_tail
    ranks[2 * n + i] = rh - fcount[2 * i + 1];   // filtered head
    ranks[3 * n + i] = rt - fcount[2 * i];       // filtered tail
}

// one-sided sweeps: query row i = triple i; ranks [2, n] = rank, filtered rank; ties [n]
__global__ void k_eval_finalize_side(const int32_t* __restrict__ rcount, const int32_t* __restrict__ fcount, int64_t n,
                                     int32_t* __restrict__ ranks, const int32_t* __restrict__ tcount, int32_t* __restrict__ ties) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (ties) ties[i] = max(0, tcount[i] - 1);
    ranks[i] = rcount[i];
    ranks[n + i] = rcount[i] - fcount[i];
}

// ------------------------------------------------------------------ ranks from materialised score rows
// (models without a pre-contracted sweep form, i.e. NTN: scores come from kge_score_forward over all candidates)
__global__ __launch_bounds__(256) void k_rank_from_scores(const float* __restrict__ scores, int64_t nq, int64_t E,
                                                          const int64_t* __restrict__ truth, const int64_t* __restrict__ off,
                                                          const int32_t* __restrict__ ids, int32_t* __restrict__ rank,
                                                          int32_t* __restrict__ frank) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    const float* s = scores + q * E;
    const int64_t tr = truth[q];
    const float st = s[tr];
    int cnt = 0, fc = 0;
    for (int64_t e = lane; e < E; e += 64) cnt += s[e] < st ? 1 : 0;
    if (off) {
        for (int64_t j = off[q] + lane; j < off[q + 1]; j += 64) {
            const int64_t e = ids[j];
            fc += (e != tr && s[e] < st) ? 1 : 0;
        }
    }
    cnt = (int)wave_sum((float)cnt);
    fc = (int)wave_sum((float)fc);
    if (lane == 0) { rank[q] = cnt; frank[q] = cnt - fc; }
}

int launch_rank_from_scores(const float* scores, int64_t nq, int64_t E, const int64_t* truth, const int64_t* off,
                            const int32_t* ids, int32_t* rank, int32_t* frank, hipStream_t s) {
    hipLaunchKernelGGL(k_rank_from_scores, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, scores, nq, E, truth, off, ids, rank,
                       frank);
    return check_launch("k_rank_from_scores");
}

// ------------------------------------------------------------------ host side
static void fill_prep(const kge_model_desc* m, const EvalPlan& p, PrepArgs* a) {
    a->nseg = 1; a->dot_tab = nullptr; a->normalize = 0; a->want_n2 = 0;
    a->E = p.E; a->K = p.K; a->Kpad = p.Kpad;
    for (int s = 0; s < 4; ++s) { a->seg[s] = nullptr; a->seg_dim[s] = 0; }
    a->seg[0] = m->tables[0]; a->seg_dim[0] = m->dim;
    switch (m->model) {
        case KGE_TRANSE: case KGE_TRANSM: a->normalize = 1; break;
        case KGE_CP:  // [sub | obj]: the head sweep reads the first half, the tail sweep the second (zero query halves)
            a->nseg = 2; a->seg[1] = m->tables[2]; a->seg_dim[1] = m->dim; break;
        case KGE_SIMPLE: case KGE_SIMPLE_IGNR:  // [tail-role row | head-role row] of entity e
            a->nseg = 2; a->seg[0] = m->tables[1]; a->seg[1] = m->tables[0]; a->seg_dim[1] = m->dim; break;
        case KGE_QUATE:
            a->nseg = 4;
            for (int q = 1; q < 4; ++q) { a->seg[q] = m->tables[q]; a->seg_dim[q] = m->dim; }
            break;
        case KGE_TRANSD: a->dot_tab = m->tables[2]; break;
        case KGE_COMPLEX: case KGE_ROTATE:
            a->nseg = 2; a->seg[1] = m->tables[1]; a->seg_dim[1] = m->dim; break;
        case KGE_ANALOGY:
            a->nseg = 3; a->seg[1] = m->tables[2]; a->seg_dim[1] = m->dim / 2;
            a->seg[2] = m->tables[3]; a->seg_dim[2] = m->dim / 2; break;
        case KGE_HEAD_1N_INTERNAL:   // candidate row = [ent row | bias]: the bias joins the logit as the chain's last term, bias * 1
            if (m->tables[1]) { a->nseg = 2; a->seg[1] = m->tables[1]; a->seg_dim[1] = 1; }
            break;
        default: break;
    }
}

// Dot-product forms go to the matrix cores when there are enough queries to fill 128-wide tiles; KGE_EVAL_GEMM=0 / 1 forces
// the choice (A/B runs).
static bool use_gemm_sweep(const EvalPlan& p, int64_t nq) {
    const int force = switch_value("EVAL_GEMM");
    if (force >= 0) return force == 1;
    return nq >= 512 && p.Kpad >= 32;
}

template <int FORM, int XFORM, int POST = P_NONE>
static void launch_tf_and_sweep(const EvalPlan& p, const kge_model_desc* m, const int64_t* triples,
                                const int64_t* tail_off, const int32_t* tail_ids, const int64_t* head_off,
                                const int32_t* head_ids, float* scores_out, hipStream_t s,
                                const int32_t* group_of_triple = nullptr, const int32_t* qdesc = nullptr,
                                int64_t n_qblocks = 0) {
    constexpr int QT = XFORM == X_NONE ? QT_PLAIN : QT_XF;
    const int64_t nq = p.nq;
    const int qblocks = qdesc ? (int)n_qblocks : (int)((nq + QT - 1) / QT);
    // tile splits S: every wave gets >= 1 tile; aim at >= ~8 waves of workgroups per CU so the last round is cheap
    const int64_t tiles_per_wave_pass = XFORM == X_NONE ? 2 : 1;
    const int64_t max_split = (p.ntiles + 4 * tiles_per_wave_pass - 1) / (4 * tiles_per_wave_pass);
    const int64_t qgroups = (qblocks + 255) / 256;
    int64_t S = (48 + qgroups - 1) / qgroups;
    if (S > max_split) S = max_split;
    if (S < 1) S = 1;
    const unsigned grid = (unsigned)(qgroups * 256 * S);
    if constexpr ((FORM == F_NEGDOT || FORM == F_SQM) && XFORM == X_NONE && POST != P_SCALE) {
        if (qdesc == nullptr && p.qT != nullptr && p.Kpad % GKS == 0 && use_gemm_sweep(p, nq)) {   // the dot-product based sweeps on the matrix cores
            constexpr bool SQM = FORM == F_SQM;
            const int qtiles = (int)((nq + GT - 1) / GT);
            const int64_t ctiles = (p.ntiles + 1) / 2;
            // candidate-tile splits: fill the 4 x 256 resident workgroup slots WITHOUT spilling into a second, mostly empty
            // generation (1029 workgroups on 1024 slots cost 25 % more than 980)
            int64_t S2 = ((KGE_GEMM_OCC) * 256) / qtiles;
            if (S2 > ctiles) S2 = ctiles;
            if (S2 < 1) S2 = 1;
            unsigned ggrid = 0;
            const GemmSplit gs = gemm_split(qtiles, ctiles, (KGE_GEMM_OCC) * 32, &ggrid, (int)S2);
            hipLaunchKernelGGL(k_eval_qt, dim3((unsigned)(qtiles * 2), (unsigned)((p.Kpad + 63) / 64)), dim3(256), 0, s, p.qvec, nq,
                               p.Kpad, p.qT);
            if (SQM)   // |q|^2 per query (p.qscale is free in this form); |c|^2 per candidate was left in p.aux by k_eval_prepare
                hipLaunchKernelGGL(k_eval_qnorm, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, p.qvec, nq, p.Kpad, p.qscale);
            if (scores_out == nullptr) {
                PrepArgs pa;
                fill_prep(m, p, &pa);
                hipLaunchKernelGGL((k_eval_target_filter_chain<FORM, POST>), dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, pa, p.aux,
                                   p.qvec, p.qscale, triples, p.n, p.Kpad, m->margin, tail_off, tail_ids, head_off, head_ids, p.st,
                                   p.fcount, p.only);
                hipLaunchKernelGGL((k_eval_gemm<false, POST, SQM>), dim3(ggrid), dim3(256), 0, s, p.cand, p.qT, p.st,
                                   nq, p.E, p.ntiles, p.Kpad, qtiles, (int)S2, gs, p.rcount, p.tcount, nullptr, p.qscale, p.aux, m->margin);
            } else {
                hipLaunchKernelGGL((k_eval_gemm<true, POST, SQM>), dim3(ggrid), dim3(256), 0, s, p.cand, p.qT, p.st,
                                   nq, p.E, p.ntiles, p.Kpad, qtiles, (int)S2, gs, p.rcount, nullptr, scores_out, p.qscale, p.aux, m->margin);
            }
            return;
        }
    }
    if (scores_out == nullptr)
        hipLaunchKernelGGL((k_eval_target_filter<FORM, XFORM, POST>), dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, p.cand,
                           p.aux, p.qvec, p.qscale, triples, p.n, p.Kpad, p.QV, m->margin, tail_off, tail_ids, head_off, head_ids,
                           p.st, p.fcount, group_of_triple, p.table_stride, p.only);
    if (scores_out == nullptr) {
        hipLaunchKernelGGL((k_eval_sweep<FORM, XFORM, QT, false, POST>), dim3(grid), dim3(256), 0, s, p.cand, p.aux, p.qvec,
                           p.qscale, p.st, nq, p.E, p.ntiles, p.Kpad, p.QV, m->margin, (int)S, qblocks, p.rcount, p.tcount, nullptr,
                           qdesc, p.table_stride);
    } else {
        hipLaunchKernelGGL((k_eval_sweep<FORM, XFORM, QT, true, POST>), dim3(grid), dim3(256), 0, s, p.cand, p.aux, p.qvec,
                           p.qscale, p.st, nq, p.E, p.ntiles, p.Kpad, p.QV, m->margin, (int)S, qblocks, p.rcount, nullptr, scores_out,
                           qdesc, p.table_stride);
    }
}

static int run_pipeline(const kge_model_desc* m, const int64_t* triples, int64_t n, const int64_t* tail_off,
                        const int32_t* tail_ids, const int64_t* head_off, const int32_t* head_ids, void* ws,
                        size_t ws_bytes, int32_t* ranks, int32_t* ties, float* scores_out, hipStream_t s, int side = 2) {
    EvalPlan p;
    if (!make_plan(m, n, ws, &p)) { set_error("kge_eval: model %d has no sweep form", m->model); return -1; }
    if (side != 2) {   // one-sided: scores [n, E], or ranks [2, n] = (rank, filtered rank) of that side (ties [n])
        if ((side != 0 && side != 1) || m->model == KGE_TRANSR) {
            set_error("kge_eval: one-sided sweeps take side 0 = tail or 1 = head (TransR projects per call: use both sides)");
            return -1;
        }
        p.nq = n;
        p.only = side;
    }
    if (ws == nullptr || ws_bytes < p.bytes) {
        set_error("kge_eval: workspace too small (%zu < %zu)", ws_bytes, p.bytes);
        return -1;
    }
    if (n <= 0) return 0;
    if (m->model == KGE_TRANSR) {  // candidates projected by the call's relation matrix (all triples share it)
        int rc = launch_transr_eval_prepare(m, triples, n, nullptr, 1, p.Kpad, p.ntiles, p.cand, p.qvec, p.qscale, s);
        if (rc) return rc;
    } else {
    PrepArgs pa;
    fill_prep(m, p, &pa);
    pa.want_n2 = (p.form == F_SQM && p.xform == X_NONE && p.qT != nullptr && use_gemm_sweep(p, p.nq)) ? 1 : 0;
    bool vec = !pa.normalize && !pa.dot_tab;
    for (int sg = 0; sg < pa.nseg; ++sg) vec = vec && pa.seg_dim[sg] % 4 == 0 && (uintptr_t)pa.seg[sg] % 16 == 0;
    if (vec) {   // 16-byte re-layout, K split over grid.y
        const int64_t nchunks = (p.Kpad + 127) / 128;
        int64_t nky = min(nchunks, (2048 + p.ntiles - 1) / p.ntiles);
        if (pa.want_n2 && nky > 1) {   // per-split partial norms borrow the (not yet written) query-tile buffer
            const int64_t cap = (int64_t)((2 * n + 127) / 128) * p.Kpad * 128 / (p.ntiles * 64);
            if (cap < nky) nky = cap;
        }
        if (nky < 1) nky = 1;
        pa.kper = (int)((nchunks + nky - 1) / nky * 128);
        nky = (p.Kpad + pa.kper - 1) / pa.kper;
        float* part = (pa.want_n2 && nky > 1) ? p.qT : nullptr;
        hipLaunchKernelGGL(k_eval_prepare4, dim3((unsigned)p.ntiles, (unsigned)nky), dim3(256), 0, s, pa, p.cand, p.aux, part);
        if (part)
            hipLaunchKernelGGL(k_eval_cnorm_parts, dim3((unsigned)((p.ntiles * 64 + 255) / 256)), dim3(256), 0, s, part, (int)nky,
                               p.ntiles * 64, p.aux);
    } else {
    // K split: only where nothing but the re-layout happens per element (no normalisation, no TransD dot product)
    int ksplit = 1;
    if (!pa.normalize && !pa.dot_tab) {
        ksplit = (int)min((int64_t)((p.Kpad + 63) / 64), (4096 + p.ntiles - 1) / p.ntiles);
        if (ksplit < 1) ksplit = 1;
    }
    pa.kper = ((p.Kpad + ksplit - 1) / ksplit + 63) / 64 * 64;
    ksplit = (p.Kpad + pa.kper - 1) / pa.kper;
    hipLaunchKernelGGL(k_eval_prepare, dim3((unsigned)p.ntiles, (unsigned)ksplit), dim3(256), 0, s, pa, p.cand, p.aux);
    if (ksplit > 1 && pa.want_n2) {
        hipLaunchKernelGGL(k_eval_cnorm, dim3((unsigned)(p.ntiles * 16)), dim3(256), 0, s, pa, p.aux);   // 64 rows per tile, 4 per block
    }
    }
    const DeviceModel dm = to_device_model(m);
    const unsigned qb = (unsigned)((n + 3) / 4);
#define KGE_Q(MID) case MID: hipLaunchKernelGGL((k_eval_queries<MID>), dim3((MID == KGE_TRANSE || MID == KGE_TRANSH || MID == KGE_TRANSD || MID == KGE_TRANSM) ? qb : (unsigned)n), dim3(256), 0, s, dm, triples, n, p.K, p.Kpad, p.QV, p.qvec, p.qscale, side); break;
    switch (m->model) {
        KGE_Q(KGE_TRANSE) KGE_Q(KGE_TRANSH) KGE_Q(KGE_TRANSD) KGE_Q(KGE_ROTATE) KGE_Q(KGE_DISTMULT)
        KGE_Q(KGE_COMPLEX) KGE_Q(KGE_ANALOGY) KGE_Q(KGE_RESCAL)
        KGE_Q(KGE_TRANSM) KGE_Q(KGE_CP) KGE_Q(KGE_SIMPLE) KGE_Q(KGE_SIMPLE_IGNR) KGE_Q(KGE_QUATE) KGE_Q(KGE_HEAD_1N_INTERNAL)
        default: set_error("kge_eval: unsupported model %d", m->model); return -1;
    }
#undef KGE_Q
    }
    if (scores_out == nullptr) (void)hipMemsetAsync(p.rcount, 0, (size_t)4 * n * sizeof(int32_t), s);   // rcount | tcount
#define KGE_S(F, X) launch_tf_and_sweep<F, X>(p, m, triples, tail_off, tail_ids, head_off, head_ids, scores_out, s)
    if (p.post == P_SCALE) {
        if (p.form == F_L1) launch_tf_and_sweep<F_L1, X_NONE, P_SCALE>(p, m, triples, tail_off, tail_ids, head_off, head_ids, scores_out, s);
        else launch_tf_and_sweep<F_L2, X_NONE, P_SCALE>(p, m, triples, tail_off, tail_ids, head_off, head_ids, scores_out, s);
    } else if (p.post == P_CLAMP) {
        launch_tf_and_sweep<F_NEGDOT, X_NONE, P_CLAMP>(p, m, triples, tail_off, tail_ids, head_off, head_ids, scores_out, s);
    } else if (p.post == P_SIGMOID) {
        launch_tf_and_sweep<F_NEGDOT, X_NONE, P_SIGMOID>(p, m, triples, tail_off, tail_ids, head_off, head_ids, scores_out, s);
    } else if (p.xform == X_NONE) {
        switch (p.form) {
            case F_L1: KGE_S(F_L1, X_NONE); break;
            case F_L2: KGE_S(F_L2, X_NONE); break;
            case F_SQM: KGE_S(F_SQM, X_NONE); break;
            default: KGE_S(F_NEGDOT, X_NONE); break;
        }
    } else if (p.xform == X_TRANSH) {
        if (p.form == F_L1) KGE_S(F_L1, X_TRANSH); else KGE_S(F_L2, X_TRANSH);
    } else {
        if (p.form == F_L1) KGE_S(F_L1, X_TRANSD); else KGE_S(F_L2, X_TRANSD);
    }
#undef KGE_S
    if (scores_out == nullptr) {
        if (side == 2) hipLaunchKernelGGL(k_eval_finalize, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p.rcount, p.fcount, n, ranks, p.tcount, ties);
        else hipLaunchKernelGGL(k_eval_finalize_side, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p.rcount, p.fcount, n, ranks, p.tcount, ties);
    }
    return check_launch("kge_eval pipeline");
}

int launch_eval_ranks(const kge_model_desc* m, const int64_t* triples, int64_t n, const int64_t* tail_off,
                      const int32_t* tail_ids, const int64_t* head_off, const int32_t* head_ids, void* ws,
                      size_t ws_bytes, int32_t* ranks, int32_t* ties, hipStream_t s) {
    if (m->model == KGE_NTN) {   // (the NTN sweep does not count ties: reported as unknown, -1)
        if (ties) (void)hipMemsetAsync(ties, 0xFF, (size_t)2 * n * sizeof(int32_t), s);
        return launch_ntn_eval_ranks(m, triples, n, tail_off, tail_ids, head_off, head_ids, ws, ws_bytes, ranks, s);
    }
    return run_pipeline(m, triples, n, tail_off, tail_ids, head_off, head_ids, ws, ws_bytes, ranks, ties, nullptr, s);
}

// TransR over several relation groups in one pass: one projected candidate table per group, every sweep workgroup
// bound to one group's queries by its descriptor {group, first query, query count}.
int launch_eval_ranks_grouped(const kge_model_desc* m, const int64_t* triples, int64_t n, const int32_t* group_of_triple,
                              const int64_t* group_rel, int64_t n_groups, const int32_t* qblocks, int64_t n_qblocks,
                              const int64_t* tail_off, const int32_t* tail_ids, const int64_t* head_off,
                              const int32_t* head_ids, void* ws, size_t ws_bytes, int32_t* ranks, int32_t* ties, hipStream_t s) {
    if (m->model != KGE_TRANSR && m->model != KGE_TRANSH && m->model != KGE_TRANSD) {
        set_error("kge_eval_ranks_grouped: TransR / TransH / TransD only (the other models' candidates do not depend on the relation)");
        return -1;
    }
    if (n_groups < 1 || n_groups > 65535) { set_error("kge_eval_ranks_grouped: %lld relation groups per call (1..65535)", (long long)n_groups); return -1; }
    EvalPlan p;
    // tables = n_groups + (n_groups == 1): a single group must still plan per-group (already transformed) candidates
    if (!make_plan(m, n, ws, &p, n_groups == 1 ? 2 : n_groups)) { set_error("kge_eval: model %d has no sweep form", m->model); return -1; }
    if (ws == nullptr || ws_bytes < p.bytes) {
        set_error("kge_eval_ranks_grouped: workspace too small (%zu < %zu)", ws_bytes, p.bytes);
        return -1;
    }
    if (m->model == KGE_TRANSR) {
        int rc = launch_transr_eval_prepare(m, triples, n, group_rel, n_groups, p.Kpad, p.ntiles, p.cand, p.qvec, p.qscale, s);
        if (rc) return rc;
    } else {
        XfPrepArgs xa;
        xa.ent = m->tables[0]; xa.group_rel = group_rel; xa.E = p.E; xa.K = p.K; xa.Kpad = p.Kpad; xa.table_stride = p.table_stride;
        xa.xform = m->model == KGE_TRANSH ? X_TRANSH : X_TRANSD;
        xa.vec_tab = m->model == KGE_TRANSH ? m->tables[2] : m->tables[3];   // w  /  rel_mappings
        xa.ent_map = m->model == KGE_TRANSD ? m->tables[2] : nullptr;       // ent_mappings
        hipLaunchKernelGGL(k_eval_prepare_xf, dim3((unsigned)p.ntiles, (unsigned)n_groups), dim3(256), 0, s, xa, p.cand);
        const DeviceModel dm = to_device_model(m);
        const unsigned qb = (unsigned)((n + 3) / 4);
        if (m->model == KGE_TRANSH)
            hipLaunchKernelGGL((k_eval_queries<KGE_TRANSH>), dim3(qb), dim3(256), 0, s, dm, triples, n, p.K, p.Kpad, p.QV, p.qvec, p.qscale, 2);
        else
            hipLaunchKernelGGL((k_eval_queries<KGE_TRANSD>), dim3(qb), dim3(256), 0, s, dm, triples, n, p.K, p.Kpad, p.QV, p.qvec, p.qscale, 2);
    }
    (void)hipMemsetAsync(p.rcount, 0, (size_t)4 * n * sizeof(int32_t), s);
    if (p.form == F_L1)
        launch_tf_and_sweep<F_L1, X_NONE>(p, m, triples, tail_off, tail_ids, head_off, head_ids, nullptr, s, group_of_triple,
                                          qblocks, n_qblocks);
    else
        launch_tf_and_sweep<F_L2, X_NONE>(p, m, triples, tail_off, tail_ids, head_off, head_ids, nullptr, s, group_of_triple,
                                          qblocks, n_qblocks);
    hipLaunchKernelGGL(k_eval_finalize, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p.rcount, p.fcount, n, ranks, p.tcount, ties);
    return check_launch("kge_eval grouped pipeline");
}

// scores: float [2n, E]: row 2i = tail-sweep energies of triple i, row 2i+1 = head-sweep energies (side 2);
// side 0 / 1: float [n, E], the tail / head sweep only
int launch_eval_sweep_scores(const kge_model_desc* m, const int64_t* triples, int64_t n, void* ws, size_t ws_bytes,
                             float* scores, hipStream_t s, int side) {
    if (m->model == KGE_NTN) {
        if (side != 2) { set_error("kge_eval_sweep_scores_side: the NTN sweep computes both sides per call"); return -1; }
        return launch_ntn_eval_scores(m, triples, n, ws, ws_bytes, scores, s);
    }
    return run_pipeline(m, triples, n, nullptr, nullptr, nullptr, nullptr, ws, ws_bytes, nullptr, nullptr, scores, s, side);
}

static kge_model_desc head_rank_desc(const float* x, int dim, const float* ent, int64_t E, const float* bias) {
    kge_model_desc m;
    memset(&m, 0, sizeof(m));
    m.model = KGE_HEAD_1N_INTERNAL;
    m.tot_entity = E; m.tot_relation = 1; m.dim = dim; m.rel_dim = dim;
    m.tables[0] = const_cast<float*>(ent); m.tables[1] = const_cast<float*>(bias); m.tables[2] = const_cast<float*>(x);
    return m;
}

size_t head_rank_workspace_bytes(int64_t n, int dim, int64_t E, bool has_bias) {
    static const float one = 1.0f;   // (only its non-NULLness is read: the plan adds the bias column)
    const kge_model_desc m = head_rank_desc(nullptr, dim, nullptr, E, has_bias ? &one : nullptr);
    EvalPlan p;
    return make_plan(&m, n, nullptr, &p) ? p.bytes : 0;
}

// triples: int64 [n, 3] whose column 2 holds the true entity of row i (columns 0 / 1 are not read beyond the id check)
int launch_head_rank(const float* x, int64_t n, int dim, const float* ent, int64_t E, const float* bias, const int64_t* triples,
                     const int64_t* off, const int32_t* ids, void* ws, size_t ws_bytes, int32_t* ranks, int32_t* ties, float* energies,
                     hipStream_t s) {
    const kge_model_desc m = head_rank_desc(x, dim, ent, E, bias);
    return run_pipeline(&m, triples, n, off, ids, nullptr, nullptr, ws, ws_bytes, ranks, ties, energies, s, 0);
}

}  // namespace kge
