// kge_ntn_eval.hip -- filtered-rank evaluation for NTN as a pre-contracted MFMA sweep.
//
// The reference ranks a test triple by pushing all E candidate triples through NTN.forward (utils/evaluator.py:254-272
// -> pairwise.py:919-960): 2*k_r*d^2 flop per candidate and a [k_r, E, d] temporary (598 MB at FB15k, d = k_r = 100).
// With the query side fixed, the bilinear term factors:
//     tail sweep (h, r, ?):  h^^T W_s t^_e = A_t[s] . t^_e      A_t[s][c] = sum_i h^_i W[s][i][c]     ([k_r, d] per query)
//     head sweep (?, r, t):  h^_e^T W_s t^ = h^_e . A_h[s]      A_h[s][c] = sum_j W[s][c][j] t^_j
// so a sweep is ONE [E, d] x [d, k_r] GEMM per query (2*k_r*d flop per candidate, d-fold less) on v_mfma_f32_32x32x2_f32
// with the epilogue  energy_e = - sum_s r^_s tanh(bil[e][s] + lin_q[s] + lin_e[s])  fused in, where lin_q = b + h^ M1
// (tail sweep) or b + t^ M2 (head sweep) and lin_e = t^_e M2 / h^_e M1 are [E, k_r] tables built once per evaluation.
// Queries are processed in chunks; scores of a chunk are materialised ([2*chunk, E] fp32) and ranked by
// k_rank_from_scores (count + CSR filter), so target / filter scores are trivially the sweep's own values.
#include "kge_internal.h"

namespace kge {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NQT = 32;          // queries per contraction tile
constexpr int kNtnChunk = 256;   // test triples per chunk (512 query sides)

struct NtnEvalWs {
    float *cand, *EM1, *EM2, *Hq, *Tq, *QA, *qlin, *qr, *scores;
    int64_t* truth;
    int32_t *rank, *frank;
    int Kpad, krp, chunk;
    int64_t ntiles;
    size_t bytes;
};

static size_t a256(size_t x) { return (x + 255) & ~(size_t)255; }

static void ntn_eval_plan(const kge_model_desc* m, int64_t n, void* ws, NtnEvalWs* w, bool want_all_scores) {
    const int d = m->dim, kr = m->rel_dim;
    w->Kpad = (d + 7) / 8 * 8;
    w->krp = (kr + 31) / 32 * 32;
    w->chunk = (int)(n < kNtnChunk ? n : kNtnChunk);
    if (w->chunk < 1) w->chunk = 1;
    w->ntiles = (m->tot_entity + 63) / 64;
    size_t off = 0;
    char* base = (char*)ws;
    auto take = [&](size_t b) { char* p = base ? base + off : nullptr; off += a256(b); return p; };
    const int64_t E = m->tot_entity, C = w->chunk;
    w->cand = (float*)take((size_t)w->ntiles * w->Kpad * 64 * 4);
    w->EM1 = (float*)take((size_t)E * w->krp * 4);
    w->EM2 = (float*)take((size_t)E * w->krp * 4);
    w->Hq = (float*)take((size_t)C * w->Kpad * 4);
    w->Tq = (float*)take((size_t)C * w->Kpad * 4);
    w->QA = (float*)take((size_t)2 * C * w->krp * w->Kpad * 4);
    w->qlin = (float*)take((size_t)2 * C * w->krp * 4);
    w->qr = (float*)take((size_t)2 * C * w->krp * 4);
    w->scores = want_all_scores ? nullptr : (float*)take((size_t)2 * C * E * 4);
    w->truth = (int64_t*)take((size_t)2 * C * 8);
    w->rank = (int32_t*)take((size_t)2 * C * 4);
    w->frank = (int32_t*)take((size_t)2 * C * 4);
    w->bytes = off;
}

size_t ntn_eval_workspace_bytes(const kge_model_desc* m, int64_t n) {
    NtnEvalWs w;
    ntn_eval_plan(m, n, nullptr, &w, false);
    return w.bytes;
}

// ---- normalised candidates in the sweep layout [tile64][k][64] and the per-entity linear maps (once per evaluation)
__global__ __launch_bounds__(256) void k_ntn_ent_tables(const float* __restrict__ ent, const float* __restrict__ M1,
                                                        const float* __restrict__ M2, int64_t E, int d, int kr, int Kpad,
                                                        int krp, float* __restrict__ cand, float* __restrict__ EM1,
                                                        float* __restrict__ EM2) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 4 + wave;
    float* sx = smem + wave * Kpad;
    if (e < E) {
        const float* x = ent + e * d;
        float n2 = 0.f;
        for (int c = lane; c < d; c += 64) n2 = fmaf(x[c], x[c], n2);
        const float inv = 1.0f / fmaxf(sqrtf(wave_sum(n2)), kEpsNormalize);
        float* out = cand + ((e >> 6) * Kpad) * 64 + (e & 63);
        for (int c = lane; c < Kpad; c += 64) {
            const float v = c < d ? x[c] * inv : 0.f;
            sx[c] = v;
            out[(int64_t)c * 64] = v;
        }
    }
    __syncthreads();
    if (e < E) {
        for (int s = lane; s < krp; s += 64) {
            float a = 0.f, b = 0.f;
            if (s < kr) {
                for (int c = 0; c < d; ++c) { a = fmaf(sx[c], M1[(int64_t)c * kr + s], a); b = fmaf(sx[c], M2[(int64_t)c * kr + s], b); }
            }
            EM1[e * krp + s] = a; EM2[e * krp + s] = b;
        }
    }
}

// ---- per query: normalised h / t rows, r^, lin_q  (one wave per test triple of the chunk)
__global__ __launch_bounds__(256) void k_ntn_q_prep(const float* __restrict__ ent, const float* __restrict__ rel,
                                                    const float* __restrict__ M1, const float* __restrict__ M2,
                                                    const float* __restrict__ b, const int64_t* __restrict__ triples,
                                                    int64_t n, int d, int kr, NtnEvalWs w) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int64_t h = triples[3 * i], r = triples[3 * i + 1], t = triples[3 * i + 2];
    const float* eh = ent + h * d; const float* et = ent + t * d; const float* er = rel + r * kr;
    float nh = 0.f, nt = 0.f, nr = 0.f;
    for (int c = lane; c < d; c += 64) { nh = fmaf(eh[c], eh[c], nh); nt = fmaf(et[c], et[c], nt); }
    for (int c = lane; c < kr; c += 64) nr = fmaf(er[c], er[c], nr);
    const float ih = 1.0f / fmaxf(sqrtf(wave_sum(nh)), kEpsNormalize), it = 1.0f / fmaxf(sqrtf(wave_sum(nt)), kEpsNormalize);
    const float ir = 1.0f / fmaxf(sqrtf(wave_sum(nr)), kEpsNormalize);
    float* Hq = w.Hq + i * w.Kpad; float* Tq = w.Tq + i * w.Kpad;
    for (int c = lane; c < w.Kpad; c += 64) { Hq[c] = c < d ? eh[c] * ih : 0.f; Tq[c] = c < d ? et[c] * it : 0.f; }
    for (int s = lane; s < w.krp; s += 64) {
        float lt = 0.f, lh = 0.f, rn = 0.f;
        if (s < kr) {
            lt = b[s]; lh = b[s];
            for (int c = 0; c < d; ++c) {
                lt = fmaf(eh[c] * ih, M1[(int64_t)c * kr + s], lt);   // tail sweep: b + h^ M1
                lh = fmaf(et[c] * it, M2[(int64_t)c * kr + s], lh);   // head sweep: b + t^ M2
            }
            rn = er[s] * ir;
        }
        w.qlin[(2 * i) * w.krp + s] = lt; w.qlin[(2 * i + 1) * w.krp + s] = lh;
        w.qr[(2 * i) * w.krp + s] = rn; w.qr[(2 * i + 1) * w.krp + s] = rn;
    }
    if (lane == 0) { w.truth[2 * i] = t; w.truth[2 * i + 1] = h; }
}

// ---- per (32-query tile, slice s): QA[2q][s][:] = h^_q^T W_s ,  QA[2q+1][s][:] = W_s t^_q     (f32 MFMA)
__global__ __launch_bounds__(256) void k_ntn_q_contract(const float* __restrict__ W, int64_t n, int d, int kr, NtnEvalWs w) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int S = w.Kpad | 1;
    float* sH = smem; float* sT = sH + NQT * S;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lk = lane >> 5;
    const int64_t q0 = (int64_t)blockIdx.x * NQT;
    const int cnt = (int)min((int64_t)NQT, n - q0);
    for (int idx = threadIdx.x; idx < NQT * w.Kpad; idx += 256) {
        const int i = idx / w.Kpad, c = idx - i * w.Kpad;
        sH[i * S + c] = i < cnt ? w.Hq[(q0 + i) * w.Kpad + c] : 0.f;
        sT[i * S + c] = i < cnt ? w.Tq[(q0 + i) * w.Kpad + c] : 0.f;
    }
    __syncthreads();
    const int s = blockIdx.y * 4 + wave;
    if (s >= kr) return;
    const float* Ws = W + (int64_t)s * d * d;
    for (int c0 = 0; c0 < d; c0 += 32) {
        const int col = c0 + li;
        f32x16 accx = {0}, accy = {0};
        for (int k0 = 0; k0 < d; k0 += 8) {
            float ah[4], at[4], bx[4], by[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = k0 + 2 * u + lk;
                const bool ok = c < d && col < d;
                ah[u] = c < d ? sH[li * S + c] : 0.f;
                at[u] = c < d ? sT[li * S + c] : 0.f;
                bx[u] = ok ? Ws[(int64_t)c * d + col] : 0.f;   // X[q][col] = sum_c h^[c] W[c][col]
                by[u] = ok ? Ws[(int64_t)col * d + c] : 0.f;   // Y[q][col] = sum_c W[col][c] t^[c]
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                accx = __builtin_amdgcn_mfma_f32_32x32x2f32(ah[u], bx[u], accx, 0, 0, 0);
                accy = __builtin_amdgcn_mfma_f32_32x32x2f32(at[u], by[u], accy, 0, 0, 0);
            }
        }
        if (col < w.Kpad) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = (q & 3) + 8 * (q >> 2) + 4 * lk;
                if (row < cnt) {
                    const int64_t qi = q0 + row;
                    w.QA[((2 * qi) * w.krp + s) * w.Kpad + col] = col < d ? accx[q] : 0.f;
                    w.QA[((2 * qi + 1) * w.krp + s) * w.Kpad + col] = col < d ? accy[q] : 0.f;
                }
            }
        }
    }
}

// ---- the sweep: one workgroup per query side; A_q transposed into LDS once, then [32 candidates] x [k_r] MFMA tiles
__global__ __launch_bounds__(256) void k_ntn_sweep(int64_t nq, int64_t E, int d, int kr, NtnEvalWs w, int64_t row0,
                                                   float* __restrict__ scores) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int SQ = w.krp + 1;                 // padded row stride of the transposed query matrix
    float* sQ = smem;                         // [Kpad][krp+1]   sQ[c][s] = A_q[s][c]
    float* sLin = sQ + w.Kpad * SQ;           // [krp]
    float* sR = sLin + w.krp;                 // [krp]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lk = lane >> 5;
    const int64_t q = blockIdx.x;
    if (q >= nq) return;
    const int side = (int)(q & 1);            // 0 tail sweep, 1 head sweep
    const float* A = w.QA + q * (int64_t)w.krp * w.Kpad;
    for (int idx = threadIdx.x; idx < w.krp * w.Kpad; idx += 256) {
        const int s = idx / w.Kpad, c = idx - s * w.Kpad;
        sQ[c * SQ + s] = s < kr ? A[idx] : 0.f;
    }
    for (int s = threadIdx.x; s < w.krp; s += 256) { sLin[s] = w.qlin[q * w.krp + s]; sR[s] = w.qr[q * w.krp + s]; }
    __syncthreads();
    const float* elin = side == 0 ? w.EM2 : w.EM1;   // tail candidates: t^_e M2 ; head candidates: h^_e M1
    const int64_t ntile32 = (E + 31) / 32;
    float* out = scores + (row0 + q) * E;
    for (int64_t tl = (int64_t)blockIdx.y * 4 + wave; tl < ntile32; tl += 4 * (int64_t)gridDim.y) {  // tiles dealt over gridDim.y workgroups
        const int64_t e0 = tl * 32;
        const float* cbase = w.cand + ((e0 >> 6) * w.Kpad) * 64 + (e0 & 63) + li;   // + k*64
        float tot[16];
#pragma unroll
        for (int qq = 0; qq < 16; ++qq) tot[qq] = 0.f;
        for (int s0 = 0; s0 < w.krp; s0 += 32) {
            f32x16 acc = {0};
            for (int k0 = 0; k0 < w.Kpad; k0 += 8) {
                float av[4], bv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int c = k0 + 2 * u + lk;
                    av[u] = cbase[(int64_t)c * 64];          // A[i = candidate li][k = c]
                    bv[u] = sQ[c * SQ + s0 + li];            // B[k = c][j = slice s0+li]
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
            }
            const int s = s0 + li;
            const float lin = sLin[s], rn = sR[s];
#pragma unroll
            for (int qq = 0; qq < 16; ++qq) {
                const int64_t e = e0 + (qq & 3) + 8 * (qq >> 2) + 4 * lk;
                const float el = (e < E) ? elin[e * w.krp + s] : 0.f;
                tot[qq] = fmaf(rn, tanhf(acc[qq] + lin + el), tot[qq]);   // padded slices: rn = 0
            }
        }
#pragma unroll
        for (int qq = 0; qq < 16; ++qq) {   // sum over the 32 lanes that hold different slices of the same candidate
            float v = tot[qq];
            v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); v += dpp_mov<0x141>(v); v += dpp_mov<0x140>(v);
            v += swz_xor16(v);
            const int64_t e = e0 + (qq & 3) + 8 * (qq >> 2) + 4 * lk;
            if (li == 0 && e < E) out[e] = -v;
        }
    }
}

__global__ void k_ntn_rank_rows(const float* __restrict__ scores, int64_t n, int64_t E, const int64_t* __restrict__ truth,
                                const int64_t* __restrict__ tail_off, const int32_t* __restrict__ tail_ids,
                                const int64_t* __restrict__ head_off, const int32_t* __restrict__ head_ids, int64_t tri0,
                                int64_t n_total, int32_t* __restrict__ ranks) {
    // one wave per query side of the chunk: rows 2i (tail sweep of triple tri0+i) and 2i+1 (head sweep)
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= 2 * n) return;
    const int64_t i = q >> 1;
    const int side = (int)(q & 1);
    const float* s = scores + q * E;
    const int64_t tr = truth[q];
    const float st = s[tr];
    int cnt = 0, fc = 0;
    for (int64_t e = lane; e < E; e += 64) cnt += s[e] < st ? 1 : 0;
    const int64_t* off = side == 0 ? tail_off : head_off;
    const int32_t* ids = side == 0 ? tail_ids : head_ids;
    if (off) {
        for (int64_t j = off[tri0 + i] + lane; j < off[tri0 + i + 1]; j += 64) {
            const int64_t e = ids[j];
            fc += (e != tr && s[e] < st) ? 1 : 0;
        }
    }
    cnt = (int)wave_sum((float)cnt);
    fc = (int)wave_sum((float)fc);
    if (lane == 0) {
        ranks[(side == 0 ? 1 : 0) * n_total + tri0 + i] = cnt;        // rows: head, tail, fhead, ftail
        ranks[(side == 0 ? 3 : 2) * n_total + tri0 + i] = cnt - fc;
    }
}

static int ntn_eval_common(const kge_model_desc* m, const int64_t* triples, int64_t n, const int64_t* tail_off,
                           const int32_t* tail_ids, const int64_t* head_off, const int32_t* head_ids, void* ws,
                           size_t ws_bytes, int32_t* ranks, float* scores_out, hipStream_t s) {
    const int d = m->dim, kr = m->rel_dim;
    if (d > 256 || kr > 256) { set_error("NTN sweep: ent_hidden_size and rel_hidden_size <= 256 supported"); return -1; }
    NtnEvalWs w;
    ntn_eval_plan(m, n, ws, &w, false);
    if (!ws || ws_bytes < w.bytes) { set_error("kge_eval (NTN): workspace too small (%zu < %zu)", ws_bytes, w.bytes); return -1; }
    const int64_t E = m->tot_entity;
    hipLaunchKernelGGL(k_ntn_ent_tables, dim3((unsigned)((E + 3) / 4)), dim3(256), (size_t)4 * w.Kpad * sizeof(float), s,
                       m->tables[0], m->tables[2], m->tables[3], E, d, kr, w.Kpad, w.krp, w.cand, w.EM1, w.EM2);
    const size_t lds_c = (size_t)2 * NQT * (w.Kpad | 1) * sizeof(float);
    const size_t lds_s = ((size_t)w.Kpad * (w.krp + 1) + 2 * w.krp) * sizeof(float);
    if (lds_s > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)k_ntn_sweep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s);
    for (int64_t lo = 0; lo < n; lo += w.chunk) {
        const int64_t c = min((int64_t)w.chunk, n - lo);
        const int64_t* tri = triples + 3 * lo;
        hipLaunchKernelGGL(k_ntn_q_prep, dim3((unsigned)((c + 3) / 4)), dim3(256), 0, s, m->tables[0], m->tables[1],
                           m->tables[2], m->tables[3], m->tables[4], tri, c, d, kr, w);
        hipLaunchKernelGGL(k_ntn_q_contract, dim3((unsigned)((c + NQT - 1) / NQT), (unsigned)((kr + 3) / 4)), dim3(256), lds_c,
                           s, m->tables[5], c, d, kr, w);
        float* sc = scores_out ? scores_out + 2 * lo * E : w.scores;
        // few queries (one workgroup each) would leave most CUs idle and every SIMD with a single wave: split each
        // query's candidate tiles over several workgroups (~3 per CU: the transposed query matrix takes ~50 KB of LDS)
        int64_t split = (768 + 2 * c - 1) / (2 * c);
        if (split > 16) split = 16;
        if (split < 1) split = 1;
        hipLaunchKernelGGL(k_ntn_sweep, dim3((unsigned)(2 * c), (unsigned)split), dim3(256), lds_s, s, 2 * c, E, d, kr, w,
                           (int64_t)0, sc);
        if (ranks)
            hipLaunchKernelGGL(k_ntn_rank_rows, dim3((unsigned)((2 * c + 3) / 4)), dim3(256), 0, s, sc, c, E, w.truth, tail_off,
                               tail_ids, head_off, head_ids, lo, n, ranks);
    }
    return check_launch("NTN sweep");
}

int launch_ntn_eval_ranks(const kge_model_desc* m, const int64_t* triples, int64_t n, const int64_t* tail_off,
                          const int32_t* tail_ids, const int64_t* head_off, const int32_t* head_ids, void* ws,
                          size_t ws_bytes, int32_t* ranks, hipStream_t s) {
    return ntn_eval_common(m, triples, n, tail_off, tail_ids, head_off, head_ids, ws, ws_bytes, ranks, nullptr, s);
}

int launch_ntn_eval_scores(const kge_model_desc* m, const int64_t* triples, int64_t n, void* ws, size_t ws_bytes,
                           float* scores, hipStream_t s) {
    return ntn_eval_common(m, triples, n, nullptr, nullptr, nullptr, nullptr, ws, ws_bytes, nullptr, scores, s);
}

}  // namespace kge
