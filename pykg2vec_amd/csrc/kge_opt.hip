// kge_opt.hip -- dense optimiser sweeps with torch.optim default semantics (utils/trainer.py:112-131).
// nn.Embedding is dense (models/Domain.py:8-13): Adam / Adagrad / RMSprop touch EVERY row every step, so the
// sweep is a pure HBM stream: float4 per lane, grid-stride, (reads+writes) = 3 (SGD) .. 7 (Adam) floats per
// parameter -- less where the gradient is zero: no clearing write, and for SGD / Adagrad nothing but the gradient read.  The gradient buffer is cleared in the same pass (optimizer.zero_grad(), utils/trainer.py:272).
#include "kge_internal.h"
#include "kge_opt_device.h"

namespace kge {

// Optional tail duty of the optimiser launch in hipGraph-replayed steps: thread 0 of block 0 derives the NEXT step's
// device-resident state (batch cursor, Philox offset, optimiser step, Adam bias terms) from the current one.  It writes
// a different state set than the one this step's kernels read (two sets, alternating), so there is no ordering hazard
// and no separate one-thread launch per step.
struct AdvanceArgs {
    const int64_t* cin;   // {start, philox offset, opt step, next batch index, next draws offset} of THIS step (NULL: off)
    int64_t* cout;        // the same five fields for the next step
    float* hout;          // {lr, step_size, bc2_sqrt} of the next step
    int64_t batch_stride, n_batches, draws_per_batch;
    float lr;
};

__device__ __forceinline__ void advance_state(const AdvanceArgs& v) {
    const int64_t b = v.cin[3];
    v.cout[0] = b * v.batch_stride;
    v.cout[1] = v.cin[4];
    v.cout[4] = v.cin[4] + v.draws_per_batch;
    v.cout[3] = (b + 1) % v.n_batches;
    const int64_t t = v.cin[2] + 1;
    v.cout[2] = t;
    const double bc1 = 1.0 - pow(0.9, (double)t), bc2 = 1.0 - pow(0.999, (double)t);
    v.hout[0] = v.lr;
    v.hout[1] = (float)((double)v.lr / bc1);
    v.hout[2] = (float)sqrt(bc2);
}

template <int KIND, bool ZERO, bool NT>
__global__ __launch_bounds__(256) void k_opt(float* __restrict__ p, float* __restrict__ g, float* __restrict__ s1,
                                             float* __restrict__ s2, int64_t numel, OptArgs a,
                                             const float* __restrict__ dev_hyper, AdvanceArgs adv) {
    if (dev_hyper) { a.lr = dev_hyper[0]; a.step_size = dev_hyper[1]; a.bc2_sqrt = dev_hyper[2]; }  // hipGraph replays
    if (adv.cin != nullptr && blockIdx.x == 0 && threadIdx.x == 0) advance_state(adv);
    const int64_t nvec = numel / 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        const float4 gv = stream_load<NT>(reinterpret_cast<const float4*>(g) + i);
        const bool gzero = gv.x == 0.f && gv.y == 0.f && gv.z == 0.f && gv.w == 0.f;
        // SGD / Adagrad with g == 0 leave parameter and state unchanged (p - lr*0/.. and s + 0*0): rows the batch did not
        // touch -- almost all of a large table -- cost one gradient read instead of six streams.  Adam / RMSprop decay
        // their moments for every row every step (dense nn.Embedding gradients, models/Domain.py:8-13) and take the full path.
        if constexpr (KIND == KGE_OPT_SGD || KIND == KGE_OPT_ADAGRAD) { if (gzero) continue; }
        float4 pv = stream_load<NT>(reinterpret_cast<const float4*>(p) + i);
        float4 av = make_float4(0, 0, 0, 0), bv = make_float4(0, 0, 0, 0);
        if constexpr (KIND != KGE_OPT_SGD) av = stream_load<NT>(reinterpret_cast<const float4*>(s1) + i);
        if constexpr (KIND == KGE_OPT_ADAM) bv = stream_load<NT>(reinterpret_cast<const float4*>(s2) + i);
        opt_update<KIND>(pv.x, gv.x, av.x, bv.x, a);
        opt_update<KIND>(pv.y, gv.y, av.y, bv.y, a);
        opt_update<KIND>(pv.z, gv.z, av.z, bv.z, a);
        opt_update<KIND>(pv.w, gv.w, av.w, bv.w, a);
        reinterpret_cast<float4*>(p)[i] = pv;   // (the next step gathers parameter rows: plain store)
        if constexpr (KIND != KGE_OPT_SGD) stream_store<NT>(reinterpret_cast<float4*>(s1) + i, av);
        if constexpr (KIND == KGE_OPT_ADAM) stream_store<NT>(reinterpret_cast<float4*>(s2) + i, bv);
        if constexpr (ZERO) { if (!gzero) reinterpret_cast<float4*>(g)[i] = make_float4(0, 0, 0, 0); }   // already clear: no write
    }
    if (blockIdx.x == 0) {  // tail (numel % 4)
        const int64_t i = nvec * 4 + threadIdx.x;
        if (threadIdx.x < 4 && i < numel) {
            float pv = p[i], gv = g[i], av = 0.f, bv = 0.f;
            if constexpr (KIND != KGE_OPT_SGD) av = s1[i];
            if constexpr (KIND == KGE_OPT_ADAM) bv = s2[i];
            opt_update<KIND>(pv, gv, av, bv, a);
            p[i] = pv;
            if constexpr (KIND != KGE_OPT_SGD) s1[i] = av;
            if constexpr (KIND == KGE_OPT_ADAM) s2[i] = bv;
            if constexpr (ZERO) g[i] = 0.f;
        }
    }
}

template <int KIND>
static int launch_kind(float* p, float* g, float* s1, float* s2, int64_t numel, OptArgs a, int zero, const float* dh,
                       const AdvanceArgs& adv, hipStream_t s) {
    int64_t blocks = (numel / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 16) blocks = 256 * 16;
    // parameters + gradient + state beyond the 256 MB Infinity Cache: stream them non-temporally (kge_opt_device.h)
    const int streams = KIND == KGE_OPT_SGD ? 2 : KIND == KGE_OPT_ADAM ? 4 : 3;
    const int force = switch_value("OPT_NT");
    const bool nt = force >= 0 ? force == 1 : (int64_t)streams * numel * 4 > ((int64_t)256 << 20);
    if (zero && nt)
        hipLaunchKernelGGL((k_opt<KIND, true, true>), dim3((int)blocks), dim3(256), 0, s, p, g, s1, s2, numel, a, dh, adv);
    else if (zero)
        hipLaunchKernelGGL((k_opt<KIND, true, false>), dim3((int)blocks), dim3(256), 0, s, p, g, s1, s2, numel, a, dh, adv);
    else if (nt)
        hipLaunchKernelGGL((k_opt<KIND, false, true>), dim3((int)blocks), dim3(256), 0, s, p, g, s1, s2, numel, a, dh, adv);
    else
        hipLaunchKernelGGL((k_opt<KIND, false, false>), dim3((int)blocks), dim3(256), 0, s, p, g, s1, s2, numel, a, dh, adv);
    return check_launch("k_opt");
}

// ---- one wave per ROW: the dense optimiser on the row, then (NORM) the row renormalisation Rescal.embed would apply at the next
// forward (kge_dense.hip: k_row_normalize -- same element-to-lane map and summation order, so the stored row is bit-identical
// to optimiser sweep + normalisation pass).  Saves the normalisation's read + write of the whole entity table per step.
template <int KIND, int NCH, bool NORM>
__global__ __launch_bounds__(256) void k_opt_rows(float* __restrict__ p, float* __restrict__ g, float* __restrict__ s1,
                                                  float* __restrict__ s2, int64_t rows, int dim, OptArgs a,
                                                  const float* __restrict__ dev_hyper, int zero) {
    if (dev_hyper) { a.lr = dev_hyper[0]; a.step_size = dev_hyper[1]; a.bc2_sqrt = dev_hyper[2]; }
    const int lane = threadIdx.x & 63;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
        const int64_t base = row * dim;
        float pv[NCH], gv[NCH], av[NCH], bv[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int e = lane + 64 * c;
            const bool on = e < dim;
            pv[c] = on ? p[base + e] : 0.f;
            gv[c] = on ? g[base + e] : 0.f;
            av[c] = (KIND != KGE_OPT_SGD && on) ? s1[base + e] : 0.f;
            bv[c] = (KIND == KGE_OPT_ADAM && on) ? s2[base + e] : 0.f;
        }
        float n2 = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            opt_update<KIND>(pv[c], gv[c], av[c], bv[c], a);
            n2 = fmaf(pv[c], pv[c], n2);   // (lanes beyond the row hold p = g = state = 0: every optimiser leaves them at 0)
        }
        float nrm = 1.f;
        if constexpr (NORM) nrm = sqrtf(wave_sum(n2));
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int e = lane + 64 * c;
            if (e < dim) {
                p[base + e] = NORM ? pv[c] / nrm : pv[c];
                if constexpr (KIND != KGE_OPT_SGD) s1[base + e] = av[c];
                if constexpr (KIND == KGE_OPT_ADAM) s2[base + e] = bv[c];
                if (zero && gv[c] != 0.f) g[base + e] = 0.f;
            }
        }
    }
}

// the same for rows of float4s (dim % 4 == 0): a 32-lane group per row, NV float4 per lane, two rows per wave, 16-byte accesses,
// optimiser state streamed non-temporally when the tables exceed the Infinity Cache (as k_opt does)
// ---- the exact lazy form of the dense optimisers (LazyRows; include/kge_hip.h: kge_lazy_*).
// nn.Embedding gradients are dense (models/Domain.py:8-13), so torch.optim moves EVERY row every step: Adam and RMSprop keep decaying
// the moments of rows no batch touched, and Rescal.embed renormalises every row at every forward (pairwise.py:843-844).  But a row
// whose gradient is zero for the steps L+1 .. t-1 has a trajectory that depends on nothing else: m <- m + 0.1 (0 - m),
// v <- 0.999 v, p <- p - step_size_i m / (sqrt(v) / bc2_i + eps), [p <- p / ||p||].  `last[row]` records the last step applied to
// the row; the sweep of step t skips rows without a gradient, and whoever needs a row -- the forward of a batch that contains it,
// or anyone who observes the tables -- first REPLAYS the missed steps in registers with the very fp32 operations of the sweep
// (opt_update<KIND> with g = +0, the same element-to-lane map and summation order of the row norm).  Bit-identical to the dense
// sweep by construction; bias-correction terms of every step come from one host-built table (kge_lazy_hyper_fill = make_opt_args).
struct LazyRows {
    int* last;                 // [rows] last optimiser step applied (0 = the initial tables); NULL: dense sweep
    const float2* hyper;       // [cap] {step_size, bc2_sqrt} of step index i
    int64_t cap;
    const int64_t* dev_cursor; // hipGraph replays: dev_cursor[2] = this step's index; NULL: `step`
    int64_t step;
};
__device__ __forceinline__ int64_t lazy_step_of(const LazyRows& z) { return z.dev_cursor ? z.dev_cursor[2] : z.step; }

// steps from .. to (inclusive) with zero gradient on the row held in registers; norm_last: also renormalise after step `to`
template <int KIND, int NV, bool NORM>
__device__ __forceinline__ void lazy_replay(float4 (&pv)[NV], float4 (&av)[NV], float4 (&bv)[NV], int64_t from, int64_t to, bool norm_last,
                                            const float2* __restrict__ hyper, float lr, int nvec, int gl) {
    // With a zero gradient only Adam can move p, and only through a non-zero first moment (p += -step_size * m / denom; SGD,
    // Adagrad and RMSprop compute p - lr * 0 / .. = p).  Where p is a fixed point of the optimiser it changes only through the
    // renormalisation, and once one renormalisation leaves every bit in place all later ones do: the replay of p stops there.  What
    // may remain is the decay of the optimiser state (Adam's second moment after the first has underflowed, RMSprop's
    // accumulator), replayed without p; rows that were never touched (all state zero -- most of a big table) have nothing left.
    float m_any = 0.f, s_any = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const float am = (av[v].x != 0.f || av[v].y != 0.f || av[v].z != 0.f || av[v].w != 0.f) ? 1.f : 0.f;
        const float bm = (bv[v].x != 0.f || bv[v].y != 0.f || bv[v].z != 0.f || bv[v].w != 0.f) ? 1.f : 0.f;
        m_any += am;
        s_any += am + bm;
    }
    const bool p_moves = KIND == KGE_OPT_ADAM && gsum<32>(m_any) > 0.f;
    const bool state_decays = (KIND == KGE_OPT_ADAM || KIND == KGE_OPT_RMSPROP) && gsum<32>(s_any) > 0.f;
    OptArgs a;
    a.lr = lr; a.step_size = 0.f; a.bc2_sqrt = 1.f;
    int64_t t = from;
    for (; t <= to; ++t) {
        if constexpr (KIND == KGE_OPT_ADAM) { const float2 h = hyper[t]; a.step_size = h.x; a.bc2_sqrt = h.y; }
        float n2 = 0.f, changed = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            opt_update<KIND>(pv[v].x, 0.f, av[v].x, bv[v].x, a); opt_update<KIND>(pv[v].y, 0.f, av[v].y, bv[v].y, a);
            opt_update<KIND>(pv[v].z, 0.f, av[v].z, bv[v].z, a); opt_update<KIND>(pv[v].w, 0.f, av[v].w, bv[v].w, a);
            n2 = fmaf(pv[v].x, pv[v].x, n2); n2 = fmaf(pv[v].y, pv[v].y, n2); n2 = fmaf(pv[v].z, pv[v].z, n2); n2 = fmaf(pv[v].w, pv[v].w, n2);
        }
        if constexpr (NORM) {
            if (t < to || norm_last) {
                const float nrm = sqrtf(gsum<32>(n2));
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    if (v * 32 + gl < nvec) {
                        const float4 before = pv[v];
                        pv[v].x = pv[v].x / nrm; pv[v].y = pv[v].y / nrm; pv[v].z = pv[v].z / nrm; pv[v].w = pv[v].w / nrm;
                        changed += (before.x != pv[v].x || before.y != pv[v].y || before.z != pv[v].z || before.w != pv[v].w) ? 1.f : 0.f;
                    }
                }
            }
        }
        if (!p_moves && gsum<32>(changed) == 0.f) { ++t; break; }   // p has reached its fixed point (without NORM: at once)
    }
    if (state_decays) {
        for (; t <= to; ++t) {   // the remaining steps touch the optimiser state only (same opt_update, p discarded)
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
                opt_update<KIND>(d.x, 0.f, av[v].x, bv[v].x, a); opt_update<KIND>(d.y, 0.f, av[v].y, bv[v].y, a);
                opt_update<KIND>(d.z, 0.f, av[v].z, bv[v].z, a); opt_update<KIND>(d.w, 0.f, av[v].w, bv[v].w, a);
            }
        }
    }
}

template <int KIND, int NV, bool NORM, bool NT>
__global__ __launch_bounds__(256) void k_opt_rows4(float* __restrict__ p, float* __restrict__ g, float* __restrict__ s1,
                                                   float* __restrict__ s2, int64_t rows, int dim, OptArgs a,
                                                   const float* __restrict__ dev_hyper, int zero,
                                                   const unsigned* __restrict__ touched, unsigned* __restrict__ touched_clear, LazyRows lazy) {
    if (dev_hyper) { a.lr = dev_hyper[0]; a.step_size = dev_hyper[1]; a.bc2_sqrt = dev_hyper[2]; }
    int64_t t_now = 0;
    if (lazy.last) {   // lazy form: the step's bias terms come from the same table the replays read
        t_now = lazy_step_of(lazy);
        const float2 h = lazy.hyper[t_now];
        a.step_size = h.x; a.bc2_sqrt = h.y;
    }
    const int gl = threadIdx.x & 31;
    const int nvec = dim >> 2;
    for (int64_t row = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5); row < rows; row += (int64_t)gridDim.x * 8) {
        // touched: one bit per row, set by the step that wrote a gradient into it; a clear bit means the row of `g` is zero
        // and is not read (one of the seven streams of a dense Adam sweep).  touched_clear: the OTHER step parity's bitmap, reset here.
        const bool has_g = !touched || ((touched[row >> 5] >> (row & 31)) & 1u);
        if (touched_clear && gl == 0 && (row & 31) == 0) touched_clear[row >> 5] = 0u;
        if (lazy.last) {
            // lazy: only rows with a gradient are stepped now -- and only if they are current through step t - 1 (the batch's rows were
            // caught up by kge_lazy_catchup before the forward; a stale bit of an earlier step with the same parity marks a row that
            // is NOT in this batch: its gradient is zero and it simply stays behind)
            if (!has_g || lazy.last[row] != (int)(t_now - 1)) continue;
        }
        float4* pr = reinterpret_cast<float4*>(p + row * dim);
        float4* gr = reinterpret_cast<float4*>(g + row * dim);
        float4* ar = reinterpret_cast<float4*>(s1 + row * dim);
        float4* br = reinterpret_cast<float4*>(s2 + row * dim);
        float4 pv[NV], gv[NV], av[NV], bv[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int i = v * 32 + gl;
            const bool on = i < nvec;
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            pv[v] = on ? pr[i] : z;
            gv[v] = (on && has_g) ? stream_load<NT>(gr + i) : z;
            av[v] = (KIND != KGE_OPT_SGD && on) ? stream_load<NT>(ar + i) : z;
            bv[v] = (KIND == KGE_OPT_ADAM && on) ? stream_load<NT>(br + i) : z;
        }
        float n2 = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            opt_update<KIND>(pv[v].x, gv[v].x, av[v].x, bv[v].x, a); opt_update<KIND>(pv[v].y, gv[v].y, av[v].y, bv[v].y, a);
            opt_update<KIND>(pv[v].z, gv[v].z, av[v].z, bv[v].z, a); opt_update<KIND>(pv[v].w, gv[v].w, av[v].w, bv[v].w, a);
            n2 = fmaf(pv[v].x, pv[v].x, n2); n2 = fmaf(pv[v].y, pv[v].y, n2); n2 = fmaf(pv[v].z, pv[v].z, n2); n2 = fmaf(pv[v].w, pv[v].w, n2);
        }
        float nrm = 1.f;
        if constexpr (NORM) nrm = sqrtf(gsum<32>(n2));
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int i = v * 32 + gl;
            if (i < nvec) {
                if constexpr (NORM) { pv[v].x = pv[v].x / nrm; pv[v].y = pv[v].y / nrm; pv[v].z = pv[v].z / nrm; pv[v].w = pv[v].w / nrm; }
                pr[i] = pv[v];   // (the next step gathers parameter rows: plain store)
                if constexpr (KIND != KGE_OPT_SGD) stream_store<NT>(ar + i, av[v]);
                if constexpr (KIND == KGE_OPT_ADAM) stream_store<NT>(br + i, bv[v]);
                if (zero && (gv[v].x != 0.f || gv[v].y != 0.f || gv[v].z != 0.f || gv[v].w != 0.f)) gr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if (lazy.last && gl == 0) lazy.last[row] = (int)t_now;
    }
}

// bring rows up to step `target`: MODE 0 = the rows named by up to four id lists (a batch's heads / tails / corrupted heads / tails:
// a row named several times is replayed by whoever raises last[row] first), target = this step - 1, every replayed step renormalised;
// MODE 1 = every row (before the tables are observed), target given, the final step renormalised only if norm_last
template <int KIND, int NV, bool NORM, int MODE>
__global__ __launch_bounds__(256) void k_lazy_rows(float* __restrict__ p, float* __restrict__ s1, float* __restrict__ s2, int64_t rows, int dim,
                                                   float lr, LazyRows lazy, const int64_t* __restrict__ ids0, const int64_t* __restrict__ ids1,
                                                   const int64_t* __restrict__ ids2, const int64_t* __restrict__ ids3, int64_t n_ids, int n_lists,
                                                   int norm_last) {
    const int gl = threadIdx.x & 31;
    const int nvec = dim >> 2;
    const int64_t target = MODE == 0 ? lazy_step_of(lazy) - 1 : lazy.step;
    const int64_t n_work = MODE == 0 ? n_ids * n_lists : rows;
    for (int64_t w = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5); w < n_work; w += (int64_t)gridDim.x * 8) {
        int64_t row = w;
        int old;
        if constexpr (MODE == 0) {
            const int list = (int)(w / n_ids);
            const int64_t i = w - (int64_t)list * n_ids;
            row = (list == 0 ? ids0 : list == 1 ? ids1 : list == 2 ? ids2 : ids3)[i];
            int claimed = 0;
            if (gl == 0) claimed = atomicMax(lazy.last + row, (int)target);
            old = __shfl(claimed, threadIdx.x & 32, 64);
        } else {
            old = lazy.last[row];
        }
        if (old >= (int)target) continue;
        float4* pr = reinterpret_cast<float4*>(p + row * dim);
        float4* ar = reinterpret_cast<float4*>(s1 + row * dim);
        float4* br = reinterpret_cast<float4*>(s2 + row * dim);
        float4 pv[NV], av[NV], bv[NV];
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int i = v * 32 + gl;
            const bool on = i < nvec;
            pv[v] = on ? pr[i] : z;
            av[v] = (KIND != KGE_OPT_SGD && on) ? ar[i] : z;
            bv[v] = (KIND == KGE_OPT_ADAM && on) ? br[i] : z;
        }
        lazy_replay<KIND, NV, NORM>(pv, av, bv, (int64_t)old + 1, target, MODE == 0 ? true : norm_last != 0, lazy.hyper, lr, nvec, gl);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int i = v * 32 + gl;
            if (i < nvec) {
                pr[i] = pv[v];
                if constexpr (KIND != KGE_OPT_SGD) ar[i] = av[v];
                if constexpr (KIND == KGE_OPT_ADAM) br[i] = bv[v];
            }
        }
        if constexpr (MODE == 1) { if (gl == 0) lazy.last[row] = (int)target; }
    }
}

static bool rows4_ok(const float* p, const float* g, const float* s1, const float* s2, int dim) {
    return (dim & 3) == 0 && dim <= 1024 && ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)s1 | (uintptr_t)s2) & 15) == 0);
}

template <int KIND>
static int launch_rows_kind(float* p, float* g, float* s1, float* s2, int64_t rows, int dim, OptArgs a, int zero, int normalize,
                            const float* dh, const unsigned* touched, unsigned* tclear, const LazyRows& lazy, hipStream_t s) {
    if (rows4_ok(p, g, s1, s2, dim)) {
        int64_t blocks4 = (rows + 7) / 8;
        if (blocks4 > 256 * 32) blocks4 = 256 * 32;
        const int streams = KIND == KGE_OPT_SGD ? 2 : KIND == KGE_OPT_ADAM ? 4 : 3;
        const bool nt = !lazy.last && (int64_t)streams * rows * dim * 4 > ((int64_t)256 << 20);   // (a lazy sweep streams nothing)
#define KGE_ROWS4(NV_)                                                                                                    \
        if (dim <= 128 * NV_) {                                                                                            \
            if (normalize && nt) hipLaunchKernelGGL((k_opt_rows4<KIND, NV_, true, true>), dim3((int)blocks4), dim3(256), 0, s, p, g, s1, s2, rows, dim, a, dh, zero, touched, tclear, lazy); \
            else if (normalize) hipLaunchKernelGGL((k_opt_rows4<KIND, NV_, true, false>), dim3((int)blocks4), dim3(256), 0, s, p, g, s1, s2, rows, dim, a, dh, zero, touched, tclear, lazy); \
            else if (nt) hipLaunchKernelGGL((k_opt_rows4<KIND, NV_, false, true>), dim3((int)blocks4), dim3(256), 0, s, p, g, s1, s2, rows, dim, a, dh, zero, touched, tclear, lazy); \
            else hipLaunchKernelGGL((k_opt_rows4<KIND, NV_, false, false>), dim3((int)blocks4), dim3(256), 0, s, p, g, s1, s2, rows, dim, a, dh, zero, touched, tclear, lazy); \
            return check_launch("k_opt_rows4");                                                                            \
        }
        KGE_ROWS4(1) KGE_ROWS4(2) KGE_ROWS4(4) KGE_ROWS4(8)
#undef KGE_ROWS4
    }
    if (lazy.last) { set_error("kge_optimizer_step_rows: the lazy form needs rows of float4s (dim %% 4 == 0, 16-byte aligned buffers, dim <= 1024)"); return -1; }
    if (tclear) {   // (the dword kernel reads every gradient row: a superset of the touched ones)
        hipError_t e = hipMemsetAsync(tclear, 0, (size_t)((rows + 31) / 32) * sizeof(unsigned), s);
        if (e != hipSuccess) { set_error("optimizer rows: memset: %s", hipGetErrorString(e)); return -2; }
    }
    int64_t blocks = (rows + 3) / 4;
    if (blocks > 256 * 32) blocks = 256 * 32;
#define KGE_ROWS(NCH_)                                                                                                   \
    if (dim <= 64 * NCH_) {                                                                                               \
        if (normalize) hipLaunchKernelGGL((k_opt_rows<KIND, NCH_, true>), dim3((int)blocks), dim3(256), 0, s, p, g, s1, s2, rows, dim, a, dh, zero); \
        else hipLaunchKernelGGL((k_opt_rows<KIND, NCH_, false>), dim3((int)blocks), dim3(256), 0, s, p, g, s1, s2, rows, dim, a, dh, zero);        \
        return check_launch("k_opt_rows");                                                                                \
    }
    KGE_ROWS(4) KGE_ROWS(8) KGE_ROWS(16)
#undef KGE_ROWS
    return -1;
}

static LazyRows to_lazy(const kge_lazy_rows* z, int64_t step) {
    LazyRows l;
    l.last = z ? z->last : nullptr;
    l.hyper = z ? reinterpret_cast<const float2*>(z->hyper) : nullptr;
    l.cap = z ? z->hyper_cap : 0;
    l.dev_cursor = z ? z->dev_cursor : nullptr;
    l.step = step;
    return l;
}

static int lazy_check(const kge_lazy_rows* z, int64_t step, const char* who) {
    if (!z) return 0;
    if (!z->last || !z->hyper || z->hyper_cap < 2) { set_error("%s: lazy rows need `last` and the hyper table", who); return -1; }
    if (!z->dev_cursor && (step < 1 || step >= z->hyper_cap)) {
        set_error("%s: step %lld outside the hyper table (%lld entries; kge_lazy_hyper_fill)", who, (long long)step, (long long)z->hyper_cap);
        return -1;
    }
    return 0;
}

int launch_optimizer_rows(int kind, float* p, float* g, float* s1, float* s2, int64_t rows, int dim, float lr, int64_t step,
                          int zero_grad, int normalize, const float* dev_hyper, const unsigned* touched, unsigned* touched_clear,
                          const kge_lazy_rows* lazy_in, hipStream_t s) {
    const OptArgs a = make_opt_args(lr, step);
    if (lazy_check(lazy_in, step, "kge_optimizer_step_rows")) return -1;
    if (lazy_in && !touched) { set_error("kge_optimizer_step_rows: the lazy form steps the rows of the touched-row bitmap: it is required"); return -1; }
    const LazyRows lazy = to_lazy(lazy_in, step);
    switch (kind) {
        case KGE_OPT_SGD: return launch_rows_kind<KGE_OPT_SGD>(p, g, s1, s2, rows, dim, a, zero_grad, normalize, dev_hyper, touched, touched_clear, lazy, s);
        case KGE_OPT_ADAM:
            if (!s1 || !s2) { set_error("adam needs two state buffers"); return -1; }
            return launch_rows_kind<KGE_OPT_ADAM>(p, g, s1, s2, rows, dim, a, zero_grad, normalize, dev_hyper, touched, touched_clear, lazy, s);
        case KGE_OPT_ADAGRAD:
            if (!s1) { set_error("adagrad needs a state buffer"); return -1; }
            return launch_rows_kind<KGE_OPT_ADAGRAD>(p, g, s1, s2, rows, dim, a, zero_grad, normalize, dev_hyper, touched, touched_clear, lazy, s);
        case KGE_OPT_RMSPROP:
            if (!s1) { set_error("rmsprop needs a state buffer"); return -1; }
            return launch_rows_kind<KGE_OPT_RMSPROP>(p, g, s1, s2, rows, dim, a, zero_grad, normalize, dev_hyper, touched, touched_clear, lazy, s);
    }
    set_error("kge_optimizer_step_rows: unknown optimizer %d", kind);
    return -1;
}

// MODE 0: catch the rows named by the id lists up to (this step - 1); MODE 1: every row up to `step` (a flush)
template <int KIND, int MODE>
static int launch_lazy_kind(float* p, float* s1, float* s2, int64_t rows, int dim, float lr, int normalize, const LazyRows& lazy,
                            const int64_t* const* ids, int64_t n_ids, int n_lists, int norm_last, hipStream_t s) {
    const int64_t work = MODE == 0 ? n_ids * n_lists : rows;
    int64_t blocks = (work + 7) / 8;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 64) blocks = 256 * 64;
    const int64_t* i0 = n_lists > 0 ? ids[0] : nullptr; const int64_t* i1 = n_lists > 1 ? ids[1] : nullptr;
    const int64_t* i2 = n_lists > 2 ? ids[2] : nullptr; const int64_t* i3 = n_lists > 3 ? ids[3] : nullptr;
#define KGE_LZ(NV_)                                                                                                              \
    if (dim <= 128 * NV_) {                                                                                                      \
        if (normalize) hipLaunchKernelGGL((k_lazy_rows<KIND, NV_, true, MODE>), dim3((int)blocks), dim3(256), 0, s, p, s1, s2, rows, dim, lr, lazy, i0, i1, i2, i3, n_ids, n_lists, norm_last); \
        else hipLaunchKernelGGL((k_lazy_rows<KIND, NV_, false, MODE>), dim3((int)blocks), dim3(256), 0, s, p, s1, s2, rows, dim, lr, lazy, i0, i1, i2, i3, n_ids, n_lists, norm_last); \
        return check_launch("k_lazy_rows");                                                                                      \
    }
    KGE_LZ(1) KGE_LZ(2) KGE_LZ(4) KGE_LZ(8)
#undef KGE_LZ
    return -1;
}

int launch_lazy_rows(int kind, int mode, float* p, float* s1, float* s2, int64_t rows, int dim, float lr, int normalize,
                     const kge_lazy_rows* lazy_in, int64_t step, const int64_t* const* ids, int64_t n_ids, int n_lists, int norm_last,
                     hipStream_t s) {
    const char* who = mode == 0 ? "kge_lazy_catchup" : "kge_lazy_flush";
    if (!lazy_in) { set_error("%s: no lazy state", who); return -1; }
    if (lazy_check(lazy_in, step, who)) return -1;
    if (!rows4_ok(p, p, s1, s2, dim)) { set_error("%s: rows of float4s only (dim %% 4 == 0, dim <= 1024, 16-byte aligned)", who); return -1; }
    if ((kind != KGE_OPT_SGD && !s1) || (kind == KGE_OPT_ADAM && !s2)) { set_error("%s: optimiser state missing", who); return -1; }
    const LazyRows lazy = to_lazy(lazy_in, step);
#define KGE_LK(K_) case K_: return mode == 0 ? launch_lazy_kind<K_, 0>(p, s1, s2, rows, dim, lr, normalize, lazy, ids, n_ids, n_lists, norm_last, s) \
                                              : launch_lazy_kind<K_, 1>(p, s1, s2, rows, dim, lr, normalize, lazy, ids, n_ids, n_lists, norm_last, s);
    switch (kind) { KGE_LK(KGE_OPT_SGD) KGE_LK(KGE_OPT_ADAM) KGE_LK(KGE_OPT_ADAGRAD) KGE_LK(KGE_OPT_RMSPROP) }
#undef KGE_LK
    set_error("%s: unknown optimizer %d", who, kind);
    return -1;
}

void lazy_hyper_fill(float lr, int64_t first_step, int64_t n, float* out) {
    for (int64_t i = 0; i < n; ++i) {
        const OptArgs a = make_opt_args(lr, first_step + i < 1 ? 1 : first_step + i);
        out[2 * i] = a.step_size;
        out[2 * i + 1] = a.bc2_sqrt;
    }
}

int launch_optimizer(int kind, float* p, float* g, float* s1, float* s2, int64_t numel, float lr, int64_t step,
                     int zero_grad, const float* dev_hyper, const int64_t* cursor_in, int64_t* cursor_out, float* hyper_out,
                     int64_t batch_stride, int64_t n_batches, int64_t draws_per_batch, hipStream_t s) {
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)s1 | (uintptr_t)s2) & 15) {
        set_error("kge_optimizer_step: buffers must be 16-byte aligned");
        return -1;
    }
    AdvanceArgs adv;
    adv.cin = cursor_in; adv.cout = cursor_out; adv.hout = hyper_out;
    adv.batch_stride = batch_stride; adv.n_batches = n_batches > 0 ? n_batches : 1; adv.draws_per_batch = draws_per_batch;
    adv.lr = lr;
    const OptArgs a = make_opt_args(lr, step);
    switch (kind) {
        case KGE_OPT_SGD: return launch_kind<KGE_OPT_SGD>(p, g, s1, s2, numel, a, zero_grad, dev_hyper, adv, s);
        case KGE_OPT_ADAM:
            if (!s1 || !s2) { set_error("adam needs two state buffers"); return -1; }
            return launch_kind<KGE_OPT_ADAM>(p, g, s1, s2, numel, a, zero_grad, dev_hyper, adv, s);
        case KGE_OPT_ADAGRAD:
            if (!s1) { set_error("adagrad needs a state buffer"); return -1; }
            return launch_kind<KGE_OPT_ADAGRAD>(p, g, s1, s2, numel, a, zero_grad, dev_hyper, adv, s);
        case KGE_OPT_RMSPROP:
            if (!s1) { set_error("rmsprop needs a state buffer"); return -1; }
            return launch_kind<KGE_OPT_RMSPROP>(p, g, s1, s2, numel, a, zero_grad, dev_hyper, adv, s);
    }
    set_error("kge_optimizer_step: unknown optimizer %d", kind);
    return -1;
}

}  // namespace kge
