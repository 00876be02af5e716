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
// Staged gradients (kge_rescal_stage, include/kge_hip.h): the gradient of a touched row is not a row of `g` but the sum of the
// rows of `gstage` registered with it (count / bucket / overflow chain, filled by the grouping launch of the pairwise step), added
// in ascending slot order -- a fixed order, whatever order the registrations arrived in -- and skipped where the slot's pair has a
// zero hinge coefficient.  The owner resets the row's list: the lists are all-zero again when the sweep ends.
struct RowStage {
    const float* gstage; const float* dsv;
    int* count; const int* bucket; int* head; const int* next; int cap;
};

constexpr int kOptNormChunk = 4096;     // = kNormChunk of kge_dense.hip (the separate normalisation pass)
// one workgroup = one chunk of one wide row: every stream of the chunk requested before the first update, the optimiser, the chunk's sum
// of squares (per-thread fmaf chain in element order, wave butterflies, the four waves as (0 + 1) + (2 + 3)) to part[blk]
template <int KIND>
__device__ __forceinline__ void opt_sumsq_chunk(int blk, float* __restrict__ p, float* __restrict__ g, float* __restrict__ s1,
                                                float* __restrict__ s2, int64_t dim, int nchunk, const OptArgs& a, int zero,
                                                float* __restrict__ part, float* sw) {
    const int64_t row = blk / nchunk;
    const int ch = blk % nchunk;
    const int64_t base = row * dim;
    const int64_t lo = (int64_t)ch * kOptNormChunk, hi = min(dim, lo + kOptNormChunk);
    constexpr int PER = kOptNormChunk / 256, SUB = 4;   // SUB elements of every stream in flight per thread (registers: the rider form
                                                        // must not cost the row owners of k_opt_rows4 their occupancy)
    float n2 = 0.f;
#pragma unroll
    for (int u0 = 0; u0 < PER; u0 += SUB) {
        float pv[SUB], gv[SUB], av[SUB], bv[SUB];
#pragma unroll
        for (int u = 0; u < SUB; ++u) {      // (clamped addresses)
            const int64_t c = base + min(lo + threadIdx.x + 256 * (u0 + u), hi - 1);
            pv[u] = p[c]; gv[u] = g[c];
            av[u] = KIND != KGE_OPT_SGD ? s1[c] : 0.f;
            bv[u] = KIND == KGE_OPT_ADAM ? s2[c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < SUB; ++u) {
            const int64_t cc = lo + threadIdx.x + 256 * (u0 + u);
            if (cc < hi) {
                const int64_t c = base + cc;
                opt_update<KIND>(pv[u], gv[u], av[u], bv[u], a);
                p[c] = pv[u];
                if constexpr (KIND != KGE_OPT_SGD) s1[c] = av[u];
                if constexpr (KIND == KGE_OPT_ADAM) s2[c] = bv[u];
                if (zero && gv[u] != 0.f) g[c] = 0.f;
                n2 = fmaf(pv[u], pv[u], n2);
            }
        }
    }
    n2 = wave_sum(n2);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = n2;
    __syncthreads();
    if (threadIdx.x == 0) part[blk] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}

// The wide-row table's optimiser as a RIDER of the row-owner sweep below (round 6): the first `nblocks` workgroups of k_opt_rows4<...,
// RIDER> each take one chunk (opt_sumsq_chunk: the arithmetic of k_opt_sumsq_chunks, bit for bit) instead of rows, so RESCAL's step has
// one optimiser launch + the rescale launch instead of three launches -- the 11.6 us of the launch-bound chunk kernel disappear under
// the 100 us sweep of the entity table.
struct RelRider {
    float* p; float* g; float* s1; float* s2;
    int64_t dim; int nchunk; int nblocks; int zero;
    float* part;
    AdvanceArgs adv;
};

constexpr int kStageChainLds = 256;   // overflow-chain entries of one row kept in LDS (8 row groups per workgroup: 8 KB)
template <int KIND, int NV, bool NORM, bool NT, bool RIDER = false>
__global__ __launch_bounds__(256) void k_opt_rows4(float* __restrict__ p, float* __restrict__ g, float* __restrict__ s1,
                                                   float* __restrict__ s2, int64_t rows, int dim, OptArgs a,
                                                   const float* __restrict__ dev_hyper, int zero,
                                                   const unsigned* __restrict__ touched, unsigned* __restrict__ touched_clear, RowStage stage,
                                                   RelRider rider) {
    __shared__ int s_chain[8][kStageChainLds];
    if (dev_hyper) { a.lr = dev_hyper[0]; a.step_size = dev_hyper[1]; a.bc2_sqrt = dev_hyper[2]; }
    int first = 0;      // workgroups in front of the row owners
    if constexpr (RIDER) {
        first = rider.nblocks;
        if ((int)blockIdx.x < first) {     // (workgroup-uniform) a chunk of the wide-row table instead of rows
            if (rider.adv.cin != nullptr && blockIdx.x == 0 && threadIdx.x == 0) advance_state(rider.adv);
            opt_sumsq_chunk<KIND>((int)blockIdx.x, rider.p, rider.g, rider.s1, rider.s2, rider.dim, rider.nchunk, a, rider.zero, rider.part,
                                  reinterpret_cast<float*>(&s_chain[0][0]));
            return;
        }
    }
    const int gl = threadIdx.x & 31;
    const int nvec = dim >> 2;
    for (int64_t row = (int64_t)((int)blockIdx.x - first) * 8 + (threadIdx.x >> 5); row < rows; row += (int64_t)((int)gridDim.x - first) * 8) {
        // touched: one bit per row, set by the step that wrote a gradient into it; a clear bit means the row of `g` is zero
        // and is not read (one of the seven streams of a dense Adam sweep).  touched_clear: the OTHER step parity's bitmap, reset here.
        const bool has_g = !touched || ((touched[row >> 5] >> (row & 31)) & 1u);
        if (touched_clear && gl == 0 && (row & 31) == 0) touched_clear[row >> 5] = 0u;
        float4* pr = reinterpret_cast<float4*>(p + row * dim);
        float4* gr = reinterpret_cast<float4*>(g + row * dim);
        float4* ar = reinterpret_cast<float4*>(s1 + row * dim);
        float4* br = reinterpret_cast<float4*>(s2 + row * dim);
        // every stream of the row requested before anything is used: the addresses are CLAMPED into the row and a lane beyond it drops
        // what it fetched (a select between two addresses -- `on ? pr[i] : z` -- made the compiler keep a zero in scratch and fetch
        // through 8-byte FLAT loads behind a wait per stream: round 6, profiles/r06_experiments.md section 7)
        float4 pv[NV], gv[NV], av[NV], bv[NV];
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool read_g = has_g && !stage.gstage;      // (uniform over the row's 32 lanes)
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int ic = min(v * 32 + gl, nvec - 1);
            pv[v] = pr[ic];
            if constexpr (KIND != KGE_OPT_SGD) av[v] = stream_load<NT>(ar + ic); else av[v] = z;
            if constexpr (KIND == KGE_OPT_ADAM) bv[v] = stream_load<NT>(br + ic); else bv[v] = z;
            gv[v] = z;
        }
        if (read_g) {
#pragma unroll
            for (int v = 0; v < NV; ++v) gv[v] = stream_load<NT>(gr + min(v * 32 + gl, nvec - 1));
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (v * 32 + gl >= nvec) { pv[v] = z; gv[v] = z; av[v] = z; bv[v] = z; }
        }
        if (stage.gstage && has_g) {
            const int cnt = stage.count[row];
            if (cnt > 0) {
                const int nb = cnt < stage.cap ? cnt : stage.cap;
                const int mine = gl < nb ? stage.bucket[row * stage.cap + gl] : 0x7FFFFFFF;   // (cap <= 32: one bucket entry per lane)
                const int chain = cnt > stage.cap ? stage.head[row] - 1 : -1;
                // The overflow chain (registrations beyond the bucket: hub entities of real graphs -- tens per 1 024-pair batch on
                // YAGO3-10 / FB15k) is walked ONCE into LDS; rounds 5's selection re-walked it -- dependent global loads -- for every
                // one of the cnt slots (O(cnt * chain) round trips, milliseconds for one hub row).  Entries beyond the
                // LDS window (kStageChainLds per row group) keep the walk: correct for any length, fast for every realistic one.
                int* const sc = s_chain[threadIdx.x >> 5];
                int nlds = 0, rest = -1;
                for (int j = chain; j >= 0; j = stage.next[j]) {     // (group-uniform: every lane walks, lane 0 files)
                    if (nlds == kStageChainLds) { rest = j; break; }
                    if (gl == 0) sc[nlds] = j;
                    ++nlds;
                }
                __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): lane 0's LDS writes have landed (same wave reads them below)
                __builtin_amdgcn_wave_barrier();
                int last = -1;
                for (int it = 0; it < cnt; ++it) {      // ascending slot order: the smallest slot above the last one taken
                    int best = mine > last ? mine : 0x7FFFFFFF;
                    for (int k = gl; k < nlds; k += 32) { const int j = sc[k]; if (j > last && j < best) best = j; }
#pragma unroll
                    for (int o = 16; o >= 1; o >>= 1) best = min(best, __shfl_xor(best, o, 64));   // over the row's 32 lanes
                    for (int j = rest; j >= 0; j = stage.next[j])
                        if (j > last && j < best) best = j;
                    if (best == 0x7FFFFFFF) break;
                    last = best;
                    if (stage.dsv[best >> 2] != 0.f) {
                        const float4* sr = reinterpret_cast<const float4*>(stage.gstage + (int64_t)best * dim);
#pragma unroll
                        for (int v = 0; v < NV; ++v) {
                            const int i = v * 32 + gl;
                            if (i < nvec) { const float4 x = sr[i]; gv[v].x += x.x; gv[v].y += x.y; gv[v].z += x.z; gv[v].w += x.w; }
                        }
                    }
                }
                if (gl == 0) { stage.count[row] = 0; if (cnt > stage.cap) stage.head[row] = 0; }
            }
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
                if (zero && !stage.gstage && (gv[v].x != 0.f || gv[v].y != 0.f || gv[v].z != 0.f || gv[v].w != 0.f)) gr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
}

static bool rows4_ok(const float* p, const float* g, const float* s1, const float* s2, int dim) {
    return (dim & 3) == 0 && dim <= 1024 && ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)s1 | (uintptr_t)s2) & 15) == 0);
}

template <int KIND>
static int launch_rows_kind(float* p, float* g, float* s1, float* s2, int64_t rows, int dim, OptArgs a, int zero, int normalize,
                            const float* dh, const unsigned* touched, unsigned* tclear, const RowStage& stage, hipStream_t s,
                            const RelRider* rider = nullptr) {
    if (rows4_ok(p, g, s1, s2, dim)) {
        int64_t blocks4 = (rows + 7) / 8;
#ifndef KGE_ROWS4_CAP
#define KGE_ROWS4_CAP (256 * 32)
#endif
        if (blocks4 > KGE_ROWS4_CAP) blocks4 = KGE_ROWS4_CAP;   // (A/B: profiles/r06_opt_rows4_ab.txt)
        const int streams = KIND == KGE_OPT_SGD ? 2 : KIND == KGE_OPT_ADAM ? 4 : 3;
        const bool nt = (int64_t)streams * rows * dim * 4 > ((int64_t)256 << 20);
        const RelRider none{};
        if (rider) {     // the wide-row table's chunks ride in front of the row owners (RESCAL: always the renormalising form)
            if (!normalize) { set_error("kge_optimizer_step_rows_rownorm: the rider form is the renormalising one"); return -1; }
            const dim3 grid((unsigned)(blocks4 + rider->nblocks));
#define KGE_ROWS4R(NV_)                                                                                                   \
            if (dim <= 128 * NV_) {                                                                                        \
                if (nt) hipLaunchKernelGGL((k_opt_rows4<KIND, NV_, true, true, true>), grid, dim3(256), 0, s, p, g, s1, s2, rows, dim, a, dh, zero, touched, tclear, stage, *rider); \
                else hipLaunchKernelGGL((k_opt_rows4<KIND, NV_, true, false, true>), grid, dim3(256), 0, s, p, g, s1, s2, rows, dim, a, dh, zero, touched, tclear, stage, *rider); \
                return check_launch("k_opt_rows4 (+ rider)");                                                              \
            }
            KGE_ROWS4R(1) KGE_ROWS4R(2) KGE_ROWS4R(4) KGE_ROWS4R(8)
#undef KGE_ROWS4R
        }
#define KGE_ROWS4(NV_)                                                                                                    \
        if (dim <= 128 * NV_) {                                                                                            \
            if (normalize && nt) hipLaunchKernelGGL((k_opt_rows4<KIND, NV_, true, true>), dim3((int)blocks4), dim3(256), 0, s, p, g, s1, s2, rows, dim, a, dh, zero, touched, tclear, stage, none); \
            else if (normalize) hipLaunchKernelGGL((k_opt_rows4<KIND, NV_, true, false>), dim3((int)blocks4), dim3(256), 0, s, p, g, s1, s2, rows, dim, a, dh, zero, touched, tclear, stage, none); \
            else if (nt) hipLaunchKernelGGL((k_opt_rows4<KIND, NV_, false, true>), dim3((int)blocks4), dim3(256), 0, s, p, g, s1, s2, rows, dim, a, dh, zero, touched, tclear, stage, none); \
            else hipLaunchKernelGGL((k_opt_rows4<KIND, NV_, false, false>), dim3((int)blocks4), dim3(256), 0, s, p, g, s1, s2, rows, dim, a, dh, zero, touched, tclear, stage, none); \
            return check_launch("k_opt_rows4");                                                                            \
        }
        KGE_ROWS4(1) KGE_ROWS4(2) KGE_ROWS4(4) KGE_ROWS4(8)
#undef KGE_ROWS4
    }
    if (rider) { set_error("kge_optimizer_step_rows_rownorm: rows of float4s only (dim %% 4 == 0, dim <= 1024, 16-byte aligned buffers)"); return -1; }
    if (stage.gstage) { set_error("kge_optimizer_step_rows_staged: rows of float4s only (dim %% 4 == 0, dim <= 1024, 16-byte aligned buffers)"); return -1; }
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

static int launch_optimizer_rows_impl(int kind, float* p, float* g, float* s1, float* s2, int64_t rows, int dim, float lr, int64_t step,
                                      int zero_grad, int normalize, const float* dev_hyper, const unsigned* touched, unsigned* touched_clear,
                                      const kge_rescal_stage* st, hipStream_t s, const RelRider* rider) {
    const OptArgs a = make_opt_args(lr, step);
    RowStage stage{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    if (st) {
        if (!st->gstage || !st->dsv || !st->count || !st->bucket || !st->head || !st->next || st->cap < 1 || st->cap > 32 || !touched) {
            set_error("kge_optimizer_step_rows_staged: incomplete stage (all buffers, 1 <= cap <= 32, and the touched-row bitmap are required)");
            return -1;
        }
        stage = RowStage{st->gstage, st->dsv, st->count, st->bucket, st->head, st->next, st->cap};
    }
    switch (kind) {
        case KGE_OPT_SGD: return launch_rows_kind<KGE_OPT_SGD>(p, g, s1, s2, rows, dim, a, zero_grad, normalize, dev_hyper, touched, touched_clear, stage, s, rider);
        case KGE_OPT_ADAM:
            if (!s1 || !s2) { set_error("adam needs two state buffers"); return -1; }
            return launch_rows_kind<KGE_OPT_ADAM>(p, g, s1, s2, rows, dim, a, zero_grad, normalize, dev_hyper, touched, touched_clear, stage, s, rider);
        case KGE_OPT_ADAGRAD:
            if (!s1) { set_error("adagrad needs a state buffer"); return -1; }
            return launch_rows_kind<KGE_OPT_ADAGRAD>(p, g, s1, s2, rows, dim, a, zero_grad, normalize, dev_hyper, touched, touched_clear, stage, s, rider);
        case KGE_OPT_RMSPROP:
            if (!s1) { set_error("rmsprop needs a state buffer"); return -1; }
            return launch_rows_kind<KGE_OPT_RMSPROP>(p, g, s1, s2, rows, dim, a, zero_grad, normalize, dev_hyper, touched, touched_clear, stage, s, rider);
    }
    set_error("kge_optimizer_step_rows: unknown optimizer %d", kind);
    return -1;
}

int launch_optimizer_rows(int kind, float* p, float* g, float* s1, float* s2, int64_t rows, int dim, float lr, int64_t step,
                          int zero_grad, int normalize, const float* dev_hyper, const unsigned* touched, unsigned* touched_clear,
                          const kge_rescal_stage* st, hipStream_t s) {
    return launch_optimizer_rows_impl(kind, p, g, s1, s2, rows, dim, lr, step, zero_grad, normalize, dev_hyper, touched, touched_clear, st, s, nullptr);
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

// ---- dense optimiser over a [rows, dim] table of WIDE rows (RESCAL's relation matrices: 37 x 40 000) followed by the in-place row
// renormalisation Rescal.embed applies at the next forward (models/pairwise.py:843-844, get_normalized_data).  Separately that is
// k_opt (all streams) + k_row_sumsq_chunks (re-reads the table) + k_row_scale_chunks: here the optimiser launch itself leaves the
// chunks' sums of squares -- same chunking (4 096 floats), same element-to-thread map and summation order as k_row_sumsq_chunks, so the
// stored rows are bit-identical to the three-launch form -- and only the rescale pass follows.
template <int KIND>
__global__ __launch_bounds__(256) void k_opt_sumsq_chunks(float* __restrict__ p, float* __restrict__ g, float* __restrict__ s1,
                                                          float* __restrict__ s2, int64_t dim, int nchunk, OptArgs a,
                                                          const float* __restrict__ dev_hyper, int zero, float* __restrict__ part,
                                                          AdvanceArgs adv) {
    if (dev_hyper) { a.lr = dev_hyper[0]; a.step_size = dev_hyper[1]; a.bc2_sqrt = dev_hyper[2]; }
    if (adv.cin != nullptr && blockIdx.x == 0 && threadIdx.x == 0) advance_state(adv);
    __shared__ float sw[4];
    opt_sumsq_chunk<KIND>((int)blockIdx.x, p, g, s1, s2, dim, nchunk, a, zero, part, sw);
}
__global__ __launch_bounds__(256) void k_opt_scale_chunks(float* __restrict__ w, int64_t dim, int nchunk, const float* __restrict__ part) {
    const int64_t row = blockIdx.x / nchunk;
    const int ch = blockIdx.x % nchunk;
    float t = 0.f;
    for (int i = 0; i < nchunk; ++i) t += part[row * nchunk + i];
    const float nrm = sqrtf(t);
    float* p = w + row * dim;
    const int64_t lo = (int64_t)ch * kOptNormChunk, hi = min(dim, lo + kOptNormChunk);
    for (int64_t c = lo + threadIdx.x; c < hi; c += 256) p[c] = p[c] / nrm;
}

bool optimizer_rownorm_ok(int64_t rows, int64_t dim, size_t scratch_floats) {
    const int64_t nchunk = (dim + kOptNormChunk - 1) / kOptNormChunk;
    return dim >= 4 * kOptNormChunk && rows >= 1 && rows < 1024 && scratch_floats >= (size_t)(rows * nchunk);
}

int launch_optimizer_rownorm(int kind, float* p, float* g, float* s1, float* s2, int64_t rows, int64_t dim, float lr, int64_t step,
                             int zero_grad, const float* dev_hyper, float* scratch, size_t scratch_floats, const int64_t* cursor_in,
                             int64_t* cursor_out, float* hyper_out, int64_t batch_stride, int64_t n_batches, int64_t draws_per_batch,
                             hipStream_t s) {
    if (!optimizer_rownorm_ok(rows, dim, scratch_floats)) { set_error("kge_optimizer_step_rownorm: rows of at least 16 384 floats, fewer than 1 024 rows, scratch of rows * ceil(dim / 4096) floats"); return -1; }
    if ((kind != KGE_OPT_SGD && !s1) || (kind == KGE_OPT_ADAM && !s2)) { set_error("kge_optimizer_step_rownorm: optimiser state missing"); return -1; }
    AdvanceArgs adv;
    adv.cin = cursor_in; adv.cout = cursor_out; adv.hout = hyper_out;
    adv.batch_stride = batch_stride; adv.n_batches = n_batches > 0 ? n_batches : 1; adv.draws_per_batch = draws_per_batch;
    adv.lr = lr;
    const OptArgs a = make_opt_args(lr, step < 1 ? 1 : step);
    const int nchunk = (int)((dim + kOptNormChunk - 1) / kOptNormChunk);
    const dim3 grid((unsigned)(rows * nchunk));
    switch (kind) {
        case KGE_OPT_SGD: hipLaunchKernelGGL((k_opt_sumsq_chunks<KGE_OPT_SGD>), grid, dim3(256), 0, s, p, g, s1, s2, dim, nchunk, a, dev_hyper, zero_grad, scratch, adv); break;
        case KGE_OPT_ADAM: hipLaunchKernelGGL((k_opt_sumsq_chunks<KGE_OPT_ADAM>), grid, dim3(256), 0, s, p, g, s1, s2, dim, nchunk, a, dev_hyper, zero_grad, scratch, adv); break;
        case KGE_OPT_ADAGRAD: hipLaunchKernelGGL((k_opt_sumsq_chunks<KGE_OPT_ADAGRAD>), grid, dim3(256), 0, s, p, g, s1, s2, dim, nchunk, a, dev_hyper, zero_grad, scratch, adv); break;
        case KGE_OPT_RMSPROP: hipLaunchKernelGGL((k_opt_sumsq_chunks<KGE_OPT_RMSPROP>), grid, dim3(256), 0, s, p, g, s1, s2, dim, nchunk, a, dev_hyper, zero_grad, scratch, adv); break;
        default: set_error("kge_optimizer_step_rownorm: unknown optimizer %d", kind); return -1;
    }
    hipLaunchKernelGGL(k_opt_scale_chunks, grid, dim3(256), 0, s, p, dim, nchunk, scratch);
    return check_launch("k_opt_sumsq_chunks / k_opt_scale_chunks");
}

// RESCAL's whole optimiser step as TWO launches: the row-owner sweep of the entity table with the relation matrices' chunks riding
// in front (k_opt_rows4<..., RIDER>), then the rescale of the relation matrices.  Stored values are bit-identical to
// launch_optimizer_rows + launch_optimizer_rownorm (same device functions, same operation order).
int launch_optimizer_rows_rownorm(int kind, float* p, float* g, float* s1, float* s2, int64_t rows, int dim, float* wp, float* wg,
                                  float* ws1, float* ws2, int64_t wrows, int64_t wdim, float lr, int64_t step, int zero_grad, int normalize,
                                  const float* dev_hyper, const unsigned* touched, unsigned* touched_clear, const kge_rescal_stage* st,
                                  float* scratch, size_t scratch_floats, const int64_t* cursor_in, int64_t* cursor_out, float* hyper_out,
                                  int64_t batch_stride, int64_t n_batches, int64_t draws_per_batch, hipStream_t s) {
    if (!optimizer_rownorm_ok(wrows, wdim, scratch_floats)) { set_error("kge_optimizer_step_rows_rownorm: wide rows of at least 16 384 floats, fewer than 1 024 of them, scratch of rows * ceil(dim / 4096) floats"); return -1; }
    if ((kind != KGE_OPT_SGD && !ws1) || (kind == KGE_OPT_ADAM && !ws2)) { set_error("kge_optimizer_step_rows_rownorm: optimiser state of the wide-row table missing"); return -1; }
    RelRider r;
    r.p = wp; r.g = wg; r.s1 = ws1; r.s2 = ws2; r.dim = wdim;
    r.nchunk = (int)((wdim + kOptNormChunk - 1) / kOptNormChunk);
    r.nblocks = (int)(wrows * r.nchunk);
    r.zero = zero_grad; r.part = scratch;
    r.adv.cin = cursor_in; r.adv.cout = cursor_out; r.adv.hout = hyper_out;
    r.adv.batch_stride = batch_stride; r.adv.n_batches = n_batches > 0 ? n_batches : 1; r.adv.draws_per_batch = draws_per_batch;
    r.adv.lr = lr;
    int rc = launch_optimizer_rows_impl(kind, p, g, s1, s2, rows, dim, lr, step < 1 ? 1 : step, st ? 0 : zero_grad, normalize, dev_hyper, touched,
                                        touched_clear, st, s, &r);
    if (rc) return rc;
    hipLaunchKernelGGL(k_opt_scale_chunks, dim3((unsigned)r.nblocks), dim3(256), 0, s, wp, wdim, r.nchunk, scratch);
    return check_launch("k_opt_scale_chunks");
}

}  // namespace kge
