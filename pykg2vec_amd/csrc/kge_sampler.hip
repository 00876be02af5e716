// kge_sampler.hip -- negative corruption on the device (replaces the python worker processes of
// pykg2vec/data/generator.py:42-97,99-158).
//
// Semantics kept from the reference: for every positive (h,r,t) and each of its neg_rate slots draw u ~ U[0,1);
// u > prob -> corrupt the TAIL else the HEAD, prob = relation_property[r] (bern) or 0.5; the replacement entity is
// drawn uniformly from [0, E) and re-drawn while the corrupted triple is a TRAIN triple; negatives of positive i
// occupy rows [i*neg_rate, (i+1)*neg_rate).
// What cannot be kept: the reference's RNG stream (unseeded numpy MT19937 inside worker processes).  Here the
// stream is counter-based Philox4x32-10, key = seed, counter = (slot index, redraw round), so a batch is a pure
// function of (seed, offset) -- reproducible and identical however the batch is split over GPUs.
//
// The train-triple membership test is an open-addressing hash set of packed 64-bit keys
// (h:24 | r:16 | t:24 bits) in HBM, ~16 B per train triple at load factor <= 0.5, linear probing.
#include "kge_internal.h"
#include "kge_sampler_device.h"

namespace kge {

__global__ void k_set_insert(const int64_t* __restrict__ triples, int64_t n, unsigned long long* __restrict__ slots,
                             unsigned long long mask) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long key = pack_triple(triples[3 * i], triples[3 * i + 1], triples[3 * i + 2]);
    unsigned long long s = mix64(key) & mask;
    for (;;) {
        const unsigned long long prev = atomicCAS(slots + s, kEmpty, key);
        if (prev == kEmpty || prev == key) return;
        s = (s + 1) & mask;
    }
}

struct SampleArgs {
    const int64_t* triples; const int64_t* perm; int64_t start;   // positives = triples[perm[start + i]] when perm != null
    const int64_t* ph; const int64_t* pr; const int64_t* pt;      // ... else explicit positives
    int64_t n_pos; int neg_rate; int64_t E;
    const float* bern; const unsigned long long* slots; unsigned long long mask;
    unsigned long long seed, offset;
    int layout;                                                    // 0 pairwise: (oph,opr,opt) + (nh,nr,nt); 1 pointwise rows
    int64_t *o0, *o1, *o2, *o3, *o4, *o5;
    const int64_t* cursor;                                         // optional device {start, offset}: hipGraph replays
};

__global__ __launch_bounds__(256) void k_sample(SampleArgs a) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.n_pos * a.neg_rate) return;
    const int64_t i = j / a.neg_rate;
    const int k = (int)(j - i * a.neg_rate);
    const int64_t start = a.cursor ? a.start + a.cursor[0] : a.start;
    const unsigned long long offset = a.cursor ? a.offset + (unsigned long long)a.cursor[1] : a.offset;
    int64_t h, r, t;
    if (a.perm) {
        const int64_t row = a.perm[start + i];
        h = a.triples[3 * row]; r = a.triples[3 * row + 1]; t = a.triples[3 * row + 2];
    } else {
        h = a.ph[i]; r = a.pr[i]; t = a.pt[i];
    }
    int64_t oh, ot;
    corrupt_one(h, r, t, a.E, a.bern, a.slots, a.mask, a.seed, offset + (unsigned long long)j, oh, ot);
    if (a.layout == 0) {
        if (k == 0 && a.o0) { a.o0[i] = h; a.o1[i] = r; a.o2[i] = t; }
        a.o3[j] = oh; a.o4[j] = r; a.o5[j] = ot;
    } else {  // data/generator.py:125-156: positive row (y=+1) followed by its neg_rate negatives (y=-1)
        const int64_t base = i * (a.neg_rate + 1);
        if (k == 0) { a.o0[base] = h; a.o1[base] = r; a.o2[base] = t; a.o3[base] = 1; }
        a.o0[base + 1 + k] = oh; a.o1[base + 1 + k] = r; a.o2[base + 1 + k] = ot; a.o3[base + 1 + k] = -1;
    }
}

static int launch_sample(const SampleArgs& a, hipStream_t s) {
    if (a.E >= (1 << 24)) { set_error("kge sampler: more than 2^24 entities not supported by the packed key"); return -1; }
    const int64_t tot = a.n_pos * a.neg_rate;
    hipLaunchKernelGGL(k_sample, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, a);
    return check_launch("k_sample");
}

int launch_triple_set_build(const int64_t* triples, int64_t n, uint64_t* slots, int64_t n_slots, hipStream_t s) {
    hipError_t e = hipMemsetAsync(slots, 0xFF, (size_t)n_slots * sizeof(uint64_t), s);
    if (e != hipSuccess) { set_error("kge_triple_set_build: memset: %s", hipGetErrorString(e)); return -2; }
    if (n > 0)
        hipLaunchKernelGGL(k_set_insert, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, triples, n,
                           (unsigned long long*)slots, (unsigned long long)(n_slots - 1));
    return check_launch("k_set_insert");
}

int launch_corrupt(const int64_t* ph, const int64_t* pr, const int64_t* pt, int64_t n_pos, int neg_rate, int64_t E,
                   const float* bern, const uint64_t* slots, int64_t n_slots, uint64_t seed, uint64_t offset,
                   int64_t* nh, int64_t* nr, int64_t* nt, hipStream_t s) {
    SampleArgs a{};
    a.ph = ph; a.pr = pr; a.pt = pt; a.n_pos = n_pos; a.neg_rate = neg_rate; a.E = E; a.bern = bern;
    a.slots = (const unsigned long long*)slots; a.mask = (unsigned long long)(slots ? n_slots - 1 : 0);
    a.seed = seed; a.offset = offset; a.layout = 0; a.o3 = nh; a.o4 = nr; a.o5 = nt;
    return launch_sample(a, s);
}

int launch_sample_batch(const int64_t* triples, const int64_t* perm, int64_t start, int64_t n_pos, int neg_rate,
                        int64_t E, const float* bern, const uint64_t* slots, int64_t n_slots, uint64_t seed,
                        uint64_t offset, int layout, int64_t* const out[6], const int64_t* cursor, hipStream_t s) {
    SampleArgs a{};
    a.cursor = cursor;
    a.triples = triples; a.perm = perm; a.start = start; a.n_pos = n_pos; a.neg_rate = neg_rate; a.E = E; a.bern = bern;
    a.slots = (const unsigned long long*)slots; a.mask = (unsigned long long)(slots ? n_slots - 1 : 0);
    a.seed = seed; a.offset = offset; a.layout = layout;
    a.o0 = out[0]; a.o1 = out[1]; a.o2 = out[2]; a.o3 = out[3]; a.o4 = out[4]; a.o5 = out[5];
    return launch_sample(a, s);
}

// ---- device-resident step state for hipGraph replays: cursor = {start, draws, opt_step, batch_idx}, hyper = {lr,
// step_size, bc2_sqrt}.  One thread; runs first in the captured step.
__global__ void k_step_advance(int64_t* cursor, float* hyper, int64_t batch_stride, int64_t n_batches,
                               int64_t draws_per_batch, float lr) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int64_t b = cursor[3];               // batch index of THIS step
    cursor[0] = b * batch_stride;              // start of the batch in the permutation
    cursor[1] = cursor[4];                     // Philox counter offset of this step
    cursor[4] += draws_per_batch;
    cursor[3] = (b + 1) % n_batches;
    const int64_t t = ++cursor[2];             // optimiser step, 1-based
    const double bc1 = 1.0 - pow(0.9, (double)t), bc2 = 1.0 - pow(0.999, (double)t);
    hyper[0] = lr;
    hyper[1] = (float)((double)lr / bc1);
    hyper[2] = (float)sqrt(bc2);
}

int launch_step_advance(int64_t* cursor, float* hyper, int64_t batch_stride, int64_t n_batches, int64_t draws_per_batch,
                        float lr, hipStream_t s) {
    hipLaunchKernelGGL(k_step_advance, dim3(1), dim3(64), 0, s, cursor, hyper, batch_stride, n_batches, draws_per_batch, lr);
    return check_launch("k_step_advance");
}

}  // namespace kge
