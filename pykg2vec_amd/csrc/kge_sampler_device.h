// kge_sampler_device.h -- device-side negative corruption shared by the stand-alone sampler (kge_sampler.hip) and the
// fused sample+score+loss+backward training kernel (kge_score.hip).  See kge_sampler.hip for the semantics.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kge {

constexpr unsigned long long kEmpty = 0xFFFFFFFFFFFFFFFFull;

__host__ __device__ __forceinline__ unsigned long long pack_triple(int64_t h, int64_t r, int64_t t) {
    return ((unsigned long long)(h & 0xFFFFFF) << 40) | ((unsigned long long)(r & 0xFFFF) << 24) |
           (unsigned long long)(t & 0xFFFFFF);
}
__host__ __device__ __forceinline__ unsigned long long mix64(unsigned long long x) {  // splitmix64 finaliser
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}

__device__ __forceinline__ bool set_contains(const unsigned long long* __restrict__ slots, unsigned long long mask,
                                             unsigned long long key) {
    unsigned long long s = mix64(key) & mask;
    for (;;) {
        const unsigned long long v = slots[s];
        if (v == key) return true;
        if (v == kEmpty) return false;
        s = (s + 1) & mask;
    }
}

struct Philox {
    uint32_t c[4];
};
__host__ __device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) {
    return (uint32_t)(((unsigned long long)a * b) >> 32);
}
// Philox4x32-10 (Salmon et al., SC'11)
__host__ __device__ __forceinline__ Philox philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                         uint32_t k1) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = mulhi32(M0, c0), lo0 = M0 * c0;
        const uint32_t hi1 = mulhi32(M1, c2), lo1 = M1 * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    Philox o;
    o.c[0] = c0; o.c[1] = c1; o.c[2] = c2; o.c[3] = c3;
    return o;
}

// one corrupted triple for negative slot `ctr` of positive (h, r, t): Philox block a = 0, 1, 2, ... per attempt,
// word 0 of block 0 decides head/tail, word 1 of each block is the candidate entity
__device__ __forceinline__ void corrupt_one(int64_t h, int64_t r, int64_t t, int64_t E, const float* __restrict__ bern,
                                            const unsigned long long* __restrict__ slots, unsigned long long mask,
                                            unsigned long long seed, unsigned long long ctr, int64_t& oh, int64_t& ot) {
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    Philox x = philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, k0, k1);
    const float u = (float)(x.c[0] >> 8) * (1.0f / 16777216.0f);  // 24-bit uniform in [0,1)
    const float prob = bern ? bern[r] : 0.5f;
    const bool corrupt_tail = u > prob;
    uint32_t attempt = 0;
    int64_t e;
    for (;;) {
        e = (int64_t)(((unsigned long long)x.c[1] * (unsigned long long)E) >> 32);  // uniform in [0,E)
        const unsigned long long key = corrupt_tail ? pack_triple(h, r, e) : pack_triple(e, r, t);
        if (slots == nullptr || !set_contains(slots, mask, key)) break;
        ++attempt;
        x = philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), attempt, 0u, k0, k1);
    }
    oh = corrupt_tail ? h : e;
    ot = corrupt_tail ? e : t;
}


}  // namespace kge
