// kge_mfma_blocks.h -- operand layout shared by the batch-as-M GEMM kernels on v_mfma_f32_16x16x4_f32 (kge_ntn.hip: k_ntn_rows /
// k_ntn_outer / k_ntn_lin; kge_dense.hip: k_rescal_rows).  A: lane holds A[row = lane & 15][k = lane >> 4]; B: B[k = lane >> 4][col];
// C / D: row = 4 (lane >> 4) + reg, col = lane & 15 of the block.
#pragma once
#include <type_traits>
#include <utility>
namespace kge {

typedef float f32x4v __attribute__((ext_vector_type(4)));

// NB column (or row) blocks of 16: a lane's operands for groups of four blocks are 16 consecutive bytes of an LDS row (then
// 8, then 4), i.e. block b of a group of four holds elements 4 l + b.  at(b, l): element of lane-in-block l of block b.
template <int NB> struct BlkMap {
    static constexpr int G4 = NB / 4, G2 = (NB % 4) / 2, G1 = NB % 2;
    __host__ __device__ static constexpr int at(int b, int l) {
        return b < 4 * G4 ? 64 * (b / 4) + 4 * l + (b % 4) : b < 4 * G4 + 2 * G2 ? 64 * G4 + 2 * l + (b - 4 * G4) : 64 * G4 + 32 * G2 + l;
    }
};
// The same reads with NATURAL output columns (block b, lane l = column 16 b + l: a kernel whose accumulators leave through row-wise
// atomics wants 16 consecutive floats per lane group): column c0 + 16 u of a slab row is STORED at pos(u, c0), the place read_blocks
// hands to block u, lane c0.
template <int NB> struct BlkMapNat {
    static constexpr int G4 = NB / 4, G2 = (NB % 4) / 2;
    __host__ __device__ static constexpr int pos(int u, int c0) {
        return u < 4 * G4 ? 64 * (u / 4) + 4 * c0 + (u % 4) : u < 4 * G4 + 2 * G2 ? 64 * G4 + 2 * c0 + (u - 4 * G4) : 64 * G4 + 32 * G2 + c0;
    }
};
template <int NB>
__device__ __forceinline__ void read_blocks(const float* __restrict__ rowp, int l, float (&b)[NB]) {
    using M = BlkMap<NB>;
#pragma unroll
    for (int g = 0; g < M::G4; ++g) {
        const float4 v = *reinterpret_cast<const float4*>(rowp + 64 * g + 4 * l);
        b[4 * g] = v.x; b[4 * g + 1] = v.y; b[4 * g + 2] = v.z; b[4 * g + 3] = v.w;
    }
    if constexpr (M::G2) {
        const float2 v = *reinterpret_cast<const float2*>(rowp + 64 * M::G4 + 2 * l);
        b[4 * M::G4] = v.x; b[4 * M::G4 + 1] = v.y;
    }
    if constexpr (M::G1) b[NB - 1] = rowp[64 * M::G4 + 32 * M::G2 + l];
}

// (compile-time check of the two maps for every block count in use: read_blocks hands LDS position 64 g + 4 l + q to block 4 g + q,
// lane l -- BlkMap calls that element at(b, l); BlkMapNat stores column 16 b + l exactly there)
template <int NB> constexpr bool blk_maps_agree() {
    for (int b = 0; b < NB; ++b)
        for (int l = 0; l < 16; ++l)
            if (BlkMapNat<NB>::pos(b, l) != BlkMap<NB>::at(b, l) || BlkMap<NB>::at(b, l) >= 16 * NB) return false;
    return true;
}
static_assert(blk_maps_agree<1>() && blk_maps_agree<2>() && blk_maps_agree<3>() && blk_maps_agree<4>() && blk_maps_agree<5>() &&
              blk_maps_agree<6>() && blk_maps_agree<7>() && blk_maps_agree<8>() && blk_maps_agree<13>(), "block maps");

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop the compiler cannot decline to unroll (register arrays
// indexed by the loop variable stay registers whatever the body size)
template <class F, int... I>
__device__ __forceinline__ void unroll_seq(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }

// Only every S-th tile of a relation starts a workgroup's work (S = 2: pairs of tiles; S = run length); launched in tile order the
// live workgroups of a relation would all sit on block ids that are S apart, i.e. on 8 / gcd(S, 8) of the 8 XCDs (blocks go to
// XCDs round robin) -- with S = 8 a whole relation on ONE XCD.  Blocks are therefore numbered class by class: block v works on tile
// (v mod Q) S + v / Q, Q = ceil(tiles / S), so that the live tiles of a relation are CONSECUTIVE block ids.  The grid has Q S blocks.
template <int S>
__device__ __forceinline__ int strided_tile(int tiles) {
    const int Q = (tiles + S - 1) / S;
    return ((int)blockIdx.x % Q) * S + (int)blockIdx.x / Q;
}

}  // namespace kge
