// kge_rescal_slab.hip -- the pairwise RESCAL step (models/pairwise.py:829-865 forward of both sides, utils/criterion.py:25-29 hinge,
// loss.backward()) at the reference's batch sizes, as TWO launches of (relation chunk, 32-column slab of M_r) workgroups.
//
// Why.  k_rescal_pair (kge_dense.hip) gives a (relation, 16 pairs) tile to ONE 1024-thread workgroup that walks the whole of M_r
// three times: at the C4 shape (YAGO3-10: 37 relations, B = 1 024, k = 200) that is 102 workgroups with 86 KB of LDS each on a
// 256-CU chip -- 40 % of the CUs, one workgroup per CU, each ~16 us of f32 MFMA issue behind ~30 us of DEPENDENT round trips
// (tile -> offsets -> perm -> ids -> rows -> M_r -> ... ; 46.9 us per launch, profiles/r04_kernel_stats.md).  A dependent global
// round trip costs ~1.5 us on this chip whatever it fetches, so the step is priced in chain depth.  The contraction h^T M_r t is
// separable over the COLUMNS of M_r up to the final sum, and the relation-matrix gradient G = sum_i ds_i h_i t_i^T over its ROWS, so
// here a workgroup owns (relation chunk of <= 64 pairs, slab of 32 columns / rows of M_r) and its chain is three trips deep:
//   [tile descriptor]  ->  [grouped ids | the slab of M_r -> LDS | energy shares | old G]  ->  [entity rows]  ->  MFMA  ->  stores
// (the grouping launch leaves one descriptor per tile and the four entity ids of every pair in grouped order, kge_dense.hip:
// k_rel_group_small).  Wave w of a workgroup owns row block w of the chunk (16 pairs: rows 0..15 the positives, 16..31 their
// negatives) over the WHOLE K range, so V and U need no cross-wave reduction and no barrier after the operands are staged.
//   k_rescal_slab_fwd   V[:, slab] = H M_r[:, slab], the slab's share of every energy  sum_{b in slab} V[i][b] T[i][b], V to the workspace.
//   k_rescal_slab_bwd   energies = the slabs' shares added in slab order, hinge (recomputed by each slab workgroup of the chunk: a few
//                       loads), U[:, slab] = T M_r[slab, :]^T -> grad_h[:, slab] = -ds U;  grad_t[:, slab] = -ds V from the workspace;
//                       G[slab, :] = sum_i ds_i H[i][slab] T[i][:] accumulated in registers over the chunk's row blocks (waves split the
//                       column tiles) and written ONCE with plain read-modify-writes -- one owner per (relation, row slab): no atomics on
//                       the relation-matrix gradient -- atomically only where a relation spans several chunks.
// 37 relations x 7 slabs = 259 four-wave workgroups fill the chip.  Every global load sits on a clamped address and is masked where
// its value is USED (profiles/r04_experiments.md section 5).  Blocks are numbered so that the slabs of one chunk share an XCD (they
// gather the same entity rows).  Entity gradients still leave through float atomics (merged for the uncorrupted side of a pair).
// k even, k <= 256; VK = 4 (k % 4 == 0, 16-byte aligned tables) or 2.
#include "kge_relgroup.h"
#include "kge_device.h"

#ifdef KGE_SLAB_TS   /* experiment build only (tools/slab_phases.py): wall_clock64 (100 MHz) of thread 0 at the phase boundaries */
namespace kge { namespace { __device__ unsigned long long g_slab_ts[2][1024][12]; } }
extern "C" int kge_ts_dump_slab(int which, unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(kge::g_slab_ts), sizeof(unsigned long long) * 1024 * 12, (size_t)which * 1024 * 12 * sizeof(unsigned long long));
}
#define SLAB_TS(which, n) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
    if (threadIdx.x == 0 && blockIdx.x < 1024) kge::g_slab_ts[which][blockIdx.x][n] = wall_clock64(); }
#else
#define SLAB_TS(which, n)
#endif

namespace kge {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kSlabBlocks = kSlabChunk / kPairTile;   // row blocks per chunk = waves per workgroup

struct SlabArgs {
    const float* ent; const float* relm; float* g_ent; float* g_rel;
    const int* tile_off;       // [R + 1]: tile_off[R] = number of tiles
    const int4* tdesc;         // per tile: (relation, first grouped position, pairs, tiles of the relation)
    const int4* gids;          // per grouped position: (ph, pt, nh, nt)
    int R, k, n_slab;
    int64_t n;
    float margin;
    float* loss;
    unsigned* touched;
    float* wsV;      // [2 n][k]      V rows in grouped order: slot 2 g + side
    float* wsP;      // [n_slab][2 n] the slabs' shares of every h^T M t
    // staged entity gradients (NULL: float atomics into g_ent): row 4 g + j of gstage = the gradient pair g contributes to its positive
    // head (j = 0), positive tail (1), negative head (2), negative tail (3), written with plain stores where the pair's hinge
    // coefficient dsv[g] is not 0; the row owners of the optimiser sum them in slot order (kge_opt.hip: k_opt_rows4<..., STAGED>)
    float* gstage;
    float* dsv;      // [n] hinge coefficient of grouped pair g
    int32_t* st_count; int32_t* st_bucket; int32_t* st_head; int32_t* st_next; int st_cap;   // per-entity lists of staged rows
};

template <int VK>
__device__ __forceinline__ void ldvec(float (&o)[VK], const float* __restrict__ p) {
    if constexpr (VK == 4) { const float4 q = *reinterpret_cast<const float4*>(p); o[0] = q.x; o[1] = q.y; o[2] = q.z; o[3] = q.w; }
    else { const float2 q = *reinterpret_cast<const float2*>(p); o[0] = q.x; o[1] = q.y; }
}

// block -> (chunk tile, slab): tile t lives on XCD t % 8 with all of its slabs (consecutive block ids go round-robin over the XCDs)
__device__ __forceinline__ void slab_block(int n_slab, int& tile, int& slab) {
    const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
    tile = x + 8 * (q / n_slab);
    slab = q % n_slab;
}

// accumulator register reg of a 32x32 f32 MFMA holds row (reg & 3) + 8 (reg >> 2) + 4 lk, column li
__device__ __forceinline__ int acc_row(int reg, int lk) { return (reg & 3) + 8 * (reg >> 2) + 4 * lk; }

template <int VK, int KF>     // KF: the largest k this instantiation covers (a multiple of 2 VK)
__global__ __launch_bounds__(256) void k_rescal_slab_fwd(SlabArgs a) {
    constexpr int NJ = KF / (2 * VK);             // operand pieces of VK floats per lane over the whole K range
    constexpr int NU = (KF + 7) / 8;
    __shared__ float sM[KF * 33];                 // the slab: sM[kk * 33 + c] = M_r[kk][slab * 32 + c]
    int tile, slab;
    slab_block(a.n_slab, tile, slab);
    SLAB_TS(0, 0)
    const int ntiles = a.tile_off[a.R];
    const int4 d = a.tdesc[min(tile, max(ntiles - 1, 0))];
    if (tile >= ntiles) return;
    SLAB_TS(0, 1)
    const int rel = d.x, g_lo = d.y, cnt = d.z;
    const int k = a.k;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lk = lane >> 5;
    const int col = slab * 32 + li, colc = min(col, k - 1);
    const float* __restrict__ M = a.relm + (int64_t)rel * k * k;
    // row li of this wave's block: li < 16 the positive of pair 16 wave + li, li >= 16 the negative of pair 16 wave + li - 16
    const int p = wave * kPairTile + (li & 15);
    const int4 id4 = a.gids[g_lo + min(p, cnt - 1)];
    {   // the slab of M_r -> LDS: thread (c, r0) takes rows r0, r0 + 8, ... of column c (coalesced over c)
        const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5, cc = min(slab * 32 + c, k - 1);
        float v[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) v[u] = M[(int64_t)min(r0 + 8 * u, k - 1) * k + cc];
        SLAB_TS(0, 2)
#pragma unroll
        for (int u = 0; u < NU; ++u)   // rows k .. KF - 1 are staged as zeros: the MFMA steps beyond k then add nothing, whatever A holds
            if (r0 + 8 * u < KF) sM[(r0 + 8 * u) * 33 + c] = r0 + 8 * u < k ? v[u] : 0.f;
    }
    const bool neg = li >= kPairTile;
    const int hid = neg ? id4.z : id4.x, tid = neg ? id4.w : id4.y;
    if (a.gstage && slab == 0 && lk == 0 && p < cnt) {
        // staged entity gradients: this triple's two gradient rows (slot 4 g + 2 side for its head, + 1 for its tail) register with
        // their entities -- count / bucket / overflow chain, all zero between steps -- and mark them for the optimiser.  Done here, by
        // the first slab's workgroup of every chunk, so that the returning atomics sit under the operand loads below.
        const int slot_h = 4 * (g_lo + p) + (neg ? 2 : 0);
        const int ph_ = atomicAdd(a.st_count + hid, 1), pt_ = atomicAdd(a.st_count + tid, 1);
        if (ph_ < a.st_cap) a.st_bucket[(int64_t)hid * a.st_cap + ph_] = slot_h;
        else a.st_next[slot_h] = atomicExch(a.st_head + hid, slot_h + 1) - 1;      // head holds slot + 1: all-zero = empty
        if (pt_ < a.st_cap) a.st_bucket[(int64_t)tid * a.st_cap + pt_] = slot_h + 1;
        else a.st_next[slot_h + 1] = atomicExch(a.st_head + tid, slot_h + 2) - 1;
        atomicOr(a.touched + (hid >> 5), 1u << (hid & 31));
        atomicOr(a.touched + (tid >> 5), 1u << (tid & 31));
    }
    const bool active = wave * kPairTile < cnt;      // (uniform) this wave's row block exists
    float av[NJ][VK];
    float tv[16];
    if (active) {
        // A operand: the head row of this lane's triple, VK floats per load.  The k order inside an MFMA step is free: lane half lk
        // carries k = kb .. kb + VK - 1 of piece j over VK consecutive steps (the B operand follows the same order)
#pragma unroll
        for (int j = 0; j < NJ; ++j) ldvec<VK>(av[j], a.ent + (int64_t)hid * k + min(2 * VK * j + VK * lk, k - VK));
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)   // the tail rows' slab columns for the energy shares (accumulator layout)
            tv[reg] = a.ent[(int64_t)__shfl(tid, acc_row(reg, lk), 64) * k + colc];
    }
    SLAB_TS(0, 3)
    __syncthreads();
    if (!active) return;
    // K loop: every piece of the instantiation (rows of the slab beyond k are zeros), no branches; the B operand comes from LDS one
    // piece ahead of the MFMA steps that consume it; two accumulators alternate (a dependent f32 MFMA chain issues every ~100
    // cycles, two interleaved chains every 64).  Rows of dead pairs carry finite garbage that is never stored.
    f32x16 acc = {0}, acc_odd = {0};
    const float* __restrict__ sB = sM + (VK * lk) * 33 + li;
    float bq[2][VK];
#pragma unroll
    for (int q = 0; q < VK; ++q) bq[0][q] = sB[q * 33];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        if (j + 1 < NJ) {
#pragma unroll
            for (int q = 0; q < VK; ++q) bq[(j + 1) & 1][q] = sB[(2 * VK * (j + 1) + q) * 33];
        }
#pragma unroll
        for (int q = 0; q < VK; ++q) {
            if (q & 1) acc_odd = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j][q], bq[j & 1][q], acc_odd, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j][q], bq[j & 1][q], acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) acc[reg] += acc_odd[reg];
#ifdef KGE_SLAB_TS
    asm volatile("s_nop 0" :: "v"(acc[0]));
#endif
    SLAB_TS(0, 4)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int i = acc_row(reg, lk);
        const int pi = wave * kPairTile + (i & 15);
        const int64_t slot = 2 * (int64_t)(g_lo + pi) + (i >> 4);
        const bool live = pi < cnt;
        if (live && col < k) a.wsV[slot * k + col] = acc[reg];
        const float pr = gsum<32>(col < k ? acc[reg] * tv[reg] : 0.f);   // over the 32 columns of this half-wave (DPP steps)
        if (li == 0 && live) a.wsP[(int64_t)slab * 2 * a.n + slot] = pr;
    }
    SLAB_TS(0, 5)
}

template <int VK, int KF>
__global__ __launch_bounds__(256, 2) void k_rescal_slab_bwd(SlabArgs a) {
    constexpr int NJ = KF / (2 * VK);
    constexpr int S2 = ((KF + 7) / 8) * 8 + 4;    // row pitch of the staged rows of M_r (4 mod 8 words): 16-byte reads of 8 lanes cover all banks
    __shared__ __attribute__((aligned(16))) float sM[32 * S2];   // sM[a * S2 + kk] = M_r[slab * 32 + a][kk]
    __shared__ int sHid[kSlabBlocks * 32], sTid[kSlabBlocks * 32];
    __shared__ float sDs[kSlabBlocks * 32];
    __shared__ int sAny;
    int tile, slab;
    slab_block(a.n_slab, tile, slab);
    SLAB_TS(1, 0)
    const int ntiles = a.tile_off[a.R];
    const int4 d = a.tdesc[min(tile, max(ntiles - 1, 0))];
    if (tile >= ntiles) return;
    SLAB_TS(1, 1)
    const int rel = d.x, g_lo = d.y, cnt = d.z;
    const bool multi = d.w > 1;                   // the relation spans several chunks: their shares of G add atomically
    const int k = a.k, n_slab = a.n_slab;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lk = lane >> 5;
    const int col = slab * 32 + li, colc = min(col, k - 1);
    const float* __restrict__ M = a.relm + (int64_t)rel * k * k;
    float* __restrict__ gM = a.g_rel + (int64_t)rel * k * k;
    const int p = wave * kPairTile + (li & 15);
    const bool on = p < cnt;
    const int gp = g_lo + min(p, cnt - 1);
    const int4 id4 = a.gids[gp];
    // hinge inputs of pair p (lanes 0..15 of each wave): the slabs' shares of both energies
    float sp = 0.f, sn = 0.f;
    if (lane < kPairTile) {
        for (int s = 0; s < n_slab; ++s) {
            sp += a.wsP[(int64_t)s * 2 * a.n + 2 * (int64_t)gp];
            sn += a.wsP[(int64_t)s * 2 * a.n + 2 * (int64_t)gp + 1];
        }
    }
    {   // rows slab * 32 .. + 31 of M_r -> LDS: 8 threads per row, VK floats per load (coalesced along the row)
        const int ar = threadIdx.x >> 3, seg = threadIdx.x & 7, arow = min(slab * 32 + ar, k - 1);
        constexpr int NU = (KF + 8 * VK - 1) / (8 * VK);
        float v[NU][VK];
#pragma unroll
        for (int u = 0; u < NU; ++u) ldvec<VK>(v[u], M + (int64_t)arow * k + min(VK * (seg + 8 * u), k - VK));
#pragma unroll
        for (int u = 0; u < NU; ++u)   // columns k .. KF - 1 are staged as zeros (see the forward kernel)
            if (VK * (seg + 8 * u) < KF) {
#pragma unroll
                for (int q = 0; q < VK; ++q) sM[ar * S2 + VK * (seg + 8 * u) + q] = VK * (seg + 8 * u) < k ? v[u][q] : 0.f;
            }
    }
    // this wave's output column tiles of G: bt = wave and wave + 4
    const bool second = wave + 4 < n_slab;
    const int bc0 = wave * 32 + li, bc1 = (wave + 4) * 32 + li;
    const int b0 = min(bc0, k - 1), b1 = min(bc1, k - 1);
    SLAB_TS(1, 2)
    const bool neg = li >= kPairTile;
    const int hid = neg ? id4.z : id4.x, tid = neg ? id4.w : id4.y;
    if (lk == 0) { sHid[wave * 32 + li] = hid; sTid[wave * 32 + li] = tid; }
    if (threadIdx.x == 0) sAny = 0;
    // margin hinge of this wave's pairs (energies = -h^T M t)
    float v = 0.f, c = 0.f;
    if (lane < kPairTile && on) {
        v = (-sp) + a.margin - (-sn);
        c = v > 0.f ? 1.f : (v == 0.f ? 0.5f : 0.f);
    }
    if (lane < kPairTile) {
        sDs[wave * 32 + lane] = c; sDs[wave * 32 + kPairTile + lane] = -c;
        if (a.dsv && slab == 0 && on) a.dsv[gp] = c;
    }
    const float tot = wave_sum(fmaxf(v, 0.f));
    const unsigned long long any = __ballot(c != 0.f);
    __syncthreads();
    if (lane == 0) {
        if (slab == 0 && tot != 0.f) unsafeAtomicAdd(a.loss + ((blockIdx.x + wave) % kLossSlots) * kLossStride, tot);
        if (any != 0ull) sAny = 1;
    }
    __syncthreads();
    if (!sAny) return;        // every pair of the chunk inside the margin: no gradient
    SLAB_TS(1, 3)
    const bool active = any != 0ull;              // (wave-uniform) this wave's row block carries a gradient
    if (a.touched && !a.gstage && slab == 0 && lk == 0 && sDs[wave * 32 + li] != 0.f) {   // entity rows this chunk writes a gradient into
        atomicOr(a.touched + (hid >> 5), 1u << (hid & 31));
        atomicOr(a.touched + (tid >> 5), 1u << (tid & 31));
    }
    // ---- operands requested together: the tail rows of this wave's block (A of U), its V rows (grad_t), and the G operands of the
    // chunk's first row block (ds_i H[i][slab column] as A, T[i][b] as B, i = 2 s + lk)
    float tvr[NJ][VK];
    float vp[8], vn[8];
    if (active) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) ldvec<VK>(tvr[j], a.ent + (int64_t)tid * k + min(2 * VK * j + VK * lk, k - VK));
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int pi = wave * kPairTile + r + 8 * lk;
            const int64_t slot = 2 * (int64_t)(g_lo + min(pi, cnt - 1));
            vp[r] = a.wsV[slot * k + colc];
            vn[r] = a.wsV[(slot + 1) * k + colc];
        }
    }
    const int nblk = (cnt + kPairTile - 1) / kPairTile;
    float ha[16], tb0[16], tb1[16];
    auto load_g = [&](int blk) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int i = blk * 32 + 2 * s + lk;
            ha[s] = a.ent[(int64_t)sHid[i] * k + colc];
            tb0[s] = a.ent[(int64_t)sTid[i] * k + b0];
            tb1[s] = a.ent[(int64_t)sTid[i] * k + b1];
        }
    };
    load_g(0);
    SLAB_TS(1, 4)
    // ---- U[i][col] = sum_b T[i][b] M[col][b] -> grad_h = -ds U
    if (active) {
        f32x16 ua = {0}, ua_odd = {0};     // (as in the forward kernel: no branches, LDS operand one piece ahead, two accumulator chains)
        const float* __restrict__ sB = sM + li * S2 + VK * lk;
        float mq[2][VK];
        auto read_m = [&](int j, float (&o)[VK]) __attribute__((always_inline)) {
            if constexpr (VK == 4) { const float4 q4 = *reinterpret_cast<const float4*>(sB + 2 * VK * j); o[0] = q4.x; o[1] = q4.y; o[2] = q4.z; o[3] = q4.w; }
            else { const float2 q2 = *reinterpret_cast<const float2*>(sB + 2 * VK * j); o[0] = q2.x; o[1] = q2.y; }
        };
        read_m(0, mq[0]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (j + 1 < NJ) read_m(j + 1, mq[(j + 1) & 1]);
#pragma unroll
            for (int q = 0; q < VK; ++q) {
                if (q & 1) ua_odd = __builtin_amdgcn_mfma_f32_32x32x2f32(tvr[j][q], mq[j & 1][q], ua_odd, 0, 0, 0);
                else ua = __builtin_amdgcn_mfma_f32_32x32x2f32(tvr[j][q], mq[j & 1][q], ua, 0, 0, 0);
            }
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) ua[reg] += ua_odd[reg];
#ifdef KGE_SLAB_TS
        asm volatile("s_nop 0" :: "v"(ua[0]));
#endif
        SLAB_TS(1, 5)
        if (col < k && a.gstage) {
            // staged: every side of every active pair owns a row of gstage (slot 4 g + j), plain stores, no atomics
#pragma unroll
            for (int reg = 0; reg < 8; ++reg) {
                const int i = acc_row(reg, lk);
                const float ds = sDs[wave * 32 + i];
                if (ds != 0.f) {
                    const int64_t slot = 4 * (int64_t)(g_lo + wave * kPairTile + i);
                    a.gstage[slot * k + col] = -ds * ua[reg];                 // positive head
                    a.gstage[(slot + 2) * k + col] = ds * ua[reg + 8];        // negative head
                }
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int i = r + 8 * lk;
                const float ds = sDs[wave * 32 + i];
                if (ds != 0.f) {
                    const int64_t slot = 4 * (int64_t)(g_lo + wave * kPairTile + i);
                    a.gstage[(slot + 1) * k + col] = -ds * vp[r];             // positive tail
                    a.gstage[(slot + 3) * k + col] = ds * vn[r];              // negative tail
                }
            }
        } else if (col < k) {   // rows i and i + 16 are the two sides of one pair (registers reg, reg + 8): the side the sampler did not
                         // corrupt is the SAME entity row and leaves as one atomic
#pragma unroll
            for (int reg = 0; reg < 8; ++reg) {
                const int i = acc_row(reg, lk);
                const float ds = sDs[wave * 32 + i];
                if (ds != 0.f) {
                    const int64_t ia = sHid[wave * 32 + i], ib = sHid[wave * 32 + i + kPairTile];
                    if (ia == ib) {
                        unsafeAtomicAdd(a.g_ent + ia * k + col, -ds * (ua[reg] - ua[reg + 8]));
                    } else {
                        unsafeAtomicAdd(a.g_ent + ia * k + col, -ds * ua[reg]);
                        unsafeAtomicAdd(a.g_ent + ib * k + col, ds * ua[reg + 8]);
                    }
                }
            }
            // grad_t = -ds V, from the rows the forward launch left in the workspace
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int i = r + 8 * lk;
                const float ds = sDs[wave * 32 + i];
                if (ds != 0.f) {
                    const int64_t ia = sTid[wave * 32 + i], ib = sTid[wave * 32 + i + kPairTile];
                    if (ia == ib) {
                        unsafeAtomicAdd(a.g_ent + ia * k + col, -ds * (vp[r] - vn[r]));
                    } else {
                        unsafeAtomicAdd(a.g_ent + ia * k + col, -ds * vp[r]);
                        unsafeAtomicAdd(a.g_ent + ib * k + col, ds * vn[r]);
                    }
                }
            }
        }
    }
    SLAB_TS(1, 6)
    // ---- G[slab rows][b] = sum_i ds_i H[i][slab row] T[i][b] over every row block of the chunk (rows of dead pairs carry ds = 0)
    f32x16 ga0 = {0}, ga1 = {0};
    float old0[16], old1[16];       // old values of this wave's share of grad_M: in flight under the G MFMAs
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int ar = min(slab * 32 + acc_row(reg, lk), k - 1);
        old0[reg] = gM[(int64_t)ar * k + b0];
        old1[reg] = gM[(int64_t)ar * k + b1];
    }
    for (int blk = 0; blk < nblk; ++blk) {
        float avg[16], t0[16], t1[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) { avg[s] = sDs[blk * 32 + 2 * s + lk] * ha[s]; t0[s] = tb0[s]; t1[s] = tb1[s]; }
        if (blk + 1 < nblk) load_g(blk + 1);       // (uniform) next block's operands in flight under this block's MFMAs
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            ga0 = __builtin_amdgcn_mfma_f32_32x32x2f32(avg[s], t0[s], ga0, 0, 0, 0);
            if (second) ga1 = __builtin_amdgcn_mfma_f32_32x32x2f32(avg[s], t1[s], ga1, 0, 0, 0);
        }
    }
#ifdef KGE_SLAB_TS
    asm volatile("s_nop 0" :: "v"(ga0[0]));
#endif
    SLAB_TS(1, 7)
    // ---- grad_M[slab rows][:] -= G
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (t == 1 && !second) break;
        const int b = t == 0 ? bc0 : bc1;
        if (b < k) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int ar = slab * 32 + acc_row(reg, lk);
                const float gv = t == 0 ? ga0[reg] : ga1[reg];
                if (ar < k && gv != 0.f) {
                    if (multi) unsafeAtomicAdd(gM + (int64_t)ar * k + b, -gv);
                    else gM[(int64_t)ar * k + b] = (t == 0 ? old0[reg] : old1[reg]) - gv;
                }
            }
        }
    }
    SLAB_TS(1, 8)
}

static size_t align256s(size_t x) { return (x + 255) & ~(size_t)255; }
static int64_t slab_tiles(int64_t R, int64_t n) { return n / kSlabChunk + R + 1; }   // >= sum_r ceil(n_r / kSlabChunk)

// layout of the slab form's workspace: tile descriptors | grouped ids | V rows | energy shares
size_t rescal_slab_ws_bytes(int k, int64_t R, int64_t n) {
    const int n_slab = (k + 31) / 32;
    return align256s((size_t)slab_tiles(R, n) * sizeof(int4)) + align256s((size_t)n * sizeof(int4)) +
           align256s(((size_t)2 * n * k + (size_t)n_slab * 2 * n) * sizeof(float));
}

void rescal_slab_gather(void* ws_slab, int k, int64_t R, int64_t n, PairGather* pg) {
    pg->tdesc = (int4*)ws_slab;
    pg->gids = (int4*)((char*)ws_slab + align256s((size_t)slab_tiles(R, n) * sizeof(int4)));
}

// the grouping (kSlabChunk pairs per tile, with the PairGather of rescal_slab_gather) has been enqueued on s before this call
int launch_rescal_slab_step(const kge_model_desc* m, int64_t n, const GroupWs& g, float margin, float* loss, unsigned* touched,
                            void* ws_slab, const kge_rescal_stage* stage, hipStream_t s) {
    const int k = m->dim;
    const int64_t R = m->tot_relation;
    PairGather pg;
    rescal_slab_gather(ws_slab, k, R, n, &pg);
    SlabArgs a;
    a.ent = m->tables[0]; a.relm = m->tables[1]; a.g_ent = m->grads[0]; a.g_rel = m->grads[1];
    a.tile_off = g.tile_off; a.tdesc = pg.tdesc; a.gids = pg.gids;
    a.R = (int)R; a.k = k; a.n_slab = (k + 31) / 32; a.n = n;
    a.margin = margin; a.loss = loss; a.touched = touched;
    a.gstage = stage ? stage->gstage : nullptr; a.dsv = stage ? stage->dsv : nullptr;
    a.st_count = stage ? stage->count : nullptr; a.st_bucket = stage ? stage->bucket : nullptr;
    a.st_head = stage ? stage->head : nullptr; a.st_next = stage ? stage->next : nullptr; a.st_cap = stage ? stage->cap : 0;
    a.wsV = (float*)((char*)pg.gids + align256s((size_t)n * sizeof(int4)));
    a.wsP = a.wsV + (size_t)2 * n * k;
    const unsigned grid = (unsigned)((slab_tiles(R, n) + 7) / 8 * 8 * a.n_slab);
    const bool v4 = (k & 3) == 0 && (reinterpret_cast<uintptr_t>(m->tables[0]) & 15) == 0 && (reinterpret_cast<uintptr_t>(m->tables[1]) & 15) == 0;
#define KGE_SLAB_GO(VK_, KF_)                                                                        \
    { hipLaunchKernelGGL((k_rescal_slab_fwd<VK_, KF_>), dim3(grid), dim3(256), 0, s, a);              \
      hipLaunchKernelGGL((k_rescal_slab_bwd<VK_, KF_>), dim3(grid), dim3(256), 0, s, a); }
    // instantiations by the largest k they cover (registers and LDS scale with it): 64, 128, 200 (the YAGO3-10 preset), 256
    if (v4) {
        if (k <= 64) KGE_SLAB_GO(4, 64) else if (k <= 128) KGE_SLAB_GO(4, 128) else if (k <= 200) KGE_SLAB_GO(4, 200) else KGE_SLAB_GO(4, 256)
    } else {
        if (k <= 64) KGE_SLAB_GO(2, 64) else if (k <= 128) KGE_SLAB_GO(2, 128) else if (k <= 200) KGE_SLAB_GO(2, 200) else KGE_SLAB_GO(2, 256)
    }
#undef KGE_SLAB_GO
    return check_launch("k_rescal_slab_fwd / k_rescal_slab_bwd");
}

}  // namespace kge
