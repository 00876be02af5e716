// kge_dense.hip -- RESCAL (NTN lives in kge_ntn.hip): the dense relation-matrix contraction on the f32 matrix cores.
//
// Reference: pykg2vec/models/pairwise.py:829-865.  energy = -h^T M_r t with M_r = rel_matrices[r].view(k,k); the
// reference gathers a [B,k,k] tensor (B*k^2 floats: 20 MB at B=128,k=200) and runs a batched mat-vec, and its
// autograd scatters B dense k^2 outer products back.  Here the batch is GROUPED BY RELATION on the device
// (histogram -> scan -> scatter, three tiny kernels), and every (relation, 32-triple tile) is one workgroup that
// runs three small GEMMs on v_mfma_f32_32x32x2_f32 (exact fp32, 157 TF peak):
//     U = T M_r^T   [32,k]x[k,k]   forward (score_i = -h_i . U_i) and grad_h = -ds * U
//     V = H M_r     [32,k]x[k,k]   grad_t = -ds * V
//     G = (ds*H)^T T   [k,32]x[32,k]   grad_M_r = -G      (one atomic per M element per 32 triples, not per triple)
// so M_r is read once per tile from L2 instead of once per triple, and the k^2-sized gradient traffic drops 32x.
// Also: the in-place table renormalisation Rescal.embed performs on every forward (pairwise.py:843-844,862-865).
//
// MFMA operand maps (cdna_hip_programming.md section 3): A: lane l holds A[i=l&31][k=l>>5]; B: lane l holds
// B[k=l>>5][j=l&31]; C/D: col j = lane&31, row i = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#include "kge_internal.h"
#include "kge_relgroup.h"
#include "kge_mfma_blocks.h"
#include <stdlib.h>
#include <type_traits>

namespace kge {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k_rel_hist(IdSplit r, int64_t n, int* __restrict__ counts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(counts + r.at(i), 1);
}

// single block: exclusive scans of counts and of ceil(counts / TILE)
__global__ __launch_bounds__(256) void k_rel_scan(const int* __restrict__ counts, int R, int* __restrict__ offsets,
                                                  int* __restrict__ tile_off, int tile) {
    __shared__ int s_a[256], s_b[256];
    int run_a = 0, run_b = 0;
    for (int base = 0; base < R; base += 256) {
        const int idx = base + threadIdx.x;
        const int c = idx < R ? counts[idx] : 0;
        const int tl = (c + tile - 1) / tile;
        s_a[threadIdx.x] = c; s_b[threadIdx.x] = tl;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {  // Hillis-Steele inclusive scan
            int va = 0, vb = 0;
            if ((int)threadIdx.x >= d) { va = s_a[threadIdx.x - d]; vb = s_b[threadIdx.x - d]; }
            __syncthreads();
            s_a[threadIdx.x] += va; s_b[threadIdx.x] += vb;
            __syncthreads();
        }
        if (idx < R) { offsets[idx] = run_a + s_a[threadIdx.x] - c; tile_off[idx] = run_b + s_b[threadIdx.x] - tl; }
        run_a += s_a[255]; run_b += s_b[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) { offsets[R] = run_a; tile_off[R] = run_b; }
}

__global__ void k_rel_scatter(IdSplit r, int64_t n, const int* __restrict__ offsets,
                              const int* __restrict__ tile_off, int* __restrict__ cursor, int* __restrict__ perm,
                              int* __restrict__ tile_rel, int tile) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int rel = (int)r.at(i);
    const int local = atomicAdd(cursor + rel, 1);
    perm[offsets[rel] + local] = (int)i;
    if (local % tile == 0) tile_rel[tile_off[rel] + local / tile] = rel;  // the first row of a tile names its relation
}

// Large batches over few relations (tens of thousands of items on a few dozen counters): the histogram and the scatter cursors
// are taken per block in LDS first, so the global counters see one atomic per (block, relation with items in the block).
constexpr int kGroupItems = 4096;      // items per 1024-thread block
__global__ __launch_bounds__(1024) void k_rel_hist_lds(IdSplit r, int64_t n, int R, int* __restrict__ counts) {
    extern __shared__ int s_h[];
    for (int i = threadIdx.x; i < R; i += 1024) s_h[i] = 0;
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * kGroupItems;
    for (int j = threadIdx.x; j < kGroupItems && lo + j < n; j += 1024) atomicAdd(&s_h[(int)r.at(lo + j)], 1);
    __syncthreads();
    for (int i = threadIdx.x; i < R; i += 1024)
        if (s_h[i]) atomicAdd(counts + i, s_h[i]);
}

__global__ __launch_bounds__(1024) void k_rel_scatter_lds(IdSplit r, int64_t n, int R, const int* __restrict__ offsets,
                                                          const int* __restrict__ tile_off, int* __restrict__ cursor,
                                                          int* __restrict__ perm, int* __restrict__ tile_rel, int tile) {
    extern __shared__ int s_h[];
    int* s_cnt = s_h;          // [R] items of this block per relation
    int* s_base = s_h + R;     // [R] first position (inside the relation's range) this block reserved
    for (int i = threadIdx.x; i < R; i += 1024) s_cnt[i] = 0;
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * kGroupItems;
    int rel[kGroupItems / 1024], rank[kGroupItems / 1024];
#pragma unroll
    for (int q = 0; q < kGroupItems / 1024; ++q) {
        const int64_t i = lo + threadIdx.x + 1024 * q;
        rel[q] = i < n ? (int)r.at(i) : -1;
        rank[q] = rel[q] >= 0 ? atomicAdd(&s_cnt[rel[q]], 1) : 0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < R; i += 1024) s_base[i] = s_cnt[i] ? atomicAdd(cursor + i, s_cnt[i]) : 0;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kGroupItems / 1024; ++q) {
        if (rel[q] < 0) continue;
        const int local = s_base[rel[q]] + rank[q];
        perm[offsets[rel[q]] + local] = (int)(lo + threadIdx.x + 1024 * q);
        if (local % tile == 0) tile_rel[tile_off[rel[q]] + local / tile] = rel[q];
    }
}

int group_by_relation(const int64_t* r, int64_t n, int64_t R, const GroupWs& g, hipStream_t s) {
    return group_by_relation_split(id_whole(r, n), n, R, g, s);
}

// Small batches (the reference's own batch sizes: a few thousand triples): histogram, both scans and the scatter in ONE launch
// of one 1024-thread workgroup with the counters in LDS -- instead of a memset and three dependent launches whose atomics all
// hit the same few dozen global counters (measured 10.6 + 4.7 + 10.6 us at the C4 shape, plus the gaps between them).
constexpr int kSmallGroupMaxR = 4096;      // relations whose three int arrays fit the workgroup's LDS (48 KB)
constexpr int kSmallGroupMaxN = 16384;
constexpr int kStageMaxN = 4096;           // pairs of a batch whose grouped order also fits the LDS (deterministic form)
__global__ __launch_bounds__(1024) void k_rel_group_small(IdSplit r, int n, int R, int* __restrict__ offsets, int* __restrict__ tile_off,
                                                          int* __restrict__ perm, int* __restrict__ tile_rel,
                                                          float* __restrict__ zero_buf, int zero_n, int tile, PairGather pg) {
    for (int i = threadIdx.x; i < zero_n; i += 1024) zero_buf[i] = 0.f;   // (the scorer's accumulation target: saves a memset launch)
    extern __shared__ int s_grp[];
    int* s_cnt = s_grp;            // [R]   counts, then scatter cursors
    int* s_off = s_cnt + R;        // [R+1] first grouped position of each relation
    int* s_toff = s_off + R + 1;   // [R+1] first tile of each relation
    __shared__ int s_wave[2][16];
    __shared__ int s_carry[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < R; i += 1024) s_cnt[i] = 0;
    if (tid < 2) s_carry[tid] = 0;
    // sorted form (n <= kStageMaxN): this thread's pairs -- their entity ids and relation -- are requested now and live in registers
    constexpr int kPer = kStageMaxN / 1024;
    int4 ids[kPer];
    int rel_of[kPer];
    if (pg.sorted) {
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            ids[u] = make_int4(0, 0, 0, 0); rel_of[u] = 0;
            if (1024 * u < n) {   // (uniform)
                const int i = min(tid + 1024 * u, n - 1);
                ids[u] = make_int4((int)pg.ph[i], (int)pg.pt[i], (int)pg.nh[i], (int)pg.nt[i]);
                rel_of[u] = (int)r.at(i);
            }
        }
    }
    __syncthreads();
    if (pg.sorted) {
#pragma unroll
        for (int u = 0; u < kPer; ++u)
            if (tid + 1024 * u < n) atomicAdd(&s_cnt[rel_of[u]], 1);
    } else {
        for (int i = tid; i < n; i += 1024) atomicAdd(&s_cnt[(int)r.at(i)], 1);
    }
    __syncthreads();
    for (int base = 0; base < R; base += 1024) {   // exclusive scans of counts and of ceil(counts / TILE), 1024 relations per pass
        const int idx = base + tid;
        const int c = idx < R ? s_cnt[idx] : 0;
        const int tl = (c + tile - 1) / tile;
        int xa = c, xb = tl;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int ya = __shfl_up(xa, o, 64), yb = __shfl_up(xb, o, 64);
            if (lane >= o) { xa += ya; xb += yb; }
        }
        if (lane == 63) { s_wave[0][wave] = xa; s_wave[1][wave] = xb; }
        __syncthreads();
        if (tid < 2) {
            int acc = s_carry[tid];
            for (int w = 0; w < 16; ++w) { const int t = s_wave[tid][w]; s_wave[tid][w] = acc; acc += t; }
            s_carry[tid] = acc;
        }
        __syncthreads();
        if (idx < R) {
            const int oa = xa - c + s_wave[0][wave], ob = xb - tl + s_wave[1][wave];
            s_off[idx] = oa; s_toff[idx] = ob; offsets[idx] = oa; tile_off[idx] = ob;
        }
        __syncthreads();
    }
    if (tid == 0) { s_off[R] = s_carry[0]; s_toff[R] = s_carry[1]; offsets[R] = s_carry[0]; tile_off[R] = s_carry[1]; }
    for (int i = tid; i < R; i += 1024) s_cnt[i] = 0;   // now the scatter cursors
    __syncthreads();
    if (pg.sorted) {
        // ---- deterministic form (staged entity gradients): the grouped order goes through LDS, relations of 2 .. 64 pairs are sorted by
        // pair index (one wave per relation, bitonic over the lanes), then ids and descriptors are written from the final order.  (n <= kStageMaxN: s_perm fits behind the three relation arrays.)
        int* s_perm = s_toff + R + 1;      // [n] grouped position -> pair
        int* s_inv = s_perm + n;           // [n] pair -> grouped position
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const int i = tid + 1024 * u;
            if (i < n) s_perm[s_off[rel_of[u]] + atomicAdd(&s_cnt[rel_of[u]], 1)] = i;
        }
        __syncthreads();
        for (int rel = wave; rel < R; rel += 16) {
            const int lo = s_off[rel], c = s_off[rel + 1] - lo;
            if (c < 2 || c > 64) continue;            // (longer relations span several chunks: their share of G adds atomically anyway)
            int v = lane < c ? s_perm[lo + lane] : 0x7FFFFFFF;
#pragma unroll
            for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
                for (int j = k >> 1; j >= 1; j >>= 1) {
                    const int o = __shfl_xor(v, j, 64);
                    const bool up = (lane & k) == 0, lower = (lane & j) == 0;
                    v = (lower == up) ? min(v, o) : max(v, o);
                }
            if (lane < c) s_perm[lo + lane] = v;
        }
        __syncthreads();
        for (int g = tid; g < n; g += 1024) {
            const int i = s_perm[g];
            perm[g] = i;
            s_inv[i] = g;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const int i = tid + 1024 * u;
            if (i < n) pg.gids[s_inv[i]] = ids[u];
            // (the entity registrations of slot 4 g + j are made by the forward launch of the slab step, spread over its workgroups
            //  and under its operand loads: 4 n returning atomics from this one workgroup cost 20 us at n = 1 024)
        }
        for (int rel = tid; rel < R; rel += 1024) {   // tile descriptors (and tile_rel) of every relation, from the counts
            const int lo = s_off[rel], c = s_off[rel + 1] - lo, t0 = s_toff[rel], nt_ = s_toff[rel + 1] - t0;
            for (int t = 0; t < nt_; ++t) {
                tile_rel[t0 + t] = rel;
                pg.tdesc[t0 + t] = make_int4(rel, lo + t * tile, min(tile, c - t * tile), nt_);
            }
        }
        return;
    }
    for (int i = tid; i < n; i += 1024) {
        const int rel = (int)r.at(i);
        const int local = atomicAdd(&s_cnt[rel], 1);
        perm[s_off[rel] + local] = i;
        if (local % tile == 0) {
            tile_rel[s_toff[rel] + local / tile] = rel;
            // one descriptor per tile: its relation, first grouped position, pairs, and the number of tiles of its relation -- what
            // a (tile, slab) workgroup of kge_rescal_slab.hip would otherwise collect over three dependent round trips
            if (pg.tdesc) pg.tdesc[s_toff[rel] + local / tile] =
                make_int4(rel, s_off[rel] + local, min(tile, s_off[rel + 1] - s_off[rel] - local), s_toff[rel + 1] - s_toff[rel]);
        }
        // the pair's four entity ids in GROUPED order (one 16-byte load instead of perm -> four id loads)
        if (pg.gids) pg.gids[s_off[rel] + local] = make_int4((int)pg.ph[i], (int)pg.pt[i], (int)pg.nh[i], (int)pg.nt[i]);
    }
}

bool group_small_ok(int64_t n, int64_t R) { return n <= kSmallGroupMaxN && R <= kSmallGroupMaxR; }

int group_by_relation_split(IdSplit r, int64_t n, int64_t R, const GroupWs& g, hipStream_t s, float* zero_buf, int64_t zero_n, int tile,
                            const PairGather* pg) {
    if (n <= kSmallGroupMaxN && R <= kSmallGroupMaxR && zero_n <= kSmallGroupMaxN) {
        if (pg && pg->sorted && n > kStageMaxN) { set_error("grouping: staged entity gradients take at most %d pairs", kStageMaxN); return -1; }
        const size_t lds = (size_t)(3 * R + 2 + (pg && pg->sorted ? 2 * n : 0)) * sizeof(int);
        hipLaunchKernelGGL(k_rel_group_small, dim3(1), dim3(1024), lds, s, r, (int)n, (int)R, g.offsets,
                           g.tile_off, g.perm, g.tile_rel, zero_buf, (int)(zero_buf ? zero_n : 0), tile, pg ? *pg : PairGather{});
        return check_launch("k_rel_group_small");
    }
    if (pg) { set_error("grouping: the pair gather needs the one-launch grouping (n <= %d, R <= %d)", kSmallGroupMaxN, kSmallGroupMaxR); return -1; }
    if (zero_buf && zero_n > 0) {
        hipError_t ez = hipMemsetAsync(zero_buf, 0, (size_t)zero_n * sizeof(float), s);
        if (ez != hipSuccess) { set_error("grouping: memset: %s", hipGetErrorString(ez)); return -2; }
    }
    hipError_t e = hipMemsetAsync(g.counts, 0, (size_t)2 * (R + 1) * sizeof(int), s);  // counts + cursor
    if (e != hipSuccess) { set_error("rescal grouping memset: %s", hipGetErrorString(e)); return -2; }
    if (R <= kSmallGroupMaxR) {   // block-local histogram / ranks in LDS: one global atomic per (block, relation) instead of one per item
        const unsigned nbl = (unsigned)((n + kGroupItems - 1) / kGroupItems);
        hipLaunchKernelGGL(k_rel_hist_lds, dim3(nbl), dim3(1024), (size_t)R * sizeof(int), s, r, n, (int)R, g.counts);
        hipLaunchKernelGGL(k_rel_scan, dim3(1), dim3(256), 0, s, g.counts, (int)R, g.offsets, g.tile_off, tile);
        hipLaunchKernelGGL(k_rel_scatter_lds, dim3(nbl), dim3(1024), (size_t)2 * R * sizeof(int), s, r, n, (int)R, g.offsets, g.tile_off,
                           g.cursor, g.perm, g.tile_rel, tile);
        return check_launch("rescal grouping");
    }
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_rel_hist, dim3(nb), dim3(256), 0, s, r, n, g.counts);
    hipLaunchKernelGGL(k_rel_scan, dim3(1), dim3(256), 0, s, g.counts, (int)R, g.offsets, g.tile_off, tile);
    hipLaunchKernelGGL(k_rel_scatter, dim3(nb), dim3(256), 0, s, r, n, g.offsets, g.tile_off, g.cursor, g.perm, g.tile_rel, tile);
    return check_launch("rescal grouping");
}

// MODE 0: scores.  MODE 1: gradients (needs dscore).
template <int MODE>
__global__ __launch_bounds__(256) void k_rescal(const float* __restrict__ ent, const float* __restrict__ relm,
                                                float* __restrict__ g_ent, float* __restrict__ g_rel,
                                                IdSplit h, IdSplit t,
                                                const int* __restrict__ offsets, const int* __restrict__ tile_off,
                                                const int* __restrict__ tile_rel, const int* __restrict__ perm, int R, int k,
                                                const float* __restrict__ dscore, float* __restrict__ scores) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int rel, tin;
    if (!locate_tile(tile_off, tile_rel, R, blockIdx.x, rel, tin)) return;
    const int S = (k + 1) | 1;                 // odd LDS row stride: conflict-free column reads
    float* sT = smem;                          // [32][S]
    float* sH = sT + TILE * S;                 // [32][S]
    float* sDs = sH + TILE * S;                // [32]
    float* sSc = sDs + TILE;                   // [32]
    int* sRow = (int*)(sSc + TILE);            // [32] original row index (-1 = padding)
    long long* sHid = (long long*)(sRow + TILE + (TILE & 1));  // [32] head ids, [32] tail ids
    long long* sTid = sHid + TILE;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g0 = offsets[rel] + tin * TILE;
    const int cnt = min(TILE, offsets[rel + 1] - g0);
    if (threadIdx.x < TILE) {
        const int row = threadIdx.x < cnt ? perm[g0 + threadIdx.x] : -1;
        sRow[threadIdx.x] = row;
        sHid[threadIdx.x] = row >= 0 ? h.at(row) : 0;
        sTid[threadIdx.x] = row >= 0 ? t.at(row) : 0;
        sDs[threadIdx.x] = (MODE == 1 && row >= 0) ? dscore[row] : 0.f;
        sSc[threadIdx.x] = 0.f;
    }
    __syncthreads();
    if ((k & 3) == 0) {
        // row gathers into LDS, 16 bytes per load, ALL of a thread's loads issued before the first LDS store: one memory round trip
        // for the whole tile instead of one per 256 elements (the staging loop used to be most of this kernel at small batches)
        const int nv = k >> 2;                       // float4 per row
        constexpr int kMaxPer = 16;                  // TILE * nv / 256 float4 per thread and matrix (k <= 512)
        float4 rt[kMaxPer], rh[kMaxPer];
#pragma unroll
        for (int j = 0; j < kMaxPer; ++j) {
            const int idx = threadIdx.x + 256 * j;
            const int i = idx / nv, c = idx - i * nv;
            const bool ok = idx < TILE * nv && i < cnt;
            rt[j] = ok ? reinterpret_cast<const float4*>(ent + sTid[i] * k)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            rh[j] = ok ? reinterpret_cast<const float4*>(ent + sHid[i] * k)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < kMaxPer; ++j) {
            const int idx = threadIdx.x + 256 * j;
            if (idx < TILE * nv) {
                const int i = idx / nv, c = (idx - i * nv) * 4;
                float* dt = sT + i * S + c;
                float* dh = sH + i * S + c;
                dt[0] = rt[j].x; dt[1] = rt[j].y; dt[2] = rt[j].z; dt[3] = rt[j].w;
                dh[0] = rh[j].x; dh[1] = rh[j].y; dh[2] = rh[j].z; dh[3] = rh[j].w;
            }
        }
    } else {
        for (int idx = threadIdx.x; idx < TILE * k; idx += 256) {  // coalesced row gathers into LDS
            const int i = idx / k, c = idx - i * k;
            const bool ok = i < cnt;
            sT[i * S + c] = ok ? ent[sTid[i] * k + c] : 0.f;
            sH[i * S + c] = ok ? ent[sHid[i] * k + c] : 0.f;
        }
    }
    __syncthreads();
    const float* M = relm + (int64_t)rel * k * k;
    const int ntile = (k + 31) / 32;
    const int li = lane & 31, lk = lane >> 5;

    // Work units of this tile, one (or four) per workgroup along gridDim.y:
    //   y in [0, ntile)              U = T M^T  column tile   (backward: grad_h)      } the four waves of the workgroup SPLIT K
    //   y in [ntile, 2 ntile)        V = H M    column tile   (forward score / grad_t) } (k/4 each) and add their accumulators
    // (the relation-matrix gradient G = (ds*H)^T T has its own relation-owner kernel, k_rescal_gm)
    // At the reference's batch sizes a tile is a handful of MFMA steps behind a chain of dependent operand loads: the chain, not
    // the matrix pipe, is the cost, so it is cut four ways (37 -> ~12 us forward, 50 -> ~20 us backward at the C4 shape).
    float* sAcc = (float*)(sTid + TILE);       // [4][16][64] accumulators of the K split
    const int kq = ((k + 7) / 8) * 2;          // K span of one wave (even: an MFMA step takes two k)
    const int k_lo = wave * kq, k_hi = min(k, k_lo + kq);
    const int y = blockIdx.y;
    const bool unit_u = MODE == 1 && y < ntile;
    const bool unit_v = MODE == 0 ? y < ntile : (y >= ntile && y < 2 * ntile);
    if (unit_u || unit_v) {
        const int ct = unit_u ? y : (MODE == 0 ? y : y - ntile);
        const int col = ct * 32 + li;      // U: a (row of M);  V: b (column of M)
        const float* sX = unit_u ? sT : sH;
        f32x16 acc = {0};
        for (int k0 = k_lo; k0 < k_hi; k0 += 16) {  // fixed-trip inner loop: 8 operand loads in flight
            float av[8], bv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int kk = k0 + 2 * q + lk;
                const bool on = kk < k_hi;
                av[q] = on ? sX[li * S + kk] : 0.f;
                // U[i][a] = sum_b T[i][b] M[a][b] (lane a reads M[a][b]: a 4-byte gather over 32 rows); V[i][b] = sum_a H[i][a] M[a][b]
                bv[q] = (on && col < k) ? (unit_u ? M[(int64_t)col * k + kk] : M[(int64_t)kk * k + col]) : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc, 0, 0, 0);
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) sAcc[(wave * 16 + reg) * 64 + lane] = acc[reg];
        __syncthreads();
        if (wave == 0 && col < k) {   // the four K shares, added in wave order
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const float v = ((sAcc[(0 * 16 + reg) * 64 + lane] + sAcc[(1 * 16 + reg) * 64 + lane]) + sAcc[(2 * 16 + reg) * 64 + lane]) +
                                sAcc[(3 * 16 + reg) * 64 + lane];
                const int i = (reg & 3) + 8 * (reg >> 2) + 4 * lk;
                if (MODE == 0) {
                    atomicAdd(&sSc[i], sT[i * S + col] * v);          // LDS atomic: score_i += V[i][b] t_i[b]
                } else if (i < cnt && sDs[i] != 0.f) {
                    unsafeAtomicAdd(g_ent + (unit_u ? sHid[i] : sTid[i]) * k + col, -sDs[i] * v);   // grad_h = -ds U ; grad_t = -ds V
                }
            }
        }
    }
    if (MODE == 0) {
        __syncthreads();
        // partial score of this block's column tiles; `scores` is zero-filled by the launcher
        if (threadIdx.x < cnt) unsafeAtomicAdd(scores + sRow[threadIdx.x], -sSc[threadIdx.x]);
    }
}

// grad_M[rel] -= sum_i ds_i h_i t_i^T, relation-owner form: the workgroup of a relation's FIRST tile of every run of kGmRun tiles
// walks that run's triples (32 per round trip: both operand rows of 16 MFMA steps in flight) and owns the 32x32 output tiles its
// four waves hold, so a relation with at most kGmRun tiles (every relation at the reference's batch sizes) is accumulated with
// plain read-modify-writes in a fixed order -- no float atomics, no 40 000 atomics per 32 triples.  Longer relations (skewed
// large batches) split into runs that add atomically.
constexpr int kGmRun = 8;
__global__ __launch_bounds__(256) void k_rescal_gm(const float* __restrict__ ent, float* __restrict__ g_rel, IdSplit h, IdSplit t,
                                                   const int* __restrict__ offsets, const int* __restrict__ tile_off,
                                                   const int* __restrict__ tile_rel, const int* __restrict__ perm, int R, int k,
                                                   const float* __restrict__ dscore) {
    __shared__ float sDs[TILE];
    __shared__ long long sHid[TILE], sTid[TILE];
    int rel, tin;
    if (!locate_tile(tile_off, tile_rel, R, blockIdx.x, rel, tin)) return;
    if (tin % kGmRun) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int ntile = (k + 31) / 32;
    const int tl = blockIdx.y * 4 + wave;
    const bool live = tl < ntile * ntile;
    const int at = live ? tl / ntile : 0, bt = live ? tl - at * ntile : 0;
    const int a_in = at * 32 + li, b_in = bt * 32 + li;
    const int r0 = offsets[rel], r1 = offsets[rel + 1];
    const int g_lo = r0 + tin * TILE, g_hi = min(r1, g_lo + kGmRun * TILE);
    const bool shared_rel = (r1 - r0) > kGmRun * TILE;     // other runs of this relation add to the same matrix
    f32x16 acc = {0};
    for (int g0 = g_lo; g0 < g_hi; g0 += TILE) {
        __syncthreads();
        if (threadIdx.x < TILE) {
            const int row = g0 + (int)threadIdx.x < g_hi ? perm[g0 + threadIdx.x] : -1;
            sHid[threadIdx.x] = row >= 0 ? h.at(row) : 0;
            sTid[threadIdx.x] = row >= 0 ? t.at(row) : 0;
            sDs[threadIdx.x] = row >= 0 ? dscore[row] : 0.f;
        }
        __syncthreads();
        if (!live) continue;
        float av[TILE / 2], bv[TILE / 2];
#pragma unroll
        for (int q = 0; q < TILE / 2; ++q) {
            const int i = 2 * q + lk;
            const float ds = sDs[i];
            av[q] = (ds != 0.f && a_in < k) ? ent[sHid[i] * k + a_in] : 0.f;
            bv[q] = (ds != 0.f && b_in < k) ? ent[sTid[i] * k + b_in] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < TILE / 2; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sDs[2 * q + lk] * av[q], bv[q], acc, 0, 0, 0);
    }
    if (!live || b_in >= k) return;
    float* gM = g_rel + (int64_t)rel * k * k;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int a = at * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lk;
        if (a < k && acc[reg] != 0.f) {
            float* p = gM + (int64_t)a * k + b_in;
            if (shared_rel) unsafeAtomicAdd(p, -acc[reg]); else *p -= acc[reg];
        }
    }
}

static size_t rescal_lds_bytes(int k) {
    const int S = (k + 1) | 1;
    return (size_t)(2 * TILE * S + 2 * TILE) * sizeof(float) + (size_t)(TILE + (TILE & 1)) * sizeof(int) +
           (size_t)2 * TILE * sizeof(long long) + (size_t)4 * 16 * 64 * sizeof(float);   // + the K-split accumulators
}



size_t dense_workspace_bytes(const kge_model_desc* m, int64_t n) {
    if (m->model == KGE_RESCAL) return group_ws_bytes(m->tot_relation, n);
    if (m->model == KGE_NTN) return ntn_workspace_bytes(m, n);
    if (m->model == KGE_TRANSR) return transr_workspace_bytes(m, n);
    return 0;
}

static int rescal_run(int mode, const kge_model_desc* m, IdSplit h, IdSplit r, IdSplit t, int64_t n,
                      const float* dscore, float* scores, void* ws, size_t ws_bytes, bool grouped, hipStream_t s) {
    const int k = m->dim;
    const int64_t R = m->tot_relation;
    if (k > 512) { set_error("RESCAL: hidden size %d exceeds the LDS-resident tile kernel (max 512)", k); return -1; }
    if (n >= (1ll << 31)) { set_error("RESCAL: batch too large"); return -1; }
    if (!ws || ws_bytes < group_ws_bytes(R, n)) {
        set_error("RESCAL needs a workspace of %zu bytes (kge_workspace_bytes)", group_ws_bytes(R, n));
        return -1;
    }
    const GroupWs g = carve_group_ws(ws, R, n);
    bool zeroed = false;
    if (!grouped) {  // the fused train step's backward reuses the grouping its forward left in this workspace
        int rc = group_by_relation_split(r, n, R, g, s, mode == 0 ? scores : nullptr, mode == 0 ? n : 0);   // (+ clears the score buffer)
        if (rc) return rc;
        zeroed = mode == 0;
    }
    const unsigned max_tiles = (unsigned)group_max_tiles(R, n);  // surplus blocks exit
    const size_t lds = rescal_lds_bytes(k);
    const int ntile = (k + 31) / 32;
    if (mode == 0) {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute((const void*)k_rescal<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (!zeroed) {
            hipError_t e = hipMemsetAsync(scores, 0, (size_t)n * sizeof(float), s);
            if (e != hipSuccess) { set_error("rescal: memset: %s", hipGetErrorString(e)); return -2; }
        }
        hipLaunchKernelGGL(k_rescal<0>, dim3(max_tiles, (unsigned)ntile), dim3(256), lds, s, m->tables[0], m->tables[1], nullptr, nullptr, h, t,
                           g.offsets, g.tile_off, g.tile_rel, g.perm, (int)R, k, nullptr, scores);
    } else {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute((const void*)k_rescal<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_rescal_gm, dim3(max_tiles, (unsigned)((ntile * ntile + 3) / 4)), dim3(256), 0, s, m->tables[0], m->grads[1], h, t,
                           g.offsets, g.tile_off, g.tile_rel, g.perm, (int)R, k, dscore);
        hipLaunchKernelGGL(k_rescal<1>, dim3(max_tiles, (unsigned)(2 * ntile)), dim3(256), lds, s, m->tables[0], m->tables[1], m->grads[0],
                           m->grads[1], h, t, g.offsets, g.tile_off, g.tile_rel, g.perm, (int)R, k, dscore, nullptr);
    }
    return check_launch("k_rescal");
}

int launch_rescal_forward(const kge_model_desc* m, const int64_t* h, const int64_t* r, const int64_t* t, int64_t n,
                          float* scores, void* ws, size_t ws_bytes, hipStream_t s) {
    return rescal_run(0, m, id_whole(h, n), id_whole(r, n), id_whole(t, n), n, nullptr, scores, ws, ws_bytes, false, s);
}
int launch_rescal_backward(const kge_model_desc* m, const int64_t* h, const int64_t* r, const int64_t* t, int64_t n,
                           const float* dscore, void* ws, size_t ws_bytes, bool grouped, hipStream_t s) {
    return rescal_run(1, m, id_whole(h, n), id_whole(r, n), id_whole(t, n), n, dscore, nullptr, ws, ws_bytes, grouped, s);
}

// The fused pairwise step scores / back-propagates positives and negatives as ONE batch of 2n triples: one grouping pass and
// one launch per direction instead of two (the kernels are latency-sized at the reference's batch sizes, so a launch costs
// the same with twice the tiles).  scores / dscore: [2n], positives first.
int launch_rescal_pair_forward(const kge_model_desc* m, const int64_t* ph, const int64_t* pr, const int64_t* pt,
                               const int64_t* nh, const int64_t* nr, const int64_t* nt, int64_t n, float* scores2, void* ws,
                               size_t ws_bytes, hipStream_t s) {
    return rescal_run(0, m, IdSplit{ph, nh, n}, IdSplit{pr, nr, n}, IdSplit{pt, nt, n}, 2 * n, nullptr, scores2, ws, ws_bytes, false, s);
}
int launch_rescal_pair_backward(const kge_model_desc* m, const int64_t* ph, const int64_t* pr, const int64_t* pt,
                                const int64_t* nh, const int64_t* nr, const int64_t* nt, int64_t n, const float* dscore2,
                                void* ws, size_t ws_bytes, hipStream_t s) {
    return rescal_run(1, m, IdSplit{ph, nh, n}, IdSplit{pr, nr, n}, IdSplit{pt, nt, n}, 2 * n, dscore2, nullptr, ws, ws_bytes, true, s);
}

// ---- W <- W / ||W_row||_2 in place (plain division, no eps: pairwise.py:862-865)
__global__ __launch_bounds__(256) void k_row_normalize(float* __restrict__ w, int64_t rows, int64_t dim) {
    const int lane = threadIdx.x & 63;
    // two rows per wave pass: both rows' loads are in flight before the first reduction; rows up to 256 floats stay in
    // registers between the norm and the scaling (one HBM read, one write)
    for (int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2; row < rows; row += (int64_t)gridDim.x * 8) {
        float* p0 = w + row * dim;
        float* p1 = p0 + dim;
        const bool has1 = row + 1 < rows;
        if (dim <= 256) {
            float a[4], b[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int64_t e = lane + 64 * c;
                a[c] = e < dim ? p0[e] : 0.f;
                b[c] = (has1 && e < dim) ? p1[e] : 0.f;
            }
            float na = 0.f, nb = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) { na = fmaf(a[c], a[c], na); nb = fmaf(b[c], b[c], nb); }
            na = sqrtf(wave_sum(na)); nb = sqrtf(wave_sum(nb));
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int64_t e = lane + 64 * c;
                if (e < dim) { p0[e] = a[c] / na; if (has1) p1[e] = b[c] / nb; }
            }
        } else {
            for (int rr = 0; rr < (has1 ? 2 : 1); ++rr) {
                float* p = rr ? p1 : p0;
                float n2 = 0.f;
                for (int64_t c = lane; c < dim; c += 64) n2 = fmaf(p[c], p[c], n2);
                const float nrm = sqrtf(wave_sum(n2));
                for (int64_t c = lane; c < dim; c += 64) p[c] = p[c] / nrm;
            }
        }
    }
}

// long rows (the k*k relation matrices): one 1024-thread workgroup per row
__global__ __launch_bounds__(1024) void k_row_normalize_wide(float* __restrict__ w, int64_t rows, int64_t dim) {
    __shared__ float part[16];
    __shared__ float s_nrm;
    float* p = w + (int64_t)blockIdx.x * dim;
    float n2 = 0.f;
    for (int64_t c = threadIdx.x; c < dim; c += 1024) n2 = fmaf(p[c], p[c], n2);
    n2 = wave_sum(n2);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = n2;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < 16; ++i) t += part[i];
        s_nrm = sqrtf(t);
    }
    __syncthreads();
    const float nrm = s_nrm;
    for (int64_t c = threadIdx.x; c < dim; c += 1024) p[c] = p[c] / nrm;
}

// very long rows, few of them (the relation matrices of a graph with a handful of relations: 37 rows of 40 000 floats):
// one workgroup per row leaves the chip idle, so a row is cut into 4096-float chunks -- pass 1 writes each chunk's sum of
// squares, pass 2 adds a row's partials in chunk order (deterministic) and scales its chunk.  `part`: [rows][nchunk].
constexpr int kNormChunk = 4096;
__global__ __launch_bounds__(256) void k_row_sumsq_chunks(const float* __restrict__ w, int64_t dim, int nchunk, float* __restrict__ part) {
    __shared__ float sw[4];
    const int64_t row = blockIdx.x / nchunk;
    const int ch = blockIdx.x % nchunk;
    const float* p = w + row * dim;
    const int64_t lo = (int64_t)ch * kNormChunk, hi = min(dim, lo + kNormChunk);
    float n2 = 0.f;
    for (int64_t c = lo + threadIdx.x; c < hi; c += 256) n2 = fmaf(p[c], p[c], n2);
    n2 = wave_sum(n2);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = n2;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}
__global__ __launch_bounds__(256) void k_row_scale_chunks(float* __restrict__ w, int64_t dim, int nchunk, const float* __restrict__ part) {
    const int64_t row = blockIdx.x / nchunk;
    const int ch = blockIdx.x % nchunk;
    float t = 0.f;
    for (int i = 0; i < nchunk; ++i) t += part[row * nchunk + i];
    const float nrm = sqrtf(t);
    float* p = w + row * dim;
    const int64_t lo = (int64_t)ch * kNormChunk, hi = min(dim, lo + kNormChunk);
    for (int64_t c = lo + threadIdx.x; c < hi; c += 256) p[c] = p[c] / nrm;
}

static void normalize_rows(float* w, int64_t rows, int64_t dim, hipStream_t s, float* scratch = nullptr, size_t scratch_floats = 0) {
    const int nchunk = (int)((dim + kNormChunk - 1) / kNormChunk);
    if (dim >= 4 * kNormChunk && rows < 1024 && scratch && scratch_floats >= (size_t)rows * nchunk) {
        hipLaunchKernelGGL(k_row_sumsq_chunks, dim3((unsigned)(rows * nchunk)), dim3(256), 0, s, w, dim, nchunk, scratch);
        hipLaunchKernelGGL(k_row_scale_chunks, dim3((unsigned)(rows * nchunk)), dim3(256), 0, s, w, dim, nchunk, scratch);
        return;
    }
    if (dim >= 2048)
        hipLaunchKernelGGL(k_row_normalize_wide, dim3((unsigned)rows), dim3(1024), 0, s, w, rows, dim);
    else
        hipLaunchKernelGGL(k_row_normalize, dim3((unsigned)min((int64_t)16384, (rows + 7) / 8)), dim3(256), 0, s, w, rows, dim);
}

int launch_rescal_normalize(float* ent, int64_t E, float* rel, int64_t R, int k, float* scratch, size_t scratch_floats,
                            hipStream_t s) {
    if (ent) normalize_rows(ent, E, (int64_t)k, s);   // (ent == NULL: the entity rows were renormalised by kge_optimizer_step_rows)
    normalize_rows(rel, R, (int64_t)k * k, s, scratch, scratch_floats);
    return check_launch("k_row_normalize");
}

// ---------------------------------------------------------------------------------------------------------------------------
// The pairwise RESCAL step (scores of both sides, margin hinge, all three gradients) as ONE launch after the grouping.
// A negative shares its positive's relation (the sampler corrupts heads and tails only), so PAIRS are grouped by relation and
// a workgroup owns (relation, 16 pairs) = 32 triples: rows 0..15 the positives, 16..31 their negatives.  With both sides of
// every pair in one workgroup the hinge is local, and the forward / coefficient / backward launches -- each a chain of
// dependent loads (tile lookup, ids, rows, matrix operands) that costs more than its arithmetic at the reference's batch sizes
// -- become one chain.  16 waves: (column tile, K half) units for V = H M and U = T M^T, every operand of a unit's MFMA steps
// loaded before the first step (one round trip), the two K halves added through LDS in a fixed order; G = (ds H)^T T from LDS.
// Entity gradients and the relation-matrix gradient leave through float atomics as in k_rescal.  k even (rows move as float4 when
// k % 4 == 0, as float2 otherwise: the reference's default k = 50), k <= 256.
// VK consecutive floats of a row as one load (VK = 4: hidden size % 4 == 0; VK = 2: any even hidden size, e.g. the reference's k = 50)
template <int VK>
__device__ __forceinline__ void ldv(float (&o)[VK], const float* __restrict__ p) {
    if constexpr (VK == 4) { const float4 q = *reinterpret_cast<const float4*>(p); o[0] = q.x; o[1] = q.y; o[2] = q.z; o[3] = q.w; }
    else { const float2 q = *reinterpret_cast<const float2*>(p); o[0] = q.x; o[1] = q.y; }
}
constexpr int kPairSteps = 52;     // V: MFMA steps (two k each) of operand loads in flight per wave
constexpr int kPairUK = 104;       // U: k covered by the operand loads one wave keeps in flight (13 float4 or 26 float2 per lane)

template <int VK>
__global__ __launch_bounds__(1024) void k_rescal_pair(const float* __restrict__ ent, const float* __restrict__ relm,
                                                      float* __restrict__ g_ent, float* __restrict__ g_rel,
                                                      const int64_t* __restrict__ ph, const int64_t* __restrict__ pt,
                                                      const int64_t* __restrict__ nh, const int64_t* __restrict__ nt,
                                                      const int* __restrict__ offsets, const int* __restrict__ tile_off,
                                                      const int* __restrict__ tile_rel, const int* __restrict__ perm, int R, int k,
                                                      float margin, float* __restrict__ loss, unsigned* __restrict__ touched,
                                                      float* __restrict__ ds_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int rel, tin;
    if (!locate_tile(tile_off, tile_rel, R, blockIdx.x, rel, tin)) return;
    const int S = (k + 1) | 1;                 // odd LDS row stride: conflict-free column reads
    float* sT = smem;                          // [32][S]
    float* sH = sT + TILE * S;                 // [32][S]
    float* sRed = sH + TILE * S;               // [8][16][64] accumulators of the second K half
    float* sPs = sRed + 8 * 16 * 64;           // [8][32]     score partials per column tile
    float* sDs = sPs + 8 * TILE;               // [32]        dL/denergy
    long long* sHid = (long long*)(sDs + TILE);
    long long* sTid = sHid + TILE;
    int* sAny = (int*)(sTid + TILE);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int g0 = offsets[rel] + tin * kPairTile;
    const int cnt = min(kPairTile, offsets[rel + 1] - g0);
    int my_pair = -1;
    if (threadIdx.x < TILE) {
        const int p = threadIdx.x & (kPairTile - 1);
        const bool neg = threadIdx.x >= kPairTile;
        const int pair = p < cnt ? perm[g0 + p] : -1;
        my_pair = pair;
        sHid[threadIdx.x] = pair >= 0 ? (neg ? nh[pair] : ph[pair]) : 0;
        sTid[threadIdx.x] = pair >= 0 ? (neg ? nt[pair] : pt[pair]) : 0;
        sDs[threadIdx.x] = 0.f;
    }
    if (threadIdx.x < 8 * TILE) sPs[threadIdx.x] = 0.f;
    __syncthreads();
    {   // row gathers into LDS: VK floats per load, every load of a thread issued before its first LDS store
        const int nv = k / VK;
        constexpr int kMaxPer = 8 / VK;              // TILE * nv / 1024 loads per thread and matrix (k <= 256)
        float rt[kMaxPer][VK], rh[kMaxPer][VK];
#pragma unroll
        for (int j = 0; j < kMaxPer; ++j) {
            const int idx = threadIdx.x + 1024 * j;
            const int i = idx / nv, c = idx - i * nv;
            const bool ok = idx < TILE * nv && (i & (kPairTile - 1)) < cnt;
#pragma unroll
            for (int q = 0; q < VK; ++q) { rt[j][q] = 0.f; rh[j][q] = 0.f; }
            if (ok) { ldv<VK>(rt[j], ent + sTid[i] * k + c * VK); ldv<VK>(rh[j], ent + sHid[i] * k + c * VK); }
        }
#pragma unroll
        for (int j = 0; j < kMaxPer; ++j) {
            const int idx = threadIdx.x + 1024 * j;
            if (idx < TILE * nv) {
                const int i = idx / nv, c = (idx - i * nv) * VK;
#pragma unroll
                for (int q = 0; q < VK; ++q) { sT[i * S + c + q] = rt[j][q]; sH[i * S + c + q] = rh[j][q]; }
            }
        }
    }
    __syncthreads();
    const float* M = relm + (int64_t)rel * k * k;
    const int ntile = (k + 31) / 32;           // <= 8
    const int khalf = ((k + 4 * VK - 1) / (4 * VK)) * (2 * VK);     // K span of one wave: a multiple of 2 VK
    const int u = wave >> 1, ks = wave & 1;    // unit: column tile u, K half ks
    const bool live = u < ntile;
    const int k_lo = ks * khalf, k_hi = min(k, k_lo + khalf);
    const int col = u * 32 + li;               // V: b (column of M);  U: a (row of M)

    // ---- V[i][b] = sum_a H[i][a] M[a][b]
    f32x16 acc = {0};
    if (live) {
        for (int kc = k_lo; kc < k_hi; kc += 2 * kPairSteps) {
            float bv[kPairSteps];
#pragma unroll
            for (int q = 0; q < kPairSteps; ++q) {
                const int kk = kc + 2 * q + lk;
                bv[q] = (kk < k_hi && col < k) ? M[(int64_t)kk * k + col] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < kPairSteps; ++q) {
                const int kk = kc + 2 * q + lk;
                const float av = kk < k_hi ? sH[li * S + kk] : 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[q], acc, 0, 0, 0);
            }
        }
        if (ks == 1) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) sRed[(u * 16 + reg) * 64 + lane] = acc[reg];
        }
    }
    __syncthreads();
    if (live && ks == 0) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            acc[reg] += sRed[(u * 16 + reg) * 64 + lane];
            const int i = (reg & 3) + 8 * (reg >> 2) + 4 * lk;
            float p = col < k ? acc[reg] * sT[i * S + col] : 0.f;
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) p += __shfl_xor(p, o, 64);      // over the 32 columns of this half-wave
            if (li == 0) sPs[u * TILE + i] = p;
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {   // margin hinge of the tile's pairs (as k_hinge_coeffs: energies = -h^T M t)
        const int p = threadIdx.x;
        float v = 0.f, c = 0.f;
        if (p < cnt) {
            float sp = 0.f, sn = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) { sp += sPs[q * TILE + p]; sn += sPs[q * TILE + kPairTile + p]; }
            v = (-sp) + margin - (-sn);
            c = v > 0.f ? 1.f : (v == 0.f ? 0.5f : 0.f);
            sDs[p] = c; sDs[kPairTile + p] = -c;
            if (ds_out) ds_out[my_pair] = c;     // large batches: the relation-matrix gradient is a second, relation-owner launch
        }
        const float tot = wave_sum(fmaxf(v, 0.f));
        const unsigned long long any = __ballot(c != 0.f);
        if (p == 0) {
            if (tot != 0.f) unsafeAtomicAdd(loss + (blockIdx.x % kLossSlots) * kLossStride, tot);
            *sAny = any != 0ull;
        }
    }
    __syncthreads();
    if (!*sAny) return;        // every pair of the tile inside the margin: no gradient
    if (touched && threadIdx.x < TILE && sDs[threadIdx.x] != 0.f) {   // entity rows this tile writes a gradient into
        const long long a = sHid[threadIdx.x], b = sTid[threadIdx.x];
        atomicOr(touched + (a >> 5), 1u << (a & 31));
        atomicOr(touched + (b >> 5), 1u << (b & 31));
    }

    // ---- grad_t = -ds V (from the registers of the first K half's waves)
    // (rows i and i + 16 are the two sides of one pair: register reg and reg + 8 of the same lane; the side the sampler did not
    // corrupt is the SAME entity row, whose two contributions leave as one atomic: a quarter of the float atomics of a batch)
    if (live && ks == 0 && col < k) {
#pragma unroll
        for (int reg = 0; reg < 8; ++reg) {
            const int i = (reg & 3) + 8 * (reg >> 2) + 4 * lk;
            const float ds = sDs[i];
            if (ds != 0.f) {
                const long long ta = sTid[i], tb = sTid[i + kPairTile];
                if (ta == tb) {
                    unsafeAtomicAdd(g_ent + ta * k + col, -ds * (acc[reg] - acc[reg + 8]));
                } else {
                    unsafeAtomicAdd(g_ent + ta * k + col, -ds * acc[reg]);
                    unsafeAtomicAdd(g_ent + tb * k + col, ds * acc[reg + 8]);
                }
            }
        }
    }
    // ---- U[i][a] = sum_b T[i][b] M[a][b]: lane a reads its row of M VK floats at a time; the VK values of a load feed VK MFMA
    // steps (lane half 0 carries k = kb .. kb+VK-1, half 1 the next VK: the k order inside a step is free)
    f32x16 ua = {0};
    if (live) {
        constexpr int kQ = kPairUK / (2 * VK);       // loads per lane and chunk
        for (int kc = k_lo; kc < k_hi; kc += kPairUK) {
            float mv[kQ][VK];
#pragma unroll
            for (int j = 0; j < kQ; ++j) {
                const int kb = kc + 2 * VK * j + VK * lk;
#pragma unroll
                for (int q = 0; q < VK; ++q) mv[j][q] = 0.f;
                if (kb < k_hi && col < k) ldv<VK>(mv[j], M + (int64_t)col * k + kb);
            }
#pragma unroll
            for (int j = 0; j < kQ; ++j) {
                const int kb = kc + 2 * VK * j + VK * lk;
                const bool on = kb < k_hi;
                const float* tr = sT + li * S + kb;
#pragma unroll
                for (int q = 0; q < VK; ++q) ua = __builtin_amdgcn_mfma_f32_32x32x2f32(on ? tr[q] : 0.f, mv[j][q], ua, 0, 0, 0);
            }
        }
        if (ks == 1) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) sRed[(u * 16 + reg) * 64 + lane] = ua[reg];
        }
    }
    __syncthreads();
    if (live && ks == 0 && col < k) {
#pragma unroll
        for (int reg = 0; reg < 8; ++reg) {
            const int i = (reg & 3) + 8 * (reg >> 2) + 4 * lk;
            const float ds = sDs[i];
            const float va = ua[reg] + sRed[(u * 16 + reg) * 64 + lane];
            const float vb = ua[reg + 8] + sRed[(u * 16 + reg + 8) * 64 + lane];
            if (ds != 0.f) {     // grad_h = -ds U
                const long long ha = sHid[i], hb = sHid[i + kPairTile];
                if (ha == hb) {
                    unsafeAtomicAdd(g_ent + ha * k + col, -ds * (va - vb));
                } else {
                    unsafeAtomicAdd(g_ent + ha * k + col, -ds * va);
                    unsafeAtomicAdd(g_ent + hb * k + col, ds * vb);
                }
            }
        }
    }
    // ---- G[a][b] = sum_i ds_i H[i][a] T[i][b] ;  grad_M = -G
    if (ds_out) return;
    float* gM = g_rel + (int64_t)rel * k * k;
    for (int tl = wave; tl < ntile * ntile; tl += 16) {
        const int at = tl / ntile, bt = tl - at * ntile;
        const int a_in = at * 32 + li, b_in = bt * 32 + li;
        f32x16 ga = {0};
#pragma unroll
        for (int kk = 0; kk < TILE; kk += 2) {
            const int i = kk + lk;
            const float av = a_in < k ? sDs[i] * sH[i * S + a_in] : 0.f;
            const float bv = b_in < k ? sT[i * S + b_in] : 0.f;
            ga = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, ga, 0, 0, 0);
        }
        if (b_in < k) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int a = at * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lk;
                if (a < k && ga[reg] != 0.f) unsafeAtomicAdd(gM + (int64_t)a * k + b_in, -ga[reg]);
            }
        }
    }
}

// Large batches: a (relation, 16 pairs) tile adding its 32-triple share of G with k^2 float atomics makes the atomics the whole
// cost (2085 tiles x 40 000 at B = 32768 on YAGO3-10's 37 relations: 485 us of a 568 us launch).  Then k_rescal_pair only
// leaves dL/denergy of every pair behind and this kernel accumulates G over runs of kPairGmRun tiles: the workgroup of a run's
// first tile stages 32 triples at a time in LDS (rows of H pre-multiplied by ds) and its sixteen waves hold ALL ceil(k/32)^2 output
// tiles in registers across the run, so the rows are read once per run and a relation of at most one run is written with k^2
// plain read-modify-writes; longer relations add atomically, once per run.
constexpr int kPairGmRun = 8;
template <int VK>
__global__ __launch_bounds__(1024) void k_rescal_pair_gm(const float* __restrict__ ent, float* __restrict__ g_rel,
                                                        const int64_t* __restrict__ ph, const int64_t* __restrict__ pt,
                                                        const int64_t* __restrict__ nh, const int64_t* __restrict__ nt,
                                                        const int* __restrict__ offsets, const int* __restrict__ tile_off,
                                                        const int* __restrict__ tile_rel, const int* __restrict__ perm, int R, int k,
                                                        const float* __restrict__ ds) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int rel, tin;
    if (!locate_tile(tile_off, tile_rel, R, blockIdx.x, rel, tin)) return;
    if (tin % kPairGmRun) return;
    const int S = (k + 1) | 1;
    float* sT = smem;                          // [32][S]
    float* sH = sT + TILE * S;                 // [32][S]  ds_i * h_i
    float* sDs = sH + TILE * S;                // [32]
    long long* sHid = (long long*)(sDs + TILE);
    long long* sTid = sHid + TILE;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: the per-tile branches below stay branches)
    const int li = lane & 31, lk = lane >> 5;
    const int ntile = (k + 31) / 32, nt2 = ntile * ntile;      // <= 64 tiles: four per wave
    const int r0 = offsets[rel], r1 = offsets[rel + 1];
    const int g_lo = r0 + tin * kPairTile, g_hi = min(r1, g_lo + kPairGmRun * kPairTile);
    const bool shared_rel = (r1 - r0) > kPairGmRun * kPairTile;
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x16{0};
    const int nv = k / VK;
    for (int g0 = g_lo; g0 < g_hi; g0 += kPairTile) {
        __syncthreads();
        if (threadIdx.x < TILE) {
            const int p = threadIdx.x & (kPairTile - 1);
            const bool neg = threadIdx.x >= kPairTile;
            const int pair = g0 + p < g_hi ? perm[g0 + p] : -1;
            const float c = pair >= 0 ? ds[pair] : 0.f;
            sHid[threadIdx.x] = pair >= 0 ? (neg ? nh[pair] : ph[pair]) : 0;
            sTid[threadIdx.x] = pair >= 0 ? (neg ? nt[pair] : pt[pair]) : 0;
            sDs[threadIdx.x] = neg ? -c : c;
        }
        __syncthreads();
        constexpr int kMaxPer = 8 / VK;              // TILE * nv / 1024 loads per thread and matrix (k <= 256)
        float rt[kMaxPer][VK], rh[kMaxPer][VK];
#pragma unroll
        for (int j = 0; j < kMaxPer; ++j) {
            const int idx = threadIdx.x + 1024 * j;
            const int i = idx / nv, c = idx - i * nv;
            const bool ok = idx < TILE * nv && sDs[i] != 0.f;
#pragma unroll
            for (int q = 0; q < VK; ++q) { rt[j][q] = 0.f; rh[j][q] = 0.f; }
            if (ok) { ldv<VK>(rt[j], ent + sTid[i] * k + c * VK); ldv<VK>(rh[j], ent + sHid[i] * k + c * VK); }
        }
#pragma unroll
        for (int j = 0; j < kMaxPer; ++j) {
            const int idx = threadIdx.x + 1024 * j;
            if (idx < TILE * nv) {
                const int i = idx / nv, c = (idx - i * nv) * VK;
                const float d = sDs[i];
#pragma unroll
                for (int q = 0; q < VK; ++q) { sT[i * S + c + q] = rt[j][q]; sH[i * S + c + q] = d * rh[j][q]; }
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int tl = wave + 16 * t;
            if (tl < nt2) {
                const int at = tl / ntile, bt = tl - at * ntile;
                int a_in = min(at * 32 + li, k - 1), b_in = min(bt * 32 + li, k - 1);   // (clamped columns are never stored)
                // (opaque to the optimiser: otherwise the 128 loop-invariant LDS addresses of the four tiles are hoisted out of the
                // chunk loop and spilled)
                asm volatile("" : "+v"(a_in), "+v"(b_in));
#pragma unroll
                for (int kk = 0; kk < TILE; kk += 2) {
                    const int i = kk + lk;
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(sH[i * S + a_in], sT[i * S + b_in], acc[t], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // (keeps the LDS operand reads of later tiles from being hoisted over this one: registers)
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int tl = wave + 16 * t;
        const int at = tl / ntile, bt = tl - at * ntile;
        const int b_in = bt * 32 + li;
        const bool on = tl < nt2 && b_in < k;
        float* gM = g_rel + (int64_t)rel * k * k;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int a = at * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lk;
            if (on && a < k && acc[t][reg] != 0.f) {
                float* q = gM + (int64_t)a * k + b_in;
                if (shared_rel) unsafeAtomicAdd(q, -acc[t][reg]); else *q -= acc[t][reg];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Large batches, second form of the pair step: V = H M_r and U = T M_r^T as GEMMs with the rows of 32 pairs (64 triples, two
// consecutive 16-pair tiles of one relation) as M, the way kge_ntn.hip treats NTN's shared tensor.  A wave owns 8 pairs = one
// 16-row block (rows 2 p and 2 p + 1: a pair's positive and its negative, so both energies of a pair meet in ONE lane of the
// accumulator layout and the hinge is a few register operations); its rows' H (then T) operand lives in registers for the whole
// pass (lane (l, lk) holds k = 16 kb + 4 lk + kk: 16 consecutive bytes of the row per slab), M_r streams through LDS in 16-deep
// slabs (transposed while staging for U), v_mfma_f32_16x16x4_f32 with NB accumulator blocks per wave.  Entity gradients leave
// through float atomics as in k_rescal_pair (the uncorrupted side of a pair: one merged atomic); dL/denergy goes to ds_out for the
// relation-matrix gradient's launch.  k <= 16 NB, any k; VEC: k % 4 == 0 and a 16-byte aligned table.
// Round 4: every global load sits on a clamped (always valid) address and is masked by a select where it is USED -- the first form
// had its 507 loads inside 824 divergent branches, each join an s_waitcnt vmcnt(0) -- and blocks are numbered so that the live
// (even) tiles of a relation are consecutive block ids (kge_mfma_blocks.h: strided_tile).
template <int NB, bool VEC>
__global__ __launch_bounds__(256, 2) void k_rescal_rows(const float* __restrict__ ent, const float* __restrict__ relm,
                                                        float* __restrict__ g_ent, const int64_t* __restrict__ ph,
                                                        const int64_t* __restrict__ pt, const int64_t* __restrict__ nh,
                                                        const int64_t* __restrict__ nt, const int* __restrict__ offsets,
                                                        const int* __restrict__ tile_off, const int* __restrict__ tile_rel,
                                                        const int* __restrict__ perm, int R, int k, float margin, float* __restrict__ loss,
                                                        unsigned* __restrict__ touched, float* __restrict__ ds_out, int tiles) {
    constexpr int DP = 16 * NB, PITCH = DP + 4, NK = 4 * NB;
    __shared__ __attribute__((aligned(16))) float sW[2][16][PITCH];
    int rel, tin;
    const int tile = strided_tile<2>(tiles);
    if (tile >= tiles || !locate_tile(tile_off, tile_rel, R, tile, rel, tin)) return;
    if (tin & 1) return;                       // (the workgroup of an even tile takes the odd one after it as well)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l = lane & 15, lk = lane >> 4;
    const int g_lo = offsets[rel] + tin * kPairTile, g_hi = min(offsets[rel + 1], g_lo + 2 * kPairTile);
    // A-operand row of this lane: row l of the wave = pair 8 wave + l / 2, side l & 1
    const int a_g = g_lo + 8 * wave + (l >> 1);
    const bool a_on = a_g < g_hi;
    const int a_pair = perm[min(a_g, g_hi - 1)];
    // accumulator rows of this lane: 4 lk + q = pairs 2 lk (q = 0 positive, 1 negative) and 2 lk + 1 (q = 2, 3)
    int c_pair[2];
    bool c_on[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int gp = g_lo + 8 * wave + 2 * lk + j;
        c_on[j] = gp < g_hi;
        c_pair[j] = perm[min(gp, g_hi - 1)];
    }
    const int a_h = (int)((l & 1) ? nh[a_pair] : ph[a_pair]), a_t = (int)((l & 1) ? nt[a_pair] : pt[a_pair]);
    int c_h[4], c_t[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        c_h[2 * j] = (int)ph[c_pair[j]]; c_h[2 * j + 1] = (int)nh[c_pair[j]];
        c_t[2 * j] = (int)pt[c_pair[j]]; c_t[2 * j + 1] = (int)nt[c_pair[j]];
    }
    float a[NK];
    auto load_a = [&](int id) __attribute__((always_inline)) {   // element 4 kb + kk = k index 16 kb + 4 lk + kk of the row
        const float* __restrict__ row = ent + (int64_t)id * k;
        if constexpr (VEC) {
            unroll_seq([&](auto kbc) __attribute__((always_inline)) {
                constexpr int kb = decltype(kbc)::value;
                const float4 v = *reinterpret_cast<const float4*>(row + min(16 * kb + 4 * lk, k - 4));
                a[4 * kb] = v.x; a[4 * kb + 1] = v.y; a[4 * kb + 2] = v.z; a[4 * kb + 3] = v.w;
            }, std::make_integer_sequence<int, NB>{});
        } else {
            unroll_seq([&](auto ksc) __attribute__((always_inline)) {
                constexpr int ks = decltype(ksc)::value;
                a[ks] = row[min(16 * (ks >> 2) + 4 * lk + (ks & 3), k - 1)];
            }, std::make_integer_sequence<int, NK>{});
        }
    };
    auto mask_a = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) a[ks] = (a_on && 16 * (ks >> 2) + 4 * lk + (ks & 3) < k) ? a[ks] : 0.f;
    };
    load_a(a_h);
    f32x4v acc[NB];
    const float* __restrict__ M = relm + (int64_t)rel * k * k;
    float st[NB];
    int buf = 0;
    // one pass over M_r: TR = false: B[kq][c] = M[16 kb + kq][c] (V = H M); TR = true: B[kq][c] = M[c][16 kb + kq] (U = T M^T)
    auto pass = [&](auto tr_tag) __attribute__((always_inline)) {
        constexpr bool TR = decltype(tr_tag)::value;
        const int kq = TR ? (threadIdx.x & 15) : (threadIdx.x >> 4), c0 = TR ? (threadIdx.x >> 4) : (threadIdx.x & 15);
        auto fetch = [&](int kb) __attribute__((always_inline)) {   // raw values off clamped addresses; masked at the LDS store
            const int kr = min(16 * kb + kq, k - 1);
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int c = min(c0 + 16 * u, k - 1);
                st[u] = TR ? M[(int64_t)c * k + kr] : M[(int64_t)kr * k + c];
            }
        };
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) acc[cb] = f32x4v{0.f, 0.f, 0.f, 0.f};
        fetch(0);
        unroll_seq([&](auto kbc) __attribute__((always_inline)) {
            constexpr int kb = decltype(kbc)::value;
#pragma unroll
            for (int u = 0; u < NB; ++u)   // (natural accumulator columns: coalesced atomics)
                sW[buf][kq][BlkMapNat<NB>::pos(u, c0)] = (16 * kb + kq < k && c0 + 16 * u < k) ? st[u] : 0.f;
            __syncthreads();   // slab kb is in LDS; everybody finished reading the buffer that is written next
            if (kb + 1 < NB) fetch(kb + 1);
            float b[2][NB];
            read_blocks<NB>(&sW[buf][4 * lk][0], l, b[0]);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                if (kk + 1 < 4) read_blocks<NB>(&sW[buf][4 * lk + kk + 1][0], l, b[(kk + 1) & 1]);
                KGE_KEEP_READS_AHEAD();
#pragma unroll
                for (int cb = 0; cb < NB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * kb + kk], b[kk & 1][cb], acc[cb], 0, 0, 0);
            }
            buf ^= 1;
        }, std::make_integer_sequence<int, NB>{});
    };
    mask_a();
    pass(std::false_type{});
    // ---- energies -s = -<V, T>, margin hinge of the lane's two pairs (as k_hinge_coeffs).  The T elements the accumulators meet are
    // fetched in accumulator layout only now: 4 NB registers that are not live during the MFMA passes (three waves per SIMD)
    float tq[NB][4];
    unroll_seq([&](auto cbc) __attribute__((always_inline)) {
        constexpr int cb = decltype(cbc)::value;
        const int col = min(16 * cb + l, k - 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) tq[cb][q] = ent[(int64_t)c_t[q] * k + col];
    }, std::make_integer_sequence<int, NB>{});
    float p[4] = {0.f, 0.f, 0.f, 0.f};
    unroll_seq([&](auto cbc) __attribute__((always_inline)) {
        constexpr int cb = decltype(cbc)::value;
        const bool in = 16 * cb + l < k;
#pragma unroll
        for (int q = 0; q < 4; ++q) p[q] = fmaf(acc[cb][q], (in && c_on[q >> 1]) ? tq[cb][q] : 0.f, p[q]);
    }, std::make_integer_sequence<int, NB>{});
#pragma unroll
    for (int q = 0; q < 4; ++q) p[q] = gsum<16>(p[q]);
    float c[2], hl = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float v = (-p[2 * j]) + margin - (-p[2 * j + 1]);
        c[j] = c_on[j] ? (v > 0.f ? 1.f : (v == 0.f ? 0.5f : 0.f)) : 0.f;
        hl += c_on[j] ? fmaxf(v, 0.f) : 0.f;
        if (l == 0 && c_on[j]) ds_out[c_pair[j]] = c[j];
    }
    {
        const float tot = wave_sum(l == 0 ? hl : 0.f);
        if (lane == 0 && tot != 0.f) unsafeAtomicAdd(loss + (blockIdx.x % kLossSlots) * kLossStride, tot);
    }
    if (!__syncthreads_or(c[0] != 0.f || c[1] != 0.f)) return;   // every pair of the workgroup inside the margin: no gradient
    if (touched && l == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (c[q >> 1] != 0.f) {
                atomicOr(touched + (c_h[q] >> 5), 1u << (c_h[q] & 31));
                atomicOr(touched + (c_t[q] >> 5), 1u << (c_t[q] & 31));
            }
    }
    // rows of a pair leave as grad = -ds x: positive row -c x_pos, negative row +c x_neg; the same entity on both sides: one atomic
    auto scatter = [&](const int (&ids)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (c[j] == 0.f) continue;
            const bool same = ids[2 * j] == ids[2 * j + 1];
            float* __restrict__ op = g_ent + (int64_t)ids[2 * j] * k + l;
            float* __restrict__ on = g_ent + (int64_t)ids[2 * j + 1] * k + l;
            unroll_seq([&](auto cbc) __attribute__((always_inline)) {
                constexpr int cb = decltype(cbc)::value;
                if (16 * cb + l < k) {
                    const float xp = acc[cb][2 * j], xn = acc[cb][2 * j + 1];
                    unsafeAtomicAdd(op + 16 * cb, same ? -c[j] * (xp - xn) : -c[j] * xp);
                    if (!same) unsafeAtomicAdd(on + 16 * cb, c[j] * xn);
                }
            }, std::make_integer_sequence<int, NB>{});
        }
    };
    load_a(a_t);                               // (the U pass's A operand: requested under the atomics of the V pass)
    scatter(c_t);                              // grad_t = -ds V
    // ---- U = T M^T, grad_h = -ds U
    mask_a();
    pass(std::true_type{});
    scatter(c_h);
}

// The relation-matrix gradient of the large-batch step as a GEMM over gathered rows, G_r = sum_i ds_i h_i t_i^T (K = the triples of
// the relation), in the style of kge_ntn.hip's k_ntn_outer: a workgroup takes a run of kGRun 16-pair tiles of one relation (256 rows)
// and ONE half of the output columns (NBJ blocks from column blockIdx.y * 16 NBJ on), stages 16 rows per slab
// -- ds_i h_i in natural order, t_i permuted so that accumulator block b, lane l is column jbase + 16 b + l -- and holds NBI x NBJ
// accumulator blocks over its four waves (row block b on wave b % 4).  The ids of a slab's rows hang on perm -> pair -> ids: they are
// resolved two slabs ahead, the rows one slab ahead.  grad_M = -G: plain read-modify-write when the relation has a single run, float
// atomics otherwise.
constexpr int kGRun = 8;
template <int NBI, int NBJ>
__global__ __launch_bounds__(256, 2) void k_rescal_g(const float* __restrict__ ent, float* __restrict__ g_rel,
                                                     const int64_t* __restrict__ ph, const int64_t* __restrict__ pt,
                                                     const int64_t* __restrict__ nh, const int64_t* __restrict__ nt,
                                                     const int* __restrict__ offsets, const int* __restrict__ tile_off,
                                                     const int* __restrict__ tile_rel, const int* __restrict__ perm, int R, int k,
                                                     const float* __restrict__ ds) {
    constexpr int DPI = 16 * NBI, DPJ = 16 * NBJ, PA = DPI + 4, PB = DPJ + 4;
    const int jbase = blockIdx.y * DPJ;   // the workgroup's half of the output columns
    constexpr int RBW = (NBI + 3) / 4;   // row blocks per wave
    __shared__ __attribute__((aligned(16))) float sA[2][16][PA], sB[2][16][PB];
    int rel, tin;
    if (!locate_tile(tile_off, tile_rel, R, blockIdx.x, rel, tin)) return;
    if (tin % kGRun) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l = lane & 15, lk = lane >> 4;
    const int r0 = offsets[rel], r1 = offsets[rel + 1];
    const int g_lo = r0 + tin * kPairTile, g_hi = min(r1, g_lo + kGRun * kPairTile);
    const bool shared_rel = (r1 - r0) > kGRun * kPairTile;
    const int nslab = (g_hi - g_lo + 7) / 8;   // 8 pairs = 16 rows per slab
    f32x4v acc[RBW][NBJ];
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
        for (int cb = 0; cb < NBJ; ++cb) acc[rb][cb] = f32x4v{0.f, 0.f, 0.f, 0.f};
    // staging roles: slab row kq = tid / 16 = pair kq / 2, side kq & 1; columns c0 + 16 u
    const int kq = threadIdx.x >> 4, c0 = threadIdx.x & 15;
    int64_t id_h = 0, id_t = 0;
    float id_s = 0.f;
    auto resolve = [&](int sl) __attribute__((always_inline)) {   // ids and signed coefficient of this thread's row of slab sl
        const int g = g_lo + 8 * sl + (kq >> 1);
        id_s = 0.f; id_h = 0; id_t = 0;
        if (sl < nslab && g < g_hi) {
            const int pair = perm[g];
            const float c = ds[pair];
            const bool neg = kq & 1;
            id_s = neg ? -c : c;
            id_h = neg ? nh[pair] : ph[pair];
            id_t = neg ? nt[pair] : pt[pair];
        }
    };
    float sta[NBI], stb[NBJ], st_s = 0.f, nx_s = 0.f;
    auto fetch = [&]() __attribute__((always_inline)) {   // rows of the slab whose ids are resolved; its coefficient travels with them
        nx_s = id_s;
        const bool live = id_s != 0.f;
        const float* __restrict__ hr = ent + id_h * k + c0;
        const float* __restrict__ tr = ent + id_t * k + jbase + c0;
#pragma unroll
        for (int u = 0; u < NBI; ++u) sta[u] = (live && c0 + 16 * u < k) ? hr[16 * u] : 0.f;
#pragma unroll
        for (int u = 0; u < NBJ; ++u) stb[u] = (live && jbase + c0 + 16 * u < k) ? tr[16 * u] : 0.f;
    };
    int buf = 0;
    resolve(0);
    fetch();
    st_s = nx_s;
    resolve(1);
    for (int sl = 0; sl < nslab; ++sl) {
#pragma unroll
        for (int u = 0; u < NBI; ++u) sA[buf][kq][c0 + 16 * u] = sta[u] * st_s;
#pragma unroll
        for (int u = 0; u < NBJ; ++u) sB[buf][kq][BlkMapNat<NBJ>::pos(u, c0)] = stb[u];
        __syncthreads();
        if (sl + 1 < nslab) { fetch(); st_s = nx_s; }   // rows of slab sl + 1 (ids resolved one iteration ago)
        resolve(sl + 2);                                   // ids of slab sl + 2: three dependent loads, two slabs of MFMAs to land
        float b[2][NBJ], av[2][RBW];
        auto operands = [&](int kk, int slot) __attribute__((always_inline)) {
            read_blocks<NBJ>(&sB[buf][4 * kk + lk][0], l, b[slot]);
#pragma unroll
            for (int rb = 0; rb < RBW; ++rb) av[slot][rb] = wave + 4 * rb < NBI ? sA[buf][4 * kk + lk][16 * (wave + 4 * rb) + l] : 0.f;
        };
        operands(0, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk + 1 < 4) operands(kk + 1, (kk + 1) & 1);
            KGE_KEEP_READS_AHEAD();
#pragma unroll
            for (int rb = 0; rb < RBW; ++rb) {
                if (wave + 4 * rb < NBI) {   // wave-uniform
#pragma unroll
                    for (int cb = 0; cb < NBJ; ++cb)
                        acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk & 1][rb], b[kk & 1][cb], acc[rb][cb], 0, 0, 0);
                }
            }
        }
        buf ^= 1;
    }
    float* __restrict__ gM = g_rel + (int64_t)rel * k * k;
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb) {
        if (wave + 4 * rb >= NBI) continue;
#pragma unroll
        for (int cb = 0; cb < NBJ; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = 16 * (wave + 4 * rb) + 4 * lk + q, j = jbase + 16 * cb + l;
                const float v = acc[rb][cb][q];
                if (i < k && j < k && v != 0.f) {
                    float* o = gM + (int64_t)i * k + j;
                    if (shared_rel) unsafeAtomicAdd(o, -v); else *o -= v;
                }
            }
    }
}

// k_rescal_g, second form (k % 4 == 0, 16-byte aligned table): the same GEMM with rows fetched as 16-byte pieces (a thread: NVA
// pieces of its row's head vector and NVB of the tail vector's column half per slab), a ring of D slabs in flight in registers, and
// no load inside a branch: within an iteration every load depends only on values loaded in EARLIER iterations (pair of slab
// sl + D + 2, then coefficient + ids of slab sl + D + 1, then the rows of slab sl + D), all on clamped addresses, masked where they
// are stored to LDS.  The first form's 164 loads sat in 172 branches with 120 s_waitcnt vmcnt(0): one exposed round trip per slab
// for the id chain and one for the rows.  Blocks are numbered so that the runs of a relation are consecutive block ids (strided_tile).
template <int NBI, int NBJ, int D>
__global__ __launch_bounds__(256, 2) void k_rescal_g2(const float* __restrict__ ent, float* __restrict__ g_rel,
                                                      const int64_t* __restrict__ ph, const int64_t* __restrict__ pt,
                                                      const int64_t* __restrict__ nh, const int64_t* __restrict__ nt,
                                                      const int* __restrict__ offsets, const int* __restrict__ tile_off,
                                                      const int* __restrict__ tile_rel, const int* __restrict__ perm, int R, int k,
                                                      const float* __restrict__ ds, int tiles) {
    constexpr int DPI = 16 * NBI, DPJ = 16 * NBJ, PA = DPI + 4, PB = DPJ + 4;
    constexpr int RBW = (NBI + 3) / 4;   // row blocks per wave
    constexpr int NVA = (NBI + 3) / 4, NVB = (NBJ + 3) / 4;   // 16-byte pieces per thread (16 threads per row: pieces f, f + 16, ...)
    __shared__ __attribute__((aligned(16))) float sA[2][16][PA], sB[2][16][PB];
    const int jbase = blockIdx.y * DPJ;   // the workgroup's half of the output columns
    int rel, tin;
    const int tile = strided_tile<kGRun>(tiles);
    if (tile >= tiles || !locate_tile(tile_off, tile_rel, R, tile, rel, tin)) return;
    if (tin % kGRun) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l = lane & 15, lk = lane >> 4;
    const int r0 = offsets[rel], r1 = offsets[rel + 1];
    const int g_lo = r0 + tin * kPairTile, g_hi = min(r1, g_lo + kGRun * kPairTile);
    const bool shared_rel = (r1 - r0) > kGRun * kPairTile;
    const int nslab = (g_hi - g_lo + 7) / 8;   // 8 pairs = 16 rows per slab
    f32x4v acc[RBW][NBJ];
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
        for (int cb = 0; cb < NBJ; ++cb) acc[rb][cb] = f32x4v{0.f, 0.f, 0.f, 0.f};
    // staging roles: slab row kq = tid / 16 = pair kq / 2, side kq & 1; pieces f + 16 v of the row
    const int kq = threadIdx.x >> 4, f = threadIdx.x & 15;
    const bool neg = kq & 1;
    const int64_t* __restrict__ hcol = neg ? nh : ph;
    const int64_t* __restrict__ tcol = neg ? nt : pt;
    float4 ra[D][NVA], rb_[D][NVB];
    float rs[D];
    auto resolve1 = [&](int sl) __attribute__((always_inline)) -> int { return perm[min(g_lo + 8 * sl + (kq >> 1), g_hi - 1)]; };
    auto fetch = [&](auto slot, int64_t idh, int64_t idt, float coef) __attribute__((always_inline)) {
        constexpr int S = decltype(slot)::value;
        rs[S] = coef;
        const float* __restrict__ hr = ent + idh * k;
        const float* __restrict__ tr = ent + idt * k + jbase;
#pragma unroll
        for (int v = 0; v < NVA; ++v) ra[S][v] = *reinterpret_cast<const float4*>(hr + min(4 * (f + 16 * v), k - 4));
#pragma unroll
        for (int v = 0; v < NVB; ++v) rb_[S][v] = *reinterpret_cast<const float4*>(tr + min(4 * (f + 16 * v), k - 4 - jbase));
    };
    auto coef_of = [&](int sl, float c) __attribute__((always_inline)) -> float {
        return g_lo + 8 * sl + (kq >> 1) < g_hi ? (neg ? -c : c) : 0.f;
    };
    // prologue: the chains of the first D slabs side by side, ids of slab D, pair of slab D + 1
    int p_pair[D + 2];
    int64_t p_h[D + 1], p_t[D + 1];
    float p_c[D + 1];
#pragma unroll
    for (int s2 = 0; s2 <= D + 1; ++s2) p_pair[s2] = resolve1(s2);
#pragma unroll
    for (int s2 = 0; s2 <= D; ++s2) { p_c[s2] = ds[p_pair[s2]]; p_h[s2] = hcol[p_pair[s2]]; p_t[s2] = tcol[p_pair[s2]]; }
    unroll_seq([&](auto sc) __attribute__((always_inline)) {
        constexpr int S = decltype(sc)::value;
        fetch(sc, p_h[S], p_t[S], coef_of(S, p_c[S]));
    }, std::make_integer_sequence<int, D>{});
    int64_t n_h = p_h[D], n_t = p_t[D];      // slab sl + D: ids and coefficient known
    float n_c = p_c[D];
    int m_pair = p_pair[D + 1];              // slab sl + D + 1: pair known
    int buf = 0;
    auto step = [&](auto slot, int sl) __attribute__((always_inline)) {
        constexpr int S = decltype(slot)::value;
        const float sc = rs[S];
#pragma unroll
        for (int v = 0; v < NVA; ++v) {
            const int c = 4 * (f + 16 * v);
            if (c < DPI) {
                const float w = c < k ? sc : 0.f;
                *reinterpret_cast<float4*>(&sA[buf][kq][c]) = float4{ra[S][v].x * w, ra[S][v].y * w, ra[S][v].z * w, ra[S][v].w * w};
            }
        }
#pragma unroll
        for (int v = 0; v < NVB; ++v) {
            const int c = 4 * (f + 16 * v);
            if (c < DPJ) {
                const bool lb = sc != 0.f && jbase + c < k;
                const int u = c >> 4, c0 = c & 15;   // columns c .. c + 3 of block u: positions 4 apart
                sB[buf][kq][BlkMapNat<NBJ>::pos(u, c0)] = lb ? rb_[S][v].x : 0.f;
                sB[buf][kq][BlkMapNat<NBJ>::pos(u, c0 + 1)] = lb ? rb_[S][v].y : 0.f;
                sB[buf][kq][BlkMapNat<NBJ>::pos(u, c0 + 2)] = lb ? rb_[S][v].z : 0.f;
                sB[buf][kq][BlkMapNat<NBJ>::pos(u, c0 + 3)] = lb ? rb_[S][v].w : 0.f;
            }
        }
        __syncthreads();
        {
            const float c1 = ds[m_pair];
            const int64_t h1 = hcol[m_pair], t1 = tcol[m_pair];
            const int pair2 = resolve1(sl + D + 2);
            fetch(slot, n_h, n_t, coef_of(sl + D, n_c));   // rows of slab sl + D into the registers just emptied (past the run: dead)
            n_h = h1; n_t = t1; n_c = c1;
            m_pair = pair2;
        }
        float b[2][NBJ], av[2][RBW];
        auto operands = [&](int kk, int s2) __attribute__((always_inline)) {
            read_blocks<NBJ>(&sB[buf][4 * kk + lk][0], l, b[s2]);
#pragma unroll
            for (int rb = 0; rb < RBW; ++rb) av[s2][rb] = wave + 4 * rb < NBI ? sA[buf][4 * kk + lk][16 * (wave + 4 * rb) + l] : 0.f;
        };
        operands(0, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk + 1 < 4) operands(kk + 1, (kk + 1) & 1);
            KGE_KEEP_READS_AHEAD();
#pragma unroll
            for (int rb = 0; rb < RBW; ++rb) {
                if (wave + 4 * rb < NBI) {   // wave-uniform
#pragma unroll
                    for (int cb = 0; cb < NBJ; ++cb)
                        acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk & 1][rb], b[kk & 1][cb], acc[rb][cb], 0, 0, 0);
                }
            }
        }
        buf ^= 1;
    };
    int sl = 0;
    for (; sl + D <= nslab; sl += D)   // whole groups of D steps without a branch between them: the memory counter waits stay partial
        unroll_seq([&](auto sc) __attribute__((always_inline)) { step(sc, sl + decltype(sc)::value); }, std::make_integer_sequence<int, D>{});
    unroll_seq([&](auto sc) __attribute__((always_inline)) {
        if (sl + decltype(sc)::value < nslab) step(sc, sl + decltype(sc)::value);   // workgroup-uniform
    }, std::make_integer_sequence<int, D>{});
    float* __restrict__ gM = g_rel + (int64_t)rel * k * k;
    if (!shared_rel) {   // sole writer: the old values of a row block are requested together, on clamped addresses, before the first store
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb) {
            float old[NBJ][4];
#pragma unroll
            for (int cb = 0; cb < NBJ; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = min(16 * (wave + 4 * rb) + 4 * lk + q, k - 1), j = min(jbase + 16 * cb + l, k - 1);
                    old[cb][q] = gM[(int64_t)i * k + j];
                }
#pragma unroll
            for (int cb = 0; cb < NBJ; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[rb][cb][q] = old[cb][q] - acc[rb][cb][q];
        }
    }
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb) {
        if (wave + 4 * rb >= NBI) continue;
#pragma unroll
        for (int cb = 0; cb < NBJ; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = 16 * (wave + 4 * rb) + 4 * lk + q, j = jbase + 16 * cb + l;
                const float v = acc[rb][cb][q];
                if (i < k && j < k) {
                    float* o = gM + (int64_t)i * k + j;
                    if (shared_rel) { if (v != 0.f) unsafeAtomicAdd(o, -v); } else *o = v;
                }
            }
    }
}

template <int NBI>
static void launch_rescal_g(const kge_model_desc* m, const int64_t* ph, const int64_t* pt, const int64_t* nh, const int64_t* nt,
                            const GroupWs& g, unsigned tiles, int R, int k, const float* ds, hipStream_t s) {
    constexpr int JA = (NBI + 1) / 2;   // column blocks per half (an odd NBI leaves one masked block in the second half)
    const bool vec = (k & 3) == 0 && (reinterpret_cast<uintptr_t>(m->tables[0]) & 15) == 0 && k >= 16 * JA + 4;
    if (vec && switch_value("RESCAL_G2") != 0) {   // (KGE_RESCAL_G2=0: the dword-gather form, A/B)
        const unsigned grid = (tiles + kGRun - 1) / kGRun * kGRun;
        hipLaunchKernelGGL((k_rescal_g2<NBI, JA, (NBI > 8 ? 2 : 3)>), dim3(grid, NBI > 1 ? 2 : 1), dim3(256), 0, s, m->tables[0], m->grads[1], ph, pt, nh, nt,
                           g.offsets, g.tile_off, g.tile_rel, g.perm, R, k, ds, (int)tiles);
        return;
    }
    hipLaunchKernelGGL((k_rescal_g<NBI, JA>), dim3(tiles, NBI > 1 ? 2 : 1), dim3(256), 0, s, m->tables[0], m->grads[1], ph, pt, nh, nt,
                       g.offsets, g.tile_off, g.tile_rel, g.perm, R, k, ds);
}

// Pairs from which dL/denergy is left behind and the relation-matrix gradient gets its own launch (with k_rescal_rows for V / U where
// k <= 208): 8 192 -- or 512 on graphs with many relations, where most 16-pair tiles of k_rescal_pair are partial and each still pays
// its k^2 atomics (profiles/r04_rescal_threshold.txt: FB15k shape k = 200, B = 512 ... 4 096: 130 / 195 / 319 / 383 -> 104 / 176 / 251 /
// 298 us; YAGO3-10 shape, 37 relations: k_rescal_pair wins up to ~6 000 pairs).  KGE_RESCAL_SPLIT_MIN=<pairs> overrides (A/B).
constexpr int64_t kPairSplitG = 8192, kPairSplitManyRel = 512, kManyRelations = 512;
static bool pair_split(int64_t R, int64_t n) {
    const int v = switch_value("RESCAL_SPLIT_MIN");
    if (v > 0) return n >= (int64_t)v;
    return n >= kPairSplitG || (R >= kManyRelations && n >= kPairSplitManyRel);
}

static size_t rescal_pair_lds_bytes(int k) {
    const int S = (k + 1) | 1;
    return (size_t)(2 * TILE * S + 8 * 16 * 64 + 8 * TILE + TILE) * sizeof(float) + (size_t)2 * TILE * sizeof(long long) + 16;
}
static size_t rescal_pair_ws_bytes(int64_t R, int64_t n) {
    return (size_t)(4 * (R + 1) + n + (n / kPairTile + R + 1) + 8 + n) * sizeof(int);   // (+ n floats of dL/denergy: used from pair_split_min() pairs on)
}

// Below the split thresholds the step runs as (relation chunk, 32-column slab) workgroups (kge_rescal_slab.hip) when the caller's
// workspace has room for the V rows and the slabs' energy shares BEHIND the pairwise step's standard layout (two grouping
// workspaces + 2 n scores, kge_workspace_bytes); KGE_RESCAL_SLAB=0: the one-launch tile kernel k_rescal_pair (A/B).
static size_t align256d(size_t x) { return (x + 255) & ~(size_t)255; }
static size_t rescal_slab_offset(int64_t R, int64_t n) {
    return 2 * align256d(group_ws_bytes(R, n)) + align256d((size_t)2 * n * sizeof(float));
}
size_t rescal_slab_extra_bytes(const kge_model_desc* m, int64_t n) {
    if (m->dim % 2 != 0 || m->dim > 256 || n >= kPairSplitG) return 0;
    if (!group_small_ok(n, m->tot_relation)) return 0;
    return align256d(rescal_slab_ws_bytes(m->dim, m->tot_relation, n));
}

bool rescal_pair_step_ok(const kge_model_desc* m, int64_t n, size_t ws_bytes) {
    return m->dim % 2 == 0 && m->dim <= 256 && n < (1ll << 31) && ws_bytes >= rescal_pair_ws_bytes(m->tot_relation, n);
}

// negatives share pr (the caller passed nr == pr); ws: the pairwise step's scorer workspace
static bool rescal_slab_taken(const kge_model_desc* m, int64_t n, size_t ws_bytes) {
    const int64_t R = m->tot_relation;
    const size_t slab_bytes = rescal_slab_extra_bytes(m, n);
    return !pair_split(R, n) && slab_bytes != 0 && ws_bytes != (size_t)-1 && ws_bytes >= rescal_slab_offset(R, n) + slab_bytes &&
           switch_value("RESCAL_SLAB") != 0;
}
bool rescal_stage_ok(const kge_model_desc* m, int64_t n, size_t ws_bytes) {
    // (the sorted grouping keeps 3 R + 2 n + 2 ints in the 64 KB of LDS a launch gets without opting in -- next to the 8 bytes of
    // static LDS k_rel_group_small declares (s_carry): at the boundary the launch would fail instead of the step falling back)
    return m->dim % 4 == 0 && n <= kStageMaxN && (3 * m->tot_relation + 2 * n + 2) * (int64_t)sizeof(int) <= 64 * 1024 - 64 &&
           m->tot_entity < (1ll << 31) && rescal_slab_taken(m, n, ws_bytes);
}

int launch_rescal_pair_step(const kge_model_desc* m, const int64_t* ph, const int64_t* pr, const int64_t* pt, const int64_t* nh,
                            const int64_t* nt, int64_t n, float margin, float* loss, void* ws, size_t ws_bytes, unsigned* touched,
                            const kge_rescal_stage* stage, hipStream_t s) {
    const int k = m->dim;
    const int64_t R = m->tot_relation;
    if (!rescal_pair_step_ok(m, n, ws_bytes)) { set_error("RESCAL pair step: unsupported shape or workspace"); return -1; }
    const GroupWs g = carve_group_ws(ws, R, n);       // (tile_rel, the last array, holds n / 16 + R + 1 entries here)
    if (stage && (!rescal_stage_ok(m, n, ws_bytes) || !touched)) {
        set_error("kge_rescal_pair_step_staged: the staged form takes hidden sizes that are multiples of 4, at most %d pairs, the slab form's "
                  "workspace (kge_workspace_bytes) and the touched-row bitmap (kge_rescal_stage_ok)", kStageMaxN);
        return -1;
    }
    if (rescal_slab_taken(m, n, ws_bytes)) {
        void* ws_slab = (char*)ws + rescal_slab_offset(R, n);
        PairGather pg;
        pg.ph = ph; pg.pt = pt; pg.nh = nh; pg.nt = nt;
        rescal_slab_gather(ws_slab, k, R, n, &pg);
        if (stage) pg.sorted = 1;
        int rcs = group_by_relation_split(id_whole(pr, n), n, R, g, s, nullptr, 0, kSlabChunk, &pg);
        if (rcs) return rcs;
        return launch_rescal_slab_step(m, n, g, margin, loss, touched, ws_slab, stage, s);
    }
    int rc = group_by_relation_split(id_whole(pr, n), n, R, g, s, nullptr, 0, kPairTile);
    if (rc) return rc;
    const size_t lds = rescal_pair_lds_bytes(k);
    // (per call, not once per process: the attribute is per device, and a process may drive several)
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((k & 3) == 0 ? (const void*)k_rescal_pair<4> : (const void*)k_rescal_pair<2>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) != hipSuccess) {
        (void)hipGetLastError();
        set_error("RESCAL pair step: the device refuses %zu bytes of dynamic LDS", lds);
        return -1;
    }
    const unsigned tiles = (unsigned)(n / kPairTile + R + 1);
    float* ds = pair_split(R, n) ? (float*)(g.tile_rel + tiles) : nullptr;
    const int S = (k + 1) | 1;
    const size_t lds_gm = (size_t)(2 * TILE * S + TILE) * sizeof(float) + (size_t)2 * TILE * sizeof(long long);
    // large batches: V / U as batch-as-M GEMMs (k_rescal_rows), dL/denergy to the relation-owner launch below (KGE_RESCAL_ROWS=0/1: A/B)
    const bool rows = ds != nullptr && k <= 208 && switch_value("RESCAL_ROWS") != 0;
    if (rows) {
        const int nb = (k + 15) / 16;
        const bool vec = (k & 3) == 0 && ((reinterpret_cast<uintptr_t>(m->tables[0])) & 15) == 0;
        const unsigned grid2 = (tiles + 1) / 2 * 2;
#define KGE_RR(J) case J: if (vec) hipLaunchKernelGGL((k_rescal_rows<J, true>), dim3(grid2), dim3(256), 0, s, m->tables[0], m->tables[1], m->grads[0], ph, pt, nh, \
                                             nt, g.offsets, g.tile_off, g.tile_rel, g.perm, (int)R, k, margin, loss, touched, ds, (int)tiles); \
                  else hipLaunchKernelGGL((k_rescal_rows<J, false>), dim3(grid2), dim3(256), 0, s, m->tables[0], m->tables[1], m->grads[0], ph, pt, nh, \
                                             nt, g.offsets, g.tile_off, g.tile_rel, g.perm, (int)R, k, margin, loss, touched, ds, (int)tiles); break;
        switch (nb) { KGE_RR(1) KGE_RR(2) KGE_RR(3) KGE_RR(4) KGE_RR(5) KGE_RR(6) KGE_RR(7) KGE_RR(8) KGE_RR(9) KGE_RR(10) KGE_RR(11) KGE_RR(12) KGE_RR(13) }
#undef KGE_RR
    }
    const int g_sw = switch_value("RESCAL_G");
    // the relation-matrix gradient as a GEMM over gathered rows (k_rescal_g) where relations span several 128-pair runs
    // (the 16-byte-gather form k_rescal_g2 wins at every relation count measured -- FB15k shape, 24 pairs per relation: 345 -> 175 us
    // against k_rescal_pair_gm; the dword form only where relations span several runs)
    const bool g2_ok = (k & 3) == 0 && (reinterpret_cast<uintptr_t>(m->tables[0]) & 15) == 0 && switch_value("RESCAL_G2") != 0;
    const bool gemm_g = rows && (g_sw >= 0 ? g_sw == 1 : (g2_ok || n >= 128 * R));
#define KGE_RP(VK_)                                                                                                              \
    {                                                                                                                            \
        if (!rows)                                                                                                               \
        hipLaunchKernelGGL(k_rescal_pair<VK_>, dim3(tiles), dim3(1024), lds, s, m->tables[0], m->tables[1], m->grads[0], m->grads[1], ph, pt, \
                           nh, nt, g.offsets, g.tile_off, g.tile_rel, g.perm, (int)R, k, margin, loss, touched, ds);              \
        if (ds && !gemm_g)                                                                                                       \
            hipLaunchKernelGGL(k_rescal_pair_gm<VK_>, dim3(tiles), dim3(1024), lds_gm, s, m->tables[0], m->grads[1], ph, pt, nh, nt, g.offsets, \
                               g.tile_off, g.tile_rel, g.perm, (int)R, k, ds);                                                   \
    }
    if ((k & 3) == 0) KGE_RP(4) else KGE_RP(2)
#undef KGE_RP
    if (gemm_g) {
#define KGE_RG(I) case I: launch_rescal_g<I>(m, ph, pt, nh, nt, g, tiles, (int)R, k, ds, s); break;
        switch ((k + 15) / 16) { KGE_RG(1) KGE_RG(2) KGE_RG(3) KGE_RG(4) KGE_RG(5) KGE_RG(6) KGE_RG(7) KGE_RG(8) KGE_RG(9) KGE_RG(10) KGE_RG(11) KGE_RG(12) KGE_RG(13) }
#undef KGE_RG
    }
    return check_launch("k_rescal_pair");
}

// ---- hinge coefficients for models scored by separate forward/backward launches (RESCAL, NTN):
// in: energies.  out (in place): pos <- dL/dpos, neg <- dL/dneg; loss += sum max(0, pos + margin - neg)
__global__ __launch_bounds__(256) void k_hinge_coeffs(float* __restrict__ pos, float* __restrict__ neg, int64_t n,
                                                      float margin, float* __restrict__ loss) {
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = pos[i] + margin - neg[i];
        acc += fmaxf(v, 0.f);
        const float c = v > 0.f ? 1.f : (v == 0.f ? 0.5f : 0.f);
        pos[i] = c; neg[i] = -c;
    }
    block_accumulate_loss<1>(acc, 0, loss);
}

int launch_hinge_coeffs(float* pos, float* neg, int64_t n, float margin, float* loss, hipStream_t s) {
    int64_t b = (n + 255) / 256;
    if (b > 1024) b = 1024;
    hipLaunchKernelGGL(k_hinge_coeffs, dim3((unsigned)b), dim3(256), 0, s, pos, neg, n, margin, loss);
    return check_launch("k_hinge_coeffs");
}

}  // namespace kge
