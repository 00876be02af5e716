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

namespace kge {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k_rel_hist(IdSplit r, int64_t n, int* __restrict__ counts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(counts + r.at(i), 1);
}

// single block: exclusive scans of counts and of ceil(counts / TILE)
__global__ __launch_bounds__(256) void k_rel_scan(const int* __restrict__ counts, int R, int* __restrict__ offsets,
                                                  int* __restrict__ tile_off) {
    __shared__ int s_a[256], s_b[256];
    int run_a = 0, run_b = 0;
    for (int base = 0; base < R; base += 256) {
        const int idx = base + threadIdx.x;
        const int c = idx < R ? counts[idx] : 0;
        const int tl = (c + TILE - 1) / TILE;
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
                              int* __restrict__ tile_rel) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int rel = (int)r.at(i);
    const int local = atomicAdd(cursor + rel, 1);
    perm[offsets[rel] + local] = (int)i;
    if (local % TILE == 0) tile_rel[tile_off[rel] + local / TILE] = rel;  // the first row of a tile names its relation
}

int group_by_relation(const int64_t* r, int64_t n, int64_t R, const GroupWs& g, hipStream_t s) {
    return group_by_relation_split(id_whole(r, n), n, R, g, s);
}

int group_by_relation_split(IdSplit r, int64_t n, int64_t R, const GroupWs& g, hipStream_t s) {
    hipError_t e = hipMemsetAsync(g.counts, 0, (size_t)2 * (R + 1) * sizeof(int), s);  // counts + cursor
    if (e != hipSuccess) { set_error("rescal grouping memset: %s", hipGetErrorString(e)); return -2; }
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_rel_hist, dim3(nb), dim3(256), 0, s, r, n, g.counts);
    hipLaunchKernelGGL(k_rel_scan, dim3(1), dim3(256), 0, s, g.counts, (int)R, g.offsets, g.tile_off);
    hipLaunchKernelGGL(k_rel_scatter, dim3(nb), dim3(256), 0, s, r, n, g.offsets, g.tile_off, g.cursor, g.perm, g.tile_rel);
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
    for (int idx = threadIdx.x; idx < TILE * k; idx += 256) {  // coalesced row gathers into LDS
        const int i = idx / k, c = idx - i * k;
        const bool ok = i < cnt;
        sT[i * S + c] = ok ? ent[sTid[i] * k + c] : 0.f;
        sH[i * S + c] = ok ? ent[sHid[i] * k + c] : 0.f;
    }
    __syncthreads();
    const float* M = relm + (int64_t)rel * k * k;
    const int ntile = (k + 31) / 32;
    const int li = lane & 31, lk = lane >> 5;

    // Work units of this tile, dealt round-robin to the 4*gridDim.y waves that share it:
    //   [0, ntile)              U = T M^T  column tile `at`   (forward score / grad_h)
    //   [ntile, 2 ntile)        V = H M    column tile `bt`   (grad_t)
    //   [2 ntile, 2 ntile + ntile^2)   G = (ds*H)^T T  tile (at, bt)   (grad_M)
    const int n_units = MODE == 0 ? ntile : 2 * ntile + ntile * ntile;
    float* gM = MODE == 1 ? g_rel + (int64_t)rel * k * k : nullptr;
    for (int u = blockIdx.y * 4 + wave; u < n_units; u += 4 * gridDim.y) {
        if (MODE == 1 && u < ntile) {  // ---- U[i][a] = sum_b T[i][b] M[a][b] ;  grad_h = -ds U
            const int a = u * 32 + li;
            f32x16 acc = {0};
            // (lane a reads M[a][b]: a 4-byte gather over 32 rows.  Staging the rows through LDS and a transposed copy of M
            // were both measured and did not pay: the unit is bound by the dependent-load latency, not by the sectors)
            for (int k0 = 0; k0 < k; k0 += 16) {  // fixed-trip inner loop: 8 operand loads in flight
                float av[8], bv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int b = k0 + 2 * q + lk;
                    av[q] = b < k ? sT[li * S + b] : 0.f;
                    bv[q] = (a < k && b < k) ? M[(int64_t)a * k + b] : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc, 0, 0, 0);
            }
            if (a < k) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int i = (reg & 3) + 8 * (reg >> 2) + 4 * lk;
                    if (i < cnt && sDs[i] != 0.f) unsafeAtomicAdd(g_ent + sHid[i] * k + a, -sDs[i] * acc[reg]);   // grad_h = -ds U
                }
            }
        } else if (MODE == 0 || u < 2 * ntile) {  // ---- V[i][b] = sum_a H[i][a] M[a][b]  (M rows read coalesced)
            // MODE 0: score_i = -V_i . t_i ;  MODE 1: grad_t = -ds V
            const int b = (MODE == 0 ? u : u - ntile) * 32 + li;
            f32x16 acc = {0};
            for (int k0 = 0; k0 < k; k0 += 16) {
                float av[8], bv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int a = k0 + 2 * u + lk;
                    av[u] = a < k ? sH[li * S + a] : 0.f;
                    bv[u] = (a < k && b < k) ? M[(int64_t)a * k + b] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
            }
            if (b < k) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int i = (reg & 3) + 8 * (reg >> 2) + 4 * lk;
                    if (MODE == 0) {
                        atomicAdd(&sSc[i], sT[i * S + b] * acc[reg]);          // LDS atomic: score_i += V[i][b] t_i[b]
                    } else if (i < cnt && sDs[i] != 0.f) {
                        unsafeAtomicAdd(g_ent + sTid[i] * k + b, -sDs[i] * acc[reg]);
                    }
                }
            }
        } else {  // ---- G[a][b] = sum_i ds_i H[i][a] T[i][b] ;  grad_M = -G
            const int tl = u - 2 * ntile;
            const int at = tl / ntile, bt = tl - at * ntile;
            const int a_in = at * 32 + li, b_in = bt * 32 + li;
            f32x16 acc = {0};
#pragma unroll 4
            for (int kk = 0; kk < TILE; kk += 2) {
                const int i = kk + lk;
                const float av = a_in < k ? sDs[i] * sH[i * S + a_in] : 0.f;
                const float bv = b_in < k ? sT[i * S + b_in] : 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
            }
            if (b_in < k) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int a = at * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lk;
                    if (a < k && acc[reg] != 0.f) unsafeAtomicAdd(gM + (int64_t)a * k + b_in, -acc[reg]);
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

static size_t rescal_lds_bytes(int k) {
    const int S = (k + 1) | 1;
    return (size_t)(2 * TILE * S + 2 * TILE) * sizeof(float) + (size_t)(TILE + (TILE & 1)) * sizeof(int) +
           (size_t)2 * TILE * sizeof(long long);
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
    if (!grouped) {  // the fused train step's backward reuses the grouping its forward left in this workspace
        int rc = group_by_relation_split(r, n, R, g, s);
        if (rc) return rc;
    }
    const unsigned max_tiles = (unsigned)group_max_tiles(R, n);  // surplus blocks exit
    const size_t lds = rescal_lds_bytes(k);
    const int ntile = (k + 31) / 32;
    if (mode == 0) {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute((const void*)k_rescal<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipError_t e = hipMemsetAsync(scores, 0, (size_t)n * sizeof(float), s);
        if (e != hipSuccess) { set_error("rescal: memset: %s", hipGetErrorString(e)); return -2; }
        hipLaunchKernelGGL(k_rescal<0>, dim3(max_tiles, (unsigned)((ntile + 3) / 4)), dim3(256), lds, s, m->tables[0], m->tables[1], nullptr, nullptr, h, t,
                           g.offsets, g.tile_off, g.tile_rel, g.perm, (int)R, k, nullptr, scores);
    } else {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute((const void*)k_rescal<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        const int units = 2 * ntile + ntile * ntile;
        const unsigned ysplit = (unsigned)min(16, (units + 3) / 4);  // ~1 unit per wave
        hipLaunchKernelGGL(k_rescal<1>, dim3(max_tiles, ysplit), dim3(256), lds, s, m->tables[0], m->tables[1], m->grads[0],
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
    normalize_rows(ent, E, (int64_t)k, s);
    normalize_rows(rel, R, (int64_t)k * k, s, scratch, scratch_floats);
    return check_launch("k_row_normalize");
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
