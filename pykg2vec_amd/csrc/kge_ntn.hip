// kge_ntn.hip -- NTN (pykg2vec/models/pairwise.py:868-963) on the f32 matrix cores.
//
//   energy = - r^ . tanh( h^T W_s t^ + h^ M1 + t^ M2 + b ),  s = 1..k_r,  x^ = F.normalize(x)
//
// The reference materialises h.repeat(k_r,1,1) and runs two bmm's (2*B*k_r*d^2 flop, a [k_r,B,d] temporary); its
// autograd then produces a dense [k_r, d*d] gradient for the shared tensor W.  Here a batch is tiled 32 triples at a
// time and the bilinear term is a chain of small GEMMs on v_mfma_f32_32x32x2_f32 (exact fp32):
//   forward   X_s = H^ W_s  ([32,d]x[d,d]) with the row-dot against T^ fused into the MFMA epilogue  -> bil[n][s]
//   backward  gT^ += gz_s * (H^ W_s),  gH^ += gz_s * (T^ W_s^T),  gW_s = (gz_s * H^)^T T^   (GEMM over the batch)
// Everything else (normalisation, the two [d,k_r] linear maps, tanh, the r^ dot) is wave-per-row VALU work.
// Intermediates live in a caller-provided workspace of n*(4d + 3k_r + 6) floats.
//
// MFMA operand maps: A: lane l holds A[i=l&31][k=l>>5]; B: lane l holds B[k=l>>5][j=l&31];
// C/D: col j = lane&31, row i = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#include "kge_internal.h"
#include "kge_mfma_blocks.h"
#include <stdlib.h>

namespace kge {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NT = 32;       // triples per tile
constexpr int SPW = 1;       // slices per wave in the bilinear backward (1: most parallelism; the batch is small)

struct NtnWs {
    float *Hn, *Tn, *Rn, *inv, *flag, *Z, *GZ, *GH, *GT;
};
static size_t ntn_ws_floats(int64_t n, int d, int kr) { return (size_t)n * (4 * (size_t)d + 3 * (size_t)kr + 6); }
static NtnWs ntn_carve(void* ws, int64_t n, int d, int kr) {
    NtnWs w;
    float* p = (float*)ws;
    w.Hn = p; p += n * d;
    w.Tn = p; p += n * d;
    w.GH = p; p += n * d;
    w.GT = p; p += n * d;
    w.Rn = p; p += n * kr;
    w.Z = p; p += n * kr;
    w.GZ = p; p += n * kr;
    w.inv = p; p += 3 * n;
    w.flag = p;
    return w;
}
size_t ntn_workspace_bytes(const kge_model_desc* m, int64_t n) { return ntn_ws_floats(n, m->dim, m->rel_dim) * sizeof(float); }

// ---- 1. normalised rows: one wave per triple
__global__ __launch_bounds__(256) void k_ntn_prep(const float* __restrict__ ent, const float* __restrict__ rel,
                                                  IdSplit h, IdSplit r, IdSplit t, int64_t n, int d, int kr, NtnWs w) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const float* eh = ent + h.at(i) * d; const float* et = ent + t.at(i) * d; const float* er = rel + r.at(i) * kr;
    float nh = 0.f, nt = 0.f, nr = 0.f;
    for (int c = lane; c < d; c += 64) { nh = fmaf(eh[c], eh[c], nh); nt = fmaf(et[c], et[c], nt); }
    for (int c = lane; c < kr; c += 64) nr = fmaf(er[c], er[c], nr);
    nh = sqrtf(wave_sum(nh)); nt = sqrtf(wave_sum(nt)); nr = sqrtf(wave_sum(nr));
    const float ih = 1.0f / fmaxf(nh, kEpsNormalize), it = 1.0f / fmaxf(nt, kEpsNormalize), ir = 1.0f / fmaxf(nr, kEpsNormalize);
    for (int c = lane; c < d; c += 64) { w.Hn[i * d + c] = eh[c] * ih; w.Tn[i * d + c] = et[c] * it; }
    for (int c = lane; c < kr; c += 64) w.Rn[i * kr + c] = er[c] * ir;
    if (lane == 0) {
        w.inv[3 * i] = ih; w.inv[3 * i + 1] = it; w.inv[3 * i + 2] = ir;
        w.flag[3 * i] = nh > kEpsNormalize; w.flag[3 * i + 1] = nt > kEpsNormalize; w.flag[3 * i + 2] = nr > kEpsNormalize;
    }
}

__device__ __forceinline__ void stage_tile(float* sX, const float* __restrict__ X, int64_t row0, int cnt, int d, int S) {
    // four loads per thread in flight before the LDS stores (a load-store-load-store chain pays the global latency per element)
    for (int base = 0; base < NT * d; base += 4 * 256) {
        float v[4];
        int pos[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * 256 + threadIdx.x;
            const int i = idx / d, c = idx - i * d;
            pos[u] = idx < NT * d ? i * S + c : -1;
            v[u] = (idx < NT * d && i < cnt) ? X[(row0 + i) * d + c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (pos[u] >= 0) sX[pos[u]] = v[u];
    }
}

// ---- 2. bil[n][s] = h^_n^T W_s t^_n : block = (tile of 32 triples, one slice); the 32-wide column tiles of W_s are dealt
// to the four waves (a wave's chain of dependent operand round trips is d/16 long instead of 4 d/16)
__global__ __launch_bounds__(256) void k_ntn_bil(const float* __restrict__ W, int64_t n, int d, int kr, NtnWs w) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int S = (d + 1) | 1;
    float* sH = smem; float* sT = sH + NT * S; float* sB = sT + NT * S;  // sB[4][32]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lk = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * NT;
    const int cnt = (int)min((int64_t)NT, n - row0);
    const int s = blockIdx.y;
    stage_tile(sH, w.Hn, row0, cnt, d, S);
    stage_tile(sT, w.Tn, row0, cnt, d, S);
    if (threadIdx.x < NT) sB[threadIdx.x] = 0.f;
    __syncthreads();
    {
        const float* Ws = W + (int64_t)s * d * d;
        float part[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) part[q] = 0.f;
        for (int j0 = wave * 32; j0 < d; j0 += 128) {
            const int j = j0 + li;
            f32x16 acc = {0};
            for (int k0 = 0; k0 < d; k0 += 16) {  // fixed-trip inner loop: 8 operand loads in flight per MFMA chain
                float av[8], bv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i2 = k0 + 2 * u + lk;
                    av[u] = i2 < d ? sH[li * S + i2] : 0.f;
                    bv[u] = (i2 < d && j < d) ? Ws[(int64_t)i2 * d + j] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
            }
            if (j < d) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int row = (q & 3) + 8 * (q >> 2) + 4 * lk;
                    part[q] = fmaf(acc[q], sT[row * S + j], part[q]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = (q & 3) + 8 * (q >> 2) + 4 * lk;
            atomicAdd(&sB[row], part[q]);  // LDS atomic: the four waves' column tiles add up
        }
    }
    __syncthreads();
    if (threadIdx.x < cnt) w.Z[(row0 + threadIdx.x) * kr + s] = sB[threadIdx.x];
}

// ---- 3. z = tanh(bil + h^ M1 + t^ M2 + b) ; score = - r^ . z
// one workgroup per triple: the d-long contraction is split over the four waves (each a quarter of c, partial sums meet
// in LDS) -- at the reference's B=128 a wave per triple would leave 7/8 of the SIMDs without a wave
__global__ __launch_bounds__(256) void k_ntn_finish(const float* __restrict__ M1, const float* __restrict__ M2,
                                                    const float* __restrict__ b, int64_t n, int d, int kr, NtnWs w,
                                                    float* __restrict__ scores) {
    __shared__ float s_part[4][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = blockIdx.x;
    if (i >= n) return;
    const float* hn = w.Hn + i * d; const float* tn = w.Tn + i * d;
    const int dq = ((d + 3) / 4 + 7) / 8 * 8;           // c-range of a wave, a multiple of the unroll
    const int c_lo = wave * dq, c_hi = min(d, c_lo + dq);
    float tot = 0.f;
    for (int sb = 0; sb < kr; sb += 128) {  // two slices per lane per pass
        const int s0 = sb + lane, s1 = sb + 64 + lane;
        const bool v0 = s0 < kr, v1 = s1 < kr;
        const int q0 = v0 ? s0 : kr - 1, q1 = v1 ? s1 : kr - 1;
        float lin0 = 0.f, lin1 = 0.f;
        for (int c0 = c_lo; c0 < c_hi; c0 += 8) {
            float m1a[8], m2a[8], m1b[8], m2b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = c0 + u < c_hi ? c0 + u : d - 1;
                m1a[u] = M1[(int64_t)c * kr + q0]; m2a[u] = M2[(int64_t)c * kr + q0];
                m1b[u] = M1[(int64_t)c * kr + q1]; m2b[u] = M2[(int64_t)c * kr + q1];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (c0 + u < c_hi) {
                    const float hv = hn[c0 + u], tv = tn[c0 + u];
                    lin0 = fmaf(hv, m1a[u], fmaf(tv, m2a[u], lin0));
                    lin1 = fmaf(hv, m1b[u], fmaf(tv, m2b[u], lin1));
                }
        }
        s_part[wave][lane] = lin0; s_part[wave][64 + lane] = lin1;
        __syncthreads();
        if (wave == 0) {
            if (v0) {
                const float lin = b[s0] + ((s_part[0][lane] + s_part[1][lane]) + (s_part[2][lane] + s_part[3][lane]));
                const float z = tanhf(w.Z[i * kr + s0] + lin);
                w.Z[i * kr + s0] = z;
                tot = fmaf(w.Rn[i * kr + s0], z, tot);
            }
            if (v1) {
                const float lin = b[s1] + ((s_part[0][64 + lane] + s_part[1][64 + lane]) + (s_part[2][64 + lane] + s_part[3][64 + lane]));
                const float z = tanhf(w.Z[i * kr + s1] + lin);
                w.Z[i * kr + s1] = z;
                tot = fmaf(w.Rn[i * kr + s1], z, tot);
            }
        }
        __syncthreads();
    }
    if (wave == 0) {
        tot = wave_sum(tot);
        if (lane == 0 && scores) scores[i] = -tot;
    }
}

// ---- 4. gz, relation-row gradient, linear parts of gH^/gT^           (one wave per triple)
__global__ __launch_bounds__(256) void k_ntn_gz(const float* __restrict__ M1, const float* __restrict__ M2,
                                                IdSplit r, const float* __restrict__ dscore,
                                                float* __restrict__ g_rel, int64_t n, int d, int kr, NtnWs w) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    float* sgz = smem + wave * kr;
    if (i < n) {
        const float ds = dscore[i];
        float dot = 0.f;
        for (int s = lane; s < kr; s += 64) {
            const float z = w.Z[i * kr + s], rn = w.Rn[i * kr + s];
            const float gz = -ds * rn * (1.f - z * z);
            w.GZ[i * kr + s] = gz;
            sgz[s] = gz;
            dot = fmaf(rn, -ds * z, dot);
        }
        dot = wave_sum(dot);
        const float ir = w.inv[3 * i + 2];
        const bool fr = w.flag[3 * i + 2] != 0.f;
        if (ds != 0.f) {
            float* gr = g_rel + r.at(i) * kr;
            for (int s = lane; s < kr; s += 64) {
                const float g = -ds * w.Z[i * kr + s];
                unsafeAtomicAdd(gr + s, fr ? (g - w.Rn[i * kr + s] * dot) * ir : g * ir);
            }
        }
    }
    __syncthreads();
    if (i < n) {
        for (int c = lane; c < d; c += 64) {
            float a = 0.f, bsum = 0.f;
#pragma unroll 8
            for (int s = 0; s < kr; ++s) {
                a = fmaf(sgz[s], M1[(int64_t)c * kr + s], a);
                bsum = fmaf(sgz[s], M2[(int64_t)c * kr + s], bsum);
            }
            w.GH[i * d + c] = a;
            w.GT[i * d + c] = bsum;
        }
    }
}

// ---- 5. gM1[c][s] += sum_n H^[n][c] gz[n][s] ; gM2 likewise ; gb[s] += sum_n gz[n][s]
__global__ __launch_bounds__(256) void k_ntn_small(float* __restrict__ gM1, float* __restrict__ gM2, float* __restrict__ gb,
                                                   int64_t n, int d, int kr, NtnWs w) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)(d + 1) * kr) return;
    const int c = (int)(idx / kr), s = (int)(idx - (int64_t)c * kr);
    if (c == d) {
        float a = 0.f;
#pragma unroll 8
        for (int64_t i = 0; i < n; ++i) a += w.GZ[i * kr + s];
        gb[s] += a;
    } else {
        float a = 0.f, b2 = 0.f;
#pragma unroll 8
        for (int64_t i = 0; i < n; ++i) {
            const float gz = w.GZ[i * kr + s];
            a = fmaf(w.Hn[i * d + c], gz, a);
            b2 = fmaf(w.Tn[i * d + c], gz, b2);
        }
        gM1[idx] += a;
        gM2[idx] += b2;
    }
}

// ---- 6. bilinear backward wrt the rows: GT += sum_s gz_s (H^ W_s),  GH += sum_s gz_s (T^ W_s^T)
// block = (tile, SPW slices); the 32-wide column tiles are dealt to the four waves (a wave's chain of dependent operand
// round trips is d/8 per slice instead of JT*d/8); each wave accumulates its column tiles over the slices, one atomic pass
template <int JT>  // JT = ceil(d / 32) <= 8
__global__ __launch_bounds__(256) void k_ntn_bil_bwd(const float* __restrict__ W, int64_t n, int d, int kr, NtnWs w) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int S = (d + 1) | 1;
    float* sH = smem; float* sT = sH + NT * S;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lk = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * NT;
    const int cnt = (int)min((int64_t)NT, n - row0);
    stage_tile(sH, w.Hn, row0, cnt, d, S);
    stage_tile(sT, w.Tn, row0, cnt, d, S);
    __syncthreads();
    constexpr int JW = (JT + 3) / 4;  // column tiles per wave
    float gt[JW][16], gh[JW][16];
#pragma unroll
    for (int a = 0; a < JW; ++a)
#pragma unroll
        for (int q = 0; q < 16; ++q) { gt[a][q] = 0.f; gh[a][q] = 0.f; }
    const int s_base = blockIdx.y * SPW;
    for (int ss = 0; ss < SPW; ++ss) {
        const int s = s_base + ss;
        if (s >= kr) break;
        const float* Ws = W + (int64_t)s * d * d;
        float gzr[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = (q & 3) + 8 * (q >> 2) + 4 * lk;
            gzr[q] = row < cnt ? w.GZ[(row0 + row) * kr + s] : 0.f;
        }
#pragma unroll
        for (int a = 0; a < JW; ++a) {
            if (wave + 4 * a >= JT) break;  // wave-uniform
            const int col = (wave + 4 * a) * 32 + li;
            f32x16 accx = {0}, accy = {0};
            for (int k0 = 0; k0 < d; k0 += 8) {
                float ah[4], at[4], bx[4], by[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int c = k0 + 2 * u + lk;
                    const bool ok = c < d && col < d;
                    ah[u] = c < d ? sH[li * S + c] : 0.f;
                    at[u] = c < d ? sT[li * S + c] : 0.f;
                    bx[u] = ok ? Ws[(int64_t)c * d + col] : 0.f;     // X = H^ W_s    : B[k=c][j=col] = W[c][col]
                    by[u] = ok ? Ws[(int64_t)col * d + c] : 0.f;     // Y = T^ W_s^T  : B[k=c][j=col] = W[col][c]
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    accx = __builtin_amdgcn_mfma_f32_32x32x2f32(ah[u], bx[u], accx, 0, 0, 0);
                    accy = __builtin_amdgcn_mfma_f32_32x32x2f32(at[u], by[u], accy, 0, 0, 0);
                }
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                gt[a][q] = fmaf(gzr[q], accx[q], gt[a][q]);
                gh[a][q] = fmaf(gzr[q], accy[q], gh[a][q]);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < JW; ++a) {
        const int col = (wave + 4 * a) * 32 + li;
        if (wave + 4 * a < JT && col < d) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = (q & 3) + 8 * (q >> 2) + 4 * lk;
                if (row < cnt) {
                    if (gt[a][q] != 0.f) unsafeAtomicAdd(w.GT + (row0 + row) * d + col, gt[a][q]);
                    if (gh[a][q] != 0.f) unsafeAtomicAdd(w.GH + (row0 + row) * d + col, gh[a][q]);
                }
            }
        }
    }
}

// ---- 7. gW[s][i][j] += sum_n gz[n][s] H^[n][i] T^[n][j] : one wave per (s, 32x32 tile), K = batch
__global__ __launch_bounds__(256) void k_ntn_gw(float* __restrict__ gW, int64_t n, int d, int kr, NtnWs w) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lk = lane >> 5;
    const int jt = (d + 31) / 32;
    const int tile = blockIdx.x * 4 + wave;
    const int s = blockIdx.y;
    if (tile >= jt * jt) return;
    const int i0 = (tile / jt) * 32, j0 = (tile % jt) * 32;
    const int ia = i0 + li, jb = j0 + li;
    f32x16 acc = {0};
    for (int64_t k0 = 0; k0 < n; k0 += 16) {
        float av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t row = k0 + 2 * u + lk;
            const bool ok = row < n;
            av[u] = (ok && ia < d) ? w.GZ[row * kr + s] * w.Hn[row * d + ia] : 0.f;
            bv[u] = (ok && jb < d) ? w.Tn[row * d + jb] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
    }
    if (jb < d) {
        float* g = gW + (int64_t)s * d * d;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int i = i0 + (q & 3) + 8 * (q >> 2) + 4 * lk;
            if (i < d) g[(int64_t)i * d + jb] += acc[q];  // single writer per element in this launch
        }
    }
}

// ---- 8. normalisation backward + scatter of the entity-row gradients    (one wave per triple)
__global__ __launch_bounds__(256) void k_ntn_scatter(IdSplit h, IdSplit t,
                                                     const float* __restrict__ dscore, float* __restrict__ g_ent,
                                                     int64_t n, int d, NtnWs w) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n || dscore[i] == 0.f) return;
    float dh = 0.f, dt = 0.f;
    for (int c = lane; c < d; c += 64) {
        dh = fmaf(w.Hn[i * d + c], w.GH[i * d + c], dh);
        dt = fmaf(w.Tn[i * d + c], w.GT[i * d + c], dt);
    }
    dh = wave_sum(dh); dt = wave_sum(dt);
    const float ih = w.inv[3 * i], it = w.inv[3 * i + 1];
    const bool fh = w.flag[3 * i] != 0.f, ft = w.flag[3 * i + 1] != 0.f;
    float* gh = g_ent + h.at(i) * d; float* gt = g_ent + t.at(i) * d;
    for (int c = lane; c < d; c += 64) {
        const float a = w.GH[i * d + c], b2 = w.GT[i * d + c];
        unsafeAtomicAdd(gh + c, fh ? (a - w.Hn[i * d + c] * dh) * ih : a * ih);
        unsafeAtomicAdd(gt + c, ft ? (b2 - w.Tn[i * d + c] * dt) * it : b2 * it);
    }
}

// ================================================================== large batches: the same contractions as dense GEMMs
// W is ONE [k_r, d, d] tensor shared by all relations (pairwise.py:884-886), so for a large batch the bilinear term is a
// plain GEMM with the batch as M:  X = H^ [n x d] * W_flat [d x (k_r d)],  bil[n][s] = <X[n][s,:], T^[n]>  (forward),
// GT^ = sum_s gz_s o (H^ W_s), GH^ = sum_s gz_s o (T^ W_s^T) (K = k_r d with the row scaling folded into the A operand), and
// gW_s = (gz_s o H^)^T T^ (K = the batch).  The tile-per-(32 triples, slice) kernels above re-read their operands from
// global memory per MFMA and leave gW to one wave per 32 x 32 tile walking the whole batch; these keep the batch-side
// operand of 128 triples in REGISTERS across all slices, stream W (or the batch, for gW) through LDS in 16-deep slabs and
// run v_mfma_f32_16x16x4_f32 with 2 x NBJ accumulator blocks per wave.
constexpr int kBigRows = 128;   // triples per workgroup of k_ntn_rows: 4 waves x 2 row blocks of 16

// MODE 0: Z[n][s] = bil[n][s] (forward);  1: GT^ += sum_s gz_s (H^ W_s);  2: GH^ += sum_s gz_s (T^ W_s^T).
// grid = (tiles of 128 triples, slice groups): a workgroup walks slices [s_lo, s_hi); with more than one slice group the
// row gradients of the groups meet through float atomics (small batches only).
// (two workgroups per CU; the forward of the widest shape parks 64 KB of T^ in LDS: one)
template <int NBJ, int MODE>
__global__ __launch_bounds__(256, (MODE == 0 && NBJ == 8) ? 1 : 2) void k_ntn_rows(const float* __restrict__ W, int64_t n, int d, int kr, int s_per, NtnWs w) {
    constexpr int DP = 16 * NBJ, PITCH = DP + 4, NK = 4 * NBJ;
    __shared__ __attribute__((aligned(16))) float sW[2][16][PITCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l = lane & 15, lk = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * kBigRows + wave * 32;
    const int s_lo = blockIdx.y * s_per, s_hi = min(kr, s_lo + s_per);
    const float* __restrict__ X = MODE == 2 ? w.Tn : w.Hn;
    // the batch-side operand of the wave's 32 triples, for every k-step: A[row = 16 rb + l][k = 4 ks + lk]
    float a[2][NK];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const int64_t row = row0 + 16 * rb + l;
            const int k = 4 * ks + lk;
            a[rb][ks] = (row < n && k < d) ? X[row * d + k] : 0.f;
        }
    // forward: the T^ elements the accumulators meet in the epilogue (C layout: row = 16 rb + 4 lk + reg, column at(cb, l)),
    // parked in LDS in accumulator order -- one 16-byte read per accumulator block and slice instead of 8 NBJ live registers
    __shared__ __attribute__((aligned(16))) float sT[MODE == 0 ? 4 : 1][MODE == 0 ? 2 * NBJ : 1][64][4];
    if constexpr (MODE == 0) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < NBJ; ++cb) {
                float4 v;
                float* pv = reinterpret_cast<float*>(&v);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t row = row0 + 16 * rb + 4 * lk + q;
                    const int col = BlkMap<NBJ>::at(cb, l);
                    pv[q] = (row < n && col < d) ? w.Tn[row * d + col] : 0.f;
                }
                *reinterpret_cast<float4*>(&sT[wave][rb * NBJ + cb][lane][0]) = v;   // (read back by the same lane only)
            }
    }
    f32x4v acc[2][NBJ];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < NBJ; ++cb) acc[rb][cb] = f32x4v{0.f, 0.f, 0.f, 0.f};
    // slab (s, kb): rows k = 16 kb .. + 15 of B_s (B_s = W_s, or W_s^T for MODE 2), DP columns.  A thread's NBJ elements sit at a
    // fixed per-thread offset + a scalar that depends on (kb, u) only: slab row kq, columns c0 + 16 u
    //   MODE 0 / 1: kq = tid / 16, c0 = tid % 16: element W_s[16 kb + kq][c0 + 16 u]      (64 contiguous bytes per 16 threads)
    //   MODE 2    : kq = tid % 16, c0 = tid / 16: element W_s[c0 + 16 u][16 kb + kq]      (the transposed read, same 64 bytes)
    const int kq = MODE == 2 ? (threadIdx.x & 15) : (threadIdx.x >> 4), c0 = MODE == 2 ? (threadIdx.x >> 4) : (threadIdx.x & 15);
    const int toff = MODE == 2 ? c0 * d + kq : kq * d + c0;
    float st[NBJ];
    auto fetch = [&](int s, int kb) __attribute__((always_inline)) {
        const float* __restrict__ Ws = W + (int64_t)s * d * d + (MODE == 2 ? 16 * kb : 16 * kb * d);
#pragma unroll
        for (int u = 0; u < NBJ; ++u)
            st[u] = (16 * kb + kq < d && c0 + 16 * u < d) ? Ws[toff + (MODE == 2 ? 16 * u * d : 16 * u)] : 0.f;
    };
    auto stash = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NBJ; ++u) sW[buf][kq][c0 + 16 * u] = st[u];
    };
    auto gz_of = [&](int s, int rb) __attribute__((always_inline)) {
        const int64_t row = row0 + 16 * rb + l;
        return (row < n && s < s_hi) ? w.GZ[row * kr + s] : 0.f;
    };
    int buf = 0;
    if (s_lo < s_hi) fetch(s_lo, 0);
    float gn0 = 0.f, gn1 = 0.f;
    if constexpr (MODE != 0) { gn0 = gz_of(s_lo, 0); gn1 = gz_of(s_lo, 1); }
    for (int s = s_lo; s < s_hi; ++s) {
        const float g0 = gn0, g1 = gn1;
        if constexpr (MODE != 0) { gn0 = gz_of(s + 1, 0); gn1 = gz_of(s + 1, 1); }   // next slice's row scales, a slice ahead
#pragma unroll
        for (int kb = 0; kb < NBJ; ++kb) {
            stash(buf);
            __syncthreads();   // slab (s, kb) is in LDS; everybody finished reading the buffer that is written next
            if (kb + 1 < NBJ) fetch(s, kb + 1);
            else if (s + 1 < s_hi) fetch(s + 1, 0);
            // operands of k-step kk + 1 are read from LDS before the MFMAs of k-step kk are issued (register double buffer)
            float b[2][NBJ];
            read_blocks<NBJ>(&sW[buf][lk][0], l, b[0]);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                if (kk + 1 < 4) read_blocks<NBJ>(&sW[buf][4 * (kk + 1) + lk][0], l, b[(kk + 1) & 1]);
                KGE_KEEP_READS_AHEAD();
                float a0 = a[0][4 * kb + kk], a1 = a[1][4 * kb + kk];
                if constexpr (MODE != 0) { a0 *= g0; a1 *= g1; }
#pragma unroll
                for (int cb = 0; cb < NBJ; ++cb) {
                    acc[0][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b[kk & 1][cb], acc[0][cb], 0, 0, 0);
                    acc[1][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b[kk & 1][cb], acc[1][cb], 0, 0, 0);
                }
            }
            buf ^= 1;
        }
        if constexpr (MODE == 0) {   // bil[row][s] = <X_s[row], T^[row]>: per-lane partial over its columns, then the 16 lanes of a row
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                float p[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int cb = 0; cb < NBJ; ++cb) {
                    const float4 t4 = *reinterpret_cast<const float4*>(&sT[wave][rb * NBJ + cb][lane][0]);
                    p[0] = fmaf(acc[rb][cb][0], t4.x, p[0]); p[1] = fmaf(acc[rb][cb][1], t4.y, p[1]);
                    p[2] = fmaf(acc[rb][cb][2], t4.z, p[2]); p[3] = fmaf(acc[rb][cb][3], t4.w, p[3]);
                    acc[rb][cb] = f32x4v{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float t = p[q];
                    t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 8, 64);
                    const int64_t row = row0 + 16 * rb + 4 * lk + q;
                    if (l == 0 && row < n) w.Z[row * kr + s] = t;
                }
            }
        }
    }
    if constexpr (MODE != 0) {
        float* __restrict__ out = MODE == 1 ? w.GT : w.GH;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < NBJ; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t row = row0 + 16 * rb + 4 * lk + q;
                    const int col = BlkMap<NBJ>::at(cb, l);
                    if (row < n && col < d) {
                        if (gridDim.y == 1) out[row * d + col] += acc[rb][cb][q];   // single writer
                        else unsafeAtomicAdd(out + row * d + col, acc[rb][cb][q]);
                    }
                }
    }
}

// out[i][j] += sum over the rows n of a batch chunk of A[n][i] B[n][j] (K = the batch), split-K over blockIdx.x:
//   MODE 0: gW_s (s = blockIdx.y): A = gz[:, s] o H^, B = T^;   MODE 1: gM1: A = H^, B = gz;   MODE 2: gM2: A = T^, B = gz.
// Row block b of the output goes to wave b % 4 (natural row order: an A operand is one 4-byte LDS read per block).
template <int NBI, int NBJ, int MODE>
__global__ __launch_bounds__(256) void k_ntn_outer(float* __restrict__ out, int64_t n, int d, int kr, int64_t rows_per, NtnWs w) {
    constexpr int DPI = 16 * NBI, DPJ = 16 * NBJ, PA = DPI + 4, PB = DPJ + 4;
    constexpr int RBW = (NBI + 3) / 4;   // row blocks per wave
    constexpr int SK = 16, SH = SK / 16;   // batch rows per slab (32 -- half the barriers -- measured 3 % slower for gW)
    __shared__ __attribute__((aligned(16))) float sA[2][SK][PA], sB[2][SK][PB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l = lane & 15, lk = lane >> 4;
    const int64_t n0 = (int64_t)blockIdx.x * rows_per, n1 = min(n, n0 + rows_per);
    const int s = blockIdx.y;
    const float* __restrict__ Asrc = MODE == 2 ? w.Tn : w.Hn;
    const float* __restrict__ Bsrc = MODE == 0 ? w.Tn : w.GZ;
    const int ni = d, nj = MODE == 0 ? d : kr;   // rows / columns of the output
    f32x4v acc[RBW][NBJ];
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
        for (int cb = 0; cb < NBJ; ++cb) acc[rb][cb] = f32x4v{0.f, 0.f, 0.f, 0.f};
    // staging roles: slab rows kq + 16 h (batch rows), columns c0 + 16 u
    const int kq = threadIdx.x >> 4, c0 = threadIdx.x & 15;
    float sta[SH][NBI], stb[SH][NBJ], stg[SH];
    auto fetch = [&](int64_t r0) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < SH; ++h) {
            const int64_t row = r0 + kq + 16 * h;
            const bool live = row < n1;
            stg[h] = 1.f;
            if constexpr (MODE == 0) stg[h] = live ? w.GZ[row * kr + s] : 0.f;
            const float* __restrict__ ar = Asrc + row * d + c0;
            const float* __restrict__ br = Bsrc + row * nj + c0;
#pragma unroll
            for (int u = 0; u < NBI; ++u) sta[h][u] = (live && c0 + 16 * u < ni) ? ar[16 * u] : 0.f;
#pragma unroll
            for (int u = 0; u < NBJ; ++u) stb[h][u] = (live && c0 + 16 * u < nj) ? br[16 * u] : 0.f;
        }
    };
    auto stash = [&](int buf) __attribute__((always_inline)) {   // (the row scale is applied HERE: a multiply inside fetch would wait for the loads)
#pragma unroll
        for (int h = 0; h < SH; ++h) {
#pragma unroll
            for (int u = 0; u < NBI; ++u) sA[buf][kq + 16 * h][c0 + 16 * u] = MODE == 0 ? sta[h][u] * stg[h] : sta[h][u];
#pragma unroll
            for (int u = 0; u < NBJ; ++u) sB[buf][kq + 16 * h][c0 + 16 * u] = stb[h][u];
        }
    };
    int buf = 0;
    if (n0 < n1) fetch(n0);
    for (int64_t r0 = n0; r0 < n1; r0 += SK) {
        stash(buf);
        __syncthreads();
        if (r0 + SK < n1) fetch(r0 + SK);
        // operands of k-step kk + 1 are read from LDS before the MFMAs of k-step kk are issued (register double buffer)
        float b[2][NBJ], av[2][RBW];
        auto operands = [&](int kk, int slot) __attribute__((always_inline)) {
            read_blocks<NBJ>(&sB[buf][4 * kk + lk][0], l, b[slot]);
#pragma unroll
            for (int rb = 0; rb < RBW; ++rb) av[slot][rb] = wave + 4 * rb < NBI ? sA[buf][4 * kk + lk][16 * (wave + 4 * rb) + l] : 0.f;
        };
        operands(0, 0);
#pragma unroll
        for (int kk = 0; kk < SK / 4; ++kk) {
            if (kk + 1 < SK / 4) operands(kk + 1, (kk + 1) & 1);
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
    float* __restrict__ o = out + (MODE == 0 ? (int64_t)s * d * d : 0);
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb) {
        if (wave + 4 * rb >= NBI) continue;
#pragma unroll
        for (int cb = 0; cb < NBJ; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = 16 * (wave + 4 * rb) + 4 * lk + q, j = BlkMap<NBJ>::at(cb, l);
                if (i < ni && j < nj) {
                    if (gridDim.x == 1) o[(int64_t)i * nj + j] += acc[rb][cb][q];
                    else unsafeAtomicAdd(o + (int64_t)i * nj + j, acc[rb][cb][q]);
                }
            }
    }
}

// gb[s] += sum_n gz[n][s]: 128 rows per workgroup, thread = (slice, row phase)
__global__ __launch_bounds__(256) void k_ntn_gb(float* __restrict__ gb, int64_t n, int kr, NtnWs w) {
    const int64_t r0 = (int64_t)blockIdx.x * 128, r1 = min(n, r0 + 128);
    for (int s = threadIdx.x; s < kr; s += 256) {
        double t = 0.0;   // (the workgroup's 128 terms without rounding; the float atomics then add n / 128 partials)
#pragma unroll 8
        for (int64_t r = r0; r < r1; ++r) t += (double)w.GZ[r * kr + s];
        unsafeAtomicAdd(gb + s, (float)t);
    }
}

// The two [d, k_r] linear maps as GEMMs with the batch as M (NB = blocks of 16 covering max(d, k_r)):
//   MODE 0 (forward):  Z[n][s] += (H^ M1)[n][s] + (T^ M2)[n][s]          two passes, K = d, B[k][col] = M[k][col]
//   MODE 1 (backward): out[n][c] = sum_s gz[n][s] M[c][s]                 K = k_r, B[k][col] = M[col][k]  (out = GH^ with M1, GT^ with M2)
template <int NB, int MODE>
__global__ __launch_bounds__(256, 2) void k_ntn_lin(const float* __restrict__ Ma, const float* __restrict__ Mb, float* __restrict__ out,
                                                    int64_t n, int d, int kr, NtnWs w) {
    constexpr int DP = 16 * NB, PITCH = DP + 4, NK = 4 * NB;
    __shared__ __attribute__((aligned(16))) float sW[2][16][PITCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l = lane & 15, lk = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * kBigRows + wave * 32;
    const int K = MODE == 0 ? d : kr, N = MODE == 0 ? kr : d;
    f32x4v acc[2][NB];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) acc[rb][cb] = f32x4v{0.f, 0.f, 0.f, 0.f};
    const int kq = MODE == 1 ? (threadIdx.x & 15) : (threadIdx.x >> 4), c0 = MODE == 1 ? (threadIdx.x >> 4) : (threadIdx.x & 15);
    const int toff = MODE == 1 ? c0 * kr + kq : kq * kr + c0;   // both matrices are [d][k_r] row-major
    int buf = 0;
    for (int pass = 0; pass < (MODE == 0 ? 2 : 1); ++pass) {
        const float* __restrict__ X = MODE == 1 ? w.GZ : pass == 0 ? w.Hn : w.Tn;
        const float* __restrict__ M = pass == 0 ? Ma : Mb;
        float a[2][NK];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                const int64_t row = row0 + 16 * rb + l;
                const int k = 4 * ks + lk;
                a[rb][ks] = (row < n && k < K) ? X[row * K + k] : 0.f;
            }
        float st[NB];
        auto fetch = [&](int kb) __attribute__((always_inline)) {
            const float* __restrict__ Ms = M + (MODE == 1 ? 16 * kb : 16 * kb * kr);
#pragma unroll
            for (int u = 0; u < NB; ++u)
                st[u] = (16 * kb + kq < K && c0 + 16 * u < N) ? Ms[toff + (MODE == 1 ? 16 * u * kr : 16 * u)] : 0.f;
        };
        fetch(0);
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
            for (int u = 0; u < NB; ++u) sW[buf][kq][c0 + 16 * u] = st[u];
            __syncthreads();
            if (kb + 1 < NB) fetch(kb + 1);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                float b[NB];
                read_blocks<NB>(&sW[buf][4 * kk + lk][0], l, b);
#pragma unroll
                for (int cb = 0; cb < NB; ++cb) {
                    acc[0][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][4 * kb + kk], b[cb], acc[0][cb], 0, 0, 0);
                    acc[1][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1][4 * kb + kk], b[cb], acc[1][cb], 0, 0, 0);
                }
            }
            buf ^= 1;
        }
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < NB; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t row = row0 + 16 * rb + 4 * lk + q;
                const int col = BlkMap<NB>::at(cb, l);
                if (row < n && col < N) {
                    if constexpr (MODE == 0) out[row * N + col] += acc[rb][cb][q];
                    else out[row * N + col] = acc[rb][cb][q];
                }
            }
}

// what is left of k_ntn_finish / k_ntn_gz once the linear maps are GEMMs: one wave per triple, element-wise over the slices
__global__ __launch_bounds__(256) void k_ntn_finish_ew(const float* __restrict__ b, int64_t n, int kr, NtnWs w, float* __restrict__ scores) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    float tot = 0.f;
    for (int s = lane; s < kr; s += 64) {
        const float z = tanhf(w.Z[i * kr + s] + b[s]);
        w.Z[i * kr + s] = z;
        tot = fmaf(w.Rn[i * kr + s], z, tot);
    }
    tot = wave_sum(tot);
    if (lane == 0 && scores) scores[i] = -tot;
}
__global__ __launch_bounds__(256) void k_ntn_gz_ew(IdSplit r, const float* __restrict__ dscore, float* __restrict__ g_rel, int64_t n, int kr, NtnWs w) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const float ds = dscore[i];
    float dot = 0.f;
    for (int s = lane; s < kr; s += 64) {
        const float z = w.Z[i * kr + s], rn = w.Rn[i * kr + s];
        w.GZ[i * kr + s] = -ds * rn * (1.f - z * z);
        dot = fmaf(rn, -ds * z, dot);
    }
    dot = wave_sum(dot);
    if (ds == 0.f) return;
    const float ir = w.inv[3 * i + 2];
    const bool fr = w.flag[3 * i + 2] != 0.f;
    float* gr = g_rel + r.at(i) * kr;
    for (int s = lane; s < kr; s += 64) {
        const float g = -ds * w.Z[i * kr + s];
        unsafeAtomicAdd(gr + s, fr ? (g - w.Rn[i * kr + s] * dot) * ir : g * ir);
    }
}

template <int MODE>
static void launch_ntn_lin(const float* Ma, const float* Mb, float* out, int64_t n, int d, int kr, const NtnWs& w, hipStream_t s) {
    const int nb = (max(d, kr) + 15) / 16;
    const dim3 grid((unsigned)((n + kBigRows - 1) / kBigRows));
#define KGE_NTN_LIN(J) case J: hipLaunchKernelGGL((k_ntn_lin<J, MODE>), grid, dim3(256), 0, s, Ma, Mb, out, n, d, kr, w); break;
    switch (nb) { KGE_NTN_LIN(1) KGE_NTN_LIN(2) KGE_NTN_LIN(3) KGE_NTN_LIN(4) KGE_NTN_LIN(5) KGE_NTN_LIN(6) KGE_NTN_LIN(7) KGE_NTN_LIN(8) }
#undef KGE_NTN_LIN
}

// the GEMM forms pay off from about a thousand rows on (slice groups fill the chip before the batch does); d and k_r up to 128
// (8 blocks of 16)
static bool ntn_big_ok(int64_t n, int d, int kr) {
    if (d > 128 || kr > 128) return false;
    const int force = switch_value("NTN_BIG");   // A/B and tests: 0 / 1
    if (force >= 0) return force == 1;
    return n >= 1024;   // profiles/r03_ntn_threshold.txt: 2 x 512 rows 590 -> 493 us per step, 2 x 128 rows 234 -> 373 us
}
static int ntn_slices_per_group(int64_t n, int kr) {
    const int64_t tiles = (n + kBigRows - 1) / kBigRows;
    int64_t groups = (512 + tiles - 1) / tiles;   // >= 512 workgroups (two per CU) where the batch alone does not give them
    if (groups > kr) groups = kr;
    if (groups < 1) groups = 1;
    return (int)((kr + groups - 1) / groups);
}

template <int MODE>
static void launch_ntn_rows(const float* W, int64_t n, int d, int kr, const NtnWs& w, hipStream_t s) {
    const int nbj = (d + 15) / 16;
    const int s_per = ntn_slices_per_group(n, kr);
    const dim3 grid((unsigned)((n + kBigRows - 1) / kBigRows), (unsigned)((kr + s_per - 1) / s_per));
#define KGE_NTN_ROWS(J) case J: hipLaunchKernelGGL((k_ntn_rows<J, MODE>), grid, dim3(256), 0, s, W, n, d, kr, s_per, w); break;
    switch (nbj) { KGE_NTN_ROWS(1) KGE_NTN_ROWS(2) KGE_NTN_ROWS(3) KGE_NTN_ROWS(4) KGE_NTN_ROWS(5) KGE_NTN_ROWS(6) KGE_NTN_ROWS(7) KGE_NTN_ROWS(8) }
#undef KGE_NTN_ROWS
}

template <int NBI, int MODE>
static void launch_ntn_outer_j(float* out, int64_t n, int d, int kr, int nbj, dim3 grid, int64_t rows_per, const NtnWs& w, hipStream_t s) {
#define KGE_NTN_OUTER(J) case J: hipLaunchKernelGGL((k_ntn_outer<NBI, J, MODE>), grid, dim3(256), 0, s, out, n, d, kr, rows_per, w); break;
    switch (nbj) { KGE_NTN_OUTER(1) KGE_NTN_OUTER(2) KGE_NTN_OUTER(3) KGE_NTN_OUTER(4) KGE_NTN_OUTER(5) KGE_NTN_OUTER(6) KGE_NTN_OUTER(7) KGE_NTN_OUTER(8) }
#undef KGE_NTN_OUTER
}
template <int MODE>
static void launch_ntn_outer(float* out, int64_t n, int d, int kr, const NtnWs& w, hipStream_t s) {
    const int nbi = (d + 15) / 16, nbj = ((MODE == 0 ? d : kr) + 15) / 16;
    const int64_t slices = MODE == 0 ? kr : 1;
    int64_t chunks = (1024 + slices - 1) / slices;               // >= 1 024 workgroups
    // blockIdx.x = chunk, and workgroups go to XCD (flat id % 8): with a multiple of 8 chunks every workgroup of a chunk -- all
    // slices, which stream the SAME batch rows -- runs on one XCD and shares its L2
    if (chunks > 8) chunks = (chunks + 7) / 8 * 8;
    int64_t rows_per = ((n + chunks - 1) / chunks + 15) / 16 * 16;
    if (rows_per < 256) rows_per = 256;                          // (a chunk amortises its k^2 atomics over >= 16 slabs)
    chunks = (n + rows_per - 1) / rows_per;
    const dim3 grid((unsigned)chunks, (unsigned)slices);
#define KGE_NTN_OUTER_I(I) case I: launch_ntn_outer_j<I, MODE>(out, n, d, kr, nbj, grid, rows_per, w, s); break;
    switch (nbi) { KGE_NTN_OUTER_I(1) KGE_NTN_OUTER_I(2) KGE_NTN_OUTER_I(3) KGE_NTN_OUTER_I(4) KGE_NTN_OUTER_I(5) KGE_NTN_OUTER_I(6) KGE_NTN_OUTER_I(7) KGE_NTN_OUTER_I(8) }
#undef KGE_NTN_OUTER_I
}

// ---- host
static int ntn_check(const kge_model_desc* m, int64_t n, void* ws, size_t ws_bytes) {
    if (m->dim > 256 || m->rel_dim > 1024) { set_error("NTN: ent_hidden_size <= 256 and rel_hidden_size <= 1024 supported"); return -1; }
    if (!ws || ws_bytes < ntn_workspace_bytes(m, n)) {
        set_error("NTN needs a workspace of %zu bytes (kge_workspace_bytes)", ntn_workspace_bytes(m, n));
        return -1;
    }
    return 0;
}

static int ntn_forward_core(const kge_model_desc* m, IdSplit h, IdSplit r, IdSplit t, int64_t n,
                            const NtnWs& w, float* scores, hipStream_t s) {
    const int d = m->dim, kr = m->rel_dim;
    const unsigned rows4 = (unsigned)((n + 3) / 4), tiles = (unsigned)((n + NT - 1) / NT);
    hipLaunchKernelGGL(k_ntn_prep, dim3(rows4), dim3(256), 0, s, m->tables[0], m->tables[1], h, r, t, n, d, kr, w);
    if (ntn_big_ok(n, d, kr)) {
        launch_ntn_rows<0>(m->tables[5], n, d, kr, w, s);
        launch_ntn_lin<0>(m->tables[2], m->tables[3], w.Z, n, d, kr, w, s);
        hipLaunchKernelGGL(k_ntn_finish_ew, dim3(rows4), dim3(256), 0, s, m->tables[4], n, kr, w, scores);
        return check_launch("ntn forward");
    } else {
        const int S = (d + 1) | 1;
        const size_t lds = (size_t)(2 * NT * S + 4 * NT) * sizeof(float);
        hipLaunchKernelGGL(k_ntn_bil, dim3(tiles, (unsigned)kr), dim3(256), lds, s, m->tables[5], n, d, kr, w);
    }
    hipLaunchKernelGGL(k_ntn_finish, dim3((unsigned)n), dim3(256), 0, s, m->tables[2], m->tables[3], m->tables[4], n, d, kr, w,
                       scores);
    return check_launch("ntn forward");
}

int launch_ntn_forward(const kge_model_desc* m, const int64_t* h, const int64_t* r, const int64_t* t, int64_t n,
                       float* scores, void* ws, size_t ws_bytes, hipStream_t s) {
    if (ntn_check(m, n, ws, ws_bytes)) return -1;
    return ntn_forward_core(m, id_whole(h, n), id_whole(r, n), id_whole(t, n), n, ntn_carve(ws, n, m->dim, m->rel_dim), scores, s);
}

// positives and negatives of the fused pairwise step as ONE batch of 2n triples: 3 + 5 launches per step instead of 6 + 10
// (the workspace is linear in n, so the two per-side workspaces of the fused step hold it)
int launch_ntn_pair_forward(const kge_model_desc* m, const int64_t* ph, const int64_t* pr, const int64_t* pt, const int64_t* nh,
                            const int64_t* nr, const int64_t* nt, int64_t n, float* scores2, void* ws, size_t ws_bytes,
                            hipStream_t s) {
    if (ntn_check(m, 2 * n, ws, ws_bytes)) return -1;
    return ntn_forward_core(m, IdSplit{ph, nh, n}, IdSplit{pr, nr, n}, IdSplit{pt, nt, n}, 2 * n,
                            ntn_carve(ws, 2 * n, m->dim, m->rel_dim), scores2, s);
}

static int ntn_backward_core(const kge_model_desc* m, IdSplit h, IdSplit r, IdSplit t, int64_t n, const float* dscore, void* ws,
                             size_t ws_bytes, bool forward_in_ws, hipStream_t s);

int launch_ntn_backward(const kge_model_desc* m, const int64_t* h, const int64_t* r, const int64_t* t, int64_t n,
                        const float* dscore, void* ws, size_t ws_bytes, bool forward_in_ws, hipStream_t s) {
    return ntn_backward_core(m, id_whole(h, n), id_whole(r, n), id_whole(t, n), n, dscore, ws, ws_bytes, forward_in_ws, s);
}
int launch_ntn_pair_backward(const kge_model_desc* m, const int64_t* ph, const int64_t* pr, const int64_t* pt, const int64_t* nh,
                             const int64_t* nr, const int64_t* nt, int64_t n, const float* dscore2, void* ws, size_t ws_bytes,
                             hipStream_t s) {
    return ntn_backward_core(m, IdSplit{ph, nh, n}, IdSplit{pr, nr, n}, IdSplit{pt, nt, n}, 2 * n, dscore2, ws, ws_bytes, true, s);
}

static int ntn_backward_core(const kge_model_desc* m, IdSplit h, IdSplit r, IdSplit t, int64_t n, const float* dscore, void* ws,
                             size_t ws_bytes, bool forward_in_ws, hipStream_t s) {
    if (ntn_check(m, n, ws, ws_bytes)) return -1;
    const int d = m->dim, kr = m->rel_dim;
    const NtnWs w = ntn_carve(ws, n, d, kr);
    if (!forward_in_ws) {  // recompute H^, T^, R^, z  (the fused train step keeps the forward's workspace instead)
        int rc = ntn_forward_core(m, h, r, t, n, w, nullptr, s);
        if (rc) return rc;
    }
    const unsigned rows4 = (unsigned)((n + 3) / 4), tiles = (unsigned)((n + NT - 1) / NT);
    if (ntn_big_ok(n, d, kr)) {
        hipLaunchKernelGGL(k_ntn_gz_ew, dim3(rows4), dim3(256), 0, s, r, dscore, m->grads[1], n, kr, w);
        launch_ntn_lin<1>(m->tables[2], nullptr, w.GH, n, d, kr, w, s);
        launch_ntn_lin<1>(m->tables[3], nullptr, w.GT, n, d, kr, w, s);
        launch_ntn_outer<1>(m->grads[2], n, d, kr, w, s);
        launch_ntn_outer<2>(m->grads[3], n, d, kr, w, s);
        hipLaunchKernelGGL(k_ntn_gb, dim3((unsigned)((n + 127) / 128)), dim3(256), 0, s, m->grads[4], n, kr, w);
        launch_ntn_rows<1>(m->tables[5], n, d, kr, w, s);
        launch_ntn_rows<2>(m->tables[5], n, d, kr, w, s);
        launch_ntn_outer<0>(m->grads[5], n, d, kr, w, s);
        hipLaunchKernelGGL(k_ntn_scatter, dim3(rows4), dim3(256), 0, s, h, t, dscore, m->grads[0], n, d, w);
        return check_launch("ntn backward");
    }
    hipLaunchKernelGGL(k_ntn_gz, dim3(rows4), dim3(256), (size_t)4 * kr * sizeof(float), s, m->tables[2], m->tables[3], r,
                       dscore, m->grads[1], n, d, kr, w);
    hipLaunchKernelGGL(k_ntn_small, dim3((unsigned)(((int64_t)(d + 1) * kr + 255) / 256)), dim3(256), 0, s, m->grads[2],
                       m->grads[3], m->grads[4], n, d, kr, w);
    const int S = (d + 1) | 1;
    const size_t lds = (size_t)(2 * NT * S) * sizeof(float);
    const dim3 gb(tiles, (unsigned)((kr + SPW - 1) / SPW));
    const int JT = (d + 31) / 32;
#define KGE_NTN_BWD(J) case J: hipLaunchKernelGGL(k_ntn_bil_bwd<J>, gb, dim3(256), lds, s, m->tables[5], n, d, kr, w); break;
    switch (JT) {
        KGE_NTN_BWD(1) KGE_NTN_BWD(2) KGE_NTN_BWD(3) KGE_NTN_BWD(4) KGE_NTN_BWD(5) KGE_NTN_BWD(6) KGE_NTN_BWD(7) KGE_NTN_BWD(8)
    }
#undef KGE_NTN_BWD
    hipLaunchKernelGGL(k_ntn_gw, dim3((unsigned)((JT * JT + 3) / 4), (unsigned)kr), dim3(256), 0, s, m->grads[5], n, d, kr, w);
    hipLaunchKernelGGL(k_ntn_scatter, dim3(rows4), dim3(256), 0, s, h, t, dscore, m->grads[0], n, d, w);
    return check_launch("ntn backward");
}

// ---- NTN.get_reg (pairwise.py:962-963): lmbda * sqrt(sum over ALL parameters of w^2), dense
__global__ __launch_bounds__(256) void k_sumsq(const float* __restrict__ p, int64_t numel, float* __restrict__ out) {
    __shared__ float part[4];
    float a = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;  // four independent chains over 16-byte loads
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if ((((uintptr_t)p) & 15) == 0) {
        const int64_t nvec = numel / 4;
        for (int64_t i = tid; i < nvec; i += stride) {
            const float4 v = reinterpret_cast<const float4*>(p)[i];
            a = fmaf(v.x, v.x, a); a1 = fmaf(v.y, v.y, a1); a2 = fmaf(v.z, v.z, a2); a3 = fmaf(v.w, v.w, a3);
        }
        for (int64_t i = nvec * 4 + tid; i < numel; i += stride) a = fmaf(p[i], p[i], a);
    } else {
        for (int64_t i = tid; i < numel; i += stride) a = fmaf(p[i], p[i], a);
    }
    a = wave_sum((a + a1) + (a2 + a3));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}
__global__ __launch_bounds__(256) void k_l2_apply(const float* __restrict__ p, float* __restrict__ g, int64_t numel, float lmbda,
                                                  const float* __restrict__ sumsq, float* __restrict__ loss) {
    const float root = sqrtf(*sumsq);
    const float c = lmbda / root;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x)
        g[i] = fmaf(c, p[i], g[i]);
    if (blockIdx.x == 0 && threadIdx.x == 0) unsafeAtomicAdd(loss, lmbda * root);
}

int launch_l2norm_reg(const float* param, float* grad, int64_t numel, float lmbda, float* scratch, float* loss, hipStream_t s) {
    hipError_t e = hipMemsetAsync(scratch, 0, sizeof(float), s);
    if (e != hipSuccess) { set_error("kge_l2norm_reg: memset: %s", hipGetErrorString(e)); return -2; }
    int64_t b = (numel + 255) / 256;
    if (b > 2048) b = 2048;
    const int64_t br = b > 256 ? 256 : b;  // one same-address atomic per block costs ~12 ns each: keep them few
    hipLaunchKernelGGL(k_sumsq, dim3((unsigned)br), dim3(256), 0, s, param, numel, scratch);
    hipLaunchKernelGGL(k_l2_apply, dim3((unsigned)b), dim3(256), 0, s, param, grad, numel, lmbda, scratch, loss);
    return check_launch("kge_l2norm_reg");
}

}  // namespace kge
