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
    const int S = (d + 1) | 1;
    const size_t lds = (size_t)(2 * NT * S + 4 * NT) * sizeof(float);
    hipLaunchKernelGGL(k_ntn_bil, dim3(tiles, (unsigned)kr), dim3(256), lds, s, m->tables[5], n, d, kr, w);
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
