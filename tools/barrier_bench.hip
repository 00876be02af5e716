// Dev tool: what does a device-wide barrier inside ONE persistent launch cost against the dependent kernel
// boundary it would replace?  The C1 step is two dependent launches (k_pull_eval -> k_pull_step) whose boundary + dispatch costs
// ~1.4 us each in the driver's clock; a persistent form pays one grid barrier per phase change instead.
//
//   pairs     N x (kernel A ; kernel B), back to back on one stream, each `blocks` workgroups of 256 threads doing `work` dependent
//             FMAs per thread: (total - N * 2 * body) / (2 N) = one kernel boundary
//   barrier   one launch of `grid` resident workgroups running 2 N phases of the same body separated by a sense-reversing barrier on
//             agent-scope atomics (arrive: fetch_add release; last arriver flips the generation; the rest spin on an acquire load)
//
// The spin is bounded (a stuck barrier sets a flag and every workgroup leaves): this tool must never hang the box.
// Build: hipcc --offload-arch=gfx950 -O3 tools/barrier_bench.hip -o tools/_libs/barrier_bench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float body(float v, int work) {
    for (int i = 0; i < work; ++i) v = fmaf(v, 1.0000001f, 1e-7f);
    return v;
}

__global__ __launch_bounds__(256) void k_phase(float* out, int work) {
    float v = body((float)threadIdx.x, work);
    if (v == 123.456f) out[blockIdx.x] = v;   // (never true: keeps the body alive)
}

struct Bar { unsigned count; unsigned gen; unsigned stuck; unsigned pad; };

__device__ __forceinline__ bool grid_barrier(Bar* b, unsigned nblocks, unsigned& my_gen) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const unsigned arrived = __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if (arrived == nblocks) {
            __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&b->gen, my_gen + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned spins = 0;
            while (__hip_atomic_load(&b->gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == my_gen) {
                if (++spins > (1u << 24) || __hip_atomic_load(&b->stuck, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(&b->stuck, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
    }
    my_gen += 1u;
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(256) void k_persistent(float* out, int work, int phases, Bar* b) {
    unsigned gen = 0;
    float v = (float)threadIdx.x;
    for (int p = 0; p < phases; ++p) {
        v = body(v, work);
        if (!grid_barrier(b, gridDim.x, gen)) break;
    }
    if (v == 123.456f) out[blockIdx.x] = v;
}

static float time_ms(hipStream_t s, void (*fn)(hipStream_t, void*), void* ctx) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    fn(s, ctx);   // warm
    CHECK(hipStreamSynchronize(s));
    CHECK(hipEventRecord(a, s));
    fn(s, ctx);
    CHECK(hipEventRecord(b, s));
    CHECK(hipStreamSynchronize(s));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms;
}

struct Ctx { float* out; int work, n, blocks, grid; Bar* bar; };

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 500;
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    float* out; Bar* bar;
    CHECK(hipMalloc(&out, 1 << 20));
    CHECK(hipMalloc(&bar, sizeof(Bar)));
    printf("# n = %d phase pairs; per-boundary / per-barrier figures in microseconds\n", n);
    for (int work : {0, 2000, 8000}) {
        for (int blocks : {256, 512, 2048}) {
            Ctx c{out, work, n, blocks, blocks > 512 ? 512 : blocks, bar};
            // one phase alone (N launches of ONE kernel would include boundaries too: time a single long launch of 2N bodies instead)
            auto body_only = [](hipStream_t st, void* p) { Ctx* c = (Ctx*)p; hipLaunchKernelGGL(k_phase, dim3(c->blocks), dim3(256), 0, st, c->out, c->work * 2 * c->n); };
            auto pairs = [](hipStream_t st, void* p) { Ctx* c = (Ctx*)p; for (int i = 0; i < 2 * c->n; ++i) hipLaunchKernelGGL(k_phase, dim3(c->blocks), dim3(256), 0, st, c->out, c->work); };
            auto persistent = [](hipStream_t st, void* p) {
                Ctx* c = (Ctx*)p;
                CHECK(hipMemsetAsync(c->bar, 0, sizeof(Bar), st));
                hipLaunchKernelGGL(k_persistent, dim3(c->grid), dim3(256), 0, st, c->out, c->work, 2 * c->n, c->bar);
            };
            const float t_body = time_ms(s, body_only, &c);
            const float t_pairs = time_ms(s, pairs, &c);
            float t_pers = -1.f;
            if (blocks <= 512) {   // the persistent grid must be resident: <= 2 workgroups of 256 threads per CU here
                t_pers = time_ms(s, persistent, &c);
                Bar h;
                CHECK(hipMemcpy(&h, bar, sizeof(Bar), hipMemcpyDeviceToHost));
                if (h.stuck) { printf("work %d blocks %d: barrier STUCK (bounded spin gave up)\n", work, blocks); continue; }
            }
            printf("work %5d blocks %4d: body %.3f us/phase | dependent launches %.3f us/phase -> boundary %.3f us | persistent %s\n", work, blocks,
                   t_body * 1e3 / (2 * n), t_pairs * 1e3 / (2 * n), (t_pairs - t_body) * 1e3 / (2 * n),
                   t_pers < 0 ? "n/a (grid not resident)" : "");
            if (t_pers >= 0) printf("                          persistent %.3f us/phase -> barrier %.3f us\n", t_pers * 1e3 / (2 * n), (t_pers - t_body) * 1e3 / (2 * n));
        }
    }
    return 0;
}
