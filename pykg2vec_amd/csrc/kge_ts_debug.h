// kge_ts_debug.h -- per-workgroup timestamps for timeline studies (profiles/r04_experiments.md sections 4 and 7).  A build with
// -DKGE_TS_DEBUG (make -C pykg2vec_amd/csrc ts: ../../tools/_libs/ts.so) records, for every workgroup of an instrumented kernel,
// wall_clock64() (100 MHz) at entry and exit, HW_REG_XCC_ID / HW_REG_HW_ID (XCD, SE, CU of the workgroup's first wave) and a tag;
// kge_ts_dump_<unit> copies the records of the LAST launch of each slot to the host, tools/wg_timeline.py turns them into lifetimes,
// start-time quantiles, resident workgroups per CU and in-flight counts.  In the product build the macros expand to nothing and the
// dump functions do not exist (the tool then says which build it needs).  One buffer per translation unit (no relocatable device
// code in this library): define KGE_TS_UNIT (pull, transr, ...) before including.
#pragma once
#ifdef KGE_TS_DEBUG
#include <hip/hip_runtime.h>
namespace kge { namespace {
constexpr int kTsSlots = 2, kTsBlocks = 16384;
__device__ unsigned long long g_ts[kTsSlots][kTsBlocks][4];
} }
#define KGE_TS_CAT_(a, b) a##b
#define KGE_TS_CAT(a, b) KGE_TS_CAT_(a, b)
extern "C" int KGE_TS_CAT(kge_ts_dump_, KGE_TS_UNIT)(int slot, unsigned long long* host, long long max_records) {
    if (slot < 0 || slot >= kge::kTsSlots || !host || max_records < 0) return -1;
    const size_t n = (size_t)(max_records < kge::kTsBlocks ? max_records : kge::kTsBlocks);
    int rc = (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(kge::g_ts), n * 4 * sizeof(unsigned long long),
                                      (size_t)slot * kge::kTsBlocks * 4 * sizeof(unsigned long long));
    void* p = nullptr;   // cleared for the next study: a record with entry time 0 was not written by the last launch
    if (!rc && hipGetSymbolAddress(&p, HIP_SYMBOL(kge::g_ts)) == hipSuccess)
        rc = (int)hipMemset((char*)p + (size_t)slot * kge::kTsBlocks * 4 * sizeof(unsigned long long), 0, (size_t)kge::kTsBlocks * 4 * sizeof(unsigned long long));
    return rc;
}
#define KGE_TS_BEGIN(slot)                                                                                                      \
    { const unsigned long long ts0_ = wall_clock64();                                                                          \
      if (threadIdx.x == 0 && blockIdx.x < kge::kTsBlocks && blockIdx.y == 0) {                                                 \
          unsigned long long* r_ = kge::g_ts[slot][blockIdx.x];                                                                 \
          r_[0] = ts0_; r_[1] = ts0_;                                                                                           \
          r_[2] = (unsigned long long)__builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) |                                  \
                  ((unsigned long long)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)) << 8);                            \
          r_[3] = 0; } }
#define KGE_TS_END(slot, tag)                                                                                                   \
    { __syncthreads();                                                                                                          \
      if (threadIdx.x == 0 && blockIdx.x < kge::kTsBlocks && blockIdx.y == 0) {                                                 \
          kge::g_ts[slot][blockIdx.x][1] = wall_clock64(); kge::g_ts[slot][blockIdx.x][3] = (unsigned long long)(tag); } }
#else
#define KGE_TS_BEGIN(slot)
#define KGE_TS_END(slot, tag)
#endif
