// kge_relgroup.h -- device-side grouping of a batch by relation id (histogram -> scan -> scatter), shared by the
// relation-matrix models (RESCAL: kge_dense.hip, TransR: kge_transr.hip).  A workgroup then owns one
// (relation, 32-triple tile) and reads that relation's matrix once per tile instead of once per triple.
#pragma once
#include "kge_internal.h"

namespace kge {

constexpr int TILE = 32;  // triples per workgroup tile
constexpr int kPairTile = 16;    // pairs per row block of the pairwise RESCAL step: rows 0..15 the positives, 16..31 their negatives
constexpr int kSlabChunk = 64;   // pairs per (relation, chunk) tile of the slab form (kge_rescal_slab.hip): 4 row blocks

struct GroupWs {           // carved from the caller's workspace
    int* counts;           // [R]   triples per relation
    int* cursor;           // [R]   scatter cursors
    int* offsets;          // [R+1] first grouped position of each relation
    int* tile_off;         // [R+1] first tile of each relation
    int* perm;             // [n]   grouped position -> original row
    int* tile_rel;         // [n/TILE + R + 1] relation of each tile (filled for the tiles that exist)
};

inline int64_t group_max_tiles(int64_t R, int64_t n) { return n / TILE + R + 1; }  // >= sum_r ceil(n_r / TILE)
inline size_t group_ws_bytes(int64_t R, int64_t n) {
    return (size_t)(4 * (R + 1) + n + group_max_tiles(R, n) + 8) * sizeof(int);
}

inline GroupWs carve_group_ws(void* ws, int64_t R, int64_t n) {
    GroupWs g;
    int* p = (int*)ws;
    g.counts = p; p += R + 1;
    g.cursor = p; p += R + 1;
    g.offsets = p; p += R + 1;
    g.tile_off = p; p += R + 1;
    g.perm = p; p += n;
    g.tile_rel = p;
    return g;
}

// optional extra outputs of the one-launch grouping (pairwise RESCAL, slab form): per tile (relation, first grouped position, pairs,
// tiles of the relation), and the four entity ids of every pair in grouped order
struct PairGather {
    const int64_t* ph = nullptr; const int64_t* pt = nullptr; const int64_t* nh = nullptr; const int64_t* nt = nullptr;
    int4* tdesc = nullptr;     // [n / tile + R + 1]
    int4* gids = nullptr;      // [n]  (ph, pt, nh, nt) of grouped position g
    // staged entity gradients (kge_rescal_stage, include/kge_hip.h): with `sorted` set the grouping orders the pairs of every relation
    // of at most 64 pairs by pair index -- a deterministic grouped order (the scatter's own order is arrival order), which fixes the slot
    // numbers 4 g + j of the staged gradient rows and the summation order of the relation-matrix gradient
    int sorted = 0;
};
bool group_small_ok(int64_t n, int64_t R);   // the grouping runs as one launch (the only form that fills a PairGather)

int group_by_relation(const int64_t* r, int64_t n, int64_t R, const GroupWs& g, hipStream_t s);  // kge_dense.hip
int group_by_relation_split(IdSplit r, int64_t n, int64_t R, const GroupWs& g, hipStream_t s, float* zero_buf = nullptr,
                            int64_t zero_n = 0, int tile = TILE, const PairGather* pg = nullptr);   // n = total length; zero_buf: optional float buffer cleared on
                                                                    // the way; tile: items per tile (tile_rel then needs n / tile + R + 1)

// the slab form of the pairwise RESCAL step (kge_rescal_slab.hip); the grouping with kSlabChunk pairs per tile is enqueued by the caller
size_t rescal_slab_ws_bytes(int k, int64_t R, int64_t n);
void rescal_slab_gather(void* ws_slab, int k, int64_t R, int64_t n, PairGather* pg);   // where the grouping leaves tdesc / gids
int launch_rescal_slab_step(const kge_model_desc* m, int64_t n, const GroupWs& g, float margin, float* loss, unsigned* touched,
                            void* ws_slab, const kge_rescal_stage* stage, hipStream_t s);

// which (relation, tile-in-relation) is block `b`?  (tile_rel is written by the grouping's scatter pass)
__device__ __forceinline__ bool locate_tile(const int* __restrict__ tile_off, const int* __restrict__ tile_rel, int R, int b,
                                            int& rel, int& tile_in_rel) {
    if (b >= tile_off[R]) return false;
    rel = tile_rel[b];
    tile_in_rel = b - tile_off[rel];
    return true;
}

}  // namespace kge
