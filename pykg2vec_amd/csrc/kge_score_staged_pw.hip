// kge_score_staged_pw.hip -- the sampler-fused pointwise-logistic bundle kernel (kge_row_kernels.h) with STAGED gradient
// output, instantiated for DistMult and ComplEx (ComplexN3 = ComplEx + N3 regulariser): the atomic-free form of
// kge_train_pointwise_logistic_sampled, consumed by kge_optimizer_step_staged (kge_staged.hip).
#include "kge_row_kernels.h"

namespace kge {

int launch_pointwise_logistic_sampled_staged(const kge_model_desc* m, const int64_t* triples, const int64_t* perm, int64_t start,
                                             int64_t n_pos, int neg_rate, const float* bern, const uint64_t* slots,
                                             int64_t n_slots, uint64_t seed, uint64_t offset, float lmbda, int reg_type,
                                             float* loss, const StageSink& sink, hipStream_t s) {
    Geometry geo;
    if (!pick_geometry(m->dim, &geo)) { set_error("hidden size %d exceeds the register-resident row kernels (max 2048)", m->dim); return -1; }
    if (neg_rate > geo.G) { set_error("fused pointwise sampler: neg_rate %d exceeds the lane group (%d)", neg_rate, geo.G); return -1; }
    FusedSampler fs;
    fs.triples = triples; fs.perm = perm; fs.start = start; fs.E = m->tot_entity; fs.bern = bern;
    fs.slots = (const unsigned long long*)slots; fs.mask = (unsigned long long)(slots ? n_slots - 1 : 0);
    fs.seed = seed; fs.offset = offset; fs.cursor = nullptr;
    const DeviceModel dm = to_device_model(m);
    const int bundle = 1 + neg_rate;
    const int64_t n = n_pos * bundle;
#define KGE_STAGED_BODY (k_pointwise_bundle<M, G, NCH, false, true><<<dim3(Launch<M, G, NCH>::grid(n_pos)), dim3(kBlock), 0, s>>>( \
        dm, nullptr, nullptr, nullptr, nullptr, n, bundle, 1, lmbda, reg_type, loss, m->tot_relation, fs, sink))
    switch (m->model) {
        KGE_FOR_MODEL(KGE_DISTMULT, KGE_STAGED_BODY)
        KGE_FOR_MODEL(KGE_COMPLEX, KGE_STAGED_BODY)
        default: break;
    }
#undef KGE_STAGED_BODY
    set_error("kge_train_pointwise_logistic_sampled_staged: DistMult / ComplEx only");
    return -1;
}

}  // namespace kge
