// kge_score_generic_pw.hip -- the fused pointwise-logistic kernels (kge_row_kernels.h) instantiated for DistMult, ComplEx
// and ANALOGY; a translation unit of its own so that it compiles in parallel with kge_score_generic.hip.
#include "kge_row_kernels.h"

namespace kge {

#define KGE_DISPATCH_POINTWISE_A(model_id, BODY)                \
    switch (model_id) {                                         \
        KGE_FOR_MODEL(KGE_DISTMULT, BODY)                       \
        KGE_FOR_MODEL(KGE_COMPLEX, BODY)                        \
        KGE_FOR_MODEL(KGE_ANALOGY, BODY)                        \
        default: break;                                         \
    }

static bool geometry_for(const kge_model_desc* m, Geometry* geo) {
    if (!pick_geometry(m->dim, geo)) {
        set_error("hidden size %d exceeds the register-resident row kernels (max 2048)", m->dim);
        return false;
    }
    return true;
}

int launch_pointwise_logistic(const kge_model_desc* m, const int64_t* h, const int64_t* r, const int64_t* t,
                              const int64_t* y, int64_t n, int bundle, float lmbda, int reg_type, float* loss,
                              const FusedSampler* fsp, hipStream_t s) {
    Geometry geo;
    if (!geometry_for(m, &geo)) return -1;
    const FusedSampler fs = fsp ? *fsp : FusedSampler{};
    if (fsp && bundle - 1 > geo.G) { set_error("fused pointwise sampler: neg_rate %d exceeds the lane group (%d)", bundle - 1, geo.G); return -1; }
    const DeviceModel dm = to_device_model(m);
    if (bundle > 1) {
        const int chb = chunk_bundles((n + bundle - 1) / bundle);
        const int64_t nb = ((n + bundle - 1) / bundle + chb - 1) / chb;
        // few relations: relation-row gradients accumulate in LDS (one flush per workgroup), every bundle its own group
        const size_t rel_lds = (size_t)m->tot_relation * (size_t)rel_span_host(m->model, m->dim) * sizeof(float);
        if (rel_lds <= 32 * 1024) {  // larger tables cost more in LDS atomics and occupancy than they save
            const int64_t nbl = (n + bundle - 1) / bundle;
            KGE_DISPATCH_POINTWISE_A(m->model, (k_pointwise_bundle<M, G, NCH, true><<<dim3(Launch<M, G, NCH>::grid(nbl)), dim3(kBlock), rel_lds, s>>>(dm, h, r, t, y, n, bundle, 1, lmbda, reg_type, loss, m->tot_relation, fs)))
        }
        KGE_DISPATCH_POINTWISE_A(m->model, (k_pointwise_bundle<M, G, NCH, false><<<dim3(Launch<M, G, NCH>::grid(nb)), dim3(kBlock), 0, s>>>(dm, h, r, t, y, n, bundle, chb, lmbda, reg_type, loss, m->tot_relation, fs)))
        return launch_pointwise_logistic_ext(m, geo, h, r, t, y, n, bundle, lmbda, reg_type, loss, fsp, s);
    }
    KGE_DISPATCH_POINTWISE_A(m->model, (k_pointwise_logistic<M, G, NCH><<<dim3(Launch<M, G, NCH>::grid(n)), dim3(kBlock), 0, s>>>(dm, h, r, t, y, n, lmbda, reg_type, loss)))
    return launch_pointwise_logistic_ext(m, geo, h, r, t, y, n, bundle, lmbda, reg_type, loss, fsp, s);
}


int launch_pointwise_logistic_sampled(const kge_model_desc* m, const int64_t* triples, const int64_t* perm, int64_t start,
                                      int64_t n_pos, int neg_rate, const float* bern, const uint64_t* slots, int64_t n_slots,
                                      uint64_t seed, uint64_t offset, const int64_t* cursor, float lmbda, int reg_type,
                                      float* loss, hipStream_t s) {
    if (m->tot_entity > (1 << 24)) { set_error("fused sampler: more than 2^24 entities not supported by the packed key"); return -1; }
    FusedSampler fs;
    fs.triples = triples; fs.perm = perm; fs.start = start; fs.E = m->tot_entity; fs.bern = bern;
    fs.slots = (const unsigned long long*)slots; fs.mask = (unsigned long long)(slots ? n_slots - 1 : 0);
    fs.seed = seed; fs.offset = offset; fs.cursor = cursor;
    return launch_pointwise_logistic(m, nullptr, nullptr, nullptr, nullptr, n_pos * (1 + (int64_t)neg_rate), 1 + neg_rate, lmbda,
                                     reg_type, loss, &fs, s);
}

}  // namespace kge
