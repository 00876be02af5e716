// kge_score_ext.hip -- the generic row kernels (kge_row_kernels.h) instantiated for the second group of
// gather-type models: TransM (pairwise.py:281-365), CP (pointwise.py:320-387), SimplE / SimplE_ignr
// (pointwise.py:461-592) and QuatE (pointwise.py:595-768).  Pairwise kernels for the pairwise model, pointwise kernels
// for the pointwise models (the reference trains each family with its own loss only, utils/trainer.py:147-180).
// Same roofline as kge_score.hip: row gather + atomic scatter, HBM/L2-bound.
#include "kge_row_kernels.h"

namespace kge {

#define KGE_DISPATCH_PAIRWISE(model_id, BODY)                   \
    switch (model_id) {                                         \
        KGE_FOR_MODEL(KGE_TRANSM, BODY)                         \
        default: break;                                         \
    }
#define KGE_DISPATCH_POINTWISE(model_id, BODY)                  \
    switch (model_id) {                                         \
        KGE_FOR_MODEL(KGE_CP, BODY)                             \
        KGE_FOR_MODEL(KGE_SIMPLE, BODY)                         \
        KGE_FOR_MODEL(KGE_SIMPLE_IGNR, BODY)                    \
        KGE_FOR_MODEL(KGE_QUATE, BODY)                          \
        default: break;                                         \
    }
#define KGE_DISPATCH_ALL(model_id, BODY)                        \
    switch (model_id) {                                         \
        KGE_FOR_MODEL(KGE_TRANSM, BODY)                         \
        KGE_FOR_MODEL(KGE_CP, BODY)                             \
        KGE_FOR_MODEL(KGE_SIMPLE, BODY)                         \
        KGE_FOR_MODEL(KGE_SIMPLE_IGNR, BODY)                    \
        KGE_FOR_MODEL(KGE_QUATE, BODY)                          \
        default: break;                                         \
    }

int launch_score_forward_ext(const kge_model_desc* m, Geometry geo, const int64_t* h, const int64_t* r, const int64_t* t,
                             int64_t n, float* scores, hipStream_t s) {
    const DeviceModel dm = to_device_model(m);
    KGE_DISPATCH_ALL(m->model, (k_score_fwd<M, G, NCH><<<dim3(Launch<M, G, NCH>::grid(n)), dim3(kBlock), 0, s>>>(dm, h, r, t, n, scores)))
    set_error("kge_score_forward: unsupported model %d", m->model);
    return -1;
}

int launch_score_backward_ext(const kge_model_desc* m, Geometry geo, const int64_t* h, const int64_t* r, const int64_t* t,
                              int64_t n, const float* dscore, hipStream_t s) {
    const DeviceModel dm = to_device_model(m);
    KGE_DISPATCH_ALL(m->model, (k_score_bwd<M, G, NCH><<<dim3(Launch<M, G, NCH>::grid(n)), dim3(kBlock), 0, s>>>(dm, h, r, t, n, dscore)))
    set_error("kge_score_backward: unsupported model %d", m->model);
    return -1;
}

int launch_pairwise_hinge_ext(const kge_model_desc* m, Geometry geo, const int64_t* ph, const int64_t* pr, const int64_t* pt,
                              const int64_t* nh, const int64_t* nr, const int64_t* nt, int64_t n, float margin,
                              float* loss, const FusedSampler* fsp, bool sampled, hipStream_t s) {
    const DeviceModel dm = to_device_model(m);
    const FusedSampler fs = *fsp;
    if (sampled) {
        KGE_DISPATCH_PAIRWISE(m->model, (k_pairwise_hinge<M, G, NCH, true><<<dim3(Launch<M, G, NCH>::grid(n)), dim3(kBlock), 0, s>>>(dm, ph, pr, pt, nh, nr, nt, n, margin, loss, fs)))
    } else {
        KGE_DISPATCH_PAIRWISE(m->model, (k_pairwise_hinge<M, G, NCH, false><<<dim3(Launch<M, G, NCH>::grid(n)), dim3(kBlock), 0, s>>>(dm, ph, pr, pt, nh, nr, nt, n, margin, loss, fs)))
    }
    set_error("kge_train_pairwise_hinge: model %d is not trained with the pairwise hinge", m->model);
    return -1;
}

int launch_pointwise_logistic_ext(const kge_model_desc* m, Geometry geo, const int64_t* h, const int64_t* r, const int64_t* t,
                                  const int64_t* y, int64_t n, int bundle, float lmbda, int reg_type, float* loss,
                                  const FusedSampler* fsp, hipStream_t s) {
    const DeviceModel dm = to_device_model(m);
    const FusedSampler fs = fsp ? *fsp : FusedSampler{};
    if (bundle > 1) {
        const int chb = chunk_bundles((n + bundle - 1) / bundle);
        const int64_t nb = ((n + bundle - 1) / bundle + chb - 1) / chb;
        // few relations: relation-row gradients accumulate in LDS (one flush per workgroup), every bundle its own group
        const size_t rel_lds = (size_t)m->tot_relation * (size_t)rel_span_host(m->model, m->dim) * sizeof(float);
        if (rel_lds <= 32 * 1024) {  // larger tables cost more in LDS atomics and occupancy than they save
            const int64_t nbl = (n + bundle - 1) / bundle;
            KGE_DISPATCH_POINTWISE(m->model, (k_pointwise_bundle<M, G, NCH, true><<<dim3(Launch<M, G, NCH>::grid(nbl)), dim3(kBlock), rel_lds, s>>>(dm, h, r, t, y, n, bundle, 1, lmbda, reg_type, loss, m->tot_relation, fs)))
        }
        KGE_DISPATCH_POINTWISE(m->model, (k_pointwise_bundle<M, G, NCH, false><<<dim3(Launch<M, G, NCH>::grid(nb)), dim3(kBlock), 0, s>>>(dm, h, r, t, y, n, bundle, chb, lmbda, reg_type, loss, m->tot_relation, fs)))
    } else {
        KGE_DISPATCH_POINTWISE(m->model, (k_pointwise_logistic<M, G, NCH><<<dim3(Launch<M, G, NCH>::grid(n)), dim3(kBlock), 0, s>>>(dm, h, r, t, y, n, lmbda, reg_type, loss)))
    }
    set_error("kge_train_pointwise_logistic: model %d is not trained with the pointwise loss", m->model);
    return -1;
}

}  // namespace kge
