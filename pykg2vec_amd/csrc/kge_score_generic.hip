// kge_score_generic.hip -- the model-generic row kernels (kge_row_kernels.h) instantiated for the first group of
// gather-type models (TransE / TransH / TransD, RotatE, DistMult / ComplEx / ANALOGY): forward, backward, fused
// pairwise hinge (explicit ids and sampler-fused), fused pointwise logistic, self-adversarial bundle.  The shared-row
// specialisations that carry the headline numbers live in kge_score.hip; this translation unit exists so that the two
// compile in parallel.
#include "kge_row_kernels.h"

namespace kge {

#define KGE_DISPATCH(model_id, BODY)                            \
    switch (model_id) {                                         \
        KGE_FOR_MODEL(KGE_TRANSE, BODY)                         \
        KGE_FOR_MODEL(KGE_TRANSH, BODY)                         \
        KGE_FOR_MODEL(KGE_TRANSD, BODY)                         \
        KGE_FOR_MODEL(KGE_ROTATE, BODY)                         \
        KGE_FOR_MODEL(KGE_DISTMULT, BODY)                       \
        KGE_FOR_MODEL(KGE_COMPLEX, BODY)                        \
        KGE_FOR_MODEL(KGE_ANALOGY, BODY)                        \
        default: break;                                         \
    }

// the reference trains each family with its own loss only (utils/trainer.py:147-180): hinge / self-adversarial kernels are
// instantiated for the pairwise models, the pointwise kernels (kge_score_generic_pw.hip) for the pointwise ones
#define KGE_DISPATCH_PAIRWISE(model_id, BODY)                   \
    switch (model_id) {                                         \
        KGE_FOR_MODEL(KGE_TRANSE, BODY)                         \
        KGE_FOR_MODEL(KGE_TRANSH, BODY)                         \
        KGE_FOR_MODEL(KGE_TRANSD, BODY)                         \
        KGE_FOR_MODEL(KGE_ROTATE, BODY)                         \
        default: break;                                         \
    }

static bool geometry_for(const kge_model_desc* m, Geometry* geo) {
    if (!pick_geometry(m->dim, geo)) {
        set_error("hidden size %d exceeds the register-resident row kernels (max 2048)", m->dim);
        return false;
    }
    return true;
}

int launch_score_forward(const kge_model_desc* m, const int64_t* h, const int64_t* r, const int64_t* t,
                         int64_t n, float* scores, hipStream_t s) {
    Geometry geo;
    if (!geometry_for(m, &geo)) return -1;
    const DeviceModel dm = to_device_model(m);
    KGE_DISPATCH(m->model, (k_score_fwd<M, G, NCH><<<dim3(Launch<M, G, NCH>::grid(n)), dim3(kBlock), 0, s>>>(dm, h, r, t, n, scores)))
    return launch_score_forward_ext(m, geo, h, r, t, n, scores, s);
}

int launch_score_backward(const kge_model_desc* m, const int64_t* h, const int64_t* r, const int64_t* t,
                          int64_t n, const float* dscore, hipStream_t s) {
    Geometry geo;
    if (!geometry_for(m, &geo)) return -1;
    const DeviceModel dm = to_device_model(m);
    KGE_DISPATCH(m->model, (k_score_bwd<M, G, NCH><<<dim3(Launch<M, G, NCH>::grid(n)), dim3(kBlock), 0, s>>>(dm, h, r, t, n, dscore)))
    return launch_score_backward_ext(m, geo, h, r, t, n, dscore, s);
}

int launch_pairwise_hinge(const kge_model_desc* m, const int64_t* ph, const int64_t* pr, const int64_t* pt,
                          const int64_t* nh, const int64_t* nr, const int64_t* nt, int64_t n, float margin,
                          float* loss, hipStream_t s) {
    Geometry geo;
    if (!geometry_for(m, &geo)) return -1;
    const DeviceModel dm = to_device_model(m);
    const FusedSampler fs{};
    KGE_DISPATCH_PAIRWISE(m->model, (k_pairwise_hinge<M, G, NCH, false><<<dim3(Launch<M, G, NCH>::grid(n)), dim3(kBlock), 0, s>>>(dm, ph, pr, pt, nh, nr, nt, n, margin, loss, fs)))
    return launch_pairwise_hinge_ext(m, geo, ph, pr, pt, nh, nr, nt, n, margin, loss, &fs, false, s);
}

int launch_pairwise_hinge_sampled_generic(const kge_model_desc* m, Geometry geo, const FusedSampler& fs, int64_t n,
                                          float margin, float* loss, hipStream_t s) {
    const DeviceModel dm = to_device_model(m);
    const int64_t* z = nullptr;
    KGE_DISPATCH_PAIRWISE(m->model, (k_pairwise_hinge<M, G, NCH, true><<<dim3(Launch<M, G, NCH>::grid(n)), dim3(kBlock), 0, s>>>(dm, z, z, z, z, z, z, n, margin, loss, fs)))
    return launch_pairwise_hinge_ext(m, geo, z, z, z, z, z, z, n, margin, loss, &fs, true, s);
}

int launch_selfadv_bundle(const kge_model_desc* m, const int64_t* ph, const int64_t* pr, const int64_t* pt,
                          const int64_t* nh, const int64_t* nr, const int64_t* nt, int64_t n_pos, int neg_rate,
                          float alpha, float* loss, hipStream_t s) {
    Geometry geo;
    if (!geometry_for(m, &geo)) return -1;
    if (neg_rate > geo.G) return 1;  // caller falls back to the three-launch path
    const DeviceModel dm = to_device_model(m);
    const int64_t n = n_pos;
    KGE_DISPATCH_PAIRWISE(m->model, (k_selfadv_bundle<M, G, NCH><<<dim3(Launch<M, G, NCH>::grid(n)), dim3(kBlock), 0, s>>>(dm, ph, pr, pt, nh, nr, nt, n_pos, neg_rate, alpha, loss)))
    set_error("kge_train_pairwise_selfadv: unsupported model %d", m->model);
    return -1;
}

}  // namespace kge
