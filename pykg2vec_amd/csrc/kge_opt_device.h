// kge_opt_device.h -- per-element dense optimiser updates with torch.optim default semantics (utils/trainer.py:112-131),
// shared by the flat-buffer sweep (kge_opt.hip) and the owner-computes training step (kge_pull.hip).
#pragma once
#include "kge_device.h"

namespace kge {

struct OptArgs {
    float lr;
    float step_size;  // Adam: lr / (1 - beta1^t)
    float bc2_sqrt;   // Adam: sqrt(1 - beta2^t)
};

template <int KIND>
__device__ __forceinline__ void opt_update(float& p, float g, float& s1, float& s2, const OptArgs& a) {
    if constexpr (KIND == KGE_OPT_SGD) {
        p = p - a.lr * g;
    } else if constexpr (KIND == KGE_OPT_ADAM) {  // torch/optim/adam.py _single_tensor_adam, defaults
        s1 = s1 + (1.0f - 0.9f) * (g - s1);          // exp_avg.lerp_(grad, 1 - beta1)
        s2 = s2 * 0.999f + (1.0f - 0.999f) * g * g;  // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
        // (the sweeps are memory-bound: a build with the 1-ulp hardware sqrt / rcp instead of these IEEE sequences streams no faster,
        //  profiles/r04_experiments.md section 2)
        const float denom = sqrtf(s2) / a.bc2_sqrt + 1e-8f;
        p = p + (-a.step_size) * s1 / denom;         // param.addcdiv_(exp_avg, denom, value=-step_size)
    } else if constexpr (KIND == KGE_OPT_ADAGRAD) {  // lr_decay 0, eps 1e-10
        s1 = s1 + g * g;
        p = p - a.lr * g / (sqrtf(s1) + 1e-10f);
    } else {  // RMSprop alpha 0.99 eps 1e-8, momentum 0, not centered
        s1 = s1 * 0.99f + (1.0f - 0.99f) * g * g;
        p = p - a.lr * g / (sqrtf(s1) + 1e-8f);
    }
}

// Streams that are read once and written once per step and do not fit any cache (optimiser state of tables beyond the
// 256 MB Infinity Cache): non-temporal accesses.  Measured on the RotatE FB15k-237 d=1000 step (117 MB tables, Adam):
// 264 -> 234 us per step (profiles/r02_experiments.md).
typedef float nt_f32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 stream_load(const float4* ptr) {
    if constexpr (NT) {
        const nt_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f32x4*>(ptr));
        return make_float4(v.x, v.y, v.z, v.w);
    } else {
        return *ptr;
    }
}
template <bool NT>
__device__ __forceinline__ void stream_store(float4* ptr, const float4& v) {
    if constexpr (NT) {
        nt_f32x4 w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
        __builtin_nontemporal_store(w, reinterpret_cast<nt_f32x4*>(ptr));
    } else {
        *ptr = v;
    }
}

// torch computes Adam's bias-correction scalars in double on the host, then applies them to fp32 tensors
inline OptArgs make_opt_args(float lr, int64_t step) {
    OptArgs a;
    a.lr = lr;
    const double bc1 = 1.0 - pow(0.9, (double)step);
    const double bc2 = 1.0 - pow(0.999, (double)step);
    a.step_size = (float)((double)lr / bc1);
    a.bc2_sqrt = (float)sqrt(bc2);
    return a;
}

}  // namespace kge
