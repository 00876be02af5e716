// kge_dense.hip -- RESCAL / NTN (dense relation-matrix contraction) -- placeholder until the MFMA path lands.
#include "kge_internal.h"
namespace kge {
int launch_rescal_normalize(float*, int64_t, float*, int64_t, int, hipStream_t) { set_error("RESCAL path not built yet"); return -3; }
int launch_rescal_forward(const kge_model_desc*, const int64_t*, const int64_t*, const int64_t*, int64_t, float*, hipStream_t) { set_error("RESCAL path not built yet"); return -3; }
int launch_rescal_backward(const kge_model_desc*, const int64_t*, const int64_t*, const int64_t*, int64_t, const float*, hipStream_t) { set_error("RESCAL path not built yet"); return -3; }
int launch_ntn_forward(const kge_model_desc*, const int64_t*, const int64_t*, const int64_t*, int64_t, float*, hipStream_t) { set_error("NTN path not built yet"); return -3; }
int launch_ntn_backward(const kge_model_desc*, const int64_t*, const int64_t*, const int64_t*, int64_t, const float*, hipStream_t) { set_error("NTN path not built yet"); return -3; }
}
