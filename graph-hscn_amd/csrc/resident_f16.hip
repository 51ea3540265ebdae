// Graph-resident HSCN engine with IEEE-half storage of node features and inter-layer activations
// (BASELINE.json configs[4]: "fp16 feat + bf16 accum" on PCQM-Contact): the kernels of resident_kernels.h
// instantiated with TS = half_t.  What is half: x_local, x_virtual, acts, xv_out and the job's x_virtual / xv_out /
// st_xv (every array that holds a node feature or an activation in HBM).  What stays float: parameters, pooled, z,
// pred, score, degree norms, gradient partials and gradients -- all sums accumulate in float registers (a superset
// of bf16 accumulation: same exponent range, 16 more mantissa bits).  An activation is rounded to half once, where
// it is produced, in LDS and in HBM alike.  H in {16, 32}.
#include "resident_kernels.h"

extern "C" {

int hscn_resident_fwd_f16(const hscn_half* x_local, const hscn_half* x_virtual, const int64_t* ei_ll, int64_t E_ll,
                          const int64_t* ei_vv, int64_t E_vv, const int64_t* ei_lv, int64_t E_lv,
                          const int32_t* lptr, const int32_t* vptr, const int32_t* eptr_ll, const int32_t* eptr_vv,
                          const int32_t* eptr_lv, int64_t N, int64_t V, int64_t B, int F, int H, int L, int C,
                          int head_act, float slope, const void* const* layer_params_host /* L x 9 */,
                          const float* W1, const float* b1, const float* W2, const float* b2, int max_n, int max_v,
                          int max_ell, int max_evv, int compute_virtual, hscn_half* acts, float* pooled, float* z,
                          float* pred, float* score, hscn_half* xv_out, int32_t* csr_rowptr_t, int32_t* csr_col_t,
                          float* dinv_out, int32_t* flag, void* stream_) {
  if (H != 16 && H != 32) return HSCN_E_UNSUPPORTED;
  return impl_resident_fwd<half_t>((const float*)x_local, (const float*)x_virtual, ei_ll, E_ll, ei_vv, E_vv, ei_lv,
                                   E_lv, lptr, vptr, eptr_ll, eptr_vv, eptr_lv, N, V, B, F, H, L, C, head_act, slope,
                                   layer_params_host, W1, b1, W2, b2, max_n, max_v, max_ell, max_evv, compute_virtual,
                                   (float*)acts, pooled, z, pred, score, (float*)xv_out, csr_rowptr_t, csr_col_t,
                                   dinv_out, flag, stream_);
}

int hscn_resident_bwd_f16(const hscn_half* x_local, const int64_t* ei_ll, int64_t E_ll, const int32_t* lptr,
                          const int32_t* eptr_ll, int64_t N, int64_t B, int F, int H, int L, int C, int head_act,
                          const void* const* W_ll_host /* L */, const float* W1, const float* W2,
                          const hscn_half* acts, const float* pooled, const float* z, const float* g_pred,
                          const float* g_scale, const int32_t* csr_rowptr_t, const int32_t* csr_col_t,
                          const float* dinv, int max_n, int max_ell, float* partials /*[B][P]*/,
                          float* grads /*[P]*/, int32_t* flag, const hscn_loss_tail* tail, void* stream_) {
  if (H != 16 && H != 32) return HSCN_E_UNSUPPORTED;
  return impl_resident_bwd<half_t>((const float*)x_local, ei_ll, E_ll, lptr, eptr_ll, N, B, F, H, L, C, head_act,
                                   W_ll_host, W1, W2, (const float*)acts, pooled, z, g_pred, g_scale, csr_rowptr_t,
                                   csr_col_t, dinv, max_n, max_ell, partials, grads, flag, tail, stream_);
}

int hscn_resident_bwd_with_virtual_f16(const hscn_half* x_local, const int64_t* ei_ll, int64_t E_ll,
                                       const int32_t* lptr, const int32_t* eptr_ll, int64_t N, int64_t B, int F,
                                       int H, int L, int C, int head_act, const void* const* W_ll_host,
                                       const float* W1, const float* W2, const hscn_half* acts, const float* pooled,
                                       const float* z, const float* g_pred, const float* g_scale,
                                       const int32_t* csr_rowptr_t, const int32_t* csr_col_t, const float* dinv,
                                       int max_n, int max_ell, float* partials, float* grads, int32_t* flag,
                                       const hscn_loss_tail* tail, const hscn_virtual_job* job, void* stream_) {
  if (H != 16 && H != 32) return HSCN_E_UNSUPPORTED;
  return impl_resident_bwd_with_virtual<half_t>((const float*)x_local, ei_ll, E_ll, lptr, eptr_ll, N, B, F, H, L, C,
                                                head_act, W_ll_host, W1, W2, (const float*)acts, pooled, z, g_pred,
                                                g_scale, csr_rowptr_t, csr_col_t, dinv, max_n, max_ell, partials,
                                                grads, flag, tail, job, stream_);
}

int hscn_resident_fwd_with_virtual_f16(const hscn_half* x_local, const int64_t* ei_ll, int64_t E_ll,
                                       const int32_t* lptr, const int32_t* eptr_ll, int64_t N, int64_t B, int F,
                                       int H, int L, int C, int head_act, const void* const* layer_params_host,
                                       const float* W1, const float* b1, const float* W2, const float* b2, int max_n,
                                       int max_ell, hscn_half* acts, float* pooled, float* z, float* pred,
                                       float* score, int32_t* csr_rowptr_t, int32_t* csr_col_t, float* dinv_out,
                                       int32_t* flag, const hscn_virtual_job* job, void* stream_) {
  if (H != 16 && H != 32) return HSCN_E_UNSUPPORTED;
  return impl_resident_fwd_with_virtual<half_t>((const float*)x_local, ei_ll, E_ll, lptr, eptr_ll, N, B, F, H, L, C,
                                                head_act, layer_params_host, W1, b1, W2, b2, max_n, max_ell,
                                                (float*)acts, pooled, z, pred, score, csr_rowptr_t, csr_col_t,
                                                dinv_out, flag, job, stream_);
}

int hscn_resident_train_step_f16(const hscn_half* x_local, const int64_t* ei_ll, int64_t E_ll, const int32_t* lptr,
                                 const int32_t* eptr_ll, int64_t N, int64_t B, int F, int H, int L, int C,
                                 int head_act, const void* const* layer_params_host, const float* W1, const float* b1,
                                 const float* W2, const float* b2, int max_n, int max_ell, const float* target,
                                 int loss_kind, float* pred, float* score, float* partials, float* grads,
                                 hscn_half* acts, uint32_t* sync, int32_t* flag, const hscn_virtual_job* job,
                                 const hscn_structure* structure, void* stream_) {
  return impl_resident_train_step<half_t>((const float*)x_local, ei_ll, E_ll, lptr, eptr_ll, N, B, F, H, L, C,
                                          head_act, layer_params_host, W1, b1, W2, b2, max_n, max_ell, target,
                                          loss_kind, pred, score, partials, grads, (float*)acts, sync, flag, job,
                                          structure, stream_);
}

}  // extern "C"
