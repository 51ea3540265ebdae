// Graph-resident HSCN engine, float32 storage: the C entry points of include/hscn.h over the kernels in
// resident_kernels.h (which holds the design notes).  resident_f16.hip instantiates the same kernels for IEEE-half
// feature / activation storage.
#include "resident_kernels.h"

extern "C" {


#ifdef HSCN_STAMPS
int hscn_diag_set_stamp_buffer(long long* buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &buf, sizeof(buf));
}
#endif

int hscn_resident_supported(int F, int H, int L, int C, int max_n, int max_v, int max_ell, int max_evv) {
  if (!(H == 16 || H == 32 || H == 64) || F < 1 || F > H || L < 1 || L > MAXL || C < 1 || C > 4096) return 0;
  if (max_n < 0 || max_v < 0 || max_ell < 0 || max_evv < 0) return 0;
  if (fwd_lds_bytes(H, C, max_n, max_v, max_ell, max_evv, 0, 0) > 160 * 1024) return 0;
  if (bwd_lds_bytes(H, C, max_n, max_ell, 1) > 160 * 1024) return 0;
  return 1;
}

int64_t hscn_resident_param_count(int F, int H, int L, int C) {
  int64_t P = 0;
  for (int l = 0; l < L; ++l) P += (int64_t)H * (l == 0 ? F : H) + H;
  return P + (int64_t)H * H + H + (int64_t)C * H + C;
}

int hscn_resident_fwd(const float* x_local, const float* x_virtual, const int64_t* ei_ll, int64_t E_ll,
                      const int64_t* ei_vv, int64_t E_vv, const int64_t* ei_lv, int64_t E_lv,
                      const int32_t* lptr, const int32_t* vptr, const int32_t* eptr_ll, const int32_t* eptr_vv,
                      const int32_t* eptr_lv, int64_t N, int64_t V, int64_t B, int F, int H, int L, int C,
                      int head_act, float slope, const void* const* layer_params_host /* L x 9 */,
                      const float* W1, const float* b1, const float* W2, const float* b2, int max_n, int max_v,
                      int max_ell, int max_evv, int compute_virtual, float* acts, float* pooled, float* z,
                      float* pred, float* score, float* xv_out, int32_t* csr_rowptr_t, int32_t* csr_col_t,
                      float* dinv_out, int32_t* flag, void* stream_) {
  return impl_resident_fwd<float>(x_local, x_virtual, ei_ll, E_ll, ei_vv, E_vv, ei_lv, E_lv, lptr, vptr, eptr_ll,
                                  eptr_vv, eptr_lv, N, V, B, F, H, L, C, head_act, slope, layer_params_host, W1, b1,
                                  W2, b2, max_n, max_v, max_ell, max_evv, compute_virtual, acts, pooled, z, pred,
                                  score, xv_out, csr_rowptr_t, csr_col_t, dinv_out, flag, stream_);
}

int hscn_resident_bwd(const float* x_local, const int64_t* ei_ll, int64_t E_ll, const int32_t* lptr,
                      const int32_t* eptr_ll, int64_t N, int64_t B, int F, int H, int L, int C, int head_act,
                      const void* const* W_ll_host /* L */, const float* W1, const float* W2, const float* acts,
                      const float* pooled, const float* z, const float* g_pred, const float* g_scale,
                      const int32_t* csr_rowptr_t, const int32_t* csr_col_t, const float* dinv, int max_n,
                      int max_ell, float* partials /*[B][P]*/, float* grads /*[P]*/, int32_t* flag,
                      const hscn_loss_tail* tail, void* stream_) {
  return impl_resident_bwd<float>(x_local, ei_ll, E_ll, lptr, eptr_ll, N, B, F, H, L, C, head_act, W_ll_host, W1, W2,
                                  acts, pooled, z, g_pred, g_scale, csr_rowptr_t, csr_col_t, dinv, max_n, max_ell,
                                  partials, grads, flag, tail, stream_);
}

int hscn_resident_bwd_with_virtual(const float* x_local, const int64_t* ei_ll, int64_t E_ll, const int32_t* lptr,
                                   const int32_t* eptr_ll, int64_t N, int64_t B, int F, int H, int L, int C,
                                   int head_act, const void* const* W_ll_host, const float* W1, const float* W2,
                                   const float* acts, const float* pooled, const float* z, const float* g_pred,
                                   const float* g_scale, const int32_t* csr_rowptr_t, const int32_t* csr_col_t,
                                   const float* dinv, int max_n, int max_ell, float* partials, float* grads,
                                   int32_t* flag, const hscn_loss_tail* tail, const hscn_virtual_job* job,
                                   void* stream_) {
  return impl_resident_bwd_with_virtual<float>(x_local, ei_ll, E_ll, lptr, eptr_ll, N, B, F, H, L, C, head_act,
                                               W_ll_host, W1, W2, acts, pooled, z, g_pred, g_scale, csr_rowptr_t,
                                               csr_col_t, dinv, max_n, max_ell, partials, grads, flag, tail, job,
                                               stream_);
}

int hscn_resident_fwd_with_virtual(const float* x_local, const int64_t* ei_ll, int64_t E_ll, const int32_t* lptr,
                                   const int32_t* eptr_ll, int64_t N, int64_t B, int F, int H, int L, int C,
                                   int head_act, const void* const* layer_params_host, const float* W1,
                                   const float* b1, const float* W2, const float* b2, int max_n, int max_ell,
                                   float* acts, float* pooled, float* z, float* pred, float* score,
                                   int32_t* csr_rowptr_t, int32_t* csr_col_t, float* dinv_out, int32_t* flag,
                                   const hscn_virtual_job* job, void* stream_) {
  return impl_resident_fwd_with_virtual<float>(x_local, ei_ll, E_ll, lptr, eptr_ll, N, B, F, H, L, C, head_act,
                                               layer_params_host, W1, b1, W2, b2, max_n, max_ell, acts, pooled, z,
                                               pred, score, csr_rowptr_t, csr_col_t, dinv_out, flag, job, stream_);
}

int hscn_resident_structure(const int64_t* ei_ll, int64_t E_ll, const int64_t* ei_vv, int64_t E_vv,
                            const int64_t* ei_lv, int64_t E_lv, const int32_t* lptr, const int32_t* vptr,
                            const int32_t* eptr_ll, const int32_t* eptr_vv, const int32_t* eptr_lv, int64_t B,
                            int max_n, int max_v, int max_ell, int max_evv, const hscn_structure* out,
                            int32_t* flag, void* stream_) {
  return impl_resident_structure(ei_ll, E_ll, ei_vv, E_vv, ei_lv, E_lv, lptr, vptr, eptr_ll, eptr_vv, eptr_lv, B, max_n,
                                 max_v, max_ell, max_evv, out, flag, stream_);
}

int hscn_resident_train_step_supported(int F, int H, int L, int C, int max_n, int max_ell, int max_v, int max_evv) {
  return step_supported(F, H, L, C, max_n, max_ell, max_v, max_evv);
}

int hscn_resident_train_step_wgs_per_cu(int F, int H, int L, int C, int max_n, int max_ell, int max_v, int max_evv) {
  if (!step_supported(F, H, L, C, max_n, max_ell, max_v, max_evv)) return 0;
  return step_wgs_per_cu(H, L, C, max_n, max_ell, max_v, max_evv);
}

int hscn_resident_train_step(const float* x_local, const int64_t* ei_ll, int64_t E_ll, const int32_t* lptr,
                             const int32_t* eptr_ll, int64_t N, int64_t B, int F, int H, int L, int C, int head_act,
                             const void* const* layer_params_host, const float* W1, const float* b1, const float* W2,
                             const float* b2, int max_n, int max_ell, const float* target, int loss_kind, float* pred,
                             float* score, float* partials, float* grads, float* acts, uint32_t* sync, int32_t* flag,
                             const hscn_virtual_job* job, const hscn_structure* structure, void* stream_) {
  return impl_resident_train_step<float>(x_local, ei_ll, E_ll, lptr, eptr_ll, N, B, F, H, L, C, head_act,
                                         layer_params_host, W1, b1, W2, b2, max_n, max_ell, target, loss_kind, pred,
                                         score, partials, grads, acts, sync, flag, job, structure, stream_);
}

}  // extern "C"
