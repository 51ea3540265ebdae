// The optimizer step behind a resident training step (reference: torch.optim.Adam / AdamW built by
// train/train.py:82 and train/train_clustering.py:30-33 from config.py's OPTIM_DICT) as ONE launch.
//
// The resident steps leave every parameter gradient in one flat buffer (graph_hscn/step.py); the parameters
// themselves stay the module's own tensors.  torch's capturable fused Adam is two launches (step counter, update):
// 8.6 us of kernel time behind a 21 us stage-A step.  This kernel walks the flat buffer once, finds each element's
// parameter tensor through a small offset table and applies torch's single-tensor update, operation for operation
// (torch/optim/adam.py::_single_tensor_adam, adamw.py):
//   [Adam, weight_decay]   g   = g + wd * p
//   [AdamW]                p   = p * (1 - lr * wd)
//   m   = m + (1 - beta1) * (g - m)                      (Tensor.lerp_)
//   v   = v * beta2;  v = v + ((1 - beta2) * g) * g       (mul_, addcmul_)
//   p   = p + (-(lr / (1 - beta1^t))) * (m / (sqrt(v) / sqrt(1 - beta2^t) + eps))     (addcdiv_)
// with the step-dependent scalars formed in double as Python does and rounded to float once.  The model family has
// a few thousand parameters: one workgroup, so the step counter needs no second launch.
#include "hscn_common.h"

namespace {

constexpr int ADAM_MAXSEG = 32;   // the tables travel in the kernel arguments: no dependent load in front of the update

struct AdamArgs {
  float* params[ADAM_MAXSEG];   // the parameter tensors, in the order of the flat buffers
  int32_t off[ADAM_MAXSEG + 1]; // element offsets of the segments in the flat buffers
  const float* grads;     // [P]
  float *m, *v;           // [P] exp_avg, exp_avg_sq
  float* step;            // [1] float step counter (torch keeps a float tensor), incremented here
  double* pows;           // [2] beta1^t, beta2^t of the LAST step (1, 1 before the first): running products -- a
                          // double pow per step costs more than the whole update
  const double* lr;       // [1] learning rate on the device (a scheduler may rewrite it between launches)
  double beta1, beta2, eps, wd;   // (doubles: Python forms the step's scalars from them in double)
  int nseg, P, decoupled;
};

__global__ void __launch_bounds__(1024) k_adam_flat(const AdamArgs A) {
  // The segment tables go from the kernel arguments to LDS through ONE vector load per table entry (lane k reads
  // entry k of the argument block as plain memory), requested together with the thread's gradients and moments; a
  // thread then finds its element's tensor by a 5-step binary search in LDS.  (A select chain over the arguments
  // costs ~150 VALU operations per element -- 4 us on one CU; a scalar loop over them one dependent load per entry.)
  __shared__ int soff[ADAM_MAXSEG + 1];
  __shared__ float* sp[ADAM_MAXSEG];
  {
    const AdamArgs* kp = reinterpret_cast<const AdamArgs*>(
        (const char*)__builtin_amdgcn_kernarg_segment_ptr());   // (generic pointer: a vector load, per-lane index)
    if (threadIdx.x <= ADAM_MAXSEG) soff[threadIdx.x] = (int)threadIdx.x <= A.nseg ? kp->off[threadIdx.x] : 0x7fffffff;
    if (threadIdx.x < ADAM_MAXSEG) sp[threadIdx.x] = (int)threadIdx.x < A.nseg ? kp->params[threadIdx.x] : nullptr;
  }
  constexpr int EPT = 4;   // (a model of this family is a few thousand parameters: one batch of requests)
  float g0[EPT], m0[EPT], v0[EPT];
#pragma unroll
  for (int u = 0; u < EPT; ++u) {
    const int i = threadIdx.x + u * 1024;
    const bool has = i < A.P;
    g0[u] = A.grads[has ? i : 0]; m0[u] = A.m[has ? i : 0]; v0[u] = A.v[has ? i : 0];
  }
  __syncthreads();
  auto addr = [&](int i) -> float* {
    int lo = 0, hi = A.nseg - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (soff[mid] <= i) lo = mid; else hi = mid - 1;
    }
    return sp[lo] + (i - soff[lo]);
  };
  float* pp0[EPT];
  float p0[EPT];
#pragma unroll
  for (int u = 0; u < EPT; ++u) {
    const int i = threadIdx.x + u * 1024;
    pp0[u] = addr(i < A.P ? i : 0);
    p0[u] = *pp0[u];
  }
  const float t = A.step[0] + 1.0f;
  const double lr = A.lr[0];
  const double b1t = A.pows[0] * A.beta1, b2t = A.pows[1] * A.beta2;
  const double bc1 = 1.0 - b1t, bc2 = 1.0 - b2t;
  const float step_size = (float)(lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  const float w1 = (float)(1.0 - A.beta1), w2 = (float)(1.0 - A.beta2), b2f = (float)A.beta2;
  const float decay = (float)(1.0 - lr * A.wd), wdf = (float)A.wd, epsf = (float)A.eps;
  auto update = [&](float* pp, int i, float p, float g, float m, float v) {
    if (A.wd != 0.0) {
      if (A.decoupled) p = p * decay;
      else g = g + wdf * p;
    }
    m = m + w1 * (g - m);
    v = v * b2f;
    v = v + (w2 * g) * g;
    const float denom = sqrtf(v) / bc2_sqrt + epsf;
    p = p + (-step_size) * (m / denom);
    *pp = p;
    A.m[i] = m;
    A.v[i] = v;
  };
#pragma unroll
  for (int u = 0; u < EPT; ++u) {
    const int i = threadIdx.x + u * 1024;
    if (i < A.P) update(pp0[u], i, p0[u], g0[u], m0[u], v0[u]);
  }
  for (int i = threadIdx.x + EPT * 1024; i < A.P; i += 1024) {
    float* pp = addr(i);
    update(pp, i, *pp, A.grads[i], A.m[i], A.v[i]);
  }
  __syncthreads();   // every thread has read the old counter
  if (threadIdx.x == 0) { A.step[0] = t; A.pows[0] = b1t; A.pows[1] = b2t; }
}

}  // namespace

extern "C" int hscn_adam_step(float* const* params_host, const int32_t* seg_off_host, int nseg, const float* grads,
                              float* exp_avg, float* exp_avg_sq, int64_t P, float* step_dev, double* beta_pows_dev,
                              const double* lr_dev, double beta1, double beta2, double eps, double weight_decay, int decoupled,
                              void* stream) {
  if (nseg < 1 || nseg > ADAM_MAXSEG || P < 0 || P > (1 << 24)) return HSCN_E_UNSUPPORTED;
  if (!params_host || !seg_off_host || !grads || !exp_avg || !exp_avg_sq || !step_dev || !beta_pows_dev || !lr_dev)
    return HSCN_E_BADARG;
  if (!(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0) || !(weight_decay >= 0.0))
    return HSCN_E_BADARG;
  if (P == 0) return 0;
  AdamArgs A;
  for (int k = 0; k < ADAM_MAXSEG; ++k) { A.params[k] = k < nseg ? params_host[k] : nullptr; A.off[k] = k <= nseg ? seg_off_host[k] : 0; }
  A.off[ADAM_MAXSEG] = nseg == ADAM_MAXSEG ? seg_off_host[nseg] : 0;
  for (int k = 0; k < nseg; ++k)
    if (!params_host[k] || seg_off_host[k + 1] < seg_off_host[k]) return HSCN_E_BADARG;
  if (seg_off_host[0] != 0 || seg_off_host[nseg] != P) return HSCN_E_BADARG;
  A.grads = grads; A.m = exp_avg; A.v = exp_avg_sq; A.step = step_dev; A.pows = beta_pows_dev;
  A.lr = lr_dev; A.beta1 = beta1; A.beta2 = beta2; A.eps = eps; A.wd = weight_decay; A.nseg = nseg; A.P = (int)P;
  A.decoupled = decoupled;
  k_adam_flat<<<1, 1024, 0, hscn_stream(stream)>>>(A);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}
