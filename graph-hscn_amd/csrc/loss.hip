// Loss tail of the training step (reference graph_hscn/loss.py:6-19 called at
// train/train.py:82): binary-cross-entropy-with-logits / L1, mean reduction, and the
// sigmoid score, on the [B, C] prediction.  One launch produces the loss, the score and
// dL/dpred (so the backward is a single scale), replacing ~8 elementwise/reduce launches
// of a few microseconds each on a 1 280-element tensor.  Ordered block reduction: reproducible.
#include "hscn_common.h"

namespace {

// One workgroup of 16 waves; a thread requests its (up to four) elements of a 4096-element slab
// before it computes any: a [128, 10] prediction is one memory round trip, not five.
__global__ void __launch_bounds__(1024) k_criterion(const float* __restrict__ pred, const float* __restrict__ target,
                                                    int64_t count, int kind, float* __restrict__ loss,
                                                    float* __restrict__ score, float* __restrict__ grad) {
  __shared__ float red[16];
  const float inv = 1.0f / (float)count;
  float s = 0.f;
  for (int64_t base = 0; base < count; base += 4096) {
    float xs[4], ys[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = base + u * 1024 + threadIdx.x;
      xs[u] = pred[i < count ? i : 0];
      ys[u] = target[i < count ? i : 0];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = base + u * 1024 + threadIdx.x;
      if (i >= count) continue;
      float l, sg, g;
      criterion_elem(kind, xs[u], ys[u], inv, l, sg, g);
      s += l;
      if (score) score[i] = sg;
      grad[i] = g;
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w];
    loss[0] = t * inv;
  }
}

__global__ void k_scale(const float* __restrict__ g, const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  const float s = g[0];
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) y[i] = s * x[i];
}

}  // namespace

extern "C" {

int hscn_criterion_fwd(const float* pred, const float* target, int64_t count, int kind, float* loss, float* score,
                       float* grad, void* stream_) {
  if (count < 1 || !pred || !target || !loss || !grad || (kind != 0 && kind != 1)) return HSCN_E_BADARG;
  k_criterion<<<1, 1024, 0, hscn_stream(stream_)>>>(pred, target, count, kind, loss, score, grad);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

int hscn_scale(const float* g, const float* x, float* y, int64_t count, void* stream_) {
  if (count < 0 || (count > 0 && (!g || !x || !y))) return HSCN_E_BADARG;
  if (count == 0) return 0;
  unsigned nb = hscn_blocks(count, 256);
  if (nb > 1024) nb = 1024;
  k_scale<<<nb, 256, 0, hscn_stream(stream_)>>>(g, x, y, count);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

}  // extern "C"
