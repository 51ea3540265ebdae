// Shared helpers for the gfx950 kernels behind include/hscn.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/hscn.h"

#define HSCN_WAVE 64

#define HSCN_RETURN_IF_LAUNCH_FAILED()                 \
  do {                                                 \
    hipError_t e__ = hipGetLastError();                \
    if (e__ != hipSuccess) return (int)e__;            \
  } while (0)

static inline hipStream_t hscn_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline unsigned hscn_blocks(int64_t work, int per_block) {
  int64_t b = (work + per_block - 1) / per_block;
  return (unsigned)(b < 1 ? 1 : b);
}

// Separately rounded multiply / add: the CPU reference accumulates messages as
// round(w*x) followed by round(acc + .) (torch index_add_), never a fused fma.
__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case HSCN_ACT_RELU: return v > 0.f ? v : 0.f;
    case HSCN_ACT_ELU: return v > 0.f ? v : expm1f(v);
    case HSCN_ACT_TANH: return tanhf(v);
    default: return v;
  }
}

// derivative expressed through the forward OUTPUT y
__device__ __forceinline__ float act_grad_from_output(float y, int act) {
  switch (act) {
    case HSCN_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case HSCN_ACT_ELU: return y > 0.f ? 1.f : y + 1.f;
    case HSCN_ACT_TANH: return 1.f - y * y;
    default: return 1.f;
  }
}

// One element of the loss tail (reference graph_hscn/loss.py:6-19): kind 0 = BCE with logits
// max(x,0) - x*y + log1p(exp(-|x|)), kind 1 = L1; l = loss term, sg = sigmoid(x) (the score of both
// branches), g = d(mean loss)/dx with inv = 1/count.  Shared by k_criterion and the graph-resident
// backward, which must agree bit for bit.
__device__ __forceinline__ void criterion_elem(int kind, float x, float y, float inv, float& l, float& sg, float& g) {
  sg = 1.0f / (1.0f + expf(-x));
  if (kind == 0) {
    l = fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
    g = (sg - y) * inv;
  } else {
    const float d = x - y;
    l = fabsf(d);
    g = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * inv;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
