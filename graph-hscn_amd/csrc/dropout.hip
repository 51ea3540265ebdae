// Inverted dropout of the MPNN baseline (reference model/mpnn.py:58, F.dropout(x, p, training)):
// y = x * keep / (1 - p).  The keep decision of element i is a pure function of (seed, i) -- a
// Philox-4x32-10 block per four consecutive elements -- so the backward regenerates the mask from
// the same (seed, count) instead of storing it: one read + one write per element either way.
#include "hscn_common.h"

namespace {

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
  const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
  c[0] = hi1 ^ c[1] ^ k0;
  c[1] = lo1;
  c[2] = hi0 ^ c[3] ^ k1;
  c[3] = lo0;
}

__device__ __forceinline__ void philox4x32_10(uint64_t seed, uint64_t ctr, uint32_t (&c)[4]) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  c[0] = (uint32_t)ctr;
  c[1] = (uint32_t)(ctr >> 32);
  c[2] = 0u;
  c[3] = 0u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

// keep iff the 32-bit draw is >= p * 2^32 (threshold computed on the host in integers)
__global__ void __launch_bounds__(256)
k_dropout(const float* __restrict__ x, float* __restrict__ y, int64_t n, uint32_t threshold, float scale,
          uint64_t seed) {
  const int64_t quads = (n + 3) >> 2;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (int64_t)gridDim.x * blockDim.x) {
    uint32_t c[4];
    philox4x32_10(seed, (uint64_t)q, c);
    const int64_t i = q << 2;
    if (i + 3 < n && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0) {
      float4 v = *reinterpret_cast<const float4*>(x + i);
      v.x = c[0] >= threshold ? v.x * scale : 0.f;
      v.y = c[1] >= threshold ? v.y * scale : 0.f;
      v.z = c[2] >= threshold ? v.z * scale : 0.f;
      v.w = c[3] >= threshold ? v.w * scale : 0.f;
      *reinterpret_cast<float4*>(y + i) = v;
    } else {
      for (int k = 0; k < 4 && i + k < n; ++k) y[i + k] = c[k] >= threshold ? x[i + k] * scale : 0.f;
    }
  }
}

}  // namespace

extern "C" int hscn_dropout(const float* x, float* y, int64_t count, float p, uint64_t seed, void* stream_) {
  if (count < 0 || !(p >= 0.f && p < 1.f) || (count > 0 && (!x || !y))) return HSCN_E_BADARG;
  if (count == 0) return 0;
  const double t = (double)p * 4294967296.0;
  const uint32_t threshold = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
  const float scale = 1.0f / (1.0f - p);
  unsigned nb = hscn_blocks((count + 3) / 4, 256);
  if (nb > 4096) nb = 4096;
  k_dropout<<<nb, 256, 0, hscn_stream(stream_)>>>(x, y, count, threshold, scale, seed);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}
