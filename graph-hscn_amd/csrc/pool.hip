// global_mean_pool (reference model/hscn.py:111): sorted-segment
// mean, one wavefront per graph, slot partials folded with __shfl_xor.
#include "hscn_common.h"

namespace {

constexpr int PL_THREADS = 256;
constexpr int PL_WAVES = PL_THREADS / 64;

template <int VEC>
__global__ void __launch_bounds__(PL_THREADS)
k_segment_mean(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ node,
               const float* __restrict__ x, float* __restrict__ out, int64_t num_seg, int width, int LPRp) {
  const int lane = threadIdx.x & 63;
  const int S = 64 / LPRp;
  const int slot = lane / LPRp;
  const int f = (lane - slot * LPRp) * VEC;
  const bool flive = f < width;
  for (int64_t g = (int64_t)blockIdx.x * PL_WAVES + (threadIdx.x >> 6); g < num_seg;
       g += (int64_t)gridDim.x * PL_WAVES) {
    const int s = rowptr[g], t = rowptr[g + 1];
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    if (flive) {
      for (int p = s + slot; p < t; p += S) {
        const int64_t i = node ? node[p] : p;
        if (VEC == 4) {
          float4 v = *reinterpret_cast<const float4*>(x + i * width + f);
          acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
        } else {
          acc[0] += x[i * width + f];
        }
      }
    }
    for (int off = 32; off >= LPRp; off >>= 1) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[k] += __shfl_xor(acc[k], off, 64);
    }
    if (slot == 0 && flive) {
      const float cnt = (float)((t - s) > 0 ? (t - s) : 1);
#pragma unroll
      for (int k = 0; k < VEC; ++k) out[g * width + f + k] = acc[k] / cnt;
    }
  }
}

__global__ void k_segment_mean_bwd(const int32_t* __restrict__ rowptr, const int64_t* __restrict__ batch,
                                   const float* __restrict__ g_out, float* __restrict__ g_x, int64_t total,
                                   int width) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; idx < total; idx += stride) {
    const int64_t i = idx / width;
    const int f = (int)(idx - i * width);
    const int64_t b = batch[i];
    const int c = rowptr[b + 1] - rowptr[b];
    g_x[idx] = g_out[b * width + f] / (float)(c > 0 ? c : 1);
  }
}

inline int pow2ceil(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

}  // namespace

extern "C" {

int hscn_segment_mean_fwd(const int32_t* rowptr, const int32_t* node, const float* x, float* out,
                          int64_t num_segments, int width, void* stream_) {
  if (num_segments < 0 || width < 1) return HSCN_E_BADARG;
  if (num_segments == 0) return 0;
  if (!rowptr || !x || !out) return HSCN_E_BADARG;
  const int VEC = (width % 4 == 0) ? 4 : 1;
  const int LPRp = pow2ceil((width + VEC - 1) / VEC);
  if (LPRp > 64) return HSCN_E_UNSUPPORTED;
  int64_t nb = (num_segments + PL_WAVES - 1) / PL_WAVES;
  if (nb > 8192) nb = 8192;
  hipStream_t st = hscn_stream(stream_);
  if (VEC == 4)
    k_segment_mean<4><<<(unsigned)nb, PL_THREADS, 0, st>>>(rowptr, node, x, out, num_segments, width, LPRp);
  else
    k_segment_mean<1><<<(unsigned)nb, PL_THREADS, 0, st>>>(rowptr, node, x, out, num_segments, width, LPRp);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

int hscn_segment_mean_bwd(const int32_t* rowptr, const int64_t* batch, const float* g_out, float* g_x,
                          int64_t num_nodes, int width, void* stream_) {
  if (num_nodes < 0 || width < 1) return HSCN_E_BADARG;
  if (num_nodes == 0) return 0;
  if (!rowptr || !batch || !g_out || !g_x) return HSCN_E_BADARG;
  const int64_t total = num_nodes * width;
  unsigned nb = hscn_blocks(total, 256);
  if (nb > 4096) nb = 4096;
  k_segment_mean_bwd<<<nb, 256, 0, hscn_stream(stream_)>>>(rowptr, batch, g_out, g_x, total, width);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

}  // extern "C"
