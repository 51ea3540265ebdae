// GATConv attention part (heads=1, bipartite local -> virtual; reference
// model/hscn.py:85-87): per-target segment softmax of
// leaky_relu(a_src[j] + a_dst[v]) and the alpha-weighted segment sum.
//
// One wavefront per target row (= per cluster in HSCN): lanes stride the row's
// edges for the max / exp-sum passes (wave __shfl reductions), then split into
// 64/LPR edge slots x LPR feature lanes for the weighted sum, slot partials are
// folded with __shfl_xor.  HBM traffic: every gathered h_src row once, alpha
// written once for the backward.
#include "hscn_common.h"

namespace {

constexpr int GAT_THREADS = 256;
constexpr int GAT_WAVES = GAT_THREADS / 64;

__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : v * slope; }

template <int VEC>
__device__ __forceinline__ void ld(const float* p, float (&v)[VEC]) {
  if (VEC == 4) {
    float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
    v[0] = *p;
  }
}
template <int VEC>
__device__ __forceinline__ void st(float* p, const float (&v)[VEC]) {
  if (VEC == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  else *p = v[0];
}

// LPRp: power of two >= ceil(width/VEC), <= 64
template <int VEC>
__global__ void __launch_bounds__(GAT_THREADS)
k_gat_fwd(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ a_src,
          const float* __restrict__ a_dst, const float* __restrict__ h_src, const float* __restrict__ bias,
          float* __restrict__ alpha, float* __restrict__ out, int64_t num_dst, int width, float slope,
          int accumulate, int act, int LPRp) {
  const int lane = threadIdx.x & 63;
  const int S = 64 / LPRp;
  const int slot = lane / LPRp;
  const int f = (lane - slot * LPRp) * VEC;
  const bool flive = f < width;
  for (int64_t v = (int64_t)blockIdx.x * GAT_WAVES + (threadIdx.x >> 6); v < num_dst;
       v += (int64_t)gridDim.x * GAT_WAVES) {
    const int s = rowptr[v], t = rowptr[v + 1];
    const float ad = a_dst[v];
    float m = -INFINITY;
    for (int p = s + lane; p < t; p += 64) m = fmaxf(m, leaky(a_src[col[p]] + ad, slope));
    m = wave_max(m);
    float sum = 0.f;
    for (int p = s + lane; p < t; p += 64) sum += expf(leaky(a_src[col[p]] + ad, slope) - m);
    sum = wave_sum(sum);
    const float denom = sum + 1e-16f;
    for (int p = s + lane; p < t; p += 64) alpha[p] = expf(leaky(a_src[col[p]] + ad, slope) - m) / denom;
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    for (int p = s + slot; p < t; p += S) {
      const int j = col[p];
      const float a = expf(leaky(a_src[j] + ad, slope) - m) / denom;
      if (flive) {
        float hv[VEC];
        ld<VEC>(h_src + (size_t)j * width + f, hv);
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = fmaf(a, hv[k], acc[k]);
      }
    }
    for (int off = 32; off >= LPRp; off >>= 1) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[k] += __shfl_xor(acc[k], off, 64);
    }
    if (slot == 0 && flive) {
      float o[VEC];
      if (bias) {
        float b[VEC];
        ld<VEC>(bias + f, b);
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] += b[k];
      }
      if (accumulate) {
        float pv[VEC];
        ld<VEC>(out + (size_t)v * width + f, pv);
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] += pv[k];
      }
#pragma unroll
      for (int k = 0; k < VEC; ++k) o[k] = apply_act(acc[k], act);
      st<VEC>(out + (size_t)v * width + f, o);
    }
  }
}

template <int VEC>
__global__ void __launch_bounds__(GAT_THREADS)
k_gat_bwd_dst(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
              const float* __restrict__ a_src, const float* __restrict__ a_dst,
              const float* __restrict__ h_src, const float* __restrict__ alpha, const float* __restrict__ g,
              float* __restrict__ g_pre, float* __restrict__ g_a_dst, int64_t num_dst, int width, float slope,
              int LPRp) {
  const int lane = threadIdx.x & 63;
  const int S = 64 / LPRp;
  const int slot = lane / LPRp;
  const int fl = lane - slot * LPRp;
  const int f = fl * VEC;
  const bool flive = f < width;
  for (int64_t v = (int64_t)blockIdx.x * GAT_WAVES + (threadIdx.x >> 6); v < num_dst;
       v += (int64_t)gridDim.x * GAT_WAVES) {
    const int s = rowptr[v], t = rowptr[v + 1];
    const float ad = a_dst[v];
    float gv[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) gv[k] = 0.f;
    if (flive) ld<VEC>(g + (size_t)v * width + f, gv);
    float tsum = 0.f;
    // pass 1: dot_p = g[v,:] . h_src[col[p],:], parked in g_pre[p] by the slot leader
    for (int p0 = s; p0 < t; p0 += S) {
      const int p = p0 + slot;
      float d = 0.f;
      if (p < t && flive) {
        float hv[VEC];
        ld<VEC>(h_src + (size_t)col[p] * width + f, hv);
#pragma unroll
        for (int k = 0; k < VEC; ++k) d = fmaf(gv[k], hv[k], d);
      }
      for (int off = LPRp >> 1; off > 0; off >>= 1) d += __shfl_xor(d, off, 64);
      if (p < t && fl == 0) {
        g_pre[p] = d;
        tsum = fmaf(alpha[p], d, tsum);
      }
    }
    tsum = wave_sum(tsum);
    // pass 2 (same leader lanes re-read what they wrote)
    float gad = 0.f;
    for (int p0 = s; p0 < t; p0 += S) {
      const int p = p0 + slot;
      if (p < t && fl == 0) {
        const float gl = alpha[p] * (g_pre[p] - tsum);
        const float pre = a_src[col[p]] + ad;
        const float gp = pre > 0.f ? gl : gl * slope;
        g_pre[p] = gp;
        gad += gp;
      }
    }
    gad = wave_sum(gad);
    if (lane == 0) g_a_dst[v] = gad;
  }
}

template <int VEC>
__global__ void __launch_bounds__(GAT_THREADS)
k_gat_bwd_src(const int32_t* __restrict__ rowptr_t, const int32_t* __restrict__ col_t,
              const int32_t* __restrict__ pos_t, const float* __restrict__ alpha,
              const float* __restrict__ g_pre, const float* __restrict__ g, const float* __restrict__ att_src,
              float* __restrict__ g_a_src, float* __restrict__ g_h_src, int64_t num_src, int width, int LPR,
              int RPB) {
  const int rl = threadIdx.x / LPR;
  const int fl = threadIdx.x - rl * LPR;
  const int f = fl * VEC;
  if (rl >= RPB) return;
  for (int64_t j = (int64_t)blockIdx.x * RPB + rl; j < num_src; j += (int64_t)gridDim.x * RPB) {
    const int s = rowptr_t[j], t = rowptr_t[j + 1];
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    float gas = 0.f;
    for (int q = s; q < t; ++q) {
      const int p = pos_t[q];
      const int v = col_t[q];
      gas = add_rn(gas, g_pre[p]);
      const float a = alpha[p];
      float gv[VEC];
      ld<VEC>(g + (size_t)v * width + f, gv);
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[k] = add_rn(acc[k], mul_rn(a, gv[k]));
    }
    float at[VEC];
    ld<VEC>(att_src + f, at);
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = fmaf(gas, at[k], acc[k]);
    st<VEC>(g_h_src + (size_t)j * width + f, acc);
    if (fl == 0) g_a_src[j] = gas;
  }
}

inline int pow2ceil(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

}  // namespace

extern "C" {

int hscn_gat_segment_fwd(const int32_t* rowptr, const int32_t* col, const float* a_src, const float* a_dst,
                         const float* h_src, const float* bias, float* alpha, float* out, int64_t num_dst,
                         int width, float slope, int accumulate, int act, void* stream_) {
  if (num_dst < 0 || width < 1) return HSCN_E_BADARG;
  if (num_dst == 0) return 0;
  if (!rowptr || !col || !a_src || !a_dst || !h_src || !alpha || !out) return HSCN_E_BADARG;
  const int VEC = (width % 4 == 0) ? 4 : 1;
  const int LPRp = pow2ceil((width + VEC - 1) / VEC);
  if (LPRp > 64) return HSCN_E_UNSUPPORTED;
  int64_t nb = (num_dst + GAT_WAVES - 1) / GAT_WAVES;
  if (nb > 8192) nb = 8192;
  hipStream_t stt = hscn_stream(stream_);
  if (VEC == 4)
    k_gat_fwd<4><<<(unsigned)nb, GAT_THREADS, 0, stt>>>(rowptr, col, a_src, a_dst, h_src, bias, alpha, out,
                                                        num_dst, width, slope, accumulate, act, LPRp);
  else
    k_gat_fwd<1><<<(unsigned)nb, GAT_THREADS, 0, stt>>>(rowptr, col, a_src, a_dst, h_src, bias, alpha, out,
                                                        num_dst, width, slope, accumulate, act, LPRp);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

int hscn_gat_segment_bwd_dst(const int32_t* rowptr, const int32_t* col, const float* a_src, const float* a_dst,
                             const float* h_src, const float* alpha, const float* g, float* g_pre,
                             float* g_a_dst, int64_t num_dst, int width, float slope, void* stream_) {
  if (num_dst < 0 || width < 1) return HSCN_E_BADARG;
  if (num_dst == 0) return 0;
  if (!rowptr || !col || !a_src || !a_dst || !h_src || !alpha || !g || !g_pre || !g_a_dst) return HSCN_E_BADARG;
  const int VEC = (width % 4 == 0) ? 4 : 1;
  const int LPRp = pow2ceil((width + VEC - 1) / VEC);
  if (LPRp > 64) return HSCN_E_UNSUPPORTED;
  int64_t nb = (num_dst + GAT_WAVES - 1) / GAT_WAVES;
  if (nb > 8192) nb = 8192;
  hipStream_t stt = hscn_stream(stream_);
  if (VEC == 4)
    k_gat_bwd_dst<4><<<(unsigned)nb, GAT_THREADS, 0, stt>>>(rowptr, col, a_src, a_dst, h_src, alpha, g, g_pre,
                                                            g_a_dst, num_dst, width, slope, LPRp);
  else
    k_gat_bwd_dst<1><<<(unsigned)nb, GAT_THREADS, 0, stt>>>(rowptr, col, a_src, a_dst, h_src, alpha, g, g_pre,
                                                            g_a_dst, num_dst, width, slope, LPRp);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

int hscn_gat_segment_bwd_src(const int32_t* rowptr_t, const int32_t* col_t, const int32_t* pos_t,
                             const float* alpha, const float* g_pre, const float* g, const float* att_src,
                             float* g_a_src, float* g_h_src, int64_t num_src, int width, void* stream_) {
  if (num_src < 0 || width < 1) return HSCN_E_BADARG;
  if (num_src == 0) return 0;
  if (!rowptr_t || !col_t || !pos_t || !alpha || !g_pre || !g || !att_src || !g_a_src || !g_h_src)
    return HSCN_E_BADARG;
  const int VEC = (width % 4 == 0) ? 4 : 1;
  const int LPR = width / VEC;
  if (LPR > GAT_THREADS) return HSCN_E_UNSUPPORTED;
  const int RPB = GAT_THREADS / LPR;
  int64_t nb = (num_src + RPB - 1) / RPB;
  if (nb > 8192) nb = 8192;
  hipStream_t stt = hscn_stream(stream_);
  if (VEC == 4)
    k_gat_bwd_src<4><<<(unsigned)nb, GAT_THREADS, 0, stt>>>(rowptr_t, col_t, pos_t, alpha, g_pre, g, att_src,
                                                            g_a_src, g_h_src, num_src, width, LPR, RPB);
  else
    k_gat_bwd_src<1><<<(unsigned)nb, GAT_THREADS, 0, stt>>>(rowptr_t, col_t, pos_t, alpha, g_pre, g, att_src,
                                                            g_a_src, g_h_src, num_src, width, LPR, RPB);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

}  // extern "C"
