// Normalisation layers of the MPNN baseline (reference graph_hscn/model/mpnn.py:34-44,53-56: nn.BatchNorm1d /
// nn.LayerNorm over the hidden width after every hidden convolution).  Row-major [N, H] activations, H <= 1024.
//
//   LayerNorm : per row   y = (x - mean) * rstd * gamma + beta,  var biased, eps inside the root
//   BatchNorm1d (training): per column the same with batch statistics (biased variance for the normalisation,
//               unbiased for the running estimate, momentum m: running = (1 - m) running + m batch);
//               (eval): running statistics.
// HBM-bound streaming passes.  Column statistics and every parameter gradient (gamma, beta) are sums over N rows:
// per-chunk partials in a fixed layout + an ordered fold -- no float atomics, bitwise reproducible.  Statistics are
// two-pass (mean first, then sum (x - mean)^2), as torch computes them.
#include "hscn_common.h"

namespace {

constexpr int NR_THREADS = 256;
constexpr int NR_CHUNK = 256;   // rows per partial-sum chunk

// ---- LayerNorm forward: one wave per row (H <= 64: lanes stride the row; H > 64: loop) ----------------------------
__global__ void __launch_bounds__(NR_THREADS)
k_layer_norm_fwd(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                 float* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out, int64_t N, int H,
                 float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (NR_THREADS / 64) + (threadIdx.x >> 6);
  if (row >= N) return;
  const float* xr = x + row * H;
  float s = 0.f;
  for (int k = lane; k < H; k += 64) s += xr[k];
  const float mean = wave_sum(s) / (float)H;
  float v = 0.f;
  for (int k = lane; k < H; k += 64) { const float d = xr[k] - mean; v = fmaf(d, d, v); }
  const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)H + eps);
  for (int k = lane; k < H; k += 64) y[row * H + k] = (xr[k] - mean) * rstd * gamma[k] + beta[k];
  if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

// LayerNorm backward, input gradient: gx = rstd * (g - mean_k(g) - xhat * mean_k(g * xhat)),  g = gy * gamma
__global__ void __launch_bounds__(NR_THREADS)
k_layer_norm_bwd_x(const float* __restrict__ gy, const float* __restrict__ x, const float* __restrict__ gamma,
                   const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ gx, int64_t N,
                   int H) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (NR_THREADS / 64) + (threadIdx.x >> 6);
  if (row >= N) return;
  const float m = mean[row], r = rstd[row];
  float s1 = 0.f, s2 = 0.f;
  for (int k = lane; k < H; k += 64) {
    const float g = gy[row * H + k] * gamma[k], xh = (x[row * H + k] - m) * r;
    s1 += g;
    s2 = fmaf(g, xh, s2);
  }
  s1 = wave_sum(s1) / (float)H;
  s2 = wave_sum(s2) / (float)H;
  for (int k = lane; k < H; k += 64) {
    const float g = gy[row * H + k] * gamma[k], xh = (x[row * H + k] - m) * r;
    gx[row * H + k] = r * (g - s1 - xh * s2);
  }
}

// ---- column sums over rows in chunks: partial[chunk][which][k] ------------------------------------------------------
// MODE 0: {sum x}                         MODE 1: {sum (x - mean_k)^2}   (batch statistics, two passes)
// MODE 2: {sum gy, sum gy * xhat}  xhat = (x - mean_row) * rstd_row        (LayerNorm: g_beta, g_gamma)
// MODE 3: {sum gy, sum gy * xhat}  xhat = (x - mean_k) * rstd_k            (BatchNorm: g_beta, g_gamma)
template <int MODE>
__global__ void __launch_bounds__(NR_THREADS)
k_col_partials(const float* __restrict__ a, const float* __restrict__ x, const float* __restrict__ mean,
               const float* __restrict__ rstd, float* __restrict__ partial, int64_t N, int H) {
  constexpr int NQ = MODE >= 2 ? 2 : 1;
  const int64_t r0 = (int64_t)blockIdx.x * NR_CHUNK;
  const int64_t r1 = r0 + NR_CHUNK < N ? r0 + NR_CHUNK : N;
  for (int k = threadIdx.x; k < H; k += NR_THREADS) {
    float s0 = 0.f, s1 = 0.f;
    for (int64_t r = r0; r < r1; ++r) {
      if (MODE == 0) s0 += a[r * H + k];
      else if (MODE == 1) { const float d = a[r * H + k] - mean[k]; s0 = fmaf(d, d, s0); }
      else {
        const float g = a[r * H + k];
        const float xh = MODE == 2 ? (x[r * H + k] - mean[r]) * rstd[r] : (x[r * H + k] - mean[k]) * rstd[k];
        s0 += g;
        s1 = fmaf(g, xh, s1);
      }
    }
    partial[((size_t)blockIdx.x * NQ + 0) * H + k] = s0;
    if (NQ == 2) partial[((size_t)blockIdx.x * NQ + 1) * H + k] = s1;
  }
}

// out[q][k] = sum over chunks in chunk order (one thread per (q, k): G is a few hundred at most)
__global__ void k_col_fold(const float* __restrict__ partial, float* __restrict__ out0, float* __restrict__ out1,
                           int G, int NQ, int H, float scale0, float scale1) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= NQ * H) return;
  const int q = idx / H, k = idx - q * H;
  float s = 0.f;
  int g = 0;
  for (; g + 8 <= G; g += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = partial[((size_t)(g + u) * NQ + q) * H + k];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; g < G; ++g) s += partial[((size_t)g * NQ + q) * H + k];
  if (q == 0) out0[k] = s * scale0;
  else out1[k] = s * scale1;
}

// BatchNorm: rstd from the biased variance, running statistics (unbiased variance), optional
__global__ void k_bn_stats(const float* __restrict__ mean, float* __restrict__ var_to_rstd, float* __restrict__ running_mean,
                           float* __restrict__ running_var, int H, float eps, float momentum, float unbias) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= H) return;
  const float var = var_to_rstd[k];                 // biased batch variance
  if (running_mean) running_mean[k] = (1.f - momentum) * running_mean[k] + momentum * mean[k];
  if (running_var) running_var[k] = (1.f - momentum) * running_var[k] + momentum * (var * unbias);
  var_to_rstd[k] = 1.0f / sqrtf(var + eps);
}

// eval mode: what the backward needs of the running statistics
__global__ void k_bn_eval_stats(const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                float* __restrict__ save_mean, float* __restrict__ save_rstd, int H, float eps) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= H) return;
  save_mean[k] = running_mean[k];
  save_rstd[k] = 1.0f / sqrtf(running_var[k] + eps);
}

// y = (x - mean_k) * rstd_k * gamma_k + beta_k        (training: batch statistics; eval: rstd_k from running_var)
__global__ void __launch_bounds__(NR_THREADS)
k_bn_apply(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd_or_var,
           const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y, int64_t total, int H,
           float eps, int is_var) {
  for (int64_t i = (int64_t)blockIdx.x * NR_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * NR_THREADS) {
    const int k = (int)(i % H);
    const float r = is_var ? 1.0f / sqrtf(rstd_or_var[k] + eps) : rstd_or_var[k];
    y[i] = (x[i] - mean[k]) * r * gamma[k] + beta[k];
  }
}

// BatchNorm backward (training): gx = gamma * rstd * (gy - g_beta / N - xhat * g_gamma / N)
__global__ void __launch_bounds__(NR_THREADS)
k_bn_bwd_x(const float* __restrict__ gy, const float* __restrict__ x, const float* __restrict__ mean,
           const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ g_beta,
           const float* __restrict__ g_gamma, float* __restrict__ gx, int64_t total, int H, float inv_n, int training) {
  for (int64_t i = (int64_t)blockIdx.x * NR_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * NR_THREADS) {
    const int k = (int)(i % H);
    const float xh = (x[i] - mean[k]) * rstd[k];
    gx[i] = training ? gamma[k] * rstd[k] * (gy[i] - g_beta[k] * inv_n - xh * g_gamma[k] * inv_n)
                     : gamma[k] * rstd[k] * gy[i];
  }
}

inline int nr_chunks(int64_t N) { return (int)((N + NR_CHUNK - 1) / NR_CHUNK); }
inline unsigned nr_grid(int64_t total) {
  int64_t nb = (total + NR_THREADS - 1) / NR_THREADS;
  return (unsigned)(nb < 1 ? 1 : (nb > 8192 ? 8192 : nb));
}

}  // namespace

extern "C" {

size_t hscn_norm_workspace_bytes(int64_t N, int H) {
  if (N < 0 || H < 1) return 0;
  return (size_t)nr_chunks(N > 0 ? N : 1) * 2 * H * sizeof(float);
}

int hscn_layer_norm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                        int64_t N, int H, float eps, void* stream_) {
  if (N < 0 || H < 1 || H > 1024) return HSCN_E_BADARG;
  if (N == 0) return 0;
  if (!x || !gamma || !beta || !y || !mean || !rstd) return HSCN_E_BADARG;
  k_layer_norm_fwd<<<hscn_blocks(N, NR_THREADS / 64), NR_THREADS, 0, hscn_stream(stream_)>>>(x, gamma, beta, y, mean, rstd,
                                                                                        N, H, eps);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

int hscn_layer_norm_bwd(const float* gy, const float* x, const float* gamma, const float* mean, const float* rstd,
                        float* gx, float* g_gamma, float* g_beta, int64_t N, int H, void* workspace,
                        size_t workspace_bytes, void* stream_) {
  if (N < 0 || H < 1 || H > 1024) return HSCN_E_BADARG;
  if (!gy || !x || !gamma || !mean || !rstd || !gx || !g_gamma || !g_beta || !workspace) return HSCN_E_BADARG;
  if (workspace_bytes < hscn_norm_workspace_bytes(N, H)) return HSCN_E_WORKSPACE;
  hipStream_t st = hscn_stream(stream_);
  const int G = nr_chunks(N > 0 ? N : 1);
  if (N > 0)
    k_layer_norm_bwd_x<<<hscn_blocks(N, NR_THREADS / 64), NR_THREADS, 0, st>>>(gy, x, gamma, mean, rstd, gx, N, H);
  k_col_partials<2><<<G, NR_THREADS, 0, st>>>(gy, x, mean, rstd, (float*)workspace, N, H);
  k_col_fold<<<hscn_blocks(2 * H, 256), 256, 0, st>>>((const float*)workspace, g_beta, g_gamma, G, 2, H, 1.f, 1.f);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

int hscn_batch_norm_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                        float* y, float* save_mean, float* save_rstd, int64_t N, int H, float eps, float momentum,
                        int training, void* workspace, size_t workspace_bytes, void* stream_) {
  if (N < 0 || H < 1 || H > 1024) return HSCN_E_BADARG;
  if (!x || !gamma || !beta || !y || !save_mean || !save_rstd || !workspace) return HSCN_E_BADARG;
  if (workspace_bytes < hscn_norm_workspace_bytes(N, H)) return HSCN_E_WORKSPACE;
  hipStream_t st = hscn_stream(stream_);
  const int64_t total = N * H;
  if (!training) {
    if (!running_mean || !running_var) return HSCN_E_BADARG;
    k_bn_eval_stats<<<hscn_blocks(H, 256), 256, 0, st>>>(running_mean, running_var, save_mean, save_rstd, H, eps);
    if (total > 0) k_bn_apply<<<nr_grid(total), NR_THREADS, 0, st>>>(x, running_mean, running_var, gamma, beta, y, total, H, eps, 1);
    HSCN_RETURN_IF_LAUNCH_FAILED();
    return 0;
  }
  if (N < 1) return HSCN_E_BADARG;      // (torch raises for a batch of no rows in training mode as well)
  const int G = nr_chunks(N);
  float* ws = (float*)workspace;
  k_col_partials<0><<<G, NR_THREADS, 0, st>>>(x, nullptr, nullptr, nullptr, ws, N, H);
  k_col_fold<<<hscn_blocks(H, 256), 256, 0, st>>>(ws, save_mean, nullptr, G, 1, H, 1.0f / (float)N, 0.f);
  k_col_partials<1><<<G, NR_THREADS, 0, st>>>(x, nullptr, save_mean, nullptr, ws, N, H);
  k_col_fold<<<hscn_blocks(H, 256), 256, 0, st>>>(ws, save_rstd, nullptr, G, 1, H, 1.0f / (float)N, 0.f);   // biased variance
  k_bn_stats<<<hscn_blocks(H, 256), 256, 0, st>>>(save_mean, save_rstd, running_mean, running_var, H, eps, momentum,
                                                  N > 1 ? (float)N / (float)(N - 1) : 1.f);
  k_bn_apply<<<nr_grid(total), NR_THREADS, 0, st>>>(x, save_mean, save_rstd, gamma, beta, y, total, H, eps, 0);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

int hscn_batch_norm_bwd(const float* gy, const float* x, const float* gamma, const float* save_mean,
                        const float* save_rstd, float* gx, float* g_gamma, float* g_beta, int64_t N, int H,
                        int training, void* workspace, size_t workspace_bytes, void* stream_) {
  if (N < 1 || H < 1 || H > 1024) return HSCN_E_BADARG;
  if (!gy || !x || !gamma || !save_mean || !save_rstd || !gx || !g_gamma || !g_beta || !workspace) return HSCN_E_BADARG;
  if (workspace_bytes < hscn_norm_workspace_bytes(N, H)) return HSCN_E_WORKSPACE;
  hipStream_t st = hscn_stream(stream_);
  const int G = nr_chunks(N);
  k_col_partials<3><<<G, NR_THREADS, 0, st>>>(gy, x, save_mean, save_rstd, (float*)workspace, N, H);
  k_col_fold<<<hscn_blocks(2 * H, 256), 256, 0, st>>>((const float*)workspace, g_beta, g_gamma, G, 2, H, 1.f, 1.f);
  k_bn_bwd_x<<<nr_grid(N * H), NR_THREADS, 0, st>>>(gy, x, save_mean, save_rstd, gamma, g_beta, g_gamma, gx, N * H, H,
                                                    1.0f / (float)N, training);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

}  // extern "C"
