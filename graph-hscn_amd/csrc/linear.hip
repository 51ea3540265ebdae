// Dense feature transforms of the hot path (torch_geometric.nn.Linear inside
// GraphConv / GCNConv / GATConv and the HSCN head; reference model/hscn.py:51,54,
// 99-100,112-113).  Widths are 9..128, so W lives in LDS and rows stream from HBM
// once; weight gradients reduce in two ordered stages (no float atomics).
#include "hscn_common.h"
#include <cstdlib>

namespace {

constexpr int LIN_THREADS = 256;

// y[r, og*VEC .. +VEC) for lane (rl, og); LPR lanes per row.
template <int VEC>
__global__ void __launch_bounds__(LIN_THREADS)
k_linear(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias,
         const float* __restrict__ x2, const float* __restrict__ W2, const float* __restrict__ att,
         float* __restrict__ a_out, float* __restrict__ y, int64_t rows, int I, int O, int w_layout,
         int act, int LPR, int RPB) {
  extern __shared__ __align__(16) float lds[];
  float* Wt = lds;                      // [I][O]
  float* W2t = lds + (size_t)I * O;     // [I][O] if x2
  for (int idx = threadIdx.x; idx < I * O; idx += LIN_THREADS) {
    int i = idx / O, o = idx - i * O;
    Wt[idx] = w_layout ? W[idx] : W[(size_t)o * I + i];
    if (x2) W2t[idx] = w_layout ? W2[idx] : W2[(size_t)o * I + i];
  }
  __syncthreads();
  const int rl = threadIdx.x / LPR;
  const int og = threadIdx.x - rl * LPR;
  const int o0 = og * VEC;
  if (rl >= RPB) return;
  for (int64_t r = (int64_t)blockIdx.x * RPB + rl; r < rows; r += (int64_t)gridDim.x * RPB) {
    float acc[VEC], acc2[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) { acc[v] = 0.f; acc2[v] = 0.f; }
    const float* xr = x + r * I;
    const float* x2r = x2 ? x2 + r * I : nullptr;
    // the row's inputs eight at a time, all requested before the first is used (one dependent global load per input
    // and lane made this kernel latency-bound: 11.6 us on 57 k rows x 16 -> 64); same fmaf chains in the same order
    const bool v4 = (I & 3) == 0;
    for (int i0 = 0; i0 < I; i0 += 8) {
      float xa[8], xb[8];
      if (v4) {
#pragma unroll
        for (int u = 0; u < 8; u += 4) {
          float4 t = make_float4(0.f, 0.f, 0.f, 0.f), t2 = t;
          if (i0 + u < I) {
            t = *reinterpret_cast<const float4*>(xr + i0 + u);
            if (x2r) t2 = *reinterpret_cast<const float4*>(x2r + i0 + u);
          }
          xa[u] = t.x; xa[u + 1] = t.y; xa[u + 2] = t.z; xa[u + 3] = t.w;
          xb[u] = t2.x; xb[u + 1] = t2.y; xb[u + 2] = t2.z; xb[u + 3] = t2.w;
        }
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          xa[u] = i0 + u < I ? xr[i0 + u] : 0.f;
          xb[u] = (x2r && i0 + u < I) ? x2r[i0 + u] : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u;
        if (i < I) {
          const float* wrow = Wt + (size_t)i * O + o0;
#pragma unroll
          for (int v = 0; v < VEC; ++v) acc[v] = fmaf(xa[u], wrow[v], acc[v]);
          if (x2r) {
            const float* w2row = W2t + (size_t)i * O + o0;
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc2[v] = fmaf(xb[u], w2row[v], acc2[v]);
          }
        }
      }
    }
    if (att) {  // host guarantees LPR is a power of two <= 64 here
      float d = 0.f;
#pragma unroll
      for (int v = 0; v < VEC; ++v) d = fmaf(acc[v], att[o0 + v], d);
      for (int off = LPR >> 1; off > 0; off >>= 1) d += __shfl_xor(d, off, 64);
      if (og == 0) a_out[r] = d;
    }
    float out[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      float t = acc[v];
      if (bias) t += bias[o0 + v];
      if (x2r) t += acc2[v];
      out[v] = apply_act(t, act);
    }
    if (VEC == 4) {
      *reinterpret_cast<float4*>(y + r * O + o0) = make_float4(out[0], out[1], out[2], out[3]);
    } else {
#pragma unroll
      for (int v = 0; v < VEC; ++v) y[r * O + o0 + v] = out[v];
    }
  }
}

// a[r] = sum_o y[r,o]*att[o]   (fallback when the fused row-dot is not possible)
__global__ void k_rowdot(const float* __restrict__ y, const float* __restrict__ att, float* __restrict__ a,
                         int64_t rows, int O) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float d = 0.f;
  for (int o = 0; o < O; ++o) d = fmaf(y[r * O + o], att[o], d);
  a[r] = d;
}

// ---- weight / bias gradient: partial[g][p] over row chunk g, pair p = o*(I+1)+i ----
constexpr int BW_T = 32;       // rows per LDS tile (fewer when in_f is wide: the tile must fit 64 KB)
constexpr int BW_PPT = 16;     // pairs per thread per pass
constexpr int BW_PAIRS = LIN_THREADS * BW_PPT;

__global__ void __launch_bounds__(LIN_THREADS)
k_linear_bwd_w_partial(const float* __restrict__ gy, const float* __restrict__ x, float* __restrict__ partial,
                       int64_t rows, int I, int O, int64_t rows_per_chunk, int P, int T) {
  extern __shared__ __align__(16) float lds[];
  float* gy_t = lds;                       // [T][O]
  float* x_t = lds + (size_t)T * O;        // [T][I+1]  (last column = 1 for the bias)
  const int I1 = I + 1;
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_chunk;
  int64_t r_end = r_begin + rows_per_chunk;
  if (r_end > rows) r_end = rows;
  const int pbase = blockIdx.y * BW_PAIRS;
  int po[BW_PPT], pi[BW_PPT];
  float acc[BW_PPT];
#pragma unroll
  for (int k = 0; k < BW_PPT; ++k) {
    int p = pbase + k * LIN_THREADS + threadIdx.x;
    if (p < P) { po[k] = p / I1; pi[k] = p - po[k] * I1; } else { po[k] = -1; pi[k] = 0; }
    acc[k] = 0.f;
  }
  for (int64_t r0 = r_begin; r0 < r_end; r0 += T) {
    int nt = (int)((r_end - r0) < T ? (r_end - r0) : T);
    __syncthreads();
    for (int idx = threadIdx.x; idx < nt * O; idx += LIN_THREADS) gy_t[idx] = gy[r0 * O + idx];
    for (int idx = threadIdx.x; idx < nt * I1; idx += LIN_THREADS) {
      int t = idx / I1, i = idx - t * I1;
      x_t[idx] = i < I ? x[(r0 + t) * I + i] : 1.0f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < BW_PPT; ++k) {
      if (po[k] < 0) continue;
      float a = acc[k];
      for (int t = 0; t < nt; ++t) a = fmaf(gy_t[t * O + po[k]], x_t[t * I1 + pi[k]], a);
      acc[k] = a;
    }
  }
#pragma unroll
  for (int k = 0; k < BW_PPT; ++k) {
    int p = pbase + k * LIN_THREADS + threadIdx.x;
    if (p < P) partial[(size_t)blockIdx.x * P + p] = acc[k];
  }
}

// The same partials on the matrix cores (O a multiple of 16, TO x TI <= 8 accumulator tiles): gW = gy^T [x | 1] is a
// product with the ROWS as the reduction index, so a wave owns a row chunk and walks it four rows per
// v_mfma_f32_16x16x4_f32 -- A[i = lane & 15][k = lane >> 4] = gy[r + k][16 to + i], B[k][j = lane & 15] =
// [x | 1][r + k][16 ti + j] -- with both operands loaded STRAIGHT from global memory into the operand registers (a
// half-row of 64 B per row and instruction; eight steps = 32 rows of requests in flight per trip): no LDS tile, no
// barrier, no load -> sync -> multiply -> sync serialisation per 32 rows (the scalar kernel above spends 26 us on
// 57 k rows x 64 x 17 for that reason).  The MFMA is a k-ordered fmaf chain: a chunk's partial is the fmaf chain over
// its rows in row order, like the scalar kernel's.  One wave per workgroup: 512 chunks fill the chip.
template <int TO, int TI>
__global__ void __launch_bounds__(64)
k_linear_bwd_w_mfma(const float* __restrict__ gy, const float* __restrict__ x, float* __restrict__ partial,
                    int64_t rows, int I, int O, int64_t rows_per_chunk, int P) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x, li = lane & 15, lk = lane >> 4;
  const int I1 = I + 1;
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_chunk;
  int64_t r_end = r_begin + rows_per_chunk;
  if (r_end > rows) r_end = rows;
  f32x4 acc[TO][TI];
#pragma unroll
  for (int a = 0; a < TO; ++a)
#pragma unroll
    for (int b = 0; b < TI; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int U = (TO + TI) <= 3 ? 8 : 4;      // k-steps of requests in flight per trip
  for (int64_t r0 = r_begin; r0 < r_end; r0 += 4 * U) {
    float av[U][TO], bv[U][TI];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = r0 + 4 * u + lk;
      const bool ok = r < r_end;
      const int64_t rc = ok ? r : r_begin;       // (a row past the chunk: any valid address, the value is dropped)
#pragma unroll
      for (int a = 0; a < TO; ++a) {
        const float v = gy[rc * O + 16 * a + li];
        av[u][a] = ok ? v : 0.f;
      }
#pragma unroll
      for (int b = 0; b < TI; ++b) {
        const int c = 16 * b + li;
        float v = c == I ? 1.f : 0.f;
        if (c < I) v = x[rc * I + c];
        bv[u][b] = ok ? v : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int a = 0; a < TO; ++a)
#pragma unroll
        for (int b = 0; b < TI; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][a], bv[u][b], acc[a][b], 0, 0, 0);
  }
  float* out = partial + (size_t)blockIdx.x * P;
#pragma unroll
  for (int a = 0; a < TO; ++a)
#pragma unroll
    for (int b = 0; b < TI; ++b) {
      const int i = 16 * b + li;
      if (i >= I1) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) out[(16 * a + 4 * lk + q) * I1 + i] = acc[a][b][q];
    }
}

// gW / gb [p] = sum over the G row chunks of partial[g][p], in a FIXED tree (bitwise reproducible): a block owns 32
// parameters x 8 contiguous slices of chunks, a slice is summed in chunk order with 16 loads in flight, the slices are
// folded in slice order.  (One thread walking all G <= 512 chunks in a dependent chain took 55 us at G = 235.)
__global__ void __launch_bounds__(256)
k_linear_bwd_w_reduce(const float* __restrict__ partial, float* __restrict__ gW, float* __restrict__ gb, int G, int I,
                      int O, int P, int accumulate) {
  __shared__ float red[8][32];
  const int pl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int p = blockIdx.x * 32 + pl;
  const int per = (G + 7) / 8;
  const int g0 = sl * per, g1 = (g0 + per) < G ? (g0 + per) : G;
  float s = 0.f;
  if (p < P) {
    int g = g0;
    for (; g + 16 <= g1; g += 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = partial[(size_t)(g + u) * P + p];
#pragma unroll
      for (int u = 0; u < 16; ++u) s += v[u];
    }
    for (; g < g1; ++g) s += partial[(size_t)g * P + p];
  }
  red[sl][pl] = s;
  __syncthreads();
  if (sl != 0 || p >= P) return;
  s = 0.f;
#pragma unroll
  for (int q = 0; q < 8; ++q) s += red[q][pl];
  int I1 = I + 1;
  int o = p / I1, i = p - o * I1;
  if (i < I) {
    if (gW) gW[(size_t)o * I + i] = accumulate ? gW[(size_t)o * I + i] + s : s;
  } else if (gb) {
    gb[o] = accumulate ? gb[o] + s : s;
  }
}

__global__ void k_act_bwd(const float* __restrict__ gy, const float* __restrict__ y, float* __restrict__ g,
                          int64_t n, int act) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) g[i] = gy[i] * act_grad_from_output(y[i], act);
}

__global__ void k_act_fwd(const float* __restrict__ x, float* __restrict__ y, int64_t n, int act) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) y[i] = apply_act(x[i], act);
}

inline int chunks_for(int64_t rows) {
  int64_t g = (rows + 63) / 64;
  if (g < 1) g = 1;
  if (g > 512) g = 512;
  return (int)g;
}

inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

}  // namespace

extern "C" {

int hscn_linear_fwd(const float* x, const float* W, const float* bias, const float* x2, const float* W2,
                    const float* att, float* a_out, float* y, int64_t rows, int I, int O, int w_layout,
                    int act, void* stream_) {
  if (rows < 0 || I < 1 || O < 1 || !W || (rows > 0 && (!x || !y))) return HSCN_E_BADARG;
  if ((x2 == nullptr) != (W2 == nullptr) || (att == nullptr) != (a_out == nullptr)) return HSCN_E_BADARG;
  if (rows == 0) return 0;
  hipStream_t st = hscn_stream(stream_);
  const int VEC = (O % 4 == 0) ? 4 : 1;
  const int LPR = O / VEC;
  if (LPR > LIN_THREADS) return HSCN_E_UNSUPPORTED;
  const int RPB = LIN_THREADS / LPR;
  size_t lds = (size_t)I * O * 4 * (x2 ? 2 : 1);
  if (lds > 160 * 1024) return HSCN_E_UNSUPPORTED;
  bool fuse_att = att && is_pow2(LPR) && LPR <= 64;
  if (att && !fuse_att && (bias || act != HSCN_ACT_IDENTITY || x2)) return HSCN_E_UNSUPPORTED;
  int64_t nb = (rows + RPB - 1) / RPB;
  if (nb > 2048) nb = 2048;
  if (VEC == 4) {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)k_linear<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k_linear<4><<<(unsigned)nb, LIN_THREADS, lds, st>>>(x, W, bias, x2, W2, fuse_att ? att : nullptr,
                                                         fuse_att ? a_out : nullptr, y, rows, I, O, w_layout,
                                                         act, LPR, RPB);
  } else {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)k_linear<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k_linear<1><<<(unsigned)nb, LIN_THREADS, lds, st>>>(x, W, bias, x2, W2, fuse_att ? att : nullptr,
                                                         fuse_att ? a_out : nullptr, y, rows, I, O, w_layout,
                                                         act, LPR, RPB);
  }
  HSCN_RETURN_IF_LAUNCH_FAILED();
  if (att && !fuse_att) {
    k_rowdot<<<hscn_blocks(rows, 256), 256, 0, st>>>(y, att, a_out, rows, O);
    HSCN_RETURN_IF_LAUNCH_FAILED();
  }
  return 0;
}

size_t hscn_linear_bwd_w_workspace_bytes(int64_t rows, int I, int O) {
  return (size_t)chunks_for(rows) * (size_t)O * (I + 1) * 4;
}

int hscn_linear_bwd_w(const float* gy, const float* x, float* gW, float* gb, int64_t rows, int I, int O,
                      int accumulate, void* workspace, size_t workspace_bytes, void* stream_) {
  // in_f == 0 with x == NULL: bias gradient only (column sums of gy)
  if (rows < 0 || I < 0 || O < 1 || !workspace || (rows > 0 && (!gy || (I > 0 && !x)))) return HSCN_E_BADARG;
  if (workspace_bytes < hscn_linear_bwd_w_workspace_bytes(rows, I, O)) return HSCN_E_WORKSPACE;
  hipStream_t st = hscn_stream(stream_);
  const int P = O * (I + 1);
  const int G = chunks_for(rows);
  const int64_t rpc = (rows + G - 1) / G;
  int T = BW_T;
  while (T > 1 && (size_t)T * (O + I + 1) * 4 > 64 * 1024) T >>= 1;
  size_t lds = (size_t)T * (O + I + 1) * 4;
  if (lds > 64 * 1024) return HSCN_E_UNSUPPORTED;
  const int TO = O / 16, TI = (I + 1 + 15) / 16;
  static const bool no_mfma = getenv("HSCN_LINEAR_BWD_W") && atoi(getenv("HSCN_LINEAR_BWD_W")) == 0;   // A/B
  bool done = false;
  if (!no_mfma && O % 16 == 0 && TO * TI <= 8 && TO <= 4 && TI <= 4) {
#define HSCN_BWM(TO_, TI_) if (TO == TO_ && TI == TI_) { k_linear_bwd_w_mfma<TO_, TI_><<<G, 64, 0, st>>>(gy, x, (float*)workspace, rows, I, O, rpc, P); done = true; }
    HSCN_BWM(1, 1) HSCN_BWM(1, 2) HSCN_BWM(1, 3) HSCN_BWM(1, 4)
    HSCN_BWM(2, 1) HSCN_BWM(2, 2) HSCN_BWM(2, 3) HSCN_BWM(2, 4)
    HSCN_BWM(3, 1) HSCN_BWM(3, 2) HSCN_BWM(4, 1) HSCN_BWM(4, 2)
#undef HSCN_BWM
  }
  if (!done) {
    dim3 grid(G, (P + BW_PAIRS - 1) / BW_PAIRS);
    k_linear_bwd_w_partial<<<grid, LIN_THREADS, lds, st>>>(gy, x, (float*)workspace, rows, I, O, rpc, P, T);
  }
  k_linear_bwd_w_reduce<<<hscn_blocks(P, 32), 256, 0, st>>>((const float*)workspace, gW, gb, G, I, O, P,
                                                            accumulate);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

int hscn_act_fwd(const float* x, float* y, int64_t count, int act, void* stream_) {
  if (count < 0 || (count > 0 && (!x || !y))) return HSCN_E_BADARG;
  if (count == 0) return 0;
  unsigned nb = hscn_blocks(count, 256);
  if (nb > 4096) nb = 4096;
  k_act_fwd<<<nb, 256, 0, hscn_stream(stream_)>>>(x, y, count, act);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

int hscn_act_bwd(const float* gy, const float* y, float* g, int64_t count, int act, void* stream_) {
  if (count < 0 || (count > 0 && (!gy || !y || !g))) return HSCN_E_BADARG;
  if (count == 0) return 0;
  unsigned nb = hscn_blocks(count, 256);
  if (nb > 4096) nb = 4096;
  k_act_bwd<<<nb, 256, 0, hscn_stream(stream_)>>>(gy, y, g, count, act);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

}  // extern "C"
