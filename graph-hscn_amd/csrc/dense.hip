// Dense route of dense_mincut_pool (reference model/hscn.py:61-63):
// the cluster-assignment contractions  A S,  S^T (A S),  S^T S,  S^T X  on the matrix cores.
//
// gfx950 has an exact-fp32 MFMA (v_mfma_f32_16x16x4_f32: D = A*B + C as a k-ordered fmaf
// chain, no reduced-precision step), so the 1e-5 parity bar holds without any split-precision
// trick.  One batched kernel serves all four products: a workgroup owns a 64 x N tile of one
// batch element (4 waves x 16 rows, N/16 accumulator tiles each), A and B tiles are staged
// through LDS with coalesced 16-byte global loads (zero padded at the ragged edges), the
// optional transpose of A (S^T ...) is a different LDS read pattern, not a different load.
// MFMA-bound only at PascalVOC-SP sizes (n ~ 480, K = 64: 33 MFLOP per graph); at Peptides
// sizes the sparse route (mincut.hip) does 8x less work.
#include "hscn_common.h"
#include <type_traits>
#include <cstdlib>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
// 16-byte global load from a 4-byte aligned address (rows of an odd-width matrix): gfx950 global
// memory takes dword-aligned dwordx4 accesses
struct __attribute__((packed, aligned(4))) F4U { float x, y, z, w; };
__device__ __forceinline__ float4 ld4u(const float* p) {
  const F4U v = *reinterpret_cast<const F4U*>(p);
  return make_float4(v.x, v.y, v.z, v.w);
}

constexpr int GM = 64;   // rows of C per workgroup
constexpr int GK = 32;   // reduction depth per LDS stage
constexpr int GNMAX = 64;

// C[b] (M x N) = op(A[b]) (M x Kd) * B[b] (Kd x N);  op = transpose when TA (A stored Kd x M)
// Two LDS stages: the next k-slab travels HBM -> registers while the matrix cores work on the current
// one and is parked in the other stage afterwards -- one barrier per slab, loads hidden under MFMA.
// NT (N tiles of 16) is a template parameter: the slab loop is branch-free, the operand reads of a slab
// are issued together and the MFMAs follow back to back.
// RS (TA = 0 only): also write the row sums of A (the degrees, adj . 1) to rsum [batch][M]: the tile is in
// LDS anyway -- four threads add a row's 32 slab entries, quads fold by DPP -- which saves a pass over the
// adjacency (the largest array of the dense route) and a launch.
// Up to three products that share op(A) run as one launch (blockIdx.z picks the right-hand side): the
// cluster-space contractions S^T (A S), S^T S, S^T X are 128-workgroup launches each -- together they fill
// the chip.
struct BRhs {
  const float* B;
  float* C;
  int N;
  int64_t ldb, ldc, sB, sC;
  int ragB, ragC;   // ragged batch: the operand of batch b starts at row nptr[b] of ONE flat [N_total, ld] array
};
// A batch of graphs of DIFFERENT sizes (PascalVOC-SP: n in [395, 500]): node-indexed operands are flat
// [N_total, width] arrays -- graph b owns rows [nptr[b], nptr[b + 1]) -- while per-graph matrices (the adjacency
// [B, nmax, nmax], zero beyond n_b; the K x K cluster-space results) keep a uniform batch stride.
// a: A is node-indexed;  m / k: M / Kd is this graph's node count n_b instead of the uniform argument.
struct Rag {
  const int32_t* nptr;
  int a, m, k;
};
struct BRhs3 {
  BRhs r[3];
};
template <int TA, int NT, int RS>
__global__ void __launch_bounds__(256)
k_bgemm(const float* __restrict__ A, const BRhs3 R, int M, int Kd, int64_t lda, int64_t sA,
        float* __restrict__ rsum, const Rag G) {
  const float* __restrict__ Bm = R.r[blockIdx.z].B;
  float* __restrict__ C = R.r[blockIdx.z].C;
  const int N = R.r[blockIdx.z].N;
  const int64_t ldb = R.r[blockIdx.z].ldb, ldc = R.r[blockIdx.z].ldc, sB = R.r[blockIdx.z].sB, sC = R.r[blockIdx.z].sC;
  const int row0 = G.nptr ? G.nptr[blockIdx.y] : 0;
  if (G.nptr) {
    const int nb = G.nptr[blockIdx.y + 1] - row0;
    if (G.m) M = nb;
    if (G.k) Kd = nb;
    if ((int)blockIdx.x * GM >= M) return;      // (the grid is sized for the largest graph; the whole workgroup leaves)
  }
  constexpr int AST = GK + 4;                           // padded row stride of the [m][k] image (bank spread, 16-B aligned)
  constexpr int ASZ = TA ? GK * GM : GM * AST;
  __shared__ __align__(16) float As[2][ASZ];          // TA=0: [m][k];  TA=1: [k][m] (stride GM)
  // [k][n], rows padded by 16 words: the four k-rows a wave reads together (lane>>4) fall on disjoint banks
  constexpr int BST = GNMAX + 16;
  __shared__ __align__(16) float Bs[2][GK * BST];
  const int b = blockIdx.y;
  const int m0 = blockIdx.x * GM;
  A += (G.nptr && G.a) ? (size_t)row0 * lda : (size_t)b * sA;
  Bm += (G.nptr && R.r[blockIdx.z].ragB) ? (size_t)row0 * ldb : (size_t)b * sB;
  C += (G.nptr && R.r[blockIdx.z].ragC) ? (size_t)row0 * ldc : (size_t)b * sC;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int NP = NT * 16;
  constexpr int APT = GM * GK / 4 / 256;   // float4 pieces of the A slab per thread (2)
  constexpr int BPT = GK * GNMAX / 4 / 256;  // of the B slab, at most (2)
  float4 ra[APT], rb[BPT];

  auto ld_guard = [](const float* p, int c, int lim) {   // p[0..3] with columns c..c+3 < lim, else 0
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c + 3 < lim) v = ld4u(p);
    else {
      if (c < lim) v.x = p[0];
      if (c + 1 < lim) v.y = p[1];
      if (c + 2 < lim) v.z = p[2];
      if (c + 3 < lim) v.w = p[3];
    }
    return v;
  };
  // a thread's pieces of a slab sit at the same (row, column) of every slab: the index arithmetic
  // (divisions by the runtime tile width) is done once, the slab loop only adds k0
  int a_r[APT], a_c[APT], b_r[BPT], b_c[BPT], a_lds[APT], b_lds[BPT];
  bool b_on[BPT];
#pragma unroll
  for (int u = 0; u < APT; ++u) {
    const int idx = threadIdx.x + u * 256;
    if (TA == 0) { a_r[u] = idx / (GK / 4); a_c[u] = (idx % (GK / 4)) * 4; a_lds[u] = a_r[u] * AST + a_c[u]; }
    else { a_r[u] = idx / (GM / 4); a_c[u] = (idx % (GM / 4)) * 4; a_lds[u] = a_r[u] * GM + a_c[u]; }
  }
#pragma unroll
  for (int u = 0; u < BPT; ++u) {
    const int idx = threadIdx.x + u * 256;
    b_on[u] = idx < GK * NP / 4;
    b_r[u] = idx / (NP / 4);
    b_c[u] = (idx % (NP / 4)) * 4;
    b_lds[u] = b_r[u] * BST + b_c[u];
  }
  auto fetch = [&](int k0) {
#pragma unroll
    for (int u = 0; u < APT; ++u) {
      ra[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (TA == 0) {   // A tile rows m0.., cols k0..k0+31
        const int gr = m0 + a_r[u], gc = k0 + a_c[u];
        if (gr < M) ra[u] = ld_guard(A + (size_t)gr * lda + gc, gc, Kd);
      } else {         // A stored [Kd][M]: tile rows k0.., cols m0..m0+63
        const int gr = k0 + a_r[u], gc = m0 + a_c[u];
        if (gr < Kd) ra[u] = ld_guard(A + (size_t)gr * lda + gc, gc, M);
      }
    }
#pragma unroll
    for (int u = 0; u < BPT; ++u) {
      rb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      const int gr = k0 + b_r[u];
      if (b_on[u] && gr < Kd) rb[u] = ld_guard(Bm + (size_t)gr * ldb + b_c[u], b_c[u], N);
    }
  };
  auto park = [&](int buf) {
#pragma unroll
    for (int u = 0; u < APT; ++u) *reinterpret_cast<float4*>(&As[buf][a_lds[u]]) = ra[u];
#pragma unroll
    for (int u = 0; u < BPT; ++u)
      if (b_on[u]) *reinterpret_cast<float4*>(&Bs[buf][b_lds[u]]) = rb[u];
  };

  fetch(0);
  park(0);
  __syncthreads();
  int buf = 0;
  float rs = 0.f;   // RS: running sum of row (threadIdx.x / 4) of the A tile
  for (int k0 = 0; k0 < Kd; k0 += GK) {
    const bool more = k0 + GK < Kd;
    if (more) fetch(k0 + GK);
    const float* as = As[buf];
    const float* bs = Bs[buf];
    // operands of the whole slab first (A: lane holds A[i = lane&15][k = lane>>4]; B[k = lane>>4][j = lane&15]),
    // then GK/4 x NT MFMAs
    float av[GK / 4], bv[GK / 4][NT];
#pragma unroll
    for (int q = 0; q < GK / 4; ++q) {
      const int kk = q * 4;
      av[q] = TA ? as[(kk + lk) * GM + wave * 16 + li] : as[(wave * 16 + li) * AST + kk + lk];
#pragma unroll
      for (int t = 0; t < NT; ++t) bv[q][t] = bs[(kk + lk) * BST + t * 16 + li];
    }
    if (RS && TA == 0) {
      const float4 p0 = *reinterpret_cast<const float4*>(&as[(threadIdx.x >> 2) * AST + (threadIdx.x & 3) * 8]);
      const float4 p1 = *reinterpret_cast<const float4*>(&as[(threadIdx.x >> 2) * AST + (threadIdx.x & 3) * 8 + 4]);
      float q8 = ((p0.x + p0.y) + (p0.z + p0.w)) + ((p1.x + p1.y) + (p1.z + p1.w));
      q8 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(q8), 0xB1, 0xF, 0xF, true));   // lane ^ 1
      q8 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(q8), 0x4E, 0xF, 0xF, true));   // lane ^ 2
      rs += q8;
    }
#pragma unroll
    for (int q = 0; q < GK / 4; ++q)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bv[q][t], acc[t], 0, 0, 0);
    if (more) park(buf ^ 1);   // the other stage: nobody reads it during this slab
    __syncthreads();
    buf ^= 1;
  }
  if (RS && TA == 0 && (threadIdx.x & 3) == 0) {
    const int row = m0 + (threadIdx.x >> 2);
    if (row < M) rsum[(G.nptr ? (size_t)row0 : (size_t)b * M) + row] = rs;
  }
  // C/D layout: col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = t * 16 + li;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + wave * 16 + lk * 4 + r;
      if (row < M && col < N) C[(size_t)row * ldc + col] = acc[t][r];
    }
  }
}

// ---- the adjacency products A S (and A^T S for the backward) on v_mfma_f32_32x32x2_f32 -----------------------------
// C[b] (n_b x K) = op(A[b]) (n_b x n_b) * S[b] (n_b x K), K <= 64: by far the largest contraction of the dense route
// (2 K n^2 of the 2 K n^2 + 2 n K^2 flops SURVEY.md 8(d) counts; 33 MFLOP per PascalVOC-SP graph).
// A workgroup (4 waves) owns 128 rows of one graph; a wave owns 32 of them across all K columns -- NT accumulators of
// 32 x 32 -- and walks the reduction in slabs of 32: 16 x NT back-to-back MFMAs per slab, 64 cycles each (the f32 MFMA
// peak: 64 FLOP / clk / SIMD).  Both operands are staged through LDS, two stages, the next slab travelling
// HBM -> registers under the current slab's MFMAs (one barrier per slab):
//   A image  [k][m] (k-major, row stride 132 words): an MFMA operand read is 32 consecutive words per half wave;
//            the transpose of a row-major adjacency tile happens in the staging stores (TA = 0), A^T needs none (TA = 1);
//   S image  [k][n] row-major.
// AT = float: the adjacency as dense_mincut_pool's caller hands it over ([B, nmax, nmax] floats: 4 n^2 bytes per graph
// and pass -- 111 MB for a PascalVOC-SP batch, which makes the product HBM-bound);
// AT = uint8_t: the counts as bytes ([B, nmax, lda8], lda8 = nmax rounded up to 32, zero filled; what the model's own
// batched route builds): 4x less traffic, converted exactly on the way into LDS (v_cvt_f32_ubyte*).
// RS (TA = 0): also the row sums of A (degrees), from the LDS image.
template <typename AT, int TA, int NT, bool RS, bool KV = false>
__global__ void __launch_bounds__(256)
k_adj_s(const AT* __restrict__ adj, const float* __restrict__ S, float* __restrict__ C, float* __restrict__ rsum,
        const int32_t* __restrict__ nptr, int n_uniform, int nmax, int64_t lda, int K,
        const int32_t* __restrict__ asym = nullptr) {
  // asym (A^T S of the backward only): asym[b] == 0 says A_b is symmetric -- A_b^T S = A_b S, which the forward kept:
  // this graph's workgroups have nothing to do (the consumer reads the forward's product for it)
  if (asym && asym[blockIdx.y] == 0) return;
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  constexpr int BM = 128, BK = 32, AS_ST = BM + 4, BS_ST = 64;
  __shared__ __align__(16) float As[2][BK * AS_ST];
  __shared__ __align__(16) float Bs[2][BK * BS_ST];
  __shared__ float rs_red[256];
  const int b = blockIdx.y, m0 = blockIdx.x * BM;
  const int row0 = nptr ? nptr[b] : b * n_uniform;
  const int n = nptr ? nptr[b + 1] - row0 : n_uniform;
  if (m0 >= n) return;
  const AT* Ab = adj + (size_t)b * nmax * lda;
  const float* Sb = S + (size_t)row0 * K;
  float* Cb = C + (size_t)row0 * K;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, lh = lane >> 5;
  f32x16 acc[NT];
#pragma unroll
  for (int q = 0; q < NT; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  // staging registers: the A slab is 128 x 32 (TA = 0) or 32 x 128 (TA = 1) elements, 16 per thread; the S slab 32 x 64
  float ra[16];
  float4 rb[2];
  uint4 wraw = make_uint4(0u, 0u, 0u, 0u);
  bool rok[2] = {true, true};
  // KV (byte adjacency, K % 4 == 0): both slabs are requested branch-free from clamped addresses and nothing touches
  // the loaded registers until they are PARKED behind the slab's MFMAs (the byte -> float conversion and the zeroing of
  // rows / columns outside the graph happen there): a request issued at the top of a slab has the whole slab to arrive.
  // Entries of A outside the graph may then be any byte -- they only ever meet rows of S that are exact zeros, or land
  // in output rows that are not stored.
  auto fetch = [&](int k0) {
    if constexpr (sizeof(AT) == 1 && KV) {
      const int r = TA ? (t >> 3) : (t >> 1), c16 = TA ? (t & 7) * 16 : (t & 1) * 16;
      const int gr = TA ? k0 + r : m0 + r, gc = TA ? m0 + c16 : k0 + c16;
      const int grc = gr < n ? gr : n - 1, gcc = gc + 16 <= lda ? gc : (int)lda - 16;
      wraw = *reinterpret_cast<const uint4*>(Ab + (size_t)grc * lda + gcc);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int sr = k0 + (t >> 4) + 16 * u, c4 = (t & 15) * 4;
        const int src = sr < n ? sr : n - 1, cc = c4 + 4 <= K ? c4 : K - 4;
        rb[u] = ld4u(Sb + (size_t)src * K + cc);
        rok[u] = sr < n && c4 < K;
      }
      return;
    }
    if constexpr (sizeof(AT) == 1) {
      // 16 consecutive bytes per thread (rows are padded to 32 and zero filled: no column guard)
      const int r = TA ? (t >> 3) : (t >> 1), c16 = TA ? (t & 7) * 16 : (t & 1) * 16;
      const int gr = TA ? k0 + r : m0 + r, gc = TA ? m0 + c16 : k0 + c16;
      uint4 w = make_uint4(0u, 0u, 0u, 0u);
      if (gr < n && gc < lda) w = *reinterpret_cast<const uint4*>(Ab + (size_t)gr * lda + gc);
      wraw = w;
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        // TA = 0: rows m0 + (t>>3) + 32u, four columns k0 + 4 (t&7);  TA = 1: rows k0 + (t>>5) + 8u, columns m0 + 4 (t&31)
        const int gr = TA ? k0 + (t >> 5) + 8 * u : m0 + (t >> 3) + 32 * u;
        const int gc = TA ? m0 + (t & 31) * 4 : k0 + (t & 7) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gr < n) {
          const float* p = reinterpret_cast<const float*>(Ab) + (size_t)gr * lda + gc;
          if (gc + 3 < n) v = ld4u(p);
          else {
            if (gc < n) v.x = p[0];
            if (gc + 1 < n) v.y = p[1];
            if (gc + 2 < n) v.z = p[2];
          }
        }
        ra[4 * u + 0] = v.x; ra[4 * u + 1] = v.y; ra[4 * u + 2] = v.z; ra[4 * u + 3] = v.w;
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int gr = k0 + (t >> 4) + 16 * u, c4 = (t & 15) * 4;
      rb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gr < n && c4 < K) {
        const float* p = Sb + (size_t)gr * K + c4;
        if (c4 + 3 < K) rb[u] = ld4u(p);
        else {
          rb[u].x = p[0];
          if (c4 + 1 < K) rb[u].y = p[1];
          if (c4 + 2 < K) rb[u].z = p[2];
        }
      }
    }
  };
  auto park = [&](int buf) {
    float* as = As[buf];
    if constexpr (sizeof(AT) == 1) {
      const unsigned ww[4] = {wraw.x, wraw.y, wraw.z, wraw.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ra[4 * j + 0] = (float)(ww[j] & 0xffu);
        ra[4 * j + 1] = (float)((ww[j] >> 8) & 0xffu);
        ra[4 * j + 2] = (float)((ww[j] >> 16) & 0xffu);
        ra[4 * j + 3] = (float)(ww[j] >> 24);
      }
      if (TA) {      // 16 consecutive m of row k
        const int k = t >> 3, c16 = (t & 7) * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<float4*>(&as[k * AS_ST + c16 + 4 * j]) = make_float4(ra[4 * j], ra[4 * j + 1], ra[4 * j + 2], ra[4 * j + 3]);
      } else {       // 16 consecutive k of row m: transposed on the way in
        const int m = t >> 1, c16 = (t & 1) * 16;
#pragma unroll
        for (int j = 0; j < 16; ++j) as[(c16 + j) * AS_ST + m] = ra[j];
      }
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (TA) {
          const int k = (t >> 5) + 8 * u, m4 = (t & 31) * 4;
          *reinterpret_cast<float4*>(&as[k * AS_ST + m4]) = make_float4(ra[4 * u], ra[4 * u + 1], ra[4 * u + 2], ra[4 * u + 3]);
        } else {
          const int m = (t >> 3) + 32 * u, k4 = (t & 7) * 4;
#pragma unroll
          for (int j = 0; j < 4; ++j) as[(k4 + j) * AS_ST + m] = ra[4 * u + j];
        }
      }
    }
    float* bs = Bs[buf];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float4 v = rb[u];
      const bool ok = rok[u];
      *reinterpret_cast<float4*>(&bs[((t >> 4) + 16 * u) * BS_ST + (t & 15) * 4]) =
          make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
    }
  };
  fetch(0);
  park(0);
  __syncthreads();
  int buf = 0;
  float rs = 0.f;
  for (int k0 = 0; k0 < n; k0 += BK) {
    const bool more = k0 + BK < n;
    if (more) fetch(k0 + BK);
    if constexpr (KV) __builtin_amdgcn_sched_barrier(0);   // the requests go out here, not where the scheduler sinks them
    const float* as = As[buf];
    const float* bs = Bs[buf];
    if (RS && !TA) {      // thread (m = t & 127, half = t >> 7) adds 16 of the slab's 32 entries of row m
      const float* p = as + (t >> 7) * 16 * AS_ST + (t & 127);
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) q += p[j * AS_ST];
      rs += q;
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float a = as[(kk + lh) * AS_ST + wave * 32 + li];
#pragma unroll
      for (int q = 0; q < NT; ++q) {
        const float bv = bs[(kk + lh) * BS_ST + q * 32 + li];
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[q], 0, 0, 0);
      }
    }
    if constexpr (KV) __builtin_amdgcn_sched_barrier(0);
    if (more) park(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  if (RS && !TA) {
    rs_red[t] = rs;
    __syncthreads();
    if (t < 128 && m0 + t < n) rsum[(size_t)row0 + m0 + t] = rs_red[t] + rs_red[t + 128];
  }
  // C/D layout of the 32x32 forms: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int q = 0; q < NT; ++q) {
    const int col = q * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row < n && col < K) Cb[(size_t)row * K + col] = acc[q][r];
    }
  }
}

// ---- A S with the adjacency read STRAIGHT INTO THE MATRIX-CORE OPERAND REGISTERS (TA = 0) ------------------------
// v_mfma_f32_32x32x2_f32 takes A[i = lane & 31][k = lane >> 5]: which two values of k a step multiplies is free as long
// as both operands agree.  Step j of a 32-deep slab uses k = 16 h + j for lane half h, so lane (i, h) needs exactly
// the 16 consecutive entries A[row i][k0 + 16 h .. + 15]: ONE 16-byte load of byte counts (four of floats), no LDS
// image of the adjacency, no transposing stores, nothing of A behind the slab barrier.  The loads run two slabs ahead
// (4 registers per slab for bytes); only the S slab (L2-resident, shared by the graph's workgroups) goes through LDS,
// read as rows 16 h + j.  The row sums (degrees) are the sums of the operands a lane feeds, its two halves folded by
// one cross-half shuffle.
// WV: waves per workgroup = 32-row groups per row tile (4: 128 rows; 2: 64 rows -- half the padding of a ragged graph's
// last tile and twice as many, smaller units for the dispatcher to balance).
template <typename AT, int NT, bool RS, bool KV = true, int WV = 4>
__global__ void __launch_bounds__(64 * WV)
k_adj_s_direct(const AT* __restrict__ adj, const float* __restrict__ S, float* __restrict__ C, float* __restrict__ rsum,
               const int32_t* __restrict__ nptr, int n_uniform, int nmax, int64_t lda, int K) {
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  constexpr int BM = 32 * WV, BK = 32, BS_ST = 64;
  constexpr int BP = 128 / (16 * WV), BR = 4 * WV;   // S slab: passes per thread, rows per pass (16 float4 per row)
  constexpr bool U8 = sizeof(AT) == 1;
  constexpr int AR = U8 ? 4 : 16;           // registers of one slab's A entries
  __shared__ __align__(16) float Bs[2][BK * BS_ST];
  const int b = blockIdx.y, m0 = blockIdx.x * BM;
  const int row0 = nptr ? nptr[b] : b * n_uniform;
  const int n = nptr ? nptr[b + 1] - row0 : n_uniform;
  if (m0 >= n) return;
  // (blockIdx.z: a slice of 32 NT columns of S and C -- the launch uses one slice: splitting K = 64 over two
  // workgroups was measured 2 us slower)
  const int c0 = blockIdx.z * 32 * NT;
  const AT* Ab = adj + (size_t)b * nmax * lda;
  const float* Sb = S + (size_t)row0 * K + c0;
  float* Cb = C + (size_t)row0 * K + c0;
  const int Kc = K - c0;      // columns left from c0 on
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, lh = lane >> 5;
  const int row = m0 + wave * 32 + li;
  const AT* arow = Ab + (size_t)(row < n ? row : 0) * lda + 16 * lh;
  f32x16 acc[NT];
#pragma unroll
  for (int q = 0; q < NT; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  const int k_last = ((n - 1) / BK) * BK;    // first column of the last slab
  auto fetchA = [&](int k0, unsigned (&ra)[AR]) {
    if constexpr (U8) {
      // branch-free: rows are padded to 32 columns and zero filled (no column guard); a row past the graph reads row 0
      // (its products land in accumulator rows that are never stored, its row sum is never stored); a slab past the
      // end re-reads the last one and is never multiplied.  An unconditional load keeps the request in flight across
      // the slab barrier instead of behind an exec branch with its own wait.
      const uint4 w = *reinterpret_cast<const uint4*>(arow + (k0 < k_last ? k0 : k_last));
      ra[0] = w.x; ra[1] = w.y; ra[2] = w.z; ra[3] = w.w;
      return;
    }
#pragma unroll
    for (int u = 0; u < AR; ++u) ra[u] = 0u;
    if (row < n && k0 < n) {
      if constexpr (U8) {
      } else {
        const float* p = reinterpret_cast<const float*>(arow) + k0;
        const int c = k0 + 16 * lh;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          const int cc = c + 4 * u;
          if (cc + 3 < n) v = ld4u(p + 4 * u);
          else {
            if (cc < n) v.x = p[4 * u];
            if (cc + 1 < n) v.y = p[4 * u + 1];
            if (cc + 2 < n) v.z = p[4 * u + 2];
          }
          ra[4 * u] = __float_as_uint(v.x); ra[4 * u + 1] = __float_as_uint(v.y);
          ra[4 * u + 2] = __float_as_uint(v.z); ra[4 * u + 3] = __float_as_uint(v.w);
        }
      }
    }
  };
  auto a_of = [&](const unsigned (&ra)[AR], int j) -> float {
    if constexpr (U8) return (float)((ra[j >> 2] >> (8 * (j & 3))) & 0xffu);
    else return __uint_as_float(ra[j]);
  };
  // KV (K % 4 == 0, what the route's cluster counts are): the S slab is requested branch-free -- clamped addresses,
  // 16-byte pieces -- and rows / columns outside the graph become exact zeros by a select applied when the piece is
  // PARKED (after the slab's MFMAs), so nothing consumes the load before then.
  // Three register sets for the S slab too: the slab parked at the end of step s was requested at the top of step s - 1.
  struct BSet { float4 v[BP]; };
  auto fetchB = [&](int k0, BSet& rb) {
#pragma unroll
    for (int u = 0; u < BP; ++u) {
      const int gr = k0 + (t >> 4) + BR * u, c4 = (t & 15) * 4;
      if constexpr (KV) {
        const int grl = n - 1, grc = gr < n ? gr : grl, cc = c4 + 4 <= Kc ? c4 : Kc - 4;
        rb.v[u] = ld4u(Sb + (size_t)grc * K + cc);
      } else {
        rb.v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gr < n && c4 < Kc && c4 < 32 * NT) {
          const float* p = Sb + (size_t)gr * K + c4;
          if (c4 + 3 < Kc) rb.v[u] = ld4u(p);
          else {
            rb.v[u].x = p[0];
            if (c4 + 1 < Kc) rb.v[u].y = p[1];
            if (c4 + 2 < Kc) rb.v[u].z = p[2];
          }
        }
      }
    }
  };
  auto parkB = [&](int buf, int k0, const BSet& rb) {      // (k0: the slab the set holds)
#pragma unroll
    for (int u = 0; u < BP; ++u) {
      const float4 v = rb.v[u];
      const int sr = (t >> 4) + BR * u;                       // row of the slab
      const int gr = k0 + sr, c4 = (t & 15) * 4;
      const bool ok = !KV || (gr < n && c4 < Kc && c4 < 32 * NT);
      // rows 16 .. 31 (the operands of lane half 1) sit 32 columns over, modulo the row: the two halves of a wave read
      // rows 16 apart -- 1 024 words, the same banks -- in one instruction; now they hit disjoint halves of the banks
      *reinterpret_cast<float4*>(&Bs[buf][sr * BS_ST + (((t & 15) * 4) ^ (32 * (sr >> 4)))]) =
          make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
    }
  };
  unsigned a0[AR], a1[AR], a2[AR];
  BSet s0, s1, s2;
  fetchB(0, s0);
  fetchA(0, a0);
  fetchB(BK, s1);              // (past the end: clamped rows, never parked)
  fetchA(BK, a1);
  parkB(0, 0, s0);
  __syncthreads();
  int buf = 0;
  float rs = 0.f;
  // One slab: request S and A of the slab after next, multiply, park the NEXT slab's S (requested a slab ago), barrier.
  // The register sets ROTATE BY NAME (the loop is unrolled three slabs deep): copying them down the pipeline would make
  // every slab wait for the load it has just issued.  The barrier waits for LDS only (`s_waitcnt lgkmcnt(0)`, not
  // __syncthreads, which drains the vector-memory counter too): a request has two full slabs to arrive.
  auto slab = [&](auto steady, const unsigned (&cur)[AR], unsigned (&tgt)[AR], const BSet& spark, BSet& sfetch, int k0) {
    const bool more = decltype(steady)::value || k0 + BK < n;
    if (decltype(steady)::value || k0 + 2 * BK < n) fetchB(k0 + 2 * BK, sfetch);
    fetchA(k0 + 2 * BK, tgt);                 // (past the end: the last slab again, never multiplied)
    __builtin_amdgcn_sched_barrier(0);        // the requests go out HERE: the scheduler had sunk them behind the MFMAs
    const float* bs = Bs[buf] + lh * 16 * BS_ST;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float a = a_of(cur, j);
      if (RS) rs += a;
#pragma unroll
      for (int q = 0; q < NT; ++q)
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bs[j * BS_ST + ((q * 32 + li) ^ (32 * lh))], acc[q], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) parkB(buf ^ 1, k0 + BK, spark);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    buf ^= 1;
  };
  // main loop: triples of slabs whose requests all lie inside the graph (no conditionals: the compiler's wait counters
  // stay exact across the back edge); then the same rotation with the conditions evaluated for what is left
  int k0 = 0;
  for (; k0 + 4 * BK < n; k0 += 3 * BK) {
    slab(std::true_type{}, a0, a2, s1, s2, k0);
    slab(std::true_type{}, a1, a0, s2, s0, k0 + BK);
    slab(std::true_type{}, a2, a1, s0, s1, k0 + 2 * BK);
  }
  for (; k0 < n; k0 += 3 * BK) {
    slab(std::false_type{}, a0, a2, s1, s2, k0);
    if (k0 + BK < n) slab(std::false_type{}, a1, a0, s2, s0, k0 + BK);
    if (k0 + 2 * BK < n) slab(std::false_type{}, a2, a1, s0, s1, k0 + 2 * BK);
  }
  if (RS) {
    rs += __shfl_xor(rs, 32, 64);
    if (lh == 0 && row < n && blockIdx.z == 0) rsum[(size_t)row0 + row] = rs;
  }
#pragma unroll
  for (int q = 0; q < NT; ++q) {
    const int col = q * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int orow = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (orow < n && col < Kc) Cb[(size_t)orow * K + col] = acc[q][r];
    }
  }
}

template <typename AT>
int launch_adj_s(const AT* adj, const float* S, float* C, float* rsum, const int32_t* nptr, int64_t B, int n, int nmax,
                 int64_t lda, int K, int transA, hipStream_t st, const int32_t* asym = nullptr) {
  dim3 grid((unsigned)((nmax + 127) / 128), (unsigned)B);
  const int NT = K > 32 ? 2 : 1;
  static const bool via_lds = getenv("HSCN_DENSE_AS") && atoi(getenv("HSCN_DENSE_AS")) == 32;   // A/B: A staged through LDS
  // measured on a PascalVOC-SP batch (B = 128, K = 64, byte adjacency; profiles/r03_dense_*): S through LDS 51-52 us
  // (49-57 % MFMA-busy), both operands in registers 65.6 us (a kernel since removed), A through LDS as well 59 us (=32), the
  // 16x16x4 kernel on a float adjacency 49 us (=16)
  if (!transA && !via_lds) {
    // K > 32: one workgroup per row tile with two 32-column accumulators per wave (two independent MFMA chains, A read
    // once).  Measured and removed: 32-column halves on separate workgroups (+2 us on the forward call), 64-row tiles
    // (+18 us), LDS padding to cap the workgroups per CU (no change).
    dim3 gd(grid.x, grid.y, 1);
    // 256-row tiles (eight waves share a staged S slab) whenever they pad the largest graph no further than 128-row tiles
    // do -- PascalVOC-SP's 395 .. 500 nodes: 512 rows either way; 42.5 vs 44.0 us, MFMA-busy 0.60 vs 0.58.
    // HSCN_DENSE_ROWS=128 / 256 forces one form (A/B)
    static const int rows_forced = getenv("HSCN_DENSE_ROWS") ? atoi(getenv("HSCN_DENSE_ROWS")) : 0;
    const int rows_env = rows_forced ? rows_forced : (((nmax + 255) / 256) * 256 == ((nmax + 127) / 128) * 128 ? 256 : 128);
    dim3 gd8((unsigned)((nmax + 255) / 256), gd.y, 1);
#define HSCN_ADJ_D(NT_, RS_, KV_) do { \
      if (rows_env == 256) k_adj_s_direct<AT, NT_, RS_, KV_, 8><<<gd8, 512, 0, st>>>(adj, S, C, rsum, nptr, n, nmax, lda, K); \
      else k_adj_s_direct<AT, NT_, RS_, KV_, 4><<<gd, 256, 0, st>>>(adj, S, C, rsum, nptr, n, nmax, lda, K); } while (0)
#define HSCN_ADJ_DK(NT_, RS_) do { if ((K & 3) == 0) HSCN_ADJ_D(NT_, RS_, true); else HSCN_ADJ_D(NT_, RS_, false); } while (0)
    if (rsum) { if (NT == 2) HSCN_ADJ_DK(2, true); else HSCN_ADJ_DK(1, true); }
    else { if (NT == 2) HSCN_ADJ_DK(2, false); else HSCN_ADJ_DK(1, false); }
#undef HSCN_ADJ_DK
#undef HSCN_ADJ_D
    HSCN_RETURN_IF_LAUNCH_FAILED();
    return 0;
  }
#define HSCN_ADJ_S(TA_, NT_, RS_) k_adj_s<AT, TA_, NT_, RS_><<<grid, 256, 0, st>>>(adj, S, C, rsum, nptr, n, nmax, lda, K)
  if (transA && sizeof(AT) == 1 && (K & 3) == 0 && lda >= 16) {     // the branch-free requests (see k_adj_s, KV)
    if (NT == 2) k_adj_s<AT, 1, 2, false, true><<<grid, 256, 0, st>>>(adj, S, C, rsum, nptr, n, nmax, lda, K, asym);
    else k_adj_s<AT, 1, 1, false, true><<<grid, 256, 0, st>>>(adj, S, C, rsum, nptr, n, nmax, lda, K, asym);
    HSCN_RETURN_IF_LAUNCH_FAILED();
    return 0;
  }
  if (transA) { if (NT == 2) HSCN_ADJ_S(1, 2, false); else HSCN_ADJ_S(1, 1, false); }
  else if (rsum) { if (NT == 2) HSCN_ADJ_S(0, 2, true); else HSCN_ADJ_S(0, 1, true); }
  else { if (NT == 2) HSCN_ADJ_S(0, 2, false); else HSCN_ADJ_S(0, 1, false); }
#undef HSCN_ADJ_S
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

// row softmax, KP = pow2 >= K lanes per row (K <= 64): consecutive lanes on consecutive addresses
__global__ void __launch_bounds__(256)
k_softmax_rows_d(const float* __restrict__ logits, float* __restrict__ S, int64_t n, int K, int KP) {
  const int k = threadIdx.x % KP;
  const int64_t rpb = 256 / KP;
  for (int64_t i = (int64_t)blockIdx.x * rpb + threadIdx.x / KP; i < n; i += (int64_t)gridDim.x * rpb) {
    const float v = k < K ? logits[i * K + k] : -INFINITY;
    float m = v;
    for (int off = KP >> 1; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    const float ex = k < K ? expf(v - m) : 0.f;
    float sum = ex;
    for (int off = KP >> 1; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (k < K) S[i * K + k] = ex / sum;
  }
}

// The same for K % 4 == 0: a lane owns FOUR columns (one 16-byte load and store), a row sits in LPR = pow2 >= K / 4 <= 16
// lanes of one DPP row, max and sum cross lanes by DPP (no ds_bpermute round trips): 17.8 -> ~9 us on 57 k x 64.
__device__ __forceinline__ float seg16_max(float v, int LPR) {
  if (LPR >= 2) v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)));
  if (LPR >= 4) v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)));
  if (LPR >= 8) v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true)));
  if (LPR >= 16) v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true)));
  return v;
}
__device__ __forceinline__ float seg16_sum(float v, int LPR) {
  if (LPR >= 2) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
  if (LPR >= 4) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
  if (LPR >= 8) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));
  if (LPR >= 16) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));
  return v;
}
__global__ void __launch_bounds__(256)
k_softmax_rows4(const float* __restrict__ logits, float* __restrict__ S, int64_t n, int K, int LPR) {
  const int l = threadIdx.x % LPR;
  const int64_t rpb = 256 / LPR;
  const bool on = 4 * l < K;
  // (every lane of the wave stays in the loop while any row of its trip exists: the DPP exchanges need their partners)
  for (int64_t i0 = (int64_t)blockIdx.x * rpb; i0 < n; i0 += (int64_t)gridDim.x * rpb) {
    const int64_t i = i0 + threadIdx.x / LPR;
    const bool ok = on && i < n;
    float4 v = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    if (ok) v = *reinterpret_cast<const float4*>(logits + i * K + 4 * l);
    const float m = seg16_max(fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)), LPR);
    float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) e = make_float4(expf(v.x - m), expf(v.y - m), expf(v.z - m), expf(v.w - m));
    const float sum = seg16_sum((e.x + e.y) + (e.z + e.w), LPR);
    if (ok) *reinterpret_cast<float4*>(S + i * K + 4 * l) = make_float4(e.x / sum, e.y / sum, e.z / sum, e.w / sum);
  }
}

// deg[b][i] = sum_k adj[b][i][k]   (wave per row, ordered fold)
__global__ void k_rowsum(const float* __restrict__ adj, float* __restrict__ deg, int64_t rows, int n) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* p = adj + row * n;
  float s = 0.f;
  for (int k = lane; k < n; k += 64) s += p[k];
  s = wave_sum(s);
  if (lane == 0) deg[row] = s;
}

// sum over the 256 threads of a workgroup, waves folded in order (red: 4 floats of LDS); every thread
// receives the total.  Two barriers.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();   // red may still be read from the previous call
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return ((red[0] + red[1]) + red[2]) + red[3];
}

// the same for NT threads (NT / 64 wave partials folded in wave order)
template <int NT>
__device__ __forceinline__ float block_sum_nt(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();   // red may still be read from the previous call
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int w = 1; w < NT / 64; ++w) t += red[w];
  return t;
}

// per graph: num = tr(oa), den = sum_i deg_i |S_i|^2, |ss|_F, ortho; normalise oa in place.
// One workgroup per graph (NT = 1 024 threads: the pass over S -- 115 KB per PascalVOC-SP graph -- is one trip of
// requests per thread instead of four), all threads in every reduction; oa and ss are staged in LDS with rows
// padded by one word (the row sums walk a row per thread).
template <int NT>
__global__ void __launch_bounds__(NT)
k_dense_finalize(const float* __restrict__ S, const float* __restrict__ deg, const float* __restrict__ ss,
                 float* __restrict__ oa, float* __restrict__ stats, int n, int K, const int32_t* __restrict__ nptr) {
  extern __shared__ float lds[];
  const int KS = K + 1, KK = K * K;
  float* oal = lds;                  // [K][KS]
  float* ssl = oal + K * KS;         // [K][KS]
  float* dn = ssl + K * KS;          // [K]
  float* red = dn + K;               // [NT / 64]
  const int g = blockIdx.x;
  const size_t r0 = nptr ? (size_t)nptr[g] : (size_t)g * n;
  if (nptr) n = nptr[g + 1] - nptr[g];
  const float* Sg = S + r0 * K;
  const float* dg = deg + r0;
  const float* ssg = ss + (size_t)g * KK;
  float* oag = oa + (size_t)g * KK;
  for (int idx = threadIdx.x; idx < KK; idx += NT) {
    const int a = idx / K, b = idx - a * K;
    oal[a * KS + b] = oag[idx];
    ssl[a * KS + b] = ssg[idx];
  }
  // den = sum_i deg_i |S_i|^2 = sum over all elements of deg[row] * S^2 (strided, coalesced)
  float part = 0.f;
  {
    const int tot = n * K;
    if ((K & 3) == 0) {   // float4 pieces, eight requests in flight per trip
      const int tot4 = tot >> 2, kq = K >> 2;
      const float4* S4 = reinterpret_cast<const float4*>(Sg);
      for (int base = threadIdx.x; base < tot4; base += 8 * NT) {
        float4 v[8];
        float d[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i4 = base + u * NT;
          const bool ok = i4 < tot4;
          v[u] = S4[ok ? i4 : 0];
          d[u] = ok ? dg[i4 / kq] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          part = fmaf(d[u], (v[u].x * v[u].x + v[u].y * v[u].y) + (v[u].z * v[u].z + v[u].w * v[u].w), part);
      }
    } else {
      for (int idx = threadIdx.x; idx < tot; idx += NT) {
        const float v = Sg[idx];
        part = fmaf(dg[idx / K], v * v, part);
      }
    }
  }
  const float den = block_sum_nt<NT>(part, red);   // (its barriers also publish oal / ssl)
  float t = 0.f;
  for (int a = threadIdx.x; a < K; a += NT) t += oal[a * KS + a];
  const float num = block_sum_nt<NT>(t, red);
  t = 0.f;
  for (int idx = threadIdx.x; idx < KK; idx += NT) {
    const float v = ssl[(idx / K) * KS + idx % K];
    t = fmaf(v, v, t);
  }
  const float nrm = sqrtf(block_sum_nt<NT>(t, red));
  const float isk = 1.0f / sqrtf((float)K);
  t = 0.f;
  for (int idx = threadIdx.x; idx < KK; idx += NT) {
    const int a = idx / K, b = idx - a * K;
    const float q = ssl[a * KS + b] / nrm - (a == b ? isk : 0.f);
    t = fmaf(q, q, t);
  }
  const float o2 = block_sum_nt<NT>(t, red);
  if (threadIdx.x == 0) {
    stats[g * 4 + 0] = num;
    stats[g * 4 + 1] = den;
    stats[g * 4 + 2] = nrm;
    stats[g * 4 + 3] = sqrtf(o2);
  }
  for (int a = threadIdx.x; a < K; a += NT) {
    float s_ = 0.f;
    for (int b = 0; b < K; ++b) s_ += (a == b) ? 0.f : oal[a * KS + b];
    dn[a] = sqrtf(s_) + 1e-15f;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < KK; idx += NT) {
    const int a = idx / K, b = idx - a * K;
    oag[idx] = (a == b) ? 0.f : (oal[a * KS + b] / dn[b]) / dn[a];
  }
}

__global__ void k_losses_d(const float* __restrict__ stats, float* __restrict__ losses, int G) {
  float mc = 0.f, o = 0.f;
  for (int g = threadIdx.x; g < G; g += 64) {
    mc += -(stats[g * 4 + 0] / stats[g * 4 + 1]);
    o += stats[g * 4 + 3];
  }
  mc = wave_sum(mc);
  o = wave_sum(o);
  if (threadIdx.x == 0) {
    losses[0] = mc / (float)G;
    losses[1] = o / (float)G;
  }
}

// Gss[g] = d ortho_g / d (S^T S)  (K x K, symmetric), scaled by 2 * dL/dortho / G
__global__ void __launch_bounds__(256)
k_dense_gss(const float* __restrict__ stats, const float* __restrict__ ss, const float* __restrict__ g_losses,
            float* __restrict__ Gss, int K, int G) {
  __shared__ float red[4];
  const int g = blockIdx.x;
  const int KK = K * K;
  const float nrm = stats[g * 4 + 2], o = stats[g * 4 + 3];
  const float go2 = 2.f * g_losses[1] / (float)G;
  const float isk = 1.0f / sqrtf((float)K);
  const float* ssg = ss + (size_t)g * KK;
  // every thread keeps its (up to 16) entries in registers: one pass over ss
  constexpr int EPT = 16;   // K <= 64: K*K / 256
  float e[EPT];
  float v = 0.f;
#pragma unroll
  for (int u = 0; u < EPT; ++u) {
    const int i = threadIdx.x + u * 256;
    e[u] = i < KK ? ssg[i] : 0.f;
  }
  if (o > 0.f) {
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
      const int i = threadIdx.x + u * 256;
      if (i < KK) {
        const int a = i / K, b = i - a * K;
        v += ((e[u] / nrm - (a == b ? isk : 0.f)) / o) * e[u];
      }
    }
  }
  const float inner = block_sum_256(v, red);
#pragma unroll
  for (int u = 0; u < EPT; ++u) {
    const int i = threadIdx.x + u * 256;
    if (i < KK) {
      const int a = i / K, b = i - a * K;
      const float gq = o > 0.f ? (e[u] / nrm - (a == b ? isk : 0.f)) / o : 0.f;
      Gss[(size_t)g * KK + i] = go2 * ((gq - inner / (nrm * nrm) * e[u]) / nrm);
    }
  }
}

// g_logits[i,:] = S_i * (dS_i - <dS_i, S_i>),  dS = c_num (AS + A^T S) + c_den 2 deg S + S Gss'
// (SG = S Gss' comes from the batched GEMM); K/4 lanes per row, row dot by __shfl_xor
__global__ void __launch_bounds__(256)
k_dense_bwd(const float* __restrict__ S, const float* __restrict__ AS, const float* __restrict__ AtS,
            const float* __restrict__ SG, const float* __restrict__ deg, const float* __restrict__ stats,
            const float* __restrict__ g_losses, float* __restrict__ g_logits, int n, int K, int G, int LPRp,
            const int32_t* __restrict__ gid, int64_t rows_total, const int32_t* __restrict__ asym = nullptr) {
  const int64_t rows = gid ? rows_total : (int64_t)G * n;
  const int RPB = 256 / LPRp;
  const int rl = threadIdx.x / LPRp, fl = threadIdx.x % LPRp;
  for (int64_t r = (int64_t)blockIdx.x * RPB + rl; r < rows; r += (int64_t)gridDim.x * RPB) {
    const int g = gid ? gid[r] : (int)(r / n);
    const float num = stats[g * 4 + 0], den = stats[g * 4 + 1];
    const float gmc = g_losses[0] / (float)G;
    const float c_num = -gmc / den, c_den = gmc * num / (den * den);
    const float di = deg[r];
    const size_t base = (size_t)r * K;
    float dS[4] = {0.f, 0.f, 0.f, 0.f}, sv[4] = {0.f, 0.f, 0.f, 0.f};
    float dot = 0.f;
    // a symmetric graph's A^T S was not computed (k_adj_s skipped it): it IS the forward's A S
    const float* __restrict__ AtSr = (asym && asym[g] == 0) ? AS : AtS;
    if ((K & 3) == 0) {      // whole 16-byte pieces: four loads and one store per lane
      const int k0 = fl * 4;
      if (k0 < K) {
        const float4 s4 = *reinterpret_cast<const float4*>(S + base + k0), a4 = *reinterpret_cast<const float4*>(AS + base + k0);
        const float4 t4 = *reinterpret_cast<const float4*>(AtSr + base + k0), g4 = *reinterpret_cast<const float4*>(SG + base + k0);
        sv[0] = s4.x; sv[1] = s4.y; sv[2] = s4.z; sv[3] = s4.w;
        dS[0] = c_num * (a4.x + t4.x) + c_den * 2.f * di * sv[0] + g4.x;
        dS[1] = c_num * (a4.y + t4.y) + c_den * 2.f * di * sv[1] + g4.y;
        dS[2] = c_num * (a4.z + t4.z) + c_den * 2.f * di * sv[2] + g4.z;
        dS[3] = c_num * (a4.w + t4.w) + c_den * 2.f * di * sv[3] + g4.w;
#pragma unroll
        for (int q = 0; q < 4; ++q) dot = fmaf(dS[q], sv[q], dot);
      }
      for (int off = LPRp >> 1; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
      if (k0 < K)
        *reinterpret_cast<float4*>(g_logits + base + k0) =
            make_float4(sv[0] * (dS[0] - dot), sv[1] * (dS[1] - dot), sv[2] * (dS[2] - dot), sv[3] * (dS[3] - dot));
      continue;
    }
    for (int k0 = fl * 4; k0 < K; k0 += LPRp * 4) {   // one pass when K <= 4*LPRp (always: LPRp = pow2ceil(K/4))
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = k0 + q;
        if (k < K) {
          sv[q] = S[base + k];
          dS[q] = c_num * (AS[base + k] + AtSr[base + k]) + c_den * 2.f * di * sv[q] + SG[base + k];
          dot = fmaf(dS[q], sv[q], dot);
        }
      }
    }
    for (int off = LPRp >> 1; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
    const int k0 = fl * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (k0 + q < K) g_logits[base + k0 + q] = sv[q] * (dS[q] - dot);
  }
}

// nr products op(A) * B_z (z < nr <= 3) in one launch; NT covers the widest right-hand side
int bgemm_multi(const float* A, const BRhs3& R, int nr, int64_t batch, int M, int Kd, int64_t lda, int64_t sA,
                int transA, hipStream_t st, float* rsum = nullptr, Rag G = Rag{nullptr, 0, 0, 0}) {
  int nmax = 0;
  for (int z = 0; z < nr; ++z) {
    if (R.r[z].N < 1 || R.r[z].N > GNMAX) return HSCN_E_UNSUPPORTED;
    if (R.r[z].N > nmax) nmax = R.r[z].N;
  }
  const int NT = (nmax + 15) / 16;
  dim3 grid((M + GM - 1) / GM, (unsigned)batch, (unsigned)nr);
  if (rsum && !transA && nr == 1) {   // A . B and the row sums of A in one pass
#define HSCN_BGEMM_RS(NT_) k_bgemm<0, NT_, 1><<<grid, 256, 0, st>>>(A, R, M, Kd, lda, sA, rsum, G)
    switch (NT) {
      case 1: HSCN_BGEMM_RS(1); break;
      case 2: HSCN_BGEMM_RS(2); break;
      case 3: HSCN_BGEMM_RS(3); break;
      default: HSCN_BGEMM_RS(4); break;
    }
#undef HSCN_BGEMM_RS
    HSCN_RETURN_IF_LAUNCH_FAILED();
    return 0;
  }
#define HSCN_BGEMM(TA_, NT_) k_bgemm<TA_, NT_, 0><<<grid, 256, 0, st>>>(A, R, M, Kd, lda, sA, nullptr, G)
  if (transA) {
    switch (NT) {
      case 1: HSCN_BGEMM(1, 1); break;
      case 2: HSCN_BGEMM(1, 2); break;
      case 3: HSCN_BGEMM(1, 3); break;
      default: HSCN_BGEMM(1, 4); break;
    }
  } else {
    switch (NT) {
      case 1: HSCN_BGEMM(0, 1); break;
      case 2: HSCN_BGEMM(0, 2); break;
      case 3: HSCN_BGEMM(0, 3); break;
      default: HSCN_BGEMM(0, 4); break;
    }
  }
#undef HSCN_BGEMM
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

int bgemm(const float* A, const float* Bm, float* C, int64_t batch, int M, int N, int Kd, int64_t lda, int64_t ldb,
          int64_t ldc, int64_t sA, int64_t sB, int64_t sC, int transA, hipStream_t st, float* rsum = nullptr,
          Rag G = Rag{nullptr, 0, 0, 0}, int ragB = 0, int ragC = 0) {
  BRhs3 R{};
  R.r[0] = BRhs{Bm, C, N, ldb, ldc, sB, sC, ragB, ragC};
  return bgemm_multi(A, R, 1, batch, M, Kd, lda, sA, transA, st, rsum, G);
}

// forward / backward of dense_mincut_pool for a batch of B graphs: n = the common node count (nptr == NULL: operands
// [B, n, .]) or the LARGEST one (ragged: node-indexed operands flat [N, .], adjacency [B, n, n] zero beyond n_b)
// adj8 != 0: `adj` points at uint8 counts, rows padded to lda8 = round_up(n, 32) (hscn_to_dense_adj_ragged_u8)
int mincut_dense_fwd_impl(const float* x, const float* adj, const float* logits, const int32_t* nptr, int64_t N,
                          int64_t B, int n, int K, int F, float* S, float* AS, float* deg, float* stats, float* ss,
                          float* pooled_x, float* pooled_adj, float* losses, hipStream_t st, int adj8 = 0) {
  const int64_t rows = nptr ? N : B * n;
  if ((K & 3) == 0 && K <= 64) {     // four columns per lane, DPP reductions
    int LPR = 1;
    while (LPR * 4 < K) LPR <<= 1;
    unsigned nb = hscn_blocks(rows, 256 / LPR);
    if (nb > 8192) nb = 8192;
    k_softmax_rows4<<<nb, 256, 0, st>>>(logits, S, rows, K, LPR);
  } else {
    int KP = 1;
    while (KP < K) KP <<= 1;
    unsigned nb = hscn_blocks(rows, 256 / KP);
    if (nb > 16384) nb = 16384;
    k_softmax_rows_d<<<nb, 256, 0, st>>>(logits, S, rows, K, KP);
  }
  HSCN_RETURN_IF_LAUNCH_FAILED();
  int rc;
  const int rg = nptr ? 1 : 0;
  // A S, and deg = A . 1 from the same pass over the adjacency (0/1 entries: the sums are exact in any order)
  static const bool old_as = getenv("HSCN_DENSE_AS") && atoi(getenv("HSCN_DENSE_AS")) == 16;   // A/B: the 16x16x4 kernel
  if (adj8) {
    if ((rc = launch_adj_s<uint8_t>(reinterpret_cast<const uint8_t*>(adj), S, AS, deg, nptr, B, n, n, (n + 31) & ~31, K, 0, st))) return rc;
  } else if (!old_as) {
    if ((rc = launch_adj_s<float>(adj, S, AS, deg, nptr, B, n, n, n, K, 0, st))) return rc;
  } else if ((rc = bgemm(adj, S, AS, B, n, K, n, n, K, K, (int64_t)n * n, (int64_t)n * K, (int64_t)n * K, 0, st, deg,
                         Rag{nptr, 0, 1, 1}, rg, rg)))
    return rc;
  // S^T (A S)  -> pooled_adj (raw), S^T S -> ss, S^T X -> pooled_x
  {
    BRhs3 R{};
    R.r[0] = BRhs{AS, pooled_adj, K, K, K, (int64_t)n * K, (int64_t)K * K, rg, 0};
    R.r[1] = BRhs{S, ss, K, K, K, (int64_t)n * K, (int64_t)K * K, rg, 0};
    int nr = 2;
    if (x && pooled_x && F > 0) R.r[nr++] = BRhs{x, pooled_x, F, F, F, (int64_t)n * F, (int64_t)K * F, rg, 0};
    if ((rc = bgemm_multi(S, R, nr, B, K, n, K, (int64_t)n * K, 1, st, nullptr, Rag{nptr, 1, 0, 1}))) return rc;
  }
  k_dense_finalize<1024><<<(unsigned)B, 1024, (size_t)(2 * K * (K + 1) + K + 16) * 4, st>>>(S, deg, ss, pooled_adj, stats, n, K, nptr);
  k_losses_d<<<1, 64, 0, st>>>(stats, losses, (int)B);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

int mincut_dense_bwd_impl(const float* adj, const float* S, const float* AS, const float* deg, const float* stats,
                          const float* ss, const float* g_losses, const int32_t* nptr, const int32_t* gid, int64_t N,
                          int64_t B, int n, int K, float* AtS, float* sg_ws, float* gss_ws, float* g_logits,
                          hipStream_t st, int adj8 = 0, const int32_t* asym = nullptr) {
  int rc;
  const int rg = nptr ? 1 : 0;
  static const bool old_as = getenv("HSCN_DENSE_AS") && atoi(getenv("HSCN_DENSE_AS")) == 16;
  if (adj8) {
    if (!(nptr && (K & 3) == 0)) asym = nullptr;      // (the skipping form exists for the ragged byte route only)
    if ((rc = launch_adj_s<uint8_t>(reinterpret_cast<const uint8_t*>(adj), S, AtS, nullptr, nptr, B, n, n, (n + 31) & ~31, K, 1, st, asym))) return rc;
  } else if (!old_as) {
    asym = nullptr;
    if ((rc = launch_adj_s<float>(adj, S, AtS, nullptr, nptr, B, n, n, n, K, 1, st))) return rc;
  } else if ((rc = bgemm(adj, S, AtS, B, n, K, n, n, K, K, (int64_t)n * n, (int64_t)n * K, (int64_t)n * K, 1, st, nullptr,
                         Rag{nptr, 0, 1, 1}, rg, rg)))
    return rc;
  // Gss' (scaled) -> gss_ws [B,K,K];  SG = S Gss' -> sg_ws
  k_dense_gss<<<(unsigned)B, 256, 0, st>>>(stats, ss, g_losses, gss_ws, K, (int)B);
  if ((rc = bgemm(S, gss_ws, sg_ws, B, n, K, K, K, K, K, (int64_t)n * K, (int64_t)K * K, (int64_t)n * K, 0, st, nullptr,
                  Rag{nptr, 1, 1, 0}, 0, rg)))
    return rc;
  int LPRp = 1;
  while (LPRp * 4 < K) LPRp <<= 1;
  const int64_t rows = nptr ? N : B * n;
  int64_t nb = (rows + 256 / LPRp - 1) / (256 / LPRp);
  if (nb > 8192) nb = 8192;
  k_dense_bwd<<<(unsigned)nb, 256, 0, st>>>(S, AS, AtS, sg_ws, deg, stats, g_losses, g_logits, n, K, (int)B, LPRp,
                                            nptr ? gid : nullptr, rows, adj8 ? asym : nullptr);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

}  // namespace

extern "C" {

int hscn_bgemm_f32(const float* A, const float* B, float* C, int64_t batch, int M, int N, int Kd, int64_t lda,
                   int64_t ldb, int64_t ldc, int64_t strideA, int64_t strideB, int64_t strideC, int transA,
                   void* stream_) {
  if (batch < 0 || M < 0 || N < 1 || Kd < 0) return HSCN_E_BADARG;
  if (batch == 0 || M == 0) return 0;
  if (!A || !B || !C) return HSCN_E_BADARG;
  return bgemm(A, B, C, batch, M, N, Kd, lda, ldb, ldc, strideA, strideB, strideC, transA, hscn_stream(stream_));
}

int hscn_mincut_dense_fwd(const float* x, const float* adj, const float* logits, int64_t B, int n, int K, int F,
                          float* S, float* AS, float* deg, float* stats, float* ss, float* pooled_x,
                          float* pooled_adj, float* losses, void* stream_) {
  if (B < 1 || n < 1 || K < 1 || K > GNMAX || F < 0 || F > GNMAX) return HSCN_E_BADARG;
  if (!adj || !logits || !S || !AS || !deg || !stats || !ss || !pooled_adj || !losses) return HSCN_E_BADARG;
  return mincut_dense_fwd_impl(x, adj, logits, nullptr, 0, B, n, K, F, S, AS, deg, stats, ss, pooled_x, pooled_adj,
                               losses, hscn_stream(stream_));
}

int hscn_mincut_dense_bwd(const float* adj, const float* S, const float* AS, const float* deg, const float* stats,
                          const float* ss, const float* g_losses, int64_t B, int n, int K, float* AtS,
                          float* sg_ws, float* gss_ws, float* g_logits, void* stream_) {
  if (B < 1 || n < 1 || K < 1 || K > GNMAX) return HSCN_E_BADARG;
  if (!adj || !S || !AS || !deg || !stats || !ss || !g_losses || !AtS || !sg_ws || !gss_ws || !g_logits)
    return HSCN_E_BADARG;
  return mincut_dense_bwd_impl(adj, S, AS, deg, stats, ss, g_losses, nullptr, nullptr, 0, B, n, K, AtS, sg_ws, gss_ws,
                               g_logits, hscn_stream(stream_));
}

int hscn_mincut_dense_ragged_fwd(const float* x, const void* adj, int adj_elem_bytes, const float* logits,
                                 const int32_t* nptr, int64_t N, int64_t B, int nmax, int K, int F, float* S, float* AS,
                                 float* deg, float* stats, float* ss, float* pooled_x, float* pooled_adj, float* losses,
                                 void* stream_) {
  if (B < 1 || N < 1 || nmax < 1 || K < 1 || K > GNMAX || F < 0 || F > GNMAX) return HSCN_E_BADARG;
  if (adj_elem_bytes != 4 && adj_elem_bytes != 1) return HSCN_E_BADARG;
  if (!adj || !logits || !nptr || !S || !AS || !deg || !stats || !ss || !pooled_adj || !losses) return HSCN_E_BADARG;
  return mincut_dense_fwd_impl(x, static_cast<const float*>(adj), logits, nptr, N, B, nmax, K, F, S, AS, deg, stats, ss,
                               pooled_x, pooled_adj, losses, hscn_stream(stream_), adj_elem_bytes == 1);
}

int hscn_mincut_dense_ragged_bwd(const void* adj, int adj_elem_bytes, const float* S, const float* AS, const float* deg,
                                 const float* stats, const float* ss, const float* g_losses, const int32_t* nptr,
                                 const int32_t* gid, int64_t N, int64_t B, int nmax, int K, float* AtS, float* sg_ws,
                                 float* gss_ws, float* g_logits, void* stream_) {
  if (B < 1 || N < 1 || nmax < 1 || K < 1 || K > GNMAX) return HSCN_E_BADARG;
  if (adj_elem_bytes != 4 && adj_elem_bytes != 1) return HSCN_E_BADARG;
  if (!adj || !S || !AS || !deg || !stats || !ss || !g_losses || !nptr || !gid || !AtS || !sg_ws || !gss_ws || !g_logits)
    return HSCN_E_BADARG;
  return mincut_dense_bwd_impl(static_cast<const float*>(adj), S, AS, deg, stats, ss, g_losses, nptr, gid, N, B, nmax, K,
                               AtS, sg_ws, gss_ws, g_logits, hscn_stream(stream_), adj_elem_bytes == 1);
}

// asym[b] |= 1 when the byte adjacency of graph b is not symmetric.  A workgroup compares one 64 x 64 tile (I, J), I <= J,
// with the transpose of tile (J, I): both are read as rows (coalesced), the second one parked in LDS and read back
// transposed.  Entries beyond a graph's nodes are zero on both sides.
static __global__ void __launch_bounds__(256)
k_adj_asym_u8(const uint8_t* __restrict__ adj, int nmax, int64_t lda, int32_t* __restrict__ asym) {
  __shared__ uint8_t tl[64][80];
  const int T = (nmax + 63) / 64;
  // blockIdx.x enumerates the pairs I <= J
  int I = 0, rem = blockIdx.x;
  while (rem >= T - I) { rem -= T - I; ++I; }
  const int J = I + rem;
  const int b = blockIdx.y;
  const uint8_t* Ab = adj + (size_t)b * nmax * lda;
  const int r = threadIdx.x >> 2, c16 = (threadIdx.x & 3) * 16;
  uint4 a = make_uint4(0u, 0u, 0u, 0u), t = a;
  const int ra = I * 64 + r, ca = J * 64 + c16;      // tile (I, J): row ra, columns ca .. ca + 15
  const int rb = J * 64 + r, cb = I * 64 + c16;      // tile (J, I)
  if (ra < nmax && ca + 16 <= lda) a = *reinterpret_cast<const uint4*>(Ab + (size_t)ra * lda + ca);
  if (rb < nmax && cb + 16 <= lda) t = *reinterpret_cast<const uint4*>(Ab + (size_t)rb * lda + cb);
  *reinterpret_cast<uint4*>(&tl[r][c16]) = t;
  __syncthreads();
  const unsigned aw[4] = {a.x, a.y, a.z, a.w};
  bool bad = false;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const unsigned mine = (aw[q >> 2] >> (8 * (q & 3))) & 0xffu;      // A[ra][ca + q]
    const int col = ca + q;
    // its mirror A[col][ra] sits in tile (J, I) at row col - 64 J, column ra - 64 I = r
    const unsigned other = tl[c16 + q][r];
    if (col < nmax && ra < nmax && mine != other) bad = true;
  }
  if (bad) atomicOr(&asym[b], 1);
}

// The adjacency product of dense_mincut_pool alone: out = op(A) S for every graph of a ragged batch (op = transpose for the
// backward's A^T S), deg (optional, transA = 0) = the row sums of A.  What hscn_mincut_dense_ragged_fwd / _bwd launch
// for it, as an entry point of its own so that the route's dominant kernel can be timed and profiled by itself.
int hscn_dense_adj_asymmetry_u8(const void* adj8, int64_t B, int nmax, int32_t* asym, void* stream_) {
  if (!adj8 || !asym || B < 1 || nmax < 1) return HSCN_E_BADARG;
  const int T = (nmax + 63) / 64;
  k_adj_asym_u8<<<dim3((unsigned)(T * (T + 1) / 2), (unsigned)B), 256, 0, hscn_stream(stream_)>>>(
      static_cast<const uint8_t*>(adj8), nmax, (int64_t)((nmax + 31) & ~31), asym);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

int hscn_mincut_dense_ragged_bwd_sym(const void* adj, int adj_elem_bytes, const float* S, const float* AS, const float* deg,
                                     const float* stats, const float* ss, const float* g_losses, const int32_t* nptr,
                                     const int32_t* gid, int64_t N, int64_t B, int nmax, int K, float* AtS, float* sg_ws,
                                     float* gss_ws, float* g_logits, const int32_t* asym, void* stream_) {
  if (B < 1 || N < 1 || nmax < 1 || K < 1 || K > GNMAX) return HSCN_E_BADARG;
  if (adj_elem_bytes != 4 && adj_elem_bytes != 1) return HSCN_E_BADARG;
  if (!adj || !S || !AS || !deg || !stats || !ss || !g_losses || !nptr || !gid || !AtS || !sg_ws || !gss_ws || !g_logits)
    return HSCN_E_BADARG;
  return mincut_dense_bwd_impl(static_cast<const float*>(adj), S, AS, deg, stats, ss, g_losses, nptr, gid, N, B, nmax, K,
                               AtS, sg_ws, gss_ws, g_logits, hscn_stream(stream_), adj_elem_bytes == 1,
                               adj_elem_bytes == 1 ? asym : nullptr);
}

int hscn_dense_adj_s(const void* adj, int adj_elem_bytes, const float* S, const int32_t* nptr, int64_t B, int nmax, int K,
                     int transA, float* out, float* deg, void* stream_) {
  if (B < 1 || nmax < 1 || K < 1 || K > 64 || !adj || !S || !nptr || !out) return HSCN_E_BADARG;
  if (adj_elem_bytes != 4 && adj_elem_bytes != 1) return HSCN_E_BADARG;
  hipStream_t st = hscn_stream(stream_);
  if (adj_elem_bytes == 1)
    return launch_adj_s<uint8_t>(static_cast<const uint8_t*>(adj), S, out, transA ? nullptr : deg, nptr, B, nmax, nmax,
                                 (nmax + 31) & ~31, K, transA, st);
  return launch_adj_s<float>(static_cast<const float*>(adj), S, out, transA ? nullptr : deg, nptr, B, nmax, nmax, nmax, K,
                             transA, st);
}

}  // extern "C"
