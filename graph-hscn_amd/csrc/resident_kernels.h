// Graph-resident HSCN engine: a workgroup keeps one graph's features, CSRs and weights in LDS for
// every layer.
//
// A batch of LRGB graphs is block-diagonal: graph g owns local nodes
// [lptr[g], lptr[g+1]), virtual nodes [vptr[g], vptr[g+1]) and a contiguous slice
// of each relation's edge list.  A Peptides graph (n <= 444, e <= ~1000, H = 16)
// fits in a fraction of one CU's 160 KB LDS, so the whole HSCN forward
// (reference model/hscn.py:102-114: L x HeteroConv{ll GCN, vv GCN, lv GAT} + ReLU,
// mean pool, 2-layer head) runs with workgroup barriers only.
//
// Launch shapes (hscn_fwd_body / hscn_bwd_body are the per-workgroup programs):
//   k_hscn_fwd         one workgroup per graph, both branches (inference, keep_virtual, large batches);
//   k_hscn_fwd_pair    2 workgroups per graph: even = local chain + head, odd = the part of the virtual
//                      branch that does not need it (its CSRs + layer 0), state left in HBM;
//   k_hscn_bwd_virtual 2 workgroups per graph: even = backward, odd = virtual layers 1.. resumed from that
//                      state and the stored local activations;
//   k_hscn_bwd         one workgroup per graph.
// The virtual branch cannot influence the prediction in the reference architecture ("local" only
// receives ll); it is computed for fidelity, on CUs a 128-graph batch would leave idle.
//
// The kernels are latency-bound (a few MB for the whole batch), so they are
// organised around the number of dependent steps, not around bandwidth:
//   prologue : every global input of the graph is requested before anything is consumed (edges first:
//              they are consumed first; features and weights are parked under the CSR builds);
//   structure: COO slices -> stable CSR in LDS, the independent CSRs side by side on wave groups between
//              the same barriers (LDS int atomics + rank by edge number; lv, whose rows are whole
//              clusters, by a wave-ballot multisplit); scratch lives inside the not-yet-used feature
//              buffer; the source-keyed ll CSR is exported for the backward launch;
//   layers   : two barriers per layer.  Local chain: X W^T on v_mfma_f32_16x16x4_f32, then the CSR
//              gather-reduce (+bias, ReLU, activations to HBM).  Virtual branch: register-blocked
//              transforms, lv segment softmax per 64-member chunk (last chunk wave of a cluster folds the
//              partials and adds the vv row).  Next layer's weights are fetched at the top of the layer
//              and parked in the other LDS weight buffer under this layer's math;
//   epilogue : mean pool over all waves, head by wave 0.
//
// The backward mirrors it (exported transposed CSR, MFMA weight / input gradients, conflict-free bias
// sums, per-graph parameter-gradient partials, then one ordered reduction over graphs): no float
// atomics, bitwise reproducible; three n x H buffers, or two when LDS is short.
//
// Numerics: k-ascending fmaf chains in the transforms, edge-order separately rounded
// multiply/add in the gather-reduce (same as the layered kernels).
#pragma once
#include "hscn_common.h"
#include "resident_common.h"
#include <cstdio>
#include <cstdlib>

namespace {

constexpr int RT_MAX = 1024;  // largest workgroup the kernels are instantiated for
constexpr int MAXL = 8;

struct LayerP {
  const float *W_ll, *b_ll, *W_vv, *b_vv, *W_src, *W_dst, *att_src, *att_dst, *b_gat;
};

// capacities of the one-launch step's virtual program in its fixed-layout form (hscn_fwd_body MODE 6): Peptides-sized
// graphs, up to 32 clusters
struct StepVCaps {
  static constexpr int N = 448, V = 32, EVV = V * (V + 1) / 2;
};

struct FwdArgs {
  const float* x_local;
  const float* x_virtual;
  const int64_t *ll_src, *ll_dst, *vv_src, *vv_dst, *lv_src, *lv_dst;
  const int32_t *lptr, *vptr, *eptr_ll, *eptr_vv, *eptr_lv;
  LayerP layer[MAXL];
  const float *W1, *b1, *W2, *b2;
  float *acts, *pooled, *z, *pred, *xv_out;
  float* score;  // optional [B][C]: sigmoid(pred), the score output of the loss tail (loss.py:9-10,17-19)
  int32_t *csr_rowptr_t, *csr_col_t;  // exported source-keyed ll CSR (graph g: rowptr at n0+g, col at e0)
  float* dinv_out;                     // exported in-degree^-1/2 of the ll relation
  int32_t* flag;
  int64_t N, V;
  int F, L, C, head_act, max_n, max_v, max_ell, max_evv, compute_virtual;
  // virtual-only launches may run a layer range [l_begin, l_end): a first part (0..l_end) exports the
  // virtual relations' CSRs and the virtual features to vs_*, a resumed part (l_begin > 0) loads them
  int l_begin, l_end;
  int32_t *vs_rowptr_lv, *vs_col_lv, *vs_rowptr_vv, *vs_col_vv;
  float *vs_dinv_v, *vs_xv;
  int spec;  // 1: ll path and virtual branch run concurrently on two wave groups (needs a 3rd n x H buffer)
  int exp;   // 1: this launch also builds + exports the source-keyed ll CSR (needs LDS for it)
  // one-launch step (resident_step.h): the local activations this virtual-only workgroup reads are published by the
  // local workgroup of the same launch; ready[g] counts them (epoch * 8 + count), NULL: they come from an earlier launch
  const uint32_t* ready;
  const uint32_t* epoch;
  int acq;   // 1: the consumer acquires (agent scope) and uses plain loads -- several workgroups share a CU (small
             // graphs), outside the envelope the acquire-free sc1-load form was measured for; 0: sc1 loads, no acquire
  // structure_build = "dataset-resident": the CSRs of the virtual relations and the virtual degree norm of every
  // graph come from HBM (include/hscn.h: hscn_structure; built once per dataset, gathered with the batch) instead
  // of being rebuilt from the COO slices every step.  Virtual-only workgroups from layer 0 (the one-launch step).
  const int32_t *pre_rp_lv, *pre_col_lv, *pre_rp_vv, *pre_col_vv;
  const float* pre_dinv_v;
  int db;        // 1: two weight buffers in LDS (the next layer's weights land under this layer's math)
  int exp_dinv;  // 1: this launch exports the ll degree norm (the workgroup that builds ll keyed by target has it)
  float slope;
};

struct BwdArgs {
  const float* x_local;
  const int64_t *ll_src, *ll_dst;
  const int32_t *lptr, *eptr_ll;
  const float* W_ll[MAXL];
  const float *W1, *W2;
  const float *acts, *pooled, *z, *g_pred;
  const float* g_scale;  // optional device scalar: the upstream gradient is g_scale[0] * g_pred
  // loss tail riding on this launch (target != NULL): the upstream gradient row is computed here from
  // (pred, target) instead of being read from g_pred, and the graph's summed loss terms go to column
  // `Pn` of its partials row (partials rows are then P = Pn + 1 wide)
  const float *pred, *target;
  int loss_kind, Pn;
  float inv_count;
  const int32_t *csr_rowptr_t, *csr_col_t;  // from the forward launch
  const float* dinv_in;
  float* partials;  // [B][P]
  int32_t* flag;
  int64_t N;
  int F, L, C, head_act, max_n, max_ell, P;
  int two;  // 1: two n x H buffers instead of three (one more barrier per layer; for graphs that need the LDS)
};

// a layer's parameter pointers as a plain object, whatever address space the argument block is read through (the
// by-value kernel argument, or the constant-address-space view of it that late_args() hands out)
template <typename AT>
__device__ __forceinline__ LayerP layer_of(const AT& A, int l) {
  LayerP P;
  __builtin_memcpy(&P, &A.layer[l], sizeof(LayerP));
  return P;
}

// ---- layer weights: global -> registers (prefetch) -> LDS (transposed Wt[k][o], rows k>=fin zero) ----
// The 4 matrices of a layer ([H][fin] each, nn.Linear layout: W_ll, W_src, W_dst, W_vv) and 5
// H-vectors (b_ll, b_vv, b_gat, att_src, att_dst) form one flat index space so every thread holds
// WPT = ceil((4*H*H + 5*H)/RT) prefetched values.
template <int H, int RT>
struct WStage {
  static constexpr int TOTAL = 4 * H * H + 5 * H;
  static constexpr int MPT = (H * H + RT - 1) / RT;  // words of each matrix per thread
  float m[4][MPT];
  float v[5];
  // One uniform base pointer per matrix / vector (no per-lane pointer table lookups), clamped
  // addresses + select instead of branches: every request of the prefetch is issued back to back.
  // ll: the local->local matrix / bias are wanted; cv: the virtual branch's are.  What a launch does
  // not use is not requested (uniform branches: no request reaches the memory pipe).
  __device__ __forceinline__ void fetch(const LayerP& P, bool ll, bool cv, int fin) {
    const float* mats[4] = {P.W_ll, P.W_src, P.W_dst, P.W_vv};
    const float* vecs[5] = {P.b_ll, P.b_vv, P.b_gat, P.att_src, P.att_dst};
#pragma unroll
    for (int mm = 0; mm < 4; ++mm) {
#pragma unroll
      for (int i = 0; i < MPT; ++i) m[mm][i] = 0.f;
      if (mm == 0 ? ll : cv) {
#pragma unroll
        for (int i = 0; i < MPT; ++i) {
          const int d = threadIdx.x + i * RT;       // destination slot k*H + o (transposed)
          const int k = d / H, o = d - k * H;
          const bool ok = d < H * H && k < fin;
          const float t = mats[mm][ok ? o * fin + k : 0];
          m[mm][i] = ok ? t : 0.f;
        }
      }
    }
    const int t_ = threadIdx.x < H ? threadIdx.x : 0;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      v[q] = 0.f;
      if (q == 0 ? ll : cv) {
        const float t = vecs[q][t_];
        v[q] = threadIdx.x < H ? t : 0.f;
      }
    }
  }
  __device__ __forceinline__ void store(float* dst) const {
#pragma unroll
    for (int mm = 0; mm < 4; ++mm) {
#pragma unroll
      for (int i = 0; i < MPT; ++i) {
        const int d = threadIdx.x + i * RT;
        if (d < H * H) dst[mm * H * H + d] = m[mm][i];
      }
    }
    if (threadIdx.x < H) {
#pragma unroll
      for (int q = 0; q < 5; ++q) dst[4 * H * H + q * H + threadIdx.x] = v[q];
    }
  }
};

// ---- Y[n][H] = X[n][H(zero padded)] * W^T, W transposed in LDS; a lane owns OPT outputs of a row ----
template <int H, int OPT>
__device__ void lin_blk(const float* X, const float* Wt, float* Y, int n, const float* att, float* a_out,
                        const Grp& G) {
  constexpr int LPR = H / OPT;
  const int RS = G.nt / LPR;
  const int og = G.t % LPR, r0 = G.t / LPR;
  const int o0 = og * OPT;
  if (r0 >= n) return;
  float w[OPT][H];
#pragma unroll
  for (int k = 0; k < H; ++k) {
#pragma unroll
    for (int q = 0; q < OPT; ++q) w[q][k] = Wt[k * H + o0 + q];
  }
  float at[OPT];
#pragma unroll
  for (int q = 0; q < OPT; ++q) at[q] = att ? att[o0 + q] : 0.f;
  for (int i = r0; i < n; i += RS) {
    const float4* xr = reinterpret_cast<const float4*>(X + i * H);
    float acc[OPT];
#pragma unroll
    for (int q = 0; q < OPT; ++q) acc[q] = 0.f;
#pragma unroll
    for (int k4 = 0; k4 < H / 4; ++k4) {
      const float4 x = xr[k4];
#pragma unroll
      for (int q = 0; q < OPT; ++q) {
        acc[q] = fmaf(x.x, w[q][4 * k4 + 0], acc[q]);
        acc[q] = fmaf(x.y, w[q][4 * k4 + 1], acc[q]);
        acc[q] = fmaf(x.z, w[q][4 * k4 + 2], acc[q]);
        acc[q] = fmaf(x.w, w[q][4 * k4 + 3], acc[q]);
      }
    }
    if (Y) {
#pragma unroll
      for (int q = 0; q < OPT; ++q) Y[i * H + o0 + q] = acc[q];
    }
    if (att) {
      float d = 0.f;
#pragma unroll
      for (int q = 0; q < OPT; ++q) d = fmaf(acc[q], at[q], d);
#pragma unroll
      for (int off = LPR >> 1; off > 0; off >>= 1) d += __shfl_xor(d, off, 64);
      if (og == 0) a_out[i] = d;
    }
  }
}

// ---- attention logits of one side of a GAT relation without the transform: --------------------------
// a[i] = att . (W x_i) = (W^T att) . x_i with w~ = Wt att; a row then costs one float4 read and a quad fold.
template <int H>
__device__ void att_logits(const float* X, const float* Wt, const float* att, float* a_out, int n, const Grp& G) {
  constexpr int LPR = H / 4;                  // lanes per row
  constexpr int RPW = 64 / LPR;               // rows per wave per pass
  const int lane = threadIdx.x & 63, q = lane % LPR, rl = lane / LPR;
  if (G.w * RPW >= n) return;
  // w~ once per wave, spread over its lanes (every wave folding all of it out of LDS by itself saturated the
  // LDS pipe of the CU): lane L owns k = L % H and one 64/H-th of the sum over o, the parts meet through
  // xor-shuffles, the lane's quarter w~[4q .. 4q+3] arrives through four permutes
  constexpr int P = 64 / H, OW = H / P;
  float wk = 0.f;
  {
    const int k = lane % H, part = lane / H;
#pragma unroll
    for (int o4 = 0; o4 < OW / 4; ++o4) {
      const float4 at = *reinterpret_cast<const float4*>(att + part * OW + 4 * o4);
      const float4 wr = *reinterpret_cast<const float4*>(Wt + k * H + part * OW + 4 * o4);
      wk = fmaf(at.w, wr.w, fmaf(at.z, wr.z, fmaf(at.y, wr.y, fmaf(at.x, wr.x, wk))));
    }
#pragma unroll
    for (int off = H; off < 64; off <<= 1) wk += __shfl_xor(wk, off, 64);
  }
  float w[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) w[c] = __shfl(wk, 4 * q + c, 64);
  for (int i0 = G.w * RPW; i0 < n; i0 += G.nw * RPW) {
    const int i = i0 + rl;
    float d = 0.f;
    if (i < n) {
      const float4 x = *reinterpret_cast<const float4*>(X + i * H + 4 * q);
      d = fmaf(x.w, w[3], fmaf(x.z, w[2], fmaf(x.y, w[1], x.x * w[0])));
    }
#pragma unroll
    for (int off = LPR >> 1; off > 0; off >>= 1) d += __shfl_xor(d, off, 64);
    if (q == 0 && i < n) a_out[i] = d;
  }
}

// ---- the same product on the matrix cores: Y[n][H] = X[n][H] * Wt[H][H] (Wt[k][o], fp32) ---------
// v_mfma_f32_16x16x4_f32: a wave owns 16-row tiles; A[row][k] is one LDS word per lane per k-step,
// B[k][o] (the weights) stays in registers for all of the wave's tiles.  MASK: multiply the result
// by relu'(M[row][o]) (the backward's input gradient).  k runs in ascending order as in lin_blk.
template <int H, bool MASK>
__device__ void lin_mfma(const float* X, const float* Wt, float* Y, int n, const float* M, const Grp& G) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  constexpr int TD = H / 16, KS = H / 4;
  const int lane = threadIdx.x & 63, li = lane & 15, lj = lane >> 4;
  const int ntile = (n + 15) >> 4;
  if (G.w >= ntile) return;
  float b[TD][KS];
#pragma unroll
  for (int ct = 0; ct < TD; ++ct)
#pragma unroll
    for (int s = 0; s < KS; ++s) b[ct][s] = Wt[(4 * s + lj) * H + ct * 16 + li];
  for (int rt = G.w; rt < ntile; rt += G.nw) {
    const int r0 = rt * 16;
    const bool ok = r0 + li < n;
    const float* xr = X + (r0 + li) * H + lj;
    f32x4 acc[TD];
#pragma unroll
    for (int ct = 0; ct < TD; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float a = ok ? xr[4 * s] : 0.f;
#pragma unroll
      for (int ct = 0; ct < TD; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[ct][s], acc[ct], 0, 0, 0);
    }
#pragma unroll
    for (int ct = 0; ct < TD; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = r0 + lj * 4 + r;
        if (row < n) {
          const int idx = row * H + ct * 16 + li;
          Y[idx] = MASK ? (M[idx] > 0.f ? acc[ct][r] : 0.f) : acc[ct][r];
        }
      }
  }
}

// ---- one GCN layer in one phase: Y = relu((A_hat X) Wt + b) ------------------------------------------
// A wave owns 16-row tiles.  Lane (li = row in the tile, lj = quarter of the input features) gathers its
// quarter of row li of A_hat X straight into the A-operand registers of v_mfma_f32_16x16x4_f32 (k runs
// over the lane's own contiguous features: the weights are fetched in the matching order), the product with
// Wt comes off the matrix cores, bias + ReLU ride in the epilogue.  The n x H intermediate X Wt and the
// workgroup barrier between "transform" and "gather-reduce" do not exist: a layer is one barrier.
// (A_hat X) Wt instead of the reference's A_hat (X Wt): same value, rounded in another order.
// pool_part (optional, [H] words of this wave): the column sums of the rows this wave produced -- the wave's share of
// global_mean_pool, taken from the accumulators (rows of a tile in row order, tiles in tile order, then the four row
// groups of the lanes): the pooling pass over the finished layer and its barrier do not exist.
// E16: the structure comes as one 16-byte RECORD per row instead of rowptr / col (build_ell16_pair: rows of at most six
// edges): six neighbour ids as uint16 in edge order (unused slots hold the row itself) + the count in the top half of
// the last word.  One LDS read delivers the row's whole neighbourhood: the gather's dependent chain is two LDS round
// trips (record -> neighbour rows) instead of three (rowptr -> col -> rows).  Same edges in the same order: same bits.
template <int H, typename TS, bool E16 = false>
__device__ void gcn_fused(const int* rowptr, const int* col, const float* dinv, const float* X, const float* Wt,
                          const float* bias, float* Y, TS* __restrict__ gout, int n, const Grp& G,
                          float* pool_part = nullptr, float* pos_part = nullptr) {
  // pos_part (optional, [H] words of this wave, with pool_part): how many of the rows this wave produced are POSITIVE in
  // each column -- the first backward layer's bias gradient is that count times the pool gradient, so it needs no pass
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  constexpr int TD = H / 16, KS = H / 4;
  const int lane = threadIdx.x & 63, li = lane & 15, lj = lane >> 4;
  const int ntile = (n + 15) >> 4;
  if (G.w >= ntile) {
    if (pool_part && lj == 0) {
#pragma unroll
      for (int ct = 0; ct < TD; ++ct) {
        pool_part[ct * 16 + li] = 0.f;
        if (pos_part) pos_part[ct * 16 + li] = 0.f;
      }
    }
    return;
  }
  float ps[TD], pc[TD];
#pragma unroll
  for (int ct = 0; ct < TD; ++ct) { ps[ct] = 0.f; pc[ct] = 0.f; }
  float b[TD][KS], bia[TD];
#pragma unroll
  for (int ct = 0; ct < TD; ++ct) {
    bia[ct] = bias[ct * 16 + li];
#pragma unroll
    for (int s = 0; s < KS; ++s) b[ct][s] = Wt[(KS * lj + s) * H + ct * 16 + li];
  }
  // (Walking a wave's two tiles' gathers jointly -- two independent chains of LDS round trips in one branch-free
  // loop -- was built and measured: 31.8 us per step against 31.0; the wave with two tiles is not what a layer waits
  // for.  One tile at a time.)
  const float* xq = X + KS * lj;
  auto finish = [&](int rt_, const float (&z)[KS]) {
    const int r0 = rt_ * 16;
    f32x4 acc[TD];
#pragma unroll
    for (int ct = 0; ct < TD; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int ct = 0; ct < TD; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(z[s], b[ct][s], acc[ct], 0, 0, 0);
#pragma unroll
    for (int ct = 0; ct < TD; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = r0 + lj * 4 + r;
        if (row < n) {
          const int idx = row * H + ct * 16 + li;
          const float v = rnd<TS>(fmaxf(acc[ct][r] + bia[ct], 0.f));   // the activation as its storage type holds it
          Y[idx] = v;
          if (gout) stf(gout, (size_t)idx, v);
          ps[ct] += v;
          pc[ct] += v > 0.f ? 1.f : 0.f;
        }
      }
  };
  for (int rt = G.w; rt < ntile; rt += G.nw) {
    const int i = rt * 16 + li;
    float z[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) z[s] = 0.f;
    if (E16) {
      if (i < n) {
        const uint4 rec = reinterpret_cast<const uint4*>(rowptr)[i];
        const float di = dinv[i];
        const int c = (int)(rec.w >> 16);
        const int jr[8] = {(int)(rec.x & 0xffffu), (int)(rec.x >> 16), (int)(rec.y & 0xffffu), (int)(rec.y >> 16),
                           (int)(rec.z & 0xffffu), (int)(rec.z >> 16), (int)(rec.w & 0xffffu), (int)(rec.w & 0xffffu)};
#pragma unroll
        for (int p = 0; p < 8; p += 4) {
          if (p < c) {
            float w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) w[u] = p + u < c ? dinv[jr[p + u]] * di : 0.f;
#pragma unroll
            for (int q = 0; q < KS / 4; ++q) {
              float4 v[4];
#pragma unroll
              for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(xq + jr[p + u] * H + 4 * q);
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                z[4 * q + 0] = fmaf(w[u], v[u].x, z[4 * q + 0]);
                z[4 * q + 1] = fmaf(w[u], v[u].y, z[4 * q + 1]);
                z[4 * q + 2] = fmaf(w[u], v[u].z, z[4 * q + 2]);
                z[4 * q + 3] = fmaf(w[u], v[u].w, z[4 * q + 3]);
              }
            }
          }
        }
      }
    } else if (i < n) {
      const int s0 = rowptr[i], t0 = rowptr[i + 1];
      const float di = dinv[i];
      for (int p = s0; p < t0; p += 4) {
        int j[4];
        float w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) j[u] = col[p + u < t0 ? p + u : t0 - 1];
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = p + u < t0 ? dinv[j[u]] * di : 0.f;
#pragma unroll
        for (int q = 0; q < KS / 4; ++q) {
          float4 v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(xq + j[u] * H + 4 * q);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            z[4 * q + 0] = fmaf(w[u], v[u].x, z[4 * q + 0]);
            z[4 * q + 1] = fmaf(w[u], v[u].y, z[4 * q + 1]);
            z[4 * q + 2] = fmaf(w[u], v[u].z, z[4 * q + 2]);
            z[4 * q + 3] = fmaf(w[u], v[u].w, z[4 * q + 3]);
          }
        }
      }
    }
    finish(rt, z);
  }
  if (pool_part) {
#pragma unroll
    for (int ct = 0; ct < TD; ++ct) {
      float t = ps[ct];
      t += __shfl_xor(t, 16, 64);
      t += __shfl_xor(t, 32, 64);
      if (lj == 0) pool_part[ct * 16 + li] = t;
      if (pos_part) {
        float c = pc[ct];
        c += __shfl_xor(c, 16, 64);
        c += __shfl_xor(c, 32, 64);
        if (lj == 0) pos_part[ct * 16 + li] = c;
      }
    }
  }
}

template <int H>
struct Blk {  // outputs per lane in lin_blk: W columns (OPT*H floats) must stay in registers at 16 waves/CU
  static constexpr int OPT = H <= 16 ? 2 : 1;
};

// ---- Out[i] = act(sum_p (dc[col[p]]*dr[i]) * Hin[col[p]] + bias) ---------------------------------
template <int H, typename TS = float>
__device__ void agg_gcn_lds(const int* rowptr, const int* col, const float* dr, const float* dc,
                            const float* Hin, const float* bias, float* Out, int n, int relu,
                            TS* __restrict__ gout /* global rows (an activation: rounded to TS) or null */, const Grp& G) {
  constexpr int LPR = H / 4;
  const int RPB = G.nt / LPR;
  const int rl = G.t / LPR, f = (G.t % LPR) * 4;
  float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias) b = *reinterpret_cast<const float4*>(bias + f);
  for (int i = rl; i < n; i += RPB) {
    const int s = rowptr[i], t = rowptr[i + 1];
    const float di = dr[i];
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = s; p < t; ++p) {
      const int j = col[p];
      const float w = mul_rn(dc[j], di);
      const float4 v = *reinterpret_cast<const float4*>(Hin + j * H + f);
      a.x = add_rn(a.x, mul_rn(w, v.x));
      a.y = add_rn(a.y, mul_rn(w, v.y));
      a.z = add_rn(a.z, mul_rn(w, v.z));
      a.w = add_rn(a.w, mul_rn(w, v.w));
    }
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    if (relu) {
      a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
    }
    if (gout) { a.x = rnd<TS>(a.x); a.y = rnd<TS>(a.y); a.z = rnd<TS>(a.z); a.w = rnd<TS>(a.w); }
    *reinterpret_cast<float4*>(Out + i * H + f) = a;
    if (gout) stf4(gout, ((size_t)i * H + f) >> 2, a);
  }
}

// LDS layout shared by host sizing and kernel carve (all counts in 4-byte words)
struct FwdLayout {
  size_t xa, bh, xva, xvb, zs, a_s, a_d, sc, dinv, dinv_v, wt, headw, part, vec;
  size_t rowptr, col, rowptr_lv, col_lv, rowptr_vv, col_vv, cursorA, tmpA, cursorB, tmpB, wsum;
  size_t rowptr_t, col_t, cursorT, tmpT, cursorV, tmpV;
  size_t ck_tab, ck_first, ck_arrive, gpart;  // softmax chunks of the lv relation (64 members each)
  size_t ek_ll, eo_ll, ek_lv, eo_lv, ek_vv, eo_vv, total;
};
__host__ __device__ inline FwdLayout fwd_layout(int H, int C, int max_n, int max_v, int max_ell, int max_evv,
                                                int db, int exp) {
  FwdLayout Y;
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o += (n + 3) & ~(size_t)3; return r; };  // keep 16-B alignment
  Y.xa = take((size_t)max_n * H);
  // scratch of the CSR builds lives inside xa while it can: the features are parked there only after the
  // builds (they wait in registers), so the counters / slot lists need no LDS of their own
  size_t xq = Y.xa;
  const size_t xend = Y.xa + (size_t)max_n * H;
  auto scratch = [&](size_t n) {
    const size_t need = (n + 3) & ~(size_t)3;
    if (xq + need <= xend) { const size_t r = xq; xq += need; return r; }
    return take(n);
  };
  // second n x H buffer (a local layer writes it, then the two swap roles); the staged COO slices (needed only
  // before layer 0) overlay it
  const size_t stage = (size_t)2 * (((size_t)max_ell + 3) / 4 * 4 + ((size_t)max_n + 3) / 4 * 4 +
                                    ((size_t)max_evv + 3) / 4 * 4);
  const size_t two = (size_t)max_n * H;
  const size_t region = take(two > stage ? two : stage);
  Y.bh = region;
  {
    size_t p = region;
    auto sub = [&](size_t n) { size_t r = p; p += (n + 3) & ~(size_t)3; return r; };
    Y.ek_ll = sub(max_ell); Y.eo_ll = sub(max_ell);
    Y.ek_lv = sub(max_n);   Y.eo_lv = sub(max_n);
    Y.ek_vv = sub(max_evv); Y.eo_vv = sub(max_evv);
  }
  Y.xva = take((size_t)max_v * H);
  Y.xvb = take((size_t)max_v * H);
  Y.zs = take(max_v ? (size_t)(RT_MAX / 64) * H : 0);   // per wave: an aggregated row of the cluster it finishes
  Y.a_s = take(max_n);
  Y.a_d = take(max_v);
  Y.sc = take(max_n);
  Y.dinv = take(max_n);
  Y.dinv_v = take(max_v);
  Y.wt = take((size_t)(db ? 2 : 1) * (4 * H * H + 5 * H));  // layer weights (double-buffered when LDS allows)
  Y.headw = take((size_t)H * H + H + (size_t)C * H + C);
  Y.part = take((size_t)(RT_MAX / 64) * H);           // one H-vector per wave (pool partials)
  Y.vec = take(128);
  Y.rowptr = take(max_n + 1);
  Y.col = take(max_ell);
  Y.rowptr_lv = take(max_v + 1);
  Y.col_lv = take(max_n);
  Y.rowptr_vv = take(max_v + 1);
  Y.col_vv = take(max_evv);
  Y.cursorA = scratch(max_n + 1);
  // softmax chunk partials (layers only) share the words of the ll build's scratch (structure only)
  const size_t maxck = (size_t)max_n / 64 + max_v + 1;   // sum over clusters of max(1, ceil(size / 64))
  const size_t gwords = max_v ? maxck * H : 0;
  Y.tmpA = take((size_t)max_ell > gwords ? (size_t)max_ell : gwords);
  Y.gpart = Y.tmpA;
  const int nchunk = (max_n + 63) / 64;
  size_t cb = (size_t)max_v + 1;
  if ((size_t)max_v * nchunk > cb) cb = (size_t)max_v * nchunk;  // multisplit counters
  Y.cursorB = scratch(cb);
  Y.tmpB = scratch(max_n > max_evv ? max_n : max_evv);
  Y.wsum = take(32);
  Y.rowptr_t = take(exp ? max_n + 1 : 0);
  Y.col_t = take(exp ? max_ell : 0);
  Y.cursorT = scratch(exp ? max_n + 1 : 0);
  Y.tmpT = scratch(exp ? max_ell : 0);
  Y.cursorV = scratch(max_v + 1);
  Y.tmpV = scratch(max_evv);
  Y.ck_tab = take(max_v ? maxck : 0);
  Y.ck_first = take(max_v ? max_v + 2 : 0);     // + the chunk count and the largest cluster's size
  Y.ck_arrive = take(max_v ? max_v + 1 : 0);   // + the work counter of the chunk list (ck_arrive[max_v])
  Y.total = o;
  return Y;
}

// MODE: 0 = what the arguments say; 1 = local chain + head only (compute_virtual == 0), 2 = virtual branch only
// (compute_virtual == 2) from its beginning, 4 = virtual branch only, resumed at layer l_begin > 0 from exported
// state, 3 = both branches (compute_virtual == 1) known at compile time -- the two workgroup programs of the paired launches are compiled
// as their own specialisations, so each fetches only the code of its own path (the generic body is 75 KB of ISA
// against a 64 KB instruction cache shared by two CUs, and a workgroup runs its program once per launch).
// TS: storage type of features / activations in HBM (float or half_t; the argument block carries them as float*).
template <int H, int RT, int MODE = 0, typename TS = float, typename AT = FwdArgs>
__device__ __forceinline__ void hscn_fwd_body(const AT& A, const int g) {
  extern __shared__ __align__(16) unsigned char smem[];
  // MODE 5 = the virtual branch of the one-launch training step (resident_step.h): what the launch fixes is a
  // compile-time constant here, so its code paths and the scalar registers that steer them do not exist
  constexpr bool STEPV = MODE == 5 || MODE == 6;
  const int a_exp = STEPV ? 0 : A.exp, a_exp_dinv = STEPV ? 0 : A.exp_dinv, a_spec = STEPV ? 1 : A.spec;
  const int l_begin = STEPV ? 0 : A.l_begin, l_end = STEPV ? A.L : A.l_end;
  const int a_acq = STEPV ? (RT <= 256 ? 1 : 0) : A.acq;
  const TS* const xl_g = reinterpret_cast<const TS*>(A.x_local);
  const TS* const xv_g = reinterpret_cast<const TS*>(A.x_virtual);
  TS* const acts_g = reinterpret_cast<TS*>(A.acts);
  TS* const vsxv_g = reinterpret_cast<TS*>(A.vs_xv);
  TS* const xvout_g = reinterpret_cast<TS*>(A.xv_out);
  constexpr int OPT = Blk<H>::OPT;
  constexpr int NW = RT / 64;
  constexpr int WSZ = 4 * H * H + 5 * H;
  const int n0 = A.lptr[g], n = A.lptr[g + 1] - n0;
  const int v0 = A.vptr[g], nv = A.vptr[g + 1] - v0;
  const bool vonly = MODE == 2 || MODE == 4 || MODE == 5 || MODE == 6 || (MODE == 0 && A.compute_virtual == 2);  // virtual branch only: the local activations come from `acts`
  // (a virtual-only workgroup touches the ll edges only when it builds the source-keyed CSR for the backward)
  const int e0 = A.eptr_ll[g], ne = (vonly && !a_exp) ? 0 : A.eptr_ll[g + 1] - e0;
  const int ev0 = A.eptr_vv[g], nev = A.eptr_vv[g + 1] - ev0;
  const int el0 = A.eptr_lv[g], nel = A.eptr_lv[g + 1] - el0;
  if ((n > A.max_n) | (nv > A.max_v) | (ne > A.max_ell) | (nev > A.max_evv) | (nel > A.max_n) | (n < 0) | (nv < 0)) {
    if (threadIdx.x == 0 && A.flag) atomicOr(A.flag, 4);
    // the launches that consume this workgroup's exports walk them without knowing about the error: leave
    // empty structure (all-zero row pointers) behind, never stale memory
    if (a_exp && A.csr_rowptr_t && n >= 0)
      for (int i = threadIdx.x; i <= n; i += RT) A.csr_rowptr_t[(size_t)n0 + g + i] = 0;
    if (vonly && l_begin == 0 && l_end < A.L && A.vs_rowptr_lv && nv >= 0)
      for (int i = threadIdx.x; i <= nv; i += RT) {
        A.vs_rowptr_lv[(size_t)v0 + g + i] = 0;
        A.vs_rowptr_vv[(size_t)v0 + g + i] = 0;
      }
    return;
  }
  // MODE 6: the capacities that size the LDS layout are compile-time constants (StepVCaps; the host takes this form
  // when the batch's maxima fit them), so every LDS address below is an immediate, not a scalar register that lives
  // -- or is spilled -- through the whole program
  constexpr bool FIXL = MODE == 6;
  const int cap_v = FIXL ? StepVCaps::V : A.max_v;
  const int a_db = FIXL ? 1 : A.db;
  const FwdLayout Y = FIXL ? fwd_layout(H, 1, StepVCaps::N, StepVCaps::V, 0, StepVCaps::EVV, 1, 0)
                           : fwd_layout(H, A.C, A.max_n, A.max_v, A.max_ell, A.max_evv, A.db, a_exp);
  float* fb = reinterpret_cast<float*>(smem);
  int* ib = reinterpret_cast<int*>(smem);
  float *xa = fb + Y.xa, *bh = fb + Y.bh, *xva = fb + Y.xva, *xvb = fb + Y.xvb, *zs = fb + Y.zs;
  float *a_s = fb + Y.a_s, *a_d = fb + Y.a_d, *sc = fb + Y.sc, *dinv = fb + Y.dinv, *dinv_v = fb + Y.dinv_v;
  float *wt = fb + Y.wt, *headw = fb + Y.headw, *part = fb + Y.part, *vec = fb + Y.vec;
  int *rowptr = ib + Y.rowptr, *col = ib + Y.col, *rowptr_lv = ib + Y.rowptr_lv, *col_lv = ib + Y.col_lv;
  int *rowptr_vv = ib + Y.rowptr_vv, *col_vv = ib + Y.col_vv;
  int* wsum = ib + Y.wsum;
  int *ck_tab = ib + Y.ck_tab, *ck_first = ib + Y.ck_first, *ck_arrive = ib + Y.ck_arrive;
  float* gpart = fb + Y.gpart;
  const bool cv = MODE >= 2 || (MODE == 0 && A.compute_virtual != 0);
  const int F = A.F;

  // wave groups: with the virtual branch on, the upper half of the waves works on it
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (scalar: "tiles of this wave" loops stay uniform)
  // structure build always splits when the virtual branch is on; a virtual-only launch keeps a
  // quarter of the waves for streaming the next layer's local activations in
  const int NA = cv ? (vonly ? (NW >= 4 ? NW / 4 : 1) : NW / 2) : NW;
  const bool inB = wave >= NA;
  const Grp ALL{(int)threadIdx.x, RT, wave, NW};
  const Grp GA{(int)threadIdx.x, NA * 64, wave, NA};
  const Grp GB{(int)threadIdx.x - NA * 64, (NW - NA) * 64, wave - NA, NW - NA};

  // softmax work list (layer independent), built by ONE wave: cluster v is cut into
  // max(1, ceil(size/64)) chunks of 64 members; entry k = (v << 8) | index inside v,
  // ck_first[v] = first chunk of v, ck_first[nv] = number of chunks, ck_first[nv + 1] = size of the largest cluster
  auto build_chunk_table = [&]() {
    const int lane = threadIdx.x & 63;
    int carry = 0, maxsz = 0;
    for (int base = 0; base < nv; base += 64) {
      const int v = base + lane;
      const int sz = v < nv ? rowptr_lv[v + 1] - rowptr_lv[v] : 0;
      maxsz = max(maxsz, wave_max_int(sz));
      const int cnt = v < nv ? (sz > 64 ? (sz + 63) >> 6 : 1) : 0;
      const int incl = wave_incl_scan(cnt);
      const int first = carry + incl - cnt;
      if (v < nv) {
        ck_first[v] = first;
        ck_arrive[v] = 0;
        for (int c = 0; c < cnt; ++c) ck_tab[first + c] = (v << 8) | c;
      }
      carry += __builtin_amdgcn_readlane(incl, 63);
    }
    if (lane == 0) { ck_first[nv] = carry; ck_first[nv + 1] = maxsz; }
  };

  // ---- prologue: request every global input of this graph, then consume -----------------------
  STAMP(0);
  WStage<H, RT> ws;
  const bool resume = MODE == 4 || (MODE == 0 && vonly && l_begin > 0);
  const bool prestruct = vonly && A.pre_rp_lv != nullptr && l_begin == 0;
  if (!resume && !prestruct) {
  constexpr int EPT = 2;   // edges per thread held in registers (covers RT*EPT edges per relation)
  constexpr int XPT = 8;   // feature words per thread held in registers
  // raw 64-bit ids first (clamped addresses, no arithmetic on the results yet): all requests of the
  // prologue are in flight together
  const int64_t* dummy = reinterpret_cast<const int64_t*>(A.lptr);
  const int64_t *pld = A.ll_dst ? A.ll_dst : dummy, *pls = A.ll_src ? A.ll_src : dummy;
  const int64_t *pvd = (cv && A.lv_dst) ? A.lv_dst : dummy, *pvs = (cv && A.lv_src) ? A.lv_src : dummy;
  const int64_t *pwd = (cv && A.vv_dst) ? A.vv_dst : dummy, *pws = (cv && A.vv_src) ? A.vv_src : dummy;
  long long rld[EPT], rls[EPT], rvd[EPT], rvs[EPT], rwd[EPT], rws[EPT];
  float xr[XPT], xvr[2], hw0 = 0.f, hw1 = 0.f;
  // a wave whose whole 64-element slice lies past the end of an array skips the request (scalar
  // branch on the wave's first index): the vector-memory pipe of the CU sees only useful requests
  const int wbase = __builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));
#pragma unroll
  for (int i = 0; i < EPT; ++i) {
    const int e = threadIdx.x + i * RT;
    const int eb = wbase + i * RT;
    const bool o1 = e < ne && A.ll_dst, o2 = cv && e < nel && A.lv_dst, o3 = cv && e < nev && A.vv_dst;
    rld[i] = 0; rls[i] = 0; rvd[i] = 0; rvs[i] = 0; rwd[i] = 0; rws[i] = 0;
    if (eb < ne) { rld[i] = pld[o1 ? e0 + e : 0]; rls[i] = pls[o1 ? e0 + e : 0]; }
    if (cv && eb < nel) { rvd[i] = pvd[o2 ? el0 + e : 0]; rvs[i] = pvs[o2 ? el0 + e : 0]; }
    if (cv && eb < nev) { rwd[i] = pwd[o3 ? ev0 + e : 0]; rws[i] = pws[o3 ? ev0 + e : 0]; }
  }
  ws.fetch(layer_of(A, 0), !vonly, cv, F);   // after the edges: they are consumed first
#pragma unroll
  for (int i = 0; i < XPT; ++i) {
    const int idx = threadIdx.x + i * RT;
    const int r = idx / H, k = idx - r * H;
    const bool ok = idx < n * H && k < F;
    xr[i] = 0.f;
    if (wbase + i * RT < n * H) {
      const float t = ldf(xl_g, ok ? (size_t)(n0 + r) * F + k : 0);
      xr[i] = ok ? t : 0.f;
    }
  }
  const TS* pxv = (cv && A.x_virtual) ? xv_g : xl_g;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = threadIdx.x + i * RT;
    const int r = idx / H, k = idx - r * H;
    const bool ok = cv && idx < nv * H && k < F;
    xvr[i] = 0.f;
    if (cv && wbase + i * RT < nv * H) {
      const float t = ldf(pxv, ok ? (size_t)(v0 + r) * F + k : 0);
      xvr[i] = ok ? t : 0.f;
    }
  }
  // head weights: W1 [H][H] | b1 [H] | W2 [C][H] | b2 [C]   (natural layout)
  constexpr int HPT = (H * H + RT - 1) / RT;
  float hw1r[HPT], hw2r[2], hb1, hb2;
#pragma unroll
  for (int i = 0; i < HPT; ++i) {
    const int idx = threadIdx.x + i * RT;
    hw1r[i] = vonly ? 0.f : A.W1[idx < H * H ? idx : 0];
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = threadIdx.x + i * RT;
    hw2r[i] = vonly ? 0.f : A.W2[idx < A.C * H ? idx : 0];
  }
  hb1 = vonly ? 0.f : A.b1[threadIdx.x < H ? threadIdx.x : 0];
  hb2 = vonly ? 0.f : A.b2[(int)threadIdx.x < A.C ? threadIdx.x : 0];
  (void)hw0; (void)hw1;
  int kll[EPT], oll[EPT], klv[EPT], olv[EPT], kvv[EPT], ovv[EPT];
#pragma unroll
  for (int i = 0; i < EPT; ++i) {
    kll[i] = (int)(rld[i] - n0); oll[i] = (int)(rls[i] - n0);
    klv[i] = (int)(rvd[i] - v0); olv[i] = (int)(rvs[i] - n0);
    kvv[i] = (int)(rwd[i] - v0); ovv[i] = (int)(rws[i] - v0);
  }
  // ---- consume: validate + park in LDS ---------------------------------------------------------
  {
    int *ek_ll = ib + Y.ek_ll, *eo_ll = ib + Y.eo_ll, *ek_lv = ib + Y.ek_lv, *eo_lv = ib + Y.eo_lv;
    int *ek_vv = ib + Y.ek_vv, *eo_vv = ib + Y.eo_vv;
    bool bad = false;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const int e = threadIdx.x + i * RT;
      if (e < ne) {
        int k = kll[i];
        int o_ = oll[i];
        if (k < 0 || k >= n || o_ < 0 || o_ >= n) { bad = true; k = -1; o_ = -1; }
        ek_ll[e] = k; eo_ll[e] = o_;
      }
      if (cv && e < nel) {
        int k = klv[i];
        if (k < 0 || k >= nv || olv[i] < 0 || olv[i] >= n) { bad = true; k = -1; }
        ek_lv[e] = k; eo_lv[e] = olv[i];
      }
      if (cv && e < nev) {
        int k = kvv[i];
        if (k < 0 || k >= nv || ovv[i] < 0 || ovv[i] >= nv) { bad = true; k = -1; }
        ek_vv[e] = k; eo_vv[e] = ovv[i];
      }
    }
    // slices longer than RT*EPT edges (not the LRGB case): straight copy
    for (int e = threadIdx.x + EPT * RT; e < ne; e += RT) {
      int k = (int)(A.ll_dst[e0 + e] - n0); int o = (int)(A.ll_src[e0 + e] - n0);
      if (k < 0 || k >= n || o < 0 || o >= n) { bad = true; k = -1; o = -1; }
      ek_ll[e] = k; eo_ll[e] = o;
    }
    if (cv) {
      for (int e = threadIdx.x + EPT * RT; e < nel; e += RT) {
        int k = (int)(A.lv_dst[el0 + e] - v0); const int o = (int)(A.lv_src[el0 + e] - n0);
        if (k < 0 || k >= nv || o < 0 || o >= n) { bad = true; k = -1; }
        ek_lv[e] = k; eo_lv[e] = o;
      }
      for (int e = threadIdx.x + EPT * RT; e < nev; e += RT) {
        int k = (int)(A.vv_dst[ev0 + e] - v0); const int o = (int)(A.vv_src[ev0 + e] - v0);
        if (k < 0 || k >= nv || o < 0 || o >= nv) { bad = true; k = -1; }
        ek_vv[e] = k; eo_vv[e] = o;
      }
    }
    if (bad && A.flag) atomicOr(A.flag, 2);
  }
  // the CSR builds count in these (they hand them back zeroed)
  for (int i = threadIdx.x; i <= n; i += RT) {
    (ib + Y.cursorA)[i] = 0;
    if (a_exp) (ib + Y.cursorT)[i] = 0;
  }
  if (cv)
    for (int i = threadIdx.x; i <= nv; i += RT) (ib + Y.cursorV)[i] = 0;
  lds_barrier();
  STAMP(1);
  // ---- structure: the CSRs are independent, so wave groups build them side by side between the
  // same barriers: ll keyed by target (forward), ll keyed by source (exported for the backward
  // launch), vv (four barriers each, one idle when the lv build runs beside them), and lv (multisplit:
  // five barriers)
  {
    int *rowptr_t = ib + Y.rowptr_t, *col_t = ib + Y.col_t;
    const int w_t = !a_exp ? 0 : (cv ? (NW * 3) / 8 : NW - NW / 2);          // 6 of 16 waves (1 of 4)
    const int w_ll = vonly ? 0 : (cv ? (a_exp ? (NW * 3) / 8 : NW / 2) : NW - w_t);
    const int w_vv = cv ? (NW - w_ll - w_t) / 2 : 0;
    const int w_lv = cv ? NW - w_ll - w_t - w_vv : 0;
    const int wa = wave < w_ll ? 0 : (wave < w_ll + w_t ? 1 : (wave < w_ll + w_t + w_vv ? 2 : 3));
    const int wbase = wa == 0 ? 0 : (wa == 1 ? w_ll : (wa == 2 ? w_ll + w_t : w_ll + w_t + w_vv));
    const int wcnt = wa == 0 ? w_ll : (wa == 1 ? w_t : (wa == 2 ? w_vv : w_lv));
    const Grp GS{(int)threadIdx.x - wbase * 64, wcnt * 64, wave - wbase, wcnt};
    static_assert(CSR_MULTISPLIT_BARRIERS == CSR_BUILD_BARRIERS + 1, "barrier sequences of the wave groups must match");
    if (wa == 0) {
      build_csr_lds(ib + Y.ek_ll, ib + Y.eo_ll, ne, n, rowptr, col, ib + Y.cursorA, ib + Y.tmpA, GS, false);
      dinv_from_rowptr(rowptr, n, dinv, GS);
      if (cv) lds_barrier();   // keeps step with the lv multisplit
    } else if (wa == 1) {
      build_csr_lds(ib + Y.eo_ll, ib + Y.ek_ll, ne, n, rowptr_t, col_t, ib + Y.cursorT, ib + Y.tmpT, GS, false);
      if (cv) lds_barrier();
    } else if (wa == 2) {
      build_csr_lds(ib + Y.ek_vv, ib + Y.eo_vv, nev, nv, rowptr_vv, col_vv, ib + Y.cursorV, ib + Y.tmpV, GS, false);
      dinv_from_rowptr(rowptr_vv, nv, dinv_v, GS);
      lds_barrier();
    } else {
      build_csr_multisplit_lds(ib + Y.ek_lv, ib + Y.eo_lv, nel, nv, rowptr_lv, col_lv, ib + Y.cursorB, ib + Y.tmpB,
                               wsum + 24, GS);
      if (GS.w == 0) build_chunk_table();
    }
    // features and weights were requested with the edges but are not needed before layer 0: they are
    // parked now, their latency spent under the CSR builds
#pragma unroll
  for (int i = 0; i < XPT; ++i) {
    const int idx = threadIdx.x + i * RT;
    if (idx < n * H) xa[idx] = xr[i];
  }
  for (int idx = threadIdx.x + XPT * RT; idx < n * H; idx += RT) {
    const int r = idx / H, k = idx - r * H;
    xa[idx] = k < F ? ldf(xl_g, (size_t)(n0 + r) * F + k) : 0.f;
  }
  if (cv) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = threadIdx.x + i * RT;
      if (idx < nv * H) xva[idx] = xvr[i];
    }
    for (int idx = threadIdx.x + 2 * RT; idx < nv * H; idx += RT) {
      const int r = idx / H, k = idx - r * H;
      xva[idx] = k < F ? ldf(xv_g, (size_t)(v0 + r) * F + k) : 0.f;
    }
  }
#pragma unroll
  for (int i = 0; i < HPT; ++i) {
    const int idx = threadIdx.x + i * RT;
    if (idx < H * H) headw[idx] = hw1r[i];
  }
  if (threadIdx.x < H) headw[H * H + threadIdx.x] = hb1;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = threadIdx.x + i * RT;
    if (idx < A.C * H) headw[H * H + H + idx] = hw2r[i];
  }
  if (!vonly) {
    for (int idx = threadIdx.x + 2 * RT; idx < A.C * H; idx += RT) headw[H * H + H + idx] = A.W2[idx];
    for (int idx = threadIdx.x; idx < A.C; idx += RT)
      headw[H * H + H + A.C * H + idx] = idx == (int)threadIdx.x ? hb2 : A.b2[idx];
  }
  ws.store(wt);
    STAMP(2);
    lds_barrier();
    // export the source-keyed CSR and the degree norm for the backward launch
    if (a_exp) {
      for (int i = threadIdx.x; i <= n; i += RT) A.csr_rowptr_t[(size_t)n0 + g + i] = rowptr_t[i];
      const int cnt_t = rowptr_t[n];
      for (int p = threadIdx.x; p < cnt_t; p += RT) A.csr_col_t[(size_t)e0 + p] = col_t[p];
    }
    if (a_exp_dinv)
      for (int i = threadIdx.x; i < n; i += RT) A.dinv_out[(size_t)n0 + i] = dinv[i];
  }
  } else if (prestruct) {
    // ---- virtual branch on dataset-resident structure: nothing is staged or built, the CSRs of the two virtual
    // relations and the degree norm are loaded (graph-local ids, exactly what the builds below would produce)
    ws.fetch(layer_of(A, 0), false, true, F);
    for (int idx = threadIdx.x; idx < n * H; idx += RT) {
      const int r = idx / H, k = idx - r * H;
      xa[idx] = k < F ? ldf(xl_g, (size_t)(n0 + r) * F + k) : 0.f;
    }
    for (int idx = threadIdx.x; idx < nv * H; idx += RT) {
      const int r = idx / H, k = idx - r * H;
      xva[idx] = k < F ? ldf(xv_g, (size_t)(v0 + r) * F + k) : 0.f;
    }
    for (int i = threadIdx.x; i <= nv; i += RT) {
      rowptr_lv[i] = A.pre_rp_lv[(size_t)v0 + g + i];
      rowptr_vv[i] = A.pre_rp_vv[(size_t)v0 + g + i];
    }
    for (int i = threadIdx.x; i < nel; i += RT) col_lv[i] = A.pre_col_lv[(size_t)el0 + i];
    for (int i = threadIdx.x; i < nev; i += RT) col_vv[i] = A.pre_col_vv[(size_t)ev0 + i];
    for (int i = threadIdx.x; i < nv; i += RT) dinv_v[i] = A.pre_dinv_v[(size_t)v0 + i];
    ws.store(wt);
    lds_barrier();
    if (wave == 0) build_chunk_table();
    lds_barrier();
  } else {
    // ---- resumed virtual branch: the structure and the virtual features come from the state the
    // first part of this step exported, the local rows from the previous layer's activations
    ws.fetch(layer_of(A, l_begin), false, true, H);
    const TS* src = acts_g + ((size_t)(l_begin - 1) * A.N + n0) * H;
    for (int i = threadIdx.x; i < n * (H / 4); i += RT) reinterpret_cast<float4*>(xa)[i] = ldf4(src, i);
    for (int i = threadIdx.x; i < nv * H; i += RT) xva[i] = ldf(vsxv_g, (size_t)v0 * H + i);
    for (int i = threadIdx.x; i <= nv; i += RT) {
      rowptr_lv[i] = A.vs_rowptr_lv[(size_t)v0 + g + i];
      rowptr_vv[i] = A.vs_rowptr_vv[(size_t)v0 + g + i];
    }
    for (int i = threadIdx.x; i < nel; i += RT) col_lv[i] = A.vs_col_lv[(size_t)el0 + i];
    for (int i = threadIdx.x; i < nev; i += RT) col_vv[i] = A.vs_col_vv[(size_t)ev0 + i];
    for (int i = threadIdx.x; i < nv; i += RT) dinv_v[i] = A.vs_dinv_v[(size_t)v0 + i];
    ws.store(wt + (a_db ? (l_begin & 1) * WSZ : 0));
    lds_barrier();
    if (wave == 0) build_chunk_table();
    lds_barrier();
  }
  STAMP(3);

  for (int l = l_begin; l < l_end; ++l) {
    const bool DB = a_db != 0;                 // two weight buffers: the next layer's land under this layer's math
    float* W = wt + (DB ? (l & 1) * WSZ : 0);
    float* Wn = wt + (DB ? ((l + 1) & 1) * WSZ : 0);
    const float* b_ll = W + 4 * H * H;
    const float* b_vv = b_ll + H;
    const float* b_gat = b_ll + 2 * H;
    const float* att_s = b_ll + 3 * H;
    const float* att_d = b_ll + 4 * H;
    const bool more = l + 1 < l_end;
    // fetch the next layer's weights now, park them in LDS under this layer's math
    if (more && DB) ws.fetch(layer_of(A, l + 1), !vonly, cv, H);
    STAMP(4 + 4 * l);
    auto transforms_ll = [&](const Grp& G_) {
      if (H <= 32) lin_mfma<H, false>(xa, W, bh, n, nullptr, G_);
      else lin_blk<H, OPT>(xa, W, bh, n, nullptr, nullptr, G_);
    };
    // virtual-only launch: layer l+1 reads the local activations the local launch stored
    auto load_next_local = [&](const Grp& G_, float* to) {
      if (!more) return;
      const TS* src = acts_g + ((size_t)l * A.N + n0) * H;
      float4* dst = reinterpret_cast<float4*>(to);
      for (int i = G_.t; i < n * (H / 4); i += G_.nt) dst[i] = ldf4(src, i);
    };
    // The virtual branch's transforms are linear and sit in front of linear aggregations, so they move BEHIND
    // them (a cluster row instead of every member row is transformed):
    //   lv GAT  out_v = W_src (sum_i alpha_i x_i) + b,  alpha from a_s[i] = (W_src^T att_src) . x_i and
    //           a_d[v] = (W_dst^T att_dst) . xv_v;       vv GCN  out_v = W_vv (sum_u norm_uv xv_u) + b.
    // Phase 1 is the n + nv attention dots; phase 2 aggregates INPUT rows and the finishing wave of a cluster
    // applies the two H x H matrices to the two aggregated rows.  Same values as transform-then-aggregate,
    // another rounding order.
    auto transforms_virtual = [&](const Grp& G_) {
      if (G_.t == 0) ck_arrive[cap_v] = 0;   // this layer's chunk counter (the phase barrier orders it before the reduce)
      att_logits<H>(xa, W + H * H, att_s, a_s, n, G_);
      att_logits<H>(xva, W + 2 * H * H, att_d, a_d, nv, G_);
    };
    auto reduce_ll = [&](const Grp& G_) {
      agg_gcn_lds<H, TS>(rowptr, col, dinv, dinv, bh, b_ll, xa, n, 1, acts_g + ((size_t)l * A.N + n0) * H, G_);
    };
    // H <= 32: the whole local layer in one phase, xa -> bh (the two buffers swap roles after the layer)
    constexpr bool FUSE = H <= 32;
    auto layer_ll = [&](const Grp& G_) {
      gcn_fused<H, TS>(rowptr, col, dinv, xa, W, b_ll, bh, acts_g + ((size_t)l * A.N + n0) * H, n, G_);
    };
    // lv segment softmax + weighted sum, one wave per 64-member chunk of a cluster (clusters are as
    // unbalanced as the assignment makes them: one wave per cluster would serialise the big one).
    // Every chunk wave recomputes the cluster's max / denominator (a few LDS reads), reduces its own
    // members into a partial row, and the wave that arrives last at the cluster's counter adds the
    // partials in chunk order, the vv GCN row and the biases.  LDS executes a wave's operations in
    // order and the counter is acquire/release, so the partials are visible to the finisher.
    auto reduce_virtual = [&](const Grp& G_) {
      constexpr int LPR = H / 4 > 64 ? 64 : H / 4;
      constexpr int S = 64 / LPR;
      const int lane = threadIdx.x & 63, slot = lane / LPR, f = (lane % LPR) * 4;
      const int nck = ck_first[nv];
      // chunks are handed out through an LDS counter (zeroed in the layer's first phase), not by wave index: whichever
      // wave is free takes the next one -- in a virtual-only workgroup the loading waves join in once their rows are
      // parked, so 16 clusters of a graph take one round on 16 waves instead of two on 12.  A chunk's arithmetic does
      // not depend on who runs it, a cluster is finished by its last arriver in chunk order: same bits as before.
      int* ck_next = ck_arrive + cap_v;
      (void)G_;
#ifndef HSCN_NO_QUAD_CLUSTERS
      if constexpr (H == 16) {
        // ---- many small clusters (a balanced assignment: K = 16 .. 32 clusters of a few members each) ----
        // One wave per cluster leaves most of the wave idle and takes nv / NW rounds.  Here a 16-lane DPP row owns a
        // cluster, four clusters per wave: lane r of the row computes the attention logit of member (r % 4) * 4 + r / 4
        // of the current 16-member tile, so the four lanes of quad q hold members q, q+4, q+8, q+12 -- exactly the
        // members slot q (= the quad, four lanes x float4 = one 16-feature row) accumulates, reached by quad
        // broadcasts; softmax max / sum are row reductions, the slots fold by row rotations, and the two 16 x 16
        // transforms read their vector through row broadcasts.  No LDS traffic besides the operands themselves.
        // Same arithmetic per element as the chunk path below, another summation order (members of a cluster
        // interleaved over four slots instead of sixteen).
        const int maxsz = ck_first[nv + 1];
        if (maxsz <= 16 || (maxsz <= 64 && nv > NW)) {
          const int grp = lane >> 4, r = lane & 15, q = r >> 2, u4 = r & 3;
          const int pm = u4 * 4 + q, fq = u4 * 4;
          const int nq = (nv + 3) >> 2;
          const float* Ws = W + H * H;                     // Wt_src[k][o]
          const float* Wv = W + 3 * H * H;                 // Wt_vv[k][o]
          for (;;) {
            int ck = 0;
            if (lane == 0) ck = __hip_atomic_fetch_add(ck_next, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            ck = __builtin_amdgcn_readfirstlane(ck);
            if (ck >= nq) break;
            const int v = 4 * ck + grp;
            const bool vok = v < nv;
            const int vc = vok ? v : nv - 1;               // (a row past the last cluster: empty ranges, nothing stored)
            const int s = rowptr_lv[vc], t = vok ? rowptr_lv[vc + 1] : s;
            const int s2 = rowptr_vv[vc], t2 = vok ? rowptr_vv[vc + 1] : s2;
            const float ad = a_d[vc], di = dinv_v[vc];
            const int cnt = t - s, cnt2 = t2 - s2;
            const int tiles = (max(max(__builtin_amdgcn_readlane(cnt, 0), __builtin_amdgcn_readlane(cnt, 16)),
                                   max(__builtin_amdgcn_readlane(cnt, 32), __builtin_amdgcn_readlane(cnt, 48))) + 15) >> 4;
            const int tiles2 = (max(max(__builtin_amdgcn_readlane(cnt2, 0), __builtin_amdgcn_readlane(cnt2, 16)),
                                    max(__builtin_amdgcn_readlane(cnt2, 32), __builtin_amdgcn_readlane(cnt2, 48))) + 15) >> 4;
            float e[4];
            int jm[4];
            float m = -INFINITY;
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) {
              e[ti] = -INFINITY;
              jm[ti] = 0;
              if (ti < tiles) {
                const int p = s + ti * 16 + pm;
                const bool on = p < t;
                jm[ti] = on ? col_lv[p] : 0;
                e[ti] = on ? leaky(a_s[jm[ti]] + ad, A.slope) : -INFINITY;
                m = fmaxf(m, e[ti]);
              }
            }
            m = row16_max(m);
            float sum = 0.f;
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) {
              if (ti < tiles) {
                e[ti] = e[ti] > -INFINITY ? expf(e[ti] - m) : 0.f;
                sum += e[ti];
              }
            }
            const float denom = row16_sum(sum) + 1e-16f;
            float4 z1 = make_float4(0.f, 0.f, 0.f, 0.f), z2 = z1;
            auto lv_trip = [&](float au, int ju) {
              const float4 x = *reinterpret_cast<const float4*>(xa + ju * H + fq);
              if (au != 0.f) {
                z1.x = fmaf(au, x.x, z1.x); z1.y = fmaf(au, x.y, z1.y);
                z1.z = fmaf(au, x.z, z1.z); z1.w = fmaf(au, x.w, z1.w);
              }
            };
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) {
              if (ti < tiles) {
                const float al = e[ti] / denom;
                lv_trip(quad_bcast<0>(al), quad_bcast<0>(jm[ti]));
                lv_trip(quad_bcast<1>(al), quad_bcast<1>(jm[ti]));
                lv_trip(quad_bcast<2>(al), quad_bcast<2>(jm[ti]));
                lv_trip(quad_bcast<3>(al), quad_bcast<3>(jm[ti]));
              }
            }
            // virtual -> virtual GCN row: separately rounded products and sums, as in the chunk path
            auto vv_trip = [&](float wu, int ju, int ou) {
              const float4 x = *reinterpret_cast<const float4*>(xva + ju * H + fq);
              if (ou) {
                z2.x = add_rn(z2.x, mul_rn(wu, x.x)); z2.y = add_rn(z2.y, mul_rn(wu, x.y));
                z2.z = add_rn(z2.z, mul_rn(wu, x.z)); z2.w = add_rn(z2.w, mul_rn(wu, x.w));
              }
            };
            for (int ti = 0; ti < tiles2; ++ti) {
              const int p = s2 + ti * 16 + pm;
              const int on = p < t2;
              const int jj = on ? col_vv[p] : 0;
              const float w = on ? mul_rn(dinv_v[jj], di) : 0.f;
              vv_trip(quad_bcast<0>(w), quad_bcast<0>(jj), quad_bcast<0>(on));
              vv_trip(quad_bcast<1>(w), quad_bcast<1>(jj), quad_bcast<1>(on));
              vv_trip(quad_bcast<2>(w), quad_bcast<2>(jj), quad_bcast<2>(on));
              vv_trip(quad_bcast<3>(w), quad_bcast<3>(jj), quad_bcast<3>(on));
            }
            // fold the four slots (every lane ends with the same sum: a + b is commutative bit for bit)
            z1.x = row_ror_add<8>(z1.x); z1.y = row_ror_add<8>(z1.y); z1.z = row_ror_add<8>(z1.z); z1.w = row_ror_add<8>(z1.w);
            z2.x = row_ror_add<8>(z2.x); z2.y = row_ror_add<8>(z2.y); z2.z = row_ror_add<8>(z2.z); z2.w = row_ror_add<8>(z2.w);
            z1.x = row_ror_add<4>(z1.x); z1.y = row_ror_add<4>(z1.y); z1.z = row_ror_add<4>(z1.z); z1.w = row_ror_add<4>(z1.w);
            z2.x = row_ror_add<4>(z2.x); z2.y = row_ror_add<4>(z2.y); z2.z = row_ror_add<4>(z2.z); z2.w = row_ror_add<4>(z2.w);
            const float og = row_matvec16(z1, Ws, r, 0.f);
            const float ov = row_matvec16(z2, Wv, r, 0.f);
            if (vok) xvb[v * H + r] = rnd<TS>(fmaxf((ov + b_vv[r]) + (og + b_gat[r]), 0.f));
          }
          return;
        }
      }
#endif
      for (;;) {
        int ck = 0;
        if (lane == 0) ck = __hip_atomic_fetch_add(ck_next, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        ck = __builtin_amdgcn_readfirstlane(ck);
        if (ck >= nck) break;
        const int code = ck_tab[ck];
        const int v = code >> 8, c = code & 255;
        const int s = rowptr_lv[v], t = rowptr_lv[v + 1];
        const float ad = a_d[v];
        const int cs = s + c * 64, ce = (cs + 64 < t) ? cs + 64 : t;   // this wave's members
        const int p_own = cs + lane;
        const bool on = p_own < ce;
        const float e_own = on ? leaky(a_s[col_lv[p_own]] + ad, A.slope) : -INFINITY;
        float m, denom;
        if (t - s <= 64) {
          m = wave_max_dpp(e_own);
          denom = wave_sum_dpp(on ? expf(e_own - m) : 0.f) + 1e-16f;
        } else {
          m = -INFINITY;
          for (int p = s + lane; p < t; p += 64) m = fmaxf(m, leaky(a_s[col_lv[p]] + ad, A.slope));
          m = wave_max_dpp(m);
          float sum = 0.f;
          for (int p = s + lane; p < t; p += 64) sum += expf(leaky(a_s[col_lv[p]] + ad, A.slope) - m);
          denom = wave_sum_dpp(sum) + 1e-16f;
        }
        if (on) sc[p_own] = expf(e_own - m) / denom;
        // LDS executes a wave's operations in order, so the slot loop below sees the alphas other lanes wrote --
        // provided the COMPILER keeps the order too: a wavefront-scope fence pins it (it costs a waitcnt)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        // Four members per slot per trip: the
        // index, alpha and row reads of a trip are independent.
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int p0 = cs + slot; p0 < ce; p0 += 4 * S) {
          int jj[4];
          float al[4];
          float4 hh[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int p = p0 + u * S;
            const bool ok = p < ce;
            jj[u] = ok ? col_lv[p] : 0;
            al[u] = ok ? sc[p] : 0.f;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) hh[u] = *reinterpret_cast<const float4*>(xa + jj[u] * H + f);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (p0 + u * S < ce) {
              acc.x = fmaf(al[u], hh[u].x, acc.x);
              acc.y = fmaf(al[u], hh[u].y, acc.y);
              acc.z = fmaf(al[u], hh[u].z, acc.z);
              acc.w = fmaf(al[u], hh[u].w, acc.w);
            }
          }
        }
        // fold the slots: across DPP rows through bpermute, inside a row through DPP rotations
#pragma unroll
        for (int off = 32; off >= 16 && off >= LPR; off >>= 1) {
          acc.x += __shfl_xor(acc.x, off, 64);
          acc.y += __shfl_xor(acc.y, off, 64);
          acc.z += __shfl_xor(acc.z, off, 64);
          acc.w += __shfl_xor(acc.w, off, 64);
        }
        if (LPR <= 8) {
          acc.x = row_ror_add<8>(acc.x); acc.y = row_ror_add<8>(acc.y);
          acc.z = row_ror_add<8>(acc.z); acc.w = row_ror_add<8>(acc.w);
        }
        if (LPR <= 4) {
          acc.x = row_ror_add<4>(acc.x); acc.y = row_ror_add<4>(acc.y);
          acc.z = row_ror_add<4>(acc.z); acc.w = row_ror_add<4>(acc.w);
        }
        const int first = ck_first[v], cntv = ck_first[v + 1] - first;
        if (slot == 0) *reinterpret_cast<float4*>(gpart + (size_t)ck * H + f) = acc;
        int arrived = 0;
        if (lane == 0)
          arrived = __hip_atomic_fetch_add(&ck_arrive[v], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
        arrived = __builtin_amdgcn_readfirstlane(arrived);
        if (arrived != cntv - 1) continue;
        // ---- last chunk of cluster v: finish row v ----
        if (lane == 0) ck_arrive[v] = 0;   // ready for the next layer
        float4 z1r = make_float4(0.f, 0.f, 0.f, 0.f), z2r = z1r;
        // virtual -> virtual GCN row v.  A row of more than eight edges (K = 32 clusters of a trained assignment: up to
        // 32) is dealt to ALL slots of the wave -- slot s takes edges s, s + S, ... -- and the slots' partial rows are
        // folded like the lv partials above (one trip instead of eight dependent ones on four lanes); a short row
        // (the untrained assignment's ~3 clusters) stays on slot 0 in edge order.  Either way separately rounded
        // products and sums; the grouping of the long rows' sums differs from the reference's edge order by rounding.
        const int s2 = rowptr_vv[v], t2 = rowptr_vv[v + 1];
        const float di = dinv_v[v];
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool wide = t2 - s2 > 8;
        if (wide) {
          for (int p = s2 + slot; p < t2; p += S) {
            const int j = col_vv[p];
            const float w = mul_rn(dinv_v[j], di);
            const float4 x = *reinterpret_cast<const float4*>(xva + j * H + f);
            a.x = add_rn(a.x, mul_rn(w, x.x)); a.y = add_rn(a.y, mul_rn(w, x.y));
            a.z = add_rn(a.z, mul_rn(w, x.z)); a.w = add_rn(a.w, mul_rn(w, x.w));
          }
#pragma unroll
          for (int off = 32; off >= 16 && off >= LPR; off >>= 1) {
            a.x += __shfl_xor(a.x, off, 64); a.y += __shfl_xor(a.y, off, 64);
            a.z += __shfl_xor(a.z, off, 64); a.w += __shfl_xor(a.w, off, 64);
          }
          if (LPR <= 8) {
            a.x = row_ror_add<8>(a.x); a.y = row_ror_add<8>(a.y); a.z = row_ror_add<8>(a.z); a.w = row_ror_add<8>(a.w);
          }
          if (LPR <= 4) {
            a.x = row_ror_add<4>(a.x); a.y = row_ror_add<4>(a.y); a.z = row_ror_add<4>(a.z); a.w = row_ror_add<4>(a.w);
          }
        }
        if (slot == 0) {
          if (!wide) {
          for (int p0 = s2; p0 < t2; p0 += 4) {
            int jj[4];
            float ww[4];
            float4 xx[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) jj[u] = (p0 + u < t2) ? col_vv[p0 + u] : 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              ww[u] = mul_rn(dinv_v[jj[u]], di);
              xx[u] = *reinterpret_cast<const float4*>(xva + jj[u] * H + f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              if (p0 + u < t2) {
                a.x = add_rn(a.x, mul_rn(ww[u], xx[u].x));
                a.y = add_rn(a.y, mul_rn(ww[u], xx[u].y));
                a.z = add_rn(a.z, mul_rn(ww[u], xx[u].z));
                a.w = add_rn(a.w, mul_rn(ww[u], xx[u].w));
              }
            }
          }
          }
          float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
          for (int c2 = 0; c2 < cntv; ++c2) {
            const float4 q = *reinterpret_cast<const float4*>(gpart + (size_t)(first + c2) * H + f);
            g.x += q.x; g.y += q.y; g.z += q.z; g.w += q.w;
          }
          // the two aggregated input rows of cluster v go through this wave's H-word scratch one after the other
          // (LDS executes a wave's operations in order: the lanes below read what these lanes wrote, and the
          // second row lands after the first has been read)
          z1r = g;                                         // z1 = sum_i alpha_i x_i
          z2r = a;                                         // z2 = sum_u norm_uv xv_u
        }
        {
          float* zw = zs + wave * H;
          const float* Ws = W + H * H;                     // Wt_src[k][o]
          const float* Wv = W + 3 * H * H;                 // Wt_vv[k][o]
          float og[(H + 63) / 64], ov_[(H + 63) / 64];
#pragma unroll
          for (int pass = 0; pass < 2; ++pass) {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // (the previous pass's reads are done)
            if (slot == 0) *reinterpret_cast<float4*>(zw + f) = pass == 0 ? z1r : z2r;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // the row is in LDS before any lane reads it
            const float* Wm = pass == 0 ? Ws : Wv;
#pragma unroll
            for (int oi = 0; oi < (H + 63) / 64; ++oi) {
              const int o = lane + 64 * oi;
              float acc_ = 0.f;
              if (o < H) {
#pragma unroll
                for (int k4 = 0; k4 < H / 4; ++k4) {
                  const float4 z = *reinterpret_cast<const float4*>(zw + 4 * k4);
                  acc_ = fmaf(z.x, Wm[(4 * k4 + 0) * H + o], acc_);
                  acc_ = fmaf(z.y, Wm[(4 * k4 + 1) * H + o], acc_);
                  acc_ = fmaf(z.z, Wm[(4 * k4 + 2) * H + o], acc_);
                  acc_ = fmaf(z.w, Wm[(4 * k4 + 3) * H + o], acc_);
                }
              }
              if (pass == 0) og[oi] = acc_; else ov_[oi] = acc_;
            }
          }
#pragma unroll
          for (int oi = 0; oi < (H + 63) / 64; ++oi) {
            const int o = lane + 64 * oi;
            if (o < H) xvb[v * H + o] = rnd<TS>(fmaxf((ov_[oi] + b_vv[o]) + (og[oi] + b_gat[o]), 0.f));
          }
        }
      }
    };
    // (the two wave groups may only run side by side while the local layer leaves xa alone until the layer ends:
    // group B gathers input rows from it in phase 2.  The unfused H = 64 local path rewrites xa in its second
    // phase, so there the branches take turns)
    if (cv && a_spec && (FUSE || vonly)) {
      // two barriers per layer: the ll path (group A) and the virtual branch (group B) side by side
      // (virtual-only launch: group A streams the next layer's local rows into the idle transform
      // buffer while group B works, the two buffers swap roles at the end of the layer)
      // Each wave group walks its own copy of the two-barrier sequence: what one group keeps in
      // registers across the barrier (the streamed rows) is not live in the other group's code.
      if (inB) {
        transforms_virtual(GB);
        STAMP_T(40 + 4 * l, NA * 64);        // group B done with its transforms
        if (more && DB) ws.store(Wn);
        lds_barrier();
        reduce_virtual(GB);
        STAMP_T(42 + 4 * l, NA * 64);        // group B (its first wave) done with its reduce
        lds_barrier();
      } else if (vonly) {
        // request up to PF float4 words per thread before the barrier, store them to LDS after it:
        // the rows travel under group B's transforms (inline, not a lambda: they stay in registers)
        constexpr int PF = 8;
        float pf[PF][4];
        // one-launch step: a_{l+1} comes from the local workgroup of THIS launch -- every loading wave polls the
        // graph's publish counter itself, then loads with sc1 buffer loads (bounded; see resident_step.h)
        const bool hand = more && A.ready != nullptr;
        const TS* nsrc = acts_g + ((size_t)l * A.N + n0) * H;
        const int ncnt = more ? n * (H / 4) : 0;
        const __amdgpu_buffer_rsrc_t nrs = rsrc_of<TS>(nsrc, ncnt * 4);
        if (hand) {
          // (the rows are fetched in phase 2, beside group B's long reduce: polling here -- the local workgroup
          // raises the counter about a layer after this point -- held the phase barrier for ~5 k cycles)
#pragma unroll
          for (int u = 0; u < PF; ++u) pf[u][0] = pf[u][1] = pf[u][2] = pf[u][3] = 0.f;
        } else {
#pragma unroll
          for (int u = 0; u < PF; ++u) {
            const int i = GA.t + u * GA.nt;
            const float4 q = ldf4(nsrc, i < ncnt ? i : 0);
            pf[u][0] = q.x; pf[u][1] = q.y; pf[u][2] = q.z; pf[u][3] = q.w;
          }
        }
        STAMP_T(41 + 4 * l, 0);              // group A done with its phase-1 work
        if (more && DB) ws.store(Wn);
        lds_barrier();
        STAMP(5 + 4 * l);
        float4* ndst = reinterpret_cast<float4*>(bh);
        const bool sc1l = hand && !a_acq;
        if (hand && a_acq) {
          wait_published<true>(A.ready + g, A.epoch[0] * 8u + (uint32_t)(l + 1), A.flag);
#pragma unroll
          for (int u = 0; u < PF; ++u) {
            const int i = GA.t + u * GA.nt;
            const float4 q = ldf4(nsrc, i < ncnt ? i : 0);
            pf[u][0] = q.x; pf[u][1] = q.y; pf[u][2] = q.z; pf[u][3] = q.w;
          }
        } else if (hand) {
          wait_published<false>(A.ready + g, A.epoch[0] * 8u + (uint32_t)(l + 1), A.flag);
#pragma unroll
          for (int u = 0; u < PF; ++u) {
            const int i = GA.t + u * GA.nt;
            const float4 q = ldf4_sc1<TS>(nrs, i < ncnt ? i : 0);
            pf[u][0] = q.x; pf[u][1] = q.y; pf[u][2] = q.z; pf[u][3] = q.w;
          }
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) {
          const int i = GA.t + u * GA.nt;
          if (i < ncnt) ndst[i] = make_float4(pf[u][0], pf[u][1], pf[u][2], pf[u][3]);
        }
        for (int i = GA.t + PF * GA.nt; i < ncnt; i += GA.nt) ndst[i] = sc1l ? ldf4_sc1<TS>(nrs, i) : ldf4(nsrc, i);
        reduce_virtual(GA);                  // (then help with whatever chunks are left)
        STAMP_T(43 + 4 * l, 0);              // group A done with its phase-2 work
        lds_barrier();
      } else {
        if (FUSE) layer_ll(GA); else transforms_ll(GA);
        STAMP_T(41 + 4 * l, 0);
        if (more && DB) ws.store(Wn);
        lds_barrier();
        STAMP(5 + 4 * l);
        if (!FUSE) reduce_ll(GA);
        STAMP_T(43 + 4 * l, 0);
        lds_barrier();
      }
      if ((vonly && more) || (FUSE && !vonly)) { float* t_ = xa; xa = bh; bh = t_; }
    } else {
      // one n x H transform buffer: the virtual branch first, then the ll path
      if (cv) {
        transforms_virtual(ALL);
        lds_barrier();
        reduce_virtual(ALL);
        lds_barrier();
      }
      if (!vonly) { if (FUSE) layer_ll(ALL); else transforms_ll(ALL); }
      STAMP_T(41 + 4 * l, 0);
      if (more && DB) ws.store(Wn);
      STAMP_T(40 + 4 * l, 0);
      lds_barrier();
      STAMP(5 + 4 * l);
      if (vonly || !FUSE) {
        if (vonly && more && A.ready) wait_published<true>(A.ready + g, A.epoch[0] * 8u + (uint32_t)(l + 1), A.flag);
        if (vonly) load_next_local(ALL, xa); else reduce_ll(ALL);
        STAMP_T(43 + 4 * l, 0);
        lds_barrier();
      } else {   // fused local layer: its output sits in bh
        float* t_ = xa; xa = bh; bh = t_;
      }
    }
    if (more && !DB) {  // single weight buffer: everybody is done with it now
      ws.fetch(layer_of(A, l + 1), !vonly, cv, H);
      ws.store(Wn);
      lds_barrier();
    }
    STAMP(6 + 4 * l);
    if (cv) {  // swap virtual buffers
      float* t_ = xva; xva = xvb; xvb = t_;
    }
  }

  if (vonly && l_end < A.L) {
    // first part of a split virtual branch: hand the state to the part that resumes at l_end
    for (int idx = threadIdx.x; idx < nv * H; idx += RT) stf(vsxv_g, (size_t)v0 * H + idx, xva[idx]);
    for (int i = threadIdx.x; i <= nv; i += RT) {
      A.vs_rowptr_lv[(size_t)v0 + g + i] = rowptr_lv[i];
      A.vs_rowptr_vv[(size_t)v0 + g + i] = rowptr_vv[i];
    }
    const int c_lv = rowptr_lv[nv], c_vv = rowptr_vv[nv];
    for (int i = threadIdx.x; i < c_lv; i += RT) A.vs_col_lv[(size_t)el0 + i] = col_lv[i];
    for (int i = threadIdx.x; i < c_vv; i += RT) A.vs_col_vv[(size_t)ev0 + i] = col_vv[i];
    for (int i = threadIdx.x; i < nv; i += RT) A.vs_dinv_v[(size_t)v0 + i] = dinv_v[i];
  } else if (cv && A.xv_out) {
    for (int idx = threadIdx.x; idx < nv * H; idx += RT) stf(xvout_g, (size_t)v0 * H + idx, xva[idx]);
  }

  STAMP(62);
  if (vonly) return;  // the prediction belongs to the local launch
  // ---- global_mean_pool: every wave sums a strided row set, wave 0 folds in wave order -----------
  {
    constexpr int LPR = H / 4 > 64 ? 64 : H / 4;
    constexpr int S = 64 / LPR;
    const int lane = threadIdx.x & 63, slot = lane / LPR, f = (lane % LPR) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = wave * S + slot; i < n; i += NW * S) {
      const float4 v = *reinterpret_cast<const float4*>(xa + i * H + f);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
#pragma unroll
    for (int off = 32; off >= LPR; off >>= 1) {
      acc.x += __shfl_xor(acc.x, off, 64);
      acc.y += __shfl_xor(acc.y, off, 64);
      acc.z += __shfl_xor(acc.z, off, 64);
      acc.w += __shfl_xor(acc.w, off, 64);
    }
    if (slot == 0) *reinterpret_cast<float4*>(part + wave * H + f) = acc;
  }
  lds_barrier();
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    float* pooled = vec;
    float* zz = vec + 64;
    const float cnt = (float)(n > 0 ? n : 1);
    if (lane < H) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += part[w * H + lane];
      s = s / cnt;
      pooled[lane] = s;
      A.pooled[(size_t)g * H + lane] = s;
    }
    // one wave: its LDS writes are visible to its own later reads (wavefront fences pin the order for the compiler)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    if (lane < H) {
      float a1 = 0.f;
      const float4* wr = reinterpret_cast<const float4*>(headw + lane * H);
#pragma unroll
      for (int k4 = 0; k4 < H / 4; ++k4) {
        const float4 w4 = wr[k4];
        const float4 p4 = *reinterpret_cast<const float4*>(pooled + 4 * k4);
        a1 = fmaf(p4.x, w4.x, a1);
        a1 = fmaf(p4.y, w4.y, a1);
        a1 = fmaf(p4.z, w4.z, a1);
        a1 = fmaf(p4.w, w4.w, a1);
      }
      a1 += headw[H * H + lane];
      a1 = apply_act(a1, A.head_act);
      zz[lane] = a1;
      A.z[(size_t)g * H + lane] = a1;
    }
    const float* W2l = headw + H * H + H;
    const float* b2l = W2l + A.C * H;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // zz (written by lanes < H just above) before it is read
    for (int c = lane; c < A.C; c += 64) {
      float a2 = 0.f;
      const float4* wr = reinterpret_cast<const float4*>(W2l + c * H);
#pragma unroll
      for (int k4 = 0; k4 < H / 4; ++k4) {
        const float4 w4 = wr[k4];
        const float4 z4 = *reinterpret_cast<const float4*>(zz + 4 * k4);
        a2 = fmaf(z4.x, w4.x, a2);
        a2 = fmaf(z4.y, w4.y, a2);
        a2 = fmaf(z4.z, w4.z, a2);
        a2 = fmaf(z4.w, w4.w, a2);
      }
      const float pc = a2 + b2l[c];
      A.pred[(size_t)g * A.C + c] = pc;
      if (A.score) A.score[(size_t)g * A.C + c] = 1.0f / (1.0f + expf(-pc));   // = criterion_elem's sg
    }
  }
  STAMP(63);
}

template <int H, int RT, int MODE, typename TS>
__global__ void __launch_bounds__(RT) k_hscn_fwd(const FwdArgs A) {
  hscn_fwd_body<H, RT, MODE, TS>(A, blockIdx.x);
}

// One launch, two kinds of workgroup: even blocks run the local chain + head of graph g, odd blocks
// the part of the virtual branch that does not need them (its CSRs and layer 0, which reads the
// input features); see k_hscn_bwd_virtual for the rest.
struct FwdPair {
  FwdArgs a[2];  // [0] local chain, [1] virtual part
};
template <int H, int RT, typename TS>
__global__ void __launch_bounds__(RT) k_hscn_fwd_pair(const FwdPair P) {
  // one copy of the body, the argument block (in the kernarg segment) chosen by the parity of the workgroup
  if (blockIdx.x & 1) hscn_fwd_body<H, RT, 2, TS>(P.a[1], blockIdx.x >> 1);
  else hscn_fwd_body<H, RT, 1, TS>(P.a[0], blockIdx.x >> 1);
}

// =============================== backward =====================================================
struct BwdLayout {
  size_t G, GH, X, dinv, vec, red, bred, wl, headw, rowptr_t, col_t, total;
};
__host__ __device__ inline BwdLayout bwd_layout(int H, int C, int max_n, int max_ell, int two) {
  BwdLayout Y;
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o += (n + 3) & ~(size_t)3; return r; };
  Y.G = take((size_t)max_n * H);
  Y.GH = take((size_t)max_n * H);
  Y.X = two ? Y.G : take((size_t)max_n * H);   // two-buffer mode: the layer input takes over the buffer of the dead gradient
  Y.dinv = take(max_n);
  Y.vec = take(256 + 64);
  Y.red = take((size_t)(RT_MAX / 64) * 256);  // weight gradient: one 16 x 16 partial tile per wave
  Y.bred = take((size_t)(RT_MAX / 64) * H);    // bias gradient: one H-vector per wave
  Y.wl = take((size_t)H * H);
  Y.headw = take((size_t)H * H + (size_t)C * H + 2 * (size_t)C);  // W1 | W2 | g_pred row | loss terms
  Y.rowptr_t = take(max_n + 1);
  Y.col_t = take(max_ell);
  Y.total = o;
  return Y;
}

template <int H, int RT, typename TS = float>
__device__ __forceinline__ void hscn_bwd_body(const BwdArgs& A, const int g) {
  extern __shared__ __align__(16) unsigned char smem[];
  const TS* const xl_g = reinterpret_cast<const TS*>(A.x_local);
  const TS* const acts_g = reinterpret_cast<const TS*>(A.acts);
  constexpr int OPT = Blk<H>::OPT;
  constexpr int NW = RT / 64;
    const int n0 = A.lptr[g], n = A.lptr[g + 1] - n0;
  const int e0 = A.eptr_ll[g], ne = A.eptr_ll[g + 1] - e0;
  float* part = A.partials + (size_t)g * A.P;
  if (n > A.max_n || ne > A.max_ell || n < 0) {
    if (threadIdx.x == 0 && A.flag) atomicOr(A.flag, 4);
    for (int i = threadIdx.x; i < A.P; i += RT) part[i] = 0.f;
    return;
  }
  const BwdLayout Y = bwd_layout(H, A.C, A.max_n, A.max_ell, A.two);
  float* fb = reinterpret_cast<float*>(smem);
  int* ib = reinterpret_cast<int*>(smem);
  // three buffers: G (gradient of a layer's output), GH (= A_hat^T G), X (the layer's input).  With two,
  // X moves into G's buffer once the gather-reduce has consumed G (one more barrier), the input gradient
  // overwrites GH in place (row tiles are independent) and the two buffers swap roles every layer.
  const bool two = A.two != 0;
  float *G = fb + Y.G, *GH = fb + Y.GH, *X = fb + Y.X, *dinv = fb + Y.dinv, *vec = fb + Y.vec;
  float *red = fb + Y.red, *bred = fb + Y.bred, *wl = fb + Y.wl, *headw = fb + Y.headw;
  int *rowptr_t = ib + Y.rowptr_t, *col_t = ib + Y.col_t;
  const int L = A.L;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;   // (scalar: "tiles of this wave" loops stay uniform)
  const Grp ALL{(int)threadIdx.x, RT, wave, NW};
  (void)ib;

  // ---- prologue: request all global inputs of the first phases at once ------------------------
  STAMP(0);
  float* W1l = headw;
  float* W2l = headw + H * H;
  float* gpl = W2l + A.C * H;
  float* zz = vec + 64;      // z
  float* gz = vec + 128;     // dL/d(lin_1 output, pre-activation)
  float* gpool = vec + 192;
  float* pol = vec + 256;    // pooled
  constexpr int EPT = 2, XPT = 8, RPT = 2;
  // everything the launch needs from HBM up front, clamped addresses, no use before the parking
  // stores below: source-keyed CSR + degree norm (exported by the forward launch), the last layer's
  // output, head weights, upstream gradient
  int cr[EPT], rr[RPT];
  float yr[XPT], dr[RPT], hw0, hw1, ty0 = 0.f, ty1 = 0.f, zv, pv;
  const int HT = H * H + A.C * H + A.C;
  auto haddr = [&](int idx) -> const float* {
    if (idx < H * H) return A.W1 + idx;
    idx -= H * H;
    if (idx < A.C * H) return A.W2 + idx;
    idx -= A.C * H;
    return (A.target ? A.pred : A.g_pred) + (size_t)g * A.C + (idx < A.C ? idx : 0);
  };
  // with a loss tail the last C words are computed from (pred, target): the target word travels with its pred word
  auto taddr = [&](int idx) -> const float* {
    idx -= HT - A.C;
    return A.target + (size_t)g * A.C + ((idx >= 0 && idx < A.C) ? idx : 0);
  };
  const int32_t* rpt = A.csr_rowptr_t + (size_t)n0 + g;
  const int32_t* cpt = A.csr_col_t + (size_t)e0;
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int idx = threadIdx.x + i * RT;
    rr[i] = rpt[idx <= n ? idx : 0];
    dr[i] = A.dinv_in[(size_t)n0 + (idx < n ? idx : 0)];
  }
#pragma unroll
  for (int i = 0; i < EPT; ++i) {
    const int e = threadIdx.x + i * RT;
    cr[i] = cpt[(e < ne && A.csr_col_t) ? e : 0];
  }
  const TS* yL = acts_g + ((size_t)(L - 1) * A.N + n0) * H;
#pragma unroll
  for (int i = 0; i < XPT; ++i) {
    const int idx = threadIdx.x + i * RT;
    yr[i] = ldf(yL, idx < n * H ? idx : 0);
  }
  const float gs = A.g_scale ? A.g_scale[0] : 1.f;
  const bool scaled = A.g_scale != nullptr;
  // (the last C words are the upstream gradient row: the loss node's scalar factor is applied here)
  const bool tail = A.target != nullptr;
  auto hval = [&](int idx, float v, float ty) {
    if (idx < HT - A.C) return v;
    if (tail) {   // v = pred, ty = target  ->  loss term (kept for the column sum) and d(mean loss)/dpred
      float l, sg;
      criterion_elem(A.loss_kind, v, ty, A.inv_count, l, sg, v);
      headw[idx + A.C] = l;
    }
    return scaled ? gs * v : v;
  };
  hw0 = *haddr((int)threadIdx.x < HT ? (int)threadIdx.x : 0);
  hw1 = *haddr((int)threadIdx.x + RT < HT ? (int)threadIdx.x + RT : 0);
  if (tail) {
    ty0 = *taddr((int)threadIdx.x < HT ? (int)threadIdx.x : 0);
    ty1 = *taddr((int)threadIdx.x + RT < HT ? (int)threadIdx.x + RT : 0);
  }
  for (int idx = threadIdx.x + 2 * RT; idx < HT; idx += RT) headw[idx] = hval(idx, *haddr(idx), tail ? *taddr(idx) : 0.f);
  zv = A.z[(size_t)g * H + (threadIdx.x < H ? threadIdx.x : 0)];
  pv = A.pooled[(size_t)g * H + (threadIdx.x < H ? threadIdx.x : 0)];
  // ---- park in LDS ----------------------------------------------------------------------------
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int idx = threadIdx.x + i * RT;
    if (idx <= n) rowptr_t[idx] = rr[i];
    if (idx < n) dinv[idx] = dr[i];
  }
  for (int idx = threadIdx.x + RPT * RT; idx <= n; idx += RT) rowptr_t[idx] = rpt[idx];
  for (int idx = threadIdx.x + RPT * RT; idx < n; idx += RT) dinv[idx] = A.dinv_in[(size_t)n0 + idx];
#pragma unroll
  for (int i = 0; i < EPT; ++i) {
    const int e = threadIdx.x + i * RT;
    if (e < ne) col_t[e] = cr[i];
  }
  for (int e = threadIdx.x + EPT * RT; e < ne; e += RT) col_t[e] = cpt[e];
  // last layer's output (ReLU mask source) -> X; it is also the next layer's input further down
#pragma unroll
  for (int i = 0; i < XPT; ++i) {
    const int idx = threadIdx.x + i * RT;
    if (idx < n * H) X[idx] = yr[i];
  }
  for (int idx = threadIdx.x + XPT * RT; idx < n * H; idx += RT) X[idx] = ldf(yL, idx);
  if ((int)threadIdx.x < HT) headw[threadIdx.x] = hval((int)threadIdx.x, hw0, ty0);
  if ((int)threadIdx.x + RT < HT) headw[threadIdx.x + RT] = hval((int)threadIdx.x + RT, hw1, ty1);
  if (threadIdx.x < H) {
    zz[threadIdx.x] = zv;
    pol[threadIdx.x] = pv;
  }
  lds_barrier();
  STAMP(1);
  STAMP(2);

  // ---- head backward ---------------------------------------------------------------------------
  // partial layout: per layer {W_ll [H*fin], b_ll [H]}, then W1 [H*H], b1 [H], W2 [C*H], b2 [C]
  int off_head = 0;
  for (int l = 0; l < L; ++l) off_head += H * (l == 0 ? A.F : H) + H;
  const int oW1 = off_head, ob1 = oW1 + H * H, oW2 = ob1 + H, ob2 = oW2 + A.C * H;
  if (threadIdx.x < H) {
    // g_zpre[k] = (sum_c g_pred[c] W2[c][k]) * act'(z[k])
    float acc = 0.f;
    for (int c = 0; c < A.C; ++c) acc = fmaf(gpl[c], W2l[c * H + threadIdx.x], acc);
    gz[threadIdx.x] = acc * act_grad_from_output(zz[threadIdx.x], A.head_act);
  }
  lds_barrier();
  for (int idx = threadIdx.x; idx < A.C * H; idx += RT) {
    const int c = idx / H, k = idx - c * H;
    part[oW2 + idx] = gpl[c] * zz[k];
  }
  for (int c = threadIdx.x; c < A.C; c += RT) part[ob2 + c] = gpl[c];
  if (tail && threadIdx.x == RT - 64) {   // the graph's loss terms, summed in class order (a wave off the head's critical path)
    float sl = 0.f;
    for (int c = 0; c < A.C; ++c) sl += headw[HT + c];
    part[A.Pn] = sl;
  }
  for (int idx = threadIdx.x; idx < H * H; idx += RT) {
    const int o = idx / H, k = idx - o * H;
    part[oW1 + idx] = gz[o] * pol[k];
  }
  if (threadIdx.x < H) {
    part[ob1 + threadIdx.x] = gz[threadIdx.x];
    float acc = 0.f;
#pragma unroll
    for (int o = 0; o < H; ++o) acc = fmaf(gz[o], W1l[o * H + threadIdx.x], acc);
    gpool[threadIdx.x] = acc;
  }
  lds_barrier();
  // dL/d x_L[i][f] = g_pool[f] / n, masked by ReLU of the saved output (in X)
  {
    const float cnt = (float)(n > 0 ? n : 1);
    for (int idx = threadIdx.x; idx < n * H; idx += RT) G[idx] = X[idx] > 0.f ? gpool[idx % H] / cnt : 0.f;
  }
  STAMP(3);

  // weight-gradient tiles: NW waves x one 16 x 16 partial tile each; when all tiles of a layer fit one
  // pass (GW1) the fold of the partial tiles is deferred past the next workgroup barrier that is
  // there anyway (the top of the next layer, or the one after the loop)
  constexpr int GW_TD = H / 16, GW_NT = GW_TD * GW_TD;
  constexpr int GW_TPP = GW_NT < NW ? GW_NT : NW, GW_RG = NW / GW_TPP;
  constexpr bool GW1 = GW_NT <= NW;
  auto fold_gw = [&](int t0, int oW_, int fin_) {
    for (int idx = threadIdx.x; idx < GW_TPP * 256; idx += RT) {
      const int t_ = t0 + idx / 256, e_ = idx & 255;
      if (t_ < GW_NT) {
        float s_ = 0.f;
#pragma unroll
        for (int r = 0; r < GW_RG; ++r) s_ += red[(r * GW_TPP + idx / 256) * 256 + e_];
        const int oo = (t_ / GW_TD) * 16 + (e_ >> 4), kk = (t_ % GW_TD) * 16 + (e_ & 15);
        if (kk < fin_) part[oW_ + oo * fin_ + kk] = s_;
      }
    }
  };
  int pend_oW = -1, pend_fin = 0;
  int off = off_head;
  for (int l = L - 1; l >= 0; --l) {
    const int fin = l == 0 ? A.F : H;
    off -= H * fin + H;
    const int oW = off, ob = off + H * fin;
    lds_barrier();  // G (masked) complete; X free to be overwritten
    if (GW1 && pend_oW >= 0) fold_gw(0, pend_oW, pend_fin);   // the previous layer's weight gradient
    // layer input -> X (zero padded) and this layer's W_ll -> LDS.  The loads are issued first and
    // parked in LDS after the gather-reduce: their HBM latency hides under it.
    float xr[XPT], wr_[(H * H + RT - 1) / RT];
    const TS* xin = l == 0 ? xl_g + (size_t)n0 * fin : acts_g + ((size_t)(l - 1) * A.N + n0) * H;
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int idx = threadIdx.x + i * RT;
      const int r = idx / H, k = idx - r * H;
      xr[i] = (idx < n * H && k < fin) ? ldf(xin, (size_t)r * fin + k) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < (H * H + RT - 1) / RT; ++i) {
      const int idx = threadIdx.x + i * RT;
      wr_[i] = (l > 0 && idx < H * H) ? A.W_ll[l][idx] : 0.f;
    }
    // bias gradient = column sums of G: a lane adds float4 pieces of a strided row set (consecutive
    // lanes read consecutive 16 B: no bank conflicts), the slots of a wave fold through DPP /
    // bpermute, the waves through LDS after the phase barrier (in wave order)
    {
      constexpr int LQ = H / 4 > 64 ? 64 : H / 4;
      constexpr int SQ = 64 / LQ;
      const int slot = lane / LQ, f = (lane % LQ) * 4;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int i = wave * SQ + slot; i < n; i += NW * SQ) {
        const float4 v = *reinterpret_cast<const float4*>(G + i * H + f);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
#pragma unroll
      for (int o_ = 32; o_ >= 16 && o_ >= LQ; o_ >>= 1) {
        acc.x += __shfl_xor(acc.x, o_, 64);
        acc.y += __shfl_xor(acc.y, o_, 64);
        acc.z += __shfl_xor(acc.z, o_, 64);
        acc.w += __shfl_xor(acc.w, o_, 64);
      }
      if (LQ <= 8) {
        acc.x = row_ror_add<8>(acc.x); acc.y = row_ror_add<8>(acc.y);
        acc.z = row_ror_add<8>(acc.z); acc.w = row_ror_add<8>(acc.w);
      }
      if (LQ <= 4) {
        acc.x = row_ror_add<4>(acc.x); acc.y = row_ror_add<4>(acc.y);
        acc.z = row_ror_add<4>(acc.z); acc.w = row_ror_add<4>(acc.w);
      }
      if (slot == 0) *reinterpret_cast<float4*>(bred + wave * H + f) = acc;
    }
    // dL/d(transform output) = A_hat^T G  (transposed CSR, edge order)
    agg_gcn_lds<H, float>(rowptr_t, col_t, dinv, dinv, G, nullptr, GH, n, 0, (float*)nullptr, ALL);
    if (two) {          // every wave is done with G before the layer input lands in its buffer
      lds_barrier();
      X = G;
    }
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int idx = threadIdx.x + i * RT;
      if (idx < n * H) X[idx] = xr[i];
    }
    for (int idx = threadIdx.x + XPT * RT; idx < n * H; idx += RT) {
      const int r = idx / H, k = idx - r * H;
      X[idx] = k < fin ? ldf(xin, (size_t)r * fin + k) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < (H * H + RT - 1) / RT; ++i) {
      const int idx = threadIdx.x + i * RT;
      if (idx < H * H) wl[idx] = wr_[i];
    }
    lds_barrier();
    STAMP(4 + 4 * l);
    if (threadIdx.x < H) {   // bias gradient: fold the waves' column sums in wave order
      float sb = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) sb += bred[w * H + threadIdx.x];
      part[ob + threadIdx.x] = sb;
    }
    // weight gradient gW[o][k] = sum_j GH[j][o] * X[j][k] = GH^T X on the matrix cores
    // (v_mfma_f32_16x16x4_f32, fp32 in and out): a wave owns one 16 x 16 tile (o, k) and a strided
    // set of 4-row chunks of j; A[o][j] and B[j][k] are single LDS words per lane, consecutive lanes
    // on consecutive addresses.  The waves that share a tile fold their partial tiles through LDS
    // in a fixed order.
    {
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      constexpr int TD = GW_TD, NT = GW_NT, TPP = GW_TPP, RG = GW_RG;
      const int li = lane & 15, lj = lane >> 4;
      for (int t0 = 0; t0 < NT; t0 += TPP) {
        const int tl = wave % TPP, rg = wave / TPP;
        const int tile = t0 + tl;
        const bool live = tile < NT && rg < RG;
        const int o0 = live ? (tile / TD) * 16 : 0, k0 = live ? (tile % TD) * 16 : 0;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (live) {
          for (int j0 = rg * 4; j0 < n; j0 += 4 * RG) {
            const int j = j0 + lj;
            const bool ok = j < n;
            const float av = ok ? GH[j * H + o0 + li] : 0.f;
            const float bv = ok ? X[j * H + k0 + li] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) red[(rg * TPP + tl) * 256 + (lj * 4 + r) * 16 + li] = acc[r];
        }
        if (GW1) {
          pend_oW = oW;
          pend_fin = fin;
        } else {
          lds_barrier();
          fold_gw(t0, oW, fin);
          if (t0 + TPP < NT) lds_barrier();   // the partial-tile buffer is reused by the next pass
        }
      }
    }
    STAMP(5 + 4 * l);
    // input gradient: G[j][k] = relu'(x_l[j][k]) * sum_o GH[j][o] * W[o][k]   (W is [H][H] here;
    // x_l = this layer's input = previous layer's output, already in X)
    if (l > 0 && H <= 32) {
      lin_mfma<H, true>(GH, wl, two ? GH : G, n, X, ALL);
    } else if (l > 0) {
      constexpr int LPR = H / OPT;
      constexpr int RS = RT / LPR;
      const int kg = threadIdx.x % LPR, r0 = threadIdx.x / LPR;
      const int k0 = kg * OPT;
      if (r0 < n) {
        float w[OPT][H];
#pragma unroll
        for (int o = 0; o < H; ++o)
#pragma unroll
          for (int q = 0; q < OPT; ++q) w[q][o] = wl[o * H + k0 + q];
        for (int j = r0; j < n; j += RS) {
          const float4* gr = reinterpret_cast<const float4*>(GH + j * H);
          float acc[OPT];
#pragma unroll
          for (int q = 0; q < OPT; ++q) acc[q] = 0.f;
#pragma unroll
          for (int o4 = 0; o4 < H / 4; ++o4) {
            const float4 v = gr[o4];
#pragma unroll
            for (int q = 0; q < OPT; ++q) {
              acc[q] = fmaf(v.x, w[q][4 * o4 + 0], acc[q]);
              acc[q] = fmaf(v.y, w[q][4 * o4 + 1], acc[q]);
              acc[q] = fmaf(v.z, w[q][4 * o4 + 2], acc[q]);
              acc[q] = fmaf(v.w, w[q][4 * o4 + 3], acc[q]);
            }
          }
#pragma unroll
          for (int q = 0; q < OPT; ++q) (two ? GH : G)[j * H + k0 + q] = X[j * H + k0 + q] > 0.f ? acc[q] : 0.f;
        }
      }
    }
    STAMP(6 + 4 * l);
    if (two) {   // the input gradient was written over GH: the buffers swap roles
      float* t_ = G;
      G = GH;
      GH = t_;
    }
  }
  if (GW1 && pend_oW >= 0) {   // layer 0's weight gradient
    lds_barrier();
    fold_gw(0, pend_oW, pend_fin);
  }
  STAMP(63);
}

template <int H, int RT, typename TS>
__global__ void __launch_bounds__(RT) k_hscn_bwd(const BwdArgs A) {
  hscn_bwd_body<H, RT, TS>(A, blockIdx.x);
}
// One launch, two kinds of workgroup: even blocks run the backward of graph g, odd blocks the
// virtual branch of the same step's forward (a mode-2 job: it depends on the local launch only and
// nothing depends on it), so the virtual branch fills the CUs a 128-graph batch leaves idle.
template <int H, int RT, typename TS>
__global__ void __launch_bounds__(RT) k_hscn_bwd_virtual(const BwdArgs Ab, const FwdArgs Af) {
  const int g = blockIdx.x >> 1;
  if (blockIdx.x & 1) {
    if (Af.l_begin > 0) hscn_fwd_body<H, RT, 4, TS>(Af, g);
    else hscn_fwd_body<H, RT, 2, TS>(Af, g);
  }
  else hscn_bwd_body<H, RT, TS>(Ab, g);
}

inline size_t fwd_lds_bytes(int H, int C, int max_n, int max_v, int max_ell, int max_evv, int db, int exp) {
  return fwd_layout(H, C, max_n, max_v, max_ell, max_evv, db, exp).total * 4;
}
// double-buffered layer weights whenever the launch still fits a CU's LDS with them (H = 64: never, 64 KB)
inline size_t pick_fwd_lds(FwdArgs& A, int H) {
  A.db = (H <= 32 && fwd_lds_bytes(H, A.C, A.max_n, A.max_v, A.max_ell, A.max_evv, 1, A.exp) <= 160 * 1024) ? 1 : 0;
  return fwd_lds_bytes(H, A.C, A.max_n, A.max_v, A.max_ell, A.max_evv, A.db, A.exp);
}
inline size_t bwd_lds_bytes(int H, int C, int max_n, int max_ell, int two) {
  return bwd_layout(H, C, max_n, max_ell, two).total * 4;
}
// three n x H buffers when they fit, else two (one more barrier per layer)
inline size_t pick_bwd_lds(BwdArgs& A, int H) {
  A.two = 0;
  size_t lds = bwd_lds_bytes(H, A.C, A.max_n, A.max_ell, 0);
  if (lds > 160 * 1024) {
    A.two = 1;
    lds = bwd_lds_bytes(H, A.C, A.max_n, A.max_ell, 1);
  }
  return lds;
}

// Workgroup size: 16 waves (4 per SIMD) hide the LDS / global latency of the many short
// phases; tiny graphs (PCQM-Contact, n <= 64) do not have the rows to feed them.
template <int H, int RT, int MODE, typename TS>
int launch_fwd_mode(const FwdArgs& A, int64_t B, size_t lds, hipStream_t st) {
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)k_hscn_fwd<H, RT, MODE, TS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
  k_hscn_fwd<H, RT, MODE, TS><<<(unsigned)B, RT, lds, st>>>(A);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}
template <int H, int RT, typename TS>
int launch_fwd_rt(const FwdArgs& A, int64_t B, size_t lds, hipStream_t st) {
  // one specialisation of the body per value of compute_virtual (see hscn_fwd_body: MODE)
  if (A.compute_virtual == 0) return launch_fwd_mode<H, RT, 1, TS>(A, B, lds, st);
  if (A.compute_virtual == 2) return launch_fwd_mode<H, RT, 0, TS>(A, B, lds, st);   // (first part or resumed: run time)
  return launch_fwd_mode<H, RT, 3, TS>(A, B, lds, st);
}
// Source-keyed ll CSR + degree norm for the backward launch when the forward launch had no LDS
// left to build them on the side (large graphs): a light kernel of its own, one workgroup per graph.
__global__ void __launch_bounds__(256) k_ll_csr_t(const FwdArgs A) {
  extern __shared__ __align__(16) unsigned char smem[];
  int* ib = reinterpret_cast<int*>(smem);
  const int g = blockIdx.x;
  const int n0 = A.lptr[g], n = A.lptr[g + 1] - n0;
  const int e0 = A.eptr_ll[g], ne = A.eptr_ll[g + 1] - e0;
  if (n > A.max_n || ne > A.max_ell || n < 0 || ne < 0) return;
  int* ek = ib;                       // [max_ell]  key = source
  int* eo = ek + A.max_ell;           // [max_ell]
  int* rowptr_t = eo + A.max_ell;     // [max_n+1]
  int* col_t = rowptr_t + A.max_n + 1;
  int* cursor = col_t + A.max_ell;    // [max_n+1]  (in-degree counts first)
  int* tmp = cursor + A.max_n + 1;
  const Grp ALL{(int)threadIdx.x, 256, (int)threadIdx.x >> 6, 4};
  for (int i = threadIdx.x; i <= n; i += 256) cursor[i] = 0;
  for (int e = threadIdx.x; e < ne; e += 256) {
    int k = (int)(A.ll_src[e0 + e] - n0), o = (int)(A.ll_dst[e0 + e] - n0);
    if (k < 0 || k >= n || o < 0 || o >= n) { k = -1; o = -1; }
    ek[e] = k; eo[e] = o;
  }
  lds_barrier();
  for (int e = threadIdx.x; e < ne; e += 256)
    if (ek[e] >= 0) atomicAdd(&cursor[eo[e]], 1);
  lds_barrier();
  for (int i = threadIdx.x; i < n; i += 256) {
    const int d = cursor[i];
    A.dinv_out[(size_t)n0 + i] = d > 0 ? 1.0f / sqrtf((float)d) : 0.f;
  }
  lds_barrier();
  build_csr_lds(ek, eo, ne, n, rowptr_t, col_t, cursor, tmp, ALL, true);
  for (int i = threadIdx.x; i <= n; i += 256) A.csr_rowptr_t[(size_t)n0 + g + i] = rowptr_t[i];
  const int cnt_t = rowptr_t[n];
  for (int p = threadIdx.x; p < cnt_t; p += 256) A.csr_col_t[(size_t)e0 + p] = col_t[p];
}

// ---- structure of every graph of a block-diagonal hetero batch (or of a whole dataset laid out as one), built once:
// the four stable CSRs (graph-local int32 ids) and the two degree norms that the resident launches otherwise rebuild
// from the COO slices in LDS every step -- with the SAME device functions, so a step that loads them computes bit
// for bit what a step that builds them computes.  One 256-thread workgroup per graph, relations one after another.
struct StructArgs {
  const int64_t *ll_src, *ll_dst, *vv_src, *vv_dst, *lv_src, *lv_dst;
  const int32_t *lptr, *vptr, *eptr_ll, *eptr_vv, *eptr_lv;
  hscn_structure out;
  int32_t* flag;
  int max_n, max_v, max_ell, max_evv;
};
inline size_t struct_lds_words(int max_n, int max_v, int max_ell, int max_evv) {
  const size_t me = (size_t)(max_ell > max_evv ? max_ell : max_evv) > (size_t)max_n ? (size_t)(max_ell > max_evv ? max_ell : max_evv) : (size_t)max_n;
  const size_t mr = (size_t)(max_n > max_v ? max_n : max_v) + 1;
  const size_t nchunk = ((size_t)max_n + 63) / 64;
  size_t cnt = (size_t)max_v * nchunk;
  if (cnt < mr) cnt = mr;
  return 4 * me + 2 * mr + cnt + 64;
}
__global__ void __launch_bounds__(256) k_structure(const StructArgs A) {
  extern __shared__ __align__(16) unsigned char smem[];
  int* ib = reinterpret_cast<int*>(smem);
  const int g = blockIdx.x;
  const int n0 = A.lptr[g], n = A.lptr[g + 1] - n0;
  const int v0 = A.vptr[g], nv = A.vptr[g + 1] - v0;
  const int e0 = A.eptr_ll[g], ne = A.eptr_ll[g + 1] - e0;
  const int ev0 = A.eptr_vv[g], nev = A.eptr_vv[g + 1] - ev0;
  const int el0 = A.eptr_lv[g], nel = A.eptr_lv[g + 1] - el0;
  if ((n > A.max_n) | (nv > A.max_v) | (ne > A.max_ell) | (nev > A.max_evv) | (nel > A.max_n) | (n < 0) | (nv < 0)) {
    if (threadIdx.x == 0 && A.flag) atomicOr(A.flag, 4);
    return;
  }
  const size_t me = (size_t)(A.max_ell > A.max_evv ? A.max_ell : A.max_evv) > (size_t)A.max_n
                        ? (size_t)(A.max_ell > A.max_evv ? A.max_ell : A.max_evv) : (size_t)A.max_n;
  const size_t mr = (size_t)(A.max_n > A.max_v ? A.max_n : A.max_v) + 1;
  int* ek = ib;
  int* eo = ek + me;
  int* colb = eo + me;
  int* tmp = colb + me;
  int* rowptr = tmp + me;
  int* cursor = rowptr + mr;            // also the multisplit's counters
  const size_t nchunk = ((size_t)A.max_n + 63) / 64;
  size_t cnt = (size_t)A.max_v * nchunk;
  if (cnt < mr) cnt = mr;
  int* wsum = cursor + cnt;
  const Grp ALL{(int)threadIdx.x, 256, (int)threadIdx.x >> 6, 4};
  bool bad = false;
  auto stage = [&](const int64_t* kp, const int64_t* op, int base, int cntE, int kb, int ob, int nk, int no) {
    for (int e = threadIdx.x; e < cntE; e += 256) {
      int k = (int)(kp[base + e] - kb), o = (int)(op[base + e] - ob);
      if (k < 0 || k >= nk || o < 0 || o >= no) { bad = true; k = -1; o = -1; }
      ek[e] = k; eo[e] = o;
    }
  };
  auto put = [&](int32_t* rp_out, int32_t* col_out, size_t rbase, size_t cbase, int rows) {
    for (int i = threadIdx.x; i <= rows; i += 256) rp_out[rbase + i] = rowptr[i];
    const int c = rowptr[rows];
    for (int p = threadIdx.x; p < c; p += 256) col_out[cbase + p] = colb[p];
  };
  // local -> local keyed by target (+ the degree norm), then keyed by source
  stage(A.ll_dst, A.ll_src, e0, ne, n0, n0, n, n);
  lds_barrier();
  build_csr_lds(ek, eo, ne, n, rowptr, colb, cursor, tmp, ALL, true);
  put(A.out.ll_rowptr_d, A.out.ll_col_d, (size_t)n0 + g, (size_t)e0, n);
  for (int i = threadIdx.x; i < n; i += 256) {
    const int d = rowptr[i + 1] - rowptr[i];
    A.out.ll_dinv[(size_t)n0 + i] = d > 0 ? 1.0f / sqrtf((float)d) : 0.f;
  }
  lds_barrier();
  build_csr_lds(eo, ek, ne, n, rowptr, colb, cursor, tmp, ALL, true);
  put(A.out.ll_rowptr_s, A.out.ll_col_s, (size_t)n0 + g, (size_t)e0, n);
  lds_barrier();
  // virtual -> virtual keyed by target (+ its degree norm)
  stage(A.vv_dst, A.vv_src, ev0, nev, v0, v0, nv, nv);
  lds_barrier();
  build_csr_lds(ek, eo, nev, nv, rowptr, colb, cursor, tmp, ALL, true);
  put(A.out.vv_rowptr, A.out.vv_col, (size_t)v0 + g, (size_t)ev0, nv);
  for (int i = threadIdx.x; i < nv; i += 256) {
    const int d = rowptr[i + 1] - rowptr[i];
    A.out.vv_dinv[(size_t)v0 + i] = d > 0 ? 1.0f / sqrtf((float)d) : 0.f;
  }
  lds_barrier();
  // local -> virtual keyed by target (rows are clusters: the multisplit)
  stage(A.lv_dst, A.lv_src, el0, nel, v0, n0, nv, n);
  lds_barrier();
  build_csr_multisplit_lds(ek, eo, nel, nv, rowptr, colb, cursor, tmp, wsum, ALL);
  put(A.out.lv_rowptr, A.out.lv_col, (size_t)v0 + g, (size_t)el0, nv);
  if (bad && A.flag) atomicOr(A.flag, 2);
}

template <int H, typename TS>
int launch_fwd(FwdArgs& A, int64_t B, hipStream_t st) {
  // preference order: concurrent wave groups + CSR export, then dropping the third n x H buffer,
  // then dropping the in-launch export (a separate light kernel builds it)
  const bool want_exp = A.csr_rowptr_t != nullptr;
  size_t lds = 0;
  bool ok = false;
  A.spec = A.compute_virtual ? 1 : 0;       // (the side-by-side wave groups cost no LDS any more)
  for (int e = 1; e >= 0 && !ok; --e) {
    A.exp = (want_exp && e) ? 1 : 0;
    lds = pick_fwd_lds(A, H);
    ok = lds <= 160 * 1024;
  }
  if (!ok) return HSCN_E_UNSUPPORTED;
  A.exp_dinv = A.exp;
  static const int rt_env = getenv("HSCN_RT") ? atoi(getenv("HSCN_RT")) : 0;
  int rc;
  if (A.max_n <= 64 || rt_env == 256) rc = launch_fwd_rt<H, 256, TS>(A, B, lds, st);
  else if (rt_env == 512 && sizeof(TS) == 4) rc = launch_fwd_rt<H, 512, float>(A, B, lds, st);
  else rc = launch_fwd_rt<H, 1024, TS>(A, B, lds, st);
  if (rc) return rc;
  if (want_exp && !A.exp) {
    const size_t l2 = ((size_t)4 * A.max_ell + 2 * ((size_t)A.max_n + 1) + 16) * 4;
    if (l2 > 160 * 1024) return HSCN_E_UNSUPPORTED;
    if (l2 > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)k_ll_csr_t, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
    k_ll_csr_t<<<(unsigned)B, 256, l2, st>>>(A);
    HSCN_RETURN_IF_LAUNCH_FAILED();
  }
  return 0;
}
template <int H, int RT, typename TS>
int launch_bwd_rt(const BwdArgs& A, int64_t B, size_t lds, hipStream_t st) {
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)k_hscn_bwd<H, RT, TS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
  k_hscn_bwd<H, RT, TS><<<(unsigned)B, RT, lds, st>>>(A);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}
template <int H, typename TS>
int launch_bwd(BwdArgs& A, int64_t B, hipStream_t st) {
  const size_t lds = pick_bwd_lds(A, H);
  if (lds > 160 * 1024) return HSCN_E_UNSUPPORTED;
  static const int rt_env = getenv("HSCN_RT") ? atoi(getenv("HSCN_RT")) : 0;
  if (A.max_n <= 64 || rt_env == 256) return launch_bwd_rt<H, 256, TS>(A, B, lds, st);
  if (rt_env == 512 && sizeof(TS) == 4) return launch_bwd_rt<H, 512, float>(A, B, lds, st);
  return launch_bwd_rt<H, 1024, TS>(A, B, lds, st);
}

template <int H, int RT, typename TS>
int launch_bwd_virtual_rt(const BwdArgs& Ab, const FwdArgs& Af, int64_t B, size_t lds, hipStream_t st) {
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)k_hscn_bwd_virtual<H, RT, TS>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  k_hscn_bwd_virtual<H, RT, TS><<<(unsigned)(2 * B), RT, lds, st>>>(Ab, Af);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}
template <int H, typename TS>
int launch_bwd_virtual(BwdArgs& Ab, FwdArgs& Af, int64_t B, hipStream_t st) {
  const size_t lb = pick_bwd_lds(Ab, H);
  size_t lf = 0;
  bool ok = false;
  Af.spec = 1;
  Af.exp = 0;
  Af.exp_dinv = 0;
  lf = pick_fwd_lds(Af, H);
  ok = lf <= 160 * 1024;
  if (!ok || lb > 160 * 1024) return HSCN_E_UNSUPPORTED;
  const size_t lds = lb > lf ? lb : lf;
  if (Ab.max_n <= 64) return launch_bwd_virtual_rt<H, 256, TS>(Ab, Af, B, lds, st);
  return launch_bwd_virtual_rt<H, 1024, TS>(Ab, Af, B, lds, st);
}

int fill_fwd_args(FwdArgs& A, const float* x_local, const float* x_virtual, const int64_t* ei_ll, int64_t E_ll,
                  const int64_t* ei_vv, int64_t E_vv, const int64_t* ei_lv, int64_t E_lv, const int32_t* lptr,
                  const int32_t* vptr, const int32_t* eptr_ll, const int32_t* eptr_vv, const int32_t* eptr_lv,
                  int64_t N, int64_t V, int F, int H, int L, int C, int head_act, float slope,
                  const void* const* layer_params_host, const float* W1, const float* b1, const float* W2,
                  const float* b2, int max_n, int max_v, int max_ell, int max_evv, int compute_virtual,
                  float* acts, float* pooled, float* z, float* pred, float* xv_out, int32_t* csr_rowptr_t,
                  int32_t* csr_col_t, float* dinv_out, int32_t* flag) {
  if (compute_virtual < 0 || compute_virtual > 2) return HSCN_E_BADARG;
  const bool vonly = compute_virtual == 2;
  if (!x_local || !lptr || !vptr || !eptr_ll || !eptr_vv || !eptr_lv || !layer_params_host || !acts)
    return HSCN_E_BADARG;
  if (!vonly && (!W1 || !b1 || !W2 || !b2 || !pooled || !z || !pred)) return HSCN_E_BADARG;
  if (vonly && (!xv_out || csr_rowptr_t || csr_col_t || dinv_out)) return HSCN_E_BADARG;
  if ((E_ll > 0 && !ei_ll && !vonly) || (compute_virtual && ((E_vv > 0 && !ei_vv) || (E_lv > 0 && !ei_lv) || !x_virtual)))
    return HSCN_E_BADARG;
  A.x_local = x_local; A.x_virtual = x_virtual;
  A.ll_src = ei_ll; A.ll_dst = ei_ll ? ei_ll + E_ll : nullptr;
  A.vv_src = ei_vv; A.vv_dst = ei_vv ? ei_vv + E_vv : nullptr;
  A.lv_src = ei_lv; A.lv_dst = ei_lv ? ei_lv + E_lv : nullptr;
  A.lptr = lptr; A.vptr = vptr; A.eptr_ll = eptr_ll; A.eptr_vv = eptr_vv; A.eptr_lv = eptr_lv;
  for (int l = 0; l < L; ++l) {
    const void* const* q = layer_params_host + (size_t)l * 9;
    for (int k = 0; k < 9; ++k)
      if (!q[k] && (compute_virtual || k < 2)) return HSCN_E_BADARG;
    A.layer[l] = LayerP{(const float*)q[0], (const float*)q[1], (const float*)q[2], (const float*)q[3],
                        (const float*)q[4], (const float*)q[5], (const float*)q[6], (const float*)q[7],
                        (const float*)q[8]};
  }
  A.W1 = W1; A.b1 = b1; A.W2 = W2; A.b2 = b2;
  A.acts = acts; A.pooled = pooled; A.z = z; A.pred = pred; A.xv_out = xv_out; A.flag = flag;
  A.score = nullptr;
  if ((csr_rowptr_t == nullptr) != (csr_col_t == nullptr) || (csr_rowptr_t == nullptr) != (dinv_out == nullptr))
    return HSCN_E_BADARG;
  A.csr_rowptr_t = csr_rowptr_t; A.csr_col_t = csr_col_t; A.dinv_out = dinv_out;
  A.N = N; A.V = V; A.F = F; A.L = L; A.C = C; A.head_act = head_act;
  A.max_n = max_n; A.max_v = max_v; A.max_ell = vonly ? 0 : max_ell; A.max_evv = max_evv;
  A.compute_virtual = compute_virtual; A.slope = slope; A.spec = 0; A.exp = 0; A.exp_dinv = 0; A.db = 0;
  A.l_begin = 0; A.l_end = L;
  A.ready = nullptr; A.epoch = nullptr; A.acq = 0;
  A.pre_rp_lv = A.pre_col_lv = A.pre_rp_vv = A.pre_col_vv = nullptr; A.pre_dinv_v = nullptr;
  A.vs_rowptr_lv = A.vs_col_lv = A.vs_rowptr_vv = A.vs_col_vv = nullptr;
  A.vs_dinv_v = A.vs_xv = nullptr;
  return 0;
}

inline bool job_has_state(const hscn_virtual_job* j) {
  return j->st_rowptr_lv && j->st_col_lv && j->st_rowptr_vv && j->st_col_vv && j->st_dinv_v && j->st_xv;
}
inline void attach_state(FwdArgs& A, const hscn_virtual_job* j) {
  A.vs_rowptr_lv = j->st_rowptr_lv; A.vs_col_lv = j->st_col_lv;
  A.vs_rowptr_vv = j->st_rowptr_vv; A.vs_col_vv = j->st_col_vv;
  A.vs_dinv_v = j->st_dinv_v; A.vs_xv = j->st_xv;
}

template <int H, int RT, typename TS>
int launch_fwd_pair_rt(const FwdArgs& Al, const FwdArgs& Av, int64_t B, size_t lds, hipStream_t st) {
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)k_hscn_fwd_pair<H, RT, TS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
  FwdPair P;
  P.a[0] = Al;
  P.a[1] = Av;
  k_hscn_fwd_pair<H, RT, TS><<<(unsigned)(2 * B), RT, lds, st>>>(P);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}
template <int H, typename TS>
int launch_fwd_pair(FwdArgs& Al, FwdArgs& Av, int64_t B, hipStream_t st) {
  // The backward launch wants the ll CSR keyed by source and the degree norm.  The local workgroup is the long
  // pole of this launch, the virtual one has slack: the VIRTUAL workgroup builds and exports the source-keyed
  // CSR (it loads the ll edges for that alone), the local one only adds the degree norm it computes anyway.
  // Fallbacks when LDS is short: export from the local workgroup, then the light side kernel.
  const bool want_exp = Al.csr_rowptr_t != nullptr;
  size_t ll = 0, lv = 0;
  bool ok = false;
  // (a virtual-only workgroup is sized without ll edges unless it takes this job: Al.max_ell is the real bound)
  const bool v_can = want_exp && Av.ll_src && Av.csr_rowptr_t &&
                     fwd_lds_bytes(H, Av.C, Av.max_n, Av.max_v, Al.max_ell, Av.max_evv, 0, 1) <= 160 * 1024;
  Av.max_ell = v_can ? Al.max_ell : 0;
  for (int e = 1; e >= 0 && !ok; --e) {
    Al.spec = 0;
    Al.exp = (want_exp && e && !v_can) ? 1 : 0;
    ll = pick_fwd_lds(Al, H);
    ok = ll <= 160 * 1024;
  }
  if (!ok) return HSCN_E_UNSUPPORTED;
  Al.exp_dinv = (Al.exp || v_can) ? 1 : 0;
  ok = false;
  Av.spec = 1;
  Av.exp = v_can ? 1 : 0;
  Av.exp_dinv = 0;
  lv = pick_fwd_lds(Av, H);
  ok = lv <= 160 * 1024;
  if (!ok) return HSCN_E_UNSUPPORTED;
  if (!v_can) { Av.ll_src = nullptr; Av.ll_dst = nullptr; }
  const size_t lds = ll > lv ? ll : lv;
  int rc = Al.max_n <= 64 ? launch_fwd_pair_rt<H, 256, TS>(Al, Av, B, lds, st)
                          : launch_fwd_pair_rt<H, 1024, TS>(Al, Av, B, lds, st);
  if (rc) return rc;
  if (want_exp && !Al.exp && !v_can) {
    const size_t l2 = ((size_t)4 * Al.max_ell + 2 * ((size_t)Al.max_n + 1) + 16) * 4;
    if (l2 > 160 * 1024) return HSCN_E_UNSUPPORTED;
    if (l2 > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)k_ll_csr_t, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
    k_ll_csr_t<<<(unsigned)B, 256, l2, st>>>(Al);
    HSCN_RETURN_IF_LAUNCH_FAILED();
  }
  return 0;
}

int fill_bwd_args(BwdArgs& A, const float* x_local, const int64_t* ei_ll, int64_t E_ll, const int32_t* lptr,
                  const int32_t* eptr_ll, int64_t N, int F, int H, int L, int C, int head_act,
                  const void* const* W_ll_host, const float* W1, const float* W2, const float* acts,
                  const float* pooled, const float* z, const float* g_pred, const float* g_scale,
                  const int32_t* csr_rowptr_t, const int32_t* csr_col_t, const float* dinv, int max_n, int max_ell,
                  float* partials, float* grads, int32_t* flag) {
  if (!x_local || !lptr || !eptr_ll || !W_ll_host || !W1 || !W2 || !acts || !pooled || !z ||
      !partials || !grads || !csr_rowptr_t || !dinv || (E_ll > 0 && !csr_col_t))
    return HSCN_E_BADARG;
  A.x_local = x_local; A.ll_src = ei_ll; A.ll_dst = ei_ll ? ei_ll + E_ll : nullptr;
  A.lptr = lptr; A.eptr_ll = eptr_ll;
  for (int l = 0; l < L; ++l) {
    if (!W_ll_host[l]) return HSCN_E_BADARG;
    A.W_ll[l] = (const float*)W_ll_host[l];
  }
  A.W1 = W1; A.W2 = W2; A.acts = acts; A.pooled = pooled; A.z = z; A.g_pred = g_pred; A.g_scale = g_scale;
  A.csr_rowptr_t = csr_rowptr_t; A.csr_col_t = csr_col_t; A.dinv_in = dinv;
  A.partials = partials; A.flag = flag; A.N = N; A.F = F; A.L = L; A.C = C; A.head_act = head_act;
  A.max_n = max_n; A.max_ell = max_ell; A.P = (int)hscn_resident_param_count(F, H, L, C);
  A.pred = nullptr; A.target = nullptr; A.loss_kind = 0; A.Pn = A.P; A.inv_count = 0.f;
  return 0;
}

// upstream gradient: either given (g_pred) or computed by the launch from the loss tail
int attach_tail(BwdArgs& A, const hscn_loss_tail* tail, int64_t B) {
  if (!tail) return A.g_pred ? 0 : HSCN_E_BADARG;
  if (!tail->pred || !tail->target || (tail->kind != 0 && tail->kind != 1)) return HSCN_E_BADARG;
  A.pred = tail->pred; A.target = tail->target; A.loss_kind = tail->kind;
  A.inv_count = 1.0f / (float)(B * (int64_t)A.C);   // k_criterion's 1 / count
  A.P = A.Pn + 1;                                     // partials rows and grads carry the loss column
  return 0;
}


#include "resident_step.h"

// ---- the C entry points, generic in the storage type (resident.hip: float, resident_f16.hip: half) ----
template <typename TS>
int impl_resident_fwd(const float* x_local, const float* x_virtual, const int64_t* ei_ll, int64_t E_ll,
                      const int64_t* ei_vv, int64_t E_vv, const int64_t* ei_lv, int64_t E_lv,
                      const int32_t* lptr, const int32_t* vptr, const int32_t* eptr_ll, const int32_t* eptr_vv,
                      const int32_t* eptr_lv, int64_t N, int64_t V, int64_t B, int F, int H, int L, int C,
                      int head_act, float slope, const void* const* layer_params_host /* L x 9 */,
                      const float* W1, const float* b1, const float* W2, const float* b2, int max_n, int max_v,
                      int max_ell, int max_evv, int compute_virtual, float* acts, float* pooled, float* z,
                      float* pred, float* score, float* xv_out, int32_t* csr_rowptr_t, int32_t* csr_col_t,
                      float* dinv_out, int32_t* flag, void* stream_) {
  if (B < 0 || N < 0 || V < 0) return HSCN_E_BADARG;
  if (B == 0) return 0;
  if (!hscn_resident_supported(F, H, L, C, max_n, max_v, max_ell, max_evv)) return HSCN_E_UNSUPPORTED;
  FwdArgs A;
  if (int rc = fill_fwd_args(A, x_local, x_virtual, ei_ll, E_ll, ei_vv, E_vv, ei_lv, E_lv, lptr, vptr, eptr_ll,
                             eptr_vv, eptr_lv, N, V, F, H, L, C, head_act, slope, layer_params_host, W1, b1, W2,
                             b2, max_n, max_v, max_ell, max_evv, compute_virtual, acts, pooled, z, pred, xv_out,
                             csr_rowptr_t, csr_col_t, dinv_out, flag))
    return rc;
  A.score = score;
  hipStream_t st = hscn_stream(stream_);
  switch (H) {
    case 16: return launch_fwd<16, TS>(A, B, st);
    case 32: return launch_fwd<32, TS>(A, B, st);
    case 64: if constexpr (sizeof(TS) == 4) return launch_fwd<64, float>(A, B, st); else break;
  }
  return HSCN_E_UNSUPPORTED;
}

template <typename TS>
int impl_resident_bwd(const float* x_local, const int64_t* ei_ll, int64_t E_ll, const int32_t* lptr,
                      const int32_t* eptr_ll, int64_t N, int64_t B, int F, int H, int L, int C, int head_act,
                      const void* const* W_ll_host /* L */, const float* W1, const float* W2, const float* acts,
                      const float* pooled, const float* z, const float* g_pred, const float* g_scale,
                      const int32_t* csr_rowptr_t, const int32_t* csr_col_t, const float* dinv, int max_n,
                      int max_ell, float* partials /*[B][P]*/, float* grads /*[P]*/, int32_t* flag,
                      const hscn_loss_tail* tail, void* stream_) {
  if (B < 0 || N < 0) return HSCN_E_BADARG;
  if (B == 0) return 0;
  if (!hscn_resident_supported(F, H, L, C, max_n, 0, max_ell, 0)) return HSCN_E_UNSUPPORTED;
  BwdArgs A;
  if (int rc0 = fill_bwd_args(A, x_local, ei_ll, E_ll, lptr, eptr_ll, N, F, H, L, C, head_act, W_ll_host, W1, W2,
                              acts, pooled, z, g_pred, g_scale, csr_rowptr_t, csr_col_t, dinv, max_n, max_ell,
                              partials, grads, flag))
    return rc0;
  if (int rct = attach_tail(A, tail, B)) return rct;
  hipStream_t st = hscn_stream(stream_);
  int rc = HSCN_E_UNSUPPORTED;
  switch (H) {
    case 16: rc = launch_bwd<16, TS>(A, B, st); break;
    case 32: rc = launch_bwd<32, TS>(A, B, st); break;
    case 64: if constexpr (sizeof(TS) == 4) rc = launch_bwd<64, float>(A, B, st); break;
  }
  if (rc) return rc;
  k_param_reduce<<<hscn_blocks(A.P, 32), 256, 0, st>>>(partials, grads, (int)B, A.P, A.target ? A.Pn : -1, A.inv_count);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

template <typename TS>
int impl_resident_bwd_with_virtual(const float* x_local, const int64_t* ei_ll, int64_t E_ll, const int32_t* lptr,
                                   const int32_t* eptr_ll, int64_t N, int64_t B, int F, int H, int L, int C,
                                   int head_act, const void* const* W_ll_host, const float* W1, const float* W2,
                                   const float* acts, const float* pooled, const float* z, const float* g_pred,
                                   const float* g_scale, const int32_t* csr_rowptr_t, const int32_t* csr_col_t,
                                   const float* dinv, int max_n, int max_ell, float* partials, float* grads,
                                   int32_t* flag, const hscn_loss_tail* tail, const hscn_virtual_job* job,
                                   void* stream_) {
  if (B < 0 || N < 0 || !job) return HSCN_E_BADARG;
  if (B == 0) return 0;
  if (!hscn_resident_supported(F, H, L, C, max_n, job->max_v, max_ell, job->max_evv)) return HSCN_E_UNSUPPORTED;
  BwdArgs Ab;
  if (int rc0 = fill_bwd_args(Ab, x_local, ei_ll, E_ll, lptr, eptr_ll, N, F, H, L, C, head_act, W_ll_host, W1,
                              W2, acts, pooled, z, g_pred, g_scale, csr_rowptr_t, csr_col_t, dinv, max_n,
                              max_ell, partials, grads, flag))
    return rc0;
  if (int rct = attach_tail(Ab, tail, B)) return rct;
  FwdArgs Af;
  if (int rc1 = fill_fwd_args(Af, x_local, job->x_virtual, nullptr, 0, job->ei_vv, job->E_vv, job->ei_lv,
                              job->E_lv, lptr, job->vptr, eptr_ll, job->eptr_vv, job->eptr_lv, N, job->V, F, H, L,
                              C, head_act, job->slope, job->layer_params_host, nullptr, nullptr, nullptr, nullptr,
                              max_n, job->max_v, max_ell, job->max_evv, 2, const_cast<float*>(acts), nullptr,
                              nullptr, nullptr, job->xv_out, nullptr, nullptr, nullptr, flag))
    return rc1;
  if (job_has_state(job)) {   // the forward launch ran structure + layer 0 (hscn_resident_fwd_with_virtual)
    if (L < 2) return HSCN_E_BADARG;
    attach_state(Af, job);
    Af.l_begin = 1;
  }
  hipStream_t st = hscn_stream(stream_);
  int rc = HSCN_E_UNSUPPORTED;
  switch (H) {
    case 16: rc = launch_bwd_virtual<16, TS>(Ab, Af, B, st); break;
    case 32: rc = launch_bwd_virtual<32, TS>(Ab, Af, B, st); break;
    case 64: if constexpr (sizeof(TS) == 4) rc = launch_bwd_virtual<64, float>(Ab, Af, B, st); break;
  }
  if (rc) return rc;
  k_param_reduce<<<hscn_blocks(Ab.P, 32), 256, 0, st>>>(partials, grads, (int)B, Ab.P, Ab.target ? Ab.Pn : -1, Ab.inv_count);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

template <typename TS>
int impl_resident_fwd_with_virtual(const float* x_local, const int64_t* ei_ll, int64_t E_ll, const int32_t* lptr,
                                   const int32_t* eptr_ll, int64_t N, int64_t B, int F, int H, int L, int C,
                                   int head_act, const void* const* layer_params_host, const float* W1,
                                   const float* b1, const float* W2, const float* b2, int max_n, int max_ell,
                                   float* acts, float* pooled, float* z, float* pred, float* score,
                                   int32_t* csr_rowptr_t, int32_t* csr_col_t, float* dinv_out, int32_t* flag,
                                   const hscn_virtual_job* job, void* stream_) {
  if (B < 0 || N < 0 || !job) return HSCN_E_BADARG;
  if (B == 0) return 0;
  if (L < 2 || !job_has_state(job)) return HSCN_E_BADARG;
  if (!hscn_resident_supported(F, H, L, C, max_n, job->max_v, max_ell, job->max_evv)) return HSCN_E_UNSUPPORTED;
  FwdArgs Al, Av;
  if (int rc0 = fill_fwd_args(Al, x_local, job->x_virtual, ei_ll, E_ll, job->ei_vv, job->E_vv, job->ei_lv,
                              job->E_lv, lptr, job->vptr, eptr_ll, job->eptr_vv, job->eptr_lv, N, job->V, F, H, L,
                              C, head_act, job->slope, layer_params_host, W1, b1, W2, b2, max_n, job->max_v,
                              max_ell, job->max_evv, 0, acts, pooled, z, pred, nullptr, csr_rowptr_t, csr_col_t,
                              dinv_out, flag))
    return rc0;
  Al.score = score;
  // virtual part 1: `acts` is not read by layer 0 (it takes the input features) but must be valid
  if (int rc1 = fill_fwd_args(Av, x_local, job->x_virtual, nullptr, 0, job->ei_vv, job->E_vv, job->ei_lv,
                              job->E_lv, lptr, job->vptr, eptr_ll, job->eptr_vv, job->eptr_lv, N, job->V, F, H, L,
                              C, head_act, job->slope, job->layer_params_host, nullptr, nullptr, nullptr, nullptr,
                              max_n, job->max_v, max_ell, job->max_evv, 2, acts, nullptr, nullptr, nullptr,
                              job->xv_out ? job->xv_out : job->st_xv, nullptr, nullptr, nullptr, flag))
    return rc1;
  attach_state(Av, job);
  Av.l_begin = 0;
  Av.l_end = 1;
  // what the virtual workgroup needs to build the backward's source-keyed ll CSR (launch_fwd_pair decides)
  Av.ll_src = Al.ll_src; Av.ll_dst = Al.ll_dst;
  Av.csr_rowptr_t = csr_rowptr_t; Av.csr_col_t = csr_col_t;
  hipStream_t st = hscn_stream(stream_);
  switch (H) {
    case 16: return launch_fwd_pair<16, TS>(Al, Av, B, st);
    case 32: return launch_fwd_pair<32, TS>(Al, Av, B, st);
    case 64: if constexpr (sizeof(TS) == 4) return launch_fwd_pair<64, float>(Al, Av, B, st); else break;
  }
  return HSCN_E_UNSUPPORTED;
}


// one-launch training step (resident_step.h).  sync: [0] = epoch word, [32 .. 32 + B) = per-graph publish counters.
template <typename TS>
int impl_resident_train_step(const float* x_local, const int64_t* ei_ll, int64_t E_ll, const int32_t* lptr,
                             const int32_t* eptr_ll, int64_t N, int64_t B, int F, int H, int L, int C, int head_act,
                             const void* const* layer_params_host, const float* W1, const float* b1, const float* W2,
                             const float* b2, int max_n, int max_ell, const float* target, int loss_kind, float* pred,
                             float* score, float* partials, float* grads, float* acts, uint32_t* sync, int32_t* flag,
                             const hscn_virtual_job* job, const hscn_structure* pre, void* stream_) {
  if (B < 0 || N < 0) return HSCN_E_BADARG;
  if (pre && (!pre->ll_rowptr_d || !pre->ll_rowptr_s || !pre->ll_dinv || (E_ll > 0 && (!pre->ll_col_d || !pre->ll_col_s))))
    return HSCN_E_BADARG;
  if (pre && job && (!pre->lv_rowptr || !pre->vv_rowptr || !pre->vv_dinv || (job->E_lv > 0 && !pre->lv_col) ||
                     (job->E_vv > 0 && !pre->vv_col)))
    return HSCN_E_BADARG;
  if (B == 0) return 0;
  if (!(H == 16 || H == 32) || F < 1 || F > H || L < 1 || L > MAXL || C < 1 || C > 4096 || max_n < 0 || max_ell < 0)
    return HSCN_E_UNSUPPORTED;
  if (!x_local || !lptr || !eptr_ll || !layer_params_host || !W1 || !b1 || !W2 || !b2 || !target || !pred ||
      !partials || !grads || (E_ll > 0 && !ei_ll) || (loss_kind != 0 && loss_kind != 1))
    return HSCN_E_BADARG;
  if (job && (!acts || !sync || !job->xv_out)) return HSCN_E_BADARG;
  StepArgs S;
  S.x_local = x_local; S.ll_src = ei_ll; S.ll_dst = ei_ll ? ei_ll + E_ll : nullptr; S.lptr = lptr; S.eptr_ll = eptr_ll;
  for (int l = 0; l < L; ++l) {
    const void* const* q = layer_params_host + (size_t)l * 9;
    if (!q[0] || !q[1]) return HSCN_E_BADARG;
    S.W_ll[l] = (const float*)q[0];
    S.b_ll[l] = (const float*)q[1];
  }
  S.W1 = W1; S.b1 = b1; S.W2 = W2; S.b2 = b2; S.target = target; S.pred = pred; S.score = score;
  S.partials = partials; S.flag = flag; S.N = N; S.F = F; S.L = L; S.C = C; S.head_act = head_act;
  S.max_n = max_n; S.max_ell = max_ell; S.Pn = (int)hscn_resident_param_count(F, H, L, C); S.P = S.Pn + 1;
  S.loss_kind = loss_kind; S.inv_count = 1.0f / (float)(B * (int64_t)C); S.B = (int)B;
  S.pre_rp_d = pre ? pre->ll_rowptr_d : nullptr; S.pre_col_d = pre ? pre->ll_col_d : nullptr;
  S.pre_rp_s = pre ? pre->ll_rowptr_s : nullptr; S.pre_col_s = pre ? pre->ll_col_s : nullptr;
  S.pre_dinv = pre ? pre->ll_dinv : nullptr;
  S.acts = (acts && L >= 2) ? acts : nullptr;
  S.ready = (job && S.acts) ? sync + 32 : nullptr;
  S.epoch = sync;
  FwdArgs V;
  if (job) {
    if (int rc1 = fill_fwd_args(V, x_local, job->x_virtual, nullptr, 0, job->ei_vv, job->E_vv, job->ei_lv, job->E_lv,
                                lptr, job->vptr, eptr_ll, job->eptr_vv, job->eptr_lv, N, job->V, F, H, L, C, head_act,
                                job->slope, job->layer_params_host, nullptr, nullptr, nullptr, nullptr, max_n,
                                job->max_v, max_ell, job->max_evv, 2, acts, nullptr, nullptr, nullptr, job->xv_out,
                                nullptr, nullptr, nullptr, flag))
      return rc1;
    V.ready = S.ready; V.epoch = sync;
    if (pre) {
      V.pre_rp_lv = pre->lv_rowptr; V.pre_col_lv = pre->lv_col; V.pre_rp_vv = pre->vv_rowptr;
      V.pre_col_vv = pre->vv_col; V.pre_dinv_v = pre->vv_dinv;
    }
  }
  hipStream_t st = hscn_stream(stream_);
  int rc = H == 16 ? launch_step<16, TS>(S, job ? &V : nullptr, st) : launch_step<32, TS>(S, job ? &V : nullptr, st);
  if (rc) return rc;
#ifdef HSCN_DIAG_REDUCE_BLOCKS   // measurement builds only (tools/build_variant.sh): what the launch costs without its work
  k_param_reduce<<<HSCN_DIAG_REDUCE_BLOCKS, 256, 0, st>>>(partials, grads, (int)B, S.P, S.Pn, S.inv_count,
                                                          S.ready ? sync : nullptr);
#else
  k_param_reduce<<<hscn_blocks(S.P, 32), 256, 0, st>>>(partials, grads, (int)B, S.P, S.Pn, S.inv_count,
                                                       S.ready ? sync : nullptr);
#endif
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

inline int impl_resident_structure(const int64_t* ei_ll, int64_t E_ll, const int64_t* ei_vv, int64_t E_vv,
                                   const int64_t* ei_lv, int64_t E_lv, const int32_t* lptr, const int32_t* vptr,
                                   const int32_t* eptr_ll, const int32_t* eptr_vv, const int32_t* eptr_lv, int64_t B,
                                   int max_n, int max_v, int max_ell, int max_evv, const hscn_structure* out,
                                   int32_t* flag, void* stream_) {
  if (B < 0 || !out || max_n < 0 || max_v < 0 || max_ell < 0 || max_evv < 0) return HSCN_E_BADARG;
  if (B == 0) return 0;
  if (!lptr || !vptr || !eptr_ll || !eptr_vv || !eptr_lv || (E_ll > 0 && !ei_ll) || (E_vv > 0 && !ei_vv) ||
      (E_lv > 0 && !ei_lv))
    return HSCN_E_BADARG;
  if (!out->ll_rowptr_d || !out->ll_rowptr_s || !out->ll_dinv || !out->lv_rowptr || !out->vv_rowptr || !out->vv_dinv ||
      (E_ll > 0 && (!out->ll_col_d || !out->ll_col_s)) || (E_lv > 0 && !out->lv_col) || (E_vv > 0 && !out->vv_col))
    return HSCN_E_BADARG;
  StructArgs A;
  A.ll_src = ei_ll; A.ll_dst = ei_ll ? ei_ll + E_ll : nullptr;
  A.vv_src = ei_vv; A.vv_dst = ei_vv ? ei_vv + E_vv : nullptr;
  A.lv_src = ei_lv; A.lv_dst = ei_lv ? ei_lv + E_lv : nullptr;
  A.lptr = lptr; A.vptr = vptr; A.eptr_ll = eptr_ll; A.eptr_vv = eptr_vv; A.eptr_lv = eptr_lv;
  A.out = *out; A.flag = flag; A.max_n = max_n; A.max_v = max_v; A.max_ell = max_ell; A.max_evv = max_evv;
  const size_t lds = struct_lds_words(max_n, max_v, max_ell, max_evv) * 4;
  if (lds > 160 * 1024) return HSCN_E_UNSUPPORTED;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)k_structure, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  k_structure<<<(unsigned)B, 256, lds, hscn_stream(stream_)>>>(A);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

// workgroups of the one-launch step that one CU takes at a time: 1 for 16-wave workgroups (they are not made to
// share a CU), up to 4 for the 4-wave workgroups of small graphs (PCQM-Contact: n <= 64) when LDS allows
inline int step_wgs_per_cu(int H, int L, int C, int max_n, int max_ell, int max_v, int max_evv) {
  if (max_n > 64) return 1;
  size_t lds = step_lds_bytes(H, L, C, max_n, max_ell);
  if (max_v > 0) {
    const size_t lv = fwd_lds_bytes(H, C, max_n, max_v, 0, max_evv, 1, 0);
    if (lv > lds) lds = lv;
  }
  if (lds == 0) return 4;
  const int by_lds = (int)((size_t)160 * 1024 / lds);
  return by_lds < 1 ? 1 : (by_lds > 4 ? 4 : by_lds);
}

inline int step_supported(int F, int H, int L, int C, int max_n, int max_ell, int max_v, int max_evv) {
  if (!(H == 16 || H == 32) || F < 1 || F > H || L < 1 || L > MAXL || C < 1 || C > 4096) return 0;
  if (max_n < 0 || max_ell < 0 || max_v < 0 || max_evv < 0) return 0;
  if (step_lds_bytes(H, L, C, max_n, max_ell) > 160 * 1024) return 0;
  if (max_v > 0 && fwd_lds_bytes(H, C, max_n, max_v, 0, max_evv, 0, 0) > 160 * 1024) return 0;
  return 1;
}

}  // namespace
