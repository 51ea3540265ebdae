// One-launch training step of stage C (reference train/train.py:73-95: pred = model(...); loss = criterion(...);
// loss.backward()): workgroup g runs the forward of graph g, its own row of d loss / d pred, and the backward,
// with the graph's structure AND every layer's activation resident in LDS from the first load to the last
// gradient partial.  Included by resident_kernels.h (inside its anonymous namespace).
//
// What disappears against the two-launch route (k_hscn_fwd_pair + k_hscn_bwd_virtual):
//   * one launch boundary and one replay gap;
//   * the backward's prologue (segment tables, exported CSR, degree norm, last activation, head weights, pooled, z:
//     two dependent HBM round trips) -- everything it loaded is still in LDS;
//   * the export + re-import of the source-keyed CSR, the degree norm and L x n x H activations (~8 of the ~17 MB
//     a step moved);
//   * the per-layer weight re-fetch of the backward: the forward's transposed W_ll stay in LDS and the backward
//     reads them transposed again.
// The per-graph loss gradient needs no other graph: d(mean loss)/d pred_g = criterion'(pred_g, y_g) / (B C); the
// mean loss itself is one more column of the per-graph partials that k_param_reduce sums anyway.
//
// LDS: NB = max(L + 1, 3) buffers of n x H floats: A[0] = input features, A[l + 1] = output of layer l.  In the
// backward the gradient G lives in A[L] (masked in place), GH = A_hat^T G in A[0] (the features are re-read from
// HBM for layer 0's weight gradient into A[1], dead by then; L = 1 keeps them and uses A[2]).  H = 16: a 444-node
// graph (Peptides' maximum) fits; H = 32 up to ~290 nodes; beyond that the caller takes the two-launch route.
//
// The virtual branch (which cannot reach the prediction: DESIGN.md section 2) runs as B more workgroups of the same
// launch (blocks B .. 2B-1, hscn_fwd_body MODE 2 over all layers).  Its layer l >= 1 reads the local activation
// a_l, so the local workgroup publishes a_1 .. a_{L-1}: write-through (sc1) 16-byte stores from LDS at the top of
// the following layer, every wave drains (s_waitcnt vmcnt(0)) in front of that layer's barrier, then ONE lane stores
// the per-graph flag (agent-scope atomic) -- no release fence, nothing on the local critical path but the store
// issue.  The virtual workgroup's loading waves poll the flag relaxed, acquire once (agent scope) and load
// (cdna_hip_programming.md Guideline 16, R1).  Local workgroups never wait for anyone; the grid puts them first;
// every spin is bounded (flag bit 8 on timeout -- the virtual features are then invalid, prediction and gradients
// are not affected).  Flags are compared against a per-step epoch kept in device memory and advanced by
// k_param_reduce, so no per-step memset is needed and nothing is frozen under hipGraph replay.

struct StepArgs {
  const float* x_local;  // TS [N][F]
  const int64_t *ll_src, *ll_dst;
  const int32_t *lptr, *eptr_ll;
  const float* W_ll[MAXL];
  const float* b_ll[MAXL];
  const float *W1, *b1, *W2, *b2;
  const float* target;   // [B][C]
  float *pred, *score;   // [B][C]
  float* partials;       // [B][P], P = Pn + 1 (last column: the graph's summed loss terms)
  float* acts;           // TS [L-1][N][H]: a_1 .. a_{L-1} for the virtual workgroups (NULL: not published)
  uint32_t* ready;       // [B] per-graph publish counters (epoch * 8 + number of published activations)
  const uint32_t* epoch; // device word, advanced once per step by k_param_reduce
  // structure_build = "dataset-resident" (include/hscn.h: hscn_structure): both CSRs of the local->local relation and
  // the degree norm come from HBM instead of being rebuilt from the COO slice (NULL: build)
  const int32_t *pre_rp_d, *pre_col_d, *pre_rp_s, *pre_col_s;
  const float* pre_dinv;
  int32_t* flag;
  int64_t N;
  int F, L, C, head_act, max_n, max_ell, P, Pn, loss_kind;
  float inv_count;
  int B;
};

struct StepLayout {
  size_t A, dinv, wt, headw, part, vec, red, rowptr, col, rowptr_t, col_t, wsum, total;
  size_t ek, eo, cursorA, tmpA, cursorT, tmpT;
  size_t cntA, cntT, ellA, ellT, ovf;   // the two-barrier build for low-degree rows (build_csr_pair_ell / build_ell16_pair)
  size_t recT;                          // source-keyed row records (the target-keyed ones sit at `rowptr`)
  int NB;
  size_t bufw;  // words per n x H buffer
};
__host__ __device__ inline StepLayout step_layout(int H, int L, int C, int max_n, int max_ell) {
  StepLayout Y;
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o += (n + 3) & ~(size_t)3; return r; };
  auto up4 = [](size_t n) { return (n + 3) & ~(size_t)3; };
  Y.NB = L + 1 > 3 ? L + 1 : 3;
  // staged COO slices live in A[1], the CSR builds' counters / slot lists in A[2] (both dead until layers 0 / 1 write
  // them); a buffer is widened if a dense graph needs more than n x H words for either
  size_t bufw = (size_t)((max_n + 15) / 16 * 16) * H;   // whole 16-row tiles (A[0] doubles as per-wave tile scratch)
  const size_t stage = 2 * up4(max_ell);
  const size_t scratch = 2 * up4((size_t)max_n + 1) + 2 * up4(max_ell);
  const size_t scratch_ell = 2 * up4((size_t)max_n + 1) + 2 * up4((size_t)max_n * ELL_D) + 4;
  if (stage > bufw) bufw = stage;
  if (scratch > bufw) bufw = scratch;
  if (scratch_ell > bufw) bufw = scratch_ell;
  Y.bufw = bufw;
  Y.A = take(bufw * Y.NB);
  Y.ek = Y.A + bufw;
  Y.eo = Y.ek + up4(max_ell);
  Y.cursorA = Y.A + 2 * bufw;
  Y.tmpA = Y.cursorA + up4((size_t)max_n + 1);
  Y.cursorT = Y.tmpA + up4(max_ell);
  Y.tmpT = Y.cursorT + up4((size_t)max_n + 1);
  Y.cntA = Y.A + 2 * bufw;                                  // (the same words as the general build's scratch)
  Y.cntT = Y.cntA + up4((size_t)max_n + 1);
  Y.ellA = Y.cntT + up4((size_t)max_n + 1);
  Y.ellT = Y.ellA + up4((size_t)max_n * ELL_D);
  Y.ovf = Y.ellT + up4((size_t)max_n * ELL_D);
  Y.dinv = take(max_n);
  Y.wt = take((size_t)L * (H * H + H));                            // per layer: Wt[k][o] (rows k >= fin zero) | b[H]
  Y.headw = take((size_t)H * H + H + (size_t)C * H + 4 * (size_t)C);  // W1 | b1 | W2 | b2 | target row | g_pred row | loss terms
  Y.part = take((size_t)3 * (RT_MAX / 64) * H);                    // pool partials, later bias-gradient partials (x3: layer mod 3)
  Y.vec = take(320);
  Y.red = take((size_t)(RT_MAX / 64) * 256);
  // the structure region: two CSRs (rowptr | col | rowptr_t | col_t) or, for low-degree graphs at H = 16, two tables of
  // 16-byte row records (build_ell16_pair: recA [max_n] | recT [max_n]) -- whichever is larger
  {
    const size_t csr_w = 2 * up4((size_t)max_n + 1) + 2 * up4(max_ell), rec_w = H == 16 ? 2 * 4 * (size_t)max_n : 0;
    Y.rowptr = take(csr_w > rec_w ? csr_w : rec_w);
    Y.col = Y.rowptr + up4((size_t)max_n + 1);
    Y.rowptr_t = Y.col + up4(max_ell);
    Y.col_t = Y.rowptr_t + up4((size_t)max_n + 1);
    Y.recT = Y.rowptr + 4 * (size_t)max_n;
  }
  Y.wsum = take(32);
  Y.total = o;
  return Y;
}

// Y[n][H] = mask(M) .* (X[n][H] * W), W given TRANSPOSED in LDS (Wt[k][o] = W[o][k], the forward's layout):
// the backward's input gradient on the matrix cores without a second copy of the weights.
template <int H>
__device__ void lin_mfma_wt_masked(const float* X, const float* Wt, float* Y, int n, const float* M, const Grp& G) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  constexpr int TD = H / 16, KS = H / 4;
  const int lane = threadIdx.x & 63, li = lane & 15, lj = lane >> 4;
  const int ntile = (n + 15) >> 4;
  if (G.w >= ntile) return;
  float b[TD][KS];
#pragma unroll
  for (int ct = 0; ct < TD; ++ct)
#pragma unroll
    for (int s = 0; s < KS; ++s) b[ct][s] = Wt[(ct * 16 + li) * H + 4 * s + lj];   // = W[4s+lj][ct*16+li]
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
          Y[idx] = M[idx] > 0.f ? acc[ct][r] : 0.f;
        }
      }
  }
}

template <int H, int RT, typename TS>
__device__ __forceinline__ void hscn_step_local(const StepArgs& A, const int g) {
  static_assert(H <= 32, "the one-launch step uses the fused local layer (H <= 32)");
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int NW = RT / 64;
  const TS* const xl_g = reinterpret_cast<const TS*>(A.x_local);
  TS* const acts_g = reinterpret_cast<TS*>(A.acts);
  const int n0 = A.lptr[g], n = A.lptr[g + 1] - n0;
  const int e0 = A.eptr_ll[g], ne = A.eptr_ll[g + 1] - e0;
  float* part = A.partials + (size_t)g * A.P;
  const int F = A.F, L = A.L, C = A.C;
  const uint32_t ep8 = A.ready ? A.epoch[0] * 8u : 0u;
  if ((n > A.max_n) | (ne > A.max_ell) | (n < 0) | (ne < 0)) {
    if (threadIdx.x == 0 && A.flag) atomicOr(A.flag, 4);
    for (int i = threadIdx.x; i < A.P; i += RT) part[i] = 0.f;
    // the virtual workgroup of this graph must not wait for activations that will never come
    if (threadIdx.x == 0 && A.ready) __hip_atomic_store(A.ready + g, ep8 + 7u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  const StepLayout Y = step_layout(H, L, C, A.max_n, A.max_ell);
  float* fb = reinterpret_cast<float*>(smem);
  int* ib = reinterpret_cast<int*>(smem);
  float* Abuf = fb + Y.A;
  auto buf = [&](int k) { return Abuf + (size_t)k * Y.bufw; };
  float *dinv = fb + Y.dinv, *wt = fb + Y.wt, *headw = fb + Y.headw, *partp = fb + Y.part, *vec = fb + Y.vec;
  float* red = fb + Y.red;
  int *rowptr = ib + Y.rowptr, *col = ib + Y.col, *rowptr_t = ib + Y.rowptr_t, *col_t = ib + Y.col_t;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;   // (scalar: "tiles of this wave" loops stay uniform)
  const Grp ALL{(int)threadIdx.x, RT, wave, NW};
  constexpr int WL = H * H + H;   // words per layer in wt

#ifndef HSCN_CSR_ELL
#define HSCN_CSR_ELL 1
#endif
#ifndef HSCN_BIAS0_COUNTS
#define HSCN_BIAS0_COUNTS 1
#endif
  // ---- prologue: every global input of the graph is requested before anything is consumed ----------------
  STAMP(0);
  constexpr int EPT = 2, XPT = 8;
  constexpr int WPT = (MAXL * WL + RT - 1) / RT > 4 ? 4 : (MAXL * WL + RT - 1) / RT;   // weight words per thread in registers
  const int64_t* dummy = reinterpret_cast<const int64_t*>(A.lptr);
  const int64_t *pld = A.ll_dst ? A.ll_dst : dummy, *pls = A.ll_src ? A.ll_src : dummy;
  long long rld[EPT], rls[EPT];
  float xr[XPT];
  const int wbase = __builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));
  const bool pre = A.pre_rp_d != nullptr;      // dataset-resident structure: loaded, not built
  int prd[EPT], prs[EPT], pcd[EPT], pcs[EPT];
  float pdv[EPT];
#pragma unroll
  for (int i = 0; i < EPT; ++i) {
    const int e = threadIdx.x + i * RT;
    const bool o1 = e < ne && A.ll_dst;
    rld[i] = 0; rls[i] = 0; prd[i] = 0; prs[i] = 0; pcd[i] = 0; pcs[i] = 0; pdv[i] = 0.f;
    if (!pre) {
      if (wbase + i * RT < ne) { rld[i] = pld[o1 ? e0 + e : 0]; rls[i] = pls[o1 ? e0 + e : 0]; }
    } else {
      if (wbase + i * RT <= n) {
        prd[i] = A.pre_rp_d[(size_t)n0 + g + (e <= n ? e : 0)];
        prs[i] = A.pre_rp_s[(size_t)n0 + g + (e <= n ? e : 0)];
        pdv[i] = A.pre_dinv[(size_t)n0 + (e < n ? e : 0)];
      }
      if (wbase + i * RT < ne) {
        pcd[i] = A.pre_col_d[(size_t)e0 + (e < ne ? e : 0)];
        pcs[i] = A.pre_col_s[(size_t)e0 + (e < ne ? e : 0)];
      }
    }
  }
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
  // all layers' local->local weights, transposed on the way in: slot d = l*WL + k*H + o  <-  W_l[o][k]; bias behind
  float wr[WPT];
  const int WTOT = L * WL;
#pragma unroll
  for (int i = 0; i < WPT; ++i) {
    const int d = threadIdx.x + i * RT;
    const int l = d / WL, q = d - l * WL;
    const int fin = l == 0 ? F : H;
    wr[i] = 0.f;
    if (wbase + i * RT < WTOT) {
      const bool inb = d < WTOT;
      const int ll = inb ? l : 0;
      const bool isb = q >= H * H;
      const int k = q / H, o = q - k * H;
      const bool ok = inb && (isb || k < fin);
      const float* src = isb ? A.b_ll[ll] + (q - H * H) : A.W_ll[ll] + (ok ? o * fin + k : 0);
      const float t = *(ok ? src : A.b_ll[0]);
      wr[i] = ok ? t : 0.f;
    }
  }
  // head weights W1 [H][H] | b1 [H] | W2 [C][H] | b2 [C] (natural layout), then this graph's target row
  const int HT = H * H + H + C * H + C;
  auto haddr = [&](int idx) -> const float* {
    if (idx < H * H) return A.W1 + idx;
    idx -= H * H;
    if (idx < H) return A.b1 + idx;
    idx -= H;
    if (idx < C * H) return A.W2 + idx;
    idx -= C * H;
    if (idx < C) return A.b2 + idx;
    idx -= C;
    return A.target + (size_t)g * C + (idx < C ? idx : 0);
  };
  const int HTT = HT + C;
  float hw0 = *haddr((int)threadIdx.x < HTT ? (int)threadIdx.x : 0);
  float hw1 = *haddr((int)threadIdx.x + RT < HTT ? (int)threadIdx.x + RT : 0);
  // ---- consume: validate + stage the edges, park features and weights -------------------------------------
  if (pre) {
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const int e = threadIdx.x + i * RT;
      if (e <= n) { rowptr[e] = prd[i]; rowptr_t[e] = prs[i]; }
      if (e < n) dinv[e] = pdv[i];
      if (e < ne) { col[e] = pcd[i]; col_t[e] = pcs[i]; }
    }
    for (int e = threadIdx.x + EPT * RT; e <= n; e += RT) {
      rowptr[e] = A.pre_rp_d[(size_t)n0 + g + e];
      rowptr_t[e] = A.pre_rp_s[(size_t)n0 + g + e];
      if (e < n) dinv[e] = A.pre_dinv[(size_t)n0 + e];
    }
    for (int e = threadIdx.x + EPT * RT; e < ne; e += RT) {
      col[e] = A.pre_col_d[(size_t)e0 + e];
      col_t[e] = A.pre_col_s[(size_t)e0 + e];
    }
  } else {
    int *ek = ib + Y.ek, *eo = ib + Y.eo;
    bool bad = false;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const int e = threadIdx.x + i * RT;
      if (e < ne) {
        int k = (int)(rld[i] - n0), o_ = (int)(rls[i] - n0);
        if (k < 0 || k >= n || o_ < 0 || o_ >= n) { bad = true; k = -1; o_ = -1; }
        ek[e] = k; eo[e] = o_;
      }
    }
    for (int e = threadIdx.x + EPT * RT; e < ne; e += RT) {
      int k = (int)(A.ll_dst[e0 + e] - n0), o_ = (int)(A.ll_src[e0 + e] - n0);
      if (k < 0 || k >= n || o_ < 0 || o_ >= n) { bad = true; k = -1; o_ = -1; }
      ek[e] = k; eo[e] = o_;
    }
    if (bad && A.flag) atomicOr(A.flag, 2);
  }
#if HSCN_CSR_ELL
  for (int i = threadIdx.x; i <= n; i += RT) { (ib + Y.cntA)[i] = 0; (ib + Y.cntT)[i] = 0; }
#else
  for (int i = threadIdx.x; i <= n; i += RT) { (ib + Y.cursorA)[i] = 0; (ib + Y.cursorT)[i] = 0; }
#endif
  if (threadIdx.x == 0) { (ib + Y.wsum)[0] = 0; (ib + Y.wsum)[1] = 0; (ib + Y.ovf)[0] = 0; }   // fold sign-offs (H = 16 backward), export sign-offs
  {
    float* x0 = buf(0);
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int idx = threadIdx.x + i * RT;
      if (idx < n * H) x0[idx] = xr[i];
    }
    for (int idx = threadIdx.x + XPT * RT; idx < n * H; idx += RT) {
      const int r = idx / H, k = idx - r * H;
      x0[idx] = k < F ? ldf(xl_g, (size_t)(n0 + r) * F + k) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int d = threadIdx.x + i * RT;
      if (d < WTOT) wt[d] = wr[i];
    }
    for (int d = threadIdx.x + WPT * RT; d < WTOT; d += RT) {
      const int l = d / WL, q = d - l * WL;
      const int fin = l == 0 ? F : H;
      const int k = q / H, o = q - k * H;
      wt[d] = q >= H * H ? A.b_ll[l][q - H * H] : (k < fin ? A.W_ll[l][o * fin + k] : 0.f);
    }
    if ((int)threadIdx.x < HTT) headw[threadIdx.x] = hw0;
    if ((int)threadIdx.x + RT < HTT) headw[threadIdx.x + RT] = hw1;
    for (int idx = threadIdx.x + 2 * RT; idx < HTT; idx += RT) headw[idx] = *haddr(idx);
  }
  lds_barrier();
  STAMP(1);
  // ---- structure: the two CSRs of the local->local relation side by side (four barriers each) -------------
  // Molecule-like graphs (every row of at most ELL_D edges) take the two-barrier build of both CSRs; a graph with a
  // denser row takes the general one (four barriers after re-zeroing its counters).  Same arrays either way.
  // H = 16: the structure as 16-byte row records (one LDS load per row in every gather, no prefix sums in the build);
  // HSCN_CSR_ELL=2 keeps the two-barrier CSR build (A/B), H = 32 always takes it (its two-phase backward walks CSRs)
  bool e16 = false;
#if HSCN_CSR_ELL
  if (!pre) {
    bool low;
    if (H == 16 && HSCN_CSR_ELL == 1) {
      low = build_ell16_pair(ib + Y.ek, ib + Y.eo, ne, n, reinterpret_cast<uint4*>(ib + Y.rowptr),
                             reinterpret_cast<uint4*>(ib + Y.recT), dinv, ib + Y.cntA, ib + Y.ellA, ib + Y.cntT,
                             ib + Y.ellT, ib + Y.ovf, RT);
      e16 = low;
    } else {
      low = build_csr_pair_ell(ib + Y.ek, ib + Y.eo, ne, n, rowptr, col, rowptr_t, col_t, dinv, ib + Y.cntA,
                               ib + Y.ellA, ib + Y.cntT, ib + Y.ellT, ib + Y.ovf, RT, wave, NW);
    }
    if (!low) {
      const int NA = NW / 2 > 0 ? NW / 2 : 1;
      const bool inB = wave >= NA && NW > 1;
      if (NW == 1) {
        build_csr_lds(ib + Y.ek, ib + Y.eo, ne, n, rowptr, col, ib + Y.cursorA, ib + Y.tmpA, ALL, true);
        build_csr_lds(ib + Y.eo, ib + Y.ek, ne, n, rowptr_t, col_t, ib + Y.cursorT, ib + Y.tmpT, ALL, true);
        dinv_from_rowptr(rowptr, n, dinv, ALL);
      } else if (!inB) {
        const Grp GA{(int)threadIdx.x, NA * 64, wave, NA};
        build_csr_lds(ib + Y.ek, ib + Y.eo, ne, n, rowptr, col, ib + Y.cursorA, ib + Y.tmpA, GA, true);
        dinv_from_rowptr(rowptr, n, dinv, GA);
      } else {
        const Grp GB{(int)threadIdx.x - NA * 64, (NW - NA) * 64, wave - NA, NW - NA};
        build_csr_lds(ib + Y.eo, ib + Y.ek, ne, n, rowptr_t, col_t, ib + Y.cursorT, ib + Y.tmpT, GB, true);
      }
    }
  }
#else
  if (!pre) {
    const int NA = NW / 2 > 0 ? NW / 2 : 1;
    const bool inB = wave >= NA && NW > 1;
    if (NW == 1) {
      build_csr_lds(ib + Y.ek, ib + Y.eo, ne, n, rowptr, col, ib + Y.cursorA, ib + Y.tmpA, ALL, false);
      build_csr_lds(ib + Y.eo, ib + Y.ek, ne, n, rowptr_t, col_t, ib + Y.cursorT, ib + Y.tmpT, ALL, false);
      dinv_from_rowptr(rowptr, n, dinv, ALL);
    } else if (!inB) {
      const Grp GA{(int)threadIdx.x, NA * 64, wave, NA};
      build_csr_lds(ib + Y.ek, ib + Y.eo, ne, n, rowptr, col, ib + Y.cursorA, ib + Y.tmpA, GA, false);
      dinv_from_rowptr(rowptr, n, dinv, GA);
    } else {
      const Grp GB{(int)threadIdx.x - NA * 64, (NW - NA) * 64, wave - NA, NW - NA};
      build_csr_lds(ib + Y.eo, ib + Y.ek, ne, n, rowptr_t, col_t, ib + Y.cursorT, ib + Y.tmpT, GB, false);
    }
  }
#endif
  lds_barrier();
  STAMP(3);
  // ---- forward layers: A[l] -> A[l + 1], one barrier each ----------------------------------------------------
  // Hand-off of a_1 .. a_{L-1} to the virtual workgroup.  Write-through (sc1) stores need > 2 us to complete, and
  // the publish counter may only be raised behind a wait for them: a wait that every wave performs (between a layer's
  // work and its barrier) stalled the whole workgroup for that long (measured: +3.5 k cycles on layer 1).  So ONE
  // wave of the workgroup does nothing but export while a hand-off is on: in phase l it first waits for the stores
  // it issued in phase l-1 (by then they are a layer old) and raises the counter for them, then copies a_l -- complete
  // since the previous barrier -- from LDS to HBM with 16-byte sc1 stores, then joins the phase's barrier.  Its
  // waiting overlaps the other waves' layer; the compute waves never wait for a store.  (A graph of 354 nodes is 23
  // row tiles: 15 compute waves need the same two rounds as 16.)
  const bool hand = acts_g != nullptr && L >= 2;
  constexpr int NE = NW >= 16 ? 2 : 1;                         // export waves (the last NE of the workgroup)
  const int NC = hand && NW > 1 ? NW - NE : NW;                // compute waves of the forward layers
  const Grp GC{(int)threadIdx.x, NC * 64, wave, NC};
  auto export_rows = [&](const float* src, TS* dst) {          // (one wave)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    const int cnt = n * (H / 4);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dst, 0, cnt * 4 * (int)sizeof(TS), 0x00020000);
    const int ew = NW > 1 ? wave - NC : 0, nes = NW > 1 ? NE : 1;      // this export wave's share
    for (int i0 = ew * 64 + lane; i0 < cnt; i0 += 4 * nes * 64) {   // four LDS reads in flight, then their stores
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * nes * 64;
        v[u] = reinterpret_cast<const float4*>(src)[i < cnt ? i : i0];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * nes * 64;
        if (i >= cnt) continue;
        if constexpr (sizeof(TS) == 4) {
          u32x4 w = {__float_as_uint(v[u].x), __float_as_uint(v[u].y), __float_as_uint(v[u].z), __float_as_uint(v[u].w)};
          __builtin_amdgcn_raw_buffer_store_b128(w, rs, i * 16, 0, 16 /* sc1 */);
        } else {
          half4_t h;
          h.x = (half_t)v[u].x; h.y = (half_t)v[u].y; h.z = (half_t)v[u].z; h.w = (half_t)v[u].w;   // exact: half already
          u32x2 w;
          __builtin_memcpy(&w, &h, 8);
          __builtin_amdgcn_raw_buffer_store_b64(w, rs, i * 8, 0, 16 /* sc1 */);
        }
      }
    }
  };
  // every export wave waits for ITS stores; the counter is raised by one lane once all of them have (they count
  // themselves off in LDS: the last one to arrive raises)
  int* exp_cnt = ib + Y.wsum + 1;
  auto raise = [&](int upto) {   // the stores of a_1 .. a_upto have completed: tell the virtual workgroup
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int last = 1;
    if (NE > 1 && NW > 1) {
      int arrived = 0;
      if (lane == 0) arrived = __hip_atomic_fetch_add(exp_cnt, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
      arrived = __builtin_amdgcn_readfirstlane(arrived);
      last = (arrived % NE) == NE - 1;
    }
    if (last && A.ready && lane == 0)
      __hip_atomic_store(A.ready + g, ep8 + (uint32_t)upto, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  for (int l = 0; l < L; ++l) {
    STAMP(4 + l);
    // the last layer leaves each wave's share of global_mean_pool behind (column sums out of its accumulators)
    float* pool_w = l == L - 1 ? partp + wave * H : nullptr;
    float* pos_w = l == L - 1 ? partp + NW * H + wave * H : nullptr;   // (slot 1 of the partials: positive counts)
    if (wave < NC) {
      if (e16) gcn_fused<H, TS, true>(rowptr, col, dinv, buf(l), wt + l * WL, wt + l * WL + H * H, buf(l + 1), (TS*)nullptr, n, GC, pool_w, pos_w);
      else gcn_fused<H, TS, false>(rowptr, col, dinv, buf(l), wt + l * WL, wt + l * WL + H * H, buf(l + 1), (TS*)nullptr, n, GC, pool_w, pos_w);
    } else if (hand) {
      if (pool_w && lane < H) { pool_w[lane] = 0.f; if (pos_w) pos_w[lane] = 0.f; }
      if (l >= 2) raise(l - 1);
      if (l >= 1) export_rows(buf(l), acts_g + ((size_t)(l - 1) * A.N + n0) * H);
    }
    lds_barrier();
  }
  if (hand && NW == 1) {                                 // a one-wave workgroup exports after its layers
    for (int l = 1; l < L; ++l) export_rows(buf(l), acts_g + ((size_t)(l - 1) * A.N + n0) * H);
    raise(L - 1);
  }
  STAMP(12);
  // ---- head + this graph's row of the loss tail (wave 0) ---------------------------------------------------
  float* aL = buf(L);
#ifdef HSCN_EARLY_RAISE   // (A/B: the last hand-off signing off beside the head -- its wait for the write-through stores held the
                          // head's barrier: 28.38 vs 28.12 us per step, uniform ids 28.40 vs 27.98)
  if (hand && wave >= NC && NW > 1) raise(L - 1);
#endif
  float* pol = vec;          // pooled
  float* zz = vec + 64;      // z = act(lin_1(pooled))
  float* gz = vec + 128;     // dL/d(lin_1 output, pre-activation)
  float* gpool = vec + 192;
  float* gpn = vec + 256;    // gpool / n: the gradient of every unmasked element of the last layer's output
  const float* W1l = headw;
  const float* b1l = headw + H * H;
  const float* W2l = b1l + H;
  const float* b2l = W2l + C * H;
  const float* tgl = b2l + C;
  float* gpl = headw + HT + C;       // g_pred row
  float* ltl = gpl + C;              // loss terms
  // H = 16, C <= 16: the head's five small products run out of REGISTERS -- every LDS word the chain needs (the
  // waves' pool partials, a row and a column of W1 and of W2 per lane, biases, the target row) is requested in one
  // batch, and a vector element crosses lanes through a DPP row broadcast (one VALU operation), not through an LDS
  // round trip with a wavefront fence on either side.  Same fmaf chains in the same order as the general path below.
  const bool fast_head = H == 16 && C <= 16;
  if (threadIdx.x < 64) {
    const float cnt = (float)(n > 0 ? n : 1);
    if (fast_head) {
      constexpr int HH = H <= 16 ? H : 16;
      const bool lh = lane < HH, lc = lane < C;
      const int lk = lh ? lane : 0, lcc = lc ? lane : 0;
      float pw[NW], w1r[HH], w1c[HH], w2r[HH], w2c[16];
#pragma unroll
      for (int w = 0; w < NW; ++w) pw[w] = partp[w * H + lk];
#pragma unroll
      for (int k4 = 0; k4 < HH / 4; ++k4) {
        const float4 a = *reinterpret_cast<const float4*>(W1l + lk * H + 4 * k4);
        w1r[4 * k4] = a.x; w1r[4 * k4 + 1] = a.y; w1r[4 * k4 + 2] = a.z; w1r[4 * k4 + 3] = a.w;
        const float4 b = *reinterpret_cast<const float4*>(W2l + lcc * H + 4 * k4);
        w2r[4 * k4] = b.x; w2r[4 * k4 + 1] = b.y; w2r[4 * k4 + 2] = b.z; w2r[4 * k4 + 3] = b.w;
      }
#pragma unroll
      for (int o = 0; o < HH; ++o) w1c[o] = W1l[o * H + lk];
#pragma unroll
      for (int c = 0; c < 16; ++c) w2c[c] = c < C ? W2l[c * H + lk] : 0.f;
      const float bb1 = b1l[lk], bb2 = b2l[lcc], tgt = tgl[lcc];
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += pw[w];
      const float polv = s / cnt;                                    // lane k: pooled[k]
      const float zv = apply_act(row_dot<HH>(polv, w1r, 0.f) + bb1, A.head_act);   // lane o: z[o]
      const float pc = row_dot<HH>(zv, w2r, 0.f) + bb2;               // lane c: pred[c]
      float lt, sg, gg;
      criterion_elem(A.loss_kind, pc, tgt, A.inv_count, lt, sg, gg);   // same element code as k_criterion
      if (lc) {
        A.pred[(size_t)g * C + lane] = pc;
        if (A.score) A.score[(size_t)g * C + lane] = sg;             // (= 1 / (1 + exp(-pred)), criterion_elem's)
      }
      lt = lc ? lt : 0.f;                                             // (lanes past C: exact zeros in the sums below)
      gg = lc ? gg : 0.f;
      const float sl = row_seq_sum<16>(lt, 0.f);                      // loss terms in class order
      const float gzv = row_dot<16>(gg, w2c, 0.f) * act_grad_from_output(zv, A.head_act);   // lane o
      const float gpa = row_dot<HH>(gzv, w1c, 0.f);                   // lane k: d loss / d pooled[k]
      if (lh) { pol[lane] = polv; zz[lane] = zv; gz[lane] = gzv; gpool[lane] = gpa; gpn[lane] = gpa / cnt; }
      if (lc) gpl[lane] = gg;
      if (lane == 0) part[A.Pn] = sl;
    } else {
    if (lane < H) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += partp[w * H + lane];
      s = s / cnt;
      pol[lane] = s;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    if (lane < H) {
      float a1 = 0.f;
      const float4* wrow = reinterpret_cast<const float4*>(W1l + lane * H);
#pragma unroll
      for (int k4 = 0; k4 < H / 4; ++k4) {
        const float4 w4 = wrow[k4];
        const float4 p4 = *reinterpret_cast<const float4*>(pol + 4 * k4);
        a1 = fmaf(p4.x, w4.x, a1);
        a1 = fmaf(p4.y, w4.y, a1);
        a1 = fmaf(p4.z, w4.z, a1);
        a1 = fmaf(p4.w, w4.w, a1);
      }
      a1 += b1l[lane];
      a1 = apply_act(a1, A.head_act);
      zz[lane] = a1;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    for (int c = lane; c < C; c += 64) {
      float a2 = 0.f;
      const float4* wrow = reinterpret_cast<const float4*>(W2l + c * H);
#pragma unroll
      for (int k4 = 0; k4 < H / 4; ++k4) {
        const float4 w4 = wrow[k4];
        const float4 z4 = *reinterpret_cast<const float4*>(zz + 4 * k4);
        a2 = fmaf(z4.x, w4.x, a2);
        a2 = fmaf(z4.y, w4.y, a2);
        a2 = fmaf(z4.z, w4.z, a2);
        a2 = fmaf(z4.w, w4.w, a2);
      }
      const float pc = a2 + b2l[c];
      float lt, sg, gg;
      criterion_elem(A.loss_kind, pc, tgl[c], A.inv_count, lt, sg, gg);   // same element code as k_criterion
      gpl[c] = gg;
      ltl[c] = lt;
      A.pred[(size_t)g * C + c] = pc;
      if (A.score) A.score[(size_t)g * C + c] = sg;
    }
    // ---- head backward, first half (same wave: no barrier needed up to gz) ----
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    if (lane < H) {
      float acc = 0.f;
      for (int c = 0; c < C; ++c) acc = fmaf(gpl[c], W2l[c * H + lane], acc);
      gz[lane] = acc * act_grad_from_output(zz[lane], A.head_act);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    if (lane < H) {   // d loss / d pooled
      float acc = 0.f;
#pragma unroll
      for (int o = 0; o < H; ++o) acc = fmaf(gz[o], W1l[o * H + lane], acc);
      gpool[lane] = acc;
      gpn[lane] = acc / cnt;
    }
    }
  }
  lds_barrier();
  STAMP(13);
#ifndef HSCN_EARLY_RAISE
  // the last hand-off signs off HERE, behind the head's barrier: the export waves own one row tile of the first backward
  // layer where the first waves own two, so their wait for the write-through stores of a_{L-1} rides on slack
  if (hand && wave >= NC && NW > 1) raise(L - 1);
#endif

  // ================================ backward ======================================================
  // partial layout: per layer {W_ll [H*fin], b_ll [H]}, then W1 [H*H], b1 [H], W2 [C*H], b2 [C], loss column
  int off_head = 0;
  for (int l = 0; l < L; ++l) off_head += H * (l == 0 ? F : H) + H;
  const int oW1 = off_head, ob1 = oW1 + H * H, oW2 = ob1 + H, ob2 = oW2 + C * H;
  for (int idx = threadIdx.x; idx < C * H; idx += RT) {
    const int c = idx / H, k = idx - c * H;
    part[oW2 + idx] = gpl[c] * zz[k];
  }
  for (int c = threadIdx.x; c < C; c += RT) part[ob2 + c] = gpl[c];
  if (!fast_head && threadIdx.x == RT - 64) {   // the graph's loss terms, summed in class order
    float sl = 0.f;
    for (int c = 0; c < C; ++c) sl += ltl[c];
    part[A.Pn] = sl;
  }
  for (int idx = threadIdx.x; idx < H * H; idx += RT) {
    const int o = idx / H, k = idx - o * H;
    part[oW1 + idx] = gz[o] * pol[k];
  }
  if (threadIdx.x < H) part[ob1 + threadIdx.x] = gz[threadIdx.x];
  float* G = aL;                  // gradient of the current layer's output, masked in place
  float* GH = L >= 2 ? buf(0) : buf(2);
  if constexpr (H != 16) {
    const float cnt = (float)(n > 0 ? n : 1);
    for (int idx = threadIdx.x; idx < n * H; idx += RT) G[idx] = G[idx] > 0.f ? gpool[idx % H] / cnt : 0.f;
  }
  STAMP(14);
  float* bred = partp;
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
  if constexpr (H == 16) {
    // ---- a backward layer in ONE phase (one barrier): a wave owns 16-row tiles.  For a tile it gathers its rows of
    // GH = A_hat^T G (source-keyed CSR, edge order, separately rounded: the same values as the two-phase backward),
    // takes the input gradient (GH W_l, masked) off the matrix cores and writes it over the tile's rows of the layer
    // input a_l IN PLACE -- those rows are read by this wave alone (mask, weight-gradient operand), and nobody
    // gathers from a_l's buffer before the next barrier -- and accumulates its part of the weight gradient
    // GH^T a_l through a wave-private 16 x 16 transposition scratch.  The 16 partial weight-gradient tiles are
    // folded after the barrier, as before; the folders sign off in an LDS counter that the next layer's writers
    // of the partial-tile buffer check (it has always been reached by then: no wait in practice, no race in theory).
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    int* fold_cnt = ib + Y.wsum;
    constexpr int NFW = NW < 4 ? NW : 4;            // waves that take part in a fold (threads 0 .. 255)
    float* ght = buf(0);                            // dead after forward layer 0 (the features are re-read from HBM)
    float* Gc = aL;
    const int li = lane & 15, lj = lane >> 4;
    const int ntile = (n + 15) >> 4;
    int pend_ob = 0;
    // The gradient of the last layer's output is never materialised: G_L[j][k] = a_L[j][k] > 0 ? gpool[k] / n : 0 is
    // applied where the first backward layer reads it (the same values as the masking pass of the launch pair).
    const float4 gq4 = *reinterpret_cast<const float4*>(gpn + 4 * lj);            // this lane's feature quarter (tile gather)
    const float4 gb4 = *reinterpret_cast<const float4*>(gpn + 4 * (lane & 3));    // (bias sums)
    auto gl4 = [](const float4 a, const float4 q) {
      return make_float4(a.x > 0.f ? q.x : 0.f, a.y > 0.f ? q.y : 0.f, a.z > 0.f ? q.z : 0.f, a.w > 0.f ? q.w : 0.f);
    };
    for (int it = 0, l = L - 1; l >= 0; --l, ++it) {
      const int fin = l == 0 ? F : H;
      off -= H * fin + H;
      const int oW = off, ob = off + H * fin;
      // bias-gradient partials of the layer of iteration j live in slot j % 3: iteration `it` folds slot it - 1, fills slot
      // `it` (it = 0 only: the pass below) and -- in its tiles' epilogue -- slot it + 1 for the next layer
      float* bredw = bred + (it % 3) * NW * H;
      float* bredn = bred + ((it + 1) % 3) * NW * H;
      float bsum = 0.f;       // this lane's share of the NEXT layer's bias gradient: column li of the rows it writes
      if (it > 0) lds_barrier();  // G complete; the previous layer's partial tiles and bias partials are in LDS
                                   // (it = 0: nothing was written to LDS since the barrier behind the head)
      if (pend_oW >= 0) {
        // (the folding threads are the workgroup's LAST ones: with 23 row tiles on 16 waves the first seven waves own
        // two tiles, the last ones one)
        const int ft = (int)threadIdx.x - (RT - NFW * 64);
        if (ft >= 0) {
          const int e_ = ft & 255;
          float s_ = 0.f;
#pragma unroll
          for (int r = 0; r < NW; ++r) s_ += red[r * 256 + e_];
          const int oo = e_ >> 4, kk = e_ & 15;
          if (kk < pend_fin) part[pend_oW + oo * pend_fin + kk] = s_;
        }
        if ((int)threadIdx.x >= RT - H) {
          const int c_ = (int)threadIdx.x - (RT - H);
          const float* bp = bred + ((it - 1) % 3) * NW * H;
          float sb = 0.f;
#pragma unroll
          for (int w = 0; w < NW; ++w) sb += bp[w * H + c_];
          part[pend_ob + c_] = sb;
        }
        if (wave >= NW - NFW) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          if (lane == 0) __hip_atomic_fetch_add(fold_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
      if (it == 0 && HSCN_BIAS0_COUNTS) {
        // first backward layer: G[row][col] = a_L[row][col] > 0 ? gpn[col] : 0, so its column sum is gpn[col] times the
        // number of positive entries of the column -- which the forward's last layer counted in its epilogue (slot 1)
        if (lane < H) bredw[wave * H + lane] = (partp + NW * H)[wave * H + lane] * gpn[lane];
      } else if (it == 0) {   // (-DHSCN_BIAS0_COUNTS=0: the row walk) bias gradient = column sums of G; later
                       // layers' sums were left by the previous layer's tile epilogues, see `bsum`
        const int slot = lane >> 2, f = (lane & 3) * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = wave * 16 + slot; i < n; i += NW * 16) {
          float4 v = *reinterpret_cast<const float4*>(Gc + i * H + f);
          if (it == 0) v = gl4(v, gb4);
          acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        acc.x += __shfl_xor(acc.x, 32, 64); acc.y += __shfl_xor(acc.y, 32, 64);
        acc.z += __shfl_xor(acc.z, 32, 64); acc.w += __shfl_xor(acc.w, 32, 64);
        acc.x += __shfl_xor(acc.x, 16, 64); acc.y += __shfl_xor(acc.y, 16, 64);
        acc.z += __shfl_xor(acc.z, 16, 64); acc.w += __shfl_xor(acc.w, 16, 64);
        acc.x = row_ror_add<8>(acc.x); acc.y = row_ror_add<8>(acc.y);
        acc.z = row_ror_add<8>(acc.z); acc.w = row_ror_add<8>(acc.w);
        acc.x = row_ror_add<4>(acc.x); acc.y = row_ror_add<4>(acc.y);
        acc.z = row_ror_add<4>(acc.z); acc.w = row_ror_add<4>(acc.w);
        if (slot == 0) *reinterpret_cast<float4*>(bredw + wave * H + f) = acc;
      }
      float* Xl = buf(l);                           // the layer input a_l (l >= 1); becomes G_l tile by tile
      const float* Wl = wt + l * WL;
      float bw[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) bw[s] = l > 0 ? Wl[li * H + 4 * lj + s] : 0.f;   // W_l[4 lj + s][li]
      f32x4 accw = {0.f, 0.f, 0.f, 0.f};
      float* sc = ght + wave * 256;
      const float* gq = Gc + 4 * lj;
      auto xrows = [&](int rt_, float (&bx)[4]) {   // l == 0: the features, straight from HBM into the operand registers
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int row = rt_ * 16 + 4 * c + lj;
          const bool okx = row < n && li < fin;
          const float tx = ldf(xl_g, okx ? (size_t)(n0 + row) * fin + li : 0);
          bx[c] = okx ? tx : 0.f;
        }
      };
      auto finish = [&](int rt_, const float4 z, const float (&bx)[4]) {
        const int r0 = rt_ * 16;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // the previous tile's reads of the scratch are done
        *reinterpret_cast<float4*>(sc + li * 16 + 4 * lj) = z;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (l > 0) {
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(z.x, bw[0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(z.y, bw[1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(z.z, bw[2], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(z.w, bw[3], acc, 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // the tile is in the scratch before it is read back
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int row = r0 + 4 * c + lj;
          const float av = sc[(4 * c + lj) * 16 + li];                       // GH[row][o = li]
          const float bv = l == 0 ? bx[c] : (row < n ? Xl[row * H + li] : 0.f);   // a_l[row][k = li]
          accw = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, accw, 0, 0, 0);
        }
        if (l > 0) {
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // this tile's rows of a_l have been read: overwrite them
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = r0 + lj * 4 + r;
            if (row < n) {
              const int idx = row * H + li;
              const float gnew = Xl[idx] > 0.f ? acc[r] : 0.f;
              Xl[idx] = gnew;
              bsum += gnew;      // column li of G_l: the bias gradient of layer l - 1
            }
          }
        }
      };
      for (int rt = wave; rt < ntile; rt += NW) {
        float bxA[4] = {0.f, 0.f, 0.f, 0.f};
        if (l == 0) xrows(rt, bxA);
        const int i = rt * 16 + li;
        float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e16) {
          if (i < n) {
            const uint4 rec = reinterpret_cast<const uint4*>(ib + Y.recT)[i];
            const float di = dinv[i];
            const int c = (int)(rec.w >> 16);
            const int jr[8] = {(int)(rec.x & 0xffffu), (int)(rec.x >> 16), (int)(rec.y & 0xffffu), (int)(rec.y >> 16),
                               (int)(rec.z & 0xffffu), (int)(rec.z >> 16), (int)(rec.w & 0xffffu), (int)(rec.w & 0xffffu)};
#pragma unroll
            for (int p = 0; p < 8; p += 4) {
              if (p < c) {
                float ww[4];
                float4 vv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                  ww[u] = mul_rn(dinv[jr[p + u]], di);
                  vv[u] = *reinterpret_cast<const float4*>(gq + jr[p + u] * H);
                  if (it == 0) vv[u] = gl4(vv[u], gq4);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                  if (p + u < c) {
#ifndef HSCN_BWD_SEPARATE   // fused multiply-add, as the forward's gather (26.82 -> 26.54 us per step); -DHSCN_BWD_SEPARATE:
                            // product and sum rounded separately, the launch pair's values
                    z.x = fmaf(ww[u], vv[u].x, z.x); z.y = fmaf(ww[u], vv[u].y, z.y);
                    z.z = fmaf(ww[u], vv[u].z, z.z); z.w = fmaf(ww[u], vv[u].w, z.w);
#else
                    z.x = add_rn(z.x, mul_rn(ww[u], vv[u].x));
                    z.y = add_rn(z.y, mul_rn(ww[u], vv[u].y));
                    z.z = add_rn(z.z, mul_rn(ww[u], vv[u].z));
                    z.w = add_rn(z.w, mul_rn(ww[u], vv[u].w));
#endif
                  }
                }
              }
            }
          }
        } else if (i < n) {
          const int s0 = rowptr_t[i], t0 = rowptr_t[i + 1];
          const float di = dinv[i];
          for (int p = s0; p < t0; p += 4) {
            int jj[4];
            float ww[4];
            float4 vv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) jj[u] = col_t[p + u < t0 ? p + u : t0 - 1];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              ww[u] = mul_rn(dinv[jj[u]], di);
              vv[u] = *reinterpret_cast<const float4*>(gq + jj[u] * H);
              if (it == 0) vv[u] = gl4(vv[u], gq4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              if (p + u < t0) {
#ifndef HSCN_BWD_SEPARATE
                z.x = fmaf(ww[u], vv[u].x, z.x); z.y = fmaf(ww[u], vv[u].y, z.y);
                z.z = fmaf(ww[u], vv[u].z, z.z); z.w = fmaf(ww[u], vv[u].w, z.w);
#else
                z.x = add_rn(z.x, mul_rn(ww[u], vv[u].x));
                z.y = add_rn(z.y, mul_rn(ww[u], vv[u].y));
                z.z = add_rn(z.z, mul_rn(ww[u], vv[u].z));
                z.w = add_rn(z.w, mul_rn(ww[u], vv[u].w));
#endif
              }
            }
          }
        }
        finish(rt, z, bxA);
      }
      if (it > 0) {   // the folders of the previous layer's partial tiles have signed off (see above)
        while (__hip_atomic_load(fold_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < it * NFW)
          __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave * 256 + (lj * 4 + r) * 16 + li] = accw[r];
      if (l > 0) {     // the wave's column sums of the G it has written (its tiles' rows): fold the four row groups
        bsum += __shfl_xor(bsum, 16, 64);
        bsum += __shfl_xor(bsum, 32, 64);
        if (lj == 0) bredn[wave * H + li] = bsum;
      }
      pend_oW = oW; pend_fin = fin; pend_ob = ob;
      Gc = Xl;
      STAMP(16 + 3 * l);
    }
    lds_barrier();
    fold_gw(0, pend_oW, pend_fin);
    if (threadIdx.x < H) {
      const float* bp = bred + ((L - 1) % 3) * NW * H;
      float sb = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) sb += bp[w * H + threadIdx.x];
      part[pend_ob + threadIdx.x] = sb;
    }
    STAMP(61);
    return;
  }
  for (int l = L - 1; l >= 0; --l) {
    const int fin = l == 0 ? F : H;
    off -= H * fin + H;
    const int oW = off, ob = off + H * fin;
    lds_barrier();  // G (masked) complete
    if (GW1 && pend_oW >= 0) fold_gw(0, pend_oW, pend_fin);
    // layer input: A[l] is still in LDS for l >= 1 (and for l = 0 when L = 1); with L >= 2 the features were
    // overwritten by GH: re-read them (requested here, parked in A[1] -- dead by now -- after the gather-reduce)
    const bool reload = l == 0 && L >= 2;
    float* X = reload ? buf(1) : buf(l);
    float xq[XPT];
    if (reload) {
#pragma unroll
      for (int i = 0; i < XPT; ++i) {
        const int idx = threadIdx.x + i * RT;
        const int r = idx / H, k = idx - r * H;
        xq[i] = (idx < n * H && k < fin) ? ldf(xl_g, (size_t)(n0 + r) * fin + k) : 0.f;
      }
    }
    {   // bias gradient = column sums of G
      constexpr int LQ = H / 4;
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
    // dL/d(transform output) = A_hat^T G  (source-keyed CSR, edge order)
    agg_gcn_lds<H, float>(rowptr_t, col_t, dinv, dinv, G, nullptr, GH, n, 0, (float*)nullptr, ALL);
    if (reload) {
#pragma unroll
      for (int i = 0; i < XPT; ++i) {
        const int idx = threadIdx.x + i * RT;
        if (idx < n * H) X[idx] = xq[i];
      }
      for (int idx = threadIdx.x + XPT * RT; idx < n * H; idx += RT) {
        const int r = idx / H, k = idx - r * H;
        X[idx] = k < fin ? ldf(xl_g, (size_t)(n0 + r) * fin + k) : 0.f;
      }
    }
    lds_barrier();
    STAMP(16 + 3 * l);
    if (threadIdx.x < H) {   // bias gradient: fold the waves' column sums in wave order
      float sb = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) sb += bred[w * H + threadIdx.x];
      part[ob + threadIdx.x] = sb;
    }
    {   // weight gradient gW[o][k] = sum_j GH[j][o] X[j][k] on the matrix cores
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
          if (t0 + TPP < NT) lds_barrier();
        }
      }
    }
    STAMP(17 + 3 * l);
    // input gradient: G[j][k] = relu'(a_l[j][k]) * sum_o GH[j][o] W_l[o][k]; the forward's transposed W_l is read
    // transposed again (a_l = X doubles as the mask; the old G is dead: GH holds what is needed of it)
    if (l > 0) lin_mfma_wt_masked<H>(GH, wt + l * WL, G, n, X, ALL);
    STAMP(18 + 3 * l);
  }
  if (GW1 && pend_oW >= 0) {
    lds_barrier();
    fold_gw(0, pend_oW, pend_fin);
  }
  STAMP(61);
}

// grid = B (no virtual branch) or 2B: blocks [0, B) run the local program, blocks [B, 2B) the virtual branch
// VMODE: 5 = the virtual program sizes its LDS layout from the arguments, 6 = from StepVCaps (compile time)
template <int H, int RT, typename TS, int VMODE = 5>
__global__ void __launch_bounds__(RT) k_hscn_step(const StepArgs S, const FwdArgs V) {
  if ((int)blockIdx.x < S.B) hscn_step_local<H, RT, TS>(S, blockIdx.x);
  else hscn_fwd_body<H, RT, VMODE, TS>(V, (int)blockIdx.x - S.B);
}
template <int H, int RT, typename TS>
__global__ void __launch_bounds__(RT) k_hscn_step_local(const StepArgs S) {
  hscn_step_local<H, RT, TS>(S, blockIdx.x);
}

inline size_t step_lds_bytes(int H, int L, int C, int max_n, int max_ell) {
  return step_layout(H, L, C, max_n, max_ell).total * 4;
}

template <int H, int RT, typename TS, int VMODE = 5>
int launch_step_rt(const StepArgs& S, const FwdArgs* V, size_t lds, hipStream_t st) {
  if (V) {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)k_hscn_step<H, RT, TS, VMODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k_hscn_step<H, RT, TS, VMODE><<<(unsigned)(2 * S.B), RT, lds, st>>>(S, *V);
  } else {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)k_hscn_step_local<H, RT, TS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds);
    k_hscn_step_local<H, RT, TS><<<(unsigned)S.B, RT, lds, st>>>(S);
  }
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

template <int H, typename TS>
int launch_step(StepArgs& S, FwdArgs* V, hipStream_t st) {
  size_t lds = step_lds_bytes(H, S.L, S.C, S.max_n, S.max_ell);
  if (lds > 160 * 1024) return HSCN_E_UNSUPPORTED;
  bool fixed_layout = false;
  if (V) {
    V->spec = 1; V->exp = 0; V->exp_dinv = 0;
    size_t lv = pick_fwd_lds(*V, H);
    if (lv > 160 * 1024) return HSCN_E_UNSUPPORTED;
    // 16-wave workgroups on graphs within StepVCaps: the virtual program with its LDS layout fixed at compile time
    // (hscn_fwd_body MODE 6: the layout's ~40 offsets are immediates instead of scalar registers -- 90 SGPR spills
    // become 9); HSCN_STEP_FIXED_LAYOUT=0 keeps the run-time layout (A/B, and what larger graphs / more clusters take)
    static const bool allow_fixed = !(getenv("HSCN_STEP_FIXED_LAYOUT") && atoi(getenv("HSCN_STEP_FIXED_LAYOUT")) == 0);
    if (allow_fixed && S.max_n > 64 && V->max_n <= StepVCaps::N && V->max_v <= StepVCaps::V && V->max_evv <= StepVCaps::EVV &&
        V->l_begin == 0 && V->l_end == V->L) {
      const size_t lf = fwd_layout(H, 1, StepVCaps::N, StepVCaps::V, 0, StepVCaps::EVV, 1, 0).total * 4;
      if (lf <= 160 * 1024) { fixed_layout = true; lv = lf; V->db = 1; }
    }
    if (lv > lds) lds = lv;
    if (S.max_n > 64) {
      // 16-wave workgroups: one per CU (the acquire-free consumer form is measured for exactly that): ask for more
      // than half of a CU's LDS
      if (lds < 81 * 1024) lds = 81 * 1024;
    } else {
      V->acq = 1;   // small graphs share CUs: the consumer acquires and uses plain loads (Guideline 16, R1 as written)
    }
  }
  if (S.max_n <= 64) return launch_step_rt<H, 256, TS>(S, V, lds, st);
  if (fixed_layout) return launch_step_rt<H, 1024, TS, 6>(S, V, lds, st);
  return launch_step_rt<H, 1024, TS>(S, V, lds, st);
}
