#!/usr/bin/env python3
"""Dense MinCUT route on the matrix cores at the PascalVOC-SP shape (BASELINE config 4):
B = 128 graphs, n = 479 superpixels, K = 64 clusters, dense [B,n,n] adjacency.
FLOPs per graph = 2Kn^2 + 2nK^2 (33.3 MFLOP, 4.27 GFLOP per batch) for
S^T A S alone; the forward also does S^T S and S^T X, the backward adds A^T S.
Reports forward / forward+backward time and the fp32-MFMA rate against the 157.3 TFLOP/s peak.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-hscn_amd")]
import torch

from graph_hscn.nn import dense_mincut_pool


def main(B=128, n=479, K=64, F=16, iters=20):
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    adj = (torch.rand(B, n, n, generator=g) < 5.65 / n).float()
    adj = ((adj + adj.transpose(1, 2)) > 0).float() + torch.eye(n)
    adj = adj.to(dev)
    x = torch.randn(B, n, F, generator=g).to(dev)
    s = torch.randn(B, n, K, generator=g).to(dev).requires_grad_()

    def fwd():
        return dense_mincut_pool(x, adj, s)

    def fwd_bwd():
        s.grad = None
        _, _, mc, o = fwd()
        (mc + o).backward()

    def timeit(fn):
        for _ in range(3):
            fn()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for a, b in ev:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) * 1e-3 for a, b in ev)
        return ts[len(ts) // 2]

    tf, tfb = timeit(fwd), timeit(fwd_bwd)
    f_as = 2.0 * n * n * K            # A S
    f_sas = 2.0 * K * n * K           # S^T (A S)
    f_ss = 2.0 * K * n * K
    f_sx = 2.0 * K * n * F
    fl_f = B * (f_as + f_sas + f_ss + f_sx)
    fl_fb = fl_f + B * f_as           # + A^T S
    print(json.dumps({"shape": {"B": B, "n": n, "K": K, "F": F}, "fwd_us": tf * 1e6, "fwd_bwd_us": tfb * 1e6,
                      "fwd_TFLOPs": fl_f / tf / 1e12, "fwd_bwd_TFLOPs": fl_fb / tfb / 1e12,
                      "peak_fp32_mfma_TFLOPs": 157.3, "fwd_frac": fl_f / tf / 157.3e12,
                      "graphs_per_s_fwd_bwd": B / tfb, "adj_bytes": B * n * n * 4,
                      "fwd_adj_GBs": B * n * n * 4 / tf / 1e9}))


if __name__ == "__main__":
    main()
