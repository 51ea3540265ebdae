#!/usr/bin/env python3
"""Streaming (layered-engine) ll SpMM at a bandwidth-resident shape.

A Peptides-func batch of 128 graphs is 3 MB per launch -- far below
one launch's latency at 8 TB/s -- so the HBM roofline of `k_spmm` is measured on
the same generator scaled up: the 128-graph block-diagonal batch tiled T times
(B = 128*T graphs) at hidden width H.  Algorithmic bytes = 4(N+1)+4E+4N+8NH.

  python tools/bench_spmm.py [--tiles 32] [--hidden 128] [--iters 30]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-hscn_amd")]
import torch

from graph_hscn import _hip
from graph_hscn.data import Batch
from graph_hscn.loader.synthetic import make_dataset
from graph_hscn.nn import functional as Fh
from graph_hscn.structure import Relation


def scaled_relation(tiles, seed=0, dev="cuda"):
    big = Batch.from_data_list(make_dataset("peptides_func", 128, seed=seed))
    n, ei = big.num_nodes, big.edge_index.to(dev)
    offs = (torch.arange(tiles, device=dev) * n).view(tiles, 1, 1)
    ei_t = (ei.unsqueeze(0) + offs).permute(1, 0, 2).reshape(2, -1).contiguous()
    return Relation(ei_t, n * tiles, n * tiles), n * tiles, ei_t.size(1)


def _median_s(fn, iters):
    for _ in range(3):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in ev:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) * 1e-3 for s, e in ev)
    return ts[len(ts) // 2]


def measure(tiles, H, iters, dev="cuda"):
    """Forward pass (target-keyed CSR, + bias + ReLU) and backward pass (the same kernel on the source-keyed CSR,
    gradient rows in, no epilogue): SURVEY.md 8(d) defines the SpMM roofline over the pair,
    ``2 * ll_bytes / (t_fwd + t_bwd)``."""
    rel, N, E = scaled_relation(tiles, dev=dev)
    rel.check()
    h = torch.randn(N, H, device=dev)
    g = torch.randn(N, H, device=dev)
    bias = torch.randn(H, device=dev)
    dinv = rel.dinv
    csr_t = rel.csr_t
    t_f = _median_s(lambda: Fh.spmm_gcn_raw(rel.csr, dinv, dinv, h, bias, 1), iters)
    t_b = _median_s(lambda: Fh.spmm_gcn_raw(csr_t, dinv, dinv, g, None, 0), iters)
    alg = 4 * (N + 1) + 4 * E + 4 * N + 8 * N * H
    return {"kernel": "k_spmm<4,0> (hscn_spmm_csr_gcn): fwd (+ bias + ReLU) and bwd (source-keyed CSR)",
            "graphs": 128 * tiles, "nodes": N, "edges": E, "hidden": H, "algorithmic_bytes": alg,
            "median_us": t_f * 1e6, "achieved_GBs": alg / t_f / 1e9, "frac_of_8TBs": alg / t_f / 8e12,
            "bwd_median_us": t_b * 1e6, "bwd_achieved_GBs": alg / t_b / 1e9, "bwd_frac_of_8TBs": alg / t_b / 8e12,
            "fwd_plus_bwd_achieved_GBs": 2 * alg / (t_f + t_b) / 1e9,
            "fwd_plus_bwd_frac_of_8TBs": 2 * alg / (t_f + t_b) / 8e12}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=32)
    ap.add_argument("--hidden", type=int, nargs="+", default=[16, 32, 64, 128])
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    _hip.lib()
    for H in a.hidden:
        print(json.dumps(measure(a.tiles, H, a.iters)))
