#!/usr/bin/env python3
"""Stage A (MinCUT spectral clustering net) throughput: gcn_norm + SCN forward + (mc+o) backward.

  batched   : B graphs as one block-diagonal batch per step (extension, shards like stage C)
  per-graph : the reference's trajectory -- one forward/backward per graph
              (train/train_clustering.py:36-50), here without the optimizer step
The CPU oracle runs the per-graph loop for comparison (notebook: 317-645 graphs/s, unknown HW).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-hscn_amd")]
import torch

from graph_hscn.data import Batch
from graph_hscn.loader.synthetic import make_dataset
from graph_hscn.model.hscn import SCN
from graph_hscn.nn import gcn_norm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--clusters", type=int, default=16)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--cpu", action="store_true")
    a = ap.parse_args()
    dev = "cuda"
    graphs = make_dataset("peptides_func", a.batch, seed=0)
    torch.manual_seed(0)
    model = SCN([16], "elu", 9, a.clusters).to(dev)
    big = Batch.from_data_list(graphs)
    x = big.x.to(dev).float()
    ei0 = big.edge_index.to(dev)
    ptr = big.ptr.to(dev).to(torch.int32)

    def step_batched():
        for p in model.parameters():
            p.grad = None
        ei, ew = gcn_norm(ei0, None, big.num_nodes, add_self_loops=True)
        _, mc, o, _ = model(x, ei, ew, node_ptr=ptr)
        (mc + o).backward()

    def timeit(fn, n):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    out = {}
    t = timeit(step_batched, a.steps)
    out["batched_eager"] = {"ms_per_step": t * 1e3, "graphs_per_s": a.batch / t}
    gs = [(g.x.to(dev).float(), g.edge_index.to(dev), g.num_nodes) for g in graphs[:32]]

    def step_per_graph():
        for xg, eg, n in gs:
            for p in model.parameters():
                p.grad = None
            ei, ew = gcn_norm(eg, None, n, add_self_loops=True)
            _, mc, o, _ = model(xg, ei, ew)
            (mc + o).backward()

    t = timeit(step_per_graph, 3)
    out["per_graph_eager"] = {"ms_per_graph": t * 1e3 / len(gs), "graphs_per_s": len(gs) / t}
    # fused graph-resident path (gcn_norm folded in): batched and per-graph
    bigd = big.to(dev)
    bigd.x = bigd.x.float()

    def step_batched_fused():
        for p in model.parameters():
            p.grad = None
        _, mc, o = model.forward_graphs(bigd)
        (mc + o).backward()

    if model.resident_ok(bigd):
        t = timeit(step_batched_fused, a.steps)
        out["batched_fused_eager"] = {"ms_per_step": t * 1e3, "graphs_per_s": a.batch / t}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step_batched_fused()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        cg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg):
            step_batched_fused()
        t = timeit(cg.replay, 200)
        out["batched_fused_hipgraph"] = {"ms_per_step": t * 1e3, "graphs_per_s": a.batch / t}
        gdev = [g.to(dev) for g in graphs[:32]]
        for g in gdev:
            g.x = g.x.float()

        def step_per_graph_fused():
            for g in gdev:
                for p in model.parameters():
                    p.grad = None
                _, mc, o = model.forward_graphs(g)
                (mc + o).backward()

        t = timeit(step_per_graph_fused, 5)
        out["per_graph_fused_eager"] = {"ms_per_graph": t * 1e3 / len(gdev), "graphs_per_s": len(gdev) / t}
    if a.cpu:
        from oracle import models as OM
        om = OM.SCN([16], "elu", 9, a.clusters)
        torch.set_num_threads(8)
        t0 = time.perf_counter()
        for g in graphs[:32]:
            om.zero_grad()
            _, mc, o, *_ = OM.scn_step_single_graph(om, g.x, g.edge_index)
            (mc + o).backward()
        out["cpu_oracle_per_graph"] = {"graphs_per_s": 32 / (time.perf_counter() - t0)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
