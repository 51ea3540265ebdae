#!/usr/bin/env python3
"""BASELINE config 1 (the MPNN GCN baseline, configs/GCN/peptides_func_GCN.yaml: hidden 16, 3 layers,
dropout 0.2, batch 32) -- forward + criterion + backward per step on the HIP path, beside the CPU oracle
on this box's host cores.  A parity-case measurement, not the bench line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-hscn_amd")]
import torch

import bench
from graph_hscn.config.config import MPNNConfig
from graph_hscn.data import Batch
from graph_hscn.loader.synthetic import make_dataset
from graph_hscn.loss import criterion
from graph_hscn.model.mpnn import build_mpnn
from oracle import models as OM


def main():
    out = {}
    for B in (1, 32, 128, 1024):
        b = Batch.from_data_list(make_dataset("peptides_func", B, seed=0))
        y = (torch.rand(B, 10, generator=torch.Generator().manual_seed(0)) < 0.2).float()
        d = b.to("cuda")
        d.x = d.x.float()
        yd = y.to("cuda")
        torch.manual_seed(0)
        m = build_mpnn(MPNNConfig("gcn", "relu"), 9, 10).to("cuda").train()

        def step():
            for p in m.parameters():
                p.grad = None
            loss, _ = criterion("cross_entropy", m(d), yd)
            loss.backward()

        for _ in range(10):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 100
        for _ in range(K):
            step()
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / K
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            g.replay()
        torch.cuda.synchronize()
        rep = (time.perf_counter() - t0) / K
        row = {"eager_ms": eager * 1e3, "replay_ms": rep * 1e3, "replay_graphs_per_s": B / rep}
        if B <= 128:
            nt = min(4, bench.host_cores())      # tiny ops: a few intra-op threads beat all cores (bench.py sweep)
            torch.set_num_threads(nt)
            om = OM.MPNN(OM.ACT["relu"], 9, 16, 10, 3, 0.2).train()
            x = b.x.float()

            def cstep():
                om.zero_grad(set_to_none=True)
                loss, _ = OM.criterion("cross_entropy", om(x, b.edge_index, b.batch, B), y)
                loss.backward()

            for _ in range(3):
                cstep()
            t0 = time.perf_counter()
            for _ in range(20):
                cstep()
            c = (time.perf_counter() - t0) / 20
            row.update(cpu_oracle_ms=c * 1e3, cpu_graphs_per_s=B / c, cpu_threads=nt)
        out[f"B={B}"] = row
        print(f"B={B}", row, file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
