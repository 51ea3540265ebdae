#!/usr/bin/env python3
"""The whole device pipeline on one 128-graph batch, eager: cluster assignment (stage A forward + argmax)
-> heterogeneous batch (stage B, on the device) -> HSCN training step (stage C)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-hscn_amd")]
import torch

from graph_hscn import _hip
from graph_hscn.config.config import ACT_DICT
from graph_hscn.data import Batch
from graph_hscn.loader.hetero_data import hetero_batch_on_device
from graph_hscn.loader.synthetic import make_dataset
from graph_hscn.loss import criterion
from graph_hscn.model.hscn import HSCN, SCN


def main(B=128, K=16, iters=50):
    dev = torch.device("cuda:0")
    graphs = make_dataset("peptides_func", B, seed=0)
    big = Batch.from_data_list(graphs).to(dev)
    big.x = big.x.float()
    torch.manual_seed(0)
    scn = SCN([16], "elu", 9, K).to(dev)
    model = HSCN("GAT", "GCN", "GCN", ACT_DICT["relu"], 9, 16, 10, 3).to(dev)
    y = (torch.rand(B, 10, device=dev) < 0.2).float()

    def step():
        with torch.no_grad():
            S, _, _ = scn.forward_graphs(big)
            ids = torch.empty(S.size(0), dtype=torch.int64, device=dev)
            _hip.call("hscn_assign_argmax", _hip.ptr(S), _hip.ptr(ids), S.size(0), K, _hip.stream())
        hb = hetero_batch_on_device(big, ids, K)
        for p in model.parameters():
            p.grad = None
        pred = model(hb.x_dict, hb.edge_index_dict, hb)
        loss, _ = criterion("cross_entropy", pred, y)
        loss.backward()

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / iters
    print(json.dumps({"graphs": B, "ms_per_batch": t * 1e3, "graphs_per_s": B / t, "engine": model.last_engine,
                      "note": "eager (stage B reads its output sizes back to the host): launch-bound"}))


if __name__ == "__main__":
    main()
