#!/usr/bin/env python3
"""Stage B throughput: cluster ids -> collated HeteroBatch (reference loader/hetero_data.py:42-87 + collate).
Host path (vectorised numpy per graph + collate) vs the device path (two launches + one 4-value host read)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-hscn_amd")]
import numpy as np
import torch

from graph_hscn.data import Batch, HeteroBatch
from graph_hscn.loader.hetero_data import hetero_batch_on_device, hetero_from_clusters
from graph_hscn.loader.synthetic import make_dataset


def main(B=128, K=16, iters=30):
    dev = torch.device("cuda:0")
    graphs = make_dataset("peptides_func", B, seed=0)
    rng = np.random.default_rng(0)
    ids = [rng.integers(0, K, g.num_nodes) for g in graphs]
    t0 = time.perf_counter()
    for _ in range(3):
        hb = HeteroBatch.from_data_list([hetero_from_clusters(g, i, K) for g, i in zip(graphs, ids)])
    t_host = (time.perf_counter() - t0) / 3
    big = Batch.from_data_list(graphs).to(dev)
    big.x = big.x.float()
    cl = torch.from_numpy(np.concatenate(ids)).to(dev)
    for _ in range(3):
        hetero_batch_on_device(big, cl, K)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        hetero_batch_on_device(big, cl, K)
    torch.cuda.synchronize()
    t_dev = (time.perf_counter() - t0) / iters
    print(json.dumps({"graphs": B, "host_ms": t_host * 1e3, "host_graphs_per_s": B / t_host,
                      "device_ms": t_dev * 1e3, "device_graphs_per_s": B / t_dev}))


if __name__ == "__main__":
    main()
