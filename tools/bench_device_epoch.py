#!/usr/bin/env python3
"""Epochs over an HBM-resident hetero dataset: the permutation and a batch counter live on the device, the gather
of the next slice (hscn_collate_gather) is captured in front of the training step: per step ONE graph replay.  Shuffled batches every epoch, no host
collate, no PCIe traffic.  Prints graphs/s over whole epochs and the dataset's footprint."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-hscn_amd")]
import numpy as np
import torch

from graph_hscn.config.config import ACT_DICT
from graph_hscn.loader.device_dataset import DeviceHeteroDataset
from graph_hscn.loader.hetero_data import hetero_from_clusters
from graph_hscn.loader.synthetic import make_dataset
from graph_hscn.model.hscn import HSCN
from graph_hscn.replay import CapturedStep


def main(G=4096, B=128, K=16, epochs=5):
    dev = torch.device("cuda:0")
    graphs = make_dataset("peptides_func", G, seed=0)
    rng = np.random.default_rng(0)
    t0 = time.perf_counter()
    hs = [hetero_from_clusters(g, rng.integers(0, K, g.num_nodes), K) for g in graphs]
    t_host_transform = time.perf_counter() - t0
    t0 = time.perf_counter()
    ds = DeviceHeteroDataset(hs, dev, B)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    torch.manual_seed(0)
    model = HSCN("GAT", "GCN", "GCN", ACT_DICT["relu"], 9, 16, 10, 3).to(dev)
    model.engine = "resident"
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, capturable=True, fused=True)
    gen = torch.Generator(device=dev).manual_seed(0)
    ds.new_epoch(gen)
    step = CapturedStep(model, ds.static, "cross_entropy", pre=ds.gather_next)     # gather + step in one graph
    step_opt = CapturedStep(model, ds.static, "cross_entropy", optimizer=opt, pre=ds.gather_next)   # + AdamW
    from graph_hscn.optim import FlatAdam
    step_flat = CapturedStep(model, ds.static, "cross_entropy", pre=ds.gather_next,      # + AdamW as ONE launch
                             optimizer=lambda st: FlatAdam.from_config("adamW", st.param_grads, st.grads, 1e-3, 0.01))
    steps = G // B

    def epoch(with_opt):
        ds.new_epoch(gen)
        st = {False: step, True: step_opt, "flat": step_flat}[with_opt]
        for i in range(steps):
            st.replay()

    out = {"graphs": G, "graphs_per_batch": B, "dataset_bytes": ds.nbytes, "static_buffer_bytes": ds.static.nbytes,
           "host_transform_s": t_host_transform, "dataset_build_s": t_build}
    for with_opt in (False, True, "flat"):
        epoch(with_opt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(epochs):
            epoch(with_opt)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / (epochs * steps)
        key = {False: "fwd_loss_bwd", True: "with_fused_adamw_in_graph", "flat": "with_flat_adamw_in_graph"}[with_opt]
        out[key] = {"ms_per_step": t * 1e3, "graphs_per_s": B / t}
    ds.check()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
