#!/usr/bin/env python3
"""A training-loop-shaped measurement: 8 DIFFERENT 128-graph batches (pre-packed on the device) cycle
through one captured step -- per iteration one flat copy into the static buffers + one hipGraph replay."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-hscn_amd")]
import torch

import bench
from graph_hscn.config.config import ACT_DICT
from graph_hscn.model.hscn import HSCN
from graph_hscn.replay import CapturedStep, StaticHeteroBatch


def main(B=128, nb=8, iters=400):
    dev = torch.device("cuda:0")
    batches = [bench.build_hetero_batch("peptides_func", B, 16, seed, dev)[0] for seed in range(nb)]
    torch.manual_seed(0)
    model = HSCN("GAT", "GCN", "GCN", ACT_DICT["relu"], 9, 16, 10, 3).to(dev)
    model.engine = "resident"
    static = StaticHeteroBatch(batches, dev)
    packed = [static.pack(hb) for hb in batches]
    static.load(packed[0])
    step = CapturedStep(model, static, "cross_entropy")
    for i in range(20):
        static.load(packed[i % nb])
        step.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        static.load(packed[i % nb])
        step.replay()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / iters
    out = {"graphs_per_batch": B, "different_batches": nb, "ms_per_step": t * 1e3, "graphs_per_s": B / t,
           "static_buffer_bytes": static.nbytes}
    # the same loop fed from pinned HOST memory: one H2D copy of the packed batch per step on the step's stream
    # (PCIe-inclusive rate: what a loader that keeps the dataset on the host delivers without overlap)
    hpacked = [p.cpu().pin_memory() for p in packed]
    for i in range(20):
        static.load(hpacked[i % nb])
        step.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        static.load(hpacked[i % nb])
        step.replay()
    torch.cuda.synchronize()
    th = (time.perf_counter() - t0) / iters
    out.update(host_fed_ms_per_step=th * 1e3, host_fed_graphs_per_s=B / th)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
