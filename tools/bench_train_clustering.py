#!/usr/bin/env python3
"""Wall clock of the stage-A driver itself (train/train_clustering.py: cluster_epochs passes with an optimizer step
per step + the assignment pass), reference trajectory (1 graph / step) and batched (128 graphs / step)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-hscn_amd")]
import torch

from graph_hscn.config.config import HSCNConfig, OptimConfig, TrainingConfig
from graph_hscn.loader.synthetic import make_dataset
from graph_hscn.model.hscn import SCN
from graph_hscn.train.train_clustering import train_clustering


def main(G=1024, epochs=5, K=16):
    graphs = make_dataset("peptides_func", G, seed=0)
    mc = HSCNConfig("relu", num_clusters=K, cluster_epochs=epochs)
    oc = OptimConfig("adam", lr=0.01)
    tc = TrainingConfig("hscn", "cross_entropy", "ap")
    out = {"graphs": G, "cluster_epochs": epochs}
    for bg in (1, 128):
        torch.manual_seed(0)
        scn = SCN(mc.mp_units, "elu", 9, K).to("cuda")
        train_clustering(None, graphs[:bg * 2], scn, HSCNConfig("relu", num_clusters=K, cluster_epochs=1), oc, tc, batch_graphs=bg)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ids = train_clustering(None, graphs, scn, mc, oc, tc, batch_graphs=bg)
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        assert len(ids) == G
        # steady state: the difference of two runs that differ in the number of epochs only (setup -- uploading the
        # graphs, building the per-step objects, capturing the optimizer step -- cancels)
        mc2 = HSCNConfig("relu", num_clusters=K, cluster_epochs=3 * epochs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        train_clustering(None, graphs, scn, mc2, oc, tc, batch_graphs=bg)
        torch.cuda.synchronize()
        t2 = time.perf_counter() - t0
        per_visit = (t2 - t) / (2 * epochs * G)
        # (batched: 8 steps per epoch -- the difference of two runs is within the noise of their setup; only the
        # whole-run figure is reported)
        steady = per_visit > 0 and bg == 1
        out[f"batch_graphs={bg}"] = {"seconds": t, "graph_visits_per_s": G * (epochs + 1) / t,
                                     "steady_state_us_per_graph_visit": per_visit * 1e6 if steady else None,
                                     "steady_state_graphs_per_s": 1.0 / per_visit if steady else None}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
