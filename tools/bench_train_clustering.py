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
        # (whole-driver figure: upload of the graphs, structure launch, the chain of visits, assignment pass.  The cost
        # of ONE visit is measured with device timers below and, per issue form, by tools/bench_stage_a_visits.py; an
        # earlier version of this tool derived it from the difference of two runs with different epoch counts and
        # read ~30 % low)
        out[f"batch_graphs={bg}"] = {"seconds": t, "graph_visits_per_s": G * (epochs + 1) / t}
    # the chain of visits alone (hscn_scn_resident_train_epoch: one call, one launch per graph visit), device timers
    from graph_hscn.data import Batch
    from graph_hscn.step import ScnEpochRunner
    big = Batch.from_data_list(graphs)
    big.x = big.x.float()
    torch.manual_seed(0)
    scn = SCN(mc.mp_units, "elu", 9, K).to("cuda")
    if ScnEpochRunner.eligible(scn, big, "adam"):
        r = ScnEpochRunner(scn, big.to("cuda"), "adam", 0.01, 0.0)
        r.run(G)
        torch.cuda.synchronize()
        per = []
        for _ in range(3):
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record()
            r.run(10 * G)
            e_.record()
            torch.cuda.synchronize()
            per.append(s_.elapsed_time(e_) * 1e3 / (10 * G))
        r.check()
        out["visit_chain"] = {"us_per_graph_visit": sorted(per)[1], "graphs_per_s": 1e6 / sorted(per)[1],
                              "runs_us": per, "visits_per_run": 10 * G}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
