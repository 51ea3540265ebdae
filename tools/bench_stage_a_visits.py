#!/usr/bin/env python3
"""Stage A in the reference's form (train/train_clustering.py:34-50: one optimizer step per graph, sequential): cost of
ONE graph visit, measured with device timers over 5 passes of 1024 Peptides-func-shaped graphs (after a warm pass), for
the issue forms the driver has had:
  pair + torch        forward launch, backward launch (+ fold), replay of torch's captured fused Adam
  one launch + torch  the one-launch step, replay of torch's captured fused Adam
  one launch + flat   the one-launch step, optim.FlatAdam as a launch of its own
  fused               the optimizer in the step's tail (one launch per visit), structure rebuilt every visit
  fused + cache       ... and the graph's CSRs / out-degrees / A_hat x loaded from HBM (per-graph Python objects)
  one call            hscn_scn_resident_train_epoch (what train_clustering does): the chain walked by one persistent
                      workgroup; "launch per visit": HSCN_PERSISTENT_EPOCH=0, the per-visit launches issued by the library
"""
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
from graph_hscn.optim import FlatAdam
from graph_hscn.replay import capture_optimizer_step
from graph_hscn.step import ScnEpochRunner, ScnStructurePool, ScnTrainStep, ScnWorkspace


def timed(fn, visits):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    s.record()
    for _ in range(5):
        fn()
    e.record()
    torch.cuda.synchronize()
    return {"us_per_visit_device": s.elapsed_time(e) * 1e3 / (5 * visits),
            "us_per_visit_wall": (time.perf_counter() - t0) * 1e6 / (5 * visits)}


def main(G=1024, K=16):
    dev = torch.device("cuda")
    graphs = make_dataset("peptides_func", G, seed=0)
    ds = [g.to(dev) for g in graphs]
    for d in ds:
        d.x = d.x.float()
    out = {"graphs": G, "passes_timed": 5}

    def fresh(one_launch, cache):
        torch.manual_seed(0)
        scn = SCN([16], "elu", 9, K).to(dev)
        ws = ScnWorkspace(dev, max(d.num_nodes for d in ds), max(d.edge_index.size(1) for d in ds), 1, 9, 16, K)
        pool = ScnStructurePool(dev, sum(d.num_nodes for d in ds), sum(d.edge_index.size(1) for d in ds), G) if cache else None
        steps = [ScnTrainStep(scn, d, workspace=ws, one_launch=one_launch, structure_pool=pool) for d in ds]
        steps[0].bind_grads()
        return scn, ws, steps

    for name, one_launch in (("pair + torch", False), ("one launch + torch", True)):
        scn, ws, steps = fresh(one_launch, False)
        opt = torch.optim.Adam(scn.parameters(), lr=0.01, capturable=True, fused=True)
        g = capture_optimizer_step(scn.parameters(), opt)

        def epoch():
            for st in steps:
                st.run()
                g.replay()
        out[name] = timed(epoch, G)
    scn, ws, steps = fresh(True, False)
    flat = FlatAdam.from_config("adam", steps[0].param_grads, ws.grads, 0.01, 0.0)

    def epoch():
        for st in steps:
            st.run()
            flat.step()
    out["one launch + flat"] = timed(epoch, G)
    for name, cache in (("fused", False), ("fused + cache", True)):
        scn, ws, steps = fresh(True, cache)
        flat = FlatAdam.from_config("adam", steps[0].param_grads, ws.grads, 0.01, 0.0)

        def epoch():
            for st in steps:
                st.run(opt=flat)
        out[name] = timed(epoch, G)
    big = Batch.from_data_list(graphs)
    big.x = big.x.float()
    torch.manual_seed(0)
    scn = SCN([16], "elu", 9, K).to(dev)
    r = ScnEpochRunner(scn, big.to(dev), "adam", 0.01, 0.0)
    out["one call"] = timed(lambda: r.run(G), G)
    os.environ["HSCN_PERSISTENT_EPOCH"] = "0"          # (read by the library at every call)
    out["one call, launch per visit"] = timed(lambda: r.run(G), G)
    del os.environ["HSCN_PERSISTENT_EPOCH"]
    # the same pass captured once as a hipGraph of G kernel nodes and replayed (what ScnEpochRunner.run does per epoch)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        r.run(G)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(cg):
        r.run(G)
    torch.cuda.synchronize()
    t_cap = time.perf_counter() - t0
    out["one call, captured pass replayed"] = timed(cg.replay, G)
    out["one call, captured pass replayed"]["capture_seconds"] = t_cap
    r.check()
    for k, v in out.items():
        if isinstance(v, dict):
            v["graphs_per_s"] = 1e6 / v["us_per_visit_device"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
