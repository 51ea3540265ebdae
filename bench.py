#!/usr/bin/env python3
"""Graph-HSCN hot-path benchmark on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N=1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = HSCN forward + criterion + backward over one Peptides-func-shaped
batch of 128 heterogeneous graphs per GPU (BASELINE.json configs[1]: K=16
clusters, hidden 16, 3 layers, 10 classes), inputs resident in HBM, including
the per-batch COO->CSR structure build.  Weak scaling: every rank owns its own
128-graph shard; gradients are all-reduced (RCCL) as one flat buffer per step.
Prints ONE JSON line on rank 0 (contract in the task statement) with the
`roofline` of the local->local SpMM kernel and a `cpu_baseline` (the CPU oracle
= restatement of the reference's PyG path, timed on this box's host cores).
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "graph-hscn_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np
import torch
import torch.distributed as dist

WORKLOADS = {
    # name: (shape, B per GPU, K clusters, classes, loss)
    "peptides_func": ("peptides_func", 128, 16, 10, "cross_entropy"),
    "peptides_struct": ("peptides_struct", 32, 32, 11, "l1"),
    "pascalvoc_sp": ("pascalvoc_sp", 128, 64, 21, "cross_entropy"),
    "pcqm_contact": ("pcqm_contact", 256, 16, 1, "l1"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# HSCN_BENCH_REHEARSAL=1: let N ranks SHARE fewer GPUs (rank r on device r % count), process group on gloo, gradients
# through the one-shot all-reduce (RCCL refuses two ranks on one device).  It exists to run the N-rank code path --
# self-launch, shard seeds, exchange, strong-scaling leg, A/B child -- on a one-GPU box; its line says so and its
# numbers are NOT a scaling measurement.
REHEARSAL = os.environ.get("HSCN_BENCH_REHEARSAL") == "1"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="peptides_func", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="graphs per GPU (default: the workload's)")
    ap.add_argument("--hidden", type=int, default=16)
    ap.add_argument("--layers", type=int, default=3)
    ap.add_argument("--mode", default="graph", choices=["graph", "eager", "eager-autograd"],
                    help="graph: the step (structure build + fwd + loss + bwd) is replayed as one hipGraph; eager: the same "
                         "direct C-ABI launches issued from Python every step (what the PMC passes profile); "
                         "eager-autograd: model(...) -> criterion -> loss.backward() through torch autograd")
    ap.add_argument("--engine", default="auto", choices=["auto", "resident", "layered"],
                    help="resident: one workgroup per graph, all layers in LDS; layered: one kernel per operator")
    ap.add_argument("--structure", default="per-step", choices=["per-step", "cached", "dataset-resident"],
                    help="per-step (the headline): every timed step builds its graphs' CSRs and degree norms from the COO "
                         "lists; dataset-resident: they were built once (hscn_resident_structure; graph structure is "
                         "epoch-invariant) and the step loads them -- a second, labelled line; cached: the layered "
                         "engine's per-batch cache")
    ap.add_argument("--steps-per-graph", type=int, default=0,
                    help="0 (default): the largest divisor of --steps that is <= 20.  "
                         "graph mode: capture this many consecutive steps (each with its gradient all-reduce) in one "
                         "hipGraph -- the host's replay overhead (~4 us per graph launch) is then paid once per group, as "
                         "in a training loop that captures several iterations per replay; the timed region is still "
                         "exactly --steps steps, and the line carries the 1-step-per-graph figure beside it")
    ap.add_argument("--cluster-ids", default="scn_untrained", choices=["scn_untrained", "uniform"],
                    help="scn_untrained: argmax of a seeded random-weight SCN (SURVEY.md 8d; collapses to ~3 clusters per "
                         "graph); uniform: ids drawn uniformly from 0..K-1 (U ~ K virtual nodes per graph, what a trained "
                         "assignment produces).  The headline line carries the other choice's time as `other_cluster_ids`")
    ap.add_argument("--no-other-ids", action="store_true", help="skip the second (other --cluster-ids) measurement")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f16"],
                    help="f16: node features and inter-layer activations stored as IEEE half in HBM, float accumulation "
                         "(BASELINE.json configs[4]: PCQM-Contact, 'fp16 feat + bf16 accum'); graph-resident engine only")
    ap.add_argument("--stage", default="c", choices=["c", "a"],
                    help="c: the headline (HSCN fwd + loss + bwd); a: the MinCUT coarsening step alone (gcn_norm + SCN fwd + "
                         "(mincut + ortho) bwd) as the headline line, with --route")
    ap.add_argument("--route", default="sparse", choices=["sparse", "dense"],
                    help="--stage a: sparse = fused graph-resident launches on the edge list; dense = to_dense_adj + "
                         "dense_mincut_pool on the matrix cores (BASELINE.json configs[3]: PascalVOC-SP, 64 clusters; "
                         "equally sized graphs, n = 479)")
    ap.add_argument("--allreduce", default=os.environ.get("HSCN_ALLREDUCE", "rccl"), choices=["rccl", "oneshot"],
                    help="gradient exchange of an N-rank run: rccl = one captured RCCL all-reduce of the flat buffer; oneshot = "
                         "hscn_allreduce_oneshot (one launch over hipIpc-mapped peer memory, one xGMI hop).  The N-rank "
                         "line of the default choice carries the other one's time as `allreduce_ab`, measured by a child job")
    ap.add_argument("--no-allreduce-ab", action="store_true", help="N > 1: skip the child job that times the other all-reduce")
    ap.add_argument("--no-strong", action="store_true",
                    help="N > 1: skip the strong-scaling reference (rank 0 alone on the whole global batch)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="after the reported timed region (`value` = that FIRST region), time the same --steps steps this many "
                         "more times and report the spread (`repeats`): a 20-step region is 0.6 ms, one scheduling hiccup moves it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-streaming-spmm", action="store_true")
    ap.add_argument("--no-stage-a", action="store_true")
    ap.add_argument("--no-stage-a-dense", action="store_true",
                    help="skip the secondary line `stage_a_dense` (BASELINE.json configs[3]: PascalVOC-SP-shaped graphs, "
                         "B = 128, K = 64, MinCUT coarsening step on the dense MFMA route, with its MFMA roofline)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    if args.steps_per_graph <= 0:
        args.steps_per_graph = max(d for d in range(1, 21) if args.steps % d == 0) if args.steps > 0 else 1
    return args


def build_hetero_batch(workload, B, K, seed, dev, cluster_ids="scn_untrained"):
    """Synthetic graphs -> cluster ids (from a seeded random-weight SCN on the HIP path, or
    uniform) -> generate_hetero_data transform -> one collated batch on the device."""
    from graph_hscn.data import Batch, HeteroBatch
    from graph_hscn.loader.hetero_data import hetero_from_clusters
    from graph_hscn.loader.synthetic import make_dataset
    from graph_hscn.model.hscn import SCN
    from graph_hscn.nn.pool import gcn_norm
    from graph_hscn import _hip

    graphs = make_dataset(workload, B, seed=seed)
    big = Batch.from_data_list(graphs)
    if cluster_ids == "uniform":
        ids = np.random.default_rng(4321 + seed).integers(0, K, big.num_nodes)
        ptr = big.ptr.numpy()
        hs = [hetero_from_clusters(g, ids[ptr[i]:ptr[i + 1]], K) for i, g in enumerate(graphs)]
        return HeteroBatch.from_data_list(hs), graphs, ids
    torch.manual_seed(1234 + seed)
    scn = SCN([16], "elu", graphs[0].num_features, K).to(dev)
    with torch.no_grad():
        ei, ew = gcn_norm(big.edge_index.to(dev), None, big.num_nodes, add_self_loops=True)
        S, _, _, _ = scn(big.x.to(dev).float(), ei, ew, node_ptr=big.ptr.to(dev).to(torch.int32))
        ids = torch.empty(big.num_nodes, dtype=torch.int64, device=dev)
        _hip.call("hscn_assign_argmax", _hip.ptr(S), _hip.ptr(ids), big.num_nodes, K, _hip.stream())
    ids = ids.cpu().numpy()
    ptr = big.ptr.numpy()
    hs = [hetero_from_clusters(g, ids[ptr[i]:ptr[i + 1]], K) for i, g in enumerate(graphs)]
    hb = HeteroBatch.from_data_list(hs)
    return hb, graphs, ids


class KernelTimer:
    """HIP-event timing of selected C-ABI launches on the stream they run on."""

    def __init__(self, names):
        self.names = set(names)
        self.events = []

    def wrap(self, name, fn, *args):
        if name.replace("_f16", "") not in self.names:      # (the half-storage twins count as their float namesakes)
            return fn(*args)
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        rc = fn(*args)
        e.record()
        self.events.append((name.replace("_f16", ""), args, s, e, name))
        return rc


def host_cores():
    """CPU cores this process may really use: min(os.cpu_count, affinity, cgroup quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(hb, args, C, loss_fn, seconds):
    """CPU oracle (PyG-order restatement of the reference path), fwd + loss + bwd
    on the same batch.  The intra-op thread count is picked by a short sweep
    (these are tiny ops: all cores is far from the fastest setting) and stated."""
    from oracle import models as OM
    avail = host_cores()
    torch.manual_seed(0)
    F = hb["local"].x.size(1)
    m = OM.HSCN("GAT", "GCN", "GCN", OM.ACT["relu"], F, args.hidden, C, args.layers)
    xd = {k: v.clone() for k, v in hb.x_dict.items()}
    ed = {k: v.clone() for k, v in hb.edge_index_dict.items()}
    bl = hb["local"].batch.clone()
    y = hb["local"].y.clone()
    B = hb.num_graphs

    def step():
        m.zero_grad(set_to_none=True)
        out = m(xd, ed, bl, B)
        loss, _ = OM.criterion(loss_fn, out, y)
        loss.backward()

    best_t, best_n = None, 1
    sweep = {}
    for nt in [c for c in (1, 2, 4, 8, 16, 32) if c <= avail]:
        torch.set_num_threads(nt)
        step()
        t0 = time.perf_counter()
        k = 0
        while k < 3 or (time.perf_counter() - t0 < 0.3 and k < 50):
            step()
            k += 1
        t = (time.perf_counter() - t0) / k
        sweep[nt] = round(1e3 * t, 3)
        if best_t is None or t < best_t:
            best_t, best_n = t, nt
    torch.set_num_threads(best_n)
    n = int(max(5, min(5000, seconds / max(best_t, 1e-4))))
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    dt = time.perf_counter() - t0
    return {"value": B * n / dt, "unit": "graphs/s", "cores": best_n, "kind": "port",
            "sample": f"{n} steps of the same {B}-graph batch (fwd+loss+bwd), torch {torch.__version__}, "
                      f"{best_n} intra-op threads (fastest of sweep ms/step {sweep}; {avail} cores usable, "
                      f"os.cpu_count {os.cpu_count()}); CPU oracle = PyG-order restatement of the reference path",
            "ms_per_step": 1e3 * dt / n}


class _StdoutToStderr:
    """RCCL prints a version banner on STDOUT when its first communicator is created; rank 0's stdout must carry
    one JSON line and nothing else (the driver parses it), so file descriptor 1 points at stderr while the process
    group comes up and the first collective runs."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *a):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


MAIN_SETTLE_S = 0.1          # one rank: untimed replays for this long before the --warmup steps
MAIN_SETTLE_STEPS = 2000     # ranks with a collective in the step: a fixed number instead (same count on every rank)


def capture(fn, warmup=3):
    """Warm ``fn`` up on a side stream, then capture it as one hipGraph."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warmup):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g


class TimedStep:
    """One training step of stage C in the form the run asks for.  ``run(n)`` issues exactly n steps."""

    def __init__(self, args, model, hb, loss_fn, reducer, B, world, spg_override=None):
        from graph_hscn import structure
        from graph_hscn.loss import criterion
        from graph_hscn.step import ResidentTrainStep
        self.args, self.reducer = args, reducer
        self.fused = None
        self.multi, self.spg = None, 1
        y = hb["local"].y
        x_dict, ei_dict = hb.x_dict, hb.edge_index_dict
        root_grad = torch.ones((), dtype=torch.float32, device=y.device)   # = loss.backward()'s implicit ones_like(loss)
        pre_structure = None
        if args.structure == "dataset-resident":      # built once, outside every timed region
            from graph_hscn.engine import build_structure
            pre_structure = build_structure(hb)

        def fwd_bwd_autograd():
            if args.structure == "per-step":
                structure.clear_cache()
            for p in model.parameters():
                p.grad = None
            pred = model(x_dict, ei_dict, hb)
            loss, _ = criterion(loss_fn, pred, y)
            loss.backward(root_grad)
            return loss

        def reduce():
            if reducer is not None:
                reducer.reduce(B, B * world)

        def step_eager():
            loss = fwd_bwd_autograd()
            reduce()
            return loss

        self.step_eager = step_eager
        self.in_graph_allreduce = False
        if args.mode == "eager-autograd" or (args.mode == "eager" and args.engine == "layered"):
            self.loss = step_eager().detach()
            self.run = lambda n: [step_eager() for _ in range(n)] and None
            return
        if args.mode == "eager":
            self.fused = ResidentTrainStep(model, hb, loss_fn, structure=pre_structure)
            self.fused.bind_grads()
            model.last_engine = "resident"
            self.loss = self.fused.loss

            def run_direct(n):
                for _ in range(n):
                    self.fused.run()
                    reduce()
            self.run = run_direct
            return
        # graph mode.  The gradient all-reduce is captured INTO the step's hipGraph (RCCL kernels are capturable): one
        # replay per step on every rank, no eager collective launch between replays.  HSCN_BENCH_GRAPH_ALLREDUCE=0
        # issues it eagerly after each replay instead (round 1's default: +12 us per step at one rank).
        self.in_graph_allreduce = reducer is not None and os.environ.get("HSCN_BENCH_GRAPH_ALLREDUCE", "1") != "0"
        one = None
        if args.engine != "layered":
            try:
                # the product's replayable step (graph_hscn.step / replay.CapturedStep): the same launches as the
                # autograd path, issued directly on preallocated buffers -- no autograd engine inside the capture
                self.fused = ResidentTrainStep(model, hb, loss_fn, structure=pre_structure)
                self.fused.bind_grads()
                model.last_engine = "resident"
                one = self.fused.run
                self.loss = self.fused.loss
            except RuntimeError:
                if args.engine == "resident":
                    raise
        if one is None:     # layered operators: the autograd path is what gets captured
            holder = {}

            def one():
                holder["loss"] = fwd_bwd_autograd()
            one()
            self.loss = None
        if self.in_graph_allreduce:
            body = lambda: (one(), reduce())
        else:
            body = one
        graph = capture(body)
        if self.loss is None:
            self.loss = holder["loss"].detach()
        # several steps per replay need the collective inside the graph (it sits between consecutive steps)
        self.spg = args.steps_per_graph if (args.steps_per_graph > 1 and (reducer is None or self.in_graph_allreduce)) else 1
        if spg_override is not None:
            self.spg = spg_override
        if self.spg > 1:
            self.multi = capture(lambda: [body() for _ in range(self.spg)], warmup=1)
            # the first launch of an instantiated graph uploads it (kernel arguments, node records): do that here, not
            # inside a timed region whose --warmup is shorter than one replay of this graph
            self.multi.replay()
            torch.cuda.synchronize()

        def run(n):
            if self.multi is not None:
                for _ in range(n // self.spg):
                    self.multi.replay()
                n %= self.spg
            for _ in range(n):
                graph.replay()
                if reducer is not None and not self.in_graph_allreduce:
                    reduce()
        self.run = run


def time_steps(ts, steps, warmup, barrier, settle_s=0.0, settle_steps=0):
    """`warmup` untimed steps, then exactly `steps` timed ones between two barriers.  settle_s (secondary legs only):
    keep replaying untimed for that long first -- those legs start after seconds of host-only work (building another
    batch), and 20 warm-up steps (< 1 ms) were seen to leave the timed steps at twice their duration now and then."""
    if settle_s > 0:
        t_end = time.perf_counter() + settle_s
        while time.perf_counter() < t_end:
            ts.run(max(1, warmup))
            torch.cuda.synchronize()
    if settle_steps > 0:      # the multi-rank form: every rank issues the same number of exchanges
        ts.run(settle_steps)
        torch.cuda.synchronize()
    ts.run(warmup)
    barrier()
    t0 = time.perf_counter()
    ts.run(steps)
    barrier()
    return time.perf_counter() - t0


def stage_a_main(args):
    print(json.dumps(stage_a_line(args)))


def stage_a_line(args):
    """--stage a: one MinCUT coarsening step (reference train/train_clustering.py:37-49 for a batch of graphs) as
    the reported line.  --route dense is BASELINE.json configs[3]: to_dense_adj + dense_mincut_pool with A S,
    S^T(A S), S^T S, S^T X on the matrix cores; the roofline is the MFMA one of the A S launch
    (2 K n^2 + 2 n K^2 flops per graph against the fp32-MFMA peak of 157.3 TFLOP/s)."""
    from graph_hscn import _hip
    from graph_hscn.data import Batch
    from graph_hscn.loader.synthetic import SHAPES, make_graph
    from graph_hscn.model.hscn import SCN
    from graph_hscn.step import ScnTrainStep
    import graph_hscn.nn.functional as Fh
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    _hip.lib()
    shape, B0, K, C, loss_fn = WORKLOADS[args.workload]
    B = args.batch or B0
    rng = np.random.default_rng(args.seed)
    fixed_n = None      # (round 2 gave every graph n = 479 on the dense route; graphs of any sizes share a batch now)
    graphs = [make_graph(rng, SHAPES[shape]) for _ in range(B)]
    F = graphs[0].x.size(1)
    torch.manual_seed(1)
    scn = SCN([16], "elu", F, K, mincut_route=args.route).to(dev)
    big = Batch.from_data_list(graphs).to(dev)
    big.x = big.x.float()
    N, E = int(big.num_nodes), int(big.edge_index.size(1))
    roofline = None
    spg_a, gm = 1, None
    if args.route == "sparse" and scn.resident_ok(big):
        st = ScnTrainStep(scn, big)
        g = capture(st.run)
        step = g.replay
        # like stage C: --steps-per-graph consecutive steps share one hipGraph (the replay overhead is paid once per group)
        spg_a = max(1, int(args.steps_per_graph))
        if spg_a > 1:
            gm = capture(lambda: [st.run() for _ in range(spg_a)], warmup=1)
            gm.replay()
            torch.cuda.synchronize()
        issue = ("hipGraph replay of the one-launch stage-A step + ordered fold (graph_hscn.step.ScnTrainStep)" if st.one_launch
                 else "hipGraph replay of the forward / backward launch pair + ordered fold (graph_hscn.step.ScnTrainStep)")
    else:
        one = torch.ones((), device=dev)

        def eager_step():
            for p in scn.parameters():
                p.grad = None
            S, mc, o, total = scn.forward_graphs(big, with_total=True)
            total.backward(one)
        step = eager_step
        issue = ("eager (layered operators + csrc/dense.hip through autograd)" if args.route == "dense" else
                 "eager (layered operators through autograd: the graphs do not fit the fused stage-A launch at this K)")
        if args.route == "dense" and args.mode == "graph":
            # every launch of the dense route has a host-known shape (gcn_norm_static, A + I straight from the raw
            # edges, ragged contractions): the whole step -- forward, losses, backward -- replays from one hipGraph
            try:
                gd = capture(eager_step)
                step = gd.replay
                issue = "hipGraph replay of the layered operators + csrc/dense.hip (autograd captured once)"
            except RuntimeError as e:      # pragma: no cover
                issue += f" [capture refused: {str(e)[:80]}]"
    def run_steps(n, one, many):
        if spg_a > 1:
            for _ in range(n // spg_a):
                many()
            n %= spg_a
        for _ in range(n):
            one()
    run_steps(args.warmup, step, gm.replay if spg_a > 1 else None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps, step, gm.replay if spg_a > 1 else None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    cached = None
    if args.route == "sparse" and scn.resident_ok(big):
        # SURVEY.md 8(d), "MinCUT sparse (per graph)": bytes 4(n+1) + 4(e+n) + 4nK, each array touched once, forward;
        # the backward moves the same bytes again (x 2).  The step is latency-bound like stage C's (one workgroup per
        # graph, the working set of a batch is a few MB): the fraction says how far from a bandwidth problem it is.
        ab = 2.0 * (4.0 * (N + B) + 4.0 * (E + N) + 4.0 * N * K)
        t_step = dt / args.steps
        roofline = {"bound": "hbm", "kernel": "k_scn_step + k_param_reduce (hscn_scn_resident_train_step)" if st.one_launch
                    else "k_scn_fwd + k_scn_bwd + k_param_reduce", "achieved": ab / t_step / 1e9, "peak": 8000.0,
                    "unit": "GB/s", "frac": ab / t_step / 8e12, "traffic": None, "algorithmic_bytes_per_launch": ab,
                    "avg_launch_us": t_step * 1e6,
                    "note": "time = one replayed step (both launches); latency-bound, see DESIGN.md section 4"}
    if args.route == "sparse" and scn.resident_ok(big) and st.one_launch:
        # second, labelled figure: the batch's CSRs / out-degrees / A_hat x kept in HBM after the first visit
        # (hscn_scn_structure: what the stage-A loop does from its second epoch on; same results bit for bit)
        from graph_hscn.step import ScnStructurePool
        st2 = ScnTrainStep(scn, big, structure_pool=ScnStructurePool(dev, N, E, B))
        st2.run()                                  # the visit that builds and stores the structure
        g2 = capture(st2.run)
        g2m = None
        if spg_a > 1:
            g2m = capture(lambda: [st2.run() for _ in range(spg_a)], warmup=1)
            g2m.replay()
        run_steps(args.warmup, g2.replay, g2m.replay if g2m else None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(args.steps, g2.replay, g2m.replay if g2m else None)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t0
        st2.check()
        assert torch.equal(st2.grads, st.grads) or st2.workspace is st.workspace
        cached = {"structure_build": "cached after the first visit (hscn_scn_structure)", "ms_per_step": 1e3 * dt2 / args.steps,
                  "graphs_per_s": B * args.steps / dt2}
    if args.route == "dense":
        # the dominant launch: A S (a [n,n] x [n,K] product per graph) inside hscn_mincut_dense_ragged_fwd -- time the C call
        sizes = np.array([int(g.num_nodes) for g in graphs], dtype=np.float64)
        timer = KernelTimer(["hscn_mincut_dense_fwd", "hscn_mincut_dense_bwd", "hscn_mincut_dense_ragged_fwd",
                             "hscn_mincut_dense_ragged_bwd", "hscn_mincut_dense_ragged_bwd_sym"])
        orig = _hip.call
        Fh.call = lambda name, *a: timer.wrap(name, lambda *b: orig(name, *b), *a)
        for _ in range(5):
            torch.cuda._sleep(400000)
            eager_step()
        torch.cuda.synchronize()
        Fh.call = orig

        def avg(name):
            hit = [(ev[4], ev[1]) for ev in timer.events if ev[0] == name][-1]
            for _ in range(3):
                orig(hit[0], *hit[1])
            torch.cuda._sleep(400000)
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record()
            for _ in range(20):
                orig(hit[0], *hit[1])
            e_.record()
            torch.cuda.synchronize()
            return s_.elapsed_time(e_) * 1e-3 / 20
        rag = any(ev[0].startswith("hscn_mincut_dense_ragged") for ev in timer.events)
        t_f = avg("hscn_mincut_dense_ragged_fwd" if rag else "hscn_mincut_dense_fwd")
        t_b = avg(("hscn_mincut_dense_ragged_bwd_sym" if any(ev[0] == "hscn_mincut_dense_ragged_bwd_sym" for ev in timer.events)
                   else "hscn_mincut_dense_ragged_bwd") if rag else "hscn_mincut_dense_bwd")
        # the dominant kernel by itself: the A S launch (hscn_dense_adj_s = what the forward call issues for it), on the
        # forward call's own operands, 20 launches between one HIP-event pair on the stream they are launched on
        t_as = None
        if rag:
            fa = [ev[1] for ev in timer.events if ev[0] == "hscn_mincut_dense_ragged_fwd"][-1]
            # (x, adj, adj_elem_bytes, logits, nptr, N, B, nmax, K, F, S, AS, deg, ...)
            as_args = (fa[1], fa[2], fa[10], fa[4], fa[6], fa[7], fa[8], 0, fa[11], fa[12], fa[-1])
            for _ in range(3):
                orig("hscn_dense_adj_s", *as_args)
            torch.cuda._sleep(400000)
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record()
            for _ in range(20):
                orig("hscn_dense_adj_s", *as_args)
            e_.record()
            torch.cuda.synchronize()
            t_as = s_.elapsed_time(e_) * 1e-3 / 20
        # S^T A S = 2 K n^2 + 2 n K^2 flops per graph, summed over the batch's actual sizes
        fl_sas = float(np.sum(2.0 * K * sizes * sizes + 2.0 * sizes * K * K))
        fl_f = float(np.sum(2.0 * sizes * sizes * K + 2.0 * K * sizes * K + 2.0 * K * sizes * K + 2.0 * K * sizes * 16))   # + S^T S + S^T X
        PEAK = 157.3
        fl_as = float(np.sum(2.0 * K * sizes * sizes))          # the A S launch: 2 K n^2 per graph
        if t_as is not None:
            kern, ach, t_dom = ("k_adj_s_direct (hscn_dense_adj_s: A S of a ragged batch on v_mfma_f32_32x32x2_f32, the "
                                "adjacency read as byte counts straight into the operand registers)"), fl_as / t_as / 1e12, t_as
        else:
            kern, ach, t_dom = ("hscn_mincut_dense_fwd (softmax, A S, merged S^T(AS) | S^T S | S^T X, statistics)",
                                fl_sas / t_f / 1e12, t_f)
        roofline = {"bound": "mfma", "kernel": kern, "achieved": ach, "peak": PEAK, "unit": "TFLOP/s",
                    "frac": ach / PEAK, "traffic": None, "avg_launch_us": t_dom * 1e6,
                    "flops_per_launch": fl_as if t_as is not None else fl_sas,
                    "flops_per_launch_SAS": fl_sas, "flops_per_launch_all_contractions": fl_f,
                    "fwd_call_us": t_f * 1e6, "frac_SAS_over_fwd_call": fl_sas / t_f / 1e12 / PEAK,
                    "frac_all_contractions": fl_f / t_f / 1e12 / PEAK,
                    "bwd_launch_us": t_b * 1e6, "adjacency_bytes": float(np.sum(sizes * sizes) * 4),
                    "adjacency_GBs_fwd": float(np.sum(sizes * sizes) * 4) / t_f / 1e9,
                    "nodes_per_graph_min_max": [int(sizes.min()), int(sizes.max())],
                    "note": "frac = the A S launch alone (2 K n^2 flops per graph over its own duration); the forward C call "
                            "spans several launches (softmax, A S, the merged cluster-space contractions, statistics): "
                            "fwd_call_us / frac_SAS_over_fwd_call; MFMA-busy from the PMC pass: profiles/r03_dense_mfma_busy.txt"}
    out = {"metric": f"graphs/sec (fwd+bwd) on {args.workload} stage A (MinCUT coarsening)", "value": B * args.steps / dt,
           "unit": "graphs/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"Graph-HSCN stage A (gcn_norm + SCN fwd + (mincut+ortho) bwd) on {args.workload}-shaped graphs",
                      "route": args.route, "graphs_per_gpu": B, "num_clusters": K, "nodes_per_gpu": N,
                      "edges_per_gpu": E, "nodes_per_graph": fixed_n, "step_issue": issue, "steps_per_graph": spg_a,
                      "parallelism": "dp1"},
           "roofline": roofline, "cpu_baseline": None}
    if cached is not None:
        out["structure_cached"] = cached
    return out


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv, timeout=None):
    """Start ``bench.py`` as n ranks (one per GPU) with torch.distributed.run as a CHILD job and return
    (return code, rank 0's JSON line or None, tail of its other output).  Called before this process has touched
    the GPU, or from a process that keeps running -- never an exec (a process that has initialised the GPU must not
    be replaced)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE",
              "TORCHELASTIC_RUN_ID", "HSCN_BENCH_FORCE_DIST"):
        env.pop(k, None)
    try:
        pr = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, timeout=timeout)
    except subprocess.TimeoutExpired as e:
        return 124, None, f"timed out after {timeout} s: {(e.stdout or b'')[-400:].decode(errors='replace')}"
    line, rest = None, []
    for ln in pr.stdout.decode(errors="replace").splitlines():
        try:
            doc = json.loads(ln)
            if isinstance(doc, dict) and "metric" in doc:
                line = doc
                continue
        except ValueError:
            pass
        rest.append(ln)
    return pr.returncode, line, "\n".join(rest[-5:])


def self_launch(args):
    """``python bench.py --gpus N`` with no launcher around it: become the launcher.  Nothing here initialises the
    GPU (``device_count`` does not), the ranks are fresh child processes, and the one line relayed is rank 0's."""
    have = torch.cuda.device_count()
    if have < args.gpus and not (REHEARSAL and have >= 1):
        raise SystemExit(f"bench.py --gpus {args.gpus}: this node shows {have} GPU(s); refusing to report a {args.gpus}-GPU "
                         f"line from fewer devices")
    rc, line, rest = launch_ranks(args.gpus, sys.argv[1:])
    if rest:
        print(rest, file=sys.stderr)
    if rc != 0 or line is None:
        raise SystemExit(f"bench.py --gpus {args.gpus}: the {args.gpus}-rank job failed (exit code {rc})")
    if line.get("n_gpus") != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: the job reported n_gpus = {line.get('n_gpus')}")
    print(json.dumps(line))


def main():
    args = parse()
    if args.stage == "a":
        return stage_a_main(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        # a line whose n_gpus differs from what was asked for is worse than no line
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} rank(s)")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    shared_gpus = REHEARSAL and world > torch.cuda.device_count()
    if world > torch.cuda.device_count() and not shared_gpus:
        raise SystemExit(f"bench.py: {world} ranks but {torch.cuda.device_count()} GPU(s) visible")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    force_dist = os.environ.get("HSCN_BENCH_FORCE_DIST") == "1"   # exercise the collective path with a single rank
    cdev = dev                                                     # where the process group's own tensors live
    if world > 1 or force_dist:
        if force_dist and world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
        if shared_gpus:
            if args.allreduce != "oneshot":
                raise SystemExit("HSCN_BENCH_REHEARSAL with ranks sharing a GPU needs --allreduce oneshot (RCCL refuses)")
            dist.init_process_group("gloo", rank=rank, world_size=world)
            cdev = torch.device("cpu")
        else:
            with _StdoutToStderr():
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
                dist.all_reduce(torch.zeros(1, device=dev))      # creates the communicator (and prints the banner) now
                torch.cuda.synchronize()
    from graph_hscn import _hip
    from graph_hscn.config.config import ACT_DICT
    from graph_hscn.distributed import FlatGradReducer
    from graph_hscn.model.hscn import HSCN

    _hip.lib()
    shape, B0, K, C, loss_fn = WORKLOADS[args.workload]
    B = args.batch or B0
    hb_host, graphs, _ = build_hetero_batch(shape, B, K, args.seed * 1000 + rank, dev, args.cluster_ids)
    hb = hb_host.to(dev)
    if args.dtype == "f16":
        hb = hb.with_feature_dtype(torch.float16)
    F = hb["local"].x.size(1)
    torch.manual_seed(0)  # identical replicas
    model = HSCN("GAT", "GCN", "GCN", ACT_DICT["relu"], F, args.hidden, C, args.layers).to(dev)
    model.engine = args.engine
    # every rank owns B graphs: equal weights, RCCL averages with no scaling launch
    reducer = (FlatGradReducer(model, single_rank_collective=force_dist, equal_weights=True, algorithm=args.allreduce)
               if (world > 1 or force_dist) else None)

    def barrier():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    ts = TimedStep(args, model, hb, loss_fn, reducer, B, world)
    # settle: the step has just been captured after seconds of host-only work (batch building, instantiation); 0.1 s of
    # untimed replays first, then the --warmup steps, then exactly --steps timed ones (reported as config.settle_s)
    dt = time_steps(ts, args.steps, args.warmup, barrier, settle_s=MAIN_SETTLE_S if reducer is None else 0.0,
                    settle_steps=0 if reducer is None else MAIN_SETTLE_STEPS)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms = 1e3 * dt / args.steps
    value = B * world * args.steps / dt
    spg = ts.spg
    if ts.fused is not None:
        ts.fused.check()
    if reducer is not None:
        reducer.check()          # a timed-out one-shot exchange would have left gradients unreduced

    # ---- spread: the same timed region again (exactly --steps steps each, max over ranks); `value` stays the first
    repeats = None
    if args.repeats > 0:
        reps = []
        for _ in range(args.repeats):
            d = time_steps(ts, args.steps, 0, barrier)
            if world > 1:
                t = torch.tensor([d], dtype=torch.float64, device=cdev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                d = float(t.item())
            reps.append(1e3 * d / args.steps)
        allr = [ms] + reps
        repeats = {"k": args.repeats, "ms_per_step": [round(r, 6) for r in reps],
                   "median_ms_per_step_incl_first": statistics.median(allr), "min_ms_per_step": min(allr),
                   "max_ms_per_step": max(allr),
                   "median_graphs_per_s": B * world / (statistics.median(allr) * 1e-3),
                   "note": "`value` / `ms_per_step` are the FIRST timed region; these are further regions of the same length"}

    # ---- strong scaling (N > 1): the weak line above runs B graphs per GPU, i.e. a global batch of B * N; the same
    # global batch on ONE GPU is measured here by rank 0 alone (the other ranks wait at the barrier), so the line can
    # be read either way: weak (value vs the 1-GPU line's value) or strong (one_gpu_ms_per_step / ms_per_step)
    strong = None
    if world > 1 and not args.no_strong:
        if rank == 0:
            hbS = build_hetero_batch(shape, B * world, K, args.seed * 1000, dev, args.cluster_ids)[0].to(dev)
            if args.dtype == "f16":
                hbS = hbS.with_feature_dtype(torch.float16)
            tsS = TimedStep(args, model, hbS, loss_fn, None, B * world, 1)
            local_barrier = torch.cuda.synchronize
            dS = time_steps(tsS, args.steps, args.warmup, local_barrier, settle_s=0.05)
            dS = min(dS, time_steps(tsS, args.steps, 0, local_barrier))
            t1 = 1e3 * dS / args.steps
            strong = {"global_batch": B * world, "one_gpu_ms_per_step": t1, "one_gpu_graphs_per_s": B * world / (t1 * 1e-3),
                      "n_gpu_ms_per_step": ms, "n_gpus": world, "speedup_vs_one_gpu_same_global_batch": t1 / ms,
                      "note": "strong reading of the same run: global batch fixed at graphs_per_gpu * n_gpus; the one-GPU "
                              "time is rank 0 alone on all of it (its best form for that batch size), measured in this job"}
            del tsS, hbS
        barrier()

    # ---- the same measurement with ONE step per hipGraph replay (what round 1 reported): the difference is the host's
    # replay overhead per graph launch, not kernel time
    single = None
    if rank == 0 and world == 1 and not force_dist and args.mode == "graph" and spg > 1:
        ts1 = TimedStep(args, model, hb, loss_fn, None, B, 1, spg_override=1)
        dt1 = time_steps(ts1, args.steps, args.warmup, barrier, settle_s=0.05)
        single = {"steps_per_graph": 1, "ms_per_step": 1e3 * dt1 / args.steps, "graphs_per_s": B * args.steps / dt1}
        del ts1

    # ---- the same step on the OTHER choice of cluster ids (one GPU only): the untrained SCN's argmax collapses to
    # ~3 clusters per graph, a trained assignment uses ~K -- the virtual branch's share of the step differs
    other_ids = None
    if rank == 0 and world == 1 and not force_dist and not args.no_other_ids:
        oc = "uniform" if args.cluster_ids == "scn_untrained" else "scn_untrained"
        hb2 = build_hetero_batch(shape, B, K, args.seed * 1000 + rank, dev, oc)[0].to(dev)
        if args.dtype == "f16":
            hb2 = hb2.with_feature_dtype(torch.float16)
        ts2 = TimedStep(args, model, hb2, loss_fn, None, B, 1)
        dt2 = time_steps(ts2, args.steps, args.warmup, barrier, settle_s=0.05)
        other_ids = {"cluster_ids": oc, "virtual_nodes_per_gpu": int(hb2["virtual"].num_nodes),
                     "vv_edges_per_gpu": int(hb2[("virtual", "to", "virtual")].edge_index.size(1)),
                     "ms_per_step": 1e3 * dt2 / args.steps, "graphs_per_s": B * args.steps / dt2}
        del ts2, hb2

    # ---- roofline of the dominant kernel: eager steps, HIP events around its C-ABI launch ----
    roofline = None
    if rank == 0:
        import graph_hscn.engine as eng
        import graph_hscn.nn.functional as Fh
        import graph_hscn.step as stepmod
        N = int(hb["local"].num_nodes)
        V = int(hb["virtual"].num_nodes)
        E = int(hb[("local", "to", "local")].edge_index.size(1))
        Evv = int(hb[("virtual", "to", "virtual")].edge_index.size(1))
        H, L = args.hidden, args.layers
        # SURVEY.md 8(d) algorithmic bytes, summed over the batch (feature values of `sz` bytes, int32 CSR indices):
        sz = 2 if args.dtype == "f16" else 4
        ll_b = 4 * (N + B) + 4 * E + 4 * N + 2 * sz * N * H     # rowptr + col + dinv + read h + write out
        lv_b = sz * N * H + 8 * N + 4 * V + sz * V * H
        vv_b = 2 * sz * V * H + 4 * Evv
        lin_b = 3 * sz * N * H + 3 * sz * V * H
        used_resident = model.last_engine == "resident"
        names = (["hscn_resident_fwd", "hscn_resident_bwd", "hscn_resident_fwd_with_virtual",
                  "hscn_resident_bwd_with_virtual", "hscn_resident_train_step"] if used_resident
                 else ["hscn_spmm_csr_gcn"])
        timer = KernelTimer(names)
        orig_call = _hip.call

        def timed_call(name, *a):
            return timer.wrap(name, lambda *b: orig_call(name, *b), *a)

        Fh.call = timed_call
        eng.call = timed_call
        stepmod.call = timed_call
        nprof = max(5, min(50, args.steps))
        prof_step = ts.fused.run if ts.fused is not None else ts.step_eager
        for _ in range(nprof):
            # park the stream for ~0.2 ms so the host runs ahead: the launches of this step then sit in the
            # queue back to back and the event pair brackets the kernel, not the host's launch latency
            torch.cuda._sleep(400000)
            prof_step()
        torch.cuda.synchronize()
        Fh.call = orig_call
        eng.call = orig_call
        stepmod.call = orig_call

        def avg_s(pred):
            """Average launch duration: the last recorded launch that matches is re-issued REP times back to
            back between ONE HIP-event pair on its stream (same arguments, same outputs), so the events'
            own cost and the host's launch latency are amortised instead of added to every launch."""
            hits = [(real, a) for (nm, a, s_, e_, real) in timer.events if pred(nm, a)]
            if not hits:
                return float("nan"), 0
            nm, a = hits[-1]
            REP = 20
            for _ in range(3):
                orig_call(nm, *a)
            torch.cuda._sleep(400000)
            s_ = torch.cuda.Event(enable_timing=True)
            e_ = torch.cuda.Event(enable_timing=True)
            s_.record()
            for _ in range(REP):
                orig_call(nm, *a)
            e_.record()
            torch.cuda.synchronize()
            return s_.elapsed_time(e_) * 1e-3 / REP, REP

        if used_resident:
            split = any(ev[0] == "hscn_resident_fwd_with_virtual" for ev in timer.events)
            one_launch = any(ev[0] == "hscn_resident_train_step" for ev in timer.events)
            virt_layer = lv_b + vv_b + 4 * N * H + 12 * V * H      # one layer of the virtual branch
            alg_f = L * (ll_b + lv_b + vv_b + lin_b)
            alg_b = L * (ll_b + lin_b)     # the backward only walks the local->local relation (+ its transforms)
            if one_launch:
                t, n_t = avg_s(lambda nm, a: nm == "hscn_resident_train_step")
                spmm_passes, unfused = 2 * L, alg_f + alg_b
                kname = ("k_hscn_step + k_param_reduce (hscn_resident_train_step: forward, loss tail and backward of a "
                         "graph in one workgroup | virtual branch, 2 workgroups/graph)")
                extra = {"step_launch_us": t * 1e6}
            else:
                t_f, n_f = avg_s(lambda nm, a: nm in ("hscn_resident_fwd", "hscn_resident_fwd_with_virtual"))
                t_b, n_b = avg_s(lambda nm, a: nm in ("hscn_resident_bwd", "hscn_resident_bwd_with_virtual"))
                if split:                      # layers 1.. of the virtual branch ride on the backward launch
                    alg_f -= (L - 1) * virt_layer
                    alg_b += (L - 1) * virt_layer
                dom_fwd = t_f >= t_b
                t, unfused, n_t = (t_f, alg_f, n_f) if dom_fwd else (t_b, alg_b, n_b)
                spmm_passes = L
                k_f = ("k_hscn_fwd_pair (hscn_resident_fwd_with_virtual: local chain + head | virtual CSRs + layer 0, "
                       "2 workgroups/graph)") if split else "k_hscn_fwd (hscn_resident_fwd: all layers, 1 workgroup/graph)"
                k_b = ("k_hscn_bwd_virtual + k_param_reduce (hscn_resident_bwd_with_virtual: backward | virtual layers 1..)"
                       if split else "k_hscn_bwd + k_param_reduce (hscn_resident_bwd)")
                kname = k_f if dom_fwd else k_b
                extra = {"fwd_us": t_f * 1e6, "bwd_us": t_b * 1e6, "fwd_unfused_model_bytes": alg_f,
                         "bwd_unfused_model_bytes": alg_b}
            # `achieved` / `frac`: SURVEY.md 8(d)'s numerator -- the local->local SpMM bytes of the passes this launch
            # performs (one per layer and direction) -- over the launch's duration.  The launch keeps every
            # intermediate in LDS, so its real HBM traffic (`traffic`, PMC) is of the same order, and both are a few
            # per cent of peak: the step is bound by the dependent chain of its largest graph, not by bandwidth.
            alg = spmm_passes * ll_b
            roofline = {"bound": "hbm", "kernel": kname,
                        "achieved": alg / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": alg / t / 1e9 / HBM_PEAK_GBS, "traffic": None, "frac_traffic": None,
                        "algorithmic_bytes_per_launch": alg, "ll_spmm_passes_per_launch": spmm_passes,
                        "ll_spmm_bytes_per_pass": ll_b, "avg_launch_us": t * 1e6, "launches_timed": n_t,
                        "frac_unfused_model": unfused / t / 1e9 / HBM_PEAK_GBS, "unfused_model_bytes": unfused,
                        "note": "latency-bound at this batch (the whole working set is a few MB, SURVEY.md 8d); "
                                "frac_unfused_model prices every operator's intermediates as if they went to HBM "
                                "(they stay in LDS) and is NOT a bandwidth figure; the HBM-resident SpMM is "
                                "streaming_spmm_scaled"}
            roofline.update(extra)
        else:
            # args[7] = num_rows, args[8] = width of hscn_spmm_csr_gcn
            t, n_l = avg_s(lambda nm, a: a[7] == N and a[8] == H)
            roofline = {"bound": "hbm", "kernel": "k_spmm<4,0> (hscn_spmm_csr_gcn, local->local, one pass)",
                        "achieved": ll_b / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ll_b / t / 1e9 / HBM_PEAK_GBS, "traffic": None, "frac_traffic": None,
                        "algorithmic_bytes_per_launch": ll_b, "avg_launch_us": t * 1e6, "launches_timed": n_l}

    # ---- the streaming (layered-engine) ll SpMM at a bandwidth-resident shape:
    # the same generator tiled to 4096 graphs, hidden 128 -- the shape where "HBM roofline" means something;
    # forward pass and backward pass (the same kernel on the source-keyed CSR), as 8(d) defines the figure
    streaming = None
    if rank == 0 and not args.no_streaming_spmm:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_spmm
        streaming = bench_spmm.measure(32, 128, 20, dev=dev)
        streaming["peak_GBs"] = HBM_PEAK_GBS

    # HBM traffic from PMC passes: separate rocprofv3 runs of this script (the counters cannot be read from inside
    # the process).  The newest profiles/rNN_pmc_traffic.json is used; it records the fingerprint of the kernel
    # sources it was collected from, and the line says whether that is still the code that is running
    if rank == 0:
        import glob
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from fingerprint import csrc_fingerprint
        now = csrc_fingerprint()
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
            fname = os.path.basename(path)
            try:
                doc = json.load(open(path))
                pmc = doc["kernels"]
            except (OSError, KeyError, ValueError):
                continue
            default_shape = (args.workload == "peptides_func" and B == 128 and args.hidden == 16 and args.layers == 3
                             and args.cluster_ids == "scn_untrained")
            stale = doc.get("_csrc_sha") != now
            if roofline and default_shape and roofline["traffic"] is None:
                key = roofline["kernel"].split(" ")[0]      # exact kernel name
                if key and key in pmc:
                    roofline["traffic"] = pmc[key]["traffic_bytes"]
                    roofline["frac_traffic"] = pmc[key]["traffic_bytes"] / (roofline["avg_launch_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS
                    roofline["traffic_commit"] = doc.get("_commit", "unrecorded")
                    roofline["traffic_csrc_sha"] = doc.get("_csrc_sha", "unrecorded")
                    roofline["csrc_sha_now"] = now
                    roofline["stale"] = bool(stale)
                    roofline["traffic_source"] = (f"profiles/{fname}: a SEPARATE rocprofv3 --pmc run of this command "
                                                  f"(FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 fetch correction), "
                                                  f"collected at commit {doc.get('_commit', 'unrecorded')}"
                                                  + ("; STALE: the kernel sources have changed since" if stale else
                                                     "; the kernel sources are unchanged since"))
                    roofline["rocprof_kernel_stats"] = (f"profiles/{fname.split('_')[0]}_step_kernel_stats.csv (rocprofv3 "
                                                        "--kernel-trace --stats of the default command): the C call timed "
                                                        "here = k_hscn_step + k_param_reduce")
            if streaming and "traffic" not in streaming and "k_spmm_scaled_H128" in pmc:
                streaming["traffic"] = pmc["k_spmm_scaled_H128"]["traffic_bytes"]
                streaming["traffic_stale"] = bool(stale)
            if roofline is None or roofline.get("traffic") is not None:
                break

    # ---- stage A (MinCUT coarsening: gcn_norm + SCN fwd + (mc+o) bwd) on the same graphs, fused engine,
    # one hipGraph replay per 128-graph step; reported beside the stage C headline
    stage_a = None
    if rank == 0 and not args.no_stage_a:
        from graph_hscn.data import Batch
        from graph_hscn.model.hscn import SCN
        from graph_hscn.step import ScnTrainStep
        torch.manual_seed(1)
        scn = SCN([16], "elu", F, K).to(dev)
        bigd = Batch.from_data_list(graphs).to(dev)
        bigd.x = bigd.x.half() if args.dtype == "f16" else bigd.x.float()
        if scn.resident_ok(bigd):
            a_step = ScnTrainStep(scn, bigd)      # (mc + o).backward() as ONE launch + the ordered fold, no autograd in the capture
            # the same grouping as the headline: steps_per_graph consecutive steps per hipGraph
            spg_a = max(1, int(spg))
            ga = capture(lambda: [a_step.run() for _ in range(spg_a)], warmup=1)
            n_rep = max(1, args.steps // spg_a)
            for _ in range(max(2, 20 // spg_a)):
                ga.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n_rep):
                ga.replay()
            torch.cuda.synchronize()
            ta = (time.perf_counter() - t0) / (n_rep * spg_a)
            a_step.check()
            stage_a = {"what": "gcn_norm + SCN fwd + (mincut+ortho) bwd, batched, graph-resident "
                               + ("one-launch step" if a_step.one_launch else "launch pair"),
                       "steps_per_graph": spg_a, "ms_per_step": ta * 1e3, "graphs_per_s": B / ta,
                       "combined_A_plus_C_graphs_per_s": B / (ta + dt / args.steps)}

    # ---- secondary line: BASELINE.json configs[3] -- the MinCUT coarsening step on PascalVOC-SP-shaped graphs (real
    # size spread, n in [395, 500]), 64 clusters, dense S^T A S route on the matrix cores, replayed from a hipGraph
    stage_a_dense = None
    if rank == 0 and world == 1 and not args.no_stage_a_dense and args.dtype == "f32":
        import copy
        a2 = copy.copy(args)
        a2.workload, a2.route, a2.stage, a2.batch, a2.mode = "pascalvoc_sp", "dense", "a", None, "graph"
        a2.steps, a2.warmup = max(20, min(args.steps, 100)), 10
        try:
            ln = stage_a_line(a2)
            stage_a_dense = {"metric": ln["metric"], "value": ln["value"], "unit": ln["unit"], "ms_per_step": ln["ms_per_step"],
                             "steps": a2.steps, "dtype": "f32", "config": ln["config"], "roofline": ln["roofline"]}
        except RuntimeError as e:      # pragma: no cover
            stage_a_dense = {"error": str(e)[:300]}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(hb_host, args, C, loss_fn, args.cpu_seconds)

    if rank == 0:
        out = {
            "metric": "graphs/sec (fwd+bwd) on Peptides-func batch=128" if args.workload == "peptides_func"
            else f"graphs/sec (fwd+bwd) on {args.workload}",
            "value": value, "unit": "graphs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.dtype == "f32" else "f16 storage (features + activations), f32 accumulate",
            "data": "synthetic",
            "config": {"workload": f"Graph-HSCN stage C (HSCN fwd+loss+bwd) on {args.workload}-shaped graphs",
                       "graphs_per_gpu": B, "global_batch": B * world, "num_clusters": K, "hidden": args.hidden,
                       "layers": args.layers, "classes": C, "nodes_per_gpu": int(hb["local"].num_nodes),
                       "ll_edges_per_gpu": int(hb[("local", "to", "local")].edge_index.size(1)),
                       "cluster_ids": args.cluster_ids,
                       "virtual_nodes_per_gpu": int(hb["virtual"].num_nodes),
                       "vv_edges_per_gpu": int(hb[("virtual", "to", "virtual")].edge_index.size(1)),
                       "mode": args.mode, "engine": model.last_engine, "structure_build": args.structure,
                       "steps_per_graph": spg,
                       "settle": (f"{MAIN_SETTLE_S} s of untimed replays" if reducer is None else f"{MAIN_SETTLE_STEPS} untimed steps")
                                 + " before the --warmup steps",
                       "step_issue": ("direct C-ABI launches (graph_hscn.step.ResidentTrainStep)" if ts.fused is not None
                                      else "autograd"),
                       "allreduce": (None if reducer is None else
                                     ("captured in the step's hipGraph" if ts.in_graph_allreduce else "eager, after each replay")),
                       "parallelism": f"dp{world}"},
            "roofline": roofline, "cpu_baseline": cpu, "one_step_per_graph": single, "other_cluster_ids": other_ids,
            "streaming_spmm_scaled": streaming, "stage_a": stage_a, "stage_a_dense": stage_a_dense, "repeats": repeats,
        }
        out["config"]["allreduce_algorithm"] = None if reducer is None else args.allreduce
        if shared_gpus:
            out["rehearsal"] = (f"{world} ranks SHARE {torch.cuda.device_count()} GPU(s) (HSCN_BENCH_REHEARSAL=1): this line "
                                f"exercises the N-rank code path and is NOT a scaling measurement")
        if strong is not None:
            out["strong_scaling"] = strong
        if cpu:
            out["vs_cpu_baseline"] = value / cpu["value"]
    if world > 1 or force_dist:
        dist.destroy_process_group()
    if rank == 0:
        # ---- A/B of the exchange (N > 1): the OTHER all-reduce algorithm in a child job of its own, started after this
        # job's ranks have left the GPUs -- a failure there (it has never met real xGMI before this run) costs its
        # own entry, not this line
        if world > 1 and not args.no_allreduce_ab and "HSCN_BENCH_AB_CHILD" not in os.environ:
            other = "oneshot" if (args.allreduce == "rccl" or shared_gpus) else "rccl"
            os.environ["HSCN_BENCH_AB_CHILD"] = "1"
            argv = ["--gpus", str(world), "--steps", str(args.steps), "--warmup", str(args.warmup), "--allreduce", other,
                    "--workload", args.workload, "--hidden", str(args.hidden), "--layers", str(args.layers),
                    "--cluster-ids", args.cluster_ids, "--dtype", args.dtype, "--structure", args.structure,
                    "--steps-per-graph", str(args.steps_per_graph), "--seed", str(args.seed),
                    "--no-cpu-baseline", "--no-streaming-spmm", "--no-stage-a", "--no-other-ids", "--no-strong",
                    "--no-allreduce-ab"] + (["--batch", str(args.batch)] if args.batch else [])
            t0 = time.perf_counter()
            rc, line, rest = launch_ranks(world, argv, timeout=240)
            ab = {"algorithm": other, "exit_code": rc, "wall_s": round(time.perf_counter() - t0, 1)}
            if rc == 0 and line is not None:
                ab.update({"ms_per_step": line["ms_per_step"], "graphs_per_s": line["value"],
                           "repeats": line.get("repeats"), "vs_this_line": line["value"] / value})
            else:
                ab["error"] = rest[-600:]
            out["allreduce_ab"] = ab
        print(json.dumps(out))


if __name__ == "__main__":
    main()
