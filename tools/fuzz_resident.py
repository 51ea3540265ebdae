#!/usr/bin/env python3
"""Randomised parity run of the graph-resident HSCN step against the CPU oracle: random graph sizes and edge
lists (self loops, repeated edges, isolated nodes, hubs, asymmetric edges), random K / H / L / C / activation,
both launch shapes (virtual branch fused into the forward, or riding on the two launches), loss tail on the
backward launch, and the one-launch training step (structure built per step and pre-built).  Not part of the test suite (minutes of oracle time); prints one line per case.

  python tools/fuzz_resident.py [cases] [seed]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-hscn_amd")]
import numpy as np
import torch

from graph_hscn.config.config import ACT_DICT
from graph_hscn.data import Data, HeteroBatch
from graph_hscn.loader.hetero_data import hetero_from_clusters
from graph_hscn.loss import criterion
from graph_hscn.model.hscn import HSCN
from graph_hscn import engine
from oracle import hetero_data as OH
from oracle import models as OM

DEV = "cuda"


def rand_graph(rng, F, C, nmax):
    n = int(rng.integers(1, nmax + 1))
    kind = rng.integers(0, 5)
    if kind == 0 or n == 1:
        e = np.zeros((2, 0), dtype=np.int64)
    else:
        m = int(rng.integers(1, 4 * n))
        e = rng.integers(0, n, (2, m))
        if kind == 1:                                   # symmetric, no loops (molecule-like)
            e = e[:, e[0] != e[1]]
            e = np.concatenate([e, e[::-1]], 1)
        elif kind == 2:                                 # a hub: node 0 receives from everybody, twice
            hub = np.stack([np.arange(n), np.zeros(n, dtype=np.int64)])
            e = np.concatenate([e, hub, hub], 1)
        elif kind == 3:                                 # nodes >= n/2 isolated
            e = e % max(n // 2, 1)
        # kind 4: raw random (self loops and repeats stay)
    x = torch.from_numpy(rng.integers(0, 6, (n, F))).float()
    y = torch.from_numpy((rng.random((1, C)) < 0.3).astype(np.float32))
    return Data(x=x, edge_index=torch.from_numpy(np.ascontiguousarray(e)).long(), y=y, num_nodes=n)


def close(a, b, atol, rtol):
    a, b = a.detach().cpu().double(), b.detach().double()
    return bool(((a - b).abs() <= atol + rtol * b.abs()).all()), float((a - b).abs().max())


def run(cases, seed, verbose=True):
    """Returns (mismatching cases, refused cases): a refusal is the engine declining a graph that does not
    fit a CU's LDS (loud, not wrong)."""
    rng = np.random.default_rng(seed)
    bad, refused = 0, 0
    for case in range(cases):
        H = int(rng.choice([16, 16, 32, 64]))
        L = int(rng.integers(1, 4))
        C = int(rng.choice([1, 3, 10, 21]))
        K = int(rng.choice([1, 4, 16, 32]))
        F = int(rng.integers(1, min(H, 14) + 1))
        B = int(rng.integers(1, 9))
        act = str(rng.choice(["relu", "elu", "tanh", "identity"]))
        loss_fn = str(rng.choice(["cross_entropy", "l1"]))
        nmax = 90 if H == 64 else (300 if H == 32 else 440)
        graphs = [rand_graph(rng, F, C, nmax) for _ in range(B)]
        ids = [rng.integers(0, K, g.num_nodes) for g in graphs]
        ob = OH.collate_hetero([OH.hetero_from_clusters(g.x, g.edge_index, g.y, i, K) for g, i in zip(graphs, ids)])
        pb = HeteroBatch.from_data_list([hetero_from_clusters(g, i, K) for g, i in zip(graphs, ids)]).to(DEV)
        torch.manual_seed(case)
        om = OM.HSCN("GAT", "GCN", "GCN", OM.ACT[act], F, H, C, L)
        with torch.no_grad():
            for n_, p in om.named_parameters():
                if n_.endswith("bias"):
                    p.normal_(0, 0.1)
        pm = HSCN("GAT", "GCN", "GCN", ACT_DICT[act], F, H, C, L).to(DEV)
        pm.load_state_dict(om.state_dict())
        pm.engine = "resident"
        out_o = om(ob["x_dict"], ob["edge_index_dict"], ob["batch_local"], B)
        lo, so = OM.criterion(loss_fn, out_o, ob["y"])
        lo.backward()
        xo = ob["x_dict"]
        for conv in om.convs:
            xo = {k: v.relu() for k, v in conv(xo, ob["edge_index_dict"]).items()}
        # float64 referee for the virtual features (two float32 evaluations of one function: tests/test_gpu_resident.py)
        import copy
        with torch.no_grad():
            x64 = {k: v.double() for k, v in ob["x_dict"].items()}
            for conv in copy.deepcopy(om).double().convs:
                x64 = {k: v.relu() for k, v in conv(x64, ob["edge_index_dict"]).items()}
        v64 = x64["virtual"]

        def virtual_ok(xv):
            """1e-5 of the features' magnitude, or -- where float32 itself is further than that from the true value --
            no further from the float64 evaluation than the float32 oracle is (+ 4 ulp of the magnitude)."""
            ref_v = xo["virtual"]
            got = xv[: ref_v.size(0)].detach().cpu().double()
            scale = max(1.0, float(v64.abs().max()))
            d = float((got - ref_v.double()).abs().max())
            if d <= 1e-5 * scale:
                return True, d
            e_hip = float((got - v64).abs().max())
            e_o32 = float((ref_v.detach().double() - v64).abs().max())
            return e_hip <= e_o32 + 4 * 2.0 ** -23 * scale, d
        msgs = []
        for overlap in (False, True):
            pm.overlap_virtual, pm.keep_virtual = overlap, not overlap
            pm.zero_grad(set_to_none=True)
            try:
                out_d = pm(pb.x_dict, pb.edge_index_dict, pb)
            except RuntimeError as e:                   # a graph too large for the LDS-resident engine
                if "does not qualify" not in str(e):
                    raise
                msgs.append("refused: " + str(e)[:60])
                break
            if pm.last_engine != "resident":
                msgs.append("fell back to layered")
                break
            ld, sd = criterion(loss_fn, out_d, pb["local"].y)
            ld.backward()
            torch.cuda.synchronize()
            pb._resident_meta.check()
            ok, d = close(out_d, out_o, 1e-5, 1e-5)
            if not ok:
                msgs.append(f"pred overlap={overlap} maxdiff {d:.2e}")
            if abs(float(ld) - float(lo.detach())) > 1e-5 * max(1.0, abs(float(lo.detach()))):
                msgs.append(f"loss {float(ld)} vs {float(lo.detach())}")
            ok, d = close(sd, so, 1e-5, 1e-5)
            if not ok:
                msgs.append(f"score maxdiff {d:.2e}")
            xv = engine.last_deferred_virtual if (overlap and L >= 2 and xo["virtual"].size(0) > 0 and 2 * B <= 256) else pm.last_virtual
            if xv is not None and xo["virtual"].size(0) > 0:
                # 1e-5 relative to the magnitude of the features (tests/helpers.py::scale_close; HIP is shown to be as
                # close to float64 as the float32 oracle in tests/test_gpu_resident.py)
                ok, d = virtual_ok(xv)
                if not ok:
                    msgs.append(f"virtual overlap={overlap} maxdiff {d:.2e}")
            for (n_, po), (_, pd) in zip(om.named_parameters(), pm.named_parameters()):
                if po.grad is None:
                    if pd.grad is not None:
                        msgs.append(f"unexpected grad {n_}")
                    continue
                ok, d = close(pd.grad, po.grad, 1e-5, 2e-3)
                if not ok:
                    msgs.append(f"grad {n_} overlap={overlap} maxdiff {d:.2e} (ref max {float(po.grad.abs().max()):.2e})")
        # the same step as ONE launch (graph_hscn.step.ResidentTrainStep: forward, loss row and backward of a graph in
        # one workgroup, virtual branch on its own workgroups through the in-launch hand-off), building its structure
        # per step and loading it pre-built: against the oracle at the tolerances above
        if not msgs:
            from graph_hscn.step import ResidentTrainStep
            pm.overlap_virtual, pm.keep_virtual = True, False
            for mode in ("per-step", "dataset-resident"):
                try:
                    rs = ResidentTrainStep(pm, pb, loss_fn, one_launch=True,
                                           structure=engine.build_structure(pb) if mode == "dataset-resident" else None)
                except RuntimeError:
                    break                                # H = 64 / graphs beyond its LDS layout: the launch pair is the route
                rs.bind_grads()
                rs.run()
                torch.cuda.synchronize()
                rs.check()
                ok, d = close(rs.pred, out_o, 1e-5, 1e-5)
                if not ok:
                    msgs.append(f"one-launch ({mode}) pred maxdiff {d:.2e}")
                if abs(float(rs.loss) - float(lo.detach())) > 1e-5 * max(1.0, abs(float(lo.detach()))):
                    msgs.append(f"one-launch ({mode}) loss {float(rs.loss)} vs {float(lo.detach())}")
                if rs.virtual is not None and xo["virtual"].size(0) > 0 and rs.idle_cus:
                    ok, d = virtual_ok(rs.virtual)
                    if not ok:
                        msgs.append(f"one-launch ({mode}) virtual maxdiff {d:.2e}")
                for (n_, po), (_, pd) in zip(om.named_parameters(), pm.named_parameters()):
                    if po.grad is None:
                        continue
                    ok, d = close(pd.grad, po.grad, 1e-5, 2e-3)
                    if not ok:
                        msgs.append(f"one-launch ({mode}) grad {n_} maxdiff {d:.2e} (ref max {float(po.grad.abs().max()):.2e})")
        is_refusal = bool(msgs) and msgs[0].startswith("refused")
        tag = "ok " if not msgs else ("REFUSED" if is_refusal else "BAD")
        bad += bool(msgs) and not is_refusal
        refused += is_refusal
        sizes = [g.num_nodes for g in graphs]
        if verbose:
            print(f"{tag} case {case}: H={H} L={L} C={C} K={K} F={F} B={B} act={act} {loss_fn} n={sizes} {'; '.join(msgs)}", flush=True)
    return bad, refused


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad, refused = run(cases, seed)
    print(f"{cases - bad - refused}/{cases} cases match the oracle, {refused} refused (graph too large for one CU's LDS), {bad} mismatch")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
