"""Graph-resident fused engine (csrc/resident.hip) vs the CPU oracle and vs the
layered per-operator engine."""
import numpy as np
import pytest
import torch

from oracle import hetero_data as OH
from oracle import models as OM
from tests.helpers import ATOL, DEV, close, scale_close

pytestmark = pytest.mark.gpu


def _batches(name, B, K, seed):
    from graph_hscn.data import HeteroBatch
    from graph_hscn.loader.hetero_data import hetero_from_clusters
    from graph_hscn.loader.synthetic import make_dataset
    graphs = make_dataset(name, B, seed=seed)
    rng = np.random.default_rng(seed)
    ids = [rng.integers(0, K, g.num_nodes) for g in graphs]
    ob = OH.collate_hetero([OH.hetero_from_clusters(g.x, g.edge_index, g.y, i, K) for g, i in zip(graphs, ids)])
    pb = HeteroBatch.from_data_list([hetero_from_clusters(g, i, K) for g, i in zip(graphs, ids)])
    return ob, pb


def _models(F, H, C, L, act="relu", seed=0):
    from graph_hscn.config.config import ACT_DICT
    from graph_hscn.model.hscn import HSCN
    torch.manual_seed(seed)
    om = OM.HSCN("GAT", "GCN", "GCN", OM.ACT[act], F, H, C, L)
    with torch.no_grad():
        for n_, p in om.named_parameters():
            if n_.endswith("bias"):
                p.normal_(0, 0.1)
    pm = HSCN("GAT", "GCN", "GCN", ACT_DICT[act], F, H, C, L).to(DEV)
    pm.load_state_dict(om.state_dict())
    return om, pm


@pytest.mark.parametrize("name,B,K,H,L,C,act", [
    ("peptides_func", 6, 16, 16, 3, 10, "relu"), ("peptides_struct", 5, 32, 32, 2, 11, "elu"),
    ("pcqm_contact", 9, 16, 16, 3, 1, "tanh"), ("pascalvoc_sp", 3, 64, 16, 2, 21, "relu"),
    ("pcqm_contact", 7, 16, 64, 1, 10, "identity"), ("peptides_func", 40, 4, 16, 3, 10, "relu")])
def test_resident_matches_oracle(name, B, K, H, L, C, act):
    ob, pb = _batches(name, B, K, seed=B + K)
    F = ob["x_dict"]["local"].size(1)
    om, pm = _models(F, H, C, L, act, seed=B)
    pm.engine, pm.keep_virtual = "resident", True
    pbd = pb.to(DEV)
    out_o = om(ob["x_dict"], ob["edge_index_dict"], ob["batch_local"], B)
    out_d = pm(pbd.x_dict, pbd.edge_index_dict, pbd)
    assert pm.last_engine == "resident"
    pbd._resident_meta.check()
    assert close(out_d, out_o, atol=ATOL, rtol=1e-5)
    # virtual branch (unused by pred): compare the final virtual features directly
    xo = ob["x_dict"]
    for conv in om.convs:
        xo = {k: v.relu() for k, v in conv(xo, ob["edge_index_dict"]).items()}
    assert scale_close(pm.last_virtual, xo["virtual"])
    g = torch.randn(B, C, generator=torch.Generator().manual_seed(1))
    out_o.backward(g)
    out_d.backward(g.to(DEV))
    for (n_, po), (_, pp) in zip(om.named_parameters(), pm.named_parameters()):
        if po.grad is None:
            assert pp.grad is None, n_
        else:
            assert close(pp.grad, po.grad, atol=1e-4, rtol=1e-3), n_


CONFIGS = [("peptides_func", 6, 16, 16, 3, 10, "relu"), ("peptides_struct", 5, 32, 32, 2, 11, "elu"),
           ("pcqm_contact", 9, 16, 16, 3, 1, "tanh"), ("pascalvoc_sp", 3, 64, 16, 2, 21, "relu"),
           ("pcqm_contact", 7, 16, 64, 1, 10, "identity"), ("peptides_func", 40, 4, 16, 3, 10, "relu")]


@pytest.mark.parametrize("name,B,K,H,L,C,act", CONFIGS)
def test_hip_is_as_close_to_float64_as_the_float32_oracle(name, B, K, H, L, C, act):
    """The resident kernels reorder two linear maps ((A_hat X) W^T for the local chain,
    W (sum alpha x) for the virtual branch), so HIP and the float32 oracle are two DIFFERENT float32 roundings of
    the same real-valued function.  The honest question is not "how far apart are they" but "is either one further
    from the true value": evaluate the oracle in float64 and require, in max norm over each output,
        |HIP - f64|  <=  |oracle_f32 - f64| + 4 ulp(scale)        (ulp(scale) = 2^-23 * max|f64|).
    The distance between the two float32 evaluations is then at most 2 |oracle_f32 - f64| + 4 ulp: that -- not a
    loosened constant -- is the tolerance test_resident_matches_oracle applies to the virtual features."""
    import copy
    ob, pb = _batches(name, B, K, seed=B + K)
    F = ob["x_dict"]["local"].size(1)
    om, pm = _models(F, H, C, L, act, seed=B)
    pm.engine, pm.keep_virtual = "resident", True
    pbd = pb.to(DEV)
    with torch.no_grad():
        out_d = pm(pbd.x_dict, pbd.edge_index_dict, pbd).cpu().double()
        xv_d = pm.last_virtual.cpu().double()
        o64 = copy.deepcopy(om).double()
        res = {}
        for tag, m, cast in (("f32", om, torch.float32), ("f64", o64, torch.float64)):
            x = {k: v.to(cast) for k, v in ob["x_dict"].items()}
            pred = m(x, ob["edge_index_dict"], ob["batch_local"], B)
            for conv in m.convs:
                x = {k: v.relu() for k, v in conv(x, ob["edge_index_dict"]).items()}
            res[tag] = (pred.double(), x["virtual"].double())
    for what, hip, i in (("prediction", out_d, 0), ("virtual features", xv_d, 1)):
        ref64, ref32 = res["f64"][i], res["f32"][i]
        e_hip = float((hip - ref64).abs().max())
        e_o32 = float((ref32 - ref64).abs().max())
        ulp = 2.0 ** -23 * max(1.0, float(ref64.abs().max()))
        print(f"[f64 check] {name} H={H} L={L} {what}: |HIP-f64|={e_hip:.3e} |oracle32-f64|={e_o32:.3e} "
              f"|HIP-oracle32|={float((hip - ref32).abs().max()):.3e} ulp(scale)={ulp:.3e}")
        assert e_hip <= e_o32 + 4 * ulp, (what, e_hip, e_o32, ulp)


def test_resident_equals_layered_engine():
    ob, pb = _batches("peptides_func", 12, 16, seed=3)
    _, pm = _models(9, 16, 10, 3)
    pbd = pb.to(DEV)
    outs, grads = {}, {}
    for eng in ("resident", "layered"):
        pm.engine = eng
        pm.zero_grad(set_to_none=True)
        out = pm(pbd.x_dict, pbd.edge_index_dict, pbd)
        assert pm.last_engine == eng
        out.square().sum().backward()
        outs[eng] = out.detach().clone()
        grads[eng] = {n: p.grad.clone() for n, p in pm.named_parameters() if p.grad is not None}
    assert close(outs["resident"], outs["layered"], atol=1e-6, rtol=1e-6)
    assert grads["resident"].keys() == grads["layered"].keys()
    for n in grads["resident"]:
        assert close(grads["resident"][n], grads["layered"][n], atol=1e-5, rtol=1e-4), n


def test_resident_is_reproducible_bitwise():
    _, pb = _batches("peptides_func", 16, 16, seed=5)
    _, pm = _models(9, 16, 10, 3)
    pm.engine = "resident"
    pbd = pb.to(DEV)
    runs = []
    for _ in range(3):
        pm.zero_grad(set_to_none=True)
        out = pm(pbd.x_dict, pbd.edge_index_dict, pbd)
        out.sum().backward()
        runs.append((out.detach().clone(), [p.grad.clone() for p in pm.parameters() if p.grad is not None]))
    for out, gs in runs[1:]:
        assert torch.equal(out, runs[0][0])
        assert all(torch.equal(a, b) for a, b in zip(gs, runs[0][1]))


def test_resident_skipping_virtual_branch_leaves_prediction_unchanged():
    _, pb = _batches("peptides_func", 8, 16, seed=7)
    _, pm = _models(9, 16, 10, 3)
    pm.engine = "resident"
    pbd = pb.to(DEV)
    a = pm(pbd.x_dict, pbd.edge_index_dict, pbd).detach().clone()
    pm.compute_virtual = False
    b = pm(pbd.x_dict, pbd.edge_index_dict, pbd).detach()
    assert torch.equal(a, b)


def test_resident_flags_edges_that_cross_graphs():
    _, pb = _batches("peptides_func", 4, 8, seed=9)
    _, pm = _models(9, 16, 10, 2)
    pm.engine = "resident"
    pbd = pb.to(DEV)
    ei = pbd[("local", "to", "local")].edge_index
    ei[0, 0] = pbd["local"].num_nodes - 1          # first graph's edge now starts in the last graph
    pm(pbd.x_dict, pbd.edge_index_dict, pbd)
    with pytest.raises(IndexError):
        pbd._resident_meta.check()


def test_auto_engine_falls_back_to_layered_for_foreign_batches():
    ob, _ = _batches("peptides_func", 3, 8, seed=1)
    _, pm = _models(9, 16, 10, 2)

    class Foreign(dict):
        num_graphs = 3
    fb = Foreign()

    class L_:
        batch = ob["batch_local"].to(DEV)
    fb["local"] = L_()
    out = pm({k: v.to(DEV) for k, v in ob["x_dict"].items()},
             {k: v.to(DEV) for k, v in ob["edge_index_dict"].items()}, fb)
    assert pm.last_engine == "layered" and out.shape == (3, 10)


def test_resident_gradients_tile_one_flat_buffer_for_the_allreduce():
    """The data-parallel reducer all-reduces the backward's single grads[P] buffer in place."""
    from graph_hscn.distributed import FlatGradReducer
    _, pb = _batches("peptides_func", 4, 8, seed=2)
    _, pm = _models(9, 16, 10, 3)
    pm.engine = "resident"
    pbd = pb.to(DEV)
    pm(pbd.x_dict, pbd.edge_index_dict, pbd).sum().backward()
    red = FlatGradReducer(pm)
    grads = [p.grad for p in pm.parameters() if p.grad is not None]
    flat = red._aliased_flat(grads)
    assert flat is not None and flat.numel() == sum(g.numel() for g in grads) == 1146
    before = [g.clone() for g in grads]
    flat.mul_(2.0)
    assert all(torch.equal(g, 2 * b) for g, b in zip(grads, before))


@pytest.mark.parametrize("name,B,K,H,L", [("peptides_func", 12, 16, 16, 3), ("pascalvoc_sp", 3, 64, 16, 2),
                                          ("pcqm_contact", 9, 16, 64, 2), ("peptides_struct", 5, 8, 32, 3)])
def test_virtual_branch_riding_on_the_backward_launch_equals_the_fused_forward(name, B, K, H, L):
    """Forward mode 0 + hscn_resident_bwd_with_virtual (the virtual branch as extra workgroups of the
    backward launch, reading the stored activations) == the one-launch forward + plain backward,
    bit for bit: prediction, final virtual features, gradients."""
    from graph_hscn import engine
    ob, pb = _batches(name, B, K, seed=3)
    F = ob["x_dict"]["local"].size(1)
    _, pm = _models(F, H, 10, L, "relu", seed=4)
    pm.engine = "resident"
    pbd = pb.to(DEV)
    g = torch.randn(B, 10, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    res = {}
    for overlap in (False, True):
        pm.overlap_virtual, pm.keep_virtual = overlap, not overlap
        engine.last_deferred_virtual = None
        pm.zero_grad(set_to_none=True)
        out = pm(pbd.x_dict, pbd.edge_index_dict, pbd)
        out.backward(g)
        torch.cuda.synchronize()
        pbd._resident_meta.check()
        xv = engine.last_deferred_virtual if overlap else pm.last_virtual
        assert xv is not None
        res[overlap] = (out.detach().clone(), xv.clone(),
                        {n: p.grad.clone() for n, p in pm.named_parameters() if p.grad is not None})
    assert torch.equal(res[False][0], res[True][0])
    assert torch.equal(res[False][1], res[True][1])
    assert res[False][2].keys() == res[True][2].keys()
    for n in res[False][2]:
        assert torch.equal(res[False][2][n], res[True][2][n]), n


def test_deferred_virtual_branch_inside_a_captured_step():
    """A captured training step (forward mode 0, loss, backward carrying the virtual branch)
    replays to the eager result."""
    from graph_hscn.loss import criterion
    B, K, H, L, C = 16, 16, 16, 3, 10
    ob, pb = _batches("peptides_func", B, K, seed=5)
    F = ob["x_dict"]["local"].size(1)
    _, pm = _models(F, H, C, L, "relu", seed=6)
    pm.engine, pm.keep_virtual, pm.overlap_virtual = "resident", False, True
    pbd = pb.to(DEV)
    y = (torch.rand(B, C, device=DEV) > 0.5).float()

    def step():
        pm.zero_grad(set_to_none=True)
        out = pm(pbd.x_dict, pbd.edge_index_dict, pbd)
        loss, _ = criterion("cross_entropy", out, y)
        loss.backward()
        return loss

    eager_loss = step().detach().clone()
    eager = {n: p.grad.clone() for n, p in pm.named_parameters() if p.grad is not None}
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        loss = step()
    for n, p in pm.named_parameters():
        if p.grad is not None:
            p.grad.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(loss, eager_loss)
    for n, p in pm.named_parameters():
        if n in eager:
            assert torch.equal(p.grad, eager[n]), n


def test_degenerate_graphs_in_a_batch():
    """Ragged / empty inputs through both launches of a training step: a single node without edges, a
    graph with nodes but no edges, a graph whose nodes all fall in one cluster, a two-node graph, next to
    ordinary graphs.  Compared with the CPU oracle (prediction, final virtual features, gradients)."""
    from graph_hscn.data import Data, HeteroBatch
    from graph_hscn.loader.hetero_data import hetero_from_clusters
    from graph_hscn.loader.synthetic import make_dataset
    from graph_hscn import engine
    K, H, L, C = 8, 16, 3, 10
    normal = make_dataset("peptides_func", 3, seed=11)
    F = normal[0].x.size(1)
    g = torch.Generator().manual_seed(5)

    def mk(n, edges):
        ei = torch.tensor(edges, dtype=torch.int64).t().reshape(2, -1) if edges else torch.zeros(2, 0, dtype=torch.int64)
        return Data(x=torch.randint(0, 5, (n, F), generator=g).float(), edge_index=ei,
                    y=torch.zeros(1, C), num_nodes=n)

    graphs = [normal[0], mk(1, []), mk(5, []), normal[1], mk(7, [(0, 1), (1, 0), (2, 3), (3, 2), (5, 6), (6, 5)]),
              mk(2, [(0, 1), (1, 0)]), normal[2]]
    rng = np.random.default_rng(3)
    ids = [rng.integers(0, K, gr.num_nodes) for gr in graphs]
    ids[4] = np.full(7, 3)                       # one cluster only
    ob = OH.collate_hetero([OH.hetero_from_clusters(gr.x, gr.edge_index, gr.y, i, K) for gr, i in zip(graphs, ids)])
    pb = HeteroBatch.from_data_list([hetero_from_clusters(gr, i, K) for gr, i in zip(graphs, ids)]).to(DEV)
    B = len(graphs)
    om, pm = _models(F, H, C, L, "relu", seed=9)
    pm.engine = "resident"
    out_o = om(ob["x_dict"], ob["edge_index_dict"], ob["batch_local"], B)
    gsel = torch.randn(B, C, generator=torch.Generator().manual_seed(1))
    (out_o * gsel).sum().backward()
    xo = ob["x_dict"]
    for conv in om.convs:
        xo = {k: v.relu() for k, v in conv(xo, ob["edge_index_dict"]).items()}
    for overlap in (False, True):
        pm.overlap_virtual, pm.keep_virtual = overlap, not overlap
        pm.zero_grad(set_to_none=True)
        out_d = pm(pb.x_dict, pb.edge_index_dict, pb)
        (out_d * gsel.to(DEV)).sum().backward()
        torch.cuda.synchronize()
        pb._resident_meta.check()
        assert close(out_d, out_o, atol=ATOL, rtol=1e-5)
        xv = engine.last_deferred_virtual if overlap else pm.last_virtual
        assert scale_close(xv[: xo["virtual"].size(0)], xo["virtual"])
        go = {n: p.grad for n, p in om.named_parameters() if p.grad is not None}
        gd = {n: p.grad for n, p in pm.named_parameters() if p.grad is not None}
        assert go.keys() == gd.keys()
        for n in go:
            assert close(gd[n], go[n], atol=1e-4, rtol=1e-3), n


def test_wide_model_on_a_large_graph_takes_the_two_buffer_backward():
    """H = 32 with a 440-node graph: the backward's three n x H buffers exceed a CU's LDS, the launch
    switches to two buffers (layer input in the dead gradient's buffer, input gradient in place).
    Prediction and gradients match the CPU oracle and the layered engine."""
    from graph_hscn.data import Data, HeteroBatch
    from graph_hscn.loader.hetero_data import hetero_from_clusters
    from graph_hscn.loader.synthetic import make_dataset
    from graph_hscn import _hip
    K, H, L, C, F = 16, 32, 3, 10, 9
    g = torch.Generator().manual_seed(2)
    n = 440
    par = torch.cat([torch.zeros(1, dtype=torch.int64), (torch.rand(n - 1, generator=g) * torch.arange(1, n)).long()])
    src = torch.arange(1, n)
    extra_a = torch.randint(0, n, (20,), generator=g)
    extra_b = (extra_a + 1 + torch.randint(0, n - 1, (20,), generator=g)) % n
    u = torch.cat([src, extra_a]); v = torch.cat([par[1:], extra_b])
    ei = torch.stack([torch.cat([u, v]), torch.cat([v, u])])
    big = Data(x=torch.randint(0, 5, (n, F), generator=g).float(), edge_index=ei, y=torch.zeros(1, C), num_nodes=n)
    graphs = make_dataset("peptides_func", 2, seed=5) + [big]
    assert _hip.lib().hscn_resident_supported(F, H, L, C, n, K, ei.size(1), K * (K + 1) // 2) == 1
    rng = np.random.default_rng(0)
    ids = [rng.integers(0, K, gr.num_nodes) for gr in graphs]
    ob = OH.collate_hetero([OH.hetero_from_clusters(gr.x, gr.edge_index, gr.y, i, K) for gr, i in zip(graphs, ids)])
    pb = HeteroBatch.from_data_list([hetero_from_clusters(gr, i, K) for gr, i in zip(graphs, ids)]).to(DEV)
    B = len(graphs)
    om, pm = _models(F, H, C, L, "relu", seed=3)
    out_o = om(ob["x_dict"], ob["edge_index_dict"], ob["batch_local"], B)
    gsel = torch.randn(B, C, generator=torch.Generator().manual_seed(1))
    (out_o * gsel).sum().backward()
    go = {k: p.grad for k, p in om.named_parameters() if p.grad is not None}
    res = {}
    for eng in ("resident", "layered"):
        pm.engine = eng
        pm.zero_grad(set_to_none=True)
        out_d = pm(pb.x_dict, pb.edge_index_dict, pb)
        assert pm.last_engine == eng
        (out_d * gsel.to(DEV)).sum().backward()
        torch.cuda.synchronize()
        assert close(out_d, out_o, atol=ATOL, rtol=1e-5)
        gd = {k: p.grad for k, p in pm.named_parameters() if p.grad is not None}
        assert gd.keys() == go.keys()
        for k in go:
            assert close(gd[k], go[k], atol=2e-4, rtol=1e-3), (eng, k)
    pb._resident_meta.check()


@pytest.mark.timeout(120)
def test_understated_batch_maxima_raise_the_flag_and_nothing_hangs():
    """A batch whose per-graph maxima (the host ints that size the launches' LDS) are too small: the workgroups of
    the offending graphs bail out, raise flag bit 4 and leave EMPTY structure behind, so the launches that consume
    their exports (the backward, the resumed virtual branch) terminate; ``check()`` reports it."""
    from graph_hscn.loss import criterion
    _, pb = _batches("peptides_func", 6, 8, seed=2)
    _, pm = _models(9, 16, 10, 3)
    pm.engine = "resident"
    for field, types in (("max_nodes", ("virtual",)), ("max_nodes", ("local",)), ("max_edges", (("local", "to", "local"),))):
        d = pb.to(DEV)
        for t in types:
            setattr(d[t], field, max(1, getattr(d[t], field) // 2))          # half of the real maximum
        pm.zero_grad(set_to_none=True)
        pred = pm(d.x_dict, d.edge_index_dict, d)
        loss, _ = criterion("cross_entropy", pred, d["local"].y)
        loss.backward()
        torch.cuda.synchronize()                                             # both launches of the step came back
        with pytest.raises(ValueError):
            d._resident_meta.check()
