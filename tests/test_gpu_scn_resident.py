"""Fused graph-resident stage A (csrc/resident_scn.hip): gcn_norm + SCN.forward + MinCUT losses
and its backward vs the CPU oracle's per-graph loop body (train_clustering.py:37-50)."""
import numpy as np
import pytest
import torch

from oracle import models as OM
from tests.helpers import ATOL, DEV, close

pytestmark = pytest.mark.gpu


def _models(K, H=16, act="elu", F=9, seed=0):
    from graph_hscn.model.hscn import SCN
    torch.manual_seed(seed)
    om = OM.SCN([H], act, F, K)
    pm = SCN([H], act, F, K).to(DEV)
    pm.load_state_dict(om.state_dict())
    return om, pm


@pytest.mark.parametrize("K,H,act,name", [(16, 16, "elu", "peptides_func"), (4, 16, "tanh", "peptides_func"),
                                          (32, 16, "relu", "pcqm_contact"), (6, 32, "elu", "pcqm_contact"),
                                          (16, 16, "identity", "pascalvoc_sp")])
def test_single_graph_step_matches_oracle(K, H, act, name):
    from graph_hscn.loader.synthetic import SHAPES, make_dataset
    F = SHAPES[name].num_features
    om, pm = _models(K, H, act, F, seed=K)
    for g in make_dataset(name, 3, seed=K + 1):
        om.zero_grad(); pm.zero_grad()
        S_o, mc_o, o_o, *_ = OM.scn_step_single_graph(om, g.x, g.edge_index)
        (mc_o + 0.5 * o_o).backward()
        fits = pm.resident_ok(g)           # (since the backward keeps x / agg in registers PascalVOC-SP sizes fit too)
        assert fits or name == "pascalvoc_sp"
        S_d, mc_d, o_d = pm.forward_graphs(g)
        assert pm.last_engine == ("resident" if fits else "layered")
        (mc_d + 0.5 * o_d).backward()
        if fits:
            g._scn_meta.check()
        assert close(S_d, S_o)
        assert abs(mc_d.item() - mc_o.item()) < ATOL and abs(o_d.item() - o_o.item()) < ATOL
        assert np.array_equal(S_d.max(1)[1].cpu().numpy(), OM.assign_clusters(S_o)) or \
            float((S_o.topk(2, 1).values[:, 0] - S_o.topk(2, 1).values[:, 1]).min()) < 1e-6
        for (n_, po), (_, pp) in zip(om.named_parameters(), pm.named_parameters()):
            assert close(pp.grad, po.grad, atol=1e-4, rtol=2e-3), n_


def test_batched_equals_mean_of_singles_and_layered():
    from graph_hscn.data import Batch
    from graph_hscn.loader.synthetic import make_dataset
    _, pm = _models(16)
    graphs = make_dataset("peptides_func", 9, seed=3)
    with torch.no_grad():
        singles = [pm.forward_graphs(g) for g in graphs]
    big = Batch.from_data_list(graphs)
    pm.zero_grad()
    S, mc, o = pm.forward_graphs(big)
    assert pm.last_engine == "resident"
    (mc + o).backward()
    gr = {n: p.grad.clone() for n, p in pm.named_parameters()}
    assert close(S, torch.cat([s[0] for s in singles]), atol=1e-6)
    assert abs(mc.item() - torch.stack([s[1] for s in singles]).mean().item()) < 1e-6
    assert abs(o.item() - torch.stack([s[2] for s in singles]).mean().item()) < 1e-6
    # the layered operators on the same batch
    from graph_hscn.nn import gcn_norm
    pm.zero_grad()
    ei, ew = gcn_norm(big.edge_index.to(DEV), None, big.num_nodes, add_self_loops=True)
    S2, mc2, o2, _ = pm(big.x.to(DEV).float(), ei, ew, node_ptr=big.ptr.to(DEV).to(torch.int32))
    (mc2 + o2).backward()
    assert close(S, S2, atol=1e-6) and abs(mc.item() - mc2.item()) < 1e-6 and abs(o.item() - o2.item()) < 1e-6
    for n, p in pm.named_parameters():
        assert close(gr[n], p.grad, atol=1e-5, rtol=1e-3), n


def test_existing_self_loops_become_the_unit_loop():
    from graph_hscn.data import Data
    om, pm = _models(4)
    g = torch.Generator().manual_seed(0)
    ei = torch.tensor([[0, 1, 1, 2, 2, 3, 1, 0], [1, 0, 2, 1, 2, 3, 1, 0]])       # self loops at 2, 3, 1, 0
    x = torch.randint(0, 9, (4, 9), generator=g)
    S_o, mc_o, o_o, *_ = OM.scn_step_single_graph(om, x, ei)
    S_d, mc_d, o_d = pm.forward_graphs(Data(x=x, edge_index=ei, num_nodes=4))
    assert close(S_d, S_o) and abs(mc_d.item() - mc_o.item()) < ATOL and abs(o_d.item() - o_o.item()) < ATOL


def test_reproducible_and_flags_bad_edges():
    from graph_hscn.data import Batch
    from graph_hscn.loader.synthetic import make_dataset
    _, pm = _models(16)
    big = Batch.from_data_list(make_dataset("peptides_func", 6, seed=1))
    outs = []
    for _ in range(2):
        pm.zero_grad()
        S, mc, o = pm.forward_graphs(big)
        (mc + o).backward()
        outs.append((S.clone(), mc.clone(), [p.grad.clone() for p in pm.parameters()]))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert all(torch.equal(a, b) for a, b in zip(outs[0][2], outs[1][2]))
    big.edge_index[0, 0] = big.num_nodes - 1
    del big._d["_scn_meta"]
    pm.forward_graphs(big)
    with pytest.raises(IndexError):
        big._scn_meta.check()


def test_total_loss_from_the_launch_equals_the_sum_and_its_gradients():
    """forward_graphs(with_total=True)[3] (mc + o written by the forward launch, reduced by the
    workgroup that finishes last) == mc + o, and backward through it == backward through the explicit
    sum, over repeated launches (the ticket counter returns to zero) and under a captured replay."""
    from graph_hscn.data import Batch
    from graph_hscn.loader.synthetic import make_dataset
    _, pm = _models(16)
    big = Batch.from_data_list(make_dataset("peptides_func", 40, seed=2)).to(DEV)
    big.x = big.x.float()
    ref = None
    for rep in range(3):
        pm.zero_grad(set_to_none=True)
        S, mc, o, total = pm.forward_graphs(big, with_total=True)
        assert torch.equal(total, mc + o)
        (mc + o).backward()
        g_sum = [p.grad.clone() for p in pm.parameters()]
        pm.zero_grad(set_to_none=True)
        pm.forward_graphs(big, with_total=True)[3].backward()
        g_tot = [p.grad.clone() for p in pm.parameters()]
        assert all(torch.equal(a, b) for a, b in zip(g_sum, g_tot))
        assert int(big._scn_meta.ticket.item()) == 0
        if ref is None:
            ref = (mc.detach().clone(), o.detach().clone(), g_tot)
        else:
            assert torch.equal(mc, ref[0]) and torch.equal(o, ref[1])
    del S, mc, o, total

    def step():
        pm.zero_grad(set_to_none=True)
        t = pm.forward_graphs(big, with_total=True)[3]
        t.backward()
        return t.detach()

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        tot = step()
    for _ in range(2):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(tot, ref[0] + ref[1])
    assert all(torch.equal(p.grad, g) for p, g in zip(pm.parameters(), ref[2]))


@pytest.mark.parametrize("K", [5, 16])
def test_degenerate_graphs_match_the_oracle_loop_body(K):
    """A single node without edges, nodes without edges, a two-node graph, a star: each through the fused
    launches as a graph of its own and as one batch (odd K takes the scalar paths of the kernels)."""
    from graph_hscn.data import Batch, Data
    om, pm = _models(K, 16, "elu", 9, seed=7)
    g = torch.Generator().manual_seed(3)

    def mk(n, edges):
        ei = torch.tensor(edges, dtype=torch.int64).t().reshape(2, -1) if edges else torch.zeros(2, 0, dtype=torch.int64)
        return Data(x=torch.randint(0, 7, (n, 9), generator=g).float(), edge_index=ei, num_nodes=n)

    star = [(0, i) for i in range(1, 9)] + [(i, 0) for i in range(1, 9)]
    graphs = [mk(1, []), mk(4, []), mk(2, [(0, 1), (1, 0)]), mk(9, star)]
    ref = []
    for gr in graphs:
        om.zero_grad(); pm.zero_grad()
        S_o, mc_o, o_o, *_ = OM.scn_step_single_graph(om, gr.x, gr.edge_index)
        (mc_o + o_o).backward()
        S_d, mc_d, o_d = pm.forward_graphs(gr)
        assert pm.last_engine == "resident"
        (mc_d + o_d).backward()
        gr._scn_meta.check()
        assert close(S_d, S_o)
        assert abs(mc_d.item() - mc_o.item()) < ATOL and abs(o_d.item() - o_o.item()) < ATOL
        for (n_, po), (_, pp) in zip(om.named_parameters(), pm.named_parameters()):
            assert close(pp.grad, po.grad, atol=1e-4, rtol=2e-3), n_
        ref.append((mc_o.item(), o_o.item()))
    big = Batch.from_data_list(graphs)
    with torch.no_grad():
        _, mc, o = pm.forward_graphs(big)
    assert abs(mc.item() - np.mean([r[0] for r in ref])) < ATOL
    assert abs(o.item() - np.mean([r[1] for r in ref])) < ATOL
