"""SCN / HSCN through the HIP path vs the CPU oracle with identical weights:
activations within 1e-5, cluster indices bit-exact (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

from oracle import hetero_data as OH
from oracle import models as OM
from oracle import pyg_ops as P
from tests.helpers import ATOL, DEV, close, scale_close, hetero_batch

pytestmark = pytest.mark.gpu


def _to_dev(d):
    return {k: v.to(DEV) for k, v in d.items()}


class _Batch(dict):
    def __init__(self, b):
        super().__init__()
        self.num_graphs = b["num_graphs"]

        class L:
            batch = b["batch_local"].to(DEV)
        self["local"] = L()


@pytest.mark.parametrize("name,B,K,H,L,C", [("peptides_func", 6, 16, 16, 3, 10), ("peptides_struct", 5, 32, 32, 2, 11),
                                            ("pcqm_contact", 9, 16, 16, 3, 1), ("pascalvoc_sp", 2, 64, 16, 2, 21)])
def test_hscn_forward_backward_matches_oracle(name, B, K, H, L, C):
    from graph_hscn.config.config import ACT_DICT
    from graph_hscn.model.hscn import HSCN
    b, _ = hetero_batch(name, B, K, seed=B)
    F = b["x_dict"]["local"].size(1)
    torch.manual_seed(B)
    om = OM.HSCN("GAT", "GCN", "GCN", OM.ACT["relu"], F, H, C, L)
    with torch.no_grad():
        for n_, p in om.named_parameters():
            if n_.endswith("bias"):
                p.normal_(0, 0.1)
    pm = HSCN("GAT", "GCN", "GCN", ACT_DICT["relu"], F, H, C, L).to(DEV)
    assert sorted(pm.state_dict()) == sorted(om.state_dict())
    pm.load_state_dict(om.state_dict())
    out_o = om(b["x_dict"], b["edge_index_dict"], b["batch_local"], B)
    out_d = pm(_to_dev(b["x_dict"]), _to_dev(b["edge_index_dict"]), _Batch(b))
    assert close(out_d, out_o, atol=ATOL, rtol=1e-5)
    g = torch.randn(B, C, generator=torch.Generator().manual_seed(1))
    out_o.backward(g)
    out_d.backward(g.to(DEV))
    for (n_, po), (_, pp) in zip(om.named_parameters(), pm.named_parameters()):
        if po.grad is None:
            assert pp.grad is None, n_
        else:
            assert close(pp.grad, po.grad, atol=1e-4, rtol=1e-3), n_


def test_hscn_virtual_branch_activations_match():
    """The virtual branch never reaches the prediction, so check it directly:
    one HeteroConv layer's 'virtual' output (vv GCN + lv GAT, summed)."""
    from graph_hscn.model.hscn import HSCN
    from graph_hscn.config.config import ACT_DICT
    b, _ = hetero_batch("peptides_func", 5, 16, seed=3)
    torch.manual_seed(0)
    om = OM.HSCN("GAT", "GCN", "GCN", OM.ACT["relu"], 9, 16, 10, 2)
    pm = HSCN("GAT", "GCN", "GCN", ACT_DICT["relu"], 9, 16, 10, 2).to(DEV)
    pm.load_state_dict(om.state_dict())
    xo, xd = b["x_dict"], _to_dev(b["x_dict"])
    eo, ed = b["edge_index_dict"], _to_dev(b["edge_index_dict"])
    for lo, lp in zip(om.convs, pm.convs):
        xo = {k: v.relu() for k, v in lo(xo, eo).items()}
        xd = {k: v.relu() for k, v in lp(xd, ed).items()}
        assert scale_close(xd["virtual"], xo["virtual"]) and close(xd["local"], xo["local"])


@pytest.mark.parametrize("K,act,units", [(16, "elu", [16]), (4, "tanh", [16]), (32, "relu", [16, 16])])
def test_scn_single_graph_step_matches_oracle(K, act, units):
    from graph_hscn.loader.synthetic import make_dataset
    from graph_hscn.model.hscn import SCN
    from graph_hscn.nn import gcn_norm
    torch.manual_seed(K)
    om = OM.SCN(units, act, 9, K)
    pm = SCN(units, act, 9, K).to(DEV)
    assert sorted(pm.state_dict()) == sorted(om.state_dict())
    pm.load_state_dict(om.state_dict())
    for g in make_dataset("peptides_func", 3, seed=K):
        om.zero_grad(); pm.zero_grad()
        S_o, mc_o, o_o, adj_o, ei_o, ew_o = OM.scn_step_single_graph(om, g.x, g.edge_index)
        (mc_o + o_o).backward()
        ei, ew = gcn_norm(g.edge_index.to(DEV), None, g.num_nodes, add_self_loops=True)
        S_d, mc_d, o_d, adj_d = pm(g.x.to(DEV).float(), ei, ew)
        (mc_d + o_d).backward()
        assert close(S_d, S_o)
        assert abs(mc_d.item() - mc_o.item()) < ATOL and abs(o_d.item() - o_o.item()) < ATOL
        assert torch.equal(adj_d.cpu(), adj_o)
        for (n_, po), (_, pp) in zip(om.named_parameters(), pm.named_parameters()):
            assert close(pp.grad, po.grad, atol=1e-4, rtol=2e-3), n_


def test_cluster_assignment_bit_exact_over_many_graphs():
    """argmax ids from the HIP path == oracle ids on EVERY node of 64 graphs (train_clustering.py:68), through the
    layered operators and through the fused stage-A launch; no flip is tolerated on these committed seeds.  Prints
    the histogram of the oracle's top-2 margins (how close the nearest tie is)."""
    from graph_hscn import _hip
    from graph_hscn.loader.synthetic import make_dataset
    from graph_hscn.model.hscn import SCN
    from graph_hscn.nn import gcn_norm
    torch.manual_seed(7)
    K = 16
    om = OM.SCN([16], "elu", 9, K)
    pm = SCN([16], "elu", 9, K).to(DEV)
    pm.load_state_dict(om.state_dict())
    margins = []
    flips = {"layered": 0, "resident": 0}
    with torch.no_grad():
        for g in make_dataset("peptides_func", 64, seed=11):
            S_o, *_ = OM.scn_step_single_graph(om, g.x, g.edge_index)
            top2 = S_o.topk(2, dim=1).values
            margins.append((top2[:, 0] - top2[:, 1]))
            want = OM.assign_clusters(S_o)
            ei, ew = gcn_norm(g.edge_index.to(DEV), None, g.num_nodes, add_self_loops=True)
            S_l, *_ = pm(g.x.to(DEV).float(), ei, ew)
            S_r, *_ = pm.forward_graphs(g.to(DEV))
            assert pm.last_engine == "resident"
            for tag, S_d in (("layered", S_l), ("resident", S_r)):
                ids = torch.empty(g.num_nodes, dtype=torch.int64, device=DEV)
                _hip.call("hscn_assign_argmax", _hip.ptr(S_d.contiguous()), _hip.ptr(ids), g.num_nodes, K, _hip.stream())
                flips[tag] += int((want != ids.cpu().numpy()).sum())
    m = torch.cat(margins)
    edges = [0.0, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 1.0]
    hist = torch.histogram(m.clamp(max=1.0), bins=torch.tensor(edges)).hist.int().tolist()
    print(f"top-2 margin histogram over {m.numel()} nodes, bin edges {edges}: {hist}; min {m.min():.3e}; flips {flips}")
    assert flips == {"layered": 0, "resident": 0}, flips


def test_scn_batched_equals_mean_of_single_graph_losses():
    from graph_hscn.data import Batch
    from graph_hscn.loader.synthetic import make_dataset
    from graph_hscn.model.hscn import SCN
    from graph_hscn.nn import gcn_norm
    torch.manual_seed(0)
    K = 16
    pm = SCN([16], "elu", 9, K).to(DEV)
    graphs = make_dataset("peptides_func", 7, seed=4)
    mcs, oos, Ss = [], [], []
    with torch.no_grad():
        for g in graphs:
            ei, ew = gcn_norm(g.edge_index.to(DEV), None, g.num_nodes, add_self_loops=True)
            S, mc, o, _ = pm(g.x.to(DEV).float(), ei, ew)
            mcs.append(mc); oos.append(o); Ss.append(S)
        big = Batch.from_data_list(graphs)
        ei, ew = gcn_norm(big.edge_index.to(DEV), None, big.num_nodes, add_self_loops=True)
        S, mc, o, adj = pm(big.x.to(DEV).float(), ei, ew, node_ptr=big.ptr.to(DEV).to(torch.int32))
    assert adj is None
    assert close(S, torch.cat(Ss), atol=1e-6)
    assert abs(mc.item() - torch.stack(mcs).mean().item()) < 1e-6
    assert abs(o.item() - torch.stack(oos).mean().item()) < 1e-6


def test_hetero_container_batch_through_model():
    """HeteroData -> DataLoader -> batch.to(device) -> HSCN, the protocol of
    train/train.py:73-77."""
    from graph_hscn.config.config import ACT_DICT, HSCNConfig
    from graph_hscn.data import DataLoader
    from graph_hscn.loader.hetero_data import hetero_from_clusters
    from graph_hscn.loader.synthetic import make_dataset
    from graph_hscn.model.hscn import HSCN, build_hscn
    graphs = make_dataset("peptides_func", 8, seed=9)
    rng = np.random.default_rng(0)
    hs = [hetero_from_clusters(g, rng.integers(0, 16, g.num_nodes), 16) for g in graphs]
    model = build_hscn(HSCNConfig(activation="ReLU"), 9, 10).to(DEV)
    assert isinstance(model, HSCN)
    for batch in DataLoader(hs, batch_size=4, shuffle=False):
        batch = batch.to(DEV)
        pred = model(batch.x_dict, batch.edge_index_dict, batch)
        assert pred.shape == (4, 10) and batch["local"].y.shape == (4, 10)
        pred.sum().backward()


@pytest.mark.parametrize("route,K", [("dense", 64), ("auto", 64), ("dense", 16)])
def test_scn_dense_mfma_route_matches_oracle_through_the_model(route, K):
    """BASELINE.json configs[3]: ``SCN(..., mincut_route="dense")`` runs the reference's literal sequence
    to_dense_adj -> dense_mincut_pool (model/hscn.py:61-63) with the contractions on the matrix cores.  Through the
    MODEL, against the oracle (which is that sequence in plain torch): assignments, both losses, the returned dense
    adjacency, every parameter gradient; cluster ids equal on every node whose top-2 margin exceeds 1e-5.  Then a
    batch of equally sized graphs == the mean over single-graph calls."""
    from graph_hscn.loader.synthetic import SHAPES, make_graph
    from graph_hscn.model.hscn import SCN
    from graph_hscn.nn import gcn_norm
    torch.manual_seed(K)
    rng = np.random.default_rng(K)
    graphs = [make_graph(rng, SHAPES["pascalvoc_sp"], n=nn) for nn in (479, 401, 479, 479)]
    F = graphs[0].x.size(1)
    om = OM.SCN([16], "elu", F, K)
    pm = SCN([16], "elu", F, K, mincut_route=route).to(DEV)
    pm.load_state_dict(om.state_dict())
    assert not pm.resident_ok(graphs[0])
    singles = []
    for g in graphs[:2]:
        om.zero_grad(); pm.zero_grad()
        S_o, mc_o, o_o, adj_o, _, _ = OM.scn_step_single_graph(om, g.x, g.edge_index)
        (mc_o + o_o).backward()
        ei, ew = gcn_norm(g.edge_index.to(DEV), None, g.num_nodes, add_self_loops=True)
        S_d, mc_d, o_d, adj_d = pm(g.x.to(DEV).float(), ei, ew)
        assert pm.last_route == "dense"
        (mc_d + o_d).backward()
        assert close(S_d, S_o)
        assert abs(mc_d.item() - mc_o.item()) < ATOL and abs(o_d.item() - o_o.item()) < ATOL
        assert torch.equal(adj_d.cpu(), adj_o)
        # cluster ids: bit-exact on EVERY node of these committed seeds (no margin rule); the oracle's smallest
        # top-2 margin is printed so a flip, should a future change cause one, can be judged
        top = S_o.topk(2, 1).values
        margin = (top[:, 0] - top[:, 1])
        flips = int((S_d.max(1)[1].cpu() != S_o.max(1)[1]).sum())
        print(f"[dense route ids] K={K} n={g.num_nodes}: flips {flips} of {g.num_nodes}, smallest top-2 margin "
              f"{float(margin.min()):.3e}, nodes with margin <= 1e-5: {int((margin <= 1e-5).sum())}")
        assert flips == 0
        for (n_, po), (_, pp) in zip(om.named_parameters(), pm.named_parameters()):
            assert close(pp.grad, po.grad, atol=1e-4, rtol=2e-3), n_
    # equally sized graphs as one [B,n,n] batch through forward_graphs: losses = mean over the graphs
    from graph_hscn.data import Batch
    same = [graphs[0], graphs[2], graphs[3]]
    with torch.no_grad():
        for g in same:
            ei, ew = gcn_norm(g.edge_index.to(DEV), None, g.num_nodes, add_self_loops=True)
            S1, mc1, o1, _ = pm(g.x.to(DEV).float(), ei, ew)
            singles.append((S1, float(mc1), float(o1)))
        Sb, mcb, ob = pm.forward_graphs(Batch.from_data_list(same).to(DEV))
    assert pm.last_route == "dense-ragged" and pm.last_engine == "layered"
    assert close(Sb, torch.cat([s[0] for s in singles]), atol=1e-6)
    assert abs(float(mcb) - np.mean([s[1] for s in singles])) < 1e-6
    assert abs(float(ob) - np.mean([s[2] for s in singles])) < 1e-6


@pytest.mark.parametrize("directed", [False, True])
@pytest.mark.parametrize("adj_format", ["u8", "f32"])
@pytest.mark.parametrize("K,sizes", [(64, (395, 500, 479, 431, 463)), (16, (40, 7, 129)), (64, (2, 64, 65, 128, 500)),
                                     (32, (130, 257, 33))])
def test_scn_dense_route_on_a_ragged_batch_matches_the_oracle_loop(K, sizes, adj_format, directed, monkeypatch):
    """BASELINE.json configs[3] with its REAL size spread (PascalVOC-SP: n in [395, 500]): a batch of graphs of
    different sizes through ``forward_graphs`` on the dense route (adjacency [B, nmax, nmax] zero beyond each graph,
    node-indexed tensors flat, hscn_mincut_dense_ragged_*) against the oracle's reference loop -- one graph at a
    time (model/hscn.py:56-64 after train_clustering.py:37-42), losses meaned: assignments, both losses, ids with zero
    flips, and the gradient of mean(mc + o) for every parameter.  The last case has graphs of 2, 64, 65 nodes (tile
    edges of the 64-row products)."""
    from graph_hscn.data import Batch
    from graph_hscn.loader.synthetic import SHAPES, make_graph
    from graph_hscn.model.hscn import SCN
    torch.manual_seed(K + len(sizes))
    rng = np.random.default_rng(K + len(sizes))
    graphs = [make_graph(rng, SHAPES["pascalvoc_sp"], n=nn) for nn in sizes]
    if directed:
        # every other graph loses a third of its edges (one direction of them): the batch then mixes symmetric
        # adjacencies -- whose A^T S the byte route's backward takes from the forward's A S -- with asymmetric ones
        for gi in range(0, len(graphs), 2):
            ei = graphs[gi].edge_index
            keep = torch.from_numpy(rng.random(ei.size(1)) > 0.33) | (ei[0] > ei[1])
            if int(keep.sum()) < ei.size(1):
                graphs[gi].edge_index = ei[:, keep].contiguous()
    F = graphs[0].x.size(1)
    om = OM.SCN([16], "elu", F, K)
    pm = SCN([16], "elu", F, K, mincut_route="dense").to(DEV)
    pm.load_state_dict(om.state_dict())
    om.zero_grad(); pm.zero_grad()
    Ss, tot = [], 0.0
    mcs, oos = [], []
    for g in graphs:
        S_o, mc_o, o_o, _, _, _ = OM.scn_step_single_graph(om, g.x, g.edge_index)
        Ss.append(S_o.detach()); mcs.append(mc_o.detach()); oos.append(o_o.detach())
        tot = tot + (mc_o + o_o)
    (tot / len(graphs)).backward()
    big = Batch.from_data_list(graphs).to(DEV)
    S_d, mc_d, o_d, total = pm.forward_graphs(big, with_total=True)
    assert pm.last_route == "dense-ragged"
    total.backward()
    S_ref = torch.cat(Ss)
    assert close(S_d, S_ref)
    assert abs(mc_d.item() - torch.stack(mcs).mean().item()) < ATOL
    assert abs(o_d.item() - torch.stack(oos).mean().item()) < ATOL
    flips = int((S_d.max(1)[1].cpu() != S_ref.max(1)[1]).sum())
    top = S_ref.topk(2, 1).values if K > 1 else None
    print(f"[ragged dense] K={K} sizes={sizes}: flips {flips} of {S_ref.size(0)}, smallest top-2 margin "
          f"{float((top[:, 0] - top[:, 1]).min()):.3e}")
    assert flips == 0
    for (n_, po), (_, pp) in zip(om.named_parameters(), pm.named_parameters()):
        assert close(pp.grad, po.grad, atol=1e-4, rtol=2e-3), n_


def test_static_gcn_norm_equals_gcn_norm_in_every_aggregate():
    """``gcn_norm_static`` (static output shape, capturable) against ``gcn_norm`` (PyG's list layout) on a multigraph
    with self loops, repeated edges and isolated nodes: same normalised weight on every surviving edge, zero on the
    placeholders, and -- what the message passing sees -- identical weighted aggregates and degrees, bit for bit."""
    from graph_hscn.nn.pool import gcn_norm, gcn_norm_static
    g = torch.Generator().manual_seed(5)
    N, E = 37, 160
    ei = torch.randint(0, N - 3, (2, E), generator=g)           # nodes N-3.. isolated
    ei[:, 5] = torch.tensor([4, 4]); ei[:, 17] = torch.tensor([9, 9]); ei[:, 60] = torch.tensor([4, 4])   # self loops (one node twice)
    w = torch.rand(E, generator=g) + 0.5
    w[60] = w[5]                                                 # (two loops on node 4: give them one weight, the winner is unspecified)
    x = torch.randn(N, 8, generator=g)
    for ew in (None, w):
        e1, w1 = gcn_norm(ei.to(DEV), None if ew is None else ew.to(DEV), N, add_self_loops=True)
        e2, w2 = gcn_norm_static(ei.to(DEV), None if ew is None else ew.to(DEV), N)
        assert e2.shape == (2, E + N) and w2.shape == (E + N,)
        loops = ei[0] == ei[1]
        assert torch.equal(e2[:, :E].cpu(), ei) and float(w2[:E][loops.to(DEV)].abs().max()) == 0.0
        assert torch.equal(w2[:E][~loops.to(DEV)], w1[: int((~loops).sum())])      # surviving edges keep their order
        assert torch.equal(w2[E:], w1[int((~loops).sum()):])                        # the N loops
        agg1 = torch.zeros(N, 8, device=DEV).index_add_(0, e1[1], w1[:, None] * x.to(DEV)[e1[0]])
        agg2 = torch.zeros(N, 8, device=DEV).index_add_(0, e2[1], w2[:, None] * x.to(DEV)[e2[0]])
        assert torch.allclose(agg1, agg2, atol=1e-6)
