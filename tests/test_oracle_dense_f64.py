"""An INDEPENDENT second formulation of the oracle's message-passing operators: dense matrices in float64.

The reference holds no vectors for this path, so the oracle (oracle/pyg_ops.py: PyG's
gather -> scale -> index_add_ restatement) is pinned by closed-form KATs and, here, by a formulation that shares
no code and no data structure with it: edge lists become dense multiplicity matrices, aggregation becomes a
matrix product, the segment softmax becomes a masked row softmax weighted by edge multiplicity.  Random
MULTIGRAPHS (repeated edges, self loops, isolated nodes) are the inputs: those are the cases in which a
misreading of PyG's edge-wise semantics (each repeated edge is its own message / its own softmax term) would
show.  A misreading shared by the oracle and the HIP kernels (which are tested against the oracle) would have
to be shared by this file too.
"""
import numpy as np
import pytest
import torch

from oracle import pyg_ops as P

F64 = torch.float64


def _multigraph(n_src, n_dst, e, seed, self_loops=True):
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(0, n_src, (e,), generator=g)
    dst = torch.randint(0, n_dst, (e,), generator=g)
    # repeat a third of the edges (multi-edges), in shuffled order
    rep = torch.randint(0, e, (e // 3,), generator=g)
    src, dst = torch.cat([src, src[rep]]), torch.cat([dst, dst[rep]])
    perm = torch.randperm(src.numel(), generator=g)
    src, dst = src[perm], dst[perm]
    if not self_loops and n_src == n_dst:
        keep = src != dst
        src, dst = src[keep], dst[keep]
    return torch.stack([src, dst])


def _mult(ei, n_src, n_dst):
    """M[i, j] = number of edges j -> i (target-major), float64."""
    M = torch.zeros(n_dst, n_src, dtype=F64)
    for j, i in ei.t().tolist():
        M[i, j] += 1.0
    return M


@pytest.mark.parametrize("n,e,seed", [(17, 60, 0), (40, 90, 1), (9, 200, 2), (33, 10, 3)])
def test_gcn_conv_equals_dense_normalised_adjacency(n, e, seed):
    """GCNConv(add_self_loops=False) == D^-1/2 M D^-1/2 (X W^T) + b, D = in-degree counting multiplicity,
    rows of degree 0 get the bias only (A.5)."""
    ei = _multigraph(n, n, e, seed)
    torch.manual_seed(seed)
    conv = P.GCNConv(7, 5, add_self_loops=False)
    with torch.no_grad():
        conv.bias.normal_()
    x = torch.randn(n, 7)
    got = conv(x, ei)
    M = _mult(ei, n, n)
    deg = M.sum(1)
    dis = torch.where(deg > 0, deg.pow(-0.5), torch.zeros_like(deg))
    A_hat = dis[:, None] * M * dis[None, :]
    want = A_hat @ (x.to(F64) @ conv.lin.weight.to(F64).t()) + conv.bias.to(F64)
    if e < n:
        assert (deg == 0).any()             # the sparse case has isolated targets (bias-only rows)
    np.testing.assert_allclose(got.detach().double().numpy(), want.detach().numpy(), atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("n,e,seed", [(12, 40, 0), (30, 50, 4)])
def test_gcn_norm_with_self_loops_equals_dense(n, e, seed):
    """gcn_norm(add_self_loops=True) on a multigraph with loops: off-diagonal multiplicities stay, every node ends
    with exactly ONE unit loop (A.1: add_remaining_self_loops), deg = row sums over targets."""
    ei = _multigraph(n, n, e, seed, self_loops=True)
    ei2, w = P.gcn_norm(ei, None, n, add_self_loops=True)
    M = _mult(ei, n, n)
    M.fill_diagonal_(0.0)
    M = M + torch.eye(n, dtype=F64)
    deg = M.sum(1)
    dis = deg.pow(-0.5)
    A_hat = dis[:, None] * M * dis[None, :]
    got = torch.zeros(n, n, dtype=F64)
    for (j, i), wv in zip(ei2.t().tolist(), w.tolist()):
        got[i, j] += wv
    np.testing.assert_allclose(got.numpy(), A_hat.numpy(), atol=1e-6)
    # and the GraphConv that consumes these weights == A_hat X W_rel^T + b + X W_root^T
    torch.manual_seed(seed)
    gc = P.GraphConv(6, 4)
    x = torch.randn(n, 6)
    out = gc(x, ei2, w)
    want = (A_hat @ x.to(F64)) @ gc.lin_rel.weight.to(F64).t() + gc.lin_rel.bias.to(F64) + x.to(F64) @ gc.lin_root.weight.to(F64).t()
    np.testing.assert_allclose(out.detach().double().numpy(), want.detach().numpy(), atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("ns,nd,e,seed", [(25, 6, 40, 0), (60, 9, 60, 1), (10, 4, 100, 2), (14, 7, 5, 3)])
def test_bipartite_gat_equals_masked_dense_softmax(ns, nd, e, seed):
    """GATConv((Fs, Fd), H, heads=1, add_self_loops=False): per target i a softmax over its in-EDGES (a repeated
    edge is a repeated term) of leaky_relu(a_s[j] + a_d[i]); targets without in-edges get the bias only (A.6)."""
    ei = _multigraph(ns, nd, e, seed)
    torch.manual_seed(seed)
    conv = P.GATConv((5, 3), 8)
    with torch.no_grad():
        conv.bias.normal_()
    xs, xd = torch.randn(ns, 5), torch.randn(nd, 3)
    got = conv((xs, xd), ei)
    hs = xs.to(F64) @ conv.lin_src.weight.to(F64).t()
    hd = xd.to(F64) @ conv.lin_dst.weight.to(F64).t()
    a_s = hs @ conv.att_src.to(F64).view(-1)
    a_d = hd @ conv.att_dst.to(F64).view(-1)
    logits = torch.nn.functional.leaky_relu(a_d[:, None] + a_s[None, :], 0.2)      # [nd, ns]
    M = _mult(ei, ns, nd)
    masked = torch.where(M > 0, logits, torch.full_like(logits, -float("inf")))
    mx = masked.max(1, keepdim=True).values
    mx = torch.where(torch.isinf(mx), torch.zeros_like(mx), mx)
    w = M * torch.exp(torch.where(M > 0, logits - mx, torch.zeros_like(logits)))
    w = torch.where(M > 0, w, torch.zeros_like(w))
    alpha = w / (w.sum(1, keepdim=True) + 1e-16)
    want = alpha @ hs + conv.bias.to(F64)
    if e < nd:
        assert (M.sum(1) == 0).any()        # targets without in-edges: bias only
    np.testing.assert_allclose(got.detach().double().numpy(), want.detach().numpy(), atol=1e-5, rtol=1e-5)


def test_hetero_layer_sums_relations_per_target_type():
    """HeteroConv(aggr='sum') of the HSCN layer: local <- ll only; virtual <- vv + lv (A.8), dense f64."""
    nl, nv = 21, 5
    e_ll, e_vv, e_lv = _multigraph(nl, nl, 50, 0), _multigraph(nv, nv, 12, 1), _multigraph(nl, nv, 30, 2)
    torch.manual_seed(0)
    ll, vv, lv = P.GCNConv(4, 6, add_self_loops=False), P.GCNConv(4, 6, add_self_loops=False), P.GATConv((4, 4), 6)
    layer = P.HeteroConv({("local", "to", "virtual"): lv, ("local", "to", "local"): ll,
                          ("virtual", "to", "virtual"): vv})
    xl, xv = torch.randn(nl, 4), torch.randn(nv, 4)
    out = layer({"local": xl, "virtual": xv},
                {("local", "to", "local"): e_ll, ("virtual", "to", "virtual"): e_vv, ("local", "to", "virtual"): e_lv})

    def gcn(conv, x, ei, n):
        M = _mult(ei, n, n)
        deg = M.sum(1)
        dis = torch.where(deg > 0, deg.pow(-0.5), torch.zeros_like(deg))
        return (dis[:, None] * M * dis[None, :]) @ (x.to(F64) @ conv.lin.weight.to(F64).t()) + conv.bias.to(F64)

    want_l = gcn(ll, xl, e_ll, nl)
    np.testing.assert_allclose(out["local"].detach().double().numpy(), want_l.detach().numpy(), atol=1e-5, rtol=1e-5)
    gat = lv((xl, xv), e_lv).detach().double()
    want_v = gcn(vv, xv, e_vv, nv) + gat
    np.testing.assert_allclose(out["virtual"].detach().double().numpy(), want_v.detach().numpy(), atol=1e-5, rtol=1e-5)


def test_global_mean_pool_equals_dense_membership_matrix():
    sizes = [3, 1, 7, 2]
    batch = torch.repeat_interleave(torch.arange(4), torch.tensor(sizes))
    x = torch.randn(sum(sizes), 5)
    Mb = torch.zeros(4, sum(sizes), dtype=F64)
    Mb[batch, torch.arange(sum(sizes))] = 1.0
    want = (Mb @ x.to(F64)) / Mb.sum(1, keepdim=True)
    np.testing.assert_allclose(P.global_mean_pool(x, batch, 4).double().numpy(), want.numpy(), atol=1e-6)


def test_oracle_runs_in_float64():
    """The f64 evaluation the GPU accuracy test leans on (tests/test_gpu_resident.py): the oracle model in
    double precision agrees with its float32 self to float32 rounding."""
    from oracle import hetero_data as OH
    from oracle import models as OM
    from graph_hscn.loader.synthetic import make_dataset
    graphs = make_dataset("peptides_func", 3, seed=0)
    rng = np.random.default_rng(0)
    b = OH.collate_hetero([OH.hetero_from_clusters(g.x, g.edge_index, g.y, rng.integers(0, 8, g.num_nodes), 8) for g in graphs])
    torch.manual_seed(0)
    m = OM.HSCN("GAT", "GCN", "GCN", OM.ACT["relu"], 9, 16, 10, 3)
    o32 = m(b["x_dict"], b["edge_index_dict"], b["batch_local"], 3)
    import copy
    m64 = copy.deepcopy(m).double()
    o64 = m64({k: v.double() for k, v in b["x_dict"].items()}, b["edge_index_dict"], b["batch_local"], 3)
    assert o64.dtype == F64
    np.testing.assert_allclose(o32.detach().double().numpy(), o64.detach().numpy(), atol=1e-5, rtol=1e-5)
