"""HIP operators (through the C ABI) vs the CPU oracle, same seeded inputs.
Float tolerance 1e-5 (BASELINE.json north_star); index work bit-exact."""
import numpy as np
import pytest
import torch

from oracle import pyg_ops as P
from tests.helpers import ATOL, DEV, close, rand_graph

pytestmark = pytest.mark.gpu


def _rel(ei, ns, nd, both=None):
    from graph_hscn.structure import Relation
    return Relation(ei.to(DEV), ns, nd, both=both)


@pytest.mark.parametrize("both", [True, False])      # True: hscn_csr_build_pair; False: two hscn_csr_build calls
@pytest.mark.parametrize("n,e,seed", [(1, 0, 0), (7, 0, 1), (50, 200, 2), (3000, 20000, 3), (70000, 300000, 4)])
def test_csr_build_is_stable_and_exact(n, e, seed, both):
    ei = rand_graph(n, e, seed, self_loops=True) if e else torch.zeros(2, 0, dtype=torch.long)
    rel = _rel(ei, n, n, both)
    rel.check()
    for csr, key, other in ((rel.csr, ei[1], ei[0]), (rel.csr_t, ei[0], ei[1])):
        order = torch.argsort(key, stable=True)
        rowptr = torch.zeros(n + 1, dtype=torch.long)
        rowptr[1:] = torch.cumsum(torch.bincount(key, minlength=n), 0)
        E = ei.size(1)
        assert torch.equal(csr.rowptr.cpu().long(), rowptr)
        assert torch.equal(csr.eid.cpu().long()[:E], order)
        assert torch.equal(csr.col.cpu().long()[:E], other[order])
    if ei.size(1):
        inv = torch.empty(ei.size(1), dtype=torch.long)
        inv[rel.csr.eid.cpu().long()] = torch.arange(ei.size(1))
        assert torch.equal(rel.pos_t.cpu().long(), inv[rel.csr_t.eid.cpu().long()])


@pytest.mark.parametrize("both", [True, False])
def test_csr_build_flags_out_of_range(both):
    ei = torch.tensor([[0, 1, 9], [1, 0, 2]])
    rel = _rel(ei, 3, 3, both)
    with pytest.raises(IndexError):
        rel.check()
    assert rel.csr.rowptr.cpu().tolist() == [0, 1, 2, 2]
    assert rel.csr_t.rowptr.cpu().tolist() == [0, 1, 2, 2]


def test_csr_pair_build_of_a_bipartite_relation_equals_the_two_single_builds():
    """num_src != num_dst, both sides beyond the single-block scan (the two halves of the pair's scan grid differ in
    length): every array of hscn_csr_build_pair equals hscn_csr_build(dst, src) / hscn_csr_build(src, dst)."""
    g = torch.Generator().manual_seed(5)
    ns, nd, E = 40000, 9000, 150000
    ei = torch.stack([torch.randint(0, ns, (E,), generator=g), torch.randint(0, nd, (E,), generator=g)])
    a, b = _rel(ei, ns, nd, True), _rel(ei, ns, nd, False)
    for x, y in ((a.csr, b.csr), (a.csr_t, b.csr_t)):
        assert torch.equal(x.rowptr, y.rowptr) and torch.equal(x.col, y.col) and torch.equal(x.eid, y.eid)


@pytest.mark.parametrize("rows,i,o,act", [(1, 9, 16, "identity"), (333, 16, 16, "relu"), (1000, 9, 10, "elu"),
                                           (257, 32, 12, "tanh"), (64, 128, 128, "relu"), (5, 16, 1, "identity")])
def test_linear_fwd_bwd(rows, i, o, act):
    from graph_hscn.nn import functional as Fh
    from oracle.models import ACT
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, i, generator=g)
    W = torch.randn(o, i, generator=g) / i ** 0.5
    b = torch.randn(o, generator=g)
    xo, Wo, bo = (t.clone().requires_grad_() for t in (x, W, b))
    yo = ACT[act](torch.nn.functional.linear(xo, Wo, bo))
    gy = torch.randn(rows, o, generator=g)
    yo.backward(gy)
    xd, Wd, bd = (t.to(DEV).requires_grad_() for t in (x, W, b))
    yd = Fh.linear(xd, Wd, bd, act)
    yd.backward(gy.to(DEV))
    assert close(yd, yo)
    assert close(xd.grad, xo.grad, atol=1e-4, rtol=1e-4)
    assert close(Wd.grad, Wo.grad, atol=1e-4 * max(1, rows / 100), rtol=1e-4)
    assert close(bd.grad, bo.grad, atol=1e-4 * max(1, rows / 100), rtol=1e-4)


@pytest.mark.parametrize("n,e,fin,h,seed", [(40, 100, 9, 16, 0), (500, 1200, 16, 16, 1), (300, 900, 16, 32, 2),
                                           (200, 300, 9, 10, 3), (1000, 5000, 64, 128, 4)])
def test_gcn_conv(n, e, fin, h, seed):
    from graph_hscn.nn import GCNConv
    ei = rand_graph(n, e, seed)
    oc = P.GCNConv(fin, h, add_self_loops=False)
    with torch.no_grad():
        oc.bias.normal_()
    pc = GCNConv(fin, h, add_self_loops=False).to(DEV)
    pc.load_state_dict(oc.state_dict())
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, fin, generator=g)
    xo = x.clone().requires_grad_()
    xd = x.to(DEV).requires_grad_()
    yo = oc(xo, ei)
    yd = pc(xd, ei.to(DEV))
    gy = torch.randn(n, h, generator=g)
    yo.backward(gy)
    yd.backward(gy.to(DEV))
    assert close(yd, yo)
    assert close(xd.grad, xo.grad, atol=1e-4, rtol=1e-4)
    assert close(pc.lin.weight.grad, oc.lin.weight.grad, atol=1e-3, rtol=1e-4)
    assert close(pc.bias.grad, oc.bias.grad, atol=1e-3, rtol=1e-4)


def test_gcn_aggregation_is_bitwise_the_cpu_order():
    """Same h on both sides: the CSR walk reproduces index_add_'s edge order and
    separately rounded mul/add, so the propagate step is bit-identical."""
    from graph_hscn.nn import functional as Fh
    n, h = 400, 16
    ei = rand_graph(n, 3000, 11)
    rel = _rel(ei, n, n)
    hx = torch.randn(n, h, generator=torch.Generator().manual_seed(0))
    _, w = P.gcn_norm(ei, None, n, add_self_loops=False)
    want = P.scatter_add(w.view(-1, 1) * hx[ei[0]], ei[1], n)
    got = Fh.spmm_gcn_raw(rel.csr, rel.dinv, rel.dinv, hx.to(DEV))
    assert torch.equal(got.cpu(), want)


@pytest.mark.parametrize("act", ["identity", "elu", "tanh", "relu"])
def test_graph_conv_weighted(act):
    from graph_hscn.nn import GraphConv
    from oracle.models import ACT
    n, fin, h = 150, 9, 16
    ei = rand_graph(n, 320, 5, symmetric=True)
    ei2, w = P.gcn_norm(ei, None, n, add_self_loops=True)
    oc = P.GraphConv(fin, h)
    pc = GraphConv(fin, h).to(DEV)
    pc.load_state_dict(oc.state_dict())
    g = torch.Generator().manual_seed(0)
    x = torch.randint(0, 12, (n, fin), generator=g).float()
    xo, xd = x.clone().requires_grad_(), x.to(DEV).requires_grad_()
    yo = ACT[act](oc(xo, ei2, w))
    yd = pc(xd, ei2.to(DEV), w.to(DEV), act=act)
    gy = torch.randn(n, h, generator=g)
    yo.backward(gy)
    yd.backward(gy.to(DEV))
    assert close(yd, yo, atol=2e-5)
    assert close(xd.grad, xo.grad, atol=1e-4, rtol=1e-4)
    for k in ("lin_rel.weight", "lin_rel.bias", "lin_root.weight"):
        assert close(dict(pc.named_parameters())[k].grad, dict(oc.named_parameters())[k].grad, atol=2e-3, rtol=1e-4), k


@pytest.mark.parametrize("ns,nd,e,fin,h,seed", [(30, 4, 30, 9, 16, 0), (400, 37, 400, 16, 16, 1), (400, 37, 1500, 16, 32, 2),
                                                (100, 10, 100, 9, 10, 3), (2000, 3, 2000, 16, 64, 4)])
def test_gat_conv_bipartite(ns, nd, e, fin, h, seed):
    from graph_hscn.nn import GATConv
    g = torch.Generator().manual_seed(seed)
    if e == ns:   # HSCN pattern: every source has exactly one out-edge
        ei = torch.stack([torch.arange(ns), torch.randint(0, nd, (ns,), generator=g)])
    else:
        ei = torch.stack([torch.randint(0, ns, (e,), generator=g), torch.randint(0, nd, (e,), generator=g)])
    oc = P.GATConv((fin, fin), h)
    with torch.no_grad():
        oc.bias.normal_()
    pc = GATConv((fin, fin), h, add_self_loops=False).to(DEV)
    pc.load_state_dict(oc.state_dict())
    xs, xd_ = torch.randn(ns, fin, generator=g), torch.randn(nd, fin, generator=g)
    xso, xdo = xs.clone().requires_grad_(), xd_.clone().requires_grad_()
    xsd, xdd = xs.to(DEV).requires_grad_(), xd_.to(DEV).requires_grad_()
    yo = oc((xso, xdo), ei)
    yd = pc((xsd, xdd), ei.to(DEV))
    gy = torch.randn(nd, h, generator=g)
    yo.backward(gy)
    yd.backward(gy.to(DEV))
    assert close(yd, yo, atol=2e-5, rtol=1e-5)
    assert close(xsd.grad, xso.grad, atol=1e-4, rtol=1e-3)
    assert close(xdd.grad, xdo.grad, atol=1e-4, rtol=1e-3)
    po, pp = dict(oc.named_parameters()), dict(pc.named_parameters())
    for k in po:
        assert close(pp[k].grad, po[k].grad, atol=2e-3, rtol=1e-3), k


@pytest.mark.parametrize("h", [16, 10, 128])
def test_global_mean_pool(h):
    from graph_hscn.nn import global_mean_pool
    g = torch.Generator().manual_seed(h)
    sizes = [1, 7, 150, 444, 8, 3]
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    x = torch.randn(int(batch.numel()), h, generator=g)
    xo, xd = x.clone().requires_grad_(), x.to(DEV).requires_grad_()
    yo = P.global_mean_pool(xo, batch)
    yd = global_mean_pool(xd, batch.to(DEV))
    gy = torch.randn(len(sizes), h, generator=g)
    yo.backward(gy)
    yd.backward(gy.to(DEV))
    assert close(yd, yo) and close(xd.grad, xo.grad)


def test_global_mean_pool_unsorted_batch_and_empty_segment():
    from graph_hscn.nn import global_mean_pool
    batch = torch.tensor([2, 0, 2, 0, 0, 3])
    x = torch.arange(12.0).view(6, 2)
    yo = P.global_mean_pool(x, batch, 4)
    yd = global_mean_pool(x.to(DEV), batch.to(DEV), 4)
    assert close(yd, yo) and float(yd[1].abs().sum()) == 0.0


def test_gcn_norm_matches_oracle_layout_and_weights():
    from graph_hscn.nn import gcn_norm
    ei = torch.tensor([[0, 1, 1, 2, 2, 3], [1, 0, 2, 1, 2, 0]])
    w = torch.tensor([1.0, 1.0, 2.0, 2.0, 5.0, 0.5])
    for ww in (None, w):
        eo, wo = P.gcn_norm(ei, ww, 5, add_self_loops=True)
        ed, wd = gcn_norm(ei.to(DEV), None if ww is None else ww.to(DEV), 5, add_self_loops=True)
        assert torch.equal(ed.cpu(), eo)
        assert torch.equal(wd.cpu(), wo)      # same summation order -> bit-exact


def test_to_dense_adj():
    from graph_hscn.nn import to_dense_adj
    ei = rand_graph(37, 200, 3, self_loops=True)
    assert torch.equal(to_dense_adj(ei.to(DEV)).cpu(), P.to_dense_adj(ei))


@pytest.mark.parametrize("K", [2, 4, 16, 32, 64])
def test_argmax_first_max_wins(K):
    from graph_hscn import _hip
    g = torch.Generator().manual_seed(K)
    S = torch.softmax(torch.randn(1000, K, generator=g) * 30, -1)   # saturating -> exact ties occur
    S[::7] = S[::7].round()
    ids = torch.empty(1000, dtype=torch.int64, device=DEV)
    Sd = S.to(DEV)
    _hip.call("hscn_assign_argmax", _hip.ptr(Sd), _hip.ptr(ids), 1000, K, _hip.stream())
    assert np.array_equal(ids.cpu().numpy(), S.max(1)[1].numpy())


@pytest.mark.parametrize("K,sizes", [(4, [12]), (16, [151]), (16, [8, 151, 444, 30]), (64, [479, 400]), (32, [60, 61])])
def test_mincut_sparse_vs_dense_oracle(K, sizes):
    from graph_hscn.nn import mincut_pool_sparse
    g = torch.Generator().manual_seed(K + len(sizes))
    eis, off = [], 0
    for n in sizes:
        ei = rand_graph(n, 2 * n, off + 1, symmetric=True)
        ei, _ = P.add_remaining_self_loops(ei, None, 1.0, n)
        eis.append(ei + off)
        off += n
    N = off
    ei = torch.cat(eis, 1)
    s = torch.randn(N, K, generator=g)
    x = torch.randn(N, 16, generator=g)
    so = s.clone().requires_grad_()
    # oracle: per-graph dense_mincut_pool, losses averaged over graphs (its own batch mean)
    mcs, oos, outs, oadjs, Ss = [], [], [], [], []
    o = 0
    for n, e in zip(sizes, eis):
        adj = P.to_dense_adj(e - o, n)
        out, oadj, mc, oo = P.dense_mincut_pool(x[o:o + n], adj, so[o:o + n])
        mcs.append(mc); oos.append(oo); outs.append(out[0]); oadjs.append(oadj[0])
        Ss.append(torch.softmax(so[o:o + n], -1))
        o += n
    mc_o, oo_o = torch.stack(mcs).mean(), torch.stack(oos).mean()
    (mc_o * 1.3 + oo_o * 0.7).backward()
    sd = s.to(DEV).requires_grad_()
    node_ptr = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int32, device=DEV)
    S, px, padj, mc_d, oo_d = mincut_pool_sparse(x.to(DEV), ei.to(DEV), sd, node_ptr)
    (mc_d * 1.3 + oo_d * 0.7).backward()
    assert close(S, torch.cat(Ss))
    assert abs(float(mc_d) - float(mc_o)) < ATOL and abs(float(oo_d) - float(oo_o)) < ATOL
    assert close(px, torch.stack(outs), atol=1e-4, rtol=1e-5)
    assert close(padj, torch.stack(oadjs), atol=1e-5, rtol=1e-4)
    assert close(sd.grad, so.grad, atol=1e-6, rtol=1e-3)


def test_ops_refuse_cpu_tensors():
    from graph_hscn.nn import functional as Fh
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Fh.linear(torch.randn(4, 4), torch.randn(4, 4))


@pytest.mark.parametrize("loss_fn,C", [("cross_entropy", 10), ("l1", 11)])
def test_fused_criterion_matches_reference_loss(loss_fn, C):
    from graph_hscn.loss import criterion
    from oracle.models import criterion as ocrit
    g = torch.Generator().manual_seed(C)
    pred = torch.randn(128, C, generator=g) * 3
    true = (torch.rand(128, C, generator=g) < 0.2).float() if loss_fn == "cross_entropy" else torch.randn(128, C, generator=g)
    po, pd = pred.clone().requires_grad_(), pred.to(DEV).requires_grad_()
    lo, so = ocrit(loss_fn, po, true)
    ld, sd = criterion(loss_fn, pd, true.to(DEV))
    (lo * 1.7).backward()
    (ld * 1.7).backward()
    assert abs(lo.item() - ld.item()) < 1e-6 and close(sd, so, atol=1e-6)
    assert close(pd.grad, po.grad, atol=1e-8, rtol=1e-5)


@pytest.mark.parametrize("B,n,K,F", [(1, 12, 4, 16), (3, 151, 16, 16), (2, 479, 64, 14), (4, 64, 32, 9), (2, 100, 5, 3)])
def test_dense_mincut_pool_mfma_matches_oracle(B, n, K, F):
    from graph_hscn.nn import dense_mincut_pool
    g = torch.Generator().manual_seed(B * 1000 + n)
    adj = (torch.rand(B, n, n, generator=g) < 4.0 / n).float()
    adj = ((adj + adj.transpose(1, 2)) > 0).float() + torch.eye(n)        # symmetric binary A + I
    adj[0, 0, 1] += 1.0                                                    # and one asymmetric multi-edge
    s = torch.randn(B, n, K, generator=g)
    x = torch.randn(B, n, F, generator=g)
    so = s.clone().requires_grad_()
    out_o, oadj_o, mc_o, oo_o = P.dense_mincut_pool(x, adj, so)
    (mc_o * 1.3 + oo_o * 0.7).backward()
    sd = s.to(DEV).requires_grad_()
    out_d, oadj_d, mc_d, oo_d = dense_mincut_pool(x.to(DEV), adj.to(DEV), sd)
    (mc_d * 1.3 + oo_d * 0.7).backward()
    assert abs(mc_d.item() - mc_o.item()) < ATOL and abs(oo_d.item() - oo_o.item()) < ATOL
    assert close(out_d, out_o, atol=1e-4, rtol=1e-5)
    assert close(oadj_d, oadj_o, atol=1e-5, rtol=1e-4)
    assert close(sd.grad, so.grad, atol=1e-6, rtol=1e-3)


def test_bgemm_mfma_asymmetric_operands():
    """A = I-like and an asymmetric B catch a transposed C write (fragment-layout check)."""
    from graph_hscn import _hip
    for (M, N, Kd, ta) in [(64, 16, 64, 0), (70, 33, 45, 0), (37, 64, 129, 1), (16, 5, 7, 1)]:
        g = torch.Generator().manual_seed(M + N)
        A = torch.randn(2, Kd, M, generator=g) if ta else torch.randn(2, M, Kd, generator=g)
        Bm = torch.randn(2, Kd, N, generator=g)
        Bm += torch.arange(N).float() * 0.5 + torch.arange(Kd).float().view(-1, 1)      # asymmetric
        want = (A.transpose(1, 2) if ta else A) @ Bm
        Ad, Bd = A.to(DEV), Bm.to(DEV)
        C = torch.empty(2, M, N, device=DEV)
        _hip.call("hscn_bgemm_f32", _hip.ptr(Ad), _hip.ptr(Bd), _hip.ptr(C), 2, M, N, Kd, Ad.stride(1), N, N,
                  Ad.stride(0), Kd * N, M * N, ta, _hip.stream())
        assert close(C, want, atol=2e-4, rtol=1e-5), (M, N, Kd, ta)


def test_dense_adj_s_entry_and_symmetry_flags():
    """hscn_dense_adj_s (the A S / A^T S launch of the dense route by itself) against torch.matmul on a ragged byte
    adjacency, and hscn_dense_adj_asymmetry_u8: 0 for undirected graphs, 1 for a graph with a single one-way edge --
    also when that edge sits across a 64 x 64 tile boundary or in the last row."""
    from graph_hscn import _hip
    from graph_hscn.nn.pool import to_dense_adj_ragged
    g = torch.Generator().manual_seed(9)
    sizes = [70, 129, 5, 200]
    K = 16
    eis, off = [], 0
    one_way = {1: (3, 100), 3: (199, 64)}        # graph -> (src, dst) of an extra directed edge
    for b, n in enumerate(sizes):
        e = torch.randint(0, n, (2, 3 * n), generator=g)
        e = torch.cat([e, e.flip(0)], 1)          # undirected
        if b in one_way:
            e = torch.cat([e, torch.tensor([[one_way[b][0]], [one_way[b][1]]])], 1)
        eis.append(e + off)
        off += n
    ei = torch.cat(eis, 1).to(DEV)
    N, B, nmax = off, len(sizes), max(sizes)
    nptr = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int32, device=DEV)
    gid = torch.repeat_interleave(torch.arange(B, dtype=torch.int32), torch.tensor(sizes)).to(DEV)
    adj8, asym = to_dense_adj_ragged(ei, nptr, gid, B, nmax, as_bytes=True, symmetry=True)
    assert asym.cpu().tolist() == [0, 1, 0, 1]
    S = torch.rand(N, K, generator=g).to(DEV)
    A = adj8[:, :, :nmax].float()
    for transA in (0, 1):
        out = torch.empty(N, K, device=DEV)
        deg = torch.empty(N, device=DEV)
        _hip.call("hscn_dense_adj_s", _hip.ptr(adj8), 1, _hip.ptr(S), _hip.ptr(nptr), B, nmax, K, transA, _hip.ptr(out),
                  _hip.ptr(deg) if not transA else None, _hip.stream())
        o = 0
        for b, n in enumerate(sizes):
            Ab = A[b, :n, :n].t() if transA else A[b, :n, :n]
            want = Ab.double() @ S[o:o + n].double()
            assert torch.allclose(out[o:o + n].double(), want, atol=1e-4, rtol=1e-6), (b, transA)
            if not transA:
                assert torch.equal(deg[o:o + n], A[b, :n, :n].sum(1))
            o += n
