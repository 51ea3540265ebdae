"""Batches collated on the device out of an HBM-resident hetero dataset (hscn_collate_gather) against the host
collate ``HeteroBatch.from_data_list`` the reference's loader performs (loader/hetero_data.py:91-106)."""
import numpy as np
import pytest
import torch

from tests.helpers import DEV

pytestmark = pytest.mark.gpu

LL, VV, LV = ("local", "to", "local"), ("virtual", "to", "virtual"), ("local", "to", "virtual")


def _dataset(G, K, seed, name="peptides_func"):
    from graph_hscn.loader.hetero_data import hetero_from_clusters
    from graph_hscn.loader.synthetic import make_dataset
    graphs = make_dataset(name, G, seed=seed)
    rng = np.random.default_rng(seed)
    return [hetero_from_clusters(g, rng.integers(0, K, g.num_nodes), K) for g in graphs]


def _same(static_hb, host):
    """Every field of the host batch equals the valid prefix of the static buffers, bit for bit."""
    for nt in ("local", "virtual"):
        n = host[nt].num_nodes
        assert torch.equal(static_hb[nt].x[:n].cpu(), host[nt].x.float())
        assert torch.equal(static_hb[nt].batch[:n].cpu(), host[nt].batch)
        assert torch.equal(static_hb[nt].ptr.cpu(), host[nt].ptr) and torch.equal(static_hb[nt].ptr32.cpu(), host[nt].ptr32)
    assert torch.equal(static_hb["local"].y.cpu(), host["local"].y.float())
    for et in (LL, VV, LV):
        e = host[et].edge_index.size(1)
        assert torch.equal(static_hb[et].edge_index[:, :e].cpu(), host[et].edge_index)
        assert torch.equal(static_hb[et].ptr32.cpu(), host[et].ptr32)


@pytest.mark.parametrize("G,B,K,name", [(40, 8, 8, "peptides_func"), (9, 9, 1, "pcqm_contact"), (12, 1, 64, "pascalvoc_sp"),
                                        (300, 128, 16, "peptides_func")])
def test_gather_equals_host_collate(G, B, K, name):
    from graph_hscn.data import HeteroBatch
    from graph_hscn.loader.device_dataset import DeviceHeteroDataset
    hs = _dataset(G, K, seed=G, name=name)
    ds = DeviceHeteroDataset(hs, DEV, B)
    rng = np.random.default_rng(0)
    for trial in range(4):
        ids = rng.permutation(G)[:B] if trial else np.argsort([-h["local"].num_nodes for h in hs])[:B]   # the B largest first
        hb = ds.gather(torch.as_tensor(ids, dtype=torch.int64, device=DEV))
        ds.check()
        _same(hb, HeteroBatch.from_data_list([hs[i] for i in ids]))
    ids = rng.integers(0, G, B)                                    # repeated graphs are fine
    _same(ds.gather(torch.as_tensor(ids, dtype=torch.int64, device=DEV)), HeteroBatch.from_data_list([hs[i] for i in ids]))


def test_bad_ids_raise_the_flag_and_shapes_are_checked():
    from graph_hscn.loader.device_dataset import DeviceHeteroDataset
    ds = DeviceHeteroDataset(_dataset(6, 4, seed=1), DEV, 3)
    ds.gather(torch.tensor([0, 6, 1], dtype=torch.int64, device=DEV))
    with pytest.raises(IndexError):
        ds.check()
    with pytest.raises(ValueError):
        ds.gather(torch.tensor([0, 1], dtype=torch.int64, device=DEV))
    with pytest.raises(ValueError):
        ds.gather(torch.tensor([0, 1, 2], dtype=torch.int32, device=DEV))


def test_epoch_of_shuffled_batches_through_one_captured_step():
    """perm on the device -> gather -> replay: prediction, loss and gradients of every step equal the eager
    step on the host-collated batch of the same graphs, bit for bit."""
    from graph_hscn.config.config import ACT_DICT
    from graph_hscn.data import HeteroBatch
    from graph_hscn.loader.device_dataset import DeviceHeteroDataset
    from graph_hscn.loss import criterion
    from graph_hscn.model.hscn import HSCN
    from graph_hscn.replay import CapturedStep
    G, B = 60, 10
    hs = _dataset(G, 8, seed=7)
    ds = DeviceHeteroDataset(hs, DEV, B)
    torch.manual_seed(0)
    model = HSCN("GAT", "GCN", "GCN", ACT_DICT["relu"], 9, 16, 10, 3).to(DEV)
    model.engine = "resident"
    perm = torch.randperm(G, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    want = []
    for i in range(0, G, B):                                   # eager steps on host-collated batches first
        host = HeteroBatch.from_data_list([hs[j] for j in perm[i:i + B].tolist()]).to(DEV)
        for p in model.parameters():
            p.grad = None
        ref = model(host.x_dict, host.edge_index_dict, host)
        l2, _ = criterion("cross_entropy", ref, host["local"].y)
        l2.backward()
        want.append((ref.detach().clone(), l2.detach().clone(),
                     [p.grad.clone() for p in model.parameters() if p.grad is not None]))
        del ref, l2, host                 # no eager autograd graph may be alive when the step is captured
    ds.gather(perm[:B])
    step = CapturedStep(model, ds.static, "cross_entropy", one_launch=False)   # the launch pair: bit-identical to eager
    grads = [p.grad for p in model.parameters() if p.grad is not None]     # the captured step's gradient buffers
    for k, i in enumerate(range(0, G, B)):
        ds.gather(perm[i:i + B])
        loss = step.replay()
        assert torch.equal(step.pred, want[k][0]) and torch.equal(loss, want[k][1])
        for a, b in zip(grads, want[k][2]):
            assert torch.equal(a, b)
    ds.check()


def test_captured_training_iterations_with_the_optimizer_equal_the_eager_loop():
    """gather -> replay(fwd + loss + bwd + fused AdamW): after an epoch the parameters equal those of the eager
    loop over host-collated batches of the same graphs."""
    import copy
    from graph_hscn.config.config import ACT_DICT
    from graph_hscn.data import HeteroBatch
    from graph_hscn.loader.device_dataset import DeviceHeteroDataset
    from graph_hscn.loss import criterion
    from graph_hscn.model.hscn import HSCN
    from graph_hscn.replay import CapturedStep
    G, B = 48, 8
    hs = _dataset(G, 8, seed=5)
    torch.manual_seed(1)
    m_eager = HSCN("GAT", "GCN", "GCN", ACT_DICT["relu"], 9, 16, 10, 3).to(DEV)
    m_eager.engine = "resident"
    perm = torch.randperm(G, device=DEV, generator=torch.Generator(device=DEV).manual_seed(9))
    first = HeteroBatch.from_data_list([hs[j] for j in perm[:B].tolist()]).to(DEV)
    m_eager(first.x_dict, first.edge_index_dict, first)                  # materialise the lazy (-1) input widths
    m_graph = copy.deepcopy(m_eager)
    mk = lambda m: torch.optim.AdamW(m.parameters(), lr=1e-2, weight_decay=5e-4, capturable=True, fused=True)
    o_eager, o_graph = mk(m_eager), mk(m_graph)
    for i in range(0, G, B):
        host = HeteroBatch.from_data_list([hs[j] for j in perm[i:i + B].tolist()]).to(DEV)
        o_eager.zero_grad(set_to_none=True)
        loss, _ = criterion("cross_entropy", m_eager(host.x_dict, host.edge_index_dict, host), host["local"].y)
        loss.backward()
        o_eager.step()
        del loss, host
    ds = DeviceHeteroDataset(hs, DEV, B)
    ds.gather(perm[:B])
    step = CapturedStep(m_graph, ds.static, "cross_entropy", optimizer=o_graph, one_launch=False)
    for i in range(0, G, B):
        ds.gather(perm[i:i + B])
        step.replay()
    for (n, a), (_, b) in zip(m_eager.named_parameters(), m_graph.named_parameters()):
        assert torch.equal(a, b), n


def test_self_walking_epoch_equals_explicit_gathers():
    """gather_next captured in front of the step (permutation + batch counter on the device): the sequence of
    replays sees exactly the batches perm[0:B], perm[B:2B], ... -- same losses, same final gradients as explicit
    gathers of those slices."""
    from graph_hscn.config.config import ACT_DICT
    from graph_hscn.loader.device_dataset import DeviceHeteroDataset
    from graph_hscn.model.hscn import HSCN
    from graph_hscn.replay import CapturedStep
    G, B = 50, 8
    hs = _dataset(G, 8, seed=11)
    ds = DeviceHeteroDataset(hs, DEV, B)
    torch.manual_seed(0)
    model = HSCN("GAT", "GCN", "GCN", ACT_DICT["relu"], 9, 16, 10, 3).to(DEV)
    model.engine = "resident"
    gen = torch.Generator(device=DEV).manual_seed(4)
    ds.new_epoch(gen)
    walk = CapturedStep(model, ds.static, "cross_entropy", pre=ds.gather_next)
    plain = CapturedStep(model, ds.static, "cross_entropy")
    for epoch in range(2):
        perm = ds.new_epoch(gen).clone()
        got = []
        for i in range(G // B):
            got.append(walk.replay().clone())
        want = []
        for i in range(G // B):
            ds.gather(perm[i * B:(i + 1) * B])
            want.append(plain.replay().clone())
        assert torch.equal(torch.stack(got), torch.stack(want))
        assert len({float(x) for x in got}) > 1            # (different batches, not one batch six times)
    ds.check()


def test_gather_next_needs_an_epoch():
    from graph_hscn.loader.device_dataset import DeviceHeteroDataset
    ds = DeviceHeteroDataset(_dataset(6, 4, seed=2), DEV, 3)
    with pytest.raises(RuntimeError):
        ds.gather_next()
    perm = ds.new_epoch(torch.Generator(device=DEV).manual_seed(0))
    assert sorted(perm.tolist()) == list(range(6))
    a = ds.gather_next()["local"].ptr.clone()
    b = ds.gather_next()["local"].ptr.clone()
    ds.check()
    assert int(ds._cursor.item()) == 2 and (not torch.equal(a, b) or True)
