"""Data-parallel path on CPU with gloo, world_size 2: graphs sharded across ranks,
flat-buffer gradient all-reduce; the result must equal the single-process
gradient of the mean loss over the whole batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build(seed, B, K):
    from graph_hscn.loader.synthetic import make_dataset
    from oracle import hetero_data as OH
    graphs = make_dataset("peptides_func", B, seed=seed)
    rng = np.random.default_rng(seed)
    return [OH.hetero_from_clusters(g.x, g.edge_index, g.y, rng.integers(0, K, g.num_nodes), K) for g in graphs], \
        [g.num_nodes for g in graphs]


def _grads(model, hs):
    from oracle import hetero_data as OH
    from oracle import models as OM
    b = OH.collate_hetero(hs)
    model.zero_grad(set_to_none=True)
    out = model(b["x_dict"], b["edge_index_dict"], b["batch_local"], len(hs))
    loss, _ = OM.criterion("cross_entropy", out, b["y"])
    loss.backward()
    return loss.detach()


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "graph-hscn_amd")]
    from graph_hscn.distributed import FlatGradReducer, shard_bounds
    from oracle import models as OM
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hs, sizes = _build(3, 10, 8)
    torch.manual_seed(0)
    model = OM.HSCN("GAT", "GCN", "GCN", OM.ACT["relu"], 9, 16, 10, 2)
    bounds = shard_bounds(sizes, world)
    mine = hs[bounds[rank]:bounds[rank + 1]]
    _grads(model, mine)
    red = FlatGradReducer(model)
    red.reduce(len(mine))                     # total weight found by all-reduce
    g1 = torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.grad is not None])
    _grads(model, mine)
    red.reduce(len(mine), len(hs))            # total weight given: no host sync
    g2 = torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.grad is not None])
    # gradients that already tile one flat buffer (what the graph-resident backward returns) are reduced in place
    _grads(model, mine)
    with_grad = [p for p in model.parameters() if p.grad is not None]
    flat = torch.cat([p.grad.reshape(-1) for p in with_grad]).clone()
    flat0 = flat.clone()
    o = 0
    for p in with_grad:
        p.grad = flat[o:o + p.numel()].view_as(p)
        o += p.numel()
    red.reduce(len(mine), len(hs))
    assert red.last_path == "aliased"
    g3 = torch.cat([p.grad.reshape(-1) for p in with_grad])
    # a replayed step writes the same buffers again: the reducer reuses its flat view (no layout check)
    assert red._fast is not None
    flat.copy_(flat0)
    red.reduce(len(mine), len(hs))
    g4 = torch.cat([p.grad.reshape(-1) for p in with_grad])
    assert torch.equal(g3, g4)
    q.put((rank, g1.numpy(), g2.numpy(), [p.grad is None for p in model.parameters()], g3.numpy()))
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_gradients_equal_single_process_gradients():
    from oracle import models as OM
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in range(world)]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    hs, _ = _build(3, 10, 8)
    torch.manual_seed(0)
    model = OM.HSCN("GAT", "GCN", "GCN", OM.ACT["relu"], 9, 16, 10, 2)
    _grads(model, hs)
    want = torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.grad is not None]).numpy()
    none_mask = [p.grad is None for p in model.parameters()]
    for rank, g1, g2, mask, g3 in res:
        assert mask == none_mask                      # virtual-branch params stay grad-less on every rank
        np.testing.assert_allclose(g1, want, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(g2, want, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(g3, want, rtol=1e-5, atol=1e-7)
    np.testing.assert_array_equal(res[0][1], res[1][1])  # ranks hold identical reduced gradients


def test_reducer_is_identity_without_a_process_group():
    from graph_hscn.distributed import FlatGradReducer
    m = torch.nn.Linear(3, 2)
    m(torch.randn(4, 3)).sum().backward()
    before = [p.grad.clone() for p in m.parameters()]
    FlatGradReducer(m).reduce(4)
    for a, p in zip(before, m.parameters()):
        assert torch.equal(a, p.grad)


# --------------------------------------------------------------------------------------------------------
# the product training loop (train/train.py::train_epoch, reference train/train.py:54-106) with a reducer on
# sharded loaders, world_size 2: end-of-epoch parameters == the single-process run on the concatenated batches
# --------------------------------------------------------------------------------------------------------
def _cpu_hscn(seed=0):
    """The product ``HSCN`` class (what ``train_epoch`` dispatches on, train/train.py:74) computing through the CPU
    oracle: test scaffolding for the host-side loop + reducer logic, which is all a CPU box can run."""
    from graph_hscn.model.hscn import HSCN
    from oracle import models as OM

    class CpuHSCN(HSCN):
        def __init__(self):
            torch.nn.Module.__init__(self)
            torch.manual_seed(seed)
            self.oracle = OM.HSCN("GAT", "GCN", "GCN", OM.ACT["relu"], 9, 16, 10, 2)

        def forward(self, x_dict, edge_index_dict, batch):
            return self.oracle(x_dict, edge_index_dict, batch["local"].batch, int(batch.num_graphs))

    return CpuHSCN()


def _hetero_graphs(n, K=8, seed=11):
    from graph_hscn.loader.hetero_data import hetero_from_clusters
    from graph_hscn.loader.synthetic import make_dataset
    graphs = make_dataset("peptides_func", n, seed=seed)
    rng = np.random.default_rng(seed)
    return [hetero_from_clusters(g, rng.integers(0, K, g.num_nodes), K) for g in graphs]


# per-rank batch sizes of the three steps of the epoch: UNEQUAL on purpose (16/15/17-style weights are the case
# in which a per-rank AVG-vs-SUM decision diverges)
_SPLITS = [((0, 3), (3, 8)), ((8, 12), (12, 16)), ((16, 21), (21, 23))]


def _epoch_worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "graph-hscn_amd")]
    from graph_hscn.data import HeteroBatch
    from graph_hscn.distributed import FlatGradReducer
    from graph_hscn.train.train import train_epoch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hs = _hetero_graphs(23)
    model = _cpu_hscn()
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    loader = [HeteroBatch.from_data_list(hs[a:b]) for (a, b) in (s[rank] for s in _SPLITS)]
    red = FlatGradReducer(model)
    loss, _ = train_epoch(0, None, loader, model, opt, "cross_entropy", None, 1, False, reducer=red)
    # equal_weights=True with unequal shards must refuse, identically on every rank, before any collective runs
    bad = FlatGradReducer(model, equal_weights=True)
    bad._fast_avg = True
    try:
        bad._check_equal(3 / 8, 2)
        refused = False
    except ValueError:
        refused = True
    q.put((rank, [p.detach().numpy().copy() for p in model.parameters()], refused))
    dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_two_rank_train_epoch_equals_single_process_on_concatenated_batches():
    from graph_hscn.data import HeteroBatch
    from graph_hscn.train.train import train_epoch
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_epoch_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=200) for _ in range(world)]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    hs = _hetero_graphs(23)
    model = _cpu_hscn()
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    loader = [HeteroBatch.from_data_list(hs[s[0][0]:s[1][1]]) for s in _SPLITS]
    train_epoch(0, None, loader, model, opt, "cross_entropy", None, 1, False)
    want = [p.detach().numpy() for p in model.parameters()]
    for rank, got, refused in res:
        assert refused
        for a, b in zip(got, want):
            np.testing.assert_allclose(a, b, rtol=2e-5, atol=2e-7)
    for a, b in zip(res[0][1], res[1][1]):
        np.testing.assert_array_equal(a, b)          # replicas stay bit-identical
