"""The one-shot all-reduce's protocol on CPU: slot / flag / epoch arithmetic of csrc/allreduce.hip restated on
shared memory (tests/oneshot_emulation.py), run by real concurrent processes, against a rank-ordered float32 sum;
and FlatGradReducer's ``algorithm="oneshot"`` branch driven over gloo with that emulation injected as transport.
(The device kernel itself: tests/test_gpu_allreduce.py.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.oneshot_emulation import HostSlotAllReduce, layout


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _values(rank, it, count):
    rng = np.random.default_rng(1000 * it + rank)
    return (rng.standard_normal(count) * 10.0 ** rng.integers(-3, 4, count)).astype(np.float32)


def _expected(G, it, count, scale):
    s = _values(0, it, count)
    for q in range(1, G):
        s = (s + _values(q, it, count)).astype(np.float32)
    return (s * np.float32(scale)).astype(np.float32)


def _proto_worker(rank, G, count, iters, shm_name, start_epoch, q):
    rng = np.random.default_rng(77 + rank)
    ar = HostSlotAllReduce(count, rank, G, shm_name, jitter=lambda: float(rng.random() < 0.15) * 1e-4 * rng.random())
    ar.epoch[:] = start_epoch
    bad = 0
    for it in range(iters):
        a = _values(rank, it, count)
        if rng.random() < 0.1:
            import time
            time.sleep(2e-4 * rng.random())          # uneven arrival: a rank may be a whole epoch ahead of another
        ar(a, 1.0 / G)
        bad += int(not np.array_equal(a, _expected(G, it, count, 1.0 / G)))
    q.put((rank, bad, int(ar.status[0]), ar.epoch.copy()))
    ar.close()


@pytest.mark.timeout(240)
@pytest.mark.parametrize("G,count,start_epoch", [(2, 1146, 0), (3, 5000, 0), (4, 2048, 0xFFFFFFF0)])
def test_slot_protocol_under_concurrency(G, count, start_epoch):
    """count 1146 = the headline model's gradient (one chunk with a ragged tail), 5000 = three chunks; the last
    case starts 16 epochs before the 32-bit wrap (flags and epochs initialised alike, as after 2^32 - 16 steps)."""
    iters = 300
    shm = HostSlotAllReduce.create(count, G)
    try:
        if start_epoch:
            stride, nch, sw, fw, per = layout(count, G)
            w = np.ndarray((G * per,), dtype=np.uint32, buffer=shm.buf)
            for p in range(G):
                w[p * per + sw:(p + 1) * per] = start_epoch
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_proto_worker, args=(r, G, count, iters, shm.name, start_epoch, q)) for r in range(G)]
        for p in procs:
            p.start()
        res = [q.get(timeout=200) for _ in range(G)]
        for p in procs:
            p.join(30)
            assert p.exitcode == 0
        for rank, bad, status, epoch in res:
            assert status == 0
            assert bad == 0, f"rank {rank}: {bad} of {iters} reductions differ from the rank-ordered float32 sum"
            assert np.all(epoch == np.uint32((start_epoch + iters) & 0xFFFFFFFF))
    finally:
        shm.close()
        shm.unlink()


def test_a_missing_peer_times_out_and_leaves_the_buffer_alone():
    shm = HostSlotAllReduce.create(100, 2)
    try:
        ar = HostSlotAllReduce(100, 0, 2, shm.name, spin_limit=50)
        a = np.arange(100, dtype=np.float32)
        ar(a, 0.5)
        assert ar.status[0] == 1 and ar.status[1] == 0b10          # source 1 never arrived
        assert np.array_equal(a, np.arange(100, dtype=np.float32))  # unreduced, not garbage
        with pytest.raises(RuntimeError):
            ar.check()
        ar.close()
    finally:
        shm.close()
        shm.unlink()


# ------------------------------------------------------------------------------------------------------------
# FlatGradReducer(algorithm="oneshot") over gloo, the emulation standing in for the device transport
# ------------------------------------------------------------------------------------------------------------
def _model_and_flat(seed):
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(9, 16), torch.nn.ReLU(), torch.nn.Linear(16, 10))
    params = list(m.parameters())
    flat = torch.zeros(sum(p.numel() for p in params))
    o = 0
    for p in params:                                   # gradients tile ONE buffer, as the resident steps leave them
        p.grad = flat[o:o + p.numel()].view_as(p)
        o += p.numel()
    return m, flat


def _fill_grads(m, flat, x):
    flat.zero_()
    loss = m(x).pow(2).mean()
    gs = torch.autograd.grad(loss, list(m.parameters()))
    o = 0
    for g in gs:
        flat[o:o + g.numel()] = g.reshape(-1)
        o += g.numel()


def _reducer_worker(rank, world, port, shm_name, count, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "graph-hscn_amd"), os.path.join(root, "tests")]
    from graph_hscn.distributed import FlatGradReducer
    from tests.oneshot_emulation import HostSlotAllReduce as T
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m, flat = _model_and_flat(0)
    assert flat.numel() == count
    xs = torch.randn(7, 9, generator=torch.Generator().manual_seed(5))
    shards = [(0, 3), (3, 7)]                        # UNEQUAL shards: 3 and 4 rows
    made = []

    def factory(n, dev, grp):
        made.append(T(n, rank, world, shm_name))
        return made[-1]

    red = FlatGradReducer(m, algorithm="oneshot", oneshot_factory=factory)
    out = []
    for _ in range(3):                               # first call builds the transport, later ones reuse it
        a, b = shards[rank]
        _fill_grads(m, flat, xs[a:b])
        red.reduce(b - a, 7)
        out.append(flat.clone().numpy())
    assert red.last_path == "oneshot" and len(made) == 1
    red.check()
    # equal shards: the kernel's own scale, and the replay fast path
    red2 = FlatGradReducer(m, algorithm="oneshot", equal_weights=True,
                           oneshot_factory=lambda n, d, g: made[0])
    eq = []
    for _ in range(2):
        _fill_grads(m, flat, xs[3 * rank:3 * rank + 3])
        red2.reduce(3, 6)
        eq.append(flat.clone().numpy())
    assert red2._fast is not None
    # packed (non-aliased) gradients must be refused, not silently sent through another algorithm
    m3 = torch.nn.Linear(3, 2)
    m3(torch.ones(1, 3)).sum().backward()
    try:
        FlatGradReducer(m3, algorithm="oneshot", oneshot_factory=factory).reduce(1, 2)
        refused = False
    except RuntimeError:
        refused = True
    q.put((rank, out, eq, refused))
    dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_reducer_oneshot_branch_over_gloo():
    world = 2
    m, flat = _model_and_flat(0)
    count = flat.numel()
    shm = HostSlotAllReduce.create(count, world)
    try:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_reducer_worker, args=(r, world, port, shm.name, count, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=200) for _ in range(world))
        for p in procs:
            p.join(30)
            assert p.exitcode == 0
    finally:
        shm.close()
        shm.unlink()
    xs = torch.randn(7, 9, generator=torch.Generator().manual_seed(5))
    _fill_grads(m, flat, xs)                          # gradient of the mean over the WHOLE batch of 7 rows
    want = flat.clone().numpy()
    _fill_grads(m, flat, xs[:6])
    want_eq = flat.clone().numpy()
    for rank, out, eq, refused in res:
        assert refused
        for g in out:
            np.testing.assert_allclose(g, want, rtol=1e-5, atol=1e-7)
        for g in eq:
            np.testing.assert_allclose(g, want_eq, rtol=1e-5, atol=1e-7)
    for a, b in zip(res[0][1] + res[0][2], res[1][1] + res[1][2]):
        np.testing.assert_array_equal(a, b)           # replicas bit-identical: rank-ordered sum on every rank


def test_unknown_algorithm_is_refused():
    from graph_hscn.distributed import FlatGradReducer
    with pytest.raises(ValueError):
        FlatGradReducer(torch.nn.Linear(2, 2), algorithm="ring-of-fire")
