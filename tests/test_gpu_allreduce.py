"""hscn_allreduce_oneshot on the device (csrc/allreduce.hip; exchange point: reference train/train.py:87-94).

One GPU is what a test box has, so the peers of a rank are (a) itself (G = 1), (b) a second rank object in the same
process, launched concurrently on another stream and addressed by pointer, and (c) a second PROCESS that shares the
GPU and maps the first one's memory through hipIpc -- the path an 8-GPU job takes, minus the xGMI hop.  Expected
values are the rank-ordered float32 sum, compared bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _vals(rank, it, count):
    rng = np.random.default_rng(1000 * it + rank)
    return (rng.standard_normal(count) * 10.0 ** rng.integers(-3, 4, count)).astype(np.float32)


def _expected(G, it, count, scale):
    s = _vals(0, it, count)
    for q in range(1, G):
        s = (s + _vals(q, it, count)).astype(np.float32)
    return (s * np.float32(scale)).astype(np.float32)


@pytest.mark.parametrize("count", [1, 1146, 2048, 5003])
def test_single_rank_is_a_scaling_and_replays_from_a_hipgraph(count):
    from graph_hscn.distributed import OneShotAllReduce
    ar = OneShotAllReduce(count, torch.device(DEV), rank=0, world=1)
    x = torch.from_numpy(_vals(0, 0, count)).to(DEV)
    want = (x.cpu().numpy() * np.float32(0.25)).astype(np.float32)
    ar(x, 0.25)
    assert np.array_equal(x.cpu().numpy(), want)
    nch = ar.epoch.numel()                     # granule form: 256 elements per workgroup; slab form: 2048
    assert nch == (count + 255) // 256
    assert ar.epoch.cpu().tolist() == [1] * nch
    ar.check()
    # captured: the epoch is device state advanced by the kernel, nothing is frozen into the graph
    buf = torch.from_numpy(_vals(0, 1, count)).to(DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ar(buf, 1.0)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ar(buf, 0.5)
    for it in range(5):
        v = _vals(0, 10 + it, count)
        buf.copy_(torch.from_numpy(v))
        g.replay()
        assert np.array_equal(buf.cpu().numpy(), (v * np.float32(0.5)).astype(np.float32))
    assert ar.epoch.cpu().tolist() == [7] * nch
    ar.check()
    ar.close()


def test_unaligned_buffer_takes_the_word_path():
    from graph_hscn.distributed import OneShotAllReduce
    ar = OneShotAllReduce(1001, torch.device(DEV), rank=0, world=1)
    base = torch.zeros(1004, device=DEV)
    x = base[1:1002]                                     # 4-byte aligned only
    v = _vals(0, 3, 1001)
    x.copy_(torch.from_numpy(v))
    ar(x, 2.0)
    assert np.array_equal(x.cpu().numpy(), (v * np.float32(2.0)).astype(np.float32))
    assert float(base[0]) == 0.0 and float(base[1002]) == 0.0
    ar.close()


def test_missing_peer_times_out_flags_it_and_leaves_the_buffer():
    """Rank 0 of a world of two whose peer never launches: bounded spin, status words set, data untouched, the
    stream drains (nothing hangs)."""
    from graph_hscn.distributed import OneShotAllReduce
    a0 = OneShotAllReduce(3000, torch.device(DEV), rank=0, world=2, spin_limit=200, connect=False)
    a1 = OneShotAllReduce(3000, torch.device(DEV), rank=1, world=2, spin_limit=200, connect=False)
    a0.connect([a0.local_info, a1.local_info])
    v = _vals(0, 0, 3000)
    x = torch.from_numpy(v).to(DEV)
    a0(x, 0.5)
    torch.cuda.synchronize()
    assert np.array_equal(x.cpu().numpy(), v)
    st = a0.status.cpu().tolist()
    assert st[0] == 1 and st[1] == 0b10
    with pytest.raises(RuntimeError, match="timed out"):
        a0.check()
    a0.close()
    a1.close()


@pytest.mark.parametrize("count", [1146, 16384, 159381])
def test_two_ranks_in_one_process_on_two_streams(count):
    """Both ranks' kernels run concurrently on one GPU (two streams) and really wait for each other: 1 146 floats =
    the headline model (granule form, 5 workgroups), 16 384 = the largest granule buffer (64 workgroups per rank),
    159 381 = the largest model of SURVEY 8(a9) (slab + flag form, 78 workgroups per rank)."""
    from graph_hscn.distributed import OneShotAllReduce
    dev = torch.device(DEV)
    ars = [OneShotAllReduce(count, dev, rank=r, world=2, connect=False) for r in range(2)]
    for a in ars:
        a.connect([b.local_info for b in ars])
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    bufs = [torch.empty(count, device=DEV) for _ in range(2)]
    torch.cuda.synchronize()
    for it in range(40):
        for r in range(2):
            bufs[r].copy_(torch.from_numpy(_vals(r, it, count)))
        torch.cuda.synchronize()
        order = (0, 1) if it % 2 == 0 else (1, 0)          # either rank may be the one that arrives first
        for r in order:
            with torch.cuda.stream(streams[r]):
                if it % 5 == 0 and r == order[0]:
                    torch.cuda._sleep(200000)               # ... or late by a whole kernel
                ars[r](bufs[r], 0.5)
        torch.cuda.synchronize()
        want = _expected(2, it, count, 0.5)
        for r in range(2):
            assert np.array_equal(bufs[r].cpu().numpy(), want), f"iteration {it}, rank {r}"
    for a in ars:
        a.check()
        a.close()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _ipc_worker(rank, world, port, count, iters, form, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "graph-hscn_amd")]
    if form:
        os.environ["HSCN_ALLREDUCE_FORM"] = form     # (read once per process, before the first call)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    from graph_hscn.distributed import OneShotAllReduce
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)     # host-side exchange of the IPC handles only
        dev = torch.device("cuda", 0)                                    # every rank on the ONE GPU of the box
        torch.cuda.set_device(dev)
        ar = OneShotAllReduce(count, dev, spin_limit=1 << 22)            # collective: all_gather_object + barrier
        buf = torch.empty(count, device=dev)
        bad = 0
        for it in range(iters):
            buf.copy_(torch.from_numpy(_vals(rank, it, count)))
            if (it + rank) % 7 == 0:
                torch.cuda._sleep(300000)
            ar(buf, 1.0 / world)
            if it % 10 == 9:
                bad += int(not np.array_equal(buf.cpu().numpy(), _expected(world, it, count, 1.0 / world)))
        torch.cuda.synchronize()
        st = ar.status.cpu().tolist()
        dist.barrier()
        ar.close()
        q.put((rank, bad, st, ar.kind, None))
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, -1, None, None, traceback.format_exc()))
        raise e


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,form", [(2, ""), (3, ""), (2, "slabs")])
def test_processes_sharing_the_gpu_through_hipipc(world, form):
    """The multi-GPU set-up path end to end -- fine-grained allocation, hipIpcGetMemHandle, all_gather_object,
    hipIpcOpenMemHandle -- and the protocol between kernels of DIFFERENT processes (separate queues, no common
    stream order), 200 exchanges with uneven arrival.  form "" = what the size selects (1 146 floats: granules),
    "slabs" forces the slab + flag form onto the same buffer."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    count, iters = 1146, 200
    procs = [ctx.Process(target=_ipc_worker, args=(r, world, port, count, iters, form, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
    for rank, bad, st, kind, err in res:
        assert err is None, err
        assert st == [0, 0], f"rank {rank}: a wait timed out, status {st}"
        assert bad == 0, f"rank {rank}: {bad} checked exchanges differ from the rank-ordered sum"
        print(f"rank {rank}: comm memory kind {kind} (0 = fine-grained)")
    for p in procs:
        assert p.exitcode == 0
