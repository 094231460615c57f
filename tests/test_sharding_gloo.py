"""CPU: the N>1 path (per-unit sharding + one gather) with world_size 2 over gloo."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gaussreg_amd import sharding


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _unit_result(u):
    # stand-in for "register pair u": a flattened 4x4 transform + 3 metrics, deterministic in u
    g = torch.Generator().manual_seed(int(u))
    return torch.cat([torch.eye(4).reshape(-1) + 0.01 * torch.rand(16, generator=g), torch.tensor([u, 2.0 * u, -1.0 * u])])


def _worker(rank, world, port, n_units, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        units = list(range(100, 100 + n_units))
        out = sharding.run_sharded(units, _unit_result)
        a, b = sharding.shard_bounds(n_units, rank, world)
        ragged = sharding.gather_rows(torch.arange(a, b, dtype=torch.float32).reshape(-1, 1))
        # by value (numpy), not as shared-memory tensors: a worker may exit before the parent has read the queue
        q.put((rank, out.numpy().copy(), ragged.reshape(-1).numpy().copy(), (a, b)))
    finally:
        dist.destroy_process_group()


def _run_world2(n_units):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_units, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=300) for _ in range(2)]
    finally:
        for p in procs:
            p.join(timeout=120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    return [(r, torch.from_numpy(o), torch.from_numpy(g), ab) for r, o, g, ab in res]


@pytest.mark.parametrize("n_units", [7, 1, 2])
def test_run_sharded_world2(n_units):
    res, err = None, None
    for _ in range(3):  # the probed port can be taken before the rendezvous binds it
        try:
            res = _run_world2(n_units)
            break
        except Exception as e:  # noqa: BLE001
            err = e
    assert res is not None, err
    want = torch.stack([_unit_result(u) for u in range(100, 100 + n_units)])
    bounds = sorted(r[3] for r in res)
    assert bounds[0][0] == 0 and bounds[-1][1] == n_units and bounds[0][1] == bounds[1][0]
    for rank, out, ragged, _ in res:
        assert out.shape == want.shape and torch.equal(out, want)
        assert torch.equal(ragged, torch.arange(n_units, dtype=torch.float32))


def test_shard_bounds_cover_everything():
    for n in (0, 1, 5, 8, 1024):
        for w in (1, 2, 3, 8):
            spans = [sharding.shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_is_identity():
    out = sharding.run_sharded([1, 2, 3], _unit_result)
    assert torch.equal(out, torch.stack([_unit_result(u) for u in (1, 2, 3)]))
