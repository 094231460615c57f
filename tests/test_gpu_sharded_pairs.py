"""configs[4] across process boundaries on ONE GPU: two processes (world size 2, gloo rendezvous, both on cuda:0) each run
PairRegistrar.register_many on their block of a pair list and the rows meet in sharding.gather_rows -- the code path of
`bench.py --gpus N` (RCCL there), so that the first real multi-GPU run is not also the first time pair results cross a process
boundary.  The gathered table must equal the one a single process computes for the whole list, row for row."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

N_PAIRS, N_POINTS, N_SAMPLE = 5, 20000, 6000


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rows(lo, hi):
    from gaussreg_amd import pair_pipeline
    dev = torch.device("cuda", 0)
    pairs = [pair_pipeline.synthetic_room_pair(1000 + i, N_POINTS, dev) for i in range(lo, hi)]
    reg = pair_pipeline.PairRegistrar(dev, num_samples=N_SAMPLE)
    out = reg.register_many(pairs, 2)
    reg.close()
    return out


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gaussreg_amd import sharding
        a, b = sharding.shard_bounds(N_PAIRS, rank, world)
        local = _rows(a, b)
        counts = [sharding.shard_bounds(N_PAIRS, r, world)[1] - sharding.shard_bounds(N_PAIRS, r, world)[0] for r in range(world)]
        allres = sharding.gather_rows(local, counts)
        assert allres.is_cuda
        q.put((rank, (a, b), allres.cpu().numpy().copy()))
    finally:
        dist.destroy_process_group()


def test_two_processes_one_gpu_register_and_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=600) for _ in range(2)]
    finally:
        for p in procs:
            p.join(timeout=120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    # the same two blocks computed by ONE process (RANSAC's seed is the pair's position in its register_many block, so the
    # blocks are formed the way the ranks form them): every kernel on the path is deterministic, the tables must be equal
    want = torch.cat([_rows(0, 3), _rows(3, 5)]).cpu().numpy()
    spans = sorted(r[1] for r in res)
    assert spans == [(0, 3), (3, 5)]
    for rank, _, got in res:
        assert got.shape == want.shape == (N_PAIRS, 20)
        assert np.array_equal(got[:, 18], want[:, 18])                       # number of correspondences
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-6)
    assert np.array_equal(res[0][2], res[1][2])                              # both ranks hold the same table
    assert (want[:, 16] < 5.0).all() and (want[:, 17] < 0.2).all()           # and the pairs did register
