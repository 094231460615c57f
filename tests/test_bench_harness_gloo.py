"""CPU: bench.py's own N>1 code path -- Harness (barrier, MAX-over-ranks timing), throughput_line (whole-job aggregate),
the pair-result gather -- with world_size 2 over gloo and the GPU kernels replaced by a stub step; plus the `--gpus N`
guard rails (never report fewer ranks than asked for)."""
import os
import socket
import sys
import time

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402
from gaussreg_amd import sharding  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    h = bench.Harness.from_env("gloo")
    try:
        assert (h.rank, h.world) == (rank, world)
        calls = {"n": 0}

        def step():  # rank 1 is the slow one: the job's time is ITS time
            calls["n"] += 1
            time.sleep(0.01 * (rank + 1))

        steps, warm, units = 5, 2, 32
        elapsed = h.timed(step, steps, warm)
        assert calls["n"] == steps + warm
        line = bench.throughput_line(h, "stub", "units/s", units, steps, warm, elapsed)
        # the configs[4] gather: this rank's block of a 5-pair list, rows = [pair id, rank]
        n_total = 5
        a, b = sharding.shard_bounds(n_total, rank, world)
        local = torch.tensor([[float(i), float(rank)] for i in range(a, b)]).reshape(-1, 2)
        counts = [sharding.shard_bounds(n_total, r, world)[1] - sharding.shard_bounds(n_total, r, world)[0] for r in range(world)]
        allres = sharding.gather_rows(local, counts)
        torch.save({"line": line, "elapsed": elapsed, "rows": allres}, os.path.join(out_dir, f"r{rank}.pt"))
    finally:
        h.close()


def test_harness_world2_max_over_ranks_and_gather(tmp_path):
    last = None
    for attempt in range(3):  # a free port can be taken between the probe and the rendezvous
        try:
            mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
            last = None
            break
        except Exception as e:  # noqa: BLE001
            last = e
    assert last is None, last
    res = [torch.load(os.path.join(str(tmp_path), f"r{r}.pt")) for r in range(2)]
    for r in res:
        line = r["line"]
        assert line["n_gpus"] == 2 and line["steps"] == 5 and line["warmup"] == 2 and line["scaling"] == "weak"
        assert r["elapsed"] >= 5 * 0.02 * 0.95                      # the slow rank's time, on BOTH ranks
        assert abs(line["value"] - 2 * 32 * 5 / r["elapsed"]) < 1e-2 * line["value"]   # units of all ranks / max time
        assert abs(line["ms_per_step"] - r["elapsed"] / 5 * 1e3) < 1e-3
        assert torch.equal(r["rows"][:, 0], torch.arange(5.0)) and r["rows"][:, 1].tolist() == [0, 0, 0, 1, 1]
    assert abs(res[0]["elapsed"] - res[1]["elapsed"]) < 1e-9         # MAX over ranks: identical everywhere


def test_gpus_flag_never_silently_shrinks(monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("this box really has 2 GPUs")
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "2"])
    assert "refusing" in str(e.value)
    monkeypatch.setenv("WORLD_SIZE", "4")
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "2"])
    assert "must agree" in str(e.value)


def test_single_rank_harness_is_plain_timing():
    h = bench.Harness(0, 1, torch.device("cpu"))
    el = h.timed(lambda: time.sleep(0.002), 3, 1)
    assert el >= 0.006 and h.max_over_ranks(1.5) == 1.5
    line = bench.throughput_line(h, "m", "u/s", 4, 3, 1, el)
    assert line["n_gpus"] == 1 and abs(line["value"] - 12 / el) < 1e-2 * line["value"]


def _dry_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import contextlib
    import io
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rc = bench.main(["--gpus", str(world), "--dry", "--backend", "gloo"])
    with open(os.path.join(out_dir, f"dry{rank}.txt"), "w") as fh:
        fh.write(f"{rc}\n{buf.getvalue()}")


@pytest.mark.parametrize("world", [2, 3])
def test_dry_mode_runs_every_collective_of_the_bench(tmp_path, world):
    """`bench.py --gpus N --dry` (tools/first_8gpu_run.md): the code a first multi-GPU node runs in its first minute, here over
    gloo -- the backend decides where the tensors live, nothing else."""
    import json
    last = None
    for attempt in range(3):
        try:
            mp.spawn(_dry_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
            last = None
            break
        except Exception as e:  # noqa: BLE001
            last = e
    assert last is None, last
    for r in range(world):
        rc, _, out = open(os.path.join(str(tmp_path), f"dry{r}.txt")).read().partition("\n")
        assert rc == "0"
        if r == 0:
            line = json.loads(out)
            assert line["dry"] and line["ok"] and line["n_gpus"] == world and line["ranks_ok"] == world and line["backend"] == "gloo"
            assert all(line["checks"].values()) and len(line["checks"]) == 5
        else:
            assert out.strip() == ""
