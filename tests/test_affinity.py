"""CPU: the rank -> NUMA node -> CPU set logic of gaussreg_amd/affinity.py on a fake topology (two sockets, eight GPUs, one
GPU without node information), and bench.py --dry over gloo with that topology injected (tools/first_8gpu_run.md)."""
import json
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from gaussreg_amd import affinity  # noqa: E402


def _fake_sysfs(root, gpu_nodes, node_cpulists):
    addrs = []
    for i, node in enumerate(gpu_nodes):
        addr = "0000:%02x:00.0" % (0x10 + i)
        d = os.path.join(root, "sys/bus/pci/devices", addr)
        os.makedirs(d)
        with open(os.path.join(d, "numa_node"), "w") as fh:
            fh.write(f"{node}\n")
        addrs.append(addr)
    for node, cl in node_cpulists.items():
        d = os.path.join(root, "sys/devices/system/node", f"node{node}")
        os.makedirs(d)
        with open(os.path.join(d, "cpulist"), "w") as fh:
            fh.write(cl + "\n")
    return addrs


def test_cpulist_round_trip():
    assert affinity.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert affinity.format_cpulist([11, 10, 8, 0, 1, 2, 3]) == "0-3,8,10-11"
    assert affinity.parse_cpulist("") == []


def test_plan_two_sockets_eight_gpus(tmp_path):
    root = str(tmp_path)
    addrs = _fake_sysfs(root, [0, 0, 0, 0, 1, 1, 1, 1], {0: "0-63,128-191", 1: "64-127,192-255"})
    pci = lambda r: addrs[r]  # noqa: E731
    allowed = list(range(256))
    seen = set()
    for r in range(8):
        cpus, info = affinity.plan(r, 8, root, pci, allowed)
        assert info["numa_node"] == (0 if r < 4 else 1) and info["source"] == "numa"
        node_cpus = set(affinity.parse_cpulist("0-63,128-191" if r < 4 else "64-127,192-255"))
        assert len(cpus) == 32 and set(cpus) <= node_cpus      # four ranks share a node's 128 CPUs
        assert not (set(cpus) & seen)                           # nobody shares a CPU
        seen |= set(cpus)
    assert len(seen) == 256


def test_plan_without_node_information_splits_evenly(tmp_path):
    root = str(tmp_path)
    addrs = _fake_sysfs(root, [-1, -1], {})
    pci = lambda r: addrs[r]  # noqa: E731
    c0, i0 = affinity.plan(0, 2, root, pci, list(range(16)))
    c1, i1 = affinity.plan(1, 2, root, pci, list(range(16)))
    assert i0["numa_node"] is None and i0["source"] == "even-split"
    assert c0 == list(range(8)) and c1 == list(range(8, 16))
    # a GPU torch cannot name (no PCI address): same fallback
    c, i = affinity.plan(0, 1, root, lambda r: None, [3, 5])
    assert c == [3, 5] and i["source"] == "even-split"


def test_bind_rank_applies_the_plan_and_honours_the_switch(tmp_path, monkeypatch):
    root = str(tmp_path)
    addrs = _fake_sysfs(root, [1, 0], {0: "0-7", 1: "8-15"})
    got = {}
    info = affinity.bind_rank(0, 2, sysfs_root=root, pci_address=lambda r: addrs[r], apply=lambda c: got.setdefault("cpus", list(c)))
    want, _ = affinity.plan(0, 2, root, lambda r: addrs[r])
    assert info["bound"] and info["numa_node"] == 1 and got["cpus"] == want
    monkeypatch.setenv("GR_AFFINITY", "0")
    assert affinity.bind_rank(0, 2)["bound"] is False


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _dry_worker(rank, world, port, out_dir, fake_root, fake_pci):
    import contextlib
    import io
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      LOCAL_WORLD_SIZE=str(world), GR_FAKE_SYSFS=fake_root, GR_FAKE_PCI=fake_pci)
    import bench
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rc = bench.main(["--gpus", str(world), "--dry", "--backend", "gloo"])
    with open(os.path.join(out_dir, f"dry{rank}.txt"), "w") as fh:
        fh.write(f"{rc}\n{buf.getvalue()}")


def test_dry_line_reports_every_ranks_placement(tmp_path):
    root = str(tmp_path / "fake")
    os.makedirs(root)
    ncpu = 10 ** 4  # CPUs this host does not have: with GR_FAKE_SYSFS the plan is reported, not applied
    addrs = _fake_sysfs(root, [0, 1], {0: f"{ncpu}-{ncpu + 7}", 1: f"{ncpu + 8}-{ncpu + 15}"})
    # (plan() intersects with the process's own mask: give the fake nodes real CPUs as well so the intersection is not empty)
    real = sorted(os.sched_getaffinity(0))
    half = max(len(real) // 2, 1)
    for node, cpus in ((0, real[:half]), (1, real[half:] or real[:half])):
        with open(os.path.join(root, "sys/devices/system/node", f"node{node}", "cpulist"), "w") as fh:
            fh.write(affinity.format_cpulist(cpus) + "\n")
    last = None
    for attempt in range(3):
        try:
            mp.spawn(_dry_worker, args=(2, _free_port(), str(tmp_path), root, ",".join(addrs)), nprocs=2, join=True)
            last = None
            break
        except Exception as e:  # noqa: BLE001
            last = e
    if last is not None:
        raise last
    rc, _, out = open(os.path.join(str(tmp_path), "dry0.txt")).read().partition("\n")
    assert rc.strip() == "0"
    line = json.loads(out.strip().splitlines()[-1])
    assert line["dry"] and line["ok"] and len(line["placement"]) == 2
    assert [p["numa_node"] for p in line["placement"]] == [0, 1]
    assert all(p["bound"] and p["cpus"] >= 1 for p in line["placement"])
    assert line["placement"][0]["cpu_affinity"] == affinity.format_cpulist(real[:half])
