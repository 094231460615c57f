"""Host-side placement of a rank next to its GPU (one process per GPU: base_trainer.py:66-69 in the reference).

Every rank of `bench.py --gpus N` (and of a pair-sharded job) spin-polls a host-mapped mailbox page and marshals a few hundred
ctypes calls per second; eight such ranks left to the scheduler share cores with each other and sit on the wrong socket for
half of the GPUs.  `bind_rank` pins the calling process to the CPUs of the NUMA node its GPU hangs off, BEFORE the library
allocates its pinned pages (first-touch then puts the mailbox page and the staging buffers on that node):

    GPU ordinal -> PCI address (torch.cuda.get_device_properties) -> /sys/bus/pci/devices/<addr>/numa_node
                -> /sys/devices/system/node/node<k>/cpulist -> os.sched_setaffinity

Ranks that share a node split its CPUs evenly.  Without a usable node (numa_node = -1: single-socket boxes, containers) the
visible CPUs are split evenly by local rank instead.  Everything that touches the system is injectable (`sysfs_root`,
`pci_address`, `apply`) so the CPU suite drives the logic with a fake topology (tests/test_affinity.py).
"""
import os


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    cpus = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            cpus.extend(range(int(a), int(b) + 1))
        else:
            cpus.append(int(part))
    return sorted(set(cpus))


def format_cpulist(cpus):
    """[0, 1, 2, 3, 8] -> '0-3,8'"""
    cpus = sorted(set(cpus))
    out, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(out)


def _read(path):
    try:
        with open(path) as fh:
            return fh.read().strip()
    except OSError:
        return None


def torch_pci_address(ordinal):
    """PCI address 'dddd:bb:dd.f' of GPU `ordinal` as torch sees it (None without a GPU / on an old torch)."""
    try:
        import torch
        p = torch.cuda.get_device_properties(ordinal)
        return "%04x:%02x:%02x.0" % (int(getattr(p, "pci_domain_id", 0)), int(p.pci_bus_id), int(p.pci_device_id))
    except Exception:  # noqa: BLE001
        return None


def gpu_numa_node(ordinal, sysfs_root="/", pci_address=torch_pci_address):
    """NUMA node of GPU `ordinal`, or None when the system does not say (numa_node = -1, no sysfs entry)."""
    addr = pci_address(ordinal)
    if not addr:
        return None
    txt = _read(os.path.join(sysfs_root, "sys/bus/pci/devices", addr, "numa_node"))
    try:
        node = int(txt)
    except (TypeError, ValueError):
        return None
    return node if node >= 0 else None


def node_cpus(node, sysfs_root="/"):
    txt = _read(os.path.join(sysfs_root, "sys/devices/system/node", f"node{node}", "cpulist"))
    return parse_cpulist(txt) if txt else []


def plan(local_rank, local_world, sysfs_root="/", pci_address=torch_pci_address, allowed=None):
    """The CPUs rank `local_rank` of `local_world` ranks on this host should run on: (cpus, info).  `allowed`: the CPUs the
    process may use at all (default: its current affinity mask)."""
    if allowed is None:
        try:
            allowed = sorted(os.sched_getaffinity(0))
        except AttributeError:
            allowed = list(range(os.cpu_count() or 1))
    allowed = sorted(allowed)
    nodes = [gpu_numa_node(r, sysfs_root, pci_address) for r in range(local_world)]
    node = nodes[local_rank] if local_rank < len(nodes) else None
    info = {"local_rank": local_rank, "numa_node": node, "source": "numa"}
    cpus = []
    if node is not None:
        on_node = [c for c in node_cpus(node, sysfs_root) if c in set(allowed)]
        sharers = [r for r in range(local_world) if nodes[r] == node]  # ranks whose GPUs hang off the same node
        k, n = sharers.index(local_rank), len(sharers)
        per = len(on_node) // n
        cpus = on_node[k * per:(k + 1) * per] if per > 0 else on_node
    if not cpus:  # no node information: an even split of what the process may use
        info["source"] = "even-split"
        per = max(len(allowed) // max(local_world, 1), 1)
        cpus = allowed[local_rank * per:(local_rank + 1) * per] or allowed
    info["cpu_affinity"] = format_cpulist(cpus)
    info["cpus"] = len(cpus)
    return cpus, info


def bind_rank(local_rank, local_world, sysfs_root=None, pci_address=None, apply=None):
    """Pin the calling process per `plan`; returns the info dict (+ "bound").  Environment hooks for tests and odd hosts:
    GR_AFFINITY=0 disables the binding, GR_FAKE_SYSFS=<dir> / GR_FAKE_PCI="addr0,addr1,..." replace the topology source."""
    if os.environ.get("GR_AFFINITY", "1") == "0":
        return {"local_rank": local_rank, "bound": False, "source": "disabled (GR_AFFINITY=0)"}
    fake_root, fake_pci = os.environ.get("GR_FAKE_SYSFS"), os.environ.get("GR_FAKE_PCI")
    if sysfs_root is None:
        sysfs_root = fake_root or "/"
    if pci_address is None:
        if fake_pci:
            table = fake_pci.split(",")
            pci_address = lambda r: table[r] if r < len(table) else None  # noqa: E731
        else:
            pci_address = torch_pci_address
    cpus, info = plan(local_rank, local_world, sysfs_root, pci_address)
    if fake_root and apply is None:
        apply = lambda c: None  # noqa: E731  (a fake topology names CPUs this host may not have)
    try:
        (apply or (lambda c: os.sched_setaffinity(0, c)))(cpus)
        info["bound"] = True
    except (OSError, AttributeError, ValueError) as e:
        info["bound"] = False
        info["error"] = str(e)
    return info
