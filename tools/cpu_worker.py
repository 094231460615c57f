"""One CPU worker of bench.py's all-cores baseline legs: waits for a common start time, runs `count` units of the chosen task
through the compiled reference core (oracle/_ref, or the repo's restatement when that is absent) and prints the epoch at
which it finished.  A separate PROCESS per core, like the reference's DataLoader workers (config.py:40 num_workers):
threads of one process serialise on the kernel's address-space lock while each call page-faults its 72 MB result."""
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))


def main():
    task, count, start, seed = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
    import numpy as np
    import torch
    torch.set_num_threads(1)
    from gaussreg_amd import synthetic
    from oracle import capi
    ref = capi.have_ref()
    rn = capi.ref_radius_neighbors if ref else capi.radius_neighbors
    gs = capi.ref_grid_subsampling if ref else capi.grid_subsampling
    if task == "radius200k":
        p, l = synthetic.cloud_200k(1, seed=100 + seed)
        p, l = p.numpy(), l.numpy()

        def unit():
            rn(p, p, l, l, 0.0625)
    elif task == "pyramid":
        from gaussreg_amd import pair_pipeline
        a, b, _ = pair_pipeline.synthetic_room_pair(seed, 30000, torch.device("cpu"))
        pts = np.concatenate([a.numpy(), b.numpy()])
        lens = np.array([30000, 30000], np.int64)

        def unit():
            plist, llist, voxel, rad = [pts], [lens], 0.025, 0.0625
            for i in range(1, 5):
                voxel *= 2
                p2, l2 = gs(plist[-1], llist[-1], voxel)
                plist.append(p2)
                llist.append(l2)
            for i in range(5):
                rn(plist[i], plist[i], llist[i], llist[i], rad)
                if i < 4:
                    rn(plist[i + 1], plist[i], llist[i + 1], llist[i], rad)
                    rn(plist[i], plist[i + 1], llist[i], llist[i + 1], 2 * rad)
                rad *= 2
    else:
        raise SystemExit("unknown task " + task)
    while time.time() < start:
        time.sleep(0.001)
    t0 = time.time()
    for _ in range(count):
        unit()
    print(f"DONE {t0:.6f} {time.time():.6f}", flush=True)


if __name__ == "__main__":
    main()
