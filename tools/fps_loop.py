"""FPS timings on the GPU box: 2 x (200 k -> 30 k) and the batched call of the configs[4] pipeline (25 clouds per call: ten workgroups per cloud on 256 CUs).
    python tools/fps_loop.py [--clouds 25]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussreg_amd import pair_pipeline
from gaussreg_amd.registration import farthest_point_sampling


def ms_of(fn, iters, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clouds", type=int, default=25)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    out = {}
    r_, s_, _ = pair_pipeline.synthetic_room_pair(0, 200000, dev)
    big = torch.cat([r_, s_]).contiguous()
    out["fps_2x200k_to_30k_ms"] = round(ms_of(lambda: farthest_point_sampling(big, [200000] * 2, [30000] * 2), 3, 1), 3)
    clouds = []
    for i in range((a.clouds + 1) // 2):
        r_, s_, _ = pair_pipeline.synthetic_room_pair(i, 200000, dev)
        clouds += [r_, s_]
    clouds = clouds[:a.clouds]
    many = torch.cat(clouds).contiguous()
    ms = ms_of(lambda: farthest_point_sampling(many, [200000] * a.clouds, [30000] * a.clouds), 3, 1)
    out[f"fps_{a.clouds}x200k_to_30k_ms"] = round(ms, 3)
    out["ms_per_cloud_batched"] = round(ms / a.clouds, 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
