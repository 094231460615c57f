"""Dev tool: does the co-operative FPS launch run next to other kernels?  One host thread loops FPS calls (25 clouds of 200 k ->
30 k) on its own stream, another loops the 64-pair pyramid on its own stream; each alone, then together."""
import os, sys, threading, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from gaussreg_amd import pair_pipeline
from gaussreg_amd.data import precompute_data_stack_mode
from gaussreg_amd.registration import farthest_point_sampling
dev = torch.device("cuda", 0)
clouds = []
for b in range(13):
    r_, s_, _ = pair_pipeline.synthetic_room_pair(b, 200000, dev)
    clouds += [r_, s_]
clouds = clouds[:25]
big = torch.cat(clouds).contiguous()
lens = [200000] * 25
small = []
for b in range(64):
    r_, s_, _ = pair_pipeline.synthetic_room_pair(b, 30000, dev)
    small += [r_, s_]
bp = torch.cat(small).contiguous()
bl = torch.tensor([30000] * 128)


gate = threading.Barrier(2)


def fps_loop(n, out, together=False):
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        farthest_point_sampling(big, lens, [30000] * 25)
        s.synchronize()
        if together:
            gate.wait()
        t0 = time.perf_counter()
        for _ in range(n):
            farthest_point_sampling(big, lens, [30000] * 25)
        s.synchronize()
        out["fps_ms_per_call"] = (time.perf_counter() - t0) / n * 1e3
        out["t0"], out["t1"] = t0, time.perf_counter()


def pyr_loop(n, out, together=False):
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        precompute_data_stack_mode(bp, bl, 5, 0.025, 0.0625, [89, 30, 43, 49, 49])
        s.synchronize()
        if together:
            gate.wait()
        t0 = time.perf_counter()
        for _ in range(n):
            precompute_data_stack_mode(bp, bl, 5, 0.025, 0.0625, [89, 30, 43, 49, 49])
        s.synchronize()
        out["pyramid_ms_per_call"] = (time.perf_counter() - t0) / n * 1e3
        out["t0"], out["t1"] = t0, time.perf_counter()


a, b = {}, {}
fps_loop(6, a)
pyr_loop(6, b)
print("alone   : fps %.2f ms per call, pyramid %.2f ms per call" % (a["fps_ms_per_call"], b["pyramid_ms_per_call"]), flush=True)
a2, b2 = {}, {}
t1 = threading.Thread(target=fps_loop, args=(8, a2, True))
t2 = threading.Thread(target=pyr_loop, args=(3, b2, True))
t1.start(); t2.start(); t1.join(); t2.join()
span = max(a2["t1"], b2["t1"]) - min(a2["t0"], b2["t0"])
print("together: fps %.2f ms per call, pyramid %.2f ms per call; both loops done in %.1f ms (sum of the two alone: %.1f ms)"
      % (a2["fps_ms_per_call"], b2["pyramid_ms_per_call"], span * 1e3, 8 * a["fps_ms_per_call"] + 3 * b["pyramid_ms_per_call"]), flush=True)
