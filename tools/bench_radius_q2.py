"""Dev tool: the round-4 search kernel (q2_kernel, gr_radius_search_mode 2) against the older paths on the bench workload
(8 x 200 k-point clouds, r = 0.0625): `radius_search(limit 40)` and the bare `ext.radius_neighbors`.  Environment switches
are read once per process, so every configuration runs in its own interpreter.
    python tools/bench_radius_q2.py            modes 0 / 1 / 2
    python tools/bench_radius_q2.py --ablate   cumulative time of the fused launch stopped after each phase"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def child():
    import ctypes
    import torch
    from gaussreg_amd import _lib, ext, synthetic
    L = _lib.lib()
    B = int(os.environ.get("BRF_CLOUDS", "8"))
    lim = int(os.environ.get("BRF_LIMIT", "40"))
    pts, lens = synthetic.cloud_200k(B, seed=0)
    d = pts.cuda()
    res = {}
    for what, fn in (("limited", lambda: ext.radius_neighbors_limited(d, d, lens, lens, 0.0625, lim)),
                     ("bare", lambda: ext.radius_neighbors(d, d, lens, lens, 0.0625))):
        if os.environ.get("GR_RADIUS_Q2_STOP", "0") != "0" and what == "bare":
            continue
        for _ in range(3):
            out = fn()
        torch.cuda.synchronize()
        L.gr_timing_enable(1)
        L.gr_timing_reset()
        n = 20
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        r = {"ms": round(dt * 1e3, 4), "width": out.shape[1]}
        for name in ("radius_bin", "radius_count", "radius_fill", "radius_fused"):
            tot, cnt = ctypes.c_double(0), ctypes.c_int64(0)
            L.gr_timing_read(name.encode(), ctypes.byref(tot), ctypes.byref(cnt))
            if cnt.value:
                r[name[7:]] = round(tot.value / n, 4)
        L.gr_timing_enable(0)
        nq = d.shape[0]
        r["frac"] = round((24.0 * nq + 8.0 * nq * out.shape[1]) / dt / 1e9 / 8000.0, 4)
        res[what] = r
    print("RESULT " + json.dumps(res), flush=True)


def run(c):
    env = dict(os.environ, BRF_CHILD="1", **c)
    r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    print(json.dumps(c), line[0][7:] if line else ("FAILED " + r.stderr[-400:]), flush=True)


def main():
    if "--ablate" in sys.argv:
        for stop in ("1", "2", "3", "5", "0"):
            run({"GR_RADIUS_MODE": "2", "GR_RADIUS_Q2_STOP": stop})
        return
    for c in ({"GR_RADIUS_MODE": "0"}, {"GR_RADIUS_MODE": "1"}, {"GR_RADIUS_MODE": "2"}):
        run(c)


if __name__ == "__main__":
    child() if os.environ.get("BRF_CHILD") == "1" else main()
