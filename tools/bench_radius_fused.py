"""Dev tool: the single-pass radius_search (gr_radius_search mode 1) against the two-pass path (mode 0), each in its own
subprocess.  8 x 200 k-point clouds, r = 0.0625, neighbor_limit 40 -- the `limited` workload of bench.py."""
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
    two = os.environ.get("BRF_BARE", "0") == "1"
    L.gr_radius_search_mode(int(os.environ.get("BRF_MODE", "1")))
    pts, lens = synthetic.cloud_200k(B, seed=0)
    d = pts.cuda()
    if two:
        fn = lambda: ext.radius_neighbors(d, d, lens, lens, 0.0625)  # noqa: E731
    else:
        fn = lambda: ext.radius_neighbors_limited(d, d, lens, lens, 0.0625, lim)  # noqa: E731
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
    res = {"ms": round(dt * 1e3, 4), "width": out.shape[1]}
    for name in ("radius_bin", "radius_count", "radius_fill", "radius_fused", "radius_tq", "radius_expand"):
        tot, cnt = ctypes.c_double(0), ctypes.c_int64(0)
        L.gr_timing_read(name.encode(), ctypes.byref(tot), ctypes.byref(cnt))
        if cnt.value:
            res[name] = round(tot.value / n, 4)
    nq = d.shape[0]
    res["end_to_end_frac"] = round((24.0 * nq + 8.0 * nq * out.shape[1]) / dt / 1e9 / 8000.0, 4)
    print("RESULT " + json.dumps(res), flush=True)


def main():
    cases = ({"BRF_MODE": "0"}, {"BRF_MODE": "1"}, {"BRF_MODE": "2"}, {"BRF_MODE": "0", "BRF_BARE": "1"}, {"BRF_MODE": "2", "BRF_BARE": "1"})
    if os.environ.get("BRF_ABLATE"):  # cumulative time of the thread-per-query kernel stopped after each phase
        cases = tuple({"BRF_MODE": "2", "TQ_STOP": str(k)} for k in (1, 2, 3, 4, 5, 0))
    for c in cases:
        env = dict(os.environ, BRF_CHILD="1", **c)
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        print(json.dumps(c), line[0][7:] if line else ("FAILED " + r.stderr[-400:]), flush=True)


if __name__ == "__main__":
    child() if os.environ.get("BRF_CHILD") == "1" else main()
