"""Dev tool: the dense distance kernel (pairwise_kernel 64 x 64 / pairwise_big_kernel 128 x 128 tiles, v_mfma_f32_32x32x2_f32)
in TFLOP/s against the 157.3 TFLOP/s fp32 matrix peak, (tile size chosen by the library: one
interpreter per configuration), plus the batched SuperPointMatching launch at the configs[4] shape (64 pairs x 767 superpoints).
The two tile sizes must return the same bits; both are checked against torch in fp64."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
PEAK = 157.3


def child():
    import hashlib
    import torch
    from gaussreg_amd import ops, _lib
    L = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(0)
    res = {}

    def timeit(fn, n=10, warm=3):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n
    for b, n, c, norm in ((1, 767, 256, False), (1, 2048, 256, False), (1, 4096, 256, False), (1, 8192, 256, False), (1, 8192, 256, True),
                          (64, 767, 256, True), (3, 1000, 37, False)):
        x = torch.randn(b, n, c, device="cuda", generator=g)
        y = torch.randn(b, n + 5, c, device="cuda", generator=g)
        if norm:
            x, y = torch.nn.functional.normalize(x, dim=-1), torch.nn.functional.normalize(y, dim=-1)
        out = ops.pairwise_distance(x, y, normalized=norm)
        t = timeit(lambda: ops.pairwise_distance(x, y, normalized=norm))
        import ctypes
        L.gr_timing_enable(1)
        L.gr_timing_reset()
        for _ in range(10):
            ops.pairwise_distance(x, y, normalized=norm)
        torch.cuda.synchronize()
        tot, cnt = ctypes.c_double(0), ctypes.c_int64(0)
        L.gr_timing_read(b"pairwise_distance", ctypes.byref(tot), ctypes.byref(cnt))
        L.gr_timing_enable(0)
        tk = tot.value / max(cnt.value, 1) / 1e3   # seconds per launch of the distance kernel alone
        want = (torch.cdist(x[:1].double(), y[:1].double()) ** 2).float() if n <= 4096 else None
        err = None if want is None else float((out[:1] - want).abs().max() / want.abs().max())
        flop = 2.0 * b * n * (n + 5) * c
        res[f"{b}x{n}x{n + 5}x{c}{'n' if norm else ''}"] = {"ms": round(t * 1e3, 4), "tflops": round(flop / t / 1e12, 1),
                                                           "frac": round(flop / t / 1e12 / PEAK, 3), "kernel_ms": round(tk * 1e3, 4),
                                                           "kernel_tflops": round(flop / tk / 1e12, 1) if tk > 0 else None, "rel_err": err,
                                                           "sha": hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]}
    print("RESULT " + json.dumps(res), flush=True)


def main():
    env = dict(os.environ, BP_CHILD="1")
    r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    if not line:
        print("FAILED", r.stderr[-600:])
        return
    for k, v in json.loads(line[0][7:]).items():
        print("   ", k, v)


if __name__ == "__main__":
    child() if os.environ.get("BP_CHILD") == "1" else main()
