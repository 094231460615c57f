import os, sys, ctypes, torch
sys.path.insert(0, '/root/repo')
from gaussreg_amd import ext, synthetic, _lib
pts, lens = synthetic.cloud_200k(8, seed=0)
dp = pts.cuda()
L = _lib.lib()
def run(tag):
    for _ in range(3): ext.radius_neighbors(dp, dp, lens, lens, 0.0625)
    L.gr_timing_reset(); L.gr_timing_enable(1)
    for _ in range(10): ext.radius_neighbors(dp, dp, lens, lens, 0.0625)
    torch.cuda.synchronize()
    out = []
    for k in (b"radius_count", b"radius_fill"):
        t, c = ctypes.c_double(0), ctypes.c_int64(0)
        L.gr_timing_read(k, ctypes.byref(t), ctypes.byref(c))
        out.append(round(t.value / max(c.value, 1), 4))
    L.gr_timing_enable(0)
    print(tag, out, flush=True)
for d in sys.argv[1:]:
    os.environ["GR_RADIUS_DEBUG"] = d
    run("dbg=" + d)
