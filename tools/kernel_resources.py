"""Register / LDS / spill figures of the kernels of one csrc file, from the compiler's resource remarks.
    python tools/kernel_resources.py radius_neighbors.hip [name regex]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gaussreg_amd", "csrc", sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else "."
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-c", src,
                    "-o", "/tmp/_kr.o", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, cwd="/tmp")
cur, rows = None, {}
for line in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = cur.replace("gr::(anonymous namespace)::", "").replace("void ", "")
        cur = re.sub(r"\(.*", "", cur)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|"
                  r"LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).split(" [")[0]] = int(m.group(2))
for k, v in rows.items():
    if re.search(pat, k):
        print(f"{k:48s}", " ".join(f"{a}={b}" for a, b in v.items()))
