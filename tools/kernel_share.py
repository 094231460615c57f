"""Shares of a rocprofv3 *_kernel_stats.csv by origin: this library's kernels (gr::), rocBLAS / hipBLASLt (Cijk_...), stock ATen.
usage: python tools/kernel_share.py <kernel_stats.csv>"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)


def share(pred):
    sel = [r for r in rows if pred(r["Name"])]
    return round(sum(float(r["TotalDurationNs"]) for r in sel) / tot, 4), sum(int(r["Calls"]) for r in sel)


blas = lambda n: "Cijk" in n or "rocblas" in n.lower()  # noqa: E731
print("total kernel ms", round(tot / 1e6, 3))
print("gr::         share, launches:", share(lambda n: "gr::" in n))
print("rocBLAS      share, launches:", share(blas))
print("at:: (ATen)  share, launches:", share(lambda n: "at::" in n and "gr::" not in n))
print("other        share, launches:", share(lambda n: "gr::" not in n and "at::" not in n and not blas(n)))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    n = re.sub(r"gr::\(anonymous namespace\)::|at::native::|\(anonymous namespace\)::|void ", "", r["Name"])[:90]
    print(f'  {float(r["TotalDurationNs"]) / tot * 100:5.1f} %  {r["Calls"]:>6} x  {n}')
