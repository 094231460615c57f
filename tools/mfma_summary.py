"""Summarise a rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE; tools/pmc.sh over
tools/bench_mfma.py) into profiles/<round>_mfma.json: per kernel and launch shape, MFMA pipe utilisation =
SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES).
usage: python tools/mfma_summary.py <counter_collection.csv> <out.json> [<stdout of the un-profiled bench_mfma run>]"""
import collections
import csv
import json
import sys

KEEP = ("pairwise_kernel", "pairwise_big_kernel", "geo_embedding", "gemm_nn", "kpconv", "rpe_")


def main():
    path, out = sys.argv[1:3]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("gr::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if not any(t in k for t in KEEP):
            continue
        key = (k, r.get("Grid_Size", ""), r.get("Workgroup_Size", ""))
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        n[key].add(r["Dispatch_Id"])
    rows = []
    for key, c in sorted(acc.items()):
        ln = max(len(n[key]), 1)
        busy, cu = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / ln, c.get("SQ_BUSY_CU_CYCLES", 0.0) / ln
        rows.append({"kernel": key[0], "grid_size": key[1], "workgroup_size": key[2], "launches": ln,
                     "mfma_busy_cycles": round(busy), "busy_cu_cycles": round(cu), "mfma_util": round(busy / (4.0 * cu), 4) if cu else None})
    doc = {"note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -- python tools/bench_mfma.py; mfma_util = "
                   "SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES), per launch shape; peak = 157.3 TFLOP/s dense fp32 MFMA "
                   "(v_mfma_f32_32x32x2_f32).  `unprofiled` = the TFLOP/s lines of the same script run without the profiler.",
           "kernels": rows}
    if len(sys.argv) > 3:
        doc["unprofiled"] = [l.strip() for l in open(sys.argv[3]) if "TFLOP/s" in l]
    json.dump(doc, open(out, "w"), indent=1)
    for r in rows:
        print(r)


if __name__ == "__main__":
    main()
