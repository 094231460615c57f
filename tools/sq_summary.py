"""Summarise rocprofv3 --pmc SQ passes (tools/pmc.sh) into profiles/<round>_sq_counters.json: per kernel, per launch,
VALU-issue time vs wall time.  usage: python tools/sq_summary.py out.json name=csv [name=csv ...]"""
import collections
import csv
import json
import sys

KEEP = tuple(__import__("os").environ.get("GR_SQ_KEEP", "blend_kernel,preprocess_kernel,traverse_kernel,fused_kernel,tq_kernel,tq_expand_kernel,coarse_kernel,fine_kernel,tile_scatter_kernel,tile_count_kernel,ds_scatter_kernel,ds_count_kernel,rpe_attention_kernel,fps_multi_kernel").split(","))
SIMDS, GHZ = 1024, 2.4


def main():
    out = sys.argv[1]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(lambda: collections.defaultdict(set))
    dur = collections.defaultdict(list)
    for spec in sys.argv[2:]:
        _, path = spec.split("=")
        seen = set()
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].replace("gr::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            if not any(t in k for t in KEEP):
                continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k][r["Counter_Name"]].add(r["Dispatch_Id"])
            if (k, r["Dispatch_Id"]) not in seen:
                seen.add((k, r["Dispatch_Id"]))
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    kernels = []
    for k, d in sorted(acc.items()):
        per = {c: v / max(len(disp[k][c]), 1) for c, v in d.items()}
        row = {"kernel": k, "launches": max(len(s) for s in disp[k].values()), "per_launch": {c: round(v) for c, v in per.items()}}
        if "SQ_INSTS_VALU" in per and "SQ_WAVES" in per:
            row["valu_insts_per_wave"] = round(per["SQ_INSTS_VALU"] / per["SQ_WAVES"], 1)
        if "SQ_INSTS_VALU" in per:
            # a wave64 VALU instruction occupies its SIMD for 4 cycles (SQ_ACTIVE_INST_VALU == SQ_INSTS_VALU quad-cycles)
            row["valu_issue_us_at_2p4GHz"] = round(per["SQ_INSTS_VALU"] * 4 / SIMDS / (GHZ * 1e3), 1)
        if "SQ_WAVE_CYCLES" in per:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if c in per:
                    row[c.lower() + "_frac_of_wave_cycles"] = round(per[c] / per["SQ_WAVE_CYCLES"], 3)
        row["profiled_duration_us_avg"] = round(sum(dur[k]) / len(dur[k]), 1)
        kernels.append(row)
    doc = {"note": "rocprofv3 --pmc (SQ counters, two passes of 8, tools/pmc.sh) over python bench.py --steps 3 --warmup 1 "
                   "--no-cpu-baseline.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md). "
                   "valu_issue_us = SQ_INSTS_VALU x 4 cycles / 1024 SIMDs / 2.4 GHz: the time the VALU pipes alone need; compare "
                   "with the kernel's duration (profiled_duration_us_avg, or profiles/*_bench_kernel_stats.csv).",
           "kernels": kernels}
    json.dump(doc, open(out, "w"), indent=1)
    for r in kernels:
        print(r["kernel"], r.get("valu_insts_per_wave"), r.get("valu_issue_us_at_2p4GHz"), r["profiled_duration_us_avg"])


if __name__ == "__main__":
    main()
