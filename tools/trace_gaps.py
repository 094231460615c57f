"""Print the GPU timeline (kernel, start offset, duration, gap to the previous kernel) of the last iteration found in a
rocprofv3 --kernel-trace CSV."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else "bbox_init"
starts = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a = starts[-2]; b = starts[-1]
t0 = int(rows[a]["Start_Timestamp"]); prev_end = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").replace("gr::(anonymous namespace)::", "").replace("gr::", "").split("(")[0][:46]
    print(f"{name:46s} start {(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:7.1f}")
    prev_end = e
print("iteration total", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3, "us")
