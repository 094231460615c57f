"""Print the last `n` kernels of a rocprofv3 --kernel-trace CSV with their queue, start, end (us): shows what overlaps when
frames run on two streams.  usage: tools/trace_streams.py <dir> [n]"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
queues = sorted({r["Queue_Id"] for r in rows})
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").replace("gr::(anonymous namespace)::", "").replace("gr::", "").split("(")[0][:40]
    q = queues.index(r["Queue_Id"])
    print(f"q{q} {'          ' * q}{name:40s} {(s - t0) / 1e3:8.1f} -> {(e - t0) / 1e3:8.1f}  ({(e - s) / 1e3:6.1f})")
