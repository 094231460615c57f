"""gpurun_out/pmcops_<op>_{FETCH_SIZE,WRITE_SIZE}/ (tools/pmc_ops.sh) -> one JSON: per operator the HBM traffic per call
(2 x FETCH_SIZE + WRITE_SIZE, KB -> bytes, MI355X_MICROARCH.md HBM section), its ratio to the algorithmic bytes, and the
per-kernel rows behind it.    usage: python tools/pmc_ops_summary.py out.json op [op ...]"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def short(name):
    name = name.replace("gr::(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:90]


def collect(op, ctr):
    d = os.path.join(ROOT, "gpurun_out", f"pmcops_{op}_{ctr}")
    info = None
    try:
        for ln in open(os.path.join(d, "stdout.log")):
            if ln.startswith("PMC_OP "):
                info = json.loads(ln[7:])
    except OSError:
        pass
    acc = collections.defaultdict(float)
    n = collections.defaultdict(set)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != ctr:
                continue
            k = short(r["Kernel_Name"])
            acc[k] += float(r["Counter_Value"])
            n[k].add(r["Dispatch_Id"])
    return info, {k: (acc[k], len(n[k])) for k in acc}


def main():
    out, ops = sys.argv[1], sys.argv[2:]
    doc = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, one process per operator (tools/pmc_ops.sh, "
                   "tools/pmc_ops.py <op> 3); hbm bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 summed over EVERY kernel of the process "
                   "(gr:: and ATen alike), divided by the operator calls; `kernels` = KB per call by kernel",
           "ops": {}}
    for op in ops:
        fi, f = collect(op, "FETCH_SIZE")
        wi, w = collect(op, "WRITE_SIZE")
        info = fi or wi
        if info is None:
            doc["ops"][op] = {"error": "no PMC_OP line (see gpurun_out/pmcops_%s_*/stdout.log)" % op}
            continue
        calls = info["calls"]
        rows = []
        for k in sorted(set(f) | set(w), key=lambda k: -(2 * f.get(k, (0, 0))[0] + w.get(k, (0, 0))[0])):
            rows.append({"kernel": k, "launches_per_call": round(max(f.get(k, (0, 0))[1], w.get(k, (0, 0))[1]) / calls, 2),
                         "fetch_kb_per_call": round(f.get(k, (0, 0))[0] / calls, 1), "write_kb_per_call": round(w.get(k, (0, 0))[0] / calls, 1)})
        skip = ()
        if op.startswith("kpconv"):  # the process also builds the pyramid once: only the KPConv kernels count
            rows_op = [r for r in rows if any(t in r["kernel"] for t in ("kp_gather", "gemm_nn", "rowflag"))]
        else:
            rows_op = rows
        hbm = sum(2 * r["fetch_kb_per_call"] + r["write_kb_per_call"] for r in rows_op) * 1024.0
        e = dict(info)
        e["hbm_bytes_per_call"] = int(hbm)
        e["traffic_ratio"] = round(hbm / info["algorithmic_bytes_per_call"], 3)
        e["kernels"] = rows_op[:24]
        doc["ops"][op] = e
        print(op, "hbm/call %.1f MB" % (hbm / 1e6), "algorithmic %.1f MB" % (info["algorithmic_bytes_per_call"] / 1e6), "ratio", e["traffic_ratio"])
    json.dump(doc, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
