#!/bin/bash
# SQ counters of the radius search kernels in one mode (GPU box, via gpurun):  bash tools/radius_sq.sh <mode> [limit]
mode=${1:-2}; lim=${2:-40}
out=$GRAFT_REPO_ROOT/gpurun_out/sq_radius_m$mode
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE"; do
  n=$(echo $pass | md5sum | cut -c1-6)
  env BRF_MODE=$mode BRF_LIMIT=$lim BRF_CHILD=1 timeout 200 rocprofv3 --pmc $pass --output-format csv -d $out/$n -o p -- python $GRAFT_REPO_ROOT/tools/bench_radius_fused.py > $out/$n.log 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("gr::(anonymous namespace)::", "").split("(")[0][:40]
        if not any(t in k for t in ("tq_kernel", "fused_kernel", "traverse_kernel")): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for k in acc:
    per = {c: acc[k][c] / len(n[k][c]) for c in acc[k]}
    q = 1.6e6
    print(k, "launches", len(n[k]["SQ_WAVES"]) if "SQ_WAVES" in n[k] else "?")
    for c in sorted(per): print("   %-24s %14.0f   per query %8.2f" % (c, per[c], per[c] / q))
PY
