#!/bin/bash
# Per-phase instruction counts of the single-pass radius kernel (fused_kernel): one rocprofv3 --pmc pass per "stop after
# phase k" build switch (GR_RADIUS_FUSED_STOP); run on the GPU box via gpurun.
#   tools/radius_phase_counters.sh
which=fused
out=$GRAFT_REPO_ROOT/gpurun_out/phase_$which
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
var=GR_RADIUS_FUSED_STOP; stops="1 2 3 4 5 6 7 0"
for s in $stops; do
  env $var=$s GR_RADIUS_SINGLE_PASS=1 BRF_CHILD=1 BRF_MODE=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
      --output-format csv -d $out/s$s -o p -- python $GRAFT_REPO_ROOT/tools/bench_radius_fused.py > $out/s$s.log 2>&1
done
python3 - "$out" "$which" <<'PY'
import csv, glob, sys, collections
out, which = sys.argv[1:3]
kern = "fused_kernel" if which == "fused" else "q2_kernel"
prev = None
for d in sorted(glob.glob(out + "/s*/"), key=lambda p: (int(p.rstrip("/").split("s")[-1]) or 99)):
    acc = collections.defaultdict(float); n = collections.defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
    if not acc: print(d, "no data"); continue
    per = {k: v / len(n[k]) for k, v in acc.items()}
    q = 1.6e6
    row = {k.replace("SQ_INSTS_", "").lower(): round(per[k] / q * 64 / 64, 1) for k in per if k.startswith("SQ_INSTS")}   # wave-instructions per query
    print(d.rstrip("/").split("/")[-1], "per query:", row, "waves", int(per.get("SQ_WAVES", 0)))
PY
