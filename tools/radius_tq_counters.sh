#!/bin/bash
# usage (GPU box, via gpurun): tools/radius_tq_counters.sh <outdir>
# SQ counters of the thread-per-query radius kernels (two --pmc passes), bare + limited loops in mode 2.
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-tq_ctr}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export GR_RADIUS_SINGLE_PASS=2 BRF_MODE=2
for pass in "SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SMEM"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --output-format csv -d $out/$tag.bare -o p -- python $GRAFT_REPO_ROOT/tools/radius_loop.py > $out/$tag.bare.log 2>&1
  rocprofv3 --pmc $pass --output-format csv -d $out/$tag.lim -o p -- python $GRAFT_REPO_ROOT/tools/radius_limited_loop.py > $out/$tag.lim.log 2>&1
done
cd $GRAFT_REPO_ROOT
GR_SQ_KEEP=tq_kernel,tq_expand,traverse_kernel,fine_kernel python tools/sq_summary.py $out/summary.json $(for f in $(find $out -name "*counter_collection.csv"); do echo x=$f; done)
python - <<PY
import json
d=json.load(open("$out/summary.json"))
for k in d["kernels"]:
    p=k["per_launch"]
    print(k["kernel"][:60], "launches",k["launches"],"valu/wave",k.get("valu_insts_per_wave"),"valu_us",k.get("valu_issue_us_at_2p4GHz"),"dur",k.get("profiled_duration_us_avg"),
      {c:p[c] for c in ("SQ_WAVES","SQ_INSTS_LDS","SQ_INSTS_VMEM_RD","SQ_INSTS_VMEM_WR","SQ_WAVE_CYCLES","SQ_BUSY_CYCLES","SQ_WAIT_INST_ANY") if c in p})
PY
