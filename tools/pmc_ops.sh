#!/bin/bash
# usage (GPU box, via gpurun): bash tools/pmc_ops.sh r05 [op ...]
# FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (no tracing flags), one process per operator and counter;
# tools/pmc_ops_summary.py turns them into gpurun_out/<R>_pmc_ops.json (copy to profiles/).
R=${1:-r05}; shift
OPS=${@:-grid_ref_64x200k grid_cell_64x200k grid_ref_1x200k radius_8x200k radius_limited_8x200k kpconv_2_2_x32 gs_fuse_2x2p55M fps_2x200k}
cd /tmp && export TMPDIR=/tmp
for op in $OPS; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    out=$GRAFT_REPO_ROOT/gpurun_out/pmcops_${op}_${ctr}
    rm -rf $out; mkdir -p $out
    timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $out -o p -- python $GRAFT_REPO_ROOT/tools/pmc_ops.py $op 3 > $out/stdout.log 2>&1
  done
done
cd $GRAFT_REPO_ROOT
python tools/pmc_ops_summary.py gpurun_out/${R}_pmc_ops.json $OPS
rm -f gpurun_out/pmcops_*/*.csv gpurun_out/pmcops_*/*/*.csv
