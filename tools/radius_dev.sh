#!/bin/bash
# Dev loop of the single-pass radius kernel (GPU box, via gpurun): parity tests in both modes, then timings per configuration.
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_radius_search.py tests/test_gpu_ext.py -x -q 2>&1 | tail -4
for cfg in "BRF_MODE=0" "BRF_MODE=1 GR_RADIUS_FUSED_SLOTS=26" "BRF_MODE=1 GR_RADIUS_FUSED_SLOTS=28" "BRF_MODE=1 GR_RADIUS_FUSED_SLOTS=24"; do
  echo "$cfg: $(env $cfg BRF_CHILD=1 python tools/bench_radius_fused.py 2>&1 | grep RESULT)"
done
if [ "$1" == "phases" ]; then bash tools/radius_phase_counters.sh 2>&1 | tail -9; fi
