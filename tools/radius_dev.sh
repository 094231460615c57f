#!/bin/bash
# Dev loop of the single-pass radius kernels (GPU box, via gpurun): parity tests in all modes, then timings per mode.
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_radius_search.py tests/test_gpu_ext.py -x -q 2>&1 | tail -6
for cfg in "BRF_MODE=0" "BRF_MODE=1" "BRF_MODE=2" "BRF_MODE=2 BRF_LIMIT=46"; do
  echo "$cfg: $(env $cfg BRF_CHILD=1 timeout 120 python tools/bench_radius_fused.py 2>&1 | grep RESULT)"
done
