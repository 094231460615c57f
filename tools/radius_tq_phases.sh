#!/bin/bash
cd $GRAFT_REPO_ROOT
for st in 1 2 3 4 5 6 0; do echo "stop $st: $(env TQ_STOP=$st BRF_MODE=2 BRF_CHILD=1 timeout 120 python tools/bench_radius_fused.py 2>&1 | grep RESULT | grep -o '"radius_tq": [0-9.]*')"; done
