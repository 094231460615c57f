for sl in 28 26 24 22; do GR_RADIUS_FUSED_SLOTS=$sl BRF_CHILD=1 BRF_MODE=1 python tools/bench_radius_fused.py 2>&1 | grep RESULT | sed "s/^/slots $sl /"; done
