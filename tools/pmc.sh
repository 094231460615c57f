#!/bin/bash
# usage: tools/pmc.sh <name> <counter list> <python script + args...>   (GPU box, via gpurun)
# One --pmc pass (no tracing flags: gpurun refuses pmc + sys/runtime trace combos).
name=$1; shift
ctr=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctr --output-format csv -d $out -o $name -- python "$@" > $out/stdout.log 2>&1
ls $out | head
