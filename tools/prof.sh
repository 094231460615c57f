#!/bin/bash
# usage: tools/prof.sh <name> <python script + args...>   (run on the GPU box via gpurun)
# writes gpurun_out/prof_<name>/<name>_kernel_stats.csv
name=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $name -- python "$@" > $out/stdout.log 2>&1
ls $out
head -25 $out/${name}_kernel_stats.csv 2>/dev/null | cut -c1-200
