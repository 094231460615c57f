# Collects everything under profiles/ for one round (GPU box, via gpurun):  bash tools/prof_all.sh r03
# bench.py with its extras or pairs section dies in rocprofv3's exit handler (after the JSON line is out), so the kernel
# stats come in parts:
#   <R>_bench_kernel_stats.csv   raster (32 views per launch on the default path -- no one-camera section, no static-scene pass:
#                                every rasterizer row is a V = 32 row of one call at a time)
#                                + radius (bare, limited in both modes) from bench.py
#   <R>_single_view_kernel_stats.csv   the one-camera loop alone (tools/single_view_loop.py): V = 1 rows
#   <R>_extras_kernel_stats.csv  everything else through tools/bench_extras.py
# The counter passes use the same command as the first CSV.
R=${1:-r03}
cd $GRAFT_REPO_ROOT
B="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --pairs 0 --no-single-view --no-static-scene"
bash tools/prof.sh bench $GRAFT_REPO_ROOT/$B > /dev/null 2>&1
bash tools/prof.sh sv $GRAFT_REPO_ROOT/tools/single_view_loop.py > /dev/null 2>&1
bash tools/prof.sh extras $GRAFT_REPO_ROOT/tools/bench_extras.py > /dev/null 2>&1
bash tools/pmc.sh fetch FETCH_SIZE $GRAFT_REPO_ROOT/$B > /dev/null 2>&1
bash tools/pmc.sh write WRITE_SIZE $GRAFT_REPO_ROOT/$B > /dev/null 2>&1
bash tools/pmc.sh sq1 "SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES" $GRAFT_REPO_ROOT/$B > /dev/null 2>&1
bash tools/pmc.sh sq2 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" $GRAFT_REPO_ROOT/$B > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
ls gpurun_out/prof_bench gpurun_out/prof_sv gpurun_out/prof_extras gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq1 gpurun_out/pmc_sq2
python tools/pmc_summary.py gpurun_out/pmc_fetch/fetch_counter_collection.csv gpurun_out/pmc_write/write_counter_collection.csv gpurun_out/${R}_pmc_hbm.json
python tools/sq_summary.py gpurun_out/${R}_sq_counters.json sq1=gpurun_out/pmc_sq1/sq1_counter_collection.csv sq2=gpurun_out/pmc_sq2/sq2_counter_collection.csv
cp gpurun_out/prof_bench/bench_kernel_stats.csv gpurun_out/${R}_bench_kernel_stats.csv
cp gpurun_out/prof_sv/sv_kernel_stats.csv gpurun_out/${R}_single_view_kernel_stats.csv
cp gpurun_out/prof_extras/extras_kernel_stats.csv gpurun_out/${R}_extras_kernel_stats.csv
python tools/trace_streams.py gpurun_out/prof_sv 64 > gpurun_out/${R}_single_view_timeline.txt 2>&1   # two frames in flight: queue, start -> end per kernel
rm -f gpurun_out/pmc_*/*.csv gpurun_out/prof_*/*trace*.csv
