#!/bin/bash
# rocprofv3 evidence for one round, summarised ON the GPU box (the raw counter CSVs are tens of MB each): kernel stats + the PMC passes
# MI355X_MICROARCH.md prescribes (separate --pmc passes, --kernel-trace only). usage: [BENCH_ARGS="--config cfg5"] bash tools/pmc_r02.sh TAG
set -o pipefail
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
S=/tmp/ug_prof_$TAG; rm -rf $S; mkdir -p $S gpurun_out profiles
B="python3 bench.py ${BENCH_ARGS:-} --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timer --no-scaling-base"
rocprofv3 --kernel-trace --stats --output-format csv -d $S/stats -- python3 bench.py ${BENCH_ARGS:-} --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-scaling-base > $S/stats.log 2>&1 || { tail -5 $S/stats.log; exit 1; }
cp $S/stats/*/*kernel_stats.csv profiles/${TAG}_bench_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d" " -f1)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $S/pmc_$n -- $B > $S/pmc_$n.log 2>&1 || { echo FAIL $n; tail -5 $S/pmc_$n.log; }
done
python3 tools/pmc_summary.py $TAG $S/pmc_FETCH_SIZE/*/*counter_collection.csv $S/pmc_WRITE_SIZE/*/*counter_collection.csv \
  $S/pmc_SQ_VALU_MFMA_BUSY_CYCLES/*/*counter_collection.csv $S/pmc_TCC_HIT_sum/*/*counter_collection.csv > gpurun_out/${TAG}_pmc_summary.log 2>&1
cp profiles/${TAG}_pmc.json profiles/${TAG}_bench_kernel_stats.csv gpurun_out/ 2>/dev/null
tail -60 gpurun_out/${TAG}_pmc_summary.log
