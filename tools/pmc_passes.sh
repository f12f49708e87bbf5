#!/bin/bash
# usage: bash tools/pmc_passes.sh TAG FILTER "<python3 script and args>" "<counters of pass 1>" ["<counters of pass 2>" ...]
# One rocprofv3 --kernel-trace --pmc pass per counter group (the program goes straight after --), then tools/pmc_kernels.py over all of them.
set -o pipefail
TAG=$1; FILTER=$2; CMD=$3; shift 3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
S=/tmp/ug_pmc_$TAG; rm -rf $S; mkdir -p $S gpurun_out
i=0; CSVS=""
for c in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $S/p$i -- $CMD > $S/p$i.log 2>&1 || { echo "FAIL pass $i ($c)"; tail -5 $S/p$i.log; continue; }
  CSVS="$CSVS $(ls $S/p$i/*/*counter_collection.csv)"
done
python3 tools/pmc_kernels.py "$FILTER" $CSVS > gpurun_out/${TAG}_pmc_kernels.log 2>&1
tail -120 gpurun_out/${TAG}_pmc_kernels.log
