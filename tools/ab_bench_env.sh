# Interleaved bench.py A/B of ONE environment switch on the probe library (tools/probe/libunigen_hip_probe.so), three runs per value on one box.
#   bash tools/ab_bench_env.sh UG_ATTN_M16 0 1 gpurun_out/out.log
set -e
VAR=$1; A=$2; Bv=$3; OUT=$4
export UG_LIB_PATH=$PWD/tools/probe/libunigen_hip_probe.so
: > $OUT
for i in 1 2 3; do
  for V in $A $Bv; do
    env $VAR=$V python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-scaling-base 2>/dev/null | python -c "
import sys, json
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$VAR=$V', round(l['value'],4), 'img/s  gemm', round(l['roofline']['achieved'],1), 'attn', round(l['roofline_attention']['achieved'],1), 'probe', round(l['mfma_probe_tflops']['shape_16x16x32'],0), 'W', l['power']['watts_median'], 'MHz', l['power']['sclk_mhz_median'])
" >> $OUT
  done
done
cat $OUT
