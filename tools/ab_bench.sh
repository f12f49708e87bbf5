# Interleaved bench.py A/B of TWO BUILDS of the library on one box, three runs each: tools/probe/bin/libunigen_<name>.so, or "tree" for the in-tree build.
#   bash tools/ab_bench.sh base tree gpurun_out/out.log
set -e
A=$1; Bn=$2; OUT=$3
: > $OUT
for i in 1 2 3; do
  for L in $A $Bn; do
    if [ "$L" = "tree" ]; then unset UG_LIB_PATH; else export UG_LIB_PATH=$PWD/tools/probe/bin/libunigen_$L.so; fi
    python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-scaling-base 2>/dev/null | python -c "
import sys, json
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$L', round(l['value'],4), 'img/s  gemm', round(l['roofline']['achieved'],1), 'attn', round(l['roofline_attention']['achieved'],1), 'probe', round(l['mfma_probe_tflops']['shape_16x16x32'],0), 'W', l['power']['watts_median'], 'MHz', l['power']['sclk_mhz_median'])
" >> $OUT
  done
done
cat $OUT
