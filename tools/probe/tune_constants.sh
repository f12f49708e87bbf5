set -e
bash tools/ab_bench_env.sh UG_GEMM_GROUP_M 4 8 gpurun_out/r05v_tune_groupm.log
bash tools/ab_bench_env.sh UG_GEMM_E128_PCT 60 75 gpurun_out/r05v_tune_e128.log
bash tools/ab_bench_env.sh UG_GEMM_SPLITK_MIN_KT 96 48 gpurun_out/r05v_tune_splitk.log
