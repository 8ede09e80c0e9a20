#!/bin/bash
# 16-bit modes: parity of the one-product K18 / K19 forms, then same-box bench lines of configs 3 (bf16) and 5 (fp16) against their fp32 runs
# and against the round-3 form (MLAGG_LP_K=0):  bash tools/scripts/r4_lp.sh TAG
TAG=${1:-r4lp}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 900 python -m pytest -m gpu tests/test_conv_wgrad_gpu.py tests/test_mixed_precision_gpu.py tests/test_configs_gpu.py -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
run() {  # name, env, args
  env $2 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline $3 > $O/$1.json 2> $O/$1.err
  echo "[$1] $(python -c "import json; d=json.load(open('$O/$1.json')); print(d['value'], d['ms_per_step'], d['dtype'])" 2>&1 | tail -1)"
}
run c3_bf16_lpk1 MLAGG_LP_K=1 "--config 3" &&
run c3_bf16_lpk0 MLAGG_LP_K=0 "--config 3" &&
run c3_fp32 MLAGG_LP_K=1 "--config 3 --precision fp32" &&
run c5_fp16_lpk1 MLAGG_LP_K=1 "--config 5" &&
run c5_fp16_lpk0 MLAGG_LP_K=0 "--config 5" &&
run c5_fp32 MLAGG_LP_K=1 "--config 5 --precision fp32" &&
run c3_bf16_lpk1_again MLAGG_LP_K=1 "--config 3"
