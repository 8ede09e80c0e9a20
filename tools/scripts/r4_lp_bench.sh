#!/bin/bash
# 16-bit modes, same-box bench lines (hipGraph replay: configs 3 / 5 are host-bound in eager mode):  bash tools/scripts/r4_lp_bench.sh TAG [extra bench args]
TAG=${1:-r4lpb}; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/$TAG; mkdir -p $O
run() {  # name, env, args
  env $2 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --graph $3 > $O/$1.json 2> $O/$1.err
  echo "[$1] $(python -c "import json; d=json.load(open('$O/$1.json')); print(d['value'], d['ms_per_step'], d['dtype'])" 2>&1 | tail -1)"
}
run c3_bf16_lpk1 MLAGG_LP_K=1 "--config 3 $*" &&
run c3_bf16_lpk0 MLAGG_LP_K=0 "--config 3 $*" &&
run c3_fp32 MLAGG_LP_K=1 "--config 3 --precision fp32 $*" &&
run c5_fp16_lpk1 MLAGG_LP_K=1 "--config 5 $*" &&
run c5_fp16_lpk0 MLAGG_LP_K=0 "--config 5 $*" &&
run c5_fp32 MLAGG_LP_K=1 "--config 5 --precision fp32 $*" &&
run c3_bf16_lpk1_again MLAGG_LP_K=1 "--config 3 $*"
