#!/bin/bash
# A/B bench lines of environment switches:  bash tools/scripts/r4_ab.sh TAG "VAR=val VAR2=val" "..." ...
TAG=$1; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/$TAG; mkdir -p $O
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/ab_$i.json 2> $O/ab_$i.err
  echo "[$cfg] $(python -c "import json; d=json.load(open('$O/ab_$i.json')); print(d['value'], d['ms_per_step'], d['config']['ms_per_step_with_loss_readback'])" 2>&1 | tail -1)"
done
