#!/bin/bash
# round 4: K5 on weight images -- parity, model goldens, A/B bench
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4i
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_linear_x3_gpu.py -x -q > $O/pytest_x3.log 2>&1 || { tail -40 $O/pytest_x3.log; exit 1; }
tail -3 $O/pytest_x3.log
timeout -k 10 900 python -m pytest tests/test_blocks_gpu.py tests/test_configs_gpu.py tests/test_graph_gpu.py tests/test_plugin_gpu.py -x -q > $O/pytest_model.log 2>&1 || { tail -40 $O/pytest_model.log; exit 1; }
tail -3 $O/pytest_model.log
B="timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
$B > $O/v2.json 2> $O/v2.err
MLAGG_K5_V2=0 $B > $O/v1.json 2> $O/v1.err
$B --graph --no-roofline > $O/v2_graph.json 2> $O/v2_graph.err
for f in $O/*.json; do echo "$f: $(python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'], d['config']['ms_per_step_with_loss_readback'], d['config']['final_loss'])" 2>&1 | tail -1)"; done
