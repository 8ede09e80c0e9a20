#!/bin/bash
# parity subset + bench line after a model-level change:  bash tools/scripts/r4_check.sh TAG
TAG=${1:-r4chk}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 900 python -m pytest -m gpu tests/test_blocks_gpu.py tests/test_configs_gpu.py tests/test_gradient_tensors_gpu.py tests/test_plugin_gpu.py tests/test_graph_gpu.py tests/test_mixed_precision_gpu.py -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for i in 1 2; do
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_$i.json 2> $O/bench_$i.err
python -c "import json; d=json.load(open('$O/bench_$i.json')); print('bench', d['value'], d['ms_per_step'], d['config']['ms_per_step_with_loss_readback'])"
done
