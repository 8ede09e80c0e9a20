#!/bin/bash
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4l; mkdir -p $O
timeout -k 10 900 python -m pytest -m gpu tests/test_plugin_gpu.py tests/test_graph_gpu.py tests/test_configs_gpu.py tests/test_linear_x3_gpu.py tests/test_ddp_rehearsal_gpu.py tests/test_blocks_gpu.py tests/test_msmm_scan_gpu.py -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
bash tools/scripts/r4_ab.sh r4l "MLAGG_LEAF_STREAM=0" "MLAGG_LEAF_STREAM=1" "MLAGG_LEAF_STREAM=0" "MLAGG_LEAF_STREAM=1"
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --graph --no-roofline > $O/graph.json 2> $O/graph.err; tail -2 $O/graph.err; cut -c1-160 $O/graph.json
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/full.json 2> $O/full.err; python -c "
import json; d=json.load(open('$O/full.json')); print(d['value'], d['ms_per_step']); print(json.dumps(d['roofline']['mfma'], indent=0)[:1500])"
