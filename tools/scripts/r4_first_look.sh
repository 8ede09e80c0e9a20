#!/bin/bash
# round 4, first GPU call: eager vs hipGraph replay (with the capturable K11), K5 / K5w row thresholds, a kernel trace of the replay
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4a
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_blocks_gpu.py tests/test_graph_gpu.py -x -q -k "adamw or graph" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
B="timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
$B > $O/eager.json 2> $O/eager.err && tail -c 600 $O/eager.err
$B --graph --no-roofline > $O/graph.json 2> $O/graph.err
MLAGG_K5_MIN_ROWS=10240 $B --no-roofline > $O/eager_k5_10240.json 2>/dev/null
MLAGG_K5_MIN_ROWS=2560 $B --no-roofline > $O/eager_k5_2560.json 2>/dev/null
MLAGG_K5_MIN_ROWS=2560 MLAGG_WGRAD_MIN_ROWS=2560 $B --no-roofline > $O/eager_k5_2560_w2560.json 2>/dev/null
MLAGG_K5_MIN_ROWS=2560 MLAGG_WGRAD_MIN_ROWS=2560 $B --no-roofline --graph > $O/graph_k5_2560_w2560.json 2>/dev/null
$B --config 3 --precision fp32 --no-roofline > $O/c3_fp32_eager.json 2>/dev/null
$B --config 3 --precision fp32 --no-roofline --graph > $O/c3_fp32_graph.json 2>/dev/null
$B --config 3 --no-roofline > $O/c3_bf16_eager.json 2>/dev/null
$B --config 3 --no-roofline --graph > $O/c3_bf16_graph.json 2>/dev/null
for f in $O/*.json; do echo "$f: $(python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'], d['config'].get('ms_per_step_with_loss_readback'), d['config']['launch'])" 2>&1 | tail -1)"; done
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_graph -o g -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --graph --no-roofline --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_graph.json 2> $GRAFT_REPO_ROOT/$O/prof_graph.err
cd $GRAFT_REPO_ROOT
ls -R $O/prof_graph | head -20
T=$(find $O/prof_graph -name "*kernel_trace.csv" | head -1)
python tools/summarize_trace.py $T --steps 10 --top 60 > $O/graph_trace_summary.md 2>&1
head -40 $O/graph_trace_summary.md
# the csv is large: keep only the summary
find $O/prof_graph -name "*.csv" -size +20M -delete
