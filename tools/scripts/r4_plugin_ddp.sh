#!/bin/bash
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4n; mkdir -p $O
timeout -k 10 900 python -m pytest -m gpu tests/test_plugin_gpu.py tests/test_ddp_rehearsal_gpu.py tests/test_graph_gpu.py -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
RAW=/tmp/raw_r4n; rm -rf $RAW; mkdir -p $RAW
MLAGG_FORCE_DDP=1 rocprofv3 --kernel-trace --output-format csv -d $RAW/ddp -o run -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $O/ddp1_bench.json 2> $O/ddp1.err
T2=$(find $RAW/ddp -name "*kernel_trace.csv" | head -1)
python3 tools/ddp_bucket_timeline.py "$T2" --steps 2 > $O/ddp_bucket_timeline.md 2>> $O/ddp1.err
head -20 $O/ddp_bucket_timeline.md
MLAGG_FORCE_DDP=1 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/ddp1_plain.json 2>/dev/null; cut -c1-200 $O/ddp1_plain.json
