#!/bin/bash
# Kernel trace of the default bench, reduced to the timed region (full per-kernel table + the library tail):
#   bash tools/scripts/trace_step.sh TAG [bench args...]
TAG=${1:-trace}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; RAW=/tmp/raw_$TAG
rm -rf "$RAW"; mkdir -p "$O" "$RAW"
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o run -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $O/bench_under_profiler.json 2> $O/trace.err
echo "trace rc=$?"; tail -2 $O/trace.err
T=$(find $RAW/trace -name "*kernel_trace.csv" | head -1)
python3 tools/summarize_trace.py "$T" --steps 10 --top 120 --aten > $O/kernel_trace_timed_region.md 2>> $O/trace.err
head -34 $O/kernel_trace_timed_region.md
