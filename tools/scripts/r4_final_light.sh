#!/bin/bash
# end-of-round evidence of the last commit: kernel trace of the default bench (timed region), whole-process kernel stats, the full bench line
TAG=${1:-round4_l}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; RAW=/tmp/raw_$TAG
rm -rf "$RAW"; mkdir -p "$O" "$RAW"
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o run -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_under_profiler.json 2> $O/trace.err
T=$(find $RAW/trace -name "*kernel_trace.csv" | head -1)
python3 tools/summarize_trace.py "$T" --steps 10 --top 80 --aten > $O/kernel_trace_timed_region.md 2>> $O/trace.err
cp $(find $RAW/trace -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_whole_process.csv
head -30 $O/kernel_trace_timed_region.md
timeout -k 10 400 python3 bench.py > $O/bench_final.json 2> $O/bench_final.err
tail -1 $O/bench_final.json | cut -c1-400
