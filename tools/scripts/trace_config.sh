#!/bin/bash
# kernel trace of one bench configuration, reduced to the timed region:  bash tools/scripts/trace_config.sh 3 round2_c_config3
CFG=${1:-3}; TAG=${2:-trace_config$CFG}; STEPS=${3:-5}; WARM=${4:-3}; MARK=${5:-selscan_bwd_(group_)?kernel}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
RAW=/tmp/raw_$TAG; rm -rf $RAW; mkdir -p $RAW gpurun_out
rocprofv3 --kernel-trace --output-format csv -d $RAW -o run -- python3 bench.py --config $CFG --steps $STEPS --warmup $WARM --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}.err
python3 tools/summarize_trace.py "$(find $RAW -name '*kernel_trace.csv' | head -1)" --steps $STEPS --mark "$MARK" --aten > gpurun_out/${TAG}_kernel_trace_timed_region.md
head -30 gpurun_out/${TAG}_kernel_trace_timed_region.md
