#!/bin/bash
# single-rank RCCL run: where the gradient buckets are enqueued inside backward:  bash tools/scripts/r4_ddp_timeline.sh TAG
TAG=${1:-ddp_tl}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; RAW=/tmp/raw_$TAG; rm -rf $RAW; mkdir -p $O $RAW
MLAGG_FORCE_DDP=1 rocprofv3 --kernel-trace --output-format csv -d $RAW/ddp -o run -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $O/ddp1_bench.json 2> $O/ddp1.err
T2=$(find $RAW/ddp -name "*kernel_trace.csv" | head -1)
python3 tools/ddp_bucket_timeline.py "$T2" --steps 3 > $O/ddp_bucket_timeline.md 2>> $O/ddp1.err
head -16 $O/ddp_bucket_timeline.md
