#!/bin/bash
# round 4 evidence: kernel trace of the default bench, data-parallel bucket timeline, PMC of the scan, attention and depthwise kernels
TAG=${1:-round4_f}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; RAW=/tmp/raw_$TAG
rm -rf "$RAW"; mkdir -p "$O" "$RAW"
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o run -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_under_profiler.json 2> $O/trace.err
T=$(find $RAW/trace -name "*kernel_trace.csv" | head -1)
python3 tools/summarize_trace.py "$T" --steps 10 --top 80 --aten > $O/kernel_trace_timed_region.md 2>> $O/trace.err
cp $(find $RAW/trace -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_whole_process.csv
head -32 $O/kernel_trace_timed_region.md
# single-rank RCCL run: where the gradient buckets are enqueued inside backward
MLAGG_FORCE_DDP=1 rocprofv3 --kernel-trace --output-format csv -d $RAW/ddp -o run -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $O/ddp1_bench.json 2> $O/ddp1.err
T2=$(find $RAW/ddp -name "*kernel_trace.csv" | head -1)
python3 tools/ddp_bucket_timeline.py "$T2" --steps 3 > $O/ddp_bucket_timeline.md 2>> $O/ddp1.err
head -30 $O/ddp_bucket_timeline.md
pmc() {  # out-name, bench_ops op, counters...
    n=$1; op=$2; shift; shift
    rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $RAW/$n -o run -- python3 tools/bench_ops.py $op --iters 5 > $O/$n.log 2>&1
    echo "$n rc=$?"
}
for op in local pooled dwconv; do
  pmc pmc_${op}_a $op SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
  pmc pmc_${op}_b $op SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
  pmc pmc_${op}_c $op SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU
  pmc pmc_${op}_f $op FETCH_SIZE
  pmc pmc_${op}_w $op WRITE_SIZE
  python3 tools/pmc_table.py $RAW/pmc_${op}_a $RAW/pmc_${op}_b $RAW/pmc_${op}_c $RAW/pmc_${op}_f $RAW/pmc_${op}_w --match "local_attn|pooled_attn|dwconv" > $O/pmc_${op}.md
done
cat $O/pmc_local.md $O/pmc_pooled.md $O/pmc_dwconv.md | cut -c1-400
