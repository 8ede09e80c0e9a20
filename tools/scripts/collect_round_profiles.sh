#!/bin/bash
# Collect the per-round evidence on the GPU box: kernel trace of the default bench (timed region summary), SQ / TCC counter
# passes and FETCH_SIZE / WRITE_SIZE passes of the scan at config-2 shapes.  Raw rocprofv3 output is reduced on the box (the
# copy-back limit is 64 MiB); what remains under gpurun_out/$TAG is copied into profiles/ by hand.
#   bash tools/scripts/collect_round_profiles.sh round2_b
TAG=${1:-round}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG
RAW=/tmp/raw_$TAG
rm -rf "$O" "$RAW"; mkdir -p "$O" "$RAW"
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o run -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_under_profiler.json 2> $O/trace.err
echo "trace rc=$?"; tail -2 $O/trace.err
T=$(find $RAW/trace -name "*kernel_trace.csv" | head -1)
python3 tools/summarize_trace.py "$T" --steps 10 > $O/kernel_trace_timed_region.md 2>> $O/trace.err
cp $(find $RAW/trace -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_whole_process.csv
pmc() {  # name, counters...
    n=$1; shift
    rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $RAW/$n -o run -- python3 tools/bench_ops.py ${SCAN_OP:-msmm} --iters 5 > $O/$n.log 2>&1
    echo "$n rc=$?"
}
pmc pmc_sq1 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES
pmc pmc_sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
pmc pmc_FETCH_SIZE FETCH_SIZE
pmc pmc_WRITE_SIZE WRITE_SIZE
pmc pmc_tcc TCC_HIT_sum TCC_MISS_sum
pmc pmc_ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pmc pmc_req TCC_REQ_sum TCC_READ_sum
python3 tools/pmc_table.py $RAW/pmc_sq1 $RAW/pmc_sq2 $RAW/pmc_tcc $RAW/pmc_ea $RAW/pmc_req --match "selscan|tok_" > $O/pmc_selscan.md
python3 tools/pmc_to_json.py $RAW/pmc_FETCH_SIZE $RAW/pmc_WRITE_SIZE > $O/pmc_traffic.json
grep -h "per fwd+bwd" $O/*.log
ls -la $O
