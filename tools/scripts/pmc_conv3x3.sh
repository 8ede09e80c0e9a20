#!/bin/bash
# SQ / memory counters of K19 on one shape:  bash tools/scripts/pmc_conv3x3.sh 96 96 128 tag
I=${1:-96}; O=${2:-96}; H=${3:-128}; TAG=${4:-k19}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
RAW=/tmp/raw_$TAG; rm -rf $RAW; mkdir -p $RAW gpurun_out
pmc() { n=$1; shift; rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $RAW/$n -o run -- python3 tools/run_one_conv3x3.py $I $O $H 5 > gpurun_out/${TAG}_$n.log 2>&1; echo "$n rc=$?"; }
pmc a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pmc b SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CU_CYCLES
pmc c SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS
pmc d TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
python3 tools/pmc_table.py $RAW/a $RAW/b $RAW/c $RAW/d --match conv3x3_kernel > gpurun_out/${TAG}_pmc.md
cat gpurun_out/${TAG}_pmc.md
