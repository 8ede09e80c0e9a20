#!/bin/bash
# SQ / memory counters of the projection kernels on one shape:  bash tools/scripts/pmc_linear.sh M K N tag
M=${1:-163840}; K=${2:-96}; N=${3:-192}; TAG=${4:-lin}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for W in new r3; do
RAW=/tmp/raw_${TAG}_$W; rm -rf $RAW; mkdir -p $RAW
pmc() { n=$1; shift; rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $RAW/$n -o run -- python3 tools/run_one_linear.py $M $K $N $W 5 > gpurun_out/${TAG}_${W}_$n.log 2>&1; echo "$W $n rc=$?"; }
pmc a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pmc b SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CU_CYCLES
pmc c SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS
pmc d TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
pmc e SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SALU
python3 tools/pmc_table.py $RAW/a $RAW/b $RAW/c $RAW/d $RAW/e --match "linear_(x3|lp)_kernel" > gpurun_out/${TAG}_${W}_pmc.md
cat gpurun_out/${TAG}_${W}_pmc.md
done
