#!/bin/bash
# SQ / memory counters of the 3 x 3 weight-gradient kernel on one shape:  bash tools/scripts/pmc_conv_wgrad.sh I O H form tag
I=${1:-48}; O=${2:-48}; H=${3:-256}; FORM=${4:-3}; TAG=${5:-wg}; KIND=${6:-wgrad}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
RAW=/tmp/raw_${TAG}; rm -rf $RAW; mkdir -p $RAW
pmc() { n=$1; shift; rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $RAW/$n -o run -- python3 tools/run_one_conv_wgrad.py $I $O $H $FORM 5 10 $KIND > gpurun_out/${TAG}_$n.log 2>&1; echo "$n rc=$?"; }
pmc a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pmc b SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CU_CYCLES
pmc c SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU
pmc d TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
pmc f FETCH_SIZE
pmc w WRITE_SIZE
pmc g GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_MFMA
python3 tools/pmc_table.py $RAW/g $RAW/a $RAW/b $RAW/c $RAW/d $RAW/f $RAW/w $RAW/g --match "conv3x3_(wgrad|kernel)" > gpurun_out/${TAG}_pmc.md
cat gpurun_out/${TAG}_pmc.md
