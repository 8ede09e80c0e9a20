#!/bin/bash
# round 4: parity of the token-major scan (K1f), the MSMM / full-model goldens on top of it, A/B bench lines
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_msmm_scan_gpu.py -x -q > $O/pytest_scan.log 2>&1 || { tail -40 $O/pytest_scan.log; exit 1; }
tail -3 $O/pytest_scan.log
timeout -k 10 600 python -m pytest tests/test_blocks_gpu.py tests/test_configs_gpu.py tests/test_selscan_gpu.py -x -q > $O/pytest_model.log 2>&1 || { tail -40 $O/pytest_model.log; exit 1; }
tail -3 $O/pytest_model.log
B="timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
$B > $O/fused.json 2> $O/fused.err
MLAGG_MSMM_FUSED=0 $B > $O/unfused.json 2> $O/unfused.err
for f in $O/*.json; do echo "$f: $(python -c "import json,sys; d=json.load(open('$f')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['kernel'], r['frac'], r['avg_launch_ms']); print({k:v for k,v in r['all_kernels_ms_per_step'].items() if 'scan' in k or 'tok_' in k})" 2>&1 | tail -2)"; done
