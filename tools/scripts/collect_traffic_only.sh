cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
RAW=/tmp/raw_t; rm -rf $RAW; mkdir -p $RAW gpurun_out/round2_c
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $RAW/pmc_FETCH_SIZE -o run -- python3 tools/bench_ops.py scanlr --iters 5 > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $RAW/pmc_WRITE_SIZE -o run -- python3 tools/bench_ops.py scanlr --iters 5 > /dev/null 2>&1
python3 tools/pmc_to_json.py $RAW/pmc_FETCH_SIZE $RAW/pmc_WRITE_SIZE > gpurun_out/round2_c/pmc_traffic.json
cat gpurun_out/round2_c/pmc_traffic.json
