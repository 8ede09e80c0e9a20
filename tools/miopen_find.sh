#!/bin/bash
# Regenerate / extend mlagg-unet_amd/miopen_db/ on an MI355X: MIOpen find over the convolutions of one bench configuration's
# train step (about 20 minutes per configuration on a fresh box; the search resumes from whatever the database already holds,
# so a run cut short keeps its progress).  Records of all configurations live in the same two files.
#   tools/miopen_find.sh [out_dir] [config]      then copy out_dir/*.txt (not *.time) into mlagg-unet_amd/miopen_db/
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$R/gpurun_out/miopen_db}
mkdir -p "$OUT"
cp -n "$R"/mlagg-unet_amd/miopen_db/*.txt "$OUT"/ 2>/dev/null || true
CFG=${2:-2}
export MLAGG_BENCH_MIOPEN=find
export MIOPEN_USER_DB_PATH=$OUT MIOPEN_CUSTOM_CACHE_DIR=${TMPDIR:-/tmp}/miopen_cache MLAGG_MIOPEN_FIND=1
# the naive reference solvers are never the fastest and take up to 0.4 s per launch to time
export MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW=0
( while true; do sleep 45; echo "[$(date +%T)] find-db entries: $(cat "$OUT"/*.ufdb.txt 2>/dev/null | wc -l)"; done ) &
HB=$!
trap 'kill $HB' EXIT
python3 "$R"/bench.py --config "$CFG" --steps 3 --warmup 1 --no-cpu-baseline --no-roofline
