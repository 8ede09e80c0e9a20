#!/bin/bash
# Regenerate mlagg-unet_amd/miopen_db/ on an MI355X: exhaustive MIOpen find over the convolutions of the config-2 train
# step (about 20 minutes on a fresh box; the search resumes from whatever the database already holds).
#   tools/miopen_find.sh [out_dir]      then copy out_dir/*.txt (not *.time) into mlagg-unet_amd/miopen_db/
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$R/gpurun_out/miopen_db}
mkdir -p "$OUT"
cp -n "$R"/mlagg-unet_amd/miopen_db/*.txt "$OUT"/ 2>/dev/null || true
export MIOPEN_USER_DB_PATH=$OUT MIOPEN_CUSTOM_CACHE_DIR=${TMPDIR:-/tmp}/miopen_cache MLAGG_MIOPEN_FIND=1
( while true; do sleep 45; echo "[$(date +%T)] find-db entries: $(cat "$OUT"/*.ufdb.txt 2>/dev/null | wc -l)"; done ) &
HB=$!
trap 'kill $HB' EXIT
python3 "$R"/bench.py --steps 10 --warmup 3 --no-cpu-baseline
