#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats and the two HBM counter passes of
# the default bench workload.  Each rocprofv3 run is its own process with its own limit.
#   scripts/profile_gpu.sh <tag> [extra bench args]
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 10 --warmup 10 --no-cpu $*"
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- $BENCH > "$OUT/kt.log" 2>&1 || { echo "kernel-trace run failed"; tail -5 "$OUT/kt.log"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $BENCH > "$OUT/fetch.log" 2>&1 || { echo "FETCH_SIZE run failed"; tail -5 "$OUT/fetch.log"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $BENCH > "$OUT/write.log" 2>&1 || { echo "WRITE_SIZE run failed"; tail -5 "$OUT/write.log"; exit 1; }
python3 "$ROOT/scripts/summarise_profile.py" "$OUT" "$TAG"
