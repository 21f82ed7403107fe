#!/bin/bash
# Copies what scripts/measure_all.sh <tag> left under gpurun_out/ into profiles/ (run here,
# after the GPU call): the bench lines, the rocprofv3 summaries and the kernel statistics.
#   scripts/collect_profiles.sh <tag>
set -eu
TAG=${1:-r01}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
M=$ROOT/gpurun_out/measure_$TAG
P=$ROOT/profiles
cp "$M/bench_default.json" "$P/${TAG}_bench.json"
: > "$P/${TAG}_bench_variants.jsonl"
for f in "$M"/*.json; do
  name=$(basename "$f" .json)
  [ "$name" = bench_default ] && continue
  python3 - "$name" "$f" >> "$P/${TAG}_bench_variants.jsonl" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(json.dumps({"run": sys.argv[1], **d}))
PY
done
for sub in "" _n2048; do
  D=$ROOT/gpurun_out/prof_$TAG$sub
  # the newest files only: gpurun merges into directories that may hold older runs
  python3 "$ROOT/scripts/summarise_profile.py" "$D" "$TAG$sub" > /dev/null
  cp "$D/summary_$TAG$sub.md" "$P/${TAG}${sub}_rocprof_summary.md"
  cp "$D/summary_$TAG$sub.json" "$P/${TAG}${sub}_rocprof_summary.json"
  cp "$(ls -t "$D"/kt/*/*kernel_stats.csv | head -1)" "$P/${TAG}${sub}_kernel_stats.csv"
done
cp "$ROOT/gpurun_out/prof_$TAG/summary_$TAG.json" "$P/traffic.json"
echo "profiles/ updated from $M"
