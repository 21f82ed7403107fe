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
for sub in "" _n2048 _b4096 _spectral; do
  [ -d "$ROOT/gpurun_out/prof_$TAG$sub" ] || continue
  D=$ROOT/gpurun_out/prof_$TAG$sub
  # the newest files only: gpurun merges into directories that may hold older runs
  python3 "$ROOT/scripts/summarise_profile.py" "$D" "$TAG$sub" > /dev/null
  cp "$D/summary_$TAG$sub.md" "$P/${TAG}${sub}_rocprof_summary.md"
  cp "$D/summary_$TAG$sub.json" "$P/${TAG}${sub}_rocprof_summary.json"
  cp "$(ls -t "$D"/kt/*/*kernel_stats.csv | head -1)" "$P/${TAG}${sub}_kernel_stats.csv"
done
# measured HBM traffic of the iteration kernel, keyed by workload + variant and stamped with the
# digest of the kernel sources the counters were taken with (bench.py reports a record only while
# both still match)
python3 - "$P" "$TAG" <<'PY'
import glob, json, os, sys
prof, tag = sys.argv[1], sys.argv[2]
path = os.path.join(prof, "traffic.json")
doc = {"entries": {}}
if os.path.exists(path):
    old = json.load(open(path))
    if "entries" in old:
        doc = old
files = glob.glob(os.path.join(prof, f"{tag}_rocprof_summary.json")) + glob.glob(os.path.join(prof, f"{tag}_*_rocprof_summary.json"))
for f in sorted(files):
    d = json.load(open(f))
    if d.get("traffic_key") and d.get("gl_engine_iter_bytes_per_launch"):
        doc["entries"][d["traffic_key"]] = {
            "bytes_per_launch": d["gl_engine_iter_bytes_per_launch"],
            "read_bytes": d["gl_engine_iter_read_bytes"], "write_bytes": d["gl_engine_iter_write_bytes"],
            "kernel": d.get("gl_engine_iter_kernel"), "engine_source_digest": d["engine_source_digest"],
            "measured_with": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; read = 2 x FETCH_SIZE KiB (gfx950), write = WRITE_SIZE KiB",
            "profile": "profiles/" + os.path.basename(f)}
json.dump(doc, open(path, "w"), indent=1)
PY
# SQ / LDS counter summaries of the iteration kernel
for d in "$ROOT"/gpurun_out/pmc_${TAG} "$ROOT"/gpurun_out/pmc_${TAG}_*; do
  [ -f "$d/summary.json" ] && cp "$d/summary.json" "$P/$(basename "$d" | sed "s/^pmc_//")_pmc_counters.json"
done
[ -s "$M/halo_loopback.json" ] && tail -1 "$M/halo_loopback.json" > "$P/${TAG}_halo_overhead.jsonl"
echo "profiles/ updated from $M"
