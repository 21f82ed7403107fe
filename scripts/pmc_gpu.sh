#!/bin/bash
# One rocprofv3 --pmc pass per argument (a quoted, space-separated counter group) over a
# short bench run; prints the per-launch average of every counter for the iteration kernel.
#   scripts/pmc_gpu.sh <tag> "<group 1>" ["<group 2>" ...]
#   (BENCH_ARGS env adds bench flags; KERNEL_FILTER env names the kernel, default gl_engine_kernel; launches of an
#   iteration-spanning kernel that only count resident workgroups are left out of the averages)
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout -s KILL 180 rocprofv3 --pmc $grp --output-format csv -d "$OUT/p$i" -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu ${BENCH_ARGS:-} > "$OUT/p$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$OUT/p$i.log"; exit 1; }
done
python3 - "$OUT" "${KERNEL_FILTER:-gl_engine_kernel}" <<'PY'
import csv, glob, sys, collections, json
out, want = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    with open(path) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"]
            if want not in k: continue
            k = k[k.index(want):][:40]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
def mean(k, v):
    if k.rstrip(">").endswith(("12, false", "13, false", "14, false")):
        v = [x for x in v if x >= 0.2 * max(v)] or v   # census launches
    return sum(v) / len(v)
res = {k: {c: mean(k, v) for c, v in d.items()} for k, d in acc.items()}
for k, d in res.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} {v:16.1f}")
json.dump(res, open(out + "/summary.json", "w"), indent=1)
PY
