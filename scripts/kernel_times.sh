#!/bin/bash
# Runs on the GPU box: average duration of every kernel of the default bench step
# (rocprofv3 --kernel-trace --stats), for the library given by $GLVOC_LIB or the shipped one.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
export TMPDIR=/tmp
rm -rf /tmp/kt_stats
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_stats -- python bench.py --no-cpu --steps 10 --warmup 5 ${EXTRA:-} > /tmp/kt_stats.log 2>&1 || { tail -5 /tmp/kt_stats.log; exit 1; }
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/kt_stats/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r['Percentage']) > 0.1:
        print(f"{r['Name'][:70]:70s} {int(r['Calls']):6d} calls {float(r['AverageNs'])/1e3:9.2f} us {float(r['Percentage']):6.2f} %")
PY
