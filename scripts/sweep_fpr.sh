#!/bin/bash
# frames_per_run sweep of the default bench workload (runs on the GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for f in "$@"; do
  echo -n "fpr=$f "
  timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu --frames-per-run $f ${BENCH_ARGS:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_launch'], d['roofline']['frac'])" || exit 1
done
