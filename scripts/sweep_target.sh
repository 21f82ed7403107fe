#!/bin/bash
# launch time of the iteration kernel against the run count, small batches (runs on the GPU box)
#   scripts/sweep_target.sh "<batches>" "<targets>"
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for b in $1; do
  for t in $2; do
    timeout -k 10 100 python bench.py --no-cpu --no-secondary --batch $b --run-target $t --steps 20 --warmup 8 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(f\"batch {sys.argv[1]:>3s} run_target {sys.argv[2]:>5s}: {d['roofline']['kernel_ms_per_iteration']*1e3:7.2f} us/iteration  {d['value']:10.1f} audio-s/s\")" $b $t || exit 1
  done
done
