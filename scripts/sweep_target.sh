#!/bin/bash
# Runs on the GPU box: launch time of the iteration kernel against the number of runs the
# automatic run table aims at (0 = the device's wave slots), for the batch sizes in $BATCHES.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for b in ${BATCHES:-8 16 32 64}; do
  for t in ${TARGETS:-0 768 1024 1536 2048 2560 2816 3072}; do
    timeout -k 10 100 python bench.py --no-cpu --batch $b --single-frame 0 --run-target $t ${EXTRA:-} > /tmp/tg.json 2>/tmp/tg.err || { echo "target $t failed"; tail -3 /tmp/tg.err; exit 1; }
    python - $b $t <<'PY'
import json, sys
d = json.loads(open("/tmp/tg.json").read().strip().splitlines()[-1])
print(f"batch {sys.argv[1]:>3s} run_target {sys.argv[2]:>5s}: {d['roofline']['kernel_ms_per_launch']*1e3:7.2f} us/launch  {d['value']:10.1f} audio-s/s")
PY
  done
done
