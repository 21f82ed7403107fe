#!/bin/bash
# Runs on the GPU box: launch time of the iteration kernel against the number of frames one
# wavefront walks (0 = the library's own run table), at the default bench shape or $EXTRA.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for f in ${FPRS:-0 8 9 10 11 12 13 14 16 18 22}; do
  timeout -k 10 100 python bench.py --no-cpu --frames-per-run $f ${EXTRA:-} > /tmp/fpr.json 2>/tmp/fpr.err || { echo "fpr $f failed"; tail -3 /tmp/fpr.err; exit 1; }
  python - $f <<'PY'
import json, sys
d = json.loads(open("/tmp/fpr.json").read().strip().splitlines()[-1])
print(f"frames_per_run {sys.argv[1]:>3s}: {d['value']:10.1f} audio-s/s  {d['ms_per_step']:7.3f} ms/step  {d['roofline']['kernel_ms_per_launch']*1e3:7.2f} us/launch")
PY
done
