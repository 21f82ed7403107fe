#!/bin/bash
# Runs on the GPU box: launch time of the iteration kernel with the library's own choices
# (mode and run table) over batch sizes, at n_fft 1024 (430 frames) and 2048 (865 frames).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
run() {
  timeout -k 10 100 python bench.py --no-cpu "$@" > /tmp/sb.json 2>/tmp/sb.err || { echo "failed: $*"; tail -3 /tmp/sb.err; exit 1; }
  python - "$*" <<'PY'
import json, sys
d = json.loads(open("/tmp/sb.json").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:<44s}: {d['roofline']['kernel_ms_per_launch']*1e3:7.2f} us/launch  {d['value']:10.1f} audio-s/s")
PY
}
for b in ${BATCHES:-1 2 4 6 7 8 10 12 16 24 32 48 64}; do run --batch $b; done
for b in ${BATCHES2:-1 2 3 4 8 16 32 64}; do run --n-fft 2048 --frames 865 --batch $b; done
