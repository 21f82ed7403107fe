#!/bin/bash
# Runs on the GPU box: times the iteration kernel of every library build under
# tts_amd/csrc/variants/ (timing-only experiments; their output is not checked) beside the
# shipped build, at the two headline shapes.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
export GLV_BENCH_NO_CHECK=1
for lib in tts_amd/csrc/libglvoc.so tts_amd/csrc/variants/*.so; do
  for shape in "--n-fft 1024 --frames 430" "--n-fft 2048 --frames 865 --steps 20 --warmup 10"; do
    GLVOC_LIB=$ROOT/$lib timeout -k 10 120 python bench.py --no-cpu $shape ${EXTRA:-} > /tmp/exp.json 2>/tmp/exp.err || { echo "$lib $shape failed"; tail -3 /tmp/exp.err; exit 1; }
    python - "$lib" "$shape" <<'PY'
import json, sys
d = json.loads(open("/tmp/exp.json").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:50s} {sys.argv[2]:45s} {d['roofline']['kernel_ms_per_launch']*1e3:8.2f} us/launch  {d['ms_per_step']:8.3f} ms/step")
PY
  done
done
