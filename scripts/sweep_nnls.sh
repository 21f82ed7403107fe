#!/bin/bash
# Runs on the GPU box: NNLS solve time of every library build under tts_amd/csrc/variants/
# beside the shipped build (same box, same call).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for rep in 1 2; do
for lib in tts_amd/csrc/libglvoc.so tts_amd/csrc/variants/*.so; do
  GLVOC_LIB=$ROOT/$lib timeout -k 10 120 python bench.py --no-cpu --inversion nnls --steps 5 --warmup 2 ${EXTRA:-} > /tmp/nn.json 2>/tmp/nn.err || { echo "$lib failed"; tail -3 /tmp/nn.err; exit 1; }
  python - "$lib" <<'PY'
import json, sys
d = json.loads(open("/tmp/nn.json").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:50s} {d['nnls']['ms_per_solve']:8.3f} ms per solve")
PY
done
done
