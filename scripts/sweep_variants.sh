#!/bin/bash
# bench the library variants under tts_amd/csrc/variants (runs on the GPU box)
#   scripts/sweep_variants.sh "<bench args>" a b c ...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
ARGS=$1; shift
for v in "$@"; do
  echo -n "variant=$v "
  GLVOC_LIB=$ROOT/tts_amd/csrc/variants/libglvoc_$v.so timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_launch'], d['roofline']['frac'])" || exit 1
done
