#!/bin/bash
# Runs on a one-GPU box (via gpurun): `python bench.py --gpus 2` for the three multi-GPU
# configurations in rehearsal mode (both ranks on GPU 0, collectives over gloo), as the driver
# would start them on a node, plus config 4 on one rank.  Each step has its own time limit.
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/rehearse_$TAG
mkdir -p "$OUT"
cd "$ROOT"
run() {  # name, limit, bench args...
  local name=$1 limit=$2; shift 2
  timeout -k 10 "$limit" python bench.py "$@" > "$OUT/$name.json" 2> "$OUT/$name.err" || { echo "$name failed"; tail -15 "$OUT/$name.err"; exit 1; }
  echo "$name: $(python -c "import json,sys; d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('n_gpus', d['n_gpus'], d['value'], d['unit'], d['ms_per_step'], 'ms/step', d.get('halo', ''))")"
}
run config1_gpus2 300 --gpus 2 --rehearse --steps 10 --warmup 3 &&
run config3_gpus2 400 --gpus 2 --rehearse --config 3 --batch 512 --steps 3 --warmup 1 &&
run config4_gpus2 400 --gpus 2 --rehearse --config 4 --steps 3 --warmup 1 &&
run config4_gpus1 300 --config 4 --no-cpu &&
echo done
