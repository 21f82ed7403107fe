#!/bin/bash
# Runs on the GPU box (via gpurun): the bench lines and rocprofv3 summaries that get copied
# into profiles/.  Every GPU step has its own time limit; the script stops at the first failure.
#   scripts/measure_all.sh <tag>
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/measure_$TAG
mkdir -p "$OUT"
cd "$ROOT"
run() {  # name, limit, bench args...
  local name=$1 limit=$2; shift 2
  timeout -k 10 "$limit" python bench.py "$@" > "$OUT/$name.json" 2> "$OUT/$name.err" || { echo "$name failed"; tail -5 "$OUT/$name.err"; exit 1; }
  echo "$name: $(python -c "import json,sys; d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print(d['value'], d['unit'], d['ms_per_step'], 'ms/step', d['roofline']['kernel_ms_per_launch'], 'ms/launch frac', d['roofline']['frac'])")"
}
run bench_default 400
run bench_unfused 200 --no-cpu --unfused
run bench_ljspeech_lengths 200 --no-cpu --lengths ljspeech
run bench_n2048_t865 200 --no-cpu --n-fft 2048 --frames 865
run bench_b4096 300 --no-cpu --batch 4096 --steps 3 --warmup 1
run bench_b128 200 --no-cpu --batch 128
run bench_b32 200 --no-cpu --batch 32
run bench_b8 200 --no-cpu --batch 8
run bench_single_ref_utterance 200 --no-cpu --batch 1 --n-fft 2048 --frames 865 --steps 50 --warmup 5
run ablation_traffic_only 200 --no-cpu --traffic-only
run ablation_compute_only 200 --no-cpu --alias-utts
run ablation_spectral_momentum 200 --no-cpu --linear 0
run ablation_recompute 200 --no-cpu --recompute 1
run ablation_n2048_traffic_only 200 --no-cpu --n-fft 2048 --frames 865 --traffic-only
run ablation_n2048_compute_only 200 --no-cpu --n-fft 2048 --frames 865 --alias-utts
run ablation_no_graph 200 --no-cpu --no-graph --no-persistent
run bench_no_persistent 200 --no-cpu --no-persistent
run bench_config3_n1 400 --no-cpu --config 3
run bench_b256 200 --no-cpu --batch 256 --steps 20 --warmup 10
run bench_b512 200 --no-cpu --batch 512 --steps 10 --warmup 5
run bench_nnls 300 --no-cpu --inversion nnls --steps 5 --warmup 2
scripts/profile_gpu.sh "$TAG" > "$OUT/profile.log" 2>&1 || { echo "profile failed"; tail -5 "$OUT/profile.log"; exit 1; }
scripts/profile_gpu.sh "${TAG}_n2048" --n-fft 2048 --frames 865 > "$OUT/profile_n2048.log" 2>&1 || { echo "profile n2048 failed"; tail -5 "$OUT/profile_n2048.log"; exit 1; }
scripts/profile_gpu.sh "${TAG}_spectral" --linear 0 > "$OUT/profile_spectral.log" 2>&1 || { echo "profile spectral failed"; tail -5 "$OUT/profile_spectral.log"; exit 1; }
# SQ / LDS counters of the iteration kernel, one pass per group (at most 8 SQ counters each)
G1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
G2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
G3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH"
scripts/pmc_gpu.sh "$TAG" "$G1" "$G2" "$G3" > "$OUT/pmc.log" 2>&1 || { echo "pmc failed"; tail -5 "$OUT/pmc.log"; exit 1; }
KERNEL_FILTER=gl_engine_kernel BENCH_ARGS="--n-fft 2048 --frames 865" scripts/pmc_gpu.sh "${TAG}_n2048" "$G1" "$G2" "$G3" > "$OUT/pmc_n2048.log" 2>&1 || { echo "pmc n2048 failed"; tail -5 "$OUT/pmc_n2048.log"; exit 1; }
# the NNLS solver: matrix-core counters of its one kernel
GM="SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE"
KERNEL_FILTER=nnls_fused_kernel BENCH_ARGS="--inversion nnls" scripts/pmc_gpu.sh "${TAG}_nnls" "$GM" > "$OUT/pmc_nnls.log" 2>&1 || { echo "pmc nnls failed"; tail -5 "$OUT/pmc_nnls.log"; exit 1; }
# the pseudo-inverse product (mel_native_kernel): the same matrix-core counters
KERNEL_FILTER=mel_native_kernel scripts/pmc_gpu.sh "${TAG}_melnative" "$GM" > "$OUT/pmc_melnative.log" 2>&1 || { echo "pmc mel_native failed"; tail -5 "$OUT/pmc_melnative.log"; exit 1; }
# the time-sharded loop at the long-form shard size, RCCL loopback: enqueued, one graph, two branches
timeout -k 10 200 python scripts/halo_loopback.py > "$OUT/halo_loopback.json" 2> "$OUT/halo_loopback.err" || { echo "halo loopback failed"; tail -5 "$OUT/halo_loopback.err"; exit 1; }
tail -1 "$OUT/halo_loopback.json"
echo done
