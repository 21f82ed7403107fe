# group size sweep for a large batch (runs on the GPU box): frames per wave slot x 3072 slots
run() { name=$1; shift; timeout -k 10 300 python bench.py --no-cpu --no-secondary "$@" > gpurun_out/g.json 2> gpurun_out/g.err || { echo "$name failed"; tail -3 gpurun_out/g.err; exit 1; }; python -c "
import json; d=json.loads(open('gpurun_out/g.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$name', d['value'], d['ms_per_step'], r['utterance_groups_per_step'], r['iterations_per_launch'], r['frac'])"; }
for f in 10 12 14 16 18 20 22 24; do
run "b1024 frames/slot=$f" --batch 1024 --steps 3 --warmup 2 --option group_frames=$((f * 3072))
done
for f in 27 40 54 80; do
run "n2048 b256 frames/slot=$f" --n-fft 2048 --frames 865 --batch 256 --steps 2 --warmup 1 --option group_frames=$((f * 2048))
done
