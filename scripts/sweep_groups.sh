run() { name=$1; shift; timeout -k 10 300 python bench.py --no-cpu --no-secondary "$@" > gpurun_out/g.json 2> gpurun_out/g.err || { echo "$name failed"; tail -3 gpurun_out/g.err; exit 1; }; python -c "
import json; d=json.loads(open('gpurun_out/g.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$name', d['value'], d['ms_per_step'], r['utterance_groups_per_step'], r['iterations_per_launch'], r['frac'])"; }
for g in -1 0; do
run "b256 group=$g" --batch 256 --steps 10 --warmup 4 --option group_frames=$g
run "b512 group=$g" --batch 512 --steps 6 --warmup 3 --option group_frames=$g
run "b192 group=$g" --batch 192 --steps 10 --warmup 4 --option group_frames=$g
run "b4096 group=$g" --batch 4096 --steps 2 --warmup 1 --option group_frames=$g
run "config3 group=$g" --config 3 --steps 2 --warmup 1 --option group_frames=$g
run "n2048 b128 group=$g" --n-fft 2048 --frames 865 --batch 128 --steps 5 --warmup 2 --option group_frames=$g
run "n2048 b256 group=$g" --n-fft 2048 --frames 865 --batch 256 --steps 3 --warmup 2 --option group_frames=$g
done
