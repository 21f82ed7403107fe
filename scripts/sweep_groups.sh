# batch sizes around the group limits, automatic groups against one piece (runs on the GPU box)
run() { name=$1; shift; timeout -k 10 300 python bench.py --no-cpu --no-secondary "$@" > gpurun_out/g.json 2> gpurun_out/g.err || { echo "$name failed"; tail -3 gpurun_out/g.err; exit 1; }; python -c "
import json; d=json.loads(open('gpurun_out/g.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$name', d['value'], d['ms_per_step'], r['utterance_groups_per_step'], r['iterations_per_launch'], r['frac'])"; }
for b in 128 144 160 192 256 384; do
run "b$b one piece" --batch $b --steps 8 --warmup 3 --option group_frames=-1
run "b$b auto" --batch $b --steps 8 --warmup 3
done
run "ljspeech b256 one piece" --batch 256 --lengths ljspeech --steps 5 --warmup 2 --option group_frames=-1
run "ljspeech b256 auto" --batch 256 --lengths ljspeech --steps 5 --warmup 2
