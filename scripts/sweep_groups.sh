run() { name=$1; shift; timeout -k 10 300 python bench.py --no-cpu --no-secondary "$@" > gpurun_out/g.json 2> gpurun_out/g.err || { echo "$name failed"; tail -3 gpurun_out/g.err; exit 1; }; grep "group of" gpurun_out/g.err | head -2; python -c "
import json; d=json.loads(open('gpurun_out/g.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$name', d['value'], d['ms_per_step'], r['utterance_groups_per_step'], r['iterations_per_launch'], r['frac'])"; }
for f in 32 24 18 14; do
run "config3 fpr=$f one piece" --config 3 --steps 2 --warmup 1 --frames-per-run $f
GLV_TMP_FPRG=1 run "config3 fpr=$f groups" --config 3 --steps 2 --warmup 1 --frames-per-run $f
done
run "ljspeech b4096 auto groups" --batch 4096 --lengths ljspeech --steps 2 --warmup 1
