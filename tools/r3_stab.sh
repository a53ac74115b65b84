set -e
mkdir -p gpurun_out
python tools/step_ab.py 15 1 0 > gpurun_out/stab_ab.txt 2>&1
for i in 1 2 3; do python bench.py --steps 10 --warmup 3 > gpurun_out/stab_bench_$i.json 2>gpurun_out/stab_bench_$i.err; done
python bench.py --steps 10 --warmup 3 --overlap-towers 0 > gpurun_out/stab_bench_nooverlap.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/stab_bench_*.json')):
    for l in open(f):
        if l.startswith('{'):
            d=json.loads(l); print(f, d['value'], d['ms_per_step'])
PY
cat gpurun_out/stab_ab.txt
