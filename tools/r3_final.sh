#!/bin/bash
# round-3 validation sequence on the GPU box: full GPU suite -> headline bench -> rocprofv3 passes -> the other BASELINE configs
set -eo pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/final_gpu_tests.log 2>&1 || { tail -30 gpurun_out/final_gpu_tests.log; exit 1; }
tail -2 gpurun_out/final_gpu_tests.log
python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err || { tail -20 gpurun_out/final_bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/final_bench.json') if l.startswith('{')][-1])
print({k: d[k] for k in ('value', 'ms_per_step', 'step_ms_spread', 'parity_rel_err_vs_reference')}); print(d['roofline']['achieved'], d['roofline']['frac'], d['cpu_baseline'])
PY
bash tools/profile_r03.sh > gpurun_out/final_profile.log 2>&1 || { tail -20 gpurun_out/final_profile.log; exit 1; }
python tools/bench_configs.py > gpurun_out/final_configs.jsonl 2> gpurun_out/final_configs.err || { tail -20 gpurun_out/final_configs.err; exit 1; }
cat gpurun_out/final_configs.jsonl | cut -c1-300
