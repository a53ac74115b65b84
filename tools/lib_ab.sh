#!/bin/bash
# A/B of two builds of the library on the headline bench, alternating processes on one box:  bash tools/lib_ab.sh outfitx_amd/libofx_hip_exp.so
set -eo pipefail
EXP=$(pwd)/$1; mkdir -p gpurun_out
for r in 1 2; do
  python bench.py --steps 10 --warmup 3 --cpu-outfits 0 > gpurun_out/ab_base_$r.json 2>/dev/null
  OFX_LIB=$EXP python bench.py --steps 10 --warmup 3 --cpu-outfits 0 > gpurun_out/ab_exp_$r.json 2>/dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/ab_*.json')):
    d = json.loads([l for l in open(f) if l.startswith('{')][-1])
    sh = {(r['M'], r['N'], r['K']): r['us_per_launch'] for r in d['roofline']['per_shape'] if r['kernel'] == 'gemm_w2f8_kernel'}
    print(f.split('/')[-1], d['ms_per_step'], d['step_ms_spread']['median'], ' '.join(f"{k[1]}x{k[2]}:{v}" for k, v in sh.items()))
PY
