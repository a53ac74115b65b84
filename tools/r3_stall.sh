#!/bin/bash
set -eo pipefail
mkdir -p gpurun_out
for r in 1 2 3 4 5 6; do python bench.py --steps 10 --warmup 3 --cpu-outfits 0 > gpurun_out/stall_$r.json 2>/dev/null; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/stall_*.json')):
    d = json.loads([l for l in open(f) if l.startswith('{')][-1]); s = d['step_ms_spread']
    print(f.split('/')[-1], d['ms_per_step'], 'allocs', s['device_allocs_in_timed_region'], s['all'], 'host', s['host_issue_ms'])
PY
