#!/bin/bash
# throughput of every tower scheme on the headline step, final build (one box)
set -eo pipefail
mkdir -p gpurun_out; : > gpurun_out/ladder_tp.txt
for s in bf16 f16 f16w2 f16w2x f16x3; do
  python bench.py --tower-precision $s --cpu-outfits 0 --secondary "" --steps 20 --warmup 5 2>/dev/null > gpurun_out/ladder_$s.json
  python - "$s" <<'PY' >> gpurun_out/ladder_tp.txt
import json, sys
d = json.loads([l for l in open(f'gpurun_out/ladder_{sys.argv[1]}.json') if l.startswith('{')][-1])
print(sys.argv[1], d['value'], d['ms_per_step'], d['step_ms_spread']['median'], d.get('parity_rel_err_vs_reference'), d['config']['launch'][:20])
PY
done
cat gpurun_out/ladder_tp.txt
