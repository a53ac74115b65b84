#!/bin/bash
set -eo pipefail
ROOT=$(pwd); mkdir -p gpurun_out; rm -rf gpurun_out/st
python3 bench.py --cpu-outfits 0 > gpurun_out/st_plain.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/st -- python3 $ROOT/bench.py --cpu-outfits 0 > $ROOT/gpurun_out/st.log 2>&1
cd $ROOT
python3 tools/trace_stalls.py $(find gpurun_out/st -name "*kernel_trace.csv" | head -1) > gpurun_out/stall_trace.txt
rm -rf gpurun_out/st
python3 - <<'PY'
import json
for f in ('gpurun_out/st_plain.json', 'gpurun_out/st.log'):
    l = [x for x in open(f) if x.startswith('{"metric')][-1]; d = json.loads(l); s = d['step_ms_spread']
    print(f, d['ms_per_step'], 'median', s['median'], [x for x in s['all'] if x > 40])
PY
cat gpurun_out/stall_trace.txt
