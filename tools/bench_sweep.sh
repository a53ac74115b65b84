#!/bin/bash
# bench.py under several flag sets in one GPU session: bash tools/bench_sweep.sh "--vit-streams 1" "--vit-streams 2" ...
mkdir -p gpurun_out
for flags in "$@"; do
  timeout -k 10 300 python bench.py --cpu-outfits 0 $flags 2>/dev/null > gpurun_out/_sweep.json
  python - "$flags" <<'PY'
import json, sys
r = json.load(open("gpurun_out/_sweep.json"))
print(f"{sys.argv[1]:40s} {r['value']:9.1f} outfits/s  {r['ms_per_step']:7.3f} ms/step", flush=True)
PY
done
