#!/bin/bash
# Tile-walk group size (row panels per L2 group, ofx_tune(0, v); default 8) on the headline bench, launch by launch: value, ms per step and the four ViT
# shapes' us per launch.  Round 4: 2 / 4 / 8 / 16 all within run-to-run noise (profiles/r04_group_m_sweep.txt).
for v in 8 4 16 2 8; do
  OFX_TUNE=0:$v timeout -k 10 200 python bench.py --steps 16 --warmup 4 --cpu-outfits 0 --graph 0 --rung "" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
ps=[(p['N'],p['K'],p['us_per_launch']) for p in d['roofline']['per_shape'][:4]]
print('group_m $v', d['value'], d['ms_per_step'], ps, flush=True)"
done
