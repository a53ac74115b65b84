#!/bin/bash
# do the GEMM kernels' clocks follow the operand values?  same kernels, same shapes: random / constant / zero operands
set -eo pipefail
mkdir -p gpurun_out; : > gpurun_out/data_clock.txt
python tools/probe_clock.py >> gpurun_out/data_clock.txt 2>&1 || true
for d in randn const zero randn; do
  echo "== OFX_DATA=$d" >> gpurun_out/data_clock.txt
  OFX_DATA=$d python tools/gemm_w2_bench.py "vit qkv" "vit fc2" 2>/dev/null | cut -c1-260 >> gpurun_out/data_clock.txt
done
cat gpurun_out/data_clock.txt
