#!/bin/bash
set -eo pipefail
EXP=$(pwd)/outfitx_amd/libofx_hip_exp.so; mkdir -p gpurun_out; : > gpurun_out/wkeep.txt
for r in 1 2 3; do
  echo "== base" >> gpurun_out/wkeep.txt; python tools/gemm_w2_bench.py 2>/dev/null | awk -F'|' '{print $1 "|" $NF}' >> gpurun_out/wkeep.txt
  echo "== weight-kept order" >> gpurun_out/wkeep.txt; OFX_LIB=$EXP python tools/gemm_w2_bench.py 2>/dev/null | awk -F'|' '{print $1 "|" $NF}' >> gpurun_out/wkeep.txt
done
echo "== x3 base" >> gpurun_out/wkeep.txt; python tools/gemm_x3_bench.py 2>/dev/null | cut -c1-200 >> gpurun_out/wkeep.txt
echo "== x3 weight-kept" >> gpurun_out/wkeep.txt; OFX_LIB=$EXP python tools/gemm_x3_bench.py 2>/dev/null | cut -c1-200 >> gpurun_out/wkeep.txt
echo "== x3 base" >> gpurun_out/wkeep.txt; python tools/gemm_x3_bench.py 2>/dev/null | cut -c1-200 >> gpurun_out/wkeep.txt
echo "== x3 weight-kept" >> gpurun_out/wkeep.txt; OFX_LIB=$EXP python tools/gemm_x3_bench.py 2>/dev/null | cut -c1-200 >> gpurun_out/wkeep.txt
cat gpurun_out/wkeep.txt
bash tools/lib_ab.sh outfitx_amd/libofx_hip_exp.so
