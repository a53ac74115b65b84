#!/bin/bash
# kernel-trace timeline of the overlapped headline step + the host's issue time per step -> gpurun_out/{timeline,host_time}.txt
set -eo pipefail
ROOT=$(pwd); mkdir -p $ROOT/gpurun_out; rm -rf $ROOT/gpurun_out/tl
python3 tools/step_host_time.py > gpurun_out/host_time.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/tl -- python3 $ROOT/bench.py --steps 4 --warmup 2 --cpu-outfits 0 > $ROOT/gpurun_out/tl.log 2>&1
cd $ROOT
python3 tools/trace_timeline.py $(find gpurun_out/tl -name "*kernel_trace.csv" | head -1) > gpurun_out/timeline.txt
rm -rf gpurun_out/tl
cat gpurun_out/host_time.txt gpurun_out/timeline.txt
