#!/bin/bash
# rocprofv3 passes of the headline bench (counters in their own passes, no --pmc together with trace domains other than --kernel-trace)
# -> gpurun_out/prof_r04/{trace,sq,fetch,write,sq2}.  Run on the GPU box from the repo root:  bash tools/profile_r04.sh [extra bench flags]
set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_r04; rm -rf $OUT; mkdir -p $OUT
# the un-profiled line first: its roofline.per_kernel carries the algorithmic bytes per launch that the PMC traffic is set against
python3 $ROOT/bench.py --steps 5 --warmup 2 --cpu-outfits 0 "$@" > $OUT/bench_line.json 2> $OUT/bench_line.err || { tail -5 $OUT/bench_line.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --steps 1 --warmup 1 --cpu-outfits 0 --graph 0 $@"       # counter passes: launch by launch (one dispatch record per kernel)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
# the default command's kernel averages (40 timed steps after 8 warm-up: the clock state bench.py's own live GEMM timing sees)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_default -- python3 $ROOT/bench.py --cpu-outfits 0 "$@" > $OUT/trace_default.log 2>&1 \
  && cp $(find $OUT/trace_default -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_default_command.csv && grep -o '{"metric.*' $OUT/trace_default.log > $OUT/bench_line_under_rocprof.json
# the same with the towers on ONE stream: per-kernel averages that are comparable with bench.py's live per-launch timing (concurrent kernels share time)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_serial -- python3 $ROOT/bench.py --cpu-outfits 0 --overlap-towers 0 --graph 0 "$@" > $OUT/trace_serial.log 2>&1 \
  && cp $(find $OUT/trace_serial -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_single_stream.csv && grep -o '{"metric.*' $OUT/trace_serial.log > $OUT/bench_line_single_stream_under_rocprof.json
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES -d $OUT/sq -- $B > $OUT/sq.log 2>&1 || { tail -5 $OUT/sq.log; exit 1; }
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/fetch -- $B > $OUT/fetch.log 2>&1 || { tail -5 $OUT/fetch.log; exit 1; }
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/write -- $B > $OUT/write.log 2>&1 || { tail -5 $OUT/write.log; exit 1; }
# where the waves' cycles go (issue / wait / LDS): optional, a missing counter name must not fail the run
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F8 -d $OUT/sq2 -- $B > $OUT/sq2.log 2>&1 \
  || rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/sq2 -- $B > $OUT/sq2.log 2>&1 || echo "sq2 pass failed (optional)"
cd $ROOT
python3 tools/pmc_mfma_util.py $OUT > $OUT/mfma_util.json && cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
# keep the merge-back small: drop the raw per-dispatch csv files
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
ls -la $OUT; head -c 1500 $OUT/mfma_util.json
