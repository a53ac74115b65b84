#!/bin/bash
# rocprofv3 kernel stats of the other BASELINE configs: cfg1 (32 precomputed outfits) and cfg4 (1000 x 100k retrieval),
# each in its own pass -> gpurun_out/prof_r04_cfg/{cfg1,cfg4,cfg4_matrix}_kernel_stats.csv.  Run on the GPU box from the repo root.
set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_r04_cfg; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in cfg1 cfg4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$c -- python3 $ROOT/tools/bench_configs.py $c > $OUT/$c.log 2>&1 || { tail -5 $OUT/$c.log; exit 1; }
  cp $(find $OUT/$c -name "*kernel_stats.csv" | head -1) $OUT/${c}_kernel_stats.csv
  grep '^{' $OUT/$c.log > $OUT/${c}_lines.jsonl
done
OFX_TUNE=17:0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg4m -- python3 $ROOT/tools/bench_configs.py cfg4 > $OUT/cfg4m.log 2>&1 \
  && cp $(find $OUT/cfg4m -name "*kernel_stats.csv" | head -1) $OUT/cfg4_matrix_path_kernel_stats.csv && grep '^{' $OUT/cfg4m.log > $OUT/cfg4_matrix_path_lines.jsonl
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $OUT/cfg4_sq -- python3 $ROOT/tools/bench_configs.py cfg4 > $OUT/cfg4_sq.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/cfg4_fetch -- python3 $ROOT/tools/bench_configs.py cfg4 > $OUT/cfg4_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/cfg4_write -- python3 $ROOT/tools/bench_configs.py cfg4 > $OUT/cfg4_write.log 2>&1
cd $ROOT
python3 - <<'PY' > $OUT/cfg4_counters.txt
import csv, glob, os, re
from collections import defaultdict
OUT = os.path.join("gpurun_out", "prof_r04_cfg")
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(set))
for d in ("cfg4_sq", "cfg4_fetch", "cfg4_write"):
    for f in glob.glob(os.path.join(OUT, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "").replace("(anonymous namespace)::", "").strip()[:48]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for k, c in acc.items():
    if not any(t in k for t in ("dist_", "topk_", "sqnorm")): continue
    disp = max(len(v) for v in n[k].values())
    mu = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(c.get("GRBM_GUI_ACTIVE", 1) / 8 * 1024, 1)
    print(f"{k:50s} dispatches {disp:4d}  MfmaUtil {mu:.3f}  FETCH_SIZE x2 {2 * c.get('FETCH_SIZE', 0) * 1024 / disp / 1e6:9.1f} MB/dispatch  WRITE_SIZE {c.get('WRITE_SIZE', 0) * 1024 / disp / 1e6:9.1f} MB/dispatch")
PY
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
ls $OUT; cat $OUT/cfg4_lines.jsonl $OUT/cfg4_matrix_path_lines.jsonl; cat $OUT/cfg4_counters.txt
