#!/bin/bash
# where the Stage-1 distillation micro-batch leaves the GPU idle (host-bound seams): kernel trace of the distill-only train leg + gap list
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
rm -rf $OUT/r03y_trace
rocprofv3 --kernel-trace --stats -d $OUT/r03y_trace -- python3 bench.py --mode train --distill-only --train-steps 12 --train-warmup 12 --no-cpu-baseline --no-roofline > $OUT/r03y_trace.log 2>&1
DBT=$(find $OUT/r03y_trace -name "*_results.db" | head -1)
NT=$(python3 - <<PY
import sqlite3
c = sqlite3.connect("$DBT")
names = [r[0] for r in c.execute("select name from kernels order by start")]
idx = [i for i, n in enumerate(names) if "cadamw_update" in n]
per = sum(1 for i in idx if i > idx[-1] - 10)
print(idx[-1] - idx[-1 - 3 * per])
PY
)
python3 tools/rocpd_summary.py $DBT --last $NT --gaps 15 > $OUT/r03y_train_distill_gaps.txt
rm -rf $OUT/r03y_trace
tail -3 $OUT/r03y_trace.log
