#!/bin/bash
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
rm -rf $OUT/r03aj_trace
rocprofv3 --kernel-trace --stats -d $OUT/r03aj_trace -- python3 bench.py --mode train2 --train-steps 12 --train-warmup 12 --no-cpu-baseline --no-roofline > $OUT/r03aj_trace.log 2>&1
DBT=$(find $OUT/r03aj_trace -name "*_results.db" | head -1)
NT=$(python3 - <<PY
import sqlite3
c = sqlite3.connect("$DBT")
names = [r[0] for r in c.execute("select name from kernels order by start")]
idx = [i for i, n in enumerate(names) if "cadamw_update" in n]
per = sum(1 for i in idx if i > idx[-1] - 10)
print(idx[-1] - idx[-1 - 3 * per])
PY
)
python3 tools/rocpd_summary.py $DBT --last $NT --gaps 100 > $OUT/r03aj_train2_gaps.txt
rm -rf $OUT/r03aj_trace
