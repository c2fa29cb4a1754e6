set -u
TAG=r02m
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_ttrace -- python3 bench.py --mode train --train-steps 12 --train-warmup 12 --no-cpu-baseline --no-roofline > $OUT/${TAG}_ttrace.log 2>&1
DBT=$(find $OUT/${TAG}_ttrace -name "*_results.db" | head -1)
NT=$(python3 - <<PY
import sqlite3
c = sqlite3.connect("$DBT")
names = [r[0] for r in c.execute("select name from kernels order by start")]
idx = [i for i, n in enumerate(names) if "cadamw_update" in n]
print(idx[-1] - idx[-7])
PY
)
{ echo "# last 12 micro-batches = $NT dispatches"; python3 tools/rocpd_summary.py $DBT --last $NT; } > $OUT/${TAG}_train_kernel_stats.txt
rm -rf $OUT/${TAG}_ttrace
head -60 $OUT/${TAG}_train_kernel_stats.txt | cut -c1-200
tail -3 $OUT/${TAG}_ttrace.log
