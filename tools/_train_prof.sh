set -u
TAG=r02n
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_t2trace -- python3 bench.py --mode train2 --train-steps 4 --train-warmup 2 --no-cpu-baseline --no-roofline > $OUT/${TAG}_t2trace.log 2>&1
DBT=$(find $OUT/${TAG}_t2trace -name "*_results.db" | head -1)
NT=$(python3 - <<PY
import sqlite3
c = sqlite3.connect("$DBT")
names = [r[0] for r in c.execute("select name from kernels order by start")]
idx = [i for i, n in enumerate(names) if "cadamw_update" in n]
per = sum(1 for i in idx if i > idx[-1] - 10)   # launches of the last optimizer step (one per arena)
print(idx[-1] - idx[-1 - 2 * per], per)
PY
)
set -- $NT
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --mode train2 --train-steps 4 --train-warmup 2 ... (last 2 optimizer steps = 4 micro-batches = $1 dispatches; $2 arenas)"; python3 tools/rocpd_summary.py $DBT --last $1; } > $OUT/${TAG}_train2_kernel_stats.txt
rm -rf $OUT/${TAG}_t2trace
head -70 $OUT/${TAG}_train2_kernel_stats.txt | cut -c1-200
tail -2 $OUT/${TAG}_t2trace.log | cut -c1-300
