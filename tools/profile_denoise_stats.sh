#!/bin/bash
# rocprofv3 --kernel-trace --stats of the graph-replayed denoise leg only (the first block of tools/profile_round.sh):  bash tools/profile_denoise_stats.sh <tag>
set -u
TAG=${1:-r06}
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
rm -rf $OUT/${TAG}_trace
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -- python3 bench.py --mode denoise --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $OUT/${TAG}_trace.log 2>&1
DB=$(find $OUT/${TAG}_trace -name "*_results.db" | head -1)
N=$(python3 - <<PY
import sqlite3
c = sqlite3.connect("$DB")
names = [r[0] for r in c.execute("select name from kernels order by start")]
idx = [i for i, n in enumerate(names) if "cfg_ddim" in n]
print((idx[-1] - idx[-2]) * 10)
PY
)
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --mode denoise --steps 20 --warmup 3 --no-cpu-baseline --no-roofline   (last 10 steps = $N dispatches)"; python3 tools/rocpd_summary.py $DB --last $N; } > $OUT/${TAG}_bench_kernel_stats.txt
python3 tools/rocprof_frac.py $OUT/${TAG}_bench_kernel_stats.txt --json $OUT/${TAG}_rocprof_frac.json > $OUT/${TAG}_rocprof_frac.txt 2>&1
rm -rf $OUT/${TAG}_trace
head -45 $OUT/${TAG}_bench_kernel_stats.txt; head -12 $OUT/${TAG}_rocprof_frac.txt
