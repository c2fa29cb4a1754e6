#!/bin/bash
# Kernel stats of the denoise leg only (the first part of profile_round.sh): bash tools/profile_denoise.sh <tag>  ->  gpurun_out/<tag>_bench_kernel_stats.txt
set -u
TAG=${1:-r04}
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
rm -rf $OUT/${TAG}_trace
