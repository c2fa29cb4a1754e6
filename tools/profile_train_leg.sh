#!/bin/bash
# rocprofv3 kernel statistics of ONE training leg (the train-leg block of tools/profile_round.sh on its own):
#   bash tools/profile_train_leg.sh r04r train2      ->  gpurun_out/r04r_train2_kernel_stats.txt   (last 3 optimizer steps = 6 micro-batches)
set -u
TAG=${1:-r04}
LEG=${2:-train2}
EXTRA=${3:-}          # e.g. --distill-only
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
rm -rf $OUT/${TAG}_ttrace
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_ttrace -- python3 bench.py --mode $LEG $EXTRA --train-steps 12 --train-warmup 12 --no-cpu-baseline --no-roofline > $OUT/${TAG}_${LEG}_trace.log 2>&1
DBT=$(find $OUT/${TAG}_ttrace -name "*_results.db" | head -1)
NT=$(python3 - <<PY
import sqlite3
c = sqlite3.connect("$DBT")
names = [r[0] for r in c.execute("select name from kernels order by start")]
idx = [i for i, n in enumerate(names) if "cadamw_update" in n]
per = sum(1 for i in idx if i > idx[-1] - 10)          # arenas = CAdamW launches of one optimizer step
print(idx[-1] - idx[-1 - 3 * per])
PY
)
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --mode $LEG $EXTRA --train-steps 12 --train-warmup 12 --no-cpu-baseline --no-roofline   (last 3 optimizer steps = 6 micro-batches = $NT dispatches; divide by 6 for one micro-batch)"; python3 tools/rocpd_summary.py $DBT --last $NT; } > $OUT/${TAG}_${LEG}${EXTRA//-/_}_kernel_stats.txt
rm -rf $OUT/${TAG}_ttrace
head -30 $OUT/${TAG}_${LEG}${EXTRA//-/_}_kernel_stats.txt | cut -c1-150
