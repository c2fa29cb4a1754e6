#!/bin/bash
# Profiles of the default bench (run on the GPU box from the repo root):   bash tools/profile_round.sh r02j
#   <tag>_bench_line.json        python bench.py (all legs)
#   <tag>_bench_kernel_stats.txt rocprofv3 --kernel-trace --stats of the denoise leg, last 10 steps
#   <tag>_train_kernel_stats.txt, <tag>_train2_kernel_stats.txt   rocprofv3 --kernel-trace --stats of the Stage-1 / Stage-2 training legs, last 6 micro-batches
#   <tag>_hbm_traffic.json       two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of 6 eager denoise steps, per kernel family
#   <tag>_train_traffic.json     the same two passes over four eager Stage-1 distillation micro-batches
set -u
TAG=${1:-r02}
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
python3 bench.py > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench.err
rm -rf $OUT/${TAG}_trace $OUT/${TAG}_pmc_f $OUT/${TAG}_pmc_w
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -- python3 bench.py --mode denoise --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $OUT/${TAG}_trace.log 2>&1
DB=$(find $OUT/${TAG}_trace -name "*_results.db" | head -1)
N=$(python3 - <<PY
import sqlite3
c = sqlite3.connect("$DB")
names = [r[0] for r in c.execute("select name from kernels order by start")]
# dispatches per denoise step = distance between the last two cfg_ddim kernels
idx = [i for i, n in enumerate(names) if "cfg_ddim" in n]
print((idx[-1] - idx[-2]) * 10)
PY
)
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --mode denoise --steps 20 --warmup 3 --no-cpu-baseline --no-roofline   (last 10 steps = $N dispatches)"; python3 tools/rocpd_summary.py $DB --last $N; } > $OUT/${TAG}_bench_kernel_stats.txt
python3 tools/rocprof_frac.py $OUT/${TAG}_bench_kernel_stats.txt --json $OUT/${TAG}_rocprof_frac.json > $OUT/${TAG}_rocprof_frac.txt 2>&1
# train legs: the last 3 optimizer steps (= 6 micro-batches, two 2,3,4-step cycles), delimited by the CAdamW kernels (one launch per arena)
for LEG in train train2; do
  rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_ttrace -- python3 bench.py --mode $LEG --train-steps 12 --train-warmup 12 --no-cpu-baseline --no-roofline > $OUT/${TAG}_${LEG}_trace.log 2>&1
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
  { echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --mode $LEG --train-steps 12 --train-warmup 12 --no-cpu-baseline --no-roofline   (last 3 optimizer steps = 6 micro-batches = $NT dispatches; divide by 6 for one micro-batch)"; python3 tools/rocpd_summary.py $DBT --last $NT; } > $OUT/${TAG}_${LEG}_kernel_stats.txt
  rm -rf $OUT/${TAG}_ttrace
done
rocprofv3 --pmc FETCH_SIZE -d $OUT/${TAG}_pmc_f -- python3 bench.py --mode denoise --steps 5 --warmup 0 --no-graph --no-cpu-baseline --no-roofline > $OUT/${TAG}_pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/${TAG}_pmc_w -- python3 bench.py --mode denoise --steps 5 --warmup 0 --no-graph --no-cpu-baseline --no-roofline > $OUT/${TAG}_pmc_w.log 2>&1
python3 tools/pmc_traffic.py $(find $OUT/${TAG}_pmc_f -name "*_results.db" | head -1) $(find $OUT/${TAG}_pmc_w -name "*_results.db" | head -1) --steps 6 --json $OUT/${TAG}_hbm_traffic.json > $OUT/${TAG}_hbm_traffic.txt 2>&1
rm -rf $OUT/${TAG}_trace $OUT/${TAG}_pmc_f $OUT/${TAG}_pmc_w
# the same two passes over four eager Stage-1 distillation micro-batches (denoising steps 2, 3, 4, 2): per-family bytes per micro-batch
rocprofv3 --pmc FETCH_SIZE -d $OUT/${TAG}_pmc_tf -- python3 bench.py --mode train --distill-only --train-steps 4 --train-warmup 0 --no-train-graphs --no-cpu-baseline --no-roofline > $OUT/${TAG}_pmc_tf.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/${TAG}_pmc_tw -- python3 bench.py --mode train --distill-only --train-steps 4 --train-warmup 0 --no-train-graphs --no-cpu-baseline --no-roofline > $OUT/${TAG}_pmc_tw.log 2>&1
python3 tools/pmc_traffic.py $(find $OUT/${TAG}_pmc_tf -name "*_results.db" | head -1) $(find $OUT/${TAG}_pmc_tw -name "*_results.db" | head -1) --steps 4 --json $OUT/${TAG}_train_traffic.json > $OUT/${TAG}_train_traffic.txt 2>&1
rm -rf $OUT/${TAG}_pmc_tf $OUT/${TAG}_pmc_tw
tail -3 $OUT/${TAG}_hbm_traffic.txt; head -12 $OUT/${TAG}_bench_kernel_stats.txt
