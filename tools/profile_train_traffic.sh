#!/bin/bash
# The two --pmc passes over four eager Stage-1 distillation micro-batches on their own (the last block of tools/profile_round.sh):
#   bash tools/profile_train_traffic.sh r04z   ->  gpurun_out/r04z_train_traffic.json
set -u
TAG=${1:-r04}
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
rm -rf $OUT/${TAG}_pmc_tf $OUT/${TAG}_pmc_tw
rocprofv3 --pmc FETCH_SIZE -d $OUT/${TAG}_pmc_tf -- python3 bench.py --mode train --distill-only --train-steps 4 --train-warmup 0 --no-train-graphs --no-cpu-baseline --no-roofline > $OUT/${TAG}_pmc_tf.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/${TAG}_pmc_tw -- python3 bench.py --mode train --distill-only --train-steps 4 --train-warmup 0 --no-train-graphs --no-cpu-baseline --no-roofline > $OUT/${TAG}_pmc_tw.log 2>&1
python3 tools/pmc_traffic.py $(find $OUT/${TAG}_pmc_tf -name "*_results.db" | head -1) $(find $OUT/${TAG}_pmc_tw -name "*_results.db" | head -1) --steps 4 --json $OUT/${TAG}_train_traffic.json > $OUT/${TAG}_train_traffic.txt 2>&1
rm -rf $OUT/${TAG}_pmc_tf $OUT/${TAG}_pmc_tw
tail -4 $OUT/${TAG}_train_traffic.txt
