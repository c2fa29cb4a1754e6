#!/bin/bash
# r06t: the halo-resident kernel's round-6 forms (256 x 128 tile, 16 x 16-pixel patches) tried on every 3x3 shape of the VAE passes that the table already holds
# (decoder at batch 4 / 1, and whatever the two training legs run: decodes at their batch sizes, the input-gradient backward), taken where > 2 % faster;
# then the training legs with the in-tree table and the new one alternating on this box.
python tools/autotune_gemm.py --try-tile 14 --new-halo-forms --vae --bench-train --bench-train2 --batches 1 --reps 6 --out gpurun_out/r06t_table.json > gpurun_out/r06t_try14.log 2>&1
tail -3 gpurun_out/r06t_try14.log
grep -c "^('" gpurun_out/r06t_try14.log
for i in 1 2 3; do
  for t in tree new; do
    unset AF_TUNE_TABLE
    [ $t = new ] && export AF_TUNE_TABLE=$PWD/gpurun_out/r06t_table.json
    for leg in train train2; do
    python bench.py --mode $leg --no-cpu-baseline --no-roofline --no-reference-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t $leg', d['ms_per_step'], d['config'].get('per_iteration_type'))"
    done
  done
done 2>&1 | tee gpurun_out/r06t_train_ab.txt
