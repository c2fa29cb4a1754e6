#!/bin/bash
# r06t (second part): the training legs with the round's earlier table and without the untabled-shape default (old) against the tree (new), alternating
for i in 1 2 3; do
  for t in old new; do
    unset AF_TUNE_TABLE AF_VAE_HALO_DEFAULT
    [ $t = old ] && export AF_TUNE_TABLE=$PWD/tools/probes/tmp_old_table.json   # (git show <parent>:adaface-dev_amd/tuning/gfx950_gemm.json > that file before the run) AF_VAE_HALO_DEFAULT=0
    for leg in train train2; do
    python bench.py --mode $leg --no-cpu-baseline --no-roofline --no-reference-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t $leg', d['ms_per_step'], d['config'].get('per_iteration_type'))"
    done
  done
done 2>&1 | tee gpurun_out/r06t_train_ab2.txt
