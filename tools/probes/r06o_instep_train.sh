#!/bin/bash
# r06o: in-step tuning of the training legs on the round-6 kernels (the halo-resident kernel ~15 % faster and with the K tail in its scope), denoise shapes
# protected; then the legs with the in-tree table and the new one alternating on this box.
P=profiles/r05w_instep_retune.log,profiles/r06h_instep_retune.log,profiles/r06l_instep_retune.log
python tools/autotune_instep.py --leg distill --keep 4 --reps 3 --protect $P --out gpurun_out/r06o_t1.json --log gpurun_out/r06o_instep_distill.log > gpurun_out/r06o_distill.out 2>&1
tail -2 gpurun_out/r06o_distill.out
AF_TUNE_TABLE=$PWD/gpurun_out/r06o_t1.json python tools/autotune_instep.py --leg train2 --keep 4 --reps 2 --protect $P,gpurun_out/r06o_instep_distill.log --out gpurun_out/r06o_t2.json --log gpurun_out/r06o_instep_train2.log > gpurun_out/r06o_train2.out 2>&1
tail -2 gpurun_out/r06o_train2.out
AF_TUNE_TABLE=$PWD/gpurun_out/r06o_t2.json python tools/autotune_instep.py --leg recon --keep 4 --reps 2 --protect $P,gpurun_out/r06o_instep_distill.log,gpurun_out/r06o_instep_train2.log --out gpurun_out/r06o_t3.json --log gpurun_out/r06o_instep_recon.log > gpurun_out/r06o_recon.out 2>&1
tail -2 gpurun_out/r06o_recon.out
for i in 1 2 3; do
  for t in tree new; do
    unset AF_TUNE_TABLE
    [ $t = new ] && export AF_TUNE_TABLE=$PWD/gpurun_out/r06o_t3.json
    for leg in train train2; do
    python bench.py --mode $leg --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t $leg', d['ms_per_step'], d['config'].get('per_iteration_type'))"
    done
  done
done
