#!/bin/bash
# Second in-step pass (wider short list; the halo-resident kernel now leaves GroupNorm statistics too) starting from the in-tree table; then the in-tree
# table and the new one alternating in the denoise leg, same box.
python tools/autotune_instep.py --keep 9 --out gpurun_out/r04w_instep_table2.json --log gpurun_out/r04w_instep2.log > gpurun_out/r04w_instep2.out 2>&1
tail -2 gpurun_out/r04w_instep2.out
for i in 1 2 3; do
  for t in tree new2; do
    unset AF_TUNE_TABLE
    [ $t = new2 ] && export AF_TUNE_TABLE=$PWD/gpurun_out/r04w_instep_table2.json
    python bench.py --mode denoise --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', d['ms_per_step'])"
  done
done
