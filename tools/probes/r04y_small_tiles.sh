#!/bin/bash
# One more in-step pass of the denoise step (GEGLU on the 128 x 128 tile admitted), then the in-tree table and the new one alternating.
python tools/autotune_instep.py --keep 9 --out gpurun_out/r04y_table.json --log gpurun_out/r04y_instep.log > gpurun_out/r04y_instep.out 2>&1
tail -2 gpurun_out/r04y_instep.out
for i in 1 2 3; do
  for t in tree new; do
    unset AF_TUNE_TABLE
    [ $t = new ] && export AF_TUNE_TABLE=$PWD/gpurun_out/r04y_table.json
    python bench.py --mode denoise --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', d['ms_per_step'])"
  done
done
