#!/bin/bash
# Small tiles with the transposed-V split: tests, one more in-step pass of the denoise step, then the in-tree table and the new one alternating.
python -m pytest tests/test_hip_kernels.py -m gpu -q -x -p no:cacheprovider -k "split_transposed or folded_layernorm" 2>&1 | tail -2
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
