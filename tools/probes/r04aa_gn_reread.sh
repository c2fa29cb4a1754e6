#!/bin/bash
# One-launch re-reading GroupNorm for few batch items (AF_GN_REREAD_MAX_B): tests, then the legs with it off / at 2 / at 8, alternating on this box.
python -m pytest tests/test_hip_kernels.py -m gpu -q -x -p no:cacheprovider -k "test_groupnorm" 2>&1 | tail -2
python -m pytest tests/test_hip_train.py -m gpu -q -x -p no:cacheprovider -k "comp_distill_iteration_reduced_width or stage1" 2>&1 | tail -2
for i in 1 2; do
  for v in 0 2 8; do
    for leg in train2 train denoise; do
    AF_GN_REREAD_MAX_B=$v python bench.py --mode $leg --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('reread_max_b=$v $leg', d['ms_per_step'], d['config'].get('per_iteration_type'))"
    done
  done
done
