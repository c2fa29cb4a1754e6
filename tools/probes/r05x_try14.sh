#!/bin/bash
# r05x: the ping-pong halo kernel (tile 14) against the table's choice on every 3x3 shape of the training legs / other batch sizes / VAE (warm, > 2 % rule);
# the denoise step's shapes were decided inside the step (r05w) and are left alone
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 2400 python tools/autotune_gemm.py --batches 8,2,4,6,16 --bench-train --bench-train2 --try-tile 14 --protect profiles/r05w_instep_retune.log --out gpurun_out/r05x_table.json > gpurun_out/r05x_try14.out 2>&1
tail -5 gpurun_out/r05x_try14.out
for i in 1 2; do
  for t in old new; do
    if [ $t = new ]; then export AF_TUNE_TABLE=$PWD/gpurun_out/r05x_table.json; else unset AF_TUNE_TABLE; fi
    python bench.py --mode train --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t train', d['ms_per_step'], d['config'].get('per_iteration_type'))"
    python bench.py --mode train2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t train2', d['ms_per_step'])"
  done
done > gpurun_out/r05x_ab.txt
cat gpurun_out/r05x_ab.txt
