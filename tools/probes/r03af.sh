#!/bin/bash
OUT=$PWD/gpurun_out
timeout 300 python tools/probes/r03ae_ln_geglu_fault.py > $OUT/r03af_fault.txt 2>&1
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_unet.py -x -q -m gpu 2>&1 | tail -4 > $OUT/r03af_tests.log
for i in 1 2; do
  python3 bench.py --mode denoise --no-cpu-baseline --no-roofline > $OUT/r03af_bench_$i.json 2> $OUT/r03af_bench_$i.err
done
grep -h -o '"ms_per_step": [0-9.]*' $OUT/r03af_bench_*.json
