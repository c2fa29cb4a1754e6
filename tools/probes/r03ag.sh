#!/bin/bash
OUT=$PWD/gpurun_out
for i in 1 2 3; do
  AF_TUNE_TABLE=$PWD/tools/probes/gfx950_gemm_before_r03ad.json python3 bench.py --mode denoise --no-cpu-baseline --no-roofline > $OUT/r03ag_base_$i.json 2> $OUT/r03ag_base_$i.err
  python3 bench.py --mode denoise --no-cpu-baseline --no-roofline > $OUT/r03ag_new_$i.json 2> $OUT/r03ag_new_$i.err
done
grep -h -o '"ms_per_step": [0-9.]*' $OUT/r03ag_base_*.json $OUT/r03ag_new_*.json
