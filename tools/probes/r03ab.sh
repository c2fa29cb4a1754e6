#!/bin/bash
OUT=$PWD/gpurun_out
for i in 1 2; do
  python3 bench.py --mode denoise --no-cpu-baseline --no-roofline > $OUT/r03ab_base_$i.json 2> $OUT/r03ab_base_$i.err
  AF_CONV_TILE14=1 python3 bench.py --mode denoise --no-cpu-baseline --no-roofline > $OUT/r03ab_t14_$i.json 2> $OUT/r03ab_t14_$i.err
done
grep -h -o '"ms_per_step": [0-9.]*' $OUT/r03ab_base_*.json $OUT/r03ab_t14_*.json
