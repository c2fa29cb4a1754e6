#!/bin/bash
OUT=$PWD/gpurun_out
for i in 1 2; do
  AF_CONV_TILE14=0 python3 bench.py --mode denoise --no-cpu-baseline --no-roofline > $OUT/r03ac_base_$i.json 2> $OUT/r03ac_base_$i.err
  python3 bench.py --mode denoise --no-cpu-baseline --no-roofline > $OUT/r03ac_t14_$i.json 2> $OUT/r03ac_t14_$i.err
done
AF_CONV_TILE14=0 python3 bench.py --mode train --distill-only --no-cpu-baseline --no-roofline > $OUT/r03ac_train_base.json 2> $OUT/r03ac_train_base.err
python3 bench.py --mode train --distill-only --no-cpu-baseline --no-roofline > $OUT/r03ac_train_t14.json 2> $OUT/r03ac_train_t14.err
grep -h -o '"ms_per_step": [0-9.]*' $OUT/r03ac_base_*.json $OUT/r03ac_t14_*.json $OUT/r03ac_train_base.json $OUT/r03ac_train_t14.json
python -m pytest tests -x -q -m gpu 2>&1 | tail -8
