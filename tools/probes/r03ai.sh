#!/bin/bash
OUT=$PWD/gpurun_out
timeout 900 python -m pytest tests/test_hip_orchestration.py tests/test_hip_train.py -x -q -m gpu -k "comp or stage2 or full_size or recon" 2>&1 | tail -4 > $OUT/r03ai_tests.log
python3 bench.py --mode train2 --no-cpu-baseline --no-roofline > $OUT/r03ai_train2.json 2> $OUT/r03ai_train2.err
python3 bench.py --mode train --no-cpu-baseline --no-roofline > $OUT/r03ai_train.json 2> $OUT/r03ai_train.err
grep -h -o '"ms_per_step": [0-9.]*' $OUT/r03ai_train2.json $OUT/r03ai_train.json
