#!/bin/bash
# Where the HOST time of a Stage-2 micro-batch goes (the leg is host-bound: wall == host time per micro-batch, GPU busy ~57 %):
# cProfile over the whole leg, cumulative times of this package's functions.  Usage: bash tools/r04p_stage2_pyprofile.sh [train2|train]
mode=${1:-train2}
mkdir -p gpurun_out
python -m cProfile -o gpurun_out/r04p_${mode}.prof bench.py --mode $mode --train-steps 8 --train-warmup 4 --no-cpu-baseline --no-roofline > gpurun_out/r04p_${mode}_line.json 2> gpurun_out/r04p_${mode}.err
python - <<PY > gpurun_out/r04p_${mode}_pyprofile.txt
import pstats
p = pstats.Stats("gpurun_out/r04p_${mode}.prof")
p.sort_stats("cumulative").print_stats(r"adaface|bench", 90)
p.sort_stats("tottime").print_stats(60)
PY
rm -f gpurun_out/r04p_${mode}.prof
tail -c 600 gpurun_out/r04p_${mode}_line.json
