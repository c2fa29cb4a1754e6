#!/bin/bash
# r05ap: host-side profile of a Stage-2 leg (cProfile, top functions by own time): where the ~250 ms of host time per micro-batch go on a slow host
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m cProfile -o gpurun_out/r05ap_train2.prof bench.py --mode train2 --train-steps 8 --train-warmup 8 --no-cpu-baseline --no-roofline > gpurun_out/r05ap_line.json 2> gpurun_out/r05ap.err
python - <<'PY' > gpurun_out/r05ap_pyprofile.txt
import pstats
p = pstats.Stats('gpurun_out/r05ap_train2.prof')
p.sort_stats('tottime').print_stats(45)
PY
head -75 gpurun_out/r05ap_pyprofile.txt | cut -c1-180
