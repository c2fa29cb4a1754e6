#!/bin/bash
# r05ap: host-side profile of a Stage-2 leg (cProfile): own time overall, cumulative time of this package's functions
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m cProfile -o /tmp/r05ap_train2.prof bench.py --mode train2 --train-steps 8 --train-warmup 8 --no-cpu-baseline --no-roofline > gpurun_out/r05ap_line.json 2> gpurun_out/r05ap.err
python - <<'PY' > gpurun_out/r05ap_pyprofile.txt
import pstats
p = pstats.Stats('/tmp/r05ap_train2.prof')
p.sort_stats('tottime').print_stats(30)
p.sort_stats('cumulative').print_stats(r'adaface', 70)
PY
python -c "
import json
d=json.loads(open('gpurun_out/r05ap_line.json').read().strip().splitlines()[-1]); print('train2 under cProfile', d['ms_per_step'])"
sed -n 45,130p gpurun_out/r05ap_pyprofile.txt | cut -c1-170
