#!/bin/bash
# r06l: in-step re-tune after the halo-resident kernel learnt the K tail (tile 14 is now a candidate for the conv2 + shortcut launches)
# tap-by-tap tiles), then old / new table alternating on this box
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1500 python tools/autotune_instep.py --out gpurun_out/r06l_instep_table.json --log gpurun_out/r06l_instep.log > gpurun_out/r06l_instep.out 2>&1
tail -3 gpurun_out/r06l_instep.out
for i in 1 2 3; do
  for t in old new; do
    if [ $t = new ]; then export AF_TUNE_TABLE=$PWD/gpurun_out/r06l_instep_table.json; else unset AF_TUNE_TABLE; fi
    python bench.py --mode denoise --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', d['ms_per_step'])"
  done
done > gpurun_out/r06l_ab.txt
cat gpurun_out/r06l_ab.txt
