#!/bin/bash
# r06h: in-step re-tune of the GEMM table with the round-6 halo kernel (tile 14 is ~15 % faster per launch than when r05w chose between it and the
# tap-by-tap tiles), then old / new table alternating on this box
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1500 python tools/autotune_instep.py --out gpurun_out/r06h_instep_table.json --log gpurun_out/r06h_instep.log > gpurun_out/r06h_instep.out 2>&1
tail -3 gpurun_out/r06h_instep.out
for i in 1 2 3; do
  for t in old new; do
    if [ $t = new ]; then export AF_TUNE_TABLE=$PWD/gpurun_out/r06h_instep_table.json; else unset AF_TUNE_TABLE; fi
    python bench.py --mode denoise --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', d['ms_per_step'])"
  done
done > gpurun_out/r06h_ab.txt
cat gpurun_out/r06h_ab.txt
