#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
( time python bench.py > gpurun_out/r05z_bench_line.json 2> gpurun_out/r05z_bench.err ) 2> gpurun_out/r05aj_time.txt
tail -3 gpurun_out/r05aj_time.txt; tail -5 gpurun_out/r05z_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05z_bench_line.json').read().strip().splitlines()[-1])
print('denoise', d['ms_per_step'], d['value'], 'frac', d['roofline']['frac'], 'traffic', d['roofline'].get('traffic'), 'cpu', d.get('cpu_baseline'))
for k in ('train','train_stage2'):
    t=d[k]; print(k, t['ms_per_step'], t['value'], t['config']['host_ms_per_micro_batch_in_timed_region'], t['config']['allocator_segments_per_micro_batch_in_timed_region'])
PY
