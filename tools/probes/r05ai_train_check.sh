#!/bin/bash
# r05ai: occasional 0.2 - 0.4 s in ONE micro-batch of a training leg: allocator segments per micro-batch for several warm-up lengths (hipGraph captures of rarely
# drawn signatures -- the from-noise recon variant, p = 0.4 -- that fall into the timed region allocate their private pools there)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
: > gpurun_out/r05ai_train_check.txt
for w in 12 20 28 36; do
python bench.py --mode train --train-warmup $w --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('warm-up $w: train', d['ms_per_step'], c['host_ms_per_micro_batch_in_timed_region'], 'segments', c['allocator_segments_per_micro_batch_in_timed_region'])" >> gpurun_out/r05ai_train_check.txt
done
for w in 12 20 28; do
python bench.py --mode train2 --train-warmup $w --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('warm-up $w: train2', d['ms_per_step'], c['host_ms_per_micro_batch_in_timed_region'], 'segments', c['allocator_segments_per_micro_batch_in_timed_region'])" >> gpurun_out/r05ai_train_check.txt
done
cat gpurun_out/r05ai_train_check.txt
