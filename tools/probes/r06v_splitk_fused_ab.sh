#!/bin/bash
# r06v: the in-launch split-K reduction of the register-staged tiles chosen by size (ops.SPLITK_FUSED_BYTES = 1 MB of slabs; round-5 review 5a: "re-measure on a
# slow host") in the Stage-2 leg, three alternating pairs; the host's speed is in the first line (a Python loop + the leg's own host-side time per micro-batch).
python - <<'PY' | tee gpurun_out/r06v_splitk_fused_ab.txt
import time
t=time.perf_counter(); s=0
for i in range(5_000_000): s+=i
print("host speed: 5M-iteration python loop %.2f s" % (time.perf_counter()-t))
PY
for i in 1 2 3; do
  for v in 0 1048576; do
    AF_SPLITK_FUSED_BYTES=$v python bench.py --mode train2 --no-cpu-baseline --no-roofline --no-reference-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('AF_SPLITK_FUSED_BYTES=$v train2', d['ms_per_step'])"
  done
done 2>&1 | tee -a gpurun_out/r06v_splitk_fused_ab.txt
