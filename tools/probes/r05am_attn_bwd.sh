#!/bin/bash
# r05am: attention backward with the MFMA groups' fragments requested ahead (the ISA had one ds_read + lgkmcnt(0) in front of EVERY MFMA): same-box comparison
# against the previous library (adaface-dev_amd/csrc/lib_old_attn_bwd.so, built from the parent commit)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
L=adaface-dev_amd/csrc
cp $L/libadaface_hip.so $L/lib_new_tmp.so
: > gpurun_out/r05am_attn_bwd.txt
for rep in 1 2; do
  for which in old new; do
    if [ $which = old ]; then cp $L/lib_old_attn_bwd.so $L/libadaface_hip.so; else cp $L/lib_new_tmp.so $L/libadaface_hip.so; fi
    echo "--- $which, batch 4" >> gpurun_out/r05am_attn_bwd.txt
    python tools/bench_attn_bwd.py 4 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05am_attn_bwd.txt
  done
done
cp $L/lib_new_tmp.so $L/libadaface_hip.so
for which in old new; do
  if [ $which = old ]; then cp $L/lib_old_attn_bwd.so $L/libadaface_hip.so; else cp $L/lib_new_tmp.so $L/libadaface_hip.so; fi
  python bench.py --mode train --distill-only --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which distill-only leg', d['ms_per_step'])" >> gpurun_out/r05am_attn_bwd.txt
done
cp $L/lib_new_tmp.so $L/libadaface_hip.so
cat gpurun_out/r05am_attn_bwd.txt
