#!/bin/bash
OUT=$PWD/gpurun_out
for i in 1 2 3; do
  AF_FUSE_XATTN=0 python3 bench.py --mode denoise --no-cpu-baseline --no-roofline > $OUT/r03av_base_$i.json 2>/dev/null
  AF_FUSE_XATTN=1 python3 bench.py --mode denoise --no-cpu-baseline --no-roofline > $OUT/r03av_fused_$i.json 2>/dev/null
done
grep -h -o '"ms_per_step": [0-9.]*' $OUT/r03av_base_*.json $OUT/r03av_fused_*.json
