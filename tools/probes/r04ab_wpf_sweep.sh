#!/bin/bash
# Weight-tile L2 prefetch depth / cooperation width under the in-step-tuned table: whole denoise leg per setting, defaults first and last.
for cfg in "4 32" "0 32" "2 32" "8 32" "4 8" "4 16" "4 64" "8 64" "4 32"; do
  set -- $cfg
  AF_GEMM3_WPREFETCH=$1 AF_GEMM3_WPF_COOP=$2 python bench.py --mode denoise --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wpf=$1 coop=$2', d['ms_per_step'])"
done
