#!/bin/bash
# r05u: in-step A/B of the ping-pong halo kernel (AF_CONV3H_PP) and the pipelined attention (AF_ATTN_PIPE), alternating runs of the denoise leg on one box
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
: > gpurun_out/r05u_instep_ab.txt
for rep in 1 2; do
  for cfg in "0 0" "1 0" "0 1" "1 1"; do
    set -- $cfg
    AF_CONV3H_PP=$1 AF_ATTN_PIPE=$2 python bench.py --mode denoise --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rep $rep conv3h_pp $1 attn_pipe $2: ms_per_step', d['ms_per_step'], 'families', d['roofline'].get('families_ms_per_step'))" >> gpurun_out/r05u_instep_ab.txt
  done
done
cat gpurun_out/r05u_instep_ab.txt
