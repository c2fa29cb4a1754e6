#!/bin/bash
# Same-box alternating A/B of the denoise leg under ONE environment switch:  bash tools/ab_denoise.sh <tag> <ENV_VAR> [reps] [extra bench args]
# Writes gpurun_out/<tag>.txt: ms per step and the per-family instrumented times of every run, arms alternating (0, 1, 0, 1, ...).
cd "$(dirname "$0")/.."
TAG=$1; VAR=$2; REPS=${3:-3}; shift 3 2>/dev/null
mkdir -p gpurun_out
: > gpurun_out/$TAG.txt
for rep in $(seq 1 $REPS); do
  for v in 0 1; do
    env $VAR=$v python bench.py --mode denoise --steps 100 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rep $rep $VAR=$v: ms_per_step', d['ms_per_step'], 'families', d['roofline'].get('families_ms_per_step'))" >> gpurun_out/$TAG.txt
  done
done
cat gpurun_out/$TAG.txt
