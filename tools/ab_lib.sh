#!/bin/bash
# Same-box alternating A/B of the denoise leg between TWO BUILDS of the library:  bash tools/ab_lib.sh <tag> "<extra hipcc flags of arm 0>" [reps]
# Arm 0 is compiled here (on the GPU box) from the tree's sources with the extra flags into /tmp/libadaface_hip_ab.so and loaded through AF_LIB;
# arm 1 is the tree's own libadaface_hip.so.  Writes gpurun_out/<tag>.txt.
cd "$(dirname "$0")/.."
TAG=$1; FLAGS=$2; REPS=${3:-3}
mkdir -p gpurun_out /tmp/ab_build
cp adaface-dev_amd/csrc/*.hip adaface-dev_amd/csrc/*.h adaface-dev_amd/csrc/Makefile /tmp/ab_build/
sed -i 's#../../include/adaface_hip.h#'$PWD'/include/adaface_hip.h#' /tmp/ab_build/af_common.h /tmp/ab_build/Makefile
make -C /tmp/ab_build -j16 EXTRA="$FLAGS" > /tmp/ab_build/build.log 2>&1 || { tail -20 /tmp/ab_build/build.log; exit 1; }
: > gpurun_out/$TAG.txt
for rep in $(seq 1 $REPS); do
  for arm in 0 1; do
    if [ $arm = 0 ]; then export AF_LIB=/tmp/ab_build/libadaface_hip.so; else unset AF_LIB; fi
    python bench.py --mode denoise --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rep $rep arm $arm (0 = built with [$FLAGS], 1 = the tree): ms_per_step', d['ms_per_step'], 'families', d['roofline'].get('families_ms_per_step'))" >> gpurun_out/$TAG.txt
  done
done
unset AF_LIB
cat gpurun_out/$TAG.txt
