#!/bin/bash
# r06ab: the round-5 review's item 1.ii MEASURED on its cost side: the halo-resident kernel built with -DAF_CONV3H_GN_PROBE normalises every halo piece in LDS
# (silu(x * g[c] * r + s), per-channel factors from memory, pad positions kept zero; one piece per stage, in the L part of the stage after the one that requested it)
# -- a timing probe, its results are not a convolution of the input -- against the tree's kernel, and the GroupNorm + SiLU launch that the fusion would remove.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out /tmp/ab_build
cp adaface-dev_amd/csrc/*.hip adaface-dev_amd/csrc/*.h adaface-dev_amd/csrc/Makefile /tmp/ab_build/
sed -i 's#../../include/adaface_hip.h#'$PWD'/include/adaface_hip.h#' /tmp/ab_build/af_common.h /tmp/ab_build/Makefile
make -C /tmp/ab_build -j16 EXTRA="-DAF_CONV3H_GN_PROBE" > /tmp/ab_build/build.log 2>&1 || { tail -20 /tmp/ab_build/build.log; exit 1; }
: > gpurun_out/r06ab_gn_in_conv_probe.txt
for rep in 1 2; do
  python tools/probes/r06ab_gn_in_conv_probe.py "tree" 2>/dev/null >> gpurun_out/r06ab_gn_in_conv_probe.txt
  AF_LIB=/tmp/ab_build/libadaface_hip.so python tools/probes/r06ab_gn_in_conv_probe.py "with the in-LDS GroupNorm + SiLU pass" 2>/dev/null >> gpurun_out/r06ab_gn_in_conv_probe.txt
done
cat gpurun_out/r06ab_gn_in_conv_probe.txt
