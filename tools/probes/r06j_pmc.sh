#!/bin/bash
# r06j: SQ counters before / after the vector-instruction diet of the halo-resident 3x3 kernel (tile 19 = round-5 loop, tile 14 = round-6 loop) and of
# the two-chain self-attention kernel with 4- and 8-wave workgroups.  Counters in their own passes; warm operands (hipGraph of 20 launches, bench_kernel.py).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06j_pmc.txt
: > $OUT
run() {   # name-substring, bench_kernel args...
  local pat=$1; shift
  rm -rf /tmp/pa /tmp/pb /tmp/pc
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d /tmp/pa -- python3 $R/tools/bench_kernel.py "$@" > /tmp/pa.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE -d /tmp/pb -- python3 $R/tools/bench_kernel.py "$@" > /tmp/pb.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM -d /tmp/pc -- python3 $R/tools/bench_kernel.py "$@" > /tmp/pc.log 2>&1
  echo "=== [AF_ATTN_NW=${AF_ATTN_NW:-}] bench_kernel.py $* (kernel ~ $pat)" >> $OUT
  tail -1 /tmp/pa.log >> $OUT
  python3 $R/tools/pmc_kernel.py "$pat" $(find /tmp/pa /tmp/pb /tmp/pc -name "*_results.db") >> $OUT 2>&1
}
run af_conv3h conv 8 64 64 320 320 19 1
run af_conv3h conv 8 64 64 320 320 14 1
run af_conv3h conv 8 32 32 640 640 19 2
run af_conv3h conv 8 32 32 640 640 14 2
export AF_ATTN_NW=4
run af_attn2 attn 8 4096 4096 8 40
export AF_ATTN_NW=8
run af_attn2 attn 8 4096 4096 8 40
cat $OUT
