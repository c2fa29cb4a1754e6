#!/bin/bash
# r06u: SQ counters of the attention backward at the 64 x 64 level (attn_bwd_dkv_kernel<3>, attn_bwd_dq_kernel<3>; the round-5 review: "none are on file") next to
# the forward's, counters in their own passes (no trace domains beside them).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06u_attn_bwd_pmc.txt
: > $OUT
rm -rf /tmp/pa /tmp/pb /tmp/pc
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d /tmp/pa -- python3 $R/tools/bench_attn_bwd.py 4 "self 64x64" > /tmp/pa.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE -d /tmp/pb -- python3 $R/tools/bench_attn_bwd.py 4 "self 64x64" > /tmp/pb.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_SCA -d /tmp/pc -- python3 $R/tools/bench_attn_bwd.py 4 "self 64x64" > /tmp/pc.log 2>&1
tail -1 /tmp/pa.log >> $OUT
for k in attn_bwd_dkv_kernel attn_bwd_dq_kernel attn_delta af_attn2_kernel af_attn_kernel; do
  echo "=== $k (self-attention 64 x 64, batch 4, 8 heads of 40)" >> $OUT
  python3 $R/tools/pmc_kernel.py "$k" $(find /tmp/pa /tmp/pb /tmp/pc -name "*_results.db") >> $OUT 2>&1
done
cat $OUT
