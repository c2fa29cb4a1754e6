cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
for P in 0 1; do
  export AF_ATTN_PIPE=$P
  rm -rf /tmp/pa /tmp/pb /tmp/pc
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d /tmp/pa -- python3 $R/tools/bench_kernel.py attn 8 4096 4096 8 40 4 > /tmp/pa.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_SALU -d /tmp/pb -- python3 $R/tools/bench_kernel.py attn 8 4096 4096 8 40 4 > /tmp/pb.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_FLAT GRBM_GUI_ACTIVE SQ_INST_LEVEL_LDS -d /tmp/pc -- python3 $R/tools/bench_kernel.py attn 8 4096 4096 8 40 4 > /tmp/pc.log 2>&1
  echo "=== AF_ATTN_PIPE=$P"
  python3 $R/tools/pmc_kernel.py af_attn $(find /tmp/pa /tmp/pb /tmp/pc -name "*_results.db")
  tail -2 /tmp/pa.log /tmp/pc.log | grep -v "^$"
done
