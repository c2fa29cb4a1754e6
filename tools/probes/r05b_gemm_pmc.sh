# Round 5: SQ counters of the CURRENT GEMM-family kernels (the only GEMM counters on file were round 1's, of a kernel that no longer runs).
#   af_gemm3w_kernel<9,2,4,5,0,2>  tile 7, 128 x 320 whole-line 3x3      conv 8 32 32 640 640 (split 2)  and the 8x8 level (split 16)
#   af_gemm3w_kernel<1,1,4,2,0,2>  tile 16, 64 x 128 (three workgroups per CU): the short-K 1x1 GEMMs     gemm 8192 640 640 / 2048 1280 1280
#   af_conv3h_kernel               tile 14, halo-resident 3x3                                             conv 8 64 64 320 320
# Counters in their own passes (no --kernel-trace mixing beyond what --pmc implies); warm operands (hipGraph of 20 launches, bench_kernel.py).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05b_gemm_pmc.txt
: > $OUT
run() {   # name-substring, bench_kernel args...
  local pat=$1; shift
  rm -rf /tmp/pa /tmp/pb
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d /tmp/pa -- python3 $R/tools/bench_kernel.py "$@" > /tmp/pa.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE -d /tmp/pb -- python3 $R/tools/bench_kernel.py "$@" > /tmp/pb.log 2>&1
  echo "=== bench_kernel.py $* (kernel ~ $pat)" >> $OUT
  tail -1 /tmp/pa.log >> $OUT
  python3 $R/tools/pmc_kernel.py "$pat" $(find /tmp/pa /tmp/pb -name "*_results.db") >> $OUT 2>&1
}
run af_gemm3w conv 8 32 32 640 640 7 2
run af_gemm3w conv 8 8 8 1280 1280 7 16
run af_gemm3w conv 8 16 16 1280 1280 7 4
run af_gemm3w gemm 8192 640 640 16 1
run af_gemm3w gemm 2048 1280 1280 16 1
run af_conv3h conv 8 64 64 320 320 14 1
cat $OUT
