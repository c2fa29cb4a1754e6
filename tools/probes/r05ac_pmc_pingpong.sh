# r05ac: SQ counters of the ping-pong halo kernel (tile 14) beside the tap-by-tap tile 7 on the same shapes (r05b had the lock-step tile 14: pipe busy 0.337)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05ac_pmc_pingpong.txt
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
run af_conv3h conv 8 64 64 320 320 14 1
run af_gemm3w conv 8 64 64 320 320 7 1
run af_conv3h conv 8 32 32 640 640 14 2
run af_conv3h conv 8 16 16 1280 1280 14 4
cat $OUT | cut -c1-260
