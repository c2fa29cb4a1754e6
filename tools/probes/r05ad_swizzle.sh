# r05ad: halo kernel with the pixel & 7 swizzle: tests, timing against tile 7, LDS conflict counters
R=$GRAFT_REPO_ROOT
cd $R
bash tools/probes/r05r_conv3h_pp.sh > /dev/null 2>&1
cp gpurun_out/r05r_conv3h_pp.txt gpurun_out/r05ad_timing.txt; cp gpurun_out/r05r_tests.txt gpurun_out/r05ad_tests.txt
cat gpurun_out/r05ad_tests.txt gpurun_out/r05ad_timing.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pb
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE -d /tmp/pb -- python3 $R/tools/bench_kernel.py conv 8 64 64 320 320 14 1 > /tmp/pb.log 2>&1
python3 $R/tools/pmc_kernel.py af_conv3h $(find /tmp/pb -name "*_results.db") > $R/gpurun_out/r05ad_pmc.txt 2>&1
cat $R/gpurun_out/r05ad_pmc.txt
