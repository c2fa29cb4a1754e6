# A/B of the pipelined two-chain attention kernel (AF_ATTN_PIPE) at the 64 x 64 level's shape, alternating runs, then its SQ counters
R=$GRAFT_REPO_ROOT
python -m pytest $R/tests/test_hip_kernels.py -m gpu -q -x -p no:cacheprovider -k "attention" 2>&1 | tail -2
for p in 0 1 0 1; do AF_ATTN_PIPE=$p python $R/tools/bench_kernel.py attn 8 4096 4096 8 40 50 2>&1 | tail -1; done
cd /tmp && export TMPDIR=/tmp
export AF_ATTN_PIPE=1
rm -rf /tmp/pa
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_LDS -d /tmp/pa -- python3 $R/tools/bench_kernel.py attn 8 4096 4096 8 40 4 > /tmp/pa.log 2>&1
python3 $R/tools/pmc_kernel.py af_attn $(find /tmp/pa -name "*_results.db")
