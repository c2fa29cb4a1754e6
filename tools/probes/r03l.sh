set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
python -m pytest tests/test_hip_unet.py -q -x -k "live_processor" 2>&1 | tail -3 > gpurun_out/r03l_tests.log
rm -rf $OUT/r03l_trace
rocprofv3 --kernel-trace --stats -d $OUT/r03l_trace -- python3 bench.py --mode denoise --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $OUT/r03l_trace.log 2>&1
DB=$(find $OUT/r03l_trace -name "*_results.db" | head -1)
N=$(python3 - <<PY
import sqlite3
c = sqlite3.connect("$DB")
names = [r[0] for r in c.execute("select name from kernels order by start")]
idx = [i for i, n in enumerate(names) if "cfg_ddim" in n]
print((idx[-1] - idx[-2]) * 10)
PY
)
python3 tools/rocpd_summary.py $DB --last $N > $OUT/r03l_bench_kernel_stats.txt
rm -rf $OUT/r03l_trace
