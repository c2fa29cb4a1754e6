set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_orchestration.py -q -x -k "on_device or device_tensors" 2>&1 | tail -8 > gpurun_out/r03g_tests.log
python -m pytest tests/test_hip_train.py -q -x -k "scheduler or reentered or scratch or graph" 2>&1 | tail -25 >> gpurun_out/r03g_tests.log
for i in 1 2; do
  for w in 0 1 2; do
    AF_GEMM3_WPREFETCH=$w python bench.py --mode denoise --no-cpu-baseline --steps 100 --warmup 10 > gpurun_out/r03g_bench_wpf${w}_$i.json 2>gpurun_out/r03g_bench_wpf${w}_$i.err
  done
done
