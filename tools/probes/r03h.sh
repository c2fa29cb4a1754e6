set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_train.py -q -x -k "scheduler or reentered" 2>&1 | tail -12 > gpurun_out/r03h_tests.log
for i in 1 2; do
  for w in 0 1 2; do
    AF_GEMM3_WPREFETCH=$w python bench.py --mode denoise --no-cpu-baseline --steps 100 --warmup 10 > gpurun_out/r03h_bench_wpf${w}_$i.json 2>gpurun_out/r03h_bench_wpf${w}_$i.err
  done
done
python bench.py --mode train --no-cpu-baseline > gpurun_out/r03h_bench_train.json 2>gpurun_out/r03h_bench_train.err
python -m pytest tests/test_hip_train.py -q -x -k "full_size_stage1" 2>&1 | tail -12 >> gpurun_out/r03h_tests.log
