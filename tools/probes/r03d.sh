set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_kernels.py -q -x -k "folded" 2>&1 | tail -5 > gpurun_out/r03d_tests.log
python tools/bench_lnfold.py 20 > gpurun_out/r03d_lnfold.txt 2>&1
python -m pytest tests/test_hip_train.py -q -x -k "graph or scratch or reentered" 2>&1 | tail -15 >> gpurun_out/r03d_tests.log
for i in 1 2; do
  AF_FOLD_LAYERNORM=1 python bench.py --mode denoise --no-cpu-baseline --steps 100 --warmup 10 > gpurun_out/r03d_bench_fold_$i.json 2>gpurun_out/r03d_bench_fold_$i.err
  AF_FOLD_LAYERNORM=0 python bench.py --mode denoise --no-cpu-baseline --steps 100 --warmup 10 > gpurun_out/r03d_bench_nofold_$i.json 2>gpurun_out/r03d_bench_nofold_$i.err
done
