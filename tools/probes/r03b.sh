set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_kernels.py -q -x -k "folded" 2>&1 | tail -5 > gpurun_out/r03b_tests.log
python tools/autotune_gemm.py --batches 8 --out gpurun_out/gfx950_gemm_r03b.json > gpurun_out/r03b_autotune.log 2>&1
for i in 1 2; do
  AF_TUNE_TABLE=gpurun_out/gfx950_gemm_r03b.json AF_FOLD_LAYERNORM=1 python bench.py --mode denoise --no-cpu-baseline --steps 100 --warmup 10 > gpurun_out/r03b_bench_fold_$i.json 2>gpurun_out/r03b_bench_fold_$i.err
  AF_FOLD_LAYERNORM=0 python bench.py --mode denoise --no-cpu-baseline --steps 100 --warmup 10 > gpurun_out/r03b_bench_nofold_$i.json 2>gpurun_out/r03b_bench_nofold_$i.err
done
