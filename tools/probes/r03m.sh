set -u
cd $GRAFT_REPO_ROOT
python bench.py --mode train --no-cpu-baseline --distill-only > gpurun_out/r03m_bench_train_distill.json 2>gpurun_out/r03m_bench_train_distill.err
AF_GEMM3_WPREFETCH=0 python bench.py --mode train --no-cpu-baseline --distill-only --no-roofline > gpurun_out/r03m_bench_train_distill_nopf.json 2>gpurun_out/r03m_bench_train_distill_nopf.err
python tools/autotune_gemm.py --batches 8 --cold-weights --retune --out gpurun_out/gfx950_gemm_r03m_cold.json > gpurun_out/r03m_autotune_cold.log 2>&1
for i in 1 2; do
  python bench.py --mode denoise --no-cpu-baseline --steps 100 --warmup 10 > gpurun_out/r03m_bench_hot_$i.json 2>gpurun_out/r03m_bench_hot_$i.err
  AF_TUNE_TABLE=gpurun_out/gfx950_gemm_r03m_cold.json python bench.py --mode denoise --no-cpu-baseline --steps 100 --warmup 10 > gpurun_out/r03m_bench_cold_$i.json 2>gpurun_out/r03m_bench_cold_$i.err
done
