set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_kernels.py -q -x 2>&1 | tail -4 > gpurun_out/r03j_tests.log
run() { name=$1; shift; env "$@" python bench.py --mode denoise --no-cpu-baseline --steps 100 --warmup 10 > gpurun_out/r03j_$name.json 2>gpurun_out/r03j_$name.err; }
for i in 1 2; do
  run base_$i AF_GEMM3_WPREFETCH=0 AF_NEXT_WEIGHT_HINT=0
  run wpf4_$i AF_NEXT_WEIGHT_HINT=0
  run wpf4_next1024_$i AF_GEMM3_NEXT_KB=1024
  run wpf4_next512_$i AF_GEMM3_NEXT_KB=512
  run wpf4_next2048_$i AF_GEMM3_NEXT_KB=2048
  run wpf0_next1024_$i AF_GEMM3_WPREFETCH=0 AF_GEMM3_NEXT_KB=1024
done
python -m pytest tests/test_hip_unet.py -q -x 2>&1 | tail -4 >> gpurun_out/r03j_tests.log
