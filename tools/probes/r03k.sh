set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_unet.py -q -x -k "live_processor" 2>&1 | tail -40 > gpurun_out/r03k_tests.log
run() { name=$1; shift; env "$@" python bench.py --mode denoise --no-cpu-baseline --steps 100 --warmup 10 > gpurun_out/r03k_$name.json 2>gpurun_out/r03k_$name.err; }
for i in 1 2; do
  run apf0_$i AF_GEMM3_APREFETCH=0
  run apf1_$i AF_GEMM3_APREFETCH=1
  run apf2_$i AF_GEMM3_APREFETCH=2
  run apf4_$i AF_GEMM3_APREFETCH=4
done
python -m pytest tests/test_hip_kernels.py -q -x -k "gemm or conv" 2>&1 | tail -3 >> gpurun_out/r03k_tests.log
