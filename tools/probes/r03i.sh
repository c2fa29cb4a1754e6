set -u
cd $GRAFT_REPO_ROOT
run() { # name env...
  name=$1; shift
  env "$@" python bench.py --mode denoise --no-cpu-baseline --steps 100 --warmup 10 > gpurun_out/r03i_$name.json 2>gpurun_out/r03i_$name.err
}
for i in 1 2; do
  run wpf0_$i AF_GEMM3_WPREFETCH=0
  run wpf2_$i AF_GEMM3_WPREFETCH=2
  run wpf4_$i AF_GEMM3_WPREFETCH=4
  run wpf8_$i AF_GEMM3_WPREFETCH=8
  run wpf4c8_$i AF_GEMM3_WPREFETCH=4 AF_GEMM3_WPF_COOP=8
  run wpf8c16_$i AF_GEMM3_WPREFETCH=8 AF_GEMM3_WPF_COOP=16
done
