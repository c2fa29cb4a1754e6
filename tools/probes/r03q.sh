set -u
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_hip_kernels.py -q -x -k "ff_fused" 2>&1 | tail -25 > gpurun_out/r03q_tests.log
timeout 300 python tools/bench_ff.py 20 > gpurun_out/r03q_bench_ff.txt 2>&1
for i in 1 2; do
  AF_FUSE_FF=0 timeout 300 python bench.py --mode denoise --no-cpu-baseline --steps 100 --warmup 10 > gpurun_out/r03q_bench_nofuse_$i.json 2>gpurun_out/r03q_bench_nofuse_$i.err
  AF_FUSE_FF=1 timeout 300 python bench.py --mode denoise --no-cpu-baseline --steps 100 --warmup 10 > gpurun_out/r03q_bench_fuse_$i.json 2>gpurun_out/r03q_bench_fuse_$i.err
done
timeout 600 python -m pytest tests/test_hip_unet.py -q -x 2>&1 | tail -5 >> gpurun_out/r03q_tests.log
