set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_orchestration.py -q -x -k "on_device or device_tensors" 2>&1 | tail -15 > gpurun_out/r03f_tests.log
python -m pytest tests/test_hip_train.py -q -x -k "comp_distill or scheduler or reentered or scratch or graph" 2>&1 | tail -25 >> gpurun_out/r03f_tests.log
python -m pytest tests/test_hip_kernels.py -q -x -k "folded" 2>&1 | tail -3 >> gpurun_out/r03f_tests.log
python bench.py --mode train2 --no-cpu-baseline --train-steps 6 --train-warmup 6 > gpurun_out/r03f_bench_train2.json 2>gpurun_out/r03f_bench_train2.err
