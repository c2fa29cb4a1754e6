R=$GRAFT_REPO_ROOT
cd $R
python tools/bench_colmix.py > gpurun_out/r05e_colmix.txt 2>&1
cat gpurun_out/r05e_colmix.txt
timeout 1500 python -m pytest tests -m gpu -q --durations=40 2>&1 | tail -70 > gpurun_out/r05e_full_gpu.log
cat gpurun_out/r05e_full_gpu.log
