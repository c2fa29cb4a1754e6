R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_hip_kernels.py -q -k "drops_producer or outside_tile14 or prefetch" 2>&1 | tail -5 > gpurun_out/r05d_tests.txt
timeout 600 python -m pytest tests/test_hip_capture_graph.py -q -k explicit 2>&1 | tail -5 >> gpurun_out/r05d_tests.txt
cat gpurun_out/r05d_tests.txt
python tools/bench_colmix.py > gpurun_out/r05d_colmix.txt 2>&1
cat gpurun_out/r05d_colmix.txt
bash tools/probes/r05c_prefetch_ab.sh
