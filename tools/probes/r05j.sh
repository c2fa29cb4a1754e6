R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_hip_kernels.py -q -k "groupnorm or statistics or gn_proj" 2>&1 | tail -30 > gpurun_out/r05j_tests.txt
cat gpurun_out/r05j_tests.txt | cut -c1-200
