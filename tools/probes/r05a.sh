# round 5, first GPU job: MFMA pair probe, robust GroupNorm statistics + guards, MFMA colmix, GEMM PMC counters
R=$GRAFT_REPO_ROOT
cd $R
(cd tools/probes && ./r05a_mfma_k16_probe) > gpurun_out/r05a_mfma_probe.txt 2>&1
cat gpurun_out/r05a_mfma_probe.txt
timeout 900 python -m pytest tests/test_hip_kernels.py -q -k "groupnorm or statistics or gn_proj or tile14" 2>&1 | tail -15 > gpurun_out/r05a_gn_tests.txt
cat gpurun_out/r05a_gn_tests.txt
timeout 600 python -m pytest tests/test_hip_capture_graph.py -q 2>&1 | tail -8 > gpurun_out/r05a_capture_tests.txt
cat gpurun_out/r05a_capture_tests.txt
python tools/bench_colmix.py > gpurun_out/r05a_colmix.txt 2>&1
cat gpurun_out/r05a_colmix.txt
bash tools/probes/r05b_gemm_pmc.sh > /dev/null 2>&1
cat gpurun_out/r05b_gemm_pmc.txt
