R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_hip_capture_graph.py -q 2>&1 | tail -12 > gpurun_out/r05p_tests.txt
cat gpurun_out/r05p_tests.txt | cut -c1-200
python tools/bench_colmix.py 4 8 4096 97 40 > gpurun_out/r05p_explicit_mfma.txt 2>/dev/null
python tools/bench_colmix.py 1 8 4096 97 40 >> gpurun_out/r05p_explicit_mfma.txt 2>/dev/null
AF_XATTN_EXPLICIT_MFMA=0 python tools/bench_colmix.py 4 8 4096 97 40 > gpurun_out/r05p_explicit_rows.txt 2>/dev/null
AF_XATTN_EXPLICIT_MFMA=0 python tools/bench_colmix.py 1 8 4096 97 40 >> gpurun_out/r05p_explicit_rows.txt 2>/dev/null
echo "--- MFMA forms"; cat gpurun_out/r05p_explicit_mfma.txt; echo "--- wave-per-row forms"; cat gpurun_out/r05p_explicit_rows.txt
for rep in 1 2; do for v in 1 0; do for mode in train train2; do
  AF_XATTN_EXPLICIT_MFMA=$v python bench.py --mode $mode --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rep $rep mfma $v $mode', d['ms_per_step'], d['config'].get('per_iteration_type'))"
done; done; done > gpurun_out/r05p_ab.txt
cat gpurun_out/r05p_ab.txt
