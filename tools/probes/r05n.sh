R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_hip_train.py tests/test_hip_capture_graph.py tests/test_hip_orchestration.py -q -x 2>&1 | tail -4 > gpurun_out/r05n_tests.txt
cat gpurun_out/r05n_tests.txt
for rep in 1 2; do
  for mode in train2 train; do
    python bench.py --mode $mode --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$mode', d['ms_per_step'], d['config'].get('per_iteration_type'))"
  done
done > gpurun_out/r05n_bench.txt
cat gpurun_out/r05n_bench.txt
python tools/torch_op_census.py --leg train2 2>/dev/null | head -28
