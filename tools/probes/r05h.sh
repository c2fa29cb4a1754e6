R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_hip_unet.py tests/test_hip_orchestration.py tests/test_hip_capture_graph.py -q -x 2>&1 | tail -15 > gpurun_out/r05h_tests.txt
cat gpurun_out/r05h_tests.txt
timeout 900 python -m pytest tests/test_hip_train.py -q -x -k "comp_distill or recon or stage2 or stall or trunk" 2>&1 | tail -8 >> gpurun_out/r05h_tests.txt
tail -8 gpurun_out/r05h_tests.txt
: > gpurun_out/r05h_ab.txt
for rep in 1 2; do
 for cfg in "1 1" "0 0" "1 0" "0 1"; do
  set -- $cfg
  for mode in train train2; do
    AF_SHARE_TRUNK=$1 AF_BATCH_UNCOND=$2 python bench.py --mode $mode --no-cpu-baseline --no-roofline > /tmp/b.json 2>/dev/null
    python - "$rep" "$1" "$2" "$mode" >> gpurun_out/r05h_ab.txt <<'PY'
import json,sys
d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
print(f"rep {sys.argv[1]} share_trunk {sys.argv[2]} batch_uncond {sys.argv[3]} {sys.argv[4]}: ms_per_step {d['ms_per_step']} value {d['value']} {d['config'].get('per_iteration_type')}")
PY
  done
 done
done
cat gpurun_out/r05h_ab.txt
