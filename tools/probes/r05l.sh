R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_hip_kernels.py -q -k "tile18 or folded_layernorm_rows" 2>&1 | tail -12 > gpurun_out/r05l_tests.txt
cat gpurun_out/r05l_tests.txt | cut -c1-220
: > gpurun_out/r05l_small_gemm.txt
for shp in "388 768 768" "388 3072 768" "388 768 3072" "1552 768 768" "1552 3072 768" "1552 768 3072" "768 768 388" "77 768 768" "4096 320 320" "4096 192 320" "4096 320 192" "1024 640 640" "256 1280 1280" "64 1280 1280" "512 1280 1280" "2048 640 640" "8 1280 1280" "616 1280 768"; do
  for cfg in "2 1" "2 3" "18 1" "16 1"; do
    set -- $cfg
    python tools/bench_kernel.py gemm $shp $1 $2 40 2>/dev/null | tail -1 >> gpurun_out/r05l_small_gemm.txt
  done
done
cat gpurun_out/r05l_small_gemm.txt
: > gpurun_out/r05l_ab.txt
for rep in 1 2; do
 for mm in 0 4096; do
  for mode in train train2; do
    AF_SMALL_GEMM_MAX_M=$mm python bench.py --mode $mode --no-cpu-baseline --no-roofline > /tmp/b.json 2>/dev/null
    python - "$rep" "$mm" "$mode" >> gpurun_out/r05l_ab.txt <<'PY'
import json,sys
d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
print(f"rep {sys.argv[1]} small_gemm_max_m {sys.argv[2]} {sys.argv[3]}: ms_per_step {d['ms_per_step']} value {d['value']} {d['config'].get('per_iteration_type')}")
PY
  done
 done
 AF_SMALL_GEMM_MAX_M=4096 python bench.py --mode train --distill-only --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('distill-only small 4096', d['ms_per_step'])" >> gpurun_out/r05l_ab.txt
 python bench.py --mode train --distill-only --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('distill-only small 0', d['ms_per_step'])" >> gpurun_out/r05l_ab.txt
done
cat gpurun_out/r05l_ab.txt
