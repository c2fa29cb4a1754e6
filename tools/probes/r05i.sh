R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_hip_kernels.py -q -k "groupnorm or statistics or gn_proj" 2>&1 | tail -4 > gpurun_out/r05i_tests.txt
timeout 900 python -m pytest tests/test_hip_unet.py tests/test_hip_vae.py -q 2>&1 | tail -4 >> gpurun_out/r05i_tests.txt
cat gpurun_out/r05i_tests.txt
for i in 1 2; do python tools/bench_gn.py 2>/dev/null | tail -12; done > gpurun_out/r05i_bench_gn.txt
cat gpurun_out/r05i_bench_gn.txt
python bench.py --mode denoise --no-cpu-baseline > gpurun_out/r05i_denoise.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05i_denoise.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['families_ms_per_step'])
PY
bash tools/profile_train_leg.sh r05i train2 > /dev/null 2>&1
head -60 gpurun_out/r05i_train2_kernel_stats.txt | cut -c1-160
