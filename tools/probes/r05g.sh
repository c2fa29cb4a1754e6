R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests -m gpu -q --durations=25 2>&1 | tail -45 > gpurun_out/r05g_full_gpu.log
cat gpurun_out/r05g_full_gpu.log
python bench.py --mode train --reference-pass-structure --no-cpu-baseline --no-roofline > gpurun_out/r05g_train_refstruct.json 2>/dev/null
python bench.py --mode train2 --reference-pass-structure --no-cpu-baseline --no-roofline > gpurun_out/r05g_train2_refstruct.json 2>/dev/null
python bench.py --mode train --no-cpu-baseline --no-roofline > gpurun_out/r05g_train_default.json 2>/dev/null
python bench.py --mode train2 --no-cpu-baseline --no-roofline > gpurun_out/r05g_train2_default.json 2>/dev/null
for f in train_refstruct train2_refstruct train_default train2_default; do python - <<PY
import json
d=json.loads(open('gpurun_out/r05g_$f.json').read().strip().splitlines()[-1])
print('$f', d['ms_per_step'], d['value'], d['config'].get('per_iteration_type'))
PY
done
