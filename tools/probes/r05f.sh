R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_hip_kernels.py -q -k "split_k_in_kernel" 2>&1 | tail -5 > gpurun_out/r05f_tests.txt
cat gpurun_out/r05f_tests.txt
timeout 900 python -m cProfile -o /tmp/prof.out -m pytest tests/test_hip_train.py -q -k "arcface_terms_back or teacher_cfg_and_shared" 2>&1 | tail -3
python - <<'PY' > gpurun_out/r05f_profile.txt 2>&1
import pstats
p = pstats.Stats('/tmp/prof.out')
p.sort_stats('cumulative').print_stats(60)
p.sort_stats('tottime').print_stats(40)
PY
head -150 gpurun_out/r05f_profile.txt | cut -c1-200
python bench.py > gpurun_out/r05f_bench.json 2> gpurun_out/r05f_bench.err
tail -c 3000 gpurun_out/r05f_bench.json
AF_SPLITK_FUSED_BYTES=1048576 python bench.py --mode train2 --no-cpu-baseline --no-roofline > gpurun_out/r05f_bench_train2_fused.json 2> gpurun_out/r05f_bench_train2_fused.err
python bench.py --mode train2 --no-cpu-baseline --no-roofline > gpurun_out/r05f_bench_train2_plain.json 2>/dev/null
AF_SPLITK_FUSED_BYTES=1048576 python bench.py --mode train --no-cpu-baseline --no-roofline > gpurun_out/r05f_bench_train_fused.json 2>/dev/null
python bench.py --mode train --no-cpu-baseline --no-roofline > gpurun_out/r05f_bench_train_plain.json 2>/dev/null
for f in train2_fused train2_plain train_fused train_plain; do echo $f; python - <<PY
import json
d=json.loads(open('gpurun_out/r05f_bench_$f.json').read().strip().splitlines()[-1])
for k in ('train','train_stage2'):
    if k in d: print(k, d[k].get('ms_per_step'), {kk:vv for kk,vv in d[k].items() if 'ms' in kk and not isinstance(vv,dict)})
PY
done
