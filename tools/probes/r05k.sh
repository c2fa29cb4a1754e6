# final-tree validation: whole GPU suite (no -x), then the round's judged artefacts
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests -m gpu -q --durations=12 2>&1 | tail -30 > gpurun_out/r05k_full_gpu.log
cat gpurun_out/r05k_full_gpu.log | cut -c1-180
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/profile_round.sh r05z > gpurun_out/r05z_profile_round.log 2>&1
tail -20 gpurun_out/r05z_profile_round.log | cut -c1-200
python tools/e2e_infer.py > gpurun_out/r05z_e2e_infer.txt 2>&1; tail -5 gpurun_out/r05z_e2e_infer.txt
