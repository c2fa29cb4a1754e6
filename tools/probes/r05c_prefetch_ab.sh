# Round 5: weight prefetch on a second stream of the captured denoise graph (ops.WeightPrefetcher), alternating runs on ONE box.
R=$GRAFT_REPO_ROOT
cd $R
OUT=gpurun_out/r05c_prefetch_stream.txt
: > $OUT
for rep in 1 2; do
  for cfg in "0 64" "2 64" "4 64" "2 16" "8 32" "3 128"; do
    set -- $cfg
    line=$(python bench.py --mode denoise --no-cpu-baseline --no-roofline --steps 100 --warmup 10 --prefetch-stream $1 --prefetch-wgs $2 2>/tmp/err.log | tail -1)
    ms=$(python -c "import json,sys; print(json.loads(sys.argv[1])['ms_per_step'])" "$line" 2>/dev/null)
    echo "rep $rep depth $1 wgs $2: ms_per_step $ms  $(grep 'weight prefetch' /tmp/err.log | tail -1)" >> $OUT
  done
done
cat $OUT
