#!/bin/bash
# the two PMC traffic passes of the Stage-1 distillation micro-batches alone (tools/profile_round.sh's last step), each under its own timeout: the WRITE_SIZE pass of the
# round's final profile run never started its workload (rocprofv3 sat after "HSA version initialized" until bench.py's 1200 s watchdog ended it)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out; TAG=r06zz
A="bench.py --mode train --distill-only --train-steps 4 --train-warmup 0 --no-train-graphs --no-cpu-baseline --no-roofline"
timeout 500 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/r06zz_pmc_tf -- python3 $A > gpurun_out/r06zz_pmc_tf.log 2>&1
timeout 500 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/r06zz_pmc_tw -- python3 $A > gpurun_out/r06zz_pmc_tw.log 2>&1
python3 tools/pmc_traffic.py $(find gpurun_out/r06zz_pmc_tf -name "*_results.db" | head -1) $(find gpurun_out/r06zz_pmc_tw -name "*_results.db" | head -1) --steps 4 --json gpurun_out/r06zz_train_traffic.json > gpurun_out/r06zz_train_traffic.txt 2>&1
tail -4 gpurun_out/r06zz_train_traffic.txt
rm -rf gpurun_out/r06zz_pmc_tf gpurun_out/r06zz_pmc_tw        # (the counter databases are far larger than what gpurun copies back)
