R=$GRAFT_REPO_ROOT
cd $R
python tools/torch_op_census.py --leg train2 > gpurun_out/r05m_census_train2.txt 2>/dev/null
head -110 gpurun_out/r05m_census_train2.txt | cut -c1-150
python tools/torch_op_census.py --leg train --micro-batches 4 > gpurun_out/r05m_census_train.txt 2>/dev/null
head -60 gpurun_out/r05m_census_train.txt | cut -c1-150
