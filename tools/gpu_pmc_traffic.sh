#!/bin/bash
# HBM traffic passes only (FETCH_SIZE, WRITE_SIZE, L2 hit) for tools/pmc_kernels.py
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmct/p$i -- python $R/tools/pmc_kernels.py > $R/gpurun_out/pmct_p$i.log 2>&1 || echo "pass $i failed"
done
