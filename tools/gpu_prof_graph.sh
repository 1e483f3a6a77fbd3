#!/bin/bash
# kernel-trace of bench.py in hipGraph mode -> per-launch timeline of one replay
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > $R/gpurun_out/$tag.log 2>&1
python $R/tools/prof_summary.py $R/gpurun_out/$tag 25 12
