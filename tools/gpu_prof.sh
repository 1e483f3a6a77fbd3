#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/gpu_prof.sh <tag> [bench args...]
# kernel-trace profile of bench.py in eager mode (kernel names) -> gpurun_out/<tag>/
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-graph "$@" > $R/gpurun_out/$tag.log 2>&1
python $R/tools/prof_summary.py $R/gpurun_out/$tag 24 40
