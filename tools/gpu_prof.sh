#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/gpu_prof.sh <tag> [bench args...]
# rocprofv3 kernel-trace profiles of bench.py -> gpurun_out/<tag>/{full,step}:
#   full : the default bench command (what the JSON line's roofline numbers come from; includes the micro-loops)
#   step : --step-only, 3 warm-up + 20 timed steps + the first eager step = 24 fused steps and nothing else
#   kernels : --kernels-only, the isolated roofline kernel loops (33 launches each): --stats averages = isolated durations
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/full -- python $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/$tag.full.log 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/step -- python $R/bench.py --steps 20 --warmup 3 --step-only --no-graph "$@" > $R/gpurun_out/$tag.step.log 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/kernels -- python $R/bench.py --kernels-only --no-graph "$@" > $R/gpurun_out/$tag.kernels.log 2>&1 &&
python $R/tools/prof_summary.py $R/gpurun_out/$tag/kernels 1 12 > $R/gpurun_out/$tag.kernels.txt &&
python $R/tools/prof_summary.py $R/gpurun_out/$tag/step 24 60 > $R/gpurun_out/$tag.step.txt &&
python $R/tools/prof_summary.py $R/gpurun_out/$tag/full 1 40 > $R/gpurun_out/$tag.full.txt
tail -3 $R/gpurun_out/$tag.step.txt
