#!/bin/bash
# usage (GPU box): bash tools/r3_graph_variants.sh <tag> "<ENV..>" ["<ENV..>" ...]
# per variant: unprofiled `bench.py --step-only --graph` time, then a rocprofv3 kernel trace of the replayed step -> timeline
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for v in "$@"; do
  i=$((i+1))
  echo "== v$i: $v" >> $R/gpurun_out/$tag.txt
  ( export $v; timeout -k 10 200 python $R/bench.py --step-only --steps 100 --warmup 10 --graph 2>>$R/gpurun_out/$tag.err | tail -1 | cut -c1-200 >> $R/gpurun_out/$tag.txt ) || exit 1
  ( export $v; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$tag/v$i -- python $R/bench.py --steps 20 --warmup 3 --step-only --graph > $R/gpurun_out/$tag.v$i.log 2>&1 ) || exit 1
  python $R/tools/timeline.py $R/gpurun_out/$tag/v$i > $R/gpurun_out/$tag.v$i.timeline.txt 2>&1
  rm -rf $R/gpurun_out/$tag/v$i
done
cat $R/gpurun_out/$tag.txt
