#!/bin/bash
# usage (GPU box): bash tools/r3_ab.sh <tag> <bench flags> "<ENV..>" ["<ENV..>" ...]   -- unprofiled bench.py --step-only timings, one line per variant
tag=$1; shift
flags=$1; shift
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  echo "== $v ($flags)" >> $R/gpurun_out/$tag.txt
  ( export $v; timeout -k 10 200 python $R/bench.py --step-only --steps 100 --warmup 10 $flags 2>>$R/gpurun_out/$tag.err | tail -1 | cut -c1-200 >> $R/gpurun_out/$tag.txt ) || exit 1
done
cat $R/gpurun_out/$tag.txt
