#!/bin/bash
# usage (GPU box): bash tools/r3_ab.sh <tag> <bench flags> "<ENV..>" ["<ENV..>" ...]   -- unprofiled bench.py --step-only timings, the variants
# interleaved REPS times (default 2) in one box; one line per run: variant, ms/step
tag=$1; shift
flags=$1; shift
R=$GRAFT_REPO_ROOT
for rep in $(seq ${REPS:-2}); do
for v in "$@"; do
  ms=$( ( export $v; timeout -k 10 200 python $R/bench.py --step-only --steps 100 --warmup 10 $flags 2>>$R/gpurun_out/$tag.err ) | tail -1 | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1/' ) || exit 1
  echo "$v ($flags): $ms ms/step" >> $R/gpurun_out/$tag.txt
done
done
cat $R/gpurun_out/$tag.txt
