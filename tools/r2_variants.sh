#!/bin/bash
# usage (GPU box): bash tools/r2_variants.sh <out-file> "<ENV1=.. ENV2=..>" ["<...>" ...]   -- step-only bench per env setting
out=$1; shift
: > $out
for v in "$@"; do
  echo "== $v" >> $out
  env $v timeout -k 10 200 python bench.py --step-only --steps 60 --warmup 10 --no-graph 2>>$out.err | tail -1 >> $out || exit 1
done
cat $out
