#!/bin/bash
# usage (GPU box): bash tools/r2_variants.sh <out-file> <bench flags, e.g. --graph or --no-graph> "<ENV1=.. ENV2=..>" ["<...>" ...]
out=$1; shift
flags=$1; shift
: > $out
for v in "$@"; do
  echo "== $v ($flags)" >> $out
  env $v timeout -k 10 200 python bench.py --step-only --steps 60 --warmup 10 $flags 2>>$out.err | tail -1 | cut -c1-200 >> $out || exit 1
done
cat $out
