#!/bin/bash
# Runs bench.py once per value of an environment variable (A/B runs on one box).
# usage: tools/bench_env.sh VAR v1 v2 ...
set -u
cd "$(dirname "$0")/.."
var=$1; shift
for v in "$@"; do
  echo "== $var=$v"
  env $var=$v timeout -k 10 200 python bench.py --steps 4 --warmup 2 --cpu-seconds 0 ${BENCH_ARGS:-} 2>/dev/null | python tools/bench_line.py || exit 1
done
