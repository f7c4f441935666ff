#!/bin/bash
# One rocprofv3 counter pass over two frames of the benchmark (one stream), and
# the per-kernel sums of its counters.
# usage: tools/pmc_pass.sh <name> COUNTER [COUNTER ...]     (run on the GPU box through gpurun)
set -u
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/pmc_$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export RT_AMD_OVERLAP=0
timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $out -o c -- python3 $root/bench.py --steps 1 --warmup 1 --cpu-seconds 0 --no-stats-pass ${BENCH_ARGS:-} > /dev/null 2> $out/err.txt || { tail -5 $out/err.txt; exit 1; }
python3 $root/tools/pmc_table.py $out/c_counter_collection.csv > $out/table.md
cat $out/table.md
