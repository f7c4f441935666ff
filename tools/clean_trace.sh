#!/bin/bash
# Kernel trace of timed frames only, every kernel on one stream (RT_AMD_OVERLAP=0):
# per-kernel durations that add up to the frame.
# usage: tools/clean_trace.sh <tag> [bench args]      (run on the GPU box through gpurun)
set -u
tag=${1:-r02}; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/clean_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export RT_AMD_OVERLAP=0
export RT_AMD_LAUNCH_EVENTS=0   # no event records between the kernels: the gaps are the pipeline's own
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- python3 $root/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-stats-pass "$@" > $out/bench.json 2> $out/bench.err || exit 1
python3 $root/tools/frame_split.py $out/trace/bench_kernel_trace.csv 3 > $out/frame_split.md || exit 1
cat $out/frame_split.md
