#!/bin/bash
# A/B of library variants on the whole benchmark frame and on an eighth of it
# (the share that is dominated by the fixed cost of the recursion levels).
# usage: tools/ab_shards.sh name1 name2 ...   (as tools/bench_variants.sh)
set -u
cd "$(dirname "$0")/.."
cp raytracer_amd/librt_amd.so /tmp/librt_amd_base.so
for v in "$@"; do
  if [ "$v" = base ]; then cp /tmp/librt_amd_base.so raytracer_amd/librt_amd.so
  else cp "variants/librt_amd_$v.so" raytracer_amd/librt_amd.so; fi
  echo "== $v"
  RT_AMD_LAUNCH_EVENTS=0 timeout -k 10 200 python tools/inflight.py ${CFG:-C4} 1,8 1 2>/dev/null | tail -2 || exit 1
done
cp /tmp/librt_amd_base.so raytracer_amd/librt_amd.so
