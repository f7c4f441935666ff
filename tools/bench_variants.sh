#!/bin/bash
# Runs bench.py once per library variant under variants/ (A/B runs on one box).
# usage: tools/bench_variants.sh name1 name2 ...   (name "base" = the in-tree build;
# variants/librt_amd_<name>.so are built beforehand with extra -D switches)
set -u
cd "$(dirname "$0")/.."
cp raytracer_amd/librt_amd.so /tmp/librt_amd_base.so
for v in "$@"; do
  if [ "$v" = base ]; then cp /tmp/librt_amd_base.so raytracer_amd/librt_amd.so
  else cp "variants/librt_amd_$v.so" raytracer_amd/librt_amd.so; fi
  echo "== $v"
  timeout -k 10 200 python bench.py --steps 4 --warmup 2 --cpu-seconds 0 ${BENCH_ARGS:-} 2>/dev/null | python tools/bench_line.py || exit 1
done
cp /tmp/librt_amd_base.so raytracer_amd/librt_amd.so
