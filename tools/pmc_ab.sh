#!/bin/bash
# One counter pass per library variant (A/B on one box): tools/pmc_ab.sh "<counters>" name1 name2 ...
# (name "base" = the in-tree build, others variants/librt_amd_<name>.so)
set -u
cd "$(dirname "$0")/.."
counters=$1; shift
cp raytracer_amd/librt_amd.so /tmp/librt_amd_base.so
for v in "$@"; do
  if [ "$v" = base ]; then cp /tmp/librt_amd_base.so raytracer_amd/librt_amd.so
  else cp "variants/librt_amd_$v.so" raytracer_amd/librt_amd.so; fi
  echo "== $v"
  tools/pmc_pass.sh "ab_$v" $counters | grep -v "^$" || { cp /tmp/librt_amd_base.so raytracer_amd/librt_amd.so; exit 1; }
done
cp /tmp/librt_amd_base.so raytracer_amd/librt_amd.so
