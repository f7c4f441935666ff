#!/bin/bash
# Wavefront-level iteration counts of the traversal kernels on the benchmark frame, from the three
# diagnostic builds variants/librt_amd_div{1,2,3}.so (-DRT_DIVERGENCE_STATS=1|2|3: the statistics
# counters then count wavefront iterations instead of lane visits).  Run on the GPU box.
#   build 1: nodes = node iterations of a wavefront, primitives = triangle iterations
#   build 2: nodes = triangle iterations in which no lane hits
#   build 3: primitives = node iterations in which every lane at a node is at the same node
set -u
cd "$(dirname "$0")/.."
cp raytracer_amd/librt_amd.so /tmp/librt_amd_base.so
for v in 1 2 3; do
  cp variants/librt_amd_div$v.so raytracer_amd/librt_amd.so
  timeout -k 10 200 python bench.py --steps 1 --warmup 1 --cpu-seconds 0 2>/dev/null | python -c "
import json, sys
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = j['roofline']
for k in ('k_trace', 'k_shadow'):
    x = r[k]; rays = x['rays_per_step']
    print('build %d  %s  %d rays  per 64 rays: nodes counter %.3f  primitives counter %.3f' % ($v, k, rays, 64.0 * x['nodes_per_ray_exact'], 64.0 * x['tris_per_ray_exact']))
" || { cp /tmp/librt_amd_base.so raytracer_amd/librt_amd.so; exit 1; }
done
cp /tmp/librt_amd_base.so raytracer_amd/librt_amd.so
