#!/usr/bin/env python3
"""Reads bench.py's JSON line on stdin and prints the few numbers an A/B run compares."""
import json
import sys

j = json.loads(sys.stdin.read().strip().splitlines()[-1])
c, r = j["config"], j.get("roofline") or {}
kt = r.get("k_trace", {})
print("%.1f Mrays/s  %.3f ms/frame  k_trace %.3f ms  k_shadow %.3f ms  batches %.2f  nodes/ray %s  tris/ray %s  fp64 frac %s" % (
    j["value"], j["ms_per_step"], c["trace_ms_per_step"], c["lights_ms_per_step"], c["batches_per_step"],
    kt.get("nodes_per_ray"), kt.get("tris_per_ray"), r.get("frac")))
