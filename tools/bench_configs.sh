#!/bin/bash
# bench.py on every workload configuration, one JSON line each into gpurun_out/configs.jsonl
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
: > gpurun_out/configs.jsonl
for c in "$@"; do
  echo "== $c" >&2
  timeout -k 10 400 python bench.py --config $c --steps 3 --warmup 1 --cpu-seconds ${CPU_SECONDS:-8} 2>> gpurun_out/configs.err | tail -1 >> gpurun_out/configs.jsonl || exit 1
done
python - <<'PY'
import json
for l in open("gpurun_out/configs.jsonl"):
    j = json.loads(l)
    print(j["config"]["workload"][:40], "|", j["value"], "Mrays/s |", j["ms_per_step"], "ms | cpu", j["cpu_baseline"]["value"] if j["cpu_baseline"] else None,
          "| batches", j["config"]["batches_per_step"], "| trace", j["config"]["trace_ms_per_step"], "lights", j["config"]["lights_ms_per_step"])
PY
