#!/bin/bash
# One rank's share of the benchmark frame for 1, 2, 4 and 8 ranks on one GPU
# (tools/shard_time.py), as a table for profiles/.
cd "$(dirname "$0")/.."
timeout -k 10 300 python tools/shard_time.py "${1:-C4}" 2>/dev/null
