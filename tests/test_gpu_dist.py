"""The RCCL leg of bench.py on the one GPU a test box has: a one-rank process
group launched the way the multi-GPU bench is launched, with the gather and
the reductions forced through the backend."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_single_rank_through_rccl(built):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RT_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", "1", "--steps", "2", "--warmup", "1", "--config", "C1", "--cpu-seconds", "0"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["roofline"]["k_trace"]["rays_per_step"] > 0
