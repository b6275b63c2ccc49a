"""bench.py under torch.distributed.run with one rank: the process group is RCCL ("nccl"), and the gather of the
results, the barrier and the max-over-ranks of the timing run on device tensors through it - the code path of
`--gpus N`, exercised on the one GPU a test box has."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_single_rank_rccl_group():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--batch", "8", "--in-flight", "2", "--steps", "4",
           "--warmup", "2", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and "nccl" in d["config"]["partition"]
    assert d["value"] > 0 and d["median_pose_error_deg_vs_synthetic_gt"] < 2.0
    assert 0 < d["roofline"]["frac"] <= 1
