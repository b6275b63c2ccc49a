"""bench.py under torch.distributed.run with one rank: the process group is RCCL ("nccl"), and the gather of the
results, the barrier and the max-over-ranks of the timing run on device tensors through it - the code path of
`--gpus N`, exercised on the one GPU a test box has."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    """A TCP port nobody listens on right now (for the rendezvous of the launched ranks)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        return str(sock.getsockname()[1])


@pytest.mark.gpu
def test_bench_single_rank_rccl_group():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--batch", "8", "--in-flight", "2", "--steps", "4",
           "--warmup", "2", "--no-cpu-baseline", "--no-other-configs"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and "nccl" in d["config"]["partition"] and "world size 1" in d["config"]["partition"]
    assert d["value"] > 0 and d["median_pose_error_deg_vs_synthetic_gt"] < 2.0
    assert 0 < d["roofline"]["frac"] <= 1


@pytest.mark.gpu
def test_bench_two_ranks_share_the_gpu(tmp_path):
    """The N > 1 control flow of bench.py on a one-GPU box: two ranks (one process each) share the GPU and talk over gloo
    (NEMO_BENCH_REHEARSAL=1; RCCL refuses two ranks on one device).  Every rank solves its own 64 images with the loss
    normaliser of the 128-image global batch, the gathered rows come rank-major, and each rank's rows equal what one
    process computes for those images with global_batch = 128."""
    import numpy as np
    import torch
    dump = str(tmp_path / "rows.npy")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", NEMO_BENCH_REHEARSAL="1", NEMO_BENCH_DUMP=dump)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "64", "--in-flight", "1", "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline", "--no-other-configs"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 128 and d["config"]["batch_per_gpu"] == 64
    assert "world size 2" in d["config"]["partition"] and "gloo" in d["config"]["partition"]
    assert d["scaling"] == "weak" and d["value"] > 0
    rows = np.load(dump)
    assert rows.shape == (128, 7)
    # the same images in this process: rank r's batch is the 64 images seeded from r * 64
    sys.path.insert(0, ROOT)
    import bench
    import nemo_amd
    from nemo_amd import ops
    from nemo_amd.utils import default_pose3d_config, get_param_samples
    dev = torch.device("cuda:0")
    mod, mem, clutter, v, f = bench.build_scene(dev)
    az, el, th, di, _, _ = get_param_samples(default_pose3d_config())
    maps, cs, ths = nemo_amd.get_pre_render_samples(mod, az, el, th, di)
    for r in range(2):
        fm, _ = bench.synthetic_images(mod, clutter, 64, first_seed=r * 64, dev=dev)
        clu, _, idx, _ = ops.clutter_and_grid_score(fm, clutter, maps._nemo_packed_grid)
        pose, _ = ops.solve_from_features(mod.mesh, mod.camera(), fm, clu, cs[idx.long()], ths[idx.long()], global_batch=128,
                                          steps=bench.EPOCHS, lr_every=bench.EPOCHS // 3)
        sph = ops.spherical_angles(pose[:, 0:3])
        want = torch.cat([sph[:, 2:3], sph[:, 1:2], pose[:, 3:4], sph[:, 0:1], pose[:, 4:7]], dim=1).cpu().numpy()
        assert np.array_equal(rows[r * 64:(r + 1) * 64], want), f"rank {r}'s rows differ from a single-process solve of its images"
    # and the normaliser matters: with the local batch size as normaliser the rows are different numbers
    alone, _ = ops.solve_from_features(mod.mesh, mod.camera(), fm, clu, cs[idx.long()], ths[idx.long()], global_batch=64,
                                       steps=bench.EPOCHS, lr_every=bench.EPOCHS // 3)
    assert not torch.equal(alone, pose)
