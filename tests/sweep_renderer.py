"""Wide random sweep of the bit-exact renderer against the oracle (face indices, barycentrics, depth), to be run by hand on
a GPU box: `SWEEP_SEEDS=60 python tests/sweep_renderer.py` (8640 poses, 90 s).  Not collected by pytest."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import refsolver as R
from tests.common import Scene, random_poses
from nemo_amd import ops
dev = torch.device("cuda:0")
t_start = time.time()
total, bad = 0, 0
for (H, W, C, nv, focal, dist) in ((40, 50, 16, 150, 187.5, (4.0, 6.0)), (40, 50, 16, 150, 187.5, (1.5, 3.0)), (64, 48, 16, 400, 300.0, (3.0, 8.0)),
                                   (80, 100, 16, 1000, 375.0, (4.0, 6.0)), (80, 100, 16, 1000, 375.0, (2.0, 3.5)), (33, 47, 16, 60, 150.0, (3.0, 12.0)), (40, 50, 16, 150, 187.5, (0.3, 1.2)), (80, 100, 16, 1000, 375.0, (0.5, 1.5)), (128, 80, 16, 300, 500.0, (3.0, 5.0))):
    sc = Scene(H, W, C, nv, focal, device=dev)
    for seed in range(int(os.environ.get('SWEEP_SEEDS', '6'))):
        campos, theta = random_poses(16, seed=100 + seed, dist=dist)
        out = ops.render(sc.mod.mesh, sc.mod.camera(), campos.to(dev), theta.to(dev), maps=False, fragments=True)
        p2f, bary, zbuf = R.fragments(sc.ref.verts, sc.ref.faces, campos, theta, sc.cam)
        nb = int((out["pix_to_face"].long().cpu() != p2f).sum()) + int((out["bary"].cpu() != bary).sum()) + int((out["zbuf"].cpu() != zbuf).sum())
        total += 16
        if nb:
            bad += 1
            print(f"MISMATCH {H}x{W} V={nv} dist={dist} seed={100 + seed}: {nb} elements", flush=True)
    print(f"{H}x{W} V={sc.verts.shape[0]} dist={dist}: done, covered fraction {float((p2f >= 0).float().mean()):.2f}, {time.time() - t_start:.0f} s", flush=True)
print(f"{total} poses, {bad} batches with mismatches")
