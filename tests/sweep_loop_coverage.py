"""How often does the optimisation loop's rasteriser (stepped edge functions, 32-bit keys) decide a pixel differently from the
bit-exact renderer (the reference rule)?  Random poses of the 80 x 100 scene; run by hand on a GPU box:
`python tests/sweep_loop_coverage.py` (40 s).  Not collected by pytest."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.common import Scene, random_poses
from nemo_amd import ops
dev = torch.device("cuda:0")
tot_px = tot_cov = d_cov = d_face = n_pose = 0
for (H, W, nv, focal, dist) in ((80, 100, 1000, 375.0, (4.0, 6.0)), (80, 100, 1000, 375.0, (2.0, 3.5)), (40, 50, 150, 187.5, (4.0, 6.0))):
    sc = Scene(H, W, 16, nv, focal, device=dev)
    cam = sc.mod.camera()
    fbits = 9
    while (1 << fbits) <= sc.mod.mesh.F:
        fbits += 1
    for seed in range(int(os.environ.get("SWEEP_SEEDS", "40"))):
        campos, theta = random_poses(64, seed=900 + seed, dist=dist)
        _, _, k0 = ops.raster_forms(sc.mod.mesh, cam, campos.to(dev), theta.to(dev), 0, reps=1)
        p2f = ops.render(sc.mod.mesh, cam, campos.to(dev), theta.to(dev), maps=False, fragments=True)["pix_to_face"].long()
        cov0, cov1 = k0 != 0xFFFFFFFF, p2f >= 0
        both = cov0 & cov1
        tot_px += cov0.numel(); tot_cov += int(cov1.sum()); n_pose += 64
        d_cov += int((cov0 ^ cov1).sum()); d_face += int(((k0 & ((1 << fbits) - 1))[both] != p2f[both]).sum())
print(f"{n_pose} poses, {tot_px} pixels, {tot_cov} covered: coverage differs at {d_cov} pixels, the winning face at {d_face} more")
