"""Wide sweep of the first optimisation steps against the oracle: per-step loss and pose after four updates, dof 3 and 6,
random starts (near the truth and anywhere), several scene sizes.  Run by hand on a GPU box:
`SWEEP_SEEDS=120 python tests/sweep_solve_steps.py` (1920 hypotheses, 90 s).  Not collected by pytest.  It reports the runs whose
error exceeds the 1e-4 of the parity tests: a pixel centre within rounding distance of an edge that the loop's stepped edge
functions and the reference rule decide differently, or a gradient component near zero whose rounding Adam's normalisation
magnifies."""
import os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import refsolver as R
from tests.common import Scene
from nemo_amd import ops
dev = torch.device("cuda:0")
t_start = time.time()
worst_loss, worst_pose, n_runs = 0.0, 0.0, 0
def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-12))
for (H, W, C, nv, focal) in ((40, 50, 32, 150, 187.5), (64, 48, 16, 400, 300.0), (33, 47, 16, 60, 150.0), (80, 100, 32, 1000, 375.0)):
    sc = Scene(H, W, C, nv, focal, device=dev)
    B = 4
    for seed in range(int(os.environ.get("SWEEP_SEEDS", "6"))):
        fm, gts = sc.feature_maps(B, 10 * seed)
        clu = R.clutter_score(fm, sc.clutter)
        g = torch.Generator().manual_seed(500 + seed)
        spread = (0.3, 1.5)[seed % 2]                   # near the truth / anywhere
        az = torch.tensor([t["azimuth"] for t in gts]) + spread * (torch.rand(B, generator=g) - 0.5)
        el = torch.tensor([t["elevation"] for t in gts]) + 0.5 * spread * (torch.rand(B, generator=g) - 0.5)
        th0 = (torch.tensor([t["theta"] for t in gts]) + 0.5 * spread * (torch.rand(B, generator=g) - 0.5)).float()
        dist = 3.5 + 3.0 * torch.rand(B, generator=g)
        C0 = R.camera_position_from_spherical_angles(dist, el, az)
        dof = (3, 6)[(seed // 2) % 2]
        pr0 = None
        if dof == 6:
            pr0 = torch.tensor([[W / 2, H / 2]]).repeat(B, 1) + 4.0 * (torch.rand(B, 2, generator=g) - 0.5)
        epochs = 4
        Cr, thr, pr, scores, hist = R.optimize_poses(sc.ref, fm, clu, C0, th0, pr0, dof, epochs=epochs, record=True)
        tab = ops.similarity_table(sc.mod.mesh, fm.to(dev))
        pose, lh = ops.solve(sc.mod.mesh, sc.mod.camera(), tab, clu.to(dev), C0.to(dev), th0.to(dev), None if pr0 is None else pr0.to(dev),
                             dof=dof, steps=epochs, history=True)
        per_image = torch.stack([h["per_image"] for h in hist], dim=1)
        e_loss = rel(lh, per_image)
        e_pose = max(rel(pose[:, 0:3], Cr), float((pose[:, 3].cpu() - thr).abs().max()))
        if dof == 6:
            e_pose = max(e_pose, float((pose[:, 4:6].cpu() - pr).abs().max()) / max(H, W))
        worst_loss, worst_pose, n_runs = max(worst_loss, e_loss), max(worst_pose, e_pose), n_runs + B
        if e_loss > 1e-4 or e_pose > 1e-4:
            print(f"ABOVE BAR {H}x{W} V={nv} seed={seed} dof={dof}: loss {e_loss:.2e} pose {e_pose:.2e}", flush=True)
    print(f"{H}x{W} V={sc.verts.shape[0]}: worst so far loss {worst_loss:.2e} pose {worst_pose:.2e}, {time.time() - t_start:.0f} s", flush=True)
print(f"{n_runs} hypotheses x 4 steps: worst per-step loss error {worst_loss:.2e} (relative), worst pose error {worst_pose:.2e}")
