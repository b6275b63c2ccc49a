"""Generates the committed fixtures under tests/golden/.  TEST INFRASTRUCTURE ONLY.

    python oracle/gen_golden.py [leaf] [small] [full] [grid] [solves]

leaf   tests/golden/leaf_golden.npz   - outputs of the reference's OWN dependency-free leaf files
       (nemo/utils/general.py, pose.py, features.py, flow_warp.py), loaded standalone from
       /root/reference with importlib (their package __init__ cannot be imported: it pulls in
       pytorch3d / BboxTools / cv2).  These pin the oracle's leaf functions to the reference.
small  tests/golden/oracle_small.npz  - oracle outputs on a 40x50x32 scene (face-index buffers of the
       144-pose grid, grid losses, a 300-step solve).
full   tests/golden/oracle_full.npz   - oracle outputs at BASELINE size (80x100x128, V=1118, F=1904):
       face-index buffers of 8 poses, a 2-image 300-step solve.
grid   tests/golden/oracle_full_grid.npz - BASELINE-size pose-grid scoring: losses and arg-min of 4 images against
       the 144-pose grid, the 300-step refinement of the 8 best grid poses of 2 of them (global batch 2), and a
       6-D slice: 2 off-centre images x the 576-pose grid (4 distances) x 3 principal-point samples.
solves tests/golden/oracle_full_solves.npz - BASELINE-size solves beyond one mesh / 3-D: render-and-compare
       classification of 2 images against 12 category cuboids (144-pose grid + 300 steps per pair, final losses, classes)
       and a 6-D solve of 2 off-centre images (576 poses x 3 principal samples, then 300 steps over 6 parameters).

The render-path fixtures come from the oracle itself (parity with PyTorch3D is unpinned: no PyTorch3D
here); they freeze the oracle so that a change in its behaviour, or a difference between machines, is
detected, and give the GPU tests full-size expected values without re-running minutes of CPU work.
Only inputs and expected outputs are stored - no source text.
"""
import importlib.util
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refsolver as R          # noqa: E402
from tests.common import GRID_CFG, Scene, categories, full_size_vertex_counts, random_poses   # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"


def _load(rel):
    spec = importlib.util.spec_from_file_location("ref_" + os.path.basename(rel)[:-3], os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _Node(dict):
    def __getattr__(self, k):
        v = self[k]
        return _Node(v) if isinstance(v, dict) else v


CFG6D = dict(GRID_CFG, dist_sample=dict(min=3.0, max=9.0, num=4), px_sample=dict(min=0.25, max=0.75, num=5),
             py_sample=dict(min=0.25, max=0.75, num=5))


def gen_leaf():
    general = _load("nemo/utils/general.py")
    pose = _load("nemo/utils/pose.py")
    features = _load("nemo/utils/features.py")
    flow = _load("nemo/utils/flow_warp.py")
    out = {}
    for name, cfg in (("3d", GRID_CFG), ("6d", CFG6D)):
        for k, arr in zip(("azim", "elev", "theta", "dist", "px", "py"), general.get_param_samples(_Node(inference=cfg))):
            out[f"grid_{name}_{k}"] = np.asarray(arr)
    g = torch.Generator().manual_seed(1)
    a, b = torch.randn(4, 3, generator=g), torch.randn(4, 3, generator=g)
    out["lin_start"], out["lin_end"] = a.numpy(), b.numpy()
    out["lin_out"] = general.tensor_linspace(a, b, 7).numpy()
    angles = np.array([[0.1, 0.2, 0.3], [-0.4, 1.0, 5.0], [0.0, 0.0, 0.0], [0.5, -0.5, 3.0]])
    out["rot_in"] = angles
    out["rot_out"] = np.stack([pose.cal_rotation_matrix(t, e, a_, 5.0) for t, e, a_ in angles])
    x = torch.randn(2560, 16, generator=g)
    out["norm_in"] = x.numpy()
    out["norm_out"] = features.normalize_features(torch.mean(x, dim=0)).numpy()
    im, fl = torch.randn(2, 3, 6, 7, generator=g), torch.randn(2, 2, 6, 7, generator=g)
    out["warp_im"], out["warp_flow"] = im.numpy(), fl.numpy()
    out["warp_out"] = flow.flow_warp(im, fl).numpy()
    np.savez_compressed(os.path.join(GOLD, "leaf_golden.npz"), **out)
    print("leaf_golden.npz:", {k: v.shape for k, v in out.items()})


def _solve_fixture(sc, B, epochs, first_seed):
    fm, gts = sc.feature_maps(B, first_seed)
    clu = R.clutter_score(fm, sc.clutter)
    g = torch.Generator().manual_seed(77 + first_seed)
    az = torch.tensor([t["azimuth"] for t in gts]) + 0.25 * (torch.rand(B, generator=g) - 0.5)
    el = torch.tensor([t["elevation"] for t in gts]) + 0.2 * (torch.rand(B, generator=g) - 0.5)
    th0 = (torch.tensor([t["theta"] for t in gts]) + 0.2 * (torch.rand(B, generator=g) - 0.5)).float()
    C0 = R.camera_position_from_spherical_angles(5.0, el, az)
    Cr, thr, _, scores, hist = R.optimize_poses(sc.ref, fm, clu, C0, th0, None, 3, epochs=epochs, record=True)
    return dict(first_seed=np.int64(first_seed), C0=C0.numpy(), th0=th0.numpy(), C=Cr.numpy(), theta=thr.numpy(),
                scores=scores.numpy(), loss_hist=torch.stack([h["per_image"] for h in hist], dim=1).numpy(),
                grad_C0=hist[0]["grad_C"].numpy(), grad_theta0=hist[0]["grad_theta"].numpy(), clutter_map=clu.numpy())


def gen_small():
    sc = Scene(40, 50, 32, 150, 375.0 / 2)
    az, el, th, di, _, _ = R.get_param_samples(GRID_CFG)
    maps, cs, ths = R.get_pre_render_samples(sc.ref, az, el, th, di)
    p2f, _, _ = R.fragments(sc.ref.verts, sc.ref.faces, cs, ths, sc.cam)
    fm, _ = sc.feature_maps(5)
    clu = R.clutter_score(fm, sc.clutter)
    loss = R.grid_losses(maps, fm, clu).T
    out = dict(grid_campos=cs.numpy(), grid_theta=ths.numpy(), grid_p2f=p2f.numpy().astype(np.int16),
               grid_loss=loss.numpy(), grid_argmin=loss.argmin(dim=1).numpy(),
               map_checksum=maps.double().sum(dim=(1, 2, 3, 4)).numpy())
    out.update({"solve_" + k: v for k, v in _solve_fixture(sc, 4, 300, 0).items()})
    np.savez_compressed(os.path.join(GOLD, "oracle_small.npz"), **out)
    print("oracle_small.npz written")


def gen_full():
    sc = Scene(80, 100, 128, 1000, 375.0, n_clutter=2560)
    campos, theta = random_poses(8, seed=11, dist=(3.0, 7.0))
    p2f, bary, zbuf = R.fragments(sc.ref.verts, sc.ref.faces, campos, theta, sc.cam)
    out = dict(campos=campos.numpy(), theta=theta.numpy(), p2f=p2f.numpy().astype(np.int16),
               bary_checksum=bary.detach().double().sum(dim=(1, 2, 3)).numpy(), zbuf_checksum=zbuf.detach().double().sum(dim=(1, 2)).numpy())
    out.update({"solve_" + k: v for k, v in _solve_fixture(sc, 2, 300, 0).items()})
    np.savez_compressed(os.path.join(GOLD, "oracle_full.npz"), **out)
    print("oracle_full.npz written")


def gen_full_grid():
    """What the matrix-core scoring instance of the BASELINE size (128 channels, 144 / 576 poses) is checked against."""
    sc = Scene(80, 100, 128, 1000, 375.0, n_clutter=2560)
    az, el, th, di, _, _ = R.get_param_samples(GRID_CFG)
    maps, cs, ths = R.get_pre_render_samples(sc.ref, az, el, th, di)
    first_seed, B, K = 200, 4, 8
    fm, _ = sc.feature_maps(B, first_seed)
    clu = R.clutter_score(fm, sc.clutter)
    loss = R.grid_losses(maps, fm, clu).T                                   # [B, 144]
    out = dict(grid_first_seed=np.int64(first_seed), grid_loss=loss.numpy(), grid_argmin=loss.argmin(dim=1).numpy(),
               grid_map_checksum=maps.double().sum(dim=(1, 2, 3, 4)).numpy())
    print("3-D grid done", flush=True)
    # the K best grid poses of images 0 and 1, refined together: every hypothesis sees the loss normaliser of the
    # 2-image batch (hypotheses are rows of the oracle's batch, global_batch rescales the batch mean)
    order = torch.sort(loss[:2], dim=1, stable=True)[1][:, :K]               # [2, K]
    C0, th0 = cs[order].reshape(2 * K, 3), ths[order].reshape(2 * K)
    Cr, thr, _, scores, hist = R.optimize_poses(sc.ref, fm[:2].repeat_interleave(K, dim=0), clu[:2].repeat_interleave(K, dim=0),
                                                C0, th0, None, 3, epochs=300, global_batch=2, record=True)
    out.update(k8_order=order.numpy(), k8_C=Cr.numpy(), k8_theta=thr.numpy(), k8_scores=scores.numpy(),
               k8_loss_hist=torch.stack([h["per_image"] for h in hist], dim=1).numpy())
    print("K=8 solve done", flush=True)
    # 6-D initialisation slice (batch_solve_pose.py:264-295): 576 poses x 3 principal samples x 2 off-centre images
    az6, el6, th6, di6, _, _ = R.get_param_samples(CFG6D)
    maps6, cs6, ths6 = R.get_pre_render_samples(sc.ref, az6, el6, th6, di6)
    cand = torch.tensor([[37.5, 30.0], [50.0, 40.0], [62.5, 50.0]])
    gt_pr = torch.tensor([[39.0, 31.0], [60.0, 49.0]])
    fms = []
    for i in range(2):
        sc.ref.principal = gt_pr[i:i + 1]
        fms.append(R.synthetic_feature_map(sc.ref, sc.clutter, 260 + i)[0])
    sc.ref.principal = None
    fm6 = torch.stack(fms)
    clu6 = R.clutter_score(fm6, sc.clutter)
    ones, centre = torch.ones(2), torch.tensor([[50.0, 40.0]]).expand(2, -1)
    l6 = []
    for pr in cand:
        pr_ = pr[None].expand(2, -1)
        t_fm = R.align_no_centered(fm6, ones, pr_, (80, 100), ones, centre, padding_mode='border')
        t_clu = R.align_no_centered(clu6[:, None], ones, pr_, (80, 100), ones, centre, padding_mode='border').squeeze(1)
        l6.append(R.grid_losses(maps6, t_fm, t_clu).T)
    l6 = torch.stack(l6)                                                    # [3, 2, 576]
    out.update(six_first_seed=np.int64(260), six_gt_principal=gt_pr.numpy(), six_candidates=cand.numpy(), six_loss=l6.numpy(),
               six_argmin=l6.argmin(dim=2).numpy(), six_campos=cs6.numpy(), six_theta=ths6.numpy())
    np.savez_compressed(os.path.join(GOLD, "oracle_full_grid.npz"), **out)
    print("oracle_full_grid.npz written")


def gen_full_solves(parts=("cls", "six")):
    """BASELINE configs[4] and [3] at full size: what MultiCategorySolver.classify and solve_pose(dof=6) are checked against.
    parts: which half to (re)compute; the other half is kept from the existing file."""
    path = os.path.join(GOLD, "oracle_full_solves.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}
    if "cls" in parts:
        out = {k: v for k, v in out.items() if not k.startswith("cls_")}
        out.update(_gen_full_classification())
    if "six" in parts:
        out = {k: v for k, v in out.items() if not k.startswith("six_")}
        out.update(_gen_full_six_dof())
    np.savez_compressed(path, **out)
    print("oracle_full_solves.npz written")


def _gen_full_classification():
    out = {}
    # --- classification: nemo_cls.py:262-282 - every image solved against every category mesh, smallest final loss wins.
    # Image seeds chosen so that the solve with the image's own category settles within the 300 steps (seed 701 of category 8
    # is still sliding down at step 299, in the oracle as on the GPU: its final loss then depends on rounding in step 30).
    n_cat, labels, seeds = 12, (3, 8), (700, 703)
    meshes, counts, mem, clutter = categories(n_cat, 128, number_vertices=full_size_vertex_counts, n_clutter=2560)
    cam = R.Camera(80, 100, 375.0)
    offs = np.concatenate([[0], np.cumsum(counts)[:-1]])
    refs = [R.MeshInterpolateModuleRef(v, f, mem[o:o + c], cam) for (v, f), o, c in zip(meshes, offs, counts)]
    fm = torch.stack([R.synthetic_feature_map(refs[lab], clutter, seed)[0] for seed, lab in zip(seeds, labels)])
    az, el, th, di, _, _ = R.get_param_samples(GRID_CFG)
    scores, poses, init_idx = np.zeros((2, n_cat), np.float32), np.zeros((2, n_cat, 4), np.float32), np.zeros((2, n_cat), np.int64)
    for c in range(n_cat):
        maps, cs, ths = R.get_pre_render_samples(refs[c], az, el, th, di)
        preds, info = R.solve_pose(refs[c], fm, clutter, cs, ths, maps, epochs=300)
        for i in range(2):
            f = preds[i]["final"][0]
            scores[i, c] = f["score"]
            poses[i, c] = (f["azimuth"], f["elevation"], f["theta"], f["distance"])
        init_idx[:, c] = info["init_idx"].numpy()
        print("category", c, scores[:, c], flush=True)
    out.update(cls_labels=np.array(labels), cls_seeds=np.array(seeds), cls_scores=scores, cls_poses=poses,
               cls_init_idx=init_idx, cls_argmin=scores.argmin(axis=1))
    return out


def _gen_full_six_dof():
    out = {}
    # --- 6-D: batch_solve_pose.py:264-338 with dof 6 (the images and candidates of the grid fixture's 6-D slice)
    sc = Scene(80, 100, 128, 1000, 375.0, n_clutter=2560)
    az6, el6, th6, di6, _, _ = R.get_param_samples(CFG6D)
    maps6, cs6, ths6 = R.get_pre_render_samples(sc.ref, az6, el6, th6, di6)
    cand = torch.tensor([[37.5, 30.0], [50.0, 40.0], [62.5, 50.0]])
    gt_pr = torch.tensor([[39.0, 31.0], [60.0, 49.0]])
    fms = []
    for i in range(2):
        sc.ref.principal = gt_pr[i:i + 1]
        fms.append(R.synthetic_feature_map(sc.ref, sc.clutter, 260 + i)[0])
    sc.ref.principal = None
    preds, info = R.solve_pose(sc.ref, torch.stack(fms), sc.clutter, cs6, ths6, maps6, principal=cand, dof=6, epochs=300)
    fin = np.array([[p["final"][0][k] for k in ("azimuth", "elevation", "theta", "distance")] + p["final"][0]["principal"]
                    + [p["final"][0]["score"]] for p in preds], dtype=np.float32)
    out.update(six_solve_first_seed=np.int64(260), six_solve_gt_principal=gt_pr.numpy(), six_solve_candidates=cand.numpy(),
               six_solve_final=fin, six_solve_use=info["init_idx"].numpy(), six_solve_init_C=info["init_C"].numpy(),
               six_solve_init_theta=info["init_theta"].numpy())
    print("6-D solve", fin, flush=True)
    return out


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    what = sys.argv[1:] or ["leaf", "small", "full", "grid", "solves"]
    torch.set_num_threads(8)
    if "leaf" in what:
        gen_leaf()
    if "small" in what:
        gen_small()
    if "full" in what:
        gen_full()
    if "grid" in what:
        gen_full_grid()
    if "solves" in what:
        gen_full_solves()
    if "solves_cls" in what:
        gen_full_solves(("cls",))
    if "solves_six" in what:
        gen_full_solves(("six",))
