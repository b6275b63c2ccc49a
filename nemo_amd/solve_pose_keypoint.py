"""Keypoint-initialised pose solver on HIP.

Drop-in for reference nemo/models/solve_pose.py: `solve_pose` (:85-347), `get_corr_pytorch` (:27-82),
`pre_compute_kp_coords` (:405-562) and `resolve_occ` (:350-402) keep their signatures and return
values.  One image at a time (the reference asserts batch size 1, nemo/models/nemo.py:396).

  1. nemo_clutter_score + nemo_similarity_table   the reference's clutter_score / kpt_score_map   (:104-124)
  2. nemo_keypoint_corr                           gather-sum over every pose x translation sample (:27-82)
  3. host: local maxima of the correlation map, one proposal per maximum                          (:161-207)
  4. nemo_solve_flow                              all proposals refined by one launch             (:220-324)

The local-maximum search restates skimage.feature.peak_local_max (scikit-image 0.19.3, the version
the reference pins) for the arguments the reference passes: min_distance, exclude_border=False.
"""
import time

import numpy as np
import torch

from . import ops
from .batch_solve_pose import _adam_from_cfg
from .utils.mesh import camera_position_from_spherical_angles, camera_position_to_spherical_angle, load_off


def loss_fg_only(obj_s, clu_s=None, device="cuda"):
    return torch.ones(1, device=obj_s.device) - torch.mean(obj_s)


def loss_fg_bg(obj_s, clu_s, device="cuda"):
    return torch.ones(1, device=obj_s.device) - (torch.mean(torch.max(obj_s, clu_s)) - torch.mean(clu_s))


# ---------------------------------------------------------------------------------------------
# correlation of the projected keypoints with the score maps
# ---------------------------------------------------------------------------------------------
def get_corr_pytorch(px_samples, py_samples, kpt_score_map, kp_coords, kp_vis, down_sample_rate, hm_h, hm_w,
                     batch_size=12, device="cuda", mesh=None):
    """corr [n_pose, n_px, n_py] (numpy f32).  `kpt_score_map` is either the similarity table of one image
    (ops.similarity_table, a device tensor) or the reference's dense [nkpt, H*W] array, which is laid out
    as a table first; `mesh` (an ops.MeshHandle) is required for the dense form.  `batch_size` is accepted
    and ignored: nothing is chunked."""
    if isinstance(kpt_score_map, torch.Tensor) and kpt_score_map.dim() == 5:
        table = kpt_score_map
        if mesh is None:
            raise ValueError("get_corr_pytorch needs mesh= (the handle the table was built for)")
    else:
        if mesh is None:
            raise ValueError("get_corr_pytorch needs mesh= to lay a dense score map out as a table")
        dense = torch.as_tensor(np.asarray(kpt_score_map), dtype=torch.float32).reshape(1, mesh.V, hm_h * hm_w)
        table = ops.table_from_dense(mesh, dense.to(device))
    corr = ops.keypoint_corr(mesh, table, torch.as_tensor(np.asarray(kp_coords), dtype=torch.float32),
                             torch.as_tensor(np.asarray(kp_vis), dtype=torch.float32), np.asarray(px_samples, dtype=np.float64),
                             np.asarray(py_samples, dtype=np.float64), down_sample_rate, hm_h, hm_w)
    return corr.cpu().numpy()


# ---------------------------------------------------------------------------------------------
# local maxima (skimage.feature.peak_local_max, exclude_border=False, default thresholds)
# ---------------------------------------------------------------------------------------------
def _maximum_filter_nearest(image, radius):
    """Maximum over the (2*radius+1)^2 window, edges replicated."""
    padded = np.pad(image, radius, mode="edge")
    out = np.full(image.shape, -np.inf, dtype=image.dtype)
    for dy in range(2 * radius + 1):
        for dx in range(2 * radius + 1):
            out = np.maximum(out, padded[dy:dy + image.shape[0], dx:dx + image.shape[1]])
    return out


def peak_local_max(image, min_distance=1, exclude_border=False):
    """Coordinates [n,2] of the local maxima of a 2-D array, highest first.
    A pixel is a peak if it equals the maximum of its (2*min_distance+1)^2 neighbourhood and exceeds the
    image minimum; a flat image has no peaks; of two peaks closer than min_distance (Chebyshev) the lower
    one is dropped.  Ties are ordered by position (stable sort)."""
    if exclude_border:
        raise NotImplementedError("the reference calls peak_local_max with exclude_border=False")
    image = np.asarray(image)
    if image.size == 1:
        mask = image > image.min()
    else:
        mask = image == _maximum_filter_nearest(image, int(min_distance))
        if np.all(mask):
            mask = np.zeros_like(mask)
        mask &= image > image.min()
    coord = np.transpose(np.nonzero(mask))
    order = np.argsort(-image[mask], kind="stable")
    coord = coord[order]
    keep = []
    for c in coord:
        if all(np.max(np.abs(c - k)) >= min_distance for k in keep):
            keep.append(c)
    return np.array(keep, dtype=np.int64).reshape(-1, 2)


def _proposals(cfg, corr, poses, px_samples, py_samples):
    """The starting points of the refinement (solve_pose.py:161-207): the correlation volume [pose, px, py] is
    reduced over poses, the translation samples that are local maxima of that map and reach
    cfg.inference.pre_rendering_thr are kept (all of them; the single highest positive one when none reaches it),
    and each kept translation sample is paired with the grid pose that correlates best there.
    -> (list of dict(azimuth, elevation, theta, distance, px, py, principal), reduced map [n_px, n_py])."""
    volume = corr.reshape(-1, len(px_samples), len(py_samples))
    heat = volume.max(axis=0)
    if cfg.inference.search_translation:
        peaks = peak_local_max(heat, min_distance=cfg.inference.min_distance, exclude_border=False)
    else:
        peaks = np.zeros((1, 2), dtype=np.int64)
    height = heat[peaks[:, 0], peaks[:, 1]] if len(peaks) else np.zeros(0, dtype=heat.dtype)
    chosen = peaks[height >= cfg.inference.pre_rendering_thr]
    if len(chosen) == 0 and len(peaks) and height.max() > 0:
        chosen = peaks[int(np.argmax(height))][None]          # argmax returns the first of equal heights
    flat_poses = poses.reshape(-1, 4)
    out = []
    for ix, iy in chosen:
        azim, elev, theta, dist = flat_poses[int(np.argmax(volume[:, ix, iy]))]
        out.append({"azimuth": azim, "elevation": elev, "theta": theta, "distance": dist, "px": px_samples[ix],
                    "py": py_samples[iy], "principal": [px_samples[ix], py_samples[iy]]})
    return out, heat


def solve_pose(cfg, feature_map, inter_module, kp_features, clutter_bank, poses, kp_coords, kp_vis, px_samples=None,
               py_samples=None, debug=False, device="cuda"):
    """Same contract as the reference.  `kp_features` must be the vertex memory the module was built with
    (the reference passes the same bank to both); it is only checked for shape."""
    mesh = inter_module.mesh
    b, c, hm_h, hm_w = feature_map.size()
    if b != 1:
        raise ValueError("the keypoint-initialised solver takes one image at a time")
    if tuple(kp_features.shape) != (mesh.V, mesh.C):
        raise ValueError("kp_features does not match the module's vertex memory")
    if cfg.inference.get("loss", "nemo.models.solve_pose.loss_fg_bg").split(".")[-1] != "loss_fg_bg":
        raise NotImplementedError("the device solver implements loss_fg_bg")
    pred = {}
    dsr = cfg.model.down_sample_rate
    fm = feature_map.to(mesh.device)

    # Step 1: foreground and background scores
    start_time = time.time()
    clutter_score = ops.clutter_score(fm, clutter_bank)
    table = ops.similarity_table(mesh, fm)
    end_time = time.time()
    pred["pre_compute_time"] = end_time - start_time

    # Step 2: initialisations
    start_time = end_time
    if px_samples is None or py_samples is None:
        if cfg.inference.search_translation:
            px_samples = np.linspace(0, hm_w * dsr, cfg.inference.num_px_samples, endpoint=True)
            py_samples = np.linspace(0, hm_h * dsr, cfg.inference.num_py_samples, endpoint=True)
        else:
            px_samples = np.array([hm_w * dsr / 2])
            py_samples = np.array([hm_h * dsr / 2])
    corr = get_corr_pytorch(px_samples, py_samples, table, kp_coords, kp_vis, dsr, hm_h, hm_w, device=device, mesh=mesh)
    extrema, corr2d_max = _proposals(cfg, corr, poses, px_samples, py_samples)
    if debug:
        pred["pre_rendering_poses"] = extrema
        pred["corr2d"] = corr2d_max
    if len(extrema) == 0:
        raise NotImplementedError
    end_time = time.time()
    pred["pre_rendering_time"] = end_time - start_time

    # Step 3: refine every proposal (one launch)
    start_time = end_time
    scale = cfg.inference.translation_scale
    n = len(extrema)
    C0 = camera_position_from_spherical_angles([e["distance"] for e in extrema], [e["elevation"] for e in extrema],
                                               [e["azimuth"] for e in extrema], degrees=False, device=mesh.device)
    th0 = torch.tensor([e["theta"] for e in extrema], dtype=torch.float32, device=mesh.device)
    fl0 = torch.tensor([[-(e["px"] - hm_w * dsr / 2) / dsr * scale, -(e["py"] - hm_h * dsr / 2) / dsr * scale] for e in extrema],
                       dtype=torch.float32, device=mesh.device)
    adam = _adam_from_cfg(cfg)
    adam.pop("lr_gamma")                   # the reference builds the scheduler but never steps it (:252-280)
    pose, _ = ops.solve_flow(mesh, inter_module.camera(), table, clutter_score, C0, th0, fl0,
                             hyp_to_image=torch.zeros(n, dtype=torch.int32), global_batch=1,
                             optimize_translation=cfg.inference.optimize_translation, translation_scale=scale,
                             steps=cfg.inference.epochs, lr_every=0, **adam)
    sph = ops.spherical_angles(pose[:, 0:3]).cpu()
    host = pose.cpu()
    refined = []
    for i in range(n):
        px_pred, py_pred = -float(host[i, 4]) / scale, -float(host[i, 5]) / scale
        refined.append({"azimuth": float(sph[i, 2]), "elevation": float(sph[i, 1]), "theta": float(host[i, 3]),
                        "distance": float(sph[i, 0]), "px": px_pred, "py": py_pred,
                        "principal": [px_pred * dsr + hm_w * dsr / 2, py_pred * dsr + hm_h * dsr / 2],
                        "score": float(host[i, 6])})
    end_time = time.time()
    pred["optimization_time"] = end_time - start_time

    if len(refined) == 1:
        pred["final"] = refined
    else:
        ret, min_score = None, 1000
        for result in refined:
            if result["score"] < min_score:
                ret, min_score = result, result["score"]
        pred["final"] = ret                 # a dict, not a list, exactly as the reference returns it (:329-338)

    if debug:
        # the reference builds these maps from the object score of the LAST forward, i.e. at the parameters before
        # the final update (:282-290): one step fewer gives exactly those parameters
        before_last, _ = ops.solve_flow(mesh, inter_module.camera(), table, clutter_score, C0, th0, fl0,
                                        hyp_to_image=torch.zeros(n, dtype=torch.int32), global_batch=1,
                                        optimize_translation=cfg.inference.optimize_translation, translation_scale=scale,
                                        steps=max(int(cfg.inference.epochs) - 1, 0), lr_every=0, **adam)
        before_last = before_last.cpu()
        seg_maps, score_maps = [], []
        for i in range(n):
            obj = _object_score_map(inter_module, fm, before_last[i], scale)
            seg_maps.append(((obj > clutter_score[0]) * (obj > 0.0)).cpu().numpy().astype(np.uint8))
            score_maps.append(obj.cpu().numpy())
        pred["object_idx_map"], pred["seg_maps"] = resolve_occ(np.array(seg_maps), np.array(score_maps))
    return pred


def _object_score_map(inter_module, fm, pose_row, scale):
    """Debug output only: the object score map of one refined proposal, [H,W]."""
    dev = fm.device
    with torch.no_grad():
        proj = inter_module(pose_row[0:3].to(dev).view(1, 3), pose_row[3:4].to(dev))       # [1,C,H,W]
        H, W = proj.shape[2:]
        gx = torch.linspace(-1.0, 1.0, W, device=dev).view(1, 1, W).expand(1, H, W) + float(pose_row[4]) / scale / ((W - 1.0) / 2.0)
        gy = torch.linspace(-1.0, 1.0, H, device=dev).view(1, H, 1).expand(1, H, W) + float(pose_row[5]) / scale / ((H - 1.0) / 2.0)
        warped = torch.nn.functional.grid_sample(proj, torch.stack([gx, gy], dim=3), mode="bilinear", padding_mode="border",
                                                 align_corners=True)
        return torch.sum(warped[0] * fm[0], dim=0)


def resolve_occ(seg_maps, score_maps):
    """Occlusion reasoning between the masks of several proposals (solve_pose.py:350-402).  Every pair of masks that
    overlap is judged on the ORIGINAL masks: a proposal's claim is its score summed over the overlap plus half its
    score over the rest of its own mask, and the proposal with the smaller claim (the later one at equal claims)
    gives the overlap up.  -> (index map: 0 = nobody, k + 1 = proposal k, the first one where several still claim a
    pixel; the reduced masks).
    Deliberate deviation: the reference's last lines add every reduced mask into mask 0 IN PLACE (`sm = seg_maps_copy[0];
    sm += seg_maps_copy[i]`) before the arg-max, so what it returns is mask 0 = the sum of all kept masks and an index
    map that says 1 wherever any proposal keeps a pixel.  That aliasing is taken for a bug (the output is only drawn
    for debugging) and is not reproduced: mask 0 stays proposal 0's and the index map names the proposals."""
    masks = np.asarray(seg_maps)
    scores = np.asarray(score_maps)
    on = masks == 1
    total = (masks * scores).reshape(len(masks), -1).sum(axis=1)
    kept = masks.copy()
    for first in range(len(masks)):
        for second in range(first + 1, len(masks)):
            both = on[first] & on[second]
            if not both.any():
                continue
            shared = [float(scores[k][both].sum()) for k in (first, second)]
            # overlap + half of what is left of the mask (masks are 0 / 1, so mask - overlap is the rest)
            claim = [shared[0] + 0.5 * (float(total[first]) - shared[0]), shared[1] + 0.5 * (float(total[second]) - shared[1])]
            loser = second if claim[0] >= claim[1] else first
            kept[loser][both] = 0
    layered = np.concatenate([np.zeros_like(kept[:1]), kept], axis=0)
    return layered.argmax(axis=0), kept


# ---------------------------------------------------------------------------------------------
# projected keypoint coordinates of the pose grid
# ---------------------------------------------------------------------------------------------
def pre_compute_kp_coords(mesh_path, azimuth_samples, elevation_samples, theta_samples, distance_samples, viewport=3000):
    """Image-plane position of every mesh vertex for every pose of the grid (solve_pose.py:405-562):
    -> (poses [A,E,T,D,4] f32, kp_coords [A*E*T*D, V, 2] f32, kp_vis [A*E*T*D, V] f32, all ones).
    `mesh_path` is an OFF file or an array of vertices [V,3] in the mesh's own axes (z up, camera circling in the
    x-y plane).  The reference's chain of rotation matrices is written out in closed form: with
    right = (cos a, sin a, 0), up = (-sin e sin a, sin e cos a, cos e) and n = (cos e sin a, -cos e cos a, sin e)
    (the unit vector from the object to the camera, which sits at dist * n), a vertex X lands at
    (u, w) = viewport * (right.X, up.X) / (dist - n.X), which is then turned by theta in the image plane and
    flipped in y.  Evaluated in float64 for all poses at once."""
    xvert = np.asarray(load_off(mesh_path)[0] if isinstance(mesh_path, str) else mesh_path, dtype=np.float64)
    grid = np.stack(np.meshgrid(np.asarray(azimuth_samples, dtype=np.float64), np.asarray(elevation_samples, dtype=np.float64),
                                np.asarray(theta_samples, dtype=np.float64), np.asarray(distance_samples, dtype=np.float64),
                                indexing="ij"), axis=-1)
    poses = grid.astype(np.float32)                           # the reference keeps the grid in float32 and projects from it
    a, e, t, d = (poses.reshape(-1, 4)[:, k].astype(np.float64) for k in range(4))
    sa, ca, se, ce = np.sin(a), np.cos(a), np.sin(e), np.cos(e)
    zero = np.zeros_like(a)
    right = np.stack([ca, sa, zero], axis=1)
    up = np.stack([-se * sa, se * ca, ce], axis=1)
    toward = np.stack([ce * sa, -ce * ca, se], axis=1)
    depth = d[:, None] - toward @ xvert.T                     # [n_pose, V]
    u = viewport * (right @ xvert.T) / depth
    w = viewport * (up @ xvert.T) / depth
    st, ct = np.sin(t)[:, None], np.cos(t)[:, None]
    kp_coords = np.stack([ct * u - st * w, -(st * u + ct * w)], axis=2).astype(np.float32)
    return poses, kp_coords, np.ones(kp_coords.shape[:2], dtype=np.float32)
