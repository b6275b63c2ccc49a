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
import copy
import math
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
    """solve_pose.py:161-207: one proposal (pose grid arg-max + translation sample) per retained maximum."""
    corr2d = corr.reshape(-1, len(px_samples), len(py_samples))
    corr2d_max = np.max(corr2d, axis=0)
    if cfg.inference.search_translation:
        extrema_2d = peak_local_max(corr2d_max, min_distance=cfg.inference.min_distance, exclude_border=False)
    else:
        extrema_2d = [[0, 0]]
    larger, max_e, max_v = [], None, 0
    for e in extrema_2d:
        if corr2d_max[e[0], e[1]] >= cfg.inference.pre_rendering_thr:
            larger.append(e)
        if corr2d_max[e[0], e[1]] > max_v:
            max_v = corr2d_max[e[0], e[1]]
            max_e = e
    iter_extrema = larger if len(larger) > 0 else [max_e]
    extrema = []
    for e in iter_extrema:
        if e is None:
            continue
        c = corr2d[:, e[0], e[1]].reshape(poses.shape[0], poses.shape[1], poses.shape[2], poses.shape[3])
        ia, ie, it, idist = np.unravel_index(np.argmax(c, axis=None), c.shape)
        p = poses[ia, ie, it, idist]
        extrema.append({"azimuth": p[0], "elevation": p[1], "theta": p[2], "distance": p[3], "px": px_samples[e[0]],
                        "py": py_samples[e[1]], "principal": [px_samples[e[0]], py_samples[e[1]]]})
    return extrema, corr2d_max


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
        seg_maps, score_maps = [], []
        for i in range(n):
            obj = _object_score_map(inter_module, fm, host[i], scale)
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
    """Pairwise occlusion reasoning between proposals: where two masks overlap, the one with the larger
    (overlap + half the rest) score keeps the overlap.  -> (object index map, masks)."""
    seg_maps_copy = copy.deepcopy(seg_maps)
    n = seg_maps.shape[0]
    for i in range(n):
        for j in range(i + 1, n):
            overlap = (seg_maps[i] == 1) * (seg_maps[j] == 1)
            if np.sum(overlap) == 0:
                continue
            score_i = np.sum(overlap * score_maps[i]) + np.sum((seg_maps[i] - overlap) * score_maps[i]) * 0.5
            score_j = np.sum(overlap * score_maps[j]) + np.sum((seg_maps[j] - overlap) * score_maps[j]) * 0.5
            if score_i >= score_j:
                seg_maps_copy[j][overlap == 1] = 0
            else:
                seg_maps_copy[i][overlap == 1] = 0
    pad = np.concatenate([np.zeros((1,) + seg_maps_copy.shape[1:], dtype=seg_maps_copy.dtype), seg_maps_copy], axis=0)
    return np.argmax(pad, axis=0), seg_maps_copy


# ---------------------------------------------------------------------------------------------
# projected keypoint coordinates of the pose grid
# ---------------------------------------------------------------------------------------------
def pre_compute_kp_coords(mesh_path, azimuth_samples, elevation_samples, theta_samples, distance_samples, viewport=3000):
    """(poses [A,E,T,D,4] f32, kp_coords [A*E*T*D, V, 2] f32, kp_vis [A*E*T*D, V] f32 (all ones)) in f64
    arithmetic like the reference.  `mesh_path` is an OFF file or an array of vertices [V,3]."""
    xvert = load_off(mesh_path)[0] if isinstance(mesh_path, str) else np.asarray(mesh_path)
    n = len(azimuth_samples) * len(elevation_samples) * len(theta_samples) * len(distance_samples)
    poses = np.zeros((n, 4), dtype=np.float32)
    count = 0
    for azim_ in azimuth_samples:
        for elev_ in elevation_samples:
            for theta_ in theta_samples:
                for dist_ in distance_samples:
                    poses[count] = [azim_, elev_, theta_, dist_]
                    count += 1
    kp_coords = np.zeros((n, len(xvert), 2), dtype=np.float32)
    kp_vis = np.ones((n, len(xvert)), dtype=np.float32)
    x3d_h = np.hstack((xvert, np.ones((len(xvert), 1)))).T
    P = np.array([[viewport, 0, 0], [0, viewport, 0], [0, 0, -1]])
    for i, (azim_, elev_, theta_, dist_) in enumerate(poses):
        C = np.zeros((3, 1))
        C[0] = dist_ * math.cos(elev_) * math.sin(azim_)
        C[1] = -dist_ * math.cos(elev_) * math.cos(azim_)
        C[2] = dist_ * math.sin(elev_)
        azimuth, elevation = -azim_, -(math.pi / 2 - elev_)
        Rz = np.array([[math.cos(azimuth), -math.sin(azimuth), 0], [math.sin(azimuth), math.cos(azimuth), 0], [0, 0, 1]])
        Rx = np.array([[1, 0, 0], [0, math.cos(elevation), -math.sin(elevation)], [0, math.sin(elevation), math.cos(elevation)]])
        R_rot = np.dot(Rx, Rz)
        R = np.hstack((R_rot, np.dot(-R_rot, C)))
        x2d = np.dot(P, np.dot(R, x3d_h))
        x2d = x2d[0:2, :] / x2d[2:3, :]
        R2d = np.array([[math.cos(theta_), -math.sin(theta_)], [math.sin(theta_), math.cos(theta_)]])
        x2d = np.dot(R2d, x2d).T
        x2d[:, 1] *= -1
        kp_coords[i] = x2d
    poses = poses.reshape(len(azimuth_samples), len(elevation_samples), len(theta_samples), len(distance_samples), 4)
    return poses, kp_coords, kp_vis
