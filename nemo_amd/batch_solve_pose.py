"""Batched render-and-compare pose solver on HIP.

Drop-in for reference nemo/models/batch_solve_pose.py: get_pre_render_samples (:43-62) and
solve_pose (:194-372) keep their signatures and return values.  The work is done by four launches:

  1. nemo_clutter_score      feature map x clutter vectors                      (:214-225)
  2. nemo_grid_score         all images x all pre-rendered poses, arg-min       (:128-164)
  3. nemo_similarity_table   vertex x pixel similarity on the matrix cores      (one GEMM per image)
  4. nemo_solve              cfg.inference.epochs Adam steps inside one launch  (:306-338)

followed by one device->host copy of the [B, 8] result.  PyTorch only allocates tensors here.
"""
import time

import numpy as np
import torch

from . import ops
from .utils.mesh import camera_position_from_spherical_angles


# ---- small torch helpers kept for API parity (not used by the hot path) -----------------------
def loss_fg_only(obj_s, clu_s=None, reduce_method=lambda x: torch.mean(x)):
    return torch.ones(1, device=obj_s.device) - reduce_method(obj_s)


def loss_fg_bg(obj_s, clu_s, reduce_method=lambda x: torch.mean(x)):
    return torch.ones(1, device=obj_s.device) - (reduce_method(torch.max(obj_s, clu_s)) - reduce_method(clu_s))


def get_pre_render_samples(inter_module, azum_samples, elev_samples, theta_samples, distance_samples=[5], device="cpu"):
    """Pre-render the pose grid (nesting azimuth -> elevation -> theta -> distance).
    Returns (maps [N,1,C,H,W], campos [N,3], theta [N]) on the module's device; all N poses are
    rendered by one launch."""
    poses = [(a, e, t, d) for a in azum_samples for e in elev_samples for t in theta_samples for d in distance_samples]
    az, el, th, di = (np.array([p[i] for p in poses], dtype=np.float64) for i in range(4))
    dev = inter_module.mesh.device
    campos = camera_position_from_spherical_angles(di, el, az, degrees=False, device=dev)
    theta = torch.as_tensor(th, dtype=torch.float32).to(dev)
    with torch.no_grad():
        maps = inter_module(campos, theta)
    return maps.unsqueeze(1), campos, theta


def get_init_pos_rendered_dim0(samples_maps, samples_pos, samples_theta, predicted_maps, clutter_scores=None,
                               batch_size=32):
    """(init campos [b,3], init theta [b], min loss [b]) - arg-min of the grid loss per image.
    `batch_size` (the reference's chunk size) is accepted and ignored: nothing is chunked."""
    pre = samples_maps.squeeze(1) if samples_maps.dim() == 5 else samples_maps
    _, idx, mn = ops.grid_score(pre, predicted_maps, clutter_scores)
    idx = idx.long()
    return samples_pos[idx], samples_theta[idx], mn


get_init_pos_rendered = get_init_pos_rendered_dim0


def _adam_from_cfg(cfg):
    opt = dict(cfg.inference.optimizer)
    if opt.pop("class_name", "torch.optim.Adam") != "torch.optim.Adam":
        raise NotImplementedError("the device optimiser implements torch.optim.Adam")
    if opt.get("weight_decay", 0) or opt.get("amsgrad", False):
        raise NotImplementedError("Adam weight_decay / amsgrad are not supported")
    sch = dict(cfg.inference.scheduler)
    if sch.pop("class_name", "torch.optim.lr_scheduler.ExponentialLR") != "torch.optim.lr_scheduler.ExponentialLR":
        raise NotImplementedError("the device optimiser implements ExponentialLR")
    betas = tuple(opt.get("betas", (0.9, 0.999)))
    return dict(lr=opt.get("lr", 1e-3), betas=betas, eps=opt.get("eps", 1e-8), lr_gamma=sch.get("gamma", 1.0))


def solve_pose(cfg, feature_map, inter_module, clutter_bank, cam_pos_pre_rendered, theta_pre_rendered,
               feature_pre_rendered, device="cuda", principal=None, pre_render=True, dof=3, bbox=None, **kwargs):
    """Same contract as the reference: list (one per image) of
    dict(final=[dict(azimuth, elevation, theta, distance, principal=[px,py], score)],
         pre_compute_time, pre_rendering_time, optimization_time).
    Extra keyword `global_batch`: size of the batch the images are a shard of (defaults to b)."""
    b, c, hm_h, hm_w = feature_map.shape
    if bbox is not None:
        raise NotImplementedError("bbox-cropped final score is not implemented")
    if principal is not None or dof == 6:
        raise NotImplementedError("re-centring (principal != None) and dof == 6 are not implemented yet")
    if dof not in (3, 4):
        raise ValueError("dof must be 3, 4 or 6")
    if cfg.inference.get("inter_mode", "bilinear") != "bilinear" or cfg.inference.get("blur_radius", 0.0) != 0.0:
        raise NotImplementedError("only inter_mode='bilinear' with blur_radius 0 is supported")
    if cfg.get("optimize_translation", False):
        raise NotImplementedError("optimize_translation is not implemented yet")
    adam = _adam_from_cfg(cfg)
    epochs = int(cfg.inference.epochs)
    pred = {}

    # Step 1: clutter score
    t0 = time.time()
    banks = clutter_bank if isinstance(clutter_bank, list) else [clutter_bank]
    clutter = torch.cat([cb.reshape(-1, c) for cb in banks], dim=0)
    clutter_score = ops.clutter_score(feature_map, clutter)
    t1 = time.time()
    pred["pre_compute_time"] = t1 - t0

    # Step 2: initialisation by pose-grid scoring
    if pre_render:
        init_C, init_theta, _ = get_init_pos_rendered_dim0(feature_pre_rendered, cam_pos_pre_rendered,
                                                           theta_pre_rendered, feature_map, clutter_score)
    else:
        with torch.no_grad():
            maps = inter_module(cam_pos_pre_rendered, theta_pre_rendered)
        init_C, init_theta, _ = get_init_pos_rendered_dim0(maps, cam_pos_pre_rendered, theta_pre_rendered,
                                                           feature_map, clutter_score)
    t2 = time.time()
    pred["pre_rendering_time"] = t2 - t1

    # Step 3: pose optimisation, all epochs on device
    cam = inter_module.camera()
    table = ops.similarity_table(inter_module.mesh, feature_map)
    pose, _ = ops.solve(inter_module.mesh, cam, table, clutter_score, init_C, init_theta, None, None,
                        global_batch=kwargs.get("global_batch", b), dof=dof, steps=epochs,
                        lr_every=epochs // 3 if epochs >= 3 else 0, **adam)
    sph = ops.spherical_angles(pose[:, 0:3])
    host = torch.cat([pose, sph], dim=1).cpu().numpy()     # the one device->host copy
    t3 = time.time()
    pred["optimization_time"] = t3 - t2

    preds = []
    for i in range(b):
        r = host[i]
        refined = [{
            "azimuth": float(r[10]), "elevation": float(r[9]), "theta": float(r[3]), "distance": float(r[8]),
            "principal": [float(r[4]), float(r[5])], "score": float(r[6]),
        }]
        preds.append(dict(final=refined, **{k: pred[k] / b for k in pred.keys()}))
    return preds
