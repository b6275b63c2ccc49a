"""Batched render-and-compare pose solver on HIP.

Drop-in for reference nemo/models/batch_solve_pose.py: get_pre_render_samples (:43-62) and
solve_pose (:194-372) keep their signatures and return values.  The work is done by three launches:

  1. nemo_clutter_score      feature map x clutter vectors                      (:214-225)
  2. nemo_grid_score         all images x all pre-rendered poses, arg-min       (:128-164)
  3. nemo_solve_features     cfg.inference.epochs Adam steps inside one launch  (:306-338); the vertex x pixel
                             similarity tiles the steps read are computed on the matrix cores inside the same launch
                             (channel counts other than 16/32/64/128: nemo_similarity_table + nemo_solve)

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
    maps = maps.unsqueeze(1)
    # the same grid in the matrix-core scoring layout (None for channel counts that form does not take)
    maps._nemo_packed_grid = ops.grid_pack(maps.squeeze(1))
    return maps, campos, theta


def get_init_pos_rendered_dim0(samples_maps, samples_pos, samples_theta, predicted_maps, clutter_scores=None,
                               batch_size=32, top_k=1, grid_scores=None):
    """(init campos [b,3], init theta [b], min loss [b]) - arg-min of the grid loss per image.
    `batch_size` (the reference's chunk size) is accepted and ignored: nothing is chunked.
    With top_k > 1 the k lowest-loss grid poses of every image come back instead, best first:
    (campos [b,k,3], theta [b,k], loss [b,k]); column 0 is the arg-min pose.
    grid_scores: (loss [b,N], argmin [b], min [b]) of these very maps against this grid when the caller has them already
    (ops.clutter_and_grid_score, ops.grid_score_on_fragments): nothing is scored again."""
    if grid_scores is not None:
        loss, idx, mn = grid_scores
    else:
        pre = samples_maps.squeeze(1) if samples_maps.dim() == 5 else samples_maps
        loss, idx, mn = ops.grid_score(pre, predicted_maps, clutter_scores, packed=getattr(samples_maps, "_nemo_packed_grid", None))
    if top_k > 1:
        if top_k > loss.shape[1]:
            raise ValueError(f"top_k={top_k} exceeds the {loss.shape[1]} poses of the grid")
        # a stable ascending sort keeps the first minimum in column 0, as the device arg-min does
        order = torch.sort(loss, dim=1, stable=True)[1][:, :top_k]
        return samples_pos[order], samples_theta[order], torch.gather(loss, 1, order)
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


def align_no_centered(maps_source, distance_source, principal_source, maps_target_shape, distance_target, principal_target,
                      padding_mode="zeros"):
    """Drop-in for reference batch_solve_pose.py:65-90 (one resample launch)."""
    n = maps_source.shape[0]
    dev = maps_source.device
    def vec(t, k):
        t = torch.as_tensor(t, dtype=torch.float32, device=dev)
        return t.reshape(-1, k).expand(n, k) if t.numel() == k else t.reshape(n, k)
    shape = (int(maps_target_shape[0]), int(maps_target_shape[1]))
    return ops.align_maps(maps_source, vec(distance_source, 1), vec(principal_source, 2), shape, vec(distance_target, 1),
                          vec(principal_target, 2), padding_mode)


def get_init_pos(inter_module, samples_pos, samples_theta, predicted_maps, clutter_scores=None, reset_distance=None):
    """Initialisation without pre-rendered maps (reference batch_solve_pose.py:168-191): the pose grid is
    rendered on the fly - once for all images, or per image at its own distance when reset_distance is given."""
    with torch.no_grad():
        if reset_distance is None:
            maps = inter_module(samples_pos, samples_theta)
            return get_init_pos_rendered_dim0(maps, samples_pos, samples_theta, predicted_maps, clutter_scores)
        unit = torch.nn.functional.normalize(samples_pos)
        idx, mn = [], []
        for i in range(predicted_maps.shape[0]):
            maps = inter_module(unit * reset_distance[i].float(), samples_theta)
            _, j, m = ops.grid_score(maps, predicted_maps[i:i + 1], None if clutter_scores is None else clutter_scores[i:i + 1])
            idx.append(j)
            mn.append(m)
        idx = torch.cat(idx).long()
        return samples_pos[idx], samples_theta[idx], torch.cat(mn)


PER_CALL_COPIES = 2048           # re-centred copies (images x principal samples) scored per grid-scoring launch
ON_DEMAND_MIN_IMAGES = 256       # from this many hypotheses per call on, the solve computes its similarity-table tiles itself


def _clutter_rows(clutter_bank, c):
    banks = clutter_bank if isinstance(clutter_bank, list) else [clutter_bank]
    return torch.cat([cb.reshape(-1, c) for cb in banks], dim=0)


def initial_poses(feature_map, clutter_score, cameras, cam_pos_pre_rendered, theta_pre_rendered, feature_pre_rendered,
                  principal=None, pre_render=True, dof=3, n_hyp=1, inter_module=None, distance_source=None, distance_target=None,
                  grid_scores=None):
    """Step 2 of solve_pose (reference batch_solve_pose.py:233-301): pose-grid scoring and the choice of the starting
    pose(s) of every image.  -> (init_C [b * n_hyp, 3], init_theta [b * n_hyp], init_principal [b * n_hyp, 2] or None),
    hypothesis-major (rows i * n_hyp + k), best first.  grid_scores: the grid losses of the batch as it is (no
    re-centring) when the caller has computed them with the clutter score (see solve_pose)."""
    b = feature_map.shape[0]
    dev = feature_map.device
    size = cameras.image_size                    # [1, 2] = (H, W)
    target_shape = (int(size[0, 0]), int(size[0, 1]))
    centre = size.flip(1) / 2                    # principal point of a centred image (x, y)
    if dof in (3, 4):
        if principal is not None:
            principal = principal.float().to(dev)
            t_feature_map = align_no_centered(feature_map, distance_source, principal, target_shape, distance_target, centre)
            t_clutter_score = align_no_centered(clutter_score[:, None], distance_source, principal, target_shape,
                                                distance_target, centre).squeeze(1)
            init_principal = principal
            cameras._N = b
        else:
            init_principal = None
            t_feature_map, t_clutter_score = feature_map, clutter_score
        if pre_render:
            init_C, init_theta, _ = get_init_pos_rendered_dim0(feature_pre_rendered, cam_pos_pre_rendered,
                                                               theta_pre_rendered, t_feature_map, t_clutter_score,
                                                               top_k=n_hyp, grid_scores=grid_scores if principal is None else None)
            if n_hyp > 1:                        # [b, K, ...] -> hypothesis-major rows i * K + k
                init_C, init_theta = init_C.reshape(b * n_hyp, 3), init_theta.reshape(b * n_hyp)
                if init_principal is not None:
                    init_principal = init_principal.reshape(-1, 2).expand(b, 2).repeat_interleave(n_hyp, dim=0)
        else:
            ds = distance_source
            init_C, init_theta, _ = get_init_pos(inter_module, cam_pos_pre_rendered, theta_pre_rendered, feature_map,
                                                 clutter_score, reset_distance=None if ds is None else ds.float())
        return init_C, init_theta, init_principal
    assert pre_render
    principal = principal.float().to(dev)
    ones = torch.ones(b, device=dev)
    # Every principal sample re-centres the whole batch (batch_solve_pose.py:264-295).  The re-centred
    # copies are scored together, a few samples per grid-scoring launch, so the pre-rendered grid is
    # streamed once per launch instead of once per sample; results per sample are unchanged.
    all_C, all_theta, all_loss = [], [], []
    per_call = max(1, PER_CALL_COPIES // max(b, 1))
    packed = getattr(feature_pre_rendered, "_nemo_packed_grid", None)
    for s0 in range(0, principal.shape[0], per_call):
        chunk = principal[s0:s0 + per_call]
        if packed is not None:
            # the re-centred copies are never written: the scoring samples the batch's own maps (copy k * b + i
            # = image i re-centred on principal sample k)
            nv = chunk.shape[0] * b
            ones_v = torch.ones(nv, device=dev)
            _, idx_, loss_ = ops.grid_score_recentred(packed, feature_map, clutter_score, ones_v, ones_v,
                                                      chunk.repeat_interleave(b, dim=0), centre.to(dev).expand(nv, 2),
                                                      padding_mode="border")
            C_, th_ = cam_pos_pre_rendered[idx_.long()], theta_pre_rendered[idx_.long()]
        else:
            fms = [align_no_centered(feature_map, ones, p_, target_shape, ones, centre, padding_mode="border") for p_ in chunk]
            cls = [align_no_centered(clutter_score[:, None], ones, p_, target_shape, ones, centre, padding_mode="border").squeeze(1)
                   for p_ in chunk]
            C_, th_, loss_ = get_init_pos_rendered_dim0(feature_pre_rendered, cam_pos_pre_rendered, theta_pre_rendered,
                                                        torch.cat(fms), torch.cat(cls))
        for k in range(chunk.shape[0]):
            all_C.append(C_[k * b:(k + 1) * b])
            all_theta.append(th_[k * b:(k + 1) * b])
            all_loss.append(loss_[k * b:(k + 1) * b])
    ar = torch.arange(b, device=dev)
    if n_hyp > 1:
        if n_hyp > principal.shape[0]:
            raise ValueError(f"hypotheses={n_hyp} exceeds the {principal.shape[0]} principal samples")
        use = torch.sort(torch.stack(all_loss), dim=0, stable=True)[1][:n_hyp].T          # [b, K], best first
        init_C = torch.stack(all_C)[use, ar[:, None]].reshape(b * n_hyp, 3)
        init_theta = torch.stack(all_theta)[use, ar[:, None]].reshape(b * n_hyp)
        init_principal = principal[use].reshape(b * n_hyp, 2)
    else:
        use = torch.min(torch.stack(all_loss), dim=0)[1]
        init_C, init_theta = torch.stack(all_C)[use, ar], torch.stack(all_theta)[use, ar]
        init_principal = principal[use]
    cameras._N = b
    return init_C, init_theta, init_principal


def score_boxes_from_bbox(bbox, b, n_hyp, rate):
    """The crop of the reported score (batch_solve_pose.py:354-357): rows / cols int(bbox / rate + .5), one row per hypothesis."""
    boxes = torch.tensor([[int(float(v) / rate + .5) for v in bbox[i][:4]] for i in range(b)], dtype=torch.int32)
    return boxes.repeat_interleave(n_hyp, dim=0)


def pack_predictions(host, b, n_hyp, times):
    """host [b * n_hyp, 11] = the 8 solve columns + (distance, elevation, azimuth) -> the reference's list of dicts
    (batch_solve_pose.py:340-372), the n_hyp refined poses of an image best score first."""
    preds = []
    for i in range(b):
        rows = host[i * n_hyp:(i + 1) * n_hyp]
        refined = [{
            "azimuth": float(r[10]), "elevation": float(r[9]), "theta": float(r[3]), "distance": float(r[8]),
            "principal": [float(r[4]), float(r[5])], "score": float(r[6]),
        } for r in rows[np.argsort(rows[:, 6], kind="stable")]]
        preds.append(dict(final=refined, **{k: times[k] / b for k in times.keys()}))
    return preds


def solve_pose(cfg, feature_map, inter_module, clutter_bank, cam_pos_pre_rendered, theta_pre_rendered,
               feature_pre_rendered, device="cuda", principal=None, pre_render=True, dof=3, bbox=None, **kwargs):
    """Same contract as the reference (batch_solve_pose.py:194-372): a list (one entry per image) of
    dict(final=[dict(azimuth, elevation, theta, distance, principal=[px,py], score)],
         pre_compute_time, pre_rendering_time, optimization_time).
    kwargs: distance_source, distance_target (re-centring), down_sample_rate (bbox), and two extras:
    `global_batch`: size of the batch the images are a shard of (defaults to b);
    `hypotheses`: K >= 1 initial poses per image (default 1, the reference's behaviour).  With K > 1 the K
    lowest-loss grid poses of every image (dof 3/4) or the best pose of each of the K best principal samples
    (dof 6) are all optimised by the same launch, every one exactly as a K = 1 run started there would be,
    and `final` lists the K refined poses best score first (`final[0]` is the answer)."""
    b, c, hm_h, hm_w = feature_map.shape
    if dof not in (3, 4, 6):
        raise ValueError("dof must be 3, 4 or 6")
    if cfg.inference.get("inter_mode", "bilinear") != "bilinear" or cfg.inference.get("blur_radius", 0.0) != 0.0:
        raise NotImplementedError("only inter_mode='bilinear' with blur_radius 0 is supported")
    adam = _adam_from_cfg(cfg)
    epochs = int(cfg.inference.epochs)
    dev = feature_map.device
    pred = {}
    n_hyp = int(kwargs.get("hypotheses", 1))
    if n_hyp < 1:
        raise ValueError("hypotheses must be >= 1")
    if n_hyp > 1 and dof != 6 and not pre_render:
        raise NotImplementedError("hypotheses > 1 needs the pre-rendered grid")

    # Step 1: clutter score.  When step 2 scores the batch as it is against a packed grid, both steps come from one
    # read of the feature maps (the grid losses are then ready before step 2 is timed).
    t0 = time.time()
    packed = getattr(feature_pre_rendered, "_nemo_packed_grid", None) if pre_render else None
    grid_scores = None
    if packed is not None and dof in (3, 4) and principal is None:
        clutter_score, g_loss, g_idx, g_min = ops.clutter_and_grid_score(feature_map, _clutter_rows(clutter_bank, c), packed)
        grid_scores = (g_loss, g_idx, g_min)
    else:
        clutter_score = ops.clutter_score(feature_map, _clutter_rows(clutter_bank, c))
    t1 = time.time()
    pred["pre_compute_time"] = t1 - t0

    # Step 2: initialisation by pose-grid scoring
    if dof in (3, 4) and principal is not None:
        ds_kw, dt_kw = kwargs["distance_source"], kwargs["distance_target"]
    else:
        ds_kw, dt_kw = kwargs.get("distance_source"), kwargs.get("distance_target")
    init_C, init_theta, init_principal = initial_poses(
        feature_map, clutter_score, inter_module.rasterizer.cameras, cam_pos_pre_rendered, theta_pre_rendered,
        feature_pre_rendered, principal=principal, pre_render=pre_render, dof=dof, n_hyp=n_hyp, inter_module=inter_module,
        distance_source=ds_kw, distance_target=dt_kw, grid_scores=grid_scores)
    t2 = time.time()
    pred["pre_rendering_time"] = t2 - t1

    # Step 3: pose optimisation, all epochs on device
    h2i = torch.arange(b, device=dev).repeat_interleave(n_hyp) if n_hyp > 1 else None
    if principal is not None and dof == 3:
        ds = kwargs["distance_source"].to(dev).float().reshape(-1).expand(b).repeat_interleave(n_hyp)
        init_C = init_C / init_C.pow(2).sum(-1).pow(.5)[..., None] * ds[..., None]
    opt_dof = 6 if (dof == 6 or cfg.get("optimize_translation", False)) else dof
    cam = inter_module.camera()
    boxes = None if bbox is None else score_boxes_from_bbox(bbox, b, n_hyp, kwargs.get("down_sample_rate", 8))
    skw = dict(global_batch=kwargs.get("global_batch", b), dof=opt_dof, steps=epochs, lr_every=epochs // 3 if epochs >= 3 else 0,
               score_boxes=boxes, **adam)
    if n_hyp <= 2 and b * n_hyp >= ON_DEMAND_MIN_IMAGES and ops.on_demand_table_supported(inter_module.mesh, cam):
        # the vertex x pixel similarity tiles are computed inside the solve, when a covered pixel first needs one
        # (three or more hypotheses per image share the image's complete table instead: each would compute its own tiles;
        # and a launch that leaves most of the chip idle has time to spare for the complete table: 0.6 ms per 64 images)
        pose, _ = ops.solve_from_features(inter_module.mesh, cam, feature_map, clutter_score, init_C, init_theta, init_principal, h2i, **skw)
    else:
        table = ops.similarity_table(inter_module.mesh, feature_map)
        pose, _ = ops.solve(inter_module.mesh, cam, table, clutter_score, init_C, init_theta, init_principal, h2i, **skw)
    sph = ops.spherical_angles(pose[:, 0:3])
    host = torch.cat([pose, sph], dim=1).cpu().numpy()     # the one device->host copy
    t3 = time.time()
    pred["optimization_time"] = t3 - t2
    return pack_predictions(host, b, n_hyp, pred)
