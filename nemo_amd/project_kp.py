"""PackedRaster - the training-side keypoint annotator (reference nemo/models/project_kp.py:32-61, 64-143,
177-247) for the triangle rasteriser (`type='near'`).  Calling it with a batch of poses returns the mesh vertices
projected to feature-map pixels, (y, x), and their visibility; the whole of get_one_standard for the poses that
share a mesh runs in one launch (nemo_project_keypoints).

The mesh of a pose is chosen by `func_of_mesh` as in the reference's training loop (nemo.py:146, :164):
  func_single        the one mesh of the annotator (or mesh i for pose i when it holds as many meshes as poses),
  func_reselect      mesh `indexs[i]` for pose i (one annotator for all categories of a multi-class bank),
  func_multi_select  a scene per pose: meshes `indexs[i][j]` moved by `transforms[i][j]` and concatenated, so that
                     the objects occlude each other.
Outputs are padded to the largest vertex count among the poses' meshes (padding slots: not visible, placed at the
principal point, where the reference's zero-padded vertices project).  VoGE rasterisation (`type='voge'`) and
deformable meshes are not provided.
"""
import torch

from . import ops
from .utils.mesh import camera_position_from_spherical_angles


def func_single(meshes, **kwargs):
    """meshes: list of (verts [V,3], faces [F,3]); -> the list itself (reference project_kp.py:32-33)."""
    return meshes


def func_reselect(meshes, indexs=None, **kwargs):
    """-> [meshes[indexs[i]] for every pose i] (reference project_kp.py:36-41)."""
    return [meshes[int(i)] for i in indexs]


def _transform_points(transform, verts):
    """Row-vector homogeneous transform, as PyTorch3D's Transform3d.transform_points: [v, 1] @ M, divided by w.
    transform: a [4,4] (or [1,4,4]) matrix or an object with get_matrix()."""
    m = transform.get_matrix() if hasattr(transform, "get_matrix") else torch.as_tensor(transform)
    m = m.reshape(4, 4).to(device=verts.device, dtype=verts.dtype)
    out = (verts[:, :, None] * m[None, :3, :]).sum(dim=1) + m[3]        # plain f32 products, no GEMM library
    return out[:, :3] / out[:, 3:]


def func_multi_select(meshes, indexs, transforms, **kwargs):
    """One scene per pose: object j of pose i is mesh indexs[i][j] moved by transforms[i][j]; vertices and
    (index-shifted) faces concatenated in that order (reference project_kp.py:44-61).  transforms: per pose a
    sequence of 4x4 matrices (row-vector convention) or Transform3d-like objects, or a [b, k, 4, 4] tensor."""
    scenes = []
    for tr, idx in zip(transforms, indexs):
        verts, faces, shift = [], [], 0
        for j in range(len(tr)):
            v, f = meshes[int(idx[j])]
            verts.append(_transform_points(tr[j], v))
            faces.append(f + shift)
            shift += v.shape[0]
        scenes.append((torch.cat(verts, dim=0), torch.cat(faces, dim=0)))
    return scenes


def _as_mesh_list(verts, faces):
    """(verts, faces) in any of the forms the reference's to_tensor takes -> list of ([V,3] f32, [F,3] i64)."""
    if isinstance(verts, (list, tuple)):
        pairs = list(zip(verts, faces))
    else:
        verts, faces = torch.as_tensor(verts), torch.as_tensor(faces)
        pairs = [(verts, faces)] if verts.dim() == 2 else list(zip(verts, faces))
    return [(torch.as_tensor(v).float(), torch.as_tensor(f).long()) for v, f in pairs]


class PackedRaster:
    def __init__(self, raster_configs, object_mesh, mesh_mode="single", device="cuda"):
        """raster_configs: dict with 'image_size' (H, W), 'down_rate', and optionally 'type' ('near'),
        'use_degree', 'focal_length' (3000), 'blur_radius' (0).  object_mesh: (verts, faces) or
        dict(verts=, faces=); one mesh (arrays or tensors, a leading batch axis is accepted) or lists of meshes."""
        raster_type = raster_configs.get("type", "near")
        if raster_type != "near" or mesh_mode != "single":
            raise NotImplementedError("PackedRaster: only type='near' with mesh_mode='single' is supported")
        if raster_configs.get("blur_radius", 0.0) not in (0, 0.0):
            raise NotImplementedError("PackedRaster: blur_radius must be 0")
        self.raster_type, self.mesh_mode = raster_type, mesh_mode
        self.use_degree = raster_configs.get("use_degree", False)
        self.raster_configs = self.kwargs = raster_configs
        self.down_rate = raster_configs.get("down_rate")
        image_size = raster_configs.get("image_size")
        self.feature_size = (image_size[0] // self.down_rate, image_size[1] // self.down_rate)
        focal = raster_configs.get("focal_length", 3000) / self.down_rate
        self.camera = ops.Camera(self.feature_size[0], self.feature_size[1], float(focal), float(self.feature_size[1] // 2),
                                 float(self.feature_size[0] // 2))
        verts, faces = (object_mesh["verts"], object_mesh["faces"]) if isinstance(object_mesh, dict) else (object_mesh[0], object_mesh[1])
        self.device = torch.device(device)
        self.meshes = [(v.to(self.device), f.to(self.device)) for v, f in _as_mesh_list(verts, faces)]
        if not self.meshes:
            raise ValueError("PackedRaster: no mesh")
        self.handles = [ops.MeshHandle(v, f) for v, f in self.meshes]
        self.verts, self.mesh = self.meshes[0][0], self.handles[0]

    def step(self):
        pass

    def get_verts_recent(self):
        """[L, Vmax, 3], zero padded (the reference's verts_padded)."""
        vmax = max(v.shape[0] for v, _ in self.meshes)
        out = torch.zeros(len(self.meshes), vmax, 3, device=self.device)
        for i, (v, _) in enumerate(self.meshes):
            out[i, :v.shape[0]] = v
        return out

    def __call__(self, azim, elev, dist, theta, **kwargs):
        """-> (project_verts [b, Vmax, 2] as (y, x) in feature-map pixels, vis_mask [b, Vmax] bool).
        kwargs: principal [b, 2] in image pixels (divided by down_rate like the reference), restrict_to_boundary,
        dist_thr, func_of_mesh (+ indexs, transforms)."""
        ext = None
        if "R" in kwargs and "T" in kwargs:               # the caller's own extrinsics (reference :127-129)
            ext = (kwargs["R"].float().to(self.device).reshape(-1, 3, 3), kwargs["T"].float().to(self.device).reshape(-1, 3))
            campos = theta = None
            n = ext[0].shape[0]
        else:
            campos = camera_position_from_spherical_angles(dist, elev, azim, degrees=self.use_degree, device=self.device)
            theta = torch.as_tensor(theta, dtype=torch.float32).reshape(-1).to(self.device)
            n = campos.shape[0]

        principal = kwargs.get("principal", None)
        if principal is not None:
            principal = (principal.float().to(self.device) / self.down_rate).reshape(-1, 2)
            if principal.shape[0] == 1 and n > 1:
                principal = principal.expand(n, 2)
        dist_thr = kwargs.get("dist_thr", self.kwargs.get("dist_thr", 1e-3))
        restrict = kwargs.get("restrict_to_boundary", self.kwargs.get("restrict_to_boundary", True))

        def annotate(handle, rows=None):
            sel = (lambda t: t) if rows is None else (lambda t: None if t is None else t[rows])
            return ops.project_keypoints(handle, self.camera, sel(campos), sel(theta), sel(principal), dist_thr=dist_thr,
                                         restrict_to_boundary=restrict, R=None if ext is None else sel(ext[0]),
                                         T=None if ext is None else sel(ext[1]))

        func = kwargs.get("func_of_mesh", None) or func_single
        name = getattr(func, "__name__", str(func))
        if name == "func_single":
            if len(self.handles) == 1:
                return annotate(self.mesh)
            if len(self.handles) != n:
                raise ValueError(f"func_single: {len(self.handles)} meshes for {n} poses")
            per_pose = list(self.handles)
        elif name == "func_reselect":
            idx = torch.as_tensor(kwargs.get("indexs")).reshape(-1).long().tolist()
            if len(idx) != n or any(i < 0 or i >= len(self.handles) for i in idx):
                raise ValueError("func_reselect: indexs must hold one mesh index per pose")
            per_pose = [self.handles[i] for i in idx]
        elif name == "func_multi_select":
            transforms, indexs = kwargs.get("transforms"), kwargs.get("indexs")
            if transforms is None or indexs is None or len(transforms) != n or len(indexs) != n:
                raise ValueError("func_multi_select: indexs and transforms must hold one entry per pose")
            per_pose = [ops.MeshHandle(v, f) for v, f in func_multi_select(self.meshes, indexs, transforms)]
        else:
            raise NotImplementedError(f"PackedRaster: unknown func_of_mesh {name}")

        vmax = max(h.V for h in per_pose)
        pad = principal if principal is not None else torch.tensor([[self.camera.px, self.camera.py]], device=self.device).expand(n, 2)
        pad = pad.flip(-1)                                                        # (y, x)
        if restrict:
            hi = torch.tensor([self.feature_size[0] - 1, self.feature_size[1] - 1], device=self.device, dtype=torch.float32)
            pad = torch.minimum(pad, hi).clamp(min=0.0)
        kp = pad[:, None, :].expand(n, vmax, 2).contiguous()
        vis = torch.zeros(n, vmax, dtype=torch.bool, device=self.device)
        groups = {}
        for i, h in enumerate(per_pose):                                           # poses that share a mesh share a launch
            groups.setdefault(id(h), (h, []))[1].append(i)
        for h, rows in groups.values():
            r = torch.tensor(rows, device=self.device)
            k_, v_ = annotate(h, r)
            kp[r, :h.V] = k_
            vis[r, :h.V] = v_
        if name == "func_multi_select":
            torch.cuda.current_stream(self.device).synchronize()                  # the per-pose scene meshes are freed here
            for h in per_pose:
                h.close()
        return kp, vis
