"""PackedRaster - the training-side keypoint annotator (reference nemo/models/project_kp.py:64-143, 175-247) for
its standard configuration: one mesh (`mesh_mode='single'`), triangle rasteriser (`type='near'`).  Calling it with
a batch of poses returns the mesh vertices projected to feature-map pixels, (y, x), and their visibility; the
whole of get_one_standard runs in one launch (nemo_project_keypoints).  VoGE rasterisation and the multi-mesh /
deformable modes are not provided.
"""
import torch

from . import ops
from .utils.mesh import camera_position_from_spherical_angles


class PackedRaster:
    def __init__(self, raster_configs, object_mesh, mesh_mode="single", device="cuda"):
        """raster_configs: dict with 'image_size' (H, W), 'down_rate', and optionally 'type' ('near'),
        'use_degree', 'focal_length' (3000), 'blur_radius' (0).  object_mesh: (verts, faces) or
        dict(verts=, faces=) of ONE mesh (arrays or tensors; a leading batch axis of 1 is accepted)."""
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
        if isinstance(verts, (list, tuple)):
            if len(verts) != 1:
                raise NotImplementedError("PackedRaster: one mesh only")
            verts, faces = verts[0], faces[0]
        verts, faces = torch.as_tensor(verts), torch.as_tensor(faces)
        if verts.dim() == 3:
            if verts.shape[0] != 1:
                raise NotImplementedError("PackedRaster: one mesh only")
            verts, faces = verts[0], faces[0]
        self.device = torch.device(device)
        self.verts = verts.float().to(self.device)
        self.mesh = ops.MeshHandle(self.verts, faces.to(self.device))

    def step(self):
        pass

    def get_verts_recent(self):
        return self.verts[None]

    def __call__(self, azim, elev, dist, theta, **kwargs):
        """-> (project_verts [b, V, 2] as (y, x) in feature-map pixels, vis_mask [b, V] bool).
        kwargs: principal [b, 2] in image pixels (divided by down_rate like the reference), restrict_to_boundary,
        dist_thr."""
        if "R" in kwargs or "T" in kwargs:
            raise NotImplementedError("PackedRaster: pass azim / elev / dist / theta (explicit R, T are not supported)")
        campos = camera_position_from_spherical_angles(dist, elev, azim, degrees=self.use_degree, device=self.device)
        theta = torch.as_tensor(theta, dtype=torch.float32).reshape(-1).to(self.device)
        principal = kwargs.get("principal", None)
        if principal is not None:
            principal = principal.float().to(self.device) / self.down_rate
        return ops.project_keypoints(self.mesh, self.camera, campos, theta, principal,
                                     dist_thr=kwargs.get("dist_thr", self.kwargs.get("dist_thr", 1e-3)),
                                     restrict_to_boundary=kwargs.get("restrict_to_boundary", self.kwargs.get("restrict_to_boundary", True)))
