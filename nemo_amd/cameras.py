"""Minimal stand-ins for the three PyTorch3D value objects the reference hands to
MeshInterpolateModule (built at reference nemo/models/nemo.py:258-280 from
config/models/nemo_runtime.yaml:40-54).  They carry parameters only; all arithmetic is in the
HIP kernels.  Attribute names follow PyTorch3D because batch_solve_pose.solve_pose reads and
mutates `rasterizer.cameras.image_size / principal_point / _N`
(reference nemo/models/batch_solve_pose.py:237,241,244,295,310,314).
"""
import torch


class PerspectiveCameras:
    def __init__(self, focal_length=1.0, principal_point=((0.0, 0.0),), image_size=None, in_ndc=False,
                 device="cpu", **kwargs):
        if in_ndc:
            raise NotImplementedError("only screen-space cameras (in_ndc=False) are supported, as the reference uses")
        if image_size is None:
            raise ValueError("image_size=((H, W),) is required for a screen-space camera")
        self.device = torch.device(device)
        self.image_size = torch.as_tensor(image_size, dtype=torch.float32, device=self.device).reshape(-1, 2)
        self.principal_point = torch.as_tensor(principal_point, dtype=torch.float32, device=self.device).reshape(-1, 2)
        fl = torch.as_tensor(focal_length, dtype=torch.float32, device=self.device).reshape(-1)
        if fl.numel() != 1 and not bool((fl == fl[0]).all()):
            raise NotImplementedError("a single focal length (fx == fy, shared by all cameras) is supported")
        self.focal_length = fl[:1].reshape(1, 1)
        self._N = 1
        self._in_ndc = False

    def in_ndc(self):
        return self._in_ndc

    def to(self, device):
        self.device = torch.device(device)
        self.image_size = self.image_size.to(device)
        self.focal_length = self.focal_length.to(device)
        if torch.is_tensor(self.principal_point):
            self.principal_point = self.principal_point.to(device)
        return self

    def cuda(self):
        return self.to("cuda")

    def get_image_size(self):
        return self.image_size


class RasterizationSettings:
    def __init__(self, image_size=256, blur_radius=0.0, faces_per_pixel=1, bin_size=None, max_faces_per_bin=None,
                 perspective_correct=None, clip_barycentric_coords=None, cull_backfaces=False, z_clip_value=None,
                 cull_to_frustum=False, **kwargs):
        self.image_size = image_size
        self.blur_radius = blur_radius
        self.faces_per_pixel = faces_per_pixel
        self.bin_size = bin_size
        self.max_faces_per_bin = max_faces_per_bin
        self.perspective_correct = perspective_correct
        self.clip_barycentric_coords = clip_barycentric_coords
        self.cull_backfaces = cull_backfaces
        self.z_clip_value = z_clip_value
        self.cull_to_frustum = cull_to_frustum


class MeshRasterizer:
    """Parameter holder: (cameras, raster_settings).  MeshInterpolateModule reads both."""

    def __init__(self, cameras=None, raster_settings=None):
        self.cameras = cameras
        self.raster_settings = raster_settings if raster_settings is not None else RasterizationSettings()

    def to(self, device):
        if self.cameras is not None:
            self.cameras = self.cameras.to(device)
        return self
