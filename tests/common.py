"""Shared builders for the parity tests: the same seeded inputs go to the oracle (CPU) and to the
HIP path (through the C ABI)."""
import math

import numpy as np
import torch

from oracle import refsolver as R

GRID_CFG = dict(azim_sample=dict(min_pi=0.0, max_pi=2.0, num=12), elev_sample=dict(min_pi=-0.1667, max_pi=0.3333, num=4),
                theta_sample=dict(min_pi=-0.1667, max_pi=0.1667, num=3), dist_sample=dict(min=5.0, max=5.0, num=1),
                px_sample=dict(min=0.0, max=0.0, num=1), py_sample=dict(min=0.0, max=0.0, num=1))


class Scene:
    """Mesh + bank + camera, oracle side and device side."""

    def __init__(self, H, W, C, number_vertices, focal, device=None, seed=0, n_clutter=64):
        self.H, self.W, self.C = H, W, C
        self.verts, self.faces = R.cuboid_mesh(number_vertices=number_vertices)
        self.mem, self.clutter = R.synthetic_bank(self.verts.shape[0], C, n_clutter=n_clutter, seed=seed)
        self.cam = R.Camera(H, W, focal)
        self.ref = R.MeshInterpolateModuleRef(self.verts, self.faces, self.mem, self.cam)
        self.device = device
        if device is not None:
            import nemo_amd
            cams = nemo_amd.PerspectiveCameras(focal_length=focal, principal_point=((W // 2, H // 2),),
                                               image_size=((H, W),), in_ndc=False)
            rs = nemo_amd.RasterizationSettings(image_size=(H, W), blur_radius=0.0, faces_per_pixel=1, bin_size=0)
            self.mod = nemo_amd.MeshInterpolateModule(torch.from_numpy(self.verts), torch.from_numpy(self.faces), self.mem,
                                                      rasterizer=nemo_amd.MeshRasterizer(cameras=cams, raster_settings=rs),
                                                      post_process=None).to(device)

    def feature_maps(self, B, first_seed=0):
        maps, gts = [], []
        for i in range(B):
            m, gt = R.synthetic_feature_map(self.ref, self.clutter, first_seed + i)
            maps.append(m)
            gts.append(gt)
        return torch.stack(maps), gts


def random_poses(n, seed=0, dist=(4.0, 6.0)):
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(n, 4, generator=g)
    az = u[:, 0] * 2 * math.pi
    el = -math.pi / 6 + u[:, 1] * math.pi / 2
    th = -math.pi / 6 + u[:, 2] * math.pi / 3
    d = dist[0] + u[:, 3] * (dist[1] - dist[0])
    return R.camera_position_from_spherical_angles(d, el, az), th.float()


def angle_diff_deg(a, b):
    d = (a - b + math.pi) % (2 * math.pi) - math.pi
    return abs(d) * 180.0 / math.pi


def categories(n, C, seed=0, number_vertices=lambda i: 110 + 9 * i, n_clutter=64):
    """n cuboids of different extents and vertex counts + one bank holding their vertex features back to back.
    -> (meshes [(verts, faces)], vertex counts, bank rows [sum V_i, C], clutter [1, C])."""
    rng = np.random.RandomState(seed)
    meshes = []
    for i in range(n):
        ex = 0.14 + 0.12 * rng.rand(3)                      # half extents
        ex[1] += 0.15                                       # longer along y, like a vehicle
        v, f = R.cuboid_mesh((-ex[0], ex[0]), (-ex[1], ex[1]), (-ex[2], ex[2]), number_vertices=number_vertices(i))
        meshes.append((v, f))
    g = torch.Generator().manual_seed(seed + 5)
    counts = [m[0].shape[0] for m in meshes]
    mem = torch.nn.functional.normalize(torch.randn(sum(counts), C, generator=g), dim=1)
    rows = torch.nn.functional.normalize(torch.randn(n_clutter, C, generator=g), dim=1)
    clutter = R.normalize_features(torch.mean(rows, dim=0)).unsqueeze(0)
    return meshes, counts, mem, clutter


def full_size_vertex_counts(i):
    """Requested vertex counts of the 12 full-size category cuboids: their face counts straddle 1024 and 2048."""
    return 560 + 70 * i
