"""CPU oracle for the render-and-compare pose solver.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain PyTorch-CPU float32, the algorithm of the reference's batched
pose solver and of the PyTorch3D 0.7.1 pieces it calls.  It is the checker for the HIP
kernels: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
The product package (nemo_amd/) never imports anything from this directory.

PARITY UNPINNED for everything that renders: the arithmetic of the reference's render path
lives in pytorch3d==0.7.1 (reference environment.yml:168), which is neither vendored under
the reference nor installed here, and the reference ships no tests, fixtures or golden
vectors.  The PyTorch3D semantics below (camera conventions, naive rasteriser, perspective
correct barycentrics, face-attribute interpolation) are restated from the published
algorithm.  The dependency-free leaf functions (pose grids, tensor_linspace, rotation
matrices, normalize_features, flow_warp) ARE pinned against the reference's own files by
tests/golden/leaf_golden.npz (made by oracle/gen_golden.py from /root/reference).

Arithmetic contract (what makes face-index buffers reproducible bit for bit on a GPU):
every geometric quantity is float32, every operation is a separate IEEE-754 rounding (torch
elementwise kernels never contract a*b+c into an fma), sums are written with explicit
association, division and sqrt are correctly rounded, and sin/cos/sqrt are evaluated in float64
and rounded once to float32 (a <=1 ulp double result rounds to the correctly rounded float
except with probability ~2^-29, which makes the value platform independent; torch's
vectorised float32 sqrt is NOT correctly rounded, its add/sub/mul/div are).

Reference files followed (read as text):
  nemo/models/batch_solve_pose.py   loss_fg_bg :31-34, get_pre_render_samples :43-62,
                                    align_no_centered :65-90,
                                    get_init_pos_rendered_dim0 :128-164, solve_pose :194-372
  nemo/models/mesh_interpolate_module.py  MeshInterpolateModuleMesh :92-217
  nemo/utils/mesh.py                load_off :15, save_off :38, camera_position_to_spherical_angle :50,
                                    forward_interpolate :98-148, pre_process_mesh_pascal :162,
                                    vertex_memory_to_face_memory :173, rotation_theta :177, campos_to_R_T :222
  nemo/utils/general.py             tensor_linspace :11-40, get_param_samples :106-145
  nemo/utils/features.py            normalize_features :4-5
  nemo/utils/pose.py, evaluate.py   cal_rotation_matrix :56-59, pose_error :8-36
  data/create_cuboid_mesh.py        meshelize :22-97
  nemo/models/solve_pose.py         get_corr_pytorch :27-82, solve_pose :85-347, pre_compute_kp_coords :405-562
                                    (skimage.feature.peak_local_max of scikit-image 0.19.3 is restated from
                                    its published algorithm on scipy.ndimage: absent here, PARITY UNPINNED)
"""
import math

import numpy as np
import torch

KEPS = 1e-8  # pytorch3d rasterizer kEpsilon


# --------------------------------------------------------------------------------------
# small leaf utilities (pinned against the reference's own files)
# --------------------------------------------------------------------------------------
def tensor_linspace(start, end, steps=10):
    """nemo/utils/general.py:11-40 - per-element linspace along a new last dim."""
    w_shape = (1,) * start.dim() + (steps,)
    sw = torch.linspace(1, 0, steps=steps).to(start).view(w_shape)
    ew = torch.linspace(0, 1, steps=steps).to(start).view(w_shape)
    return sw * start.contiguous()[..., None] + ew * end.contiguous()[..., None]


def get_param_samples(inference_cfg):
    """nemo/utils/general.py:106-145.  `inference_cfg` is a nested dict with the YAML keys
    (azim_sample/elev_sample/theta_sample: min_pi,max_pi,num; dist/px/py: min,max,num)."""
    c = inference_cfg
    az = np.linspace(c["azim_sample"]["min_pi"] * np.pi, c["azim_sample"]["max_pi"] * np.pi,
                     c["azim_sample"]["num"], endpoint=False)
    el = np.linspace(c["elev_sample"]["min_pi"] * np.pi, c["elev_sample"]["max_pi"] * np.pi,
                     c["elev_sample"]["num"])
    th = np.linspace(c["theta_sample"]["min_pi"] * np.pi, c["theta_sample"]["max_pi"] * np.pi,
                     c["theta_sample"]["num"])
    di = np.linspace(c["dist_sample"]["min"], c["dist_sample"]["max"], c["dist_sample"]["num"])
    px = np.linspace(c["px_sample"]["min"], c["px_sample"]["max"], c["px_sample"]["num"])
    py = np.linspace(c["py_sample"]["min"], c["py_sample"]["max"], c["py_sample"]["num"])
    return az, el, th, di, px, py


def normalize_features(x, dim=0):
    """nemo/utils/features.py:4-5 (note the [0]: divides by the norm of the first slice)."""
    return x / torch.sum(x ** 2, dim=dim, keepdim=True)[0] ** 0.5


def cal_rotation_matrix(theta, elev, azim, dis):
    """nemo/utils/pose.py:6-59 (numpy float64)."""
    if dis <= 1e-10:
        dis = 0.5
    a, e = -azim, -(math.pi / 2 - elev)
    Rz = np.array([[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]])
    Rx = np.array([[1, 0, 0], [0, math.cos(e), -math.sin(e)], [0, math.sin(e), math.cos(e)]])
    Rt = np.array([[np.cos(theta), -np.sin(theta), 0], [np.sin(theta), np.cos(theta), 0], [0, 0, 1]])
    return Rt @ np.dot(Rx, Rz)


def pose_error(gt, pred):
    """nemo/utils/evaluate.py:8-36 - geodesic distance between two rotations (radians)."""
    from scipy.linalg import logm
    if pred is None:
        return np.pi
    A = cal_rotation_matrix(float(gt["theta"]), float(gt["elevation"]), float(gt["azimuth"]), 5.0)
    P = cal_rotation_matrix(float(pred["theta"]), float(pred["elevation"]), float(pred["azimuth"]), 5.0)
    if np.any(np.isnan(A)) or np.any(np.isnan(P)) or np.any(np.isinf(A)) or np.any(np.isinf(P)):
        return np.pi
    return ((logm(np.dot(np.transpose(P), A)) ** 2).sum()) ** 0.5 / (2.0 ** 0.5)


def flow_warp(im, flow):
    """nemo/utils/flow_warp.py:6-46 - translation warp, bilinear, border, align_corners=True."""
    N, _, H, W = flow.shape
    gx = torch.linspace(-1.0, 1.0, W).view(1, 1, 1, W).expand(N, -1, H, -1)
    gy = torch.linspace(-1.0, 1.0, H).view(1, 1, H, 1).expand(N, -1, -1, W)
    fl = torch.cat([flow[:, 0:1] / ((im.size(3) - 1.0) / 2.0), flow[:, 1:2] / ((im.size(2) - 1.0) / 2.0)], 1)
    grid = torch.cat([gx, gy], 1) + fl
    return torch.nn.functional.grid_sample(im, grid.permute(0, 2, 3, 1), mode="bilinear",
                                           padding_mode="border", align_corners=True)


# --------------------------------------------------------------------------------------
# meshes
# --------------------------------------------------------------------------------------
def meshelize(x_range, y_range, z_range, number_vertices):
    """data/create_cuboid_mesh.py:22-97 - six gridded cuboid faces, each with its own vertex grid."""
    w, h, d = x_range[1] - x_range[0], y_range[1] - y_range[0], z_range[1] - z_range[0]
    total_area = (w * h + h * d + w * d) * 2
    mesh_size = total_area / (number_vertices * 2)
    edge = (mesh_size * 2) ** .5
    xs = x_range[0] + np.linspace(0, w, int(w / edge + 1))
    ys = y_range[0] + np.linspace(0, h, int(h / edge + 1))
    zs = z_range[0] + np.linspace(0, d, int(d / edge + 1))
    verts, faces = [], []
    base = 0

    def grid(us, vs, make):
        nonlocal base
        nu, nv = len(us), len(vs)
        for n in range(nv):
            for m in range(nu):
                verts.append(make(us[m], vs[n]))
        for m in range(nv - 1):
            for n in range(nu - 1):
                faces.append((base + m * nu + n, base + m * nu + n + 1, base + (m + 1) * nu + n))
                faces.append((base + (m + 1) * nu + n + 1, base + m * nu + n + 1, base + (m + 1) * nu + n))
        base += nu * nv

    grid(xs, ys, lambda u, v: (u, v, zs[0]))
    grid(xs, ys, lambda u, v: (u, v, zs[-1]))
    grid(xs, zs, lambda u, v: (u, ys[0], v))
    grid(xs, zs, lambda u, v: (u, ys[-1], v))
    grid(ys, zs, lambda u, v: (xs[0], u, v))
    grid(ys, zs, lambda u, v: (xs[-1], u, v))
    return np.array(verts), np.array(faces)


def off_text(vertices, faces):
    """nemo/utils/mesh.py:38-47 (save_off) - returns the file text."""
    out = "OFF\n%d %d 0\n" % (vertices.shape[0], faces.shape[0])
    for v in vertices:
        out += "{:.16f} {:.16f} {:.16f}\n".format(v[0], v[1], v[2])
    for f in faces:
        out += "3 %d %d %d\n" % (f[0], f[1], f[2])
    return out


def parse_off(text):
    """nemo/utils/mesh.py:15-35 (load_off) on file text: float32 verts, int32 faces[:,1:]."""
    lines = text.splitlines(keepends=True)
    n_points = int(lines[1].split(" ")[0])
    v = np.array("".join(lines[2:2 + n_points]).split(), dtype=np.float32).reshape(-1, 3)
    f = np.array("".join(lines[2 + n_points:]).split(), dtype=np.int32).reshape(-1, 4)[:, 1:]
    return v, f


def cuboid_mesh(x_range=(-0.225, 0.225), y_range=(-0.5, 0.5), z_range=(-0.18, 0.18), number_vertices=1000):
    """Synthetic car-like cuboid (PASCAL3D+ CAD models are normalised to about unit length: 1.0 long
    is 75 px of the 100 px wide stride-8 map at distance 5, focal 375), round-tripped through the OFF
    text format like the reference's mesh files.  Default: V = 1118, F = 1904."""
    v, f = meshelize(x_range, y_range, z_range, number_vertices)
    return parse_off(off_text(v, f))


def pre_process_mesh_pascal(verts):
    """nemo/utils/mesh.py:162-170 - (x, y, z) -> (x, z, -y)."""
    return torch.cat((verts[:, 0:1], verts[:, 2:3], -verts[:, 1:2]), dim=1)


# --------------------------------------------------------------------------------------
# camera math (PyTorch3D semantics, explicit float32 op order)
# --------------------------------------------------------------------------------------
def _sincos32(x):
    """sin/cos evaluated in float64 and rounded once to float32 (differentiable)."""
    xd = x.double()
    return torch.sin(xd).float(), torch.cos(xd).float()


def camera_position_from_spherical_angles(distance, elevation, azimuth):
    """pytorch3d.renderer.camera_position_from_spherical_angles (radians): float32 inputs,
    x = d*cos(e)*sin(a), y = d*sin(e), z = d*cos(e)*cos(a)."""
    d = torch.as_tensor(distance, dtype=torch.float32).reshape(-1)
    e = torch.as_tensor(elevation, dtype=torch.float32).reshape(-1)
    a = torch.as_tensor(azimuth, dtype=torch.float32).reshape(-1)
    d, e, a = torch.broadcast_tensors(d, e, a)
    se, ce = _sincos32(e)
    sa, ca = _sincos32(a)
    return torch.stack([d * ce * sa, d * se, d * ce * ca], dim=1)


def camera_position_to_spherical_angle(camera_pose):
    """nemo/utils/mesh.py:50-56."""
    distance = torch.sum(camera_pose ** 2, axis=1) ** 0.5
    azimuth = torch.atan(camera_pose[:, 0] / camera_pose[:, 2]) % np.pi + np.pi * (
        camera_pose[:, 0] < 0).type(camera_pose.dtype)
    elevation = torch.asin(camera_pose[:, 1] / distance)
    return distance, elevation, azimuth


def _normalize3(x, y, z, eps=1e-5):
    """F.normalize over a 3-vector given as components: v / max(||v||, eps), norm = sqrt((x*x+y*y)+z*z)."""
    n = torch.sqrt(((x * x + y * y) + z * z).double()).float()   # torch's float32 sqrt is not correctly rounded
    n = torch.clamp(n, min=eps)
    return x / n, y / n, z / n


def _cross(ax, ay, az, bx, by, bz):
    return ay * bz - az * by, az * bx - ax * bz, ax * by - ay * bx


def look_at_rotation(campos):
    """pytorch3d.renderer.look_at_rotation with at=(0,0,0), up=(0,1,0).  Returns the nine
    entries r[i][j] (lists of [n] tensors): columns of R are the camera x, y, z axes."""
    cx, cy, cz = campos[:, 0], campos[:, 1], campos[:, 2]
    zx, zy, zz = _normalize3(0.0 - cx, 0.0 - cy, 0.0 - cz)
    one, zero = torch.ones_like(zx), torch.zeros_like(zx)
    xx, xy, xz = _normalize3(*_cross(zero, one, zero, zx, zy, zz))
    yx, yy, yz = _normalize3(*_cross(zx, zy, zz, xx, xy, xz))
    close = (xx.abs() <= 5e-3) & (xy.abs() <= 5e-3) & (xz.abs() <= 5e-3)
    if bool(close.any()):
        rx, ry, rz = _normalize3(*_cross(yx, yy, yz, zx, zy, zz))
        xx, xy, xz = torch.where(close, rx, xx), torch.where(close, ry, xy), torch.where(close, rz, xz)
    return [[xx, yx, zx], [xy, yy, zy], [xz, yz, zz]]


def campos_to_R_T(campos, theta):
    """nemo/utils/mesh.py:222-226: R = look_at(C) @ Rz(theta), T = -R^T C.
    Returns R as r[i][j] lists of [n] tensors and T as three [n] tensors."""
    r0 = look_at_rotation(campos)
    s, c = _sincos32(theta.reshape(-1))
    R = [[r0[i][0] * c + r0[i][1] * s, r0[i][1] * c - r0[i][0] * s, r0[i][2]] for i in range(3)]
    cx, cy, cz = campos[:, 0], campos[:, 1], campos[:, 2]
    T = [-((R[0][j] * cx + R[1][j] * cy) + R[2][j] * cz) for j in range(3)]
    return R, T


def R_T_tensors(campos, theta):
    R, T = campos_to_R_T(campos, theta)
    return torch.stack([torch.stack(r, dim=1) for r in R], dim=1), torch.stack(T, dim=1)


class Camera:
    """PerspectiveCameras(in_ndc=False, focal_length=f, principal_point=(px,py), image_size=(H,W))
    with RasterizationSettings(image_size=(H,W), blur_radius=0, faces_per_pixel=1, bin_size=0)."""

    def __init__(self, H, W, focal, principal=None):
        self.H, self.W, self.focal = int(H), int(W), float(focal)
        self.principal = (float(W // 2), float(H // 2)) if principal is None else principal


def pixel_centers(H, W):
    """pytorch3d PixToNonSquareNdc: NDC coordinate of each pixel row / column centre (float32).
    Row yi (from the top) uses index H-1-yi, column xi uses W-1-xi (+X left, +Y up)."""
    f32 = np.float32

    def ndc(i, S1, S2):
        rng = f32(2.0)
        if S1 > S2:
            rng = f32(f32(S1) * rng) / f32(S2)
        off = rng / f32(2.0)
        return -off + (rng * f32(i) + off) / f32(S1)

    yf = np.array([ndc(H - 1 - yi, H, W) for yi in range(H)], dtype=np.float32)
    xf = np.array([ndc(W - 1 - xi, W, H) for xi in range(W)], dtype=np.float32)
    return torch.from_numpy(xf), torch.from_numpy(yf)


def project_verts(verts, campos, theta, cam, principal=None, rt=None):
    """World verts [V,3] -> per camera view coords and NDC xy.  X_view = X @ R + T (row vectors);
    x_ndc = (a*X)/Z + bx with a = f/s, bx = (W/2 - px)/s, s = min(H,W)/2 (screen-space pinhole
    x_s = f X/Z + px followed by the screen->NDC map of a non-square image); depth = Z_view.
    rt = (R [n,3,3], T [n,3]): the caller's extrinsics instead of campos / theta."""
    if rt is not None:
        R = [[rt[0][:, i, j] for j in range(3)] for i in range(3)]
        T = [rt[1][:, j] for j in range(3)]
        campos = rt[1]                                   # only its leading size is used below
    else:
        R, T = campos_to_R_T(campos, theta)
    X0, X1, X2 = verts[None, :, 0], verts[None, :, 1], verts[None, :, 2]
    Xv = [((X0 * R[0][j][:, None] + X1 * R[1][j][:, None]) + X2 * R[2][j][:, None]) + T[j][:, None]
          for j in range(3)]
    n = campos.shape[0]
    if principal is None:
        principal = torch.tensor([cam.principal], dtype=torch.float32).expand(n, -1)
    s = torch.tensor(min(cam.H, cam.W) / 2.0, dtype=torch.float32)
    a = torch.tensor(cam.focal, dtype=torch.float32) / s
    bx = (torch.tensor(cam.W / 2.0, dtype=torch.float32) - principal[:, 0]) / s
    by = (torch.tensor(cam.H / 2.0, dtype=torch.float32) - principal[:, 1]) / s
    xn = (a * Xv[0]) / Xv[2] + bx[:, None]
    yn = (a * Xv[1]) / Xv[2] + by[:, None]
    return xn, yn, Xv[2], Xv


def _edge(px, py, ax, ay, bx, by):
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax)


def _face_eval(xf, yf, x0, y0, z0, x1, y1, z1, x2, y2, z2):
    """Barycentrics of pixel (xf,yf) in one face: pytorch3d BarycentricCoordsForward then
    BarycentricPerspectiveCorrectionForward; returns (b0,b1,b2,pz)."""
    area = _edge(x2, y2, x0, y0, x1, y1) + KEPS
    w0 = _edge(xf, yf, x1, y1, x2, y2) / area
    w1 = _edge(xf, yf, x2, y2, x0, y0) / area
    w2 = _edge(xf, yf, x0, y0, x1, y1) / area
    t0 = w0 * z1 * z2
    t1 = z0 * w1 * z2
    t2 = z0 * z1 * w2
    den = torch.clamp((t0 + t1) + t2, min=KEPS)
    b0, b1, b2 = t0 / den, t1 / den, t2 / den
    pz = (b0 * z0 + b1 * z1) + b2 * z2
    return b0, b1, b2, pz


@torch.no_grad()
def rasterize_naive(xn, yn, zv, faces, H, W, rows_per_chunk=8):
    """pytorch3d RasterizeMeshesNaive (blur 0, K=1, perspective_correct, no culling): per pixel the
    face with the smallest interpolated depth among faces whose corrected barycentrics are all > 0;
    ties go to the smaller face index.  Returns pix_to_face [n,H,W] int64 (local face index, -1 bg)."""
    n = xn.shape[0]
    fl = faces.long()
    xf, yf = pixel_centers(H, W)
    inf = torch.tensor(float("inf"))
    out = torch.full((n, H, W), -1, dtype=torch.int64)
    for i in range(n):
        x0, x1, x2 = xn[i, fl[:, 0]], xn[i, fl[:, 1]], xn[i, fl[:, 2]]
        y0, y1, y2 = yn[i, fl[:, 0]], yn[i, fl[:, 1]], yn[i, fl[:, 2]]
        z0, z1, z2 = zv[i, fl[:, 0]], zv[i, fl[:, 1]], zv[i, fl[:, 2]]
        xmin, xmax = torch.minimum(torch.minimum(x0, x1), x2), torch.maximum(torch.maximum(x0, x1), x2)
        ymin, ymax = torch.minimum(torch.minimum(y0, y1), y2), torch.maximum(torch.maximum(y0, y1), y2)
        zmax = torch.maximum(torch.maximum(z0, z1), z2)
        farea = _edge(x0, y0, x1, y1, x2, y2)
        face_ok = ~((zmax < 0) | ((farea <= KEPS) & (farea >= -KEPS)))     # behind the camera (rasterize_meshes.cu: zmax < 0) / zero area
        for r0 in range(0, H, rows_per_chunk):
            r1 = min(H, r0 + rows_per_chunk)
            px = xf[None, :, None].expand(r1 - r0, -1, 1)
            py = yf[r0:r1, None, None].expand(-1, W, 1)
            b0, b1, b2, pz = _face_eval(px, py, x0, y0, z0, x1, y1, z1, x2, y2, z2)
            outside = (px > xmax) | (px < xmin) | (py > ymax) | (py < ymin)
            ok = face_ok & ~outside & ~(pz < 0) & (b0 > 0) & (b1 > 0) & (b2 > 0)
            key = torch.where(ok, pz, inf)
            idx = torch.argmin(key, dim=-1)
            hit = torch.gather(ok, -1, idx[..., None])[..., 0]
            out[i, r0:r1] = torch.where(hit, idx, torch.full_like(idx, -1))
    return out


def fragments(verts, faces, campos, theta, cam, principal=None, rt=None):
    """Rasterise `n` cameras.  pix_to_face [n,H,W] (local face idx, -1 = background), bary
    [n,H,W,3] and zbuf [n,H,W] (differentiable w.r.t. campos/theta/principal; -1 on background)."""
    xn, yn, zv, _ = project_verts(verts, campos, theta, cam, principal, rt)
    p2f = rasterize_naive(xn.detach(), yn.detach(), zv.detach(), faces, cam.H, cam.W)
    n = xn.shape[0]
    xf, yf = pixel_centers(cam.H, cam.W)
    fsel = faces.long()[p2f.clamp(min=0)]                      # [n,H,W,3]
    ar = torch.arange(n)[:, None, None]
    g = lambda t, j: t[ar, fsel[..., j]]
    b0, b1, b2, pz = _face_eval(xf[None, None, :], yf[None, :, None],
                                g(xn, 0), g(yn, 0), g(zv, 0), g(xn, 1), g(yn, 1), g(zv, 1),
                                g(xn, 2), g(yn, 2), g(zv, 2))
    hit = p2f >= 0
    bary = torch.where(hit[..., None], torch.stack([b0, b1, b2], dim=-1), torch.full((), -1.0))
    zbuf = torch.where(hit, pz, torch.full((), -1.0))
    return p2f, bary, zbuf


def interpolate_face_attributes(p2f, bary, face_memory):
    """pytorch3d interpolate_face_attributes: sum_j bary_j * attr[face, j, :], zeros on background."""
    attr = face_memory[p2f.clamp(min=0)]                       # [n,H,W,3,C]
    out = (bary[..., 0, None] * attr[..., 0, :] + bary[..., 1, None] * attr[..., 1, :]) \
        + bary[..., 2, None] * attr[..., 2, :]
    return torch.where((p2f >= 0)[..., None], out, torch.zeros(()))


def set_bary_coords_to_nearest(bary):
    """nemo/utils/mesh.py:59-70: one-hot of the largest barycentric, formed as softmax + (one_hot - softmax), plus the
    negative parts of the input (the background's -1, which interpolate_face_attributes masks anyway)."""
    shape = bary.shape
    exr = bary * (bary < 0)
    flat = bary.reshape(-1, shape[-1])
    target = torch.zeros_like(flat).scatter(1, flat.argmax(1).unsqueeze(1), 1.0)
    soft = torch.softmax(flat, dim=1)
    return (soft + (target - soft).detach()).view(*shape) + exr      # straight-through: the gradient is the softmax's


class MeshInterpolateModuleRef:
    """nemo/models/mesh_interpolate_module.py:92-217 (single mesh, blur 0, faces_per_pixel 1)."""

    def __init__(self, vertices, faces, memory_bank, cam):
        self.verts = pre_process_mesh_pascal(torch.as_tensor(vertices, dtype=torch.float32))
        self.faces = torch.as_tensor(faces).to(torch.int32)
        self.cam = cam
        self.principal = None      # [n,2] tensor overrides cam.principal (solve_pose mutates it)
        self.update_memory(memory_bank)

    def update_memory(self, memory_bank):
        self.face_memory = torch.as_tensor(memory_bank, dtype=torch.float32)[self.faces.long()]

    def fragments(self, campos, theta):
        pr = self.principal
        if pr is not None and pr.shape[0] == 1 and campos.shape[0] > 1:
            pr = pr.expand(campos.shape[0], -1)
        return fragments(self.verts, self.faces, campos, theta, self.cam, pr)

    def __call__(self, campos, theta, mode="bilinear"):
        p2f, bary, _ = self.fragments(campos, theta)
        if mode == "nearest":
            bary = set_bary_coords_to_nearest(bary)
        return interpolate_face_attributes(p2f, bary, self.face_memory).permute(0, 3, 1, 2)


# --------------------------------------------------------------------------------------
# batched solver (nemo/models/batch_solve_pose.py)
# --------------------------------------------------------------------------------------
def loss_fg_bg(obj_s, clu_s, reduce_method=lambda x: torch.mean(x)):
    return torch.ones(1) - (reduce_method(torch.max(obj_s, clu_s)) - reduce_method(clu_s))


def clutter_score(feature_map, clutter_bank):
    """batch_solve_pose.py:214-225: 1x1 conv with each clutter vector, max over banks."""
    banks = clutter_bank if isinstance(clutter_bank, list) else [clutter_bank]
    score = None
    for cb in banks:
        s = torch.nn.functional.conv2d(feature_map, cb.unsqueeze(2).unsqueeze(3)).squeeze(1)
        score = s if score is None else torch.max(score, s)
    return score


def get_pre_render_samples(inter_module, azim_samples, elev_samples, theta_samples, distance_samples=(5,)):
    """batch_solve_pose.py:43-62: loop nesting azim -> elev -> theta -> dist, one render per pose."""
    with torch.no_grad():
        cs, ths, maps = [], [], []
        for a in azim_samples:
            for e in elev_samples:
                for t in theta_samples:
                    for d in distance_samples:
                        th = torch.ones(1) * t
                        C = camera_position_from_spherical_angles(d, e, a)
                        maps.append(inter_module(C, th))
                        cs.append(C)
                        ths.append(th)
        return torch.stack(maps), torch.stack(cs).squeeze(1), torch.cat(ths)


@torch.no_grad()
def grid_losses(samples_maps, predicted_maps, clutter_scores):
    """batch_solve_pose.py:128-158: loss[n, b] of every pre-rendered pose n against every image b."""
    P = samples_maps.squeeze(1)
    out = []
    for i in range(predicted_maps.shape[0]):
        obj = torch.einsum('nchw,chw->nhw', P, predicted_maps[i])
        clu = clutter_scores[None, i]
        red = lambda x: torch.mean(x, dim=(1, 2))
        out.append(torch.ones(1) - (red(torch.max(obj, clu)) - red(clu)))
    return torch.stack(out).T


def get_init_pos_rendered_dim0(samples_maps, samples_pos, samples_theta, predicted_maps, clutter_scores):
    """batch_solve_pose.py:128-164: arg-min (first minimum) over the pose grid."""
    loss = grid_losses(samples_maps, predicted_maps, clutter_scores)
    idx = torch.min(loss, dim=0)[1]
    return samples_pos[idx], samples_theta[idx], torch.min(loss, dim=0)[0], idx


@torch.no_grad()
def get_init_pos(inter_module, samples_pos, samples_theta, predicted_maps, clutter_scores=None, reset_distance=None):
    """batch_solve_pose.py:168-191 (pre_render: false): every grid pose is rendered when it is scored - once for the whole
    batch, or, with reset_distance [b], per image at that image's own distance along the pose's viewing direction.
    Without clutter scores the loss is loss_fg_only (:27-28).  -> (campos [b,3], theta [b], min loss [b], index [b])."""
    b = predicted_maps.shape[0]
    red = lambda x: torch.mean(x, dim=(1, 2))
    out = []
    for pos_, theta_ in zip(samples_pos, samples_theta):
        if reset_distance is not None:
            C = torch.nn.functional.normalize(pos_[None]) * reset_distance[:, None]
            maps_ = inter_module(C, theta_[None].expand(b))
        else:
            maps_ = inter_module(pos_[None], theta_[None]).expand(b, -1, -1, -1)
        obj = torch.einsum('nchw,nchw->nhw', maps_, predicted_maps)
        if clutter_scores is None:
            out.append(torch.ones(1) - red(obj))
        else:
            out.append(torch.ones(1) - (red(torch.max(obj, clutter_scores)) - red(clutter_scores)))
    loss = torch.stack(out)                       # [n, b]
    mn, idx = torch.min(loss, dim=0)
    return samples_pos[idx], samples_theta[idx], mn, idx


def align_no_centered(maps_source, distance_source, principal_source, maps_target_shape, distance_target,
                      principal_target, padding_mode='zeros'):
    """batch_solve_pose.py:65-90."""
    n, c, h1, w1 = maps_source.shape
    h0, w0 = maps_target_shape
    rr = (distance_source / distance_target).float()
    x_min = 2 * principal_source[:, 0] / w1 - w0 / (w1 * rr) - principal_target[:, 0] * 2 / w0
    x_max = 2 * principal_source[:, 0] / w1 + w0 / (w1 * rr) - principal_target[:, 0] * 2 / w0
    y_min = 2 * principal_source[:, 1] / h1 - h0 / (h1 * rr) - principal_target[:, 1] * 2 / h0
    y_max = 2 * principal_source[:, 1] / h1 + h0 / (h1 * rr) - principal_target[:, 1] * 2 / h0
    gx = tensor_linspace(x_min, x_max, int(w0))[:, None, :].expand(-1, int(h0), -1)
    gy = tensor_linspace(y_min, y_max, int(h0))[:, :, None].expand(-1, -1, int(w0))
    grids = torch.cat([gx[..., None], gy[..., None]], dim=3)
    return torch.nn.functional.grid_sample(maps_source, grids, padding_mode=padding_mode, align_corners=False)


def optimize_poses(inter, feature_map, clu, init_C, init_theta, init_principal=None, dof=3,
                   epochs=300, lr=0.05, betas=(0.4, 0.6), gamma=0.2, global_batch=None, record=False, bbox=None,
                   down_sample_rate=8):
    """batch_solve_pose.py:306-338 - the Adam loop.  `global_batch` rescales the batch-mean loss so
    a shard of a larger batch reproduces the gradients it would see inside that batch."""
    b = feature_map.shape[0]
    C = torch.nn.Parameter(init_C.clone())
    theta = torch.nn.Parameter(init_theta.clone())
    params = [C, theta]
    if dof == 6:
        principals = torch.nn.Parameter(init_principal.clone().float())
        inter.principal = principals
        params.append(principals)
    else:
        principals = init_principal
        inter.principal = init_principal
    optim = torch.optim.Adam(params, lr=lr, betas=betas)
    sched = torch.optim.lr_scheduler.ExponentialLR(optim, gamma=gamma)
    scale = 1.0 if global_batch is None else b / float(global_batch)
    hist = []
    object_score = None
    for epo in range(epochs):
        projected = inter(C, theta)
        object_score = torch.sum(projected * feature_map, dim=1)
        loss = loss_fg_bg(object_score, clu)
        if record:
            hist.append(dict(loss=float(loss.detach()), C=C.detach().clone(), theta=theta.detach().clone(),
                             per_image=torch.stack([loss_fg_bg(object_score[i, None].detach(), clu[i, None])[0]
                                                    for i in range(b)])))
        (loss * scale if scale != 1.0 else loss).backward()
        if record:
            hist[-1]["grad_C"] = C.grad.detach().clone()
            hist[-1]["grad_theta"] = theta.grad.detach().clone()
            if dof == 6:
                hist[-1]["grad_principal"] = principals.grad.detach().clone()
        optim.step()
        optim.zero_grad()
        if (epo + 1) % (epochs // 3) == 0:
            sched.step()
    inter.principal = None
    if bbox is not None:
        crop = lambda v: int(float(v) / down_sample_rate + .5)
        scores = torch.stack([object_loss_fg_bg(object_score[i].detach(), clu[i], (crop(bbox[i][0]), crop(bbox[i][1])),
                                                (crop(bbox[i][2]), crop(bbox[i][3])))[0] for i in range(b)])
    else:
        scores = torch.stack([loss_fg_bg(object_score[i, None].detach(), clu[i, None])[0] for i in range(b)])
    pr = principals.detach() if principals is not None else None
    return C.detach(), theta.detach(), pr, scores, hist


def object_loss_fg_bg(obj_s, clu_s, object_height, object_width):
    """batch_solve_pose.py:36-41."""
    obj_s = obj_s[object_height[0]:object_height[1], object_width[0]:object_width[1]]
    clu_s = clu_s[object_height[0]:object_height[1], object_width[0]:object_width[1]]
    return torch.ones(1) - (torch.mean(torch.max(obj_s, clu_s)) - torch.mean(clu_s))


def solve_pose(inter, feature_map, clutter_bank, cam_pos_pre_rendered, theta_pre_rendered,
               feature_pre_rendered, epochs=300, lr=0.05, betas=(0.4, 0.6), gamma=0.2,
               global_batch=None, record=False, principal=None, dof=3, bbox=None, distance_source=None,
               distance_target=None, down_sample_rate=8):
    """batch_solve_pose.py:194-372 with pre_render=True: dof 3/4 (centred or re-centred) and dof 6."""
    b = feature_map.shape[0]
    clu = clutter_score(feature_map, clutter_bank)
    H, W = inter.cam.H, inter.cam.W
    target = torch.tensor([[H, W]], dtype=torch.float32)
    centre = target.flip(1) / 2
    if dof in (3, 4):
        if principal is not None:
            t_fm = align_no_centered(feature_map, distance_source, principal, (H, W), distance_target,
                                     centre.expand(b, -1))
            t_clu = align_no_centered(clu[:, None], distance_source, principal, (H, W), distance_target,
                                      centre.expand(b, -1)).squeeze(1)
            init_principal = principal.float()
        else:
            t_fm, t_clu, init_principal = feature_map, clu, None
        init_C, init_theta, init_loss, idx = get_init_pos_rendered_dim0(
            feature_pre_rendered, cam_pos_pre_rendered, theta_pre_rendered, t_fm, t_clu)
    else:
        ones = torch.ones(b)
        all_C, all_t, all_l = [], [], []
        for pr in principal:
            pr_ = pr[None].expand(b, -1).float()
            t_fm = align_no_centered(feature_map, ones, pr_, (H, W), ones, centre.expand(b, -1), padding_mode='border')
            t_clu = align_no_centered(clu[:, None], ones, pr_, (H, W), ones, centre.expand(b, -1),
                                      padding_mode='border').squeeze(1)
            c_, t_, l_, _ = get_init_pos_rendered_dim0(feature_pre_rendered, cam_pos_pre_rendered, theta_pre_rendered,
                                                        t_fm, t_clu)
            all_C.append(c_); all_t.append(t_); all_l.append(l_)
        use = torch.min(torch.stack(all_l), dim=0)[1]
        ar = torch.arange(b)
        init_C, init_theta = torch.stack(all_C)[use, ar], torch.stack(all_t)[use, ar]
        init_principal = principal[use].float()
        init_loss, idx = torch.stack(all_l)[use, ar], use
    if principal is not None and dof == 3:
        init_C = init_C / init_C.pow(2).sum(-1).pow(.5)[..., None] * distance_source[..., None].float()
    C, theta, pr_out, scores, hist = optimize_poses(inter, feature_map, clu, init_C, init_theta, init_principal, dof,
                                                    epochs, lr, betas, gamma, global_batch, record, bbox=bbox,
                                                    down_sample_rate=down_sample_rate)
    dist, elev, azim = camera_position_to_spherical_angle(C)
    px, py = inter.cam.principal
    preds = []
    for i in range(b):
        pp = [px, py] if pr_out is None else [pr_out[i, 0].item(), pr_out[i, 1].item()]
        preds.append(dict(final=[dict(azimuth=azim[i].item(), elevation=elev[i].item(), theta=theta[i].item(),
                                      distance=dist[i].item(), principal=pp, score=scores[i].item())]))
    return preds, dict(init_idx=idx, init_loss=init_loss, init_C=init_C, init_theta=init_theta, C=C, theta=theta,
                       scores=scores, clutter=clu, hist=hist)


# --------------------------------------------------------------------------------------
# keypoint-initialised solver (nemo/models/solve_pose.py)
# --------------------------------------------------------------------------------------
def pre_compute_kp_coords(xvert, azimuth_samples, elevation_samples, theta_samples, distance_samples, viewport=3000):
    """solve_pose.py:405-562 in the reference's f64 numpy arithmetic; kp_vis is all ones there."""
    poses = np.array([[a, e, t, d] for a in azimuth_samples for e in elevation_samples for t in theta_samples
                      for d in distance_samples], dtype=np.float32)
    kp_coords = np.zeros((len(poses), len(xvert), 2), dtype=np.float32)
    for i, pose_ in enumerate(poses):
        azim_, elev_, theta_, dist_ = pose_
        C = np.zeros((3, 1))
        C[0] = dist_ * math.cos(elev_) * math.sin(azim_)
        C[1] = -dist_ * math.cos(elev_) * math.cos(azim_)
        C[2] = dist_ * math.sin(elev_)
        azimuth = -azim_
        elevation = -(math.pi / 2 - elev_)
        Rz = np.array([[math.cos(azimuth), -math.sin(azimuth), 0], [math.sin(azimuth), math.cos(azimuth), 0], [0, 0, 1]])
        Rx = np.array([[1, 0, 0], [0, math.cos(elevation), -math.sin(elevation)], [0, math.sin(elevation), math.cos(elevation)]])
        R_rot = np.dot(Rx, Rz)
        R = np.hstack((R_rot, np.dot(-R_rot, C)))
        P = np.array([[viewport, 0, 0], [0, viewport, 0], [0, 0, -1]])
        x3d_ = np.hstack((xvert, np.ones((len(xvert), 1)))).T
        x3d_ = np.dot(R, x3d_)
        x2d = np.dot(P, x3d_)
        x2d[0, :] = x2d[0, :] / x2d[2, :]
        x2d[1, :] = x2d[1, :] / x2d[2, :]
        x2d = x2d[0:2, :]
        R2d = np.array([[math.cos(theta_), -math.sin(theta_)], [math.sin(theta_), math.cos(theta_)]])
        x2d = np.dot(R2d, x2d).T
        x2d[:, 1] *= -1
        kp_coords[i] = x2d
    poses = poses.reshape(len(azimuth_samples), len(elevation_samples), len(theta_samples), len(distance_samples), 4)
    return poses, kp_coords, np.ones(kp_coords.shape[:2], dtype=np.float32)


def get_corr(px_samples, py_samples, kpt_score_map, kp_coords, kp_vis, down_sample_rate, hm_h, hm_w):
    """solve_pose.py:27-82 without the chunking: corr [n_pose, n_px, n_py]."""
    px_s, py_s = torch.from_numpy(np.asarray(px_samples)), torch.from_numpy(np.asarray(py_samples))
    kpc = torch.from_numpy(np.asarray(kp_coords)).clone()
    kpv = torch.from_numpy(np.asarray(kp_vis)).clone()
    kps = torch.as_tensor(kpt_score_map)
    xv, yv = torch.meshgrid(px_s, py_s, indexing="ij")
    principal_samples = torch.stack([xv, yv], dim=2).reshape(-1, 1, 2).repeat(1, kpc.shape[1], 1)
    kpc = kpc.unsqueeze(1).repeat(1, principal_samples.shape[0], 1, 1)
    kpc += principal_samples
    kpc = kpc.reshape(-1, kpc.shape[2], 2)
    kpc = torch.round(kpc / down_sample_rate)
    kpv = kpv.unsqueeze(1).repeat(1, principal_samples.shape[0], 1)
    kpv = kpv.reshape(-1, kpv.shape[2])
    kpv[kpc[:, :, 0] < 0] = 0
    kpv[kpc[:, :, 0] >= hm_w - 1] = 0
    kpv[kpc[:, :, 1] < 0] = 0
    kpv[kpc[:, :, 1] >= hm_h - 1] = 0
    kpc[:, :, 0] = torch.clamp(kpc[:, :, 0], min=0, max=hm_w - 1)
    kpc[:, :, 1] = torch.clamp(kpc[:, :, 1], min=0, max=hm_h - 1)
    kpc = (kpc[:, :, 1:2] * hm_w + kpc[:, :, 0:1]).long()
    corr = torch.take_along_dim(kps.unsqueeze(0), kpc, dim=2)[:, :, 0]
    corr = torch.sum(corr * kpv, dim=1)
    return corr.reshape(-1, len(px_s), len(py_s)).numpy()


def peak_local_max(image, min_distance=1):
    """skimage.feature.peak_local_max(image, min_distance, exclude_border=False) of scikit-image 0.19.3,
    restated on scipy.ndimage (default thresholds: strictly above the image minimum; footprint
    (2*min_distance+1)^2, maximum filter with mode='nearest'; no peaks in a flat image; highest first;
    ensure_spacing drops a peak closer than min_distance (Chebyshev, strict) to a higher kept peak)."""
    from scipy import ndimage as ndi
    image = np.asarray(image)
    size = 2 * min_distance + 1
    if image.size == 1:
        mask = image > image.min()
    else:
        mask = image == ndi.maximum_filter(image, footprint=np.ones((size, size), dtype=bool), mode="nearest")
        if np.all(mask):
            mask[:] = False
        mask &= image > image.min()
    coord = np.nonzero(mask)
    idx = np.argsort(-image[coord], kind="stable")
    coord = np.transpose(coord)[idx]
    kept = []
    for c in coord:
        if all(np.abs(c - k).max() >= min_distance for k in kept):
            kept.append(c)
    return np.array(kept, dtype=np.int64).reshape(-1, 2)


def keypoint_proposals(corr, poses, px_samples, py_samples, search_translation, min_distance, pre_rendering_thr):
    """solve_pose.py:161-207."""
    corr2d = corr.reshape(-1, len(px_samples), len(py_samples))
    corr2d_max = np.max(corr2d, axis=0)
    extrema_2d = peak_local_max(corr2d_max, min_distance=min_distance) if search_translation else [[0, 0]]
    larger_thr_ext_2d, max_e, max_v = [], None, 0
    for e in extrema_2d:
        if corr2d_max[e[0], e[1]] >= pre_rendering_thr:
            larger_thr_ext_2d.append(e)
        if corr2d_max[e[0], e[1]] > max_v:
            max_v = corr2d_max[e[0], e[1]]
            max_e = e
    extrema = []
    for e in (larger_thr_ext_2d if len(larger_thr_ext_2d) > 0 else [max_e]):
        c = corr2d[:, e[0], e[1]].reshape(poses.shape[0], poses.shape[1], poses.shape[2], poses.shape[3])
        e_azim, e_elev, e_the, e_dist = np.unravel_index(np.argmax(c, axis=None), c.shape)
        p = poses[e_azim, e_elev, e_the, e_dist]
        extrema.append(dict(azimuth=p[0], elevation=p[1], theta=p[2], distance=p[3], px=px_samples[e[0]], py=py_samples[e[1]]))
    return extrema, corr2d_max


def refine_proposal(inter, feature_map, clu, C0, theta0, flow0, optimize_translation=True, translation_scale=0.1,
                    epochs=300, lr=0.05, betas=(0.4, 0.6), record=False):
    """solve_pose.py:223-324 for one proposal: Adam on (C [1,3], theta, flow [2]); the scheduler is built but
    never stepped there, so the learning rate is constant.  feature_map [1,C,H,W], clu [1,H,W]."""
    H, W = feature_map.shape[2:]
    C = torch.nn.Parameter(C0.clone().view(1, 3))
    theta = torch.nn.Parameter(theta0.clone().view(()))
    flow = torch.nn.Parameter(flow0.clone().view(2))
    params = [C, theta] + ([flow] if optimize_translation else [])
    optim = torch.optim.Adam(params, lr=lr, betas=betas)
    hist = []
    loss = None
    for epo in range(epochs):
        projected = inter(C, theta.view(1))[0]                                   # [C,H,W]
        flow_map = flow.view(1, 2, 1, 1).repeat(1, 1, H, W)
        projected = flow_warp(projected.unsqueeze(0), flow_map / translation_scale)[0]
        object_score = torch.sum(projected * feature_map[0], dim=0)
        loss = torch.ones(1) - (torch.mean(torch.max(object_score, clu[0])) - torch.mean(clu[0]))
        if record:
            hist.append(float(loss.detach()))
        loss.backward()
        optim.step()
        optim.zero_grad()
    return C.detach()[0], theta.detach(), flow.detach(), float(loss.detach()), hist


def solve_pose_keypoint(inter, feature_map, kp_features, clutter_bank, poses, kp_coords, kp_vis, px_samples, py_samples,
                        down_sample_rate=8, search_translation=True, optimize_translation=True, min_distance=1,
                        pre_rendering_thr=80, translation_scale=0.1, epochs=300, lr=0.05, betas=(0.4, 0.6)):
    """solve_pose.py:85-347 (debug outputs omitted).  Returns (final, refined list, extrema, corr)."""
    _, c, hm_h, hm_w = feature_map.shape
    clu = clutter_score(feature_map, clutter_bank)
    kpt_score_map = torch.einsum("vc,chw->vhw", kp_features, feature_map[0]).reshape(kp_features.shape[0], -1)
    corr = get_corr(px_samples, py_samples, kpt_score_map, kp_coords, kp_vis, down_sample_rate, hm_h, hm_w)
    extrema, _ = keypoint_proposals(corr, poses, px_samples, py_samples, search_translation, min_distance, pre_rendering_thr)
    refined = []
    for e in extrema:
        C0 = camera_position_from_spherical_angles(torch.tensor([float(e["distance"])]), torch.tensor([float(e["elevation"])]),
                                                   torch.tensor([float(e["azimuth"])]))
        flow0 = torch.tensor([-(e["px"] - hm_w * down_sample_rate / 2) / down_sample_rate * translation_scale,
                              -(e["py"] - hm_h * down_sample_rate / 2) / down_sample_rate * translation_scale], dtype=torch.float32)
        Cr, thr, fl, score, _ = refine_proposal(inter, feature_map, clu, C0, torch.tensor(float(e["theta"])), flow0,
                                                optimize_translation, translation_scale, epochs, lr, betas)
        d, el, az = camera_position_to_spherical_angle(Cr.view(1, 3))
        px_pred, py_pred = -float(fl[0]) / translation_scale, -float(fl[1]) / translation_scale
        refined.append(dict(azimuth=float(az), elevation=float(el), theta=float(thr), distance=float(d), px=px_pred, py=py_pred,
                            principal=[px_pred * down_sample_rate + hm_w * down_sample_rate / 2,
                                       py_pred * down_sample_rate + hm_h * down_sample_rate / 2], score=score))
    final = refined if len(refined) == 1 else min(refined, key=lambda r: r["score"])
    return final, refined, extrema, corr


# --------------------------------------------------------------------------------------
# synthetic workload (seeded; no dataset)
# --------------------------------------------------------------------------------------
def synthetic_bank(V, C=128, n_clutter=2560, seed=0):
    g = torch.Generator().manual_seed(seed)
    mem = torch.nn.functional.normalize(torch.randn(V, C, generator=g), dim=1)
    clutter_rows = torch.nn.functional.normalize(torch.randn(n_clutter, C, generator=g), dim=1)
    clutter = normalize_features(torch.mean(clutter_rows, dim=0)).unsqueeze(0)
    return mem, clutter


def synthetic_pose(seed):
    g = torch.Generator().manual_seed(1000 + seed)
    u = torch.rand(3, generator=g)
    return dict(azimuth=float(u[0] * 2 * math.pi), elevation=float(-math.pi / 6 + u[1] * math.pi / 2),
                theta=float(-math.pi / 6 + u[2] * math.pi / 3), distance=5.0)


def synthetic_feature_map(inter, clutter, seed, noise=0.5):
    """An "image" feature map: the bank rendered at a seeded ground-truth pose, clutter direction on
    the background, gaussian noise, L2-normalised over channels (the backbone's output contract)."""
    gt = synthetic_pose(seed)
    C = camera_position_from_spherical_angles(gt["distance"], gt["elevation"], gt["azimuth"])
    with torch.no_grad():
        p2f, bary, _ = inter.fragments(C, torch.tensor([gt["theta"]]))
        m = interpolate_face_attributes(p2f, bary, inter.face_memory)[0]       # [H,W,C]
    m = torch.where((p2f[0] >= 0)[..., None], m, clutter[0][None, None, :].expand_as(m))
    g = torch.Generator().manual_seed(2000 + seed)
    m = m + noise * torch.randn(m.shape, generator=g) / math.sqrt(m.shape[-1])
    m = torch.nn.functional.normalize(m, dim=-1)
    return m.permute(2, 0, 1).contiguous(), gt


# --------------------------------------------------------------------------------------
# training-side keypoint annotation (nemo/models/project_kp.py:175-247, one mesh per pose)
# --------------------------------------------------------------------------------------
def project_keypoints(verts, faces, campos, theta, cam, principal=None, restrict_to_boundary=True, dist_thr=1e-3, rt=None):
    """get_one_standard (project_kp.py:175-247) with func_single: -> (project_verts [n,V,2] as (y, x) in
    feature-map pixels, vis_mask [n,V] bool, margin [n,V] = | |sampled - true| - dist_thr | for tests that
    want to skip vertices sitting on the threshold).  The PyTorch3D pinhole is restated as
    x_t = (f X + px Z) / Z (PerspectiveCameras(in_ndc=False).transform_points: K then divide by w)."""
    R, T = R_T_tensors(campos, theta) if rt is None else rt      # X_view = X @ R + T
    n, V = R.shape[0], verts.shape[0]
    pr = torch.tensor([cam.principal], dtype=torch.float32).expand(n, 2) if principal is None else principal.float()
    cam_loc = -torch.matmul(torch.inverse(R), T[..., None])[:, :, 0]                       # :183
    _, _, _, Xv = project_verts(verts, campos, theta, cam, principal, rt)
    xt = (cam.focal * Xv[0] + pr[:, 0:1] * Xv[2]) / Xv[2]
    yt = (cam.focal * Xv[1] + pr[:, 1:2] * Xv[2]) / Xv[2]
    project = torch.stack([2 * pr[:, 1:2] - yt, 2 * pr[:, 0:1] - xt], dim=-1)            # :186-188, (y, x)
    size = torch.tensor([[float(cam.H), float(cam.W)]])
    if restrict_to_boundary:                                                               # :194-197
        project = torch.min(project, (size.unsqueeze(1) - 1) * torch.ones_like(project))
        project = torch.max(project, torch.zeros_like(project))
    p2f, bary, _ = fragments(verts, faces, campos, theta, cam, principal, rt)
    true_dist = (cam_loc[:, None] - verts[None]).pow(2).sum(-1).pow(.5)                   # [n,V] :201
    face_dist = true_dist[:, faces.long()]                                                # [n,F,3] :202
    depth = torch.zeros(n, cam.H, cam.W)
    for i in range(n):
        cov = p2f[i] >= 0
        fd = face_dist[i][p2f[i][cov]]                                                    # [m,3]
        b = bary[i][cov].detach()
        depth[i][cov] = (b[:, 0] * fd[:, 0] + b[:, 1] * fd[:, 1]) + b[:, 2] * fd[:, 2]     # interpolate_face_attributes :209
    grid = project[:, None] / size * 2 - 1                                                # :211
    sampled = torch.nn.functional.grid_sample(depth[:, None], grid.flip(-1), align_corners=False, mode="nearest")[:, 0, 0, :]
    diff = torch.abs(sampled - true_dist)
    return project, diff < dist_thr, torch.abs(diff - dist_thr)
