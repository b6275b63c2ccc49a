"""Mesh / camera helpers of the pose-solver path (host side).

Counterparts in the reference: nemo/utils/mesh.py (load_off :15, save_off :38,
camera_position_to_spherical_angle :50, pre_process_mesh_pascal :162,
vertex_memory_to_face_memory :173, rotation_theta :177, campos_to_R_T :222),
pytorch3d.renderer.camera_position_from_spherical_angles, data/create_cuboid_mesh.py (meshelize :22-97).
"""
import math

import numpy as np
import torch


def load_off(off_file_name, to_torch=False):
    """OFF reader: float32 vertices [V,3]; faces are the last three columns of the (n,4) int32 rows."""
    with open(off_file_name) as fh:
        lines = fh.readlines()
    n_points = int(lines[1].split(" ")[0])
    verts = np.array("".join(lines[2:2 + n_points]).split(), dtype=np.float32).reshape(-1, 3)
    faces = np.array("".join(lines[2 + n_points:]).split(), dtype=np.int32).reshape(-1, 4)[:, 1:]
    if to_torch:
        return torch.from_numpy(verts), torch.from_numpy(np.ascontiguousarray(faces))
    return verts, faces


def save_off(off_file_name, vertices, faces):
    rows = ["OFF", "%d %d 0" % (vertices.shape[0], faces.shape[0])]
    rows += ["{:.16f} {:.16f} {:.16f}".format(v[0], v[1], v[2]) for v in vertices]
    rows += ["3 %d %d %d" % (f[0], f[1], f[2]) for f in faces]
    with open(off_file_name, "w") as fh:
        fh.write("\n".join(rows) + "\n")


def meshelize(x_range, y_range, z_range, number_vertices):
    """Cuboid with six independently gridded sides (about `number_vertices` vertices in total)."""
    ext = [r[1] - r[0] for r in (x_range, y_range, z_range)]
    w, h, d = ext
    area = 2 * (w * h + h * d + w * d)
    edge = (2 * (area / (number_vertices * 2))) ** .5
    axes = [lo[0] + np.linspace(0, e, int(e / edge + 1)) for lo, e in zip((x_range, y_range, z_range), ext)]
    xs, ys, zs = axes
    verts, faces = [], []

    def side(us, vs, put):
        base, nu, nv = len(verts), len(us), len(vs)
        verts.extend(put(us[m], vs[n]) for n in range(nv) for m in range(nu))
        for m in range(nv - 1):
            for n in range(nu - 1):
                a = base + m * nu + n
                faces.append((a, a + 1, a + nu))
                faces.append((a + nu + 1, a + 1, a + nu))

    side(xs, ys, lambda u, v: (u, v, zs[0]))
    side(xs, ys, lambda u, v: (u, v, zs[-1]))
    side(xs, zs, lambda u, v: (u, ys[0], v))
    side(xs, zs, lambda u, v: (u, ys[-1], v))
    side(ys, zs, lambda u, v: (xs[0], u, v))
    side(ys, zs, lambda u, v: (xs[-1], u, v))
    return np.array(verts), np.array(faces)


def pre_process_mesh_pascal(verts, *args):
    """PASCAL3D+ axes -> render axes: (x, y, z) -> (x, z, -y)."""
    if torch.is_tensor(verts):
        out = torch.stack((verts[:, 0], verts[:, 2], -verts[:, 1]), dim=1)
    else:
        out = np.stack((verts[:, 0], verts[:, 2], -verts[:, 1]), axis=1)
    return out if not args else (out,) + tuple(args)


def vertex_memory_to_face_memory(memory_bank, faces):
    return memory_bank[faces.type(torch.long)]


def camera_position_from_spherical_angles(distance, elevation, azimuth, degrees=False, device="cpu"):
    """[N,3] float32 camera positions; sin/cos evaluated in float64 and rounded once."""
    d, e, a = (torch.as_tensor(t, dtype=torch.float32).reshape(-1) for t in (distance, elevation, azimuth))
    d, e, a = torch.broadcast_tensors(d, e, a)
    if degrees:
        e, a = e * (math.pi / 180.0), a * (math.pi / 180.0)
    ed, ad = e.double(), a.double()
    ce, se, ca, sa = (t.float() for t in (torch.cos(ed), torch.sin(ed), torch.cos(ad), torch.sin(ad)))
    return torch.stack([d * ce * sa, d * se, d * ce * ca], dim=1).to(device)


def camera_position_to_spherical_angle(camera_pose):
    """(distance, elevation, azimuth) of camera positions [n,3]; device tensors go through the HIP kernel."""
    if camera_pose.is_cuda:
        from .. import ops
        out = ops.spherical_angles(camera_pose)
        return out[:, 0], out[:, 1], out[:, 2]
    dist = torch.sum(camera_pose ** 2, axis=1) ** 0.5
    azim = torch.atan(camera_pose[:, 0] / camera_pose[:, 2]) % np.pi + np.pi * (camera_pose[:, 0] < 0).type(
        camera_pose.dtype)
    elev = torch.asin(camera_pose[:, 1] / dist)
    return dist, elev, azim
