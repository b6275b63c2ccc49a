"""CPU tests (no GPU): the oracle against the reference-made leaf fixtures and known answers, the host
logic of the product package, the C-ABI library's symbol table, and the 2-process gather path (gloo)."""
import ctypes
import math
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import refsolver as R
from tests.common import GRID_CFG, Scene, angle_diff_deg, random_poses

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def leaf():
    return np.load(os.path.join(GOLD, "leaf_golden.npz"))


CFG6D = dict(GRID_CFG, dist_sample=dict(min=3.0, max=9.0, num=4), px_sample=dict(min=0.25, max=0.75, num=5),
             py_sample=dict(min=0.25, max=0.75, num=5))


# ---- leaf functions pinned to the reference's own files -----------------------------------------
@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_pose_grid_matches_reference(leaf, impl):
    from nemo_amd.utils import get_param_samples
    for name, cfg in (("3d", GRID_CFG), ("6d", CFG6D)):
        got = R.get_param_samples(cfg) if impl == "oracle" else get_param_samples({"inference": cfg})
        for k, arr in zip(("azim", "elev", "theta", "dist", "px", "py"), got):
            assert np.array_equal(np.asarray(arr), leaf[f"grid_{name}_{k}"]), (name, k)
    # the YAML decimals are not pi/6, pi/3: the grid must come from the decimals
    assert abs(leaf["grid_3d_elev"][0] - (-0.1667 * math.pi)) < 1e-15
    assert abs(leaf["grid_3d_elev"][0] + math.pi / 6) > 1e-5


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_leaf_functions_match_reference(leaf, impl):
    import nemo_amd.utils as U
    mod = R if impl == "oracle" else U
    a, b = torch.from_numpy(leaf["lin_start"]), torch.from_numpy(leaf["lin_end"])
    assert torch.equal(mod.tensor_linspace(a, b, 7), torch.from_numpy(leaf["lin_out"]))
    for (t, e, az), ref in zip(leaf["rot_in"], leaf["rot_out"]):
        assert np.allclose(mod.cal_rotation_matrix(t, e, az, 5.0), ref, rtol=0, atol=1e-15)
    x = torch.from_numpy(leaf["norm_in"])
    assert torch.allclose(mod.normalize_features(torch.mean(x, dim=0)), torch.from_numpy(leaf["norm_out"]), rtol=1e-6, atol=0)


def test_flow_warp_matches_reference(leaf):
    got = R.flow_warp(torch.from_numpy(leaf["warp_im"]), torch.from_numpy(leaf["warp_flow"]))
    assert torch.allclose(got, torch.from_numpy(leaf["warp_out"]), rtol=1e-6, atol=1e-7)


# ---- known answers ----------------------------------------------------------------------------------
def test_spherical_round_trip():
    d = torch.tensor([3.0, 5.0, 7.5, 5.0])
    e = torch.tensor([-0.5, 0.3, 1.0, 0.0])
    a = torch.tensor([0.1, 2.0, 4.0, 6.0])
    C = R.camera_position_from_spherical_angles(d, e, a)
    d2, e2, a2 = R.camera_position_to_spherical_angle(C)
    assert torch.allclose(d2, d, atol=1e-5) and torch.allclose(e2, e, atol=1e-5) and torch.allclose(a2, a, atol=1e-5)
    from nemo_amd.utils import camera_position_from_spherical_angles as cp
    assert torch.equal(cp(d, e, a), C)


def test_look_at_is_orthonormal_and_looks_at_origin():
    campos, theta = random_poses(32, seed=5)
    Rm, T = R.R_T_tensors(campos, torch.zeros(32))
    eye = torch.eye(3).expand(32, 3, 3)
    assert torch.allclose(Rm.transpose(1, 2) @ Rm, eye, atol=1e-5)
    # the origin lies on the optical axis at depth |C| : X_view = 0 @ R + T = (0, 0, |C|)
    assert torch.allclose(T[:, :2], torch.zeros(32, 2), atol=1e-5)
    assert torch.allclose(T[:, 2], campos.norm(dim=1), atol=1e-5)
    # theta rotates about the optical axis: T is unchanged
    _, T2 = R.R_T_tensors(campos, theta)
    assert torch.allclose(T2, T, atol=1e-5)


def test_single_triangle_coverage():
    # a large triangle in front of a camera on the +z axis covers the image centre, not the corners
    cam = R.Camera(16, 20, 20.0)
    verts = torch.tensor([[-1.0, -1.0, 0.0], [1.0, -1.0, 0.0], [0.0, 1.5, 0.0]])
    faces = torch.tensor([[0, 1, 2]], dtype=torch.int32)
    C = torch.tensor([[0.0, 0.0, 4.0]])
    p2f, bary, zbuf = R.fragments(verts, faces, C, torch.zeros(1), cam)
    assert p2f[0, 8, 10] == 0 and p2f[0, 0, 0] == -1 and p2f[0, 15, 19] == -1
    hit = p2f[0] >= 0
    assert torch.allclose(bary[0][hit].sum(-1), torch.ones(int(hit.sum())), atol=1e-5)
    assert torch.allclose(zbuf[0][hit], torch.full((int(hit.sum()),), 4.0), atol=1e-5)
    assert float(bary[0][~hit].max()) == -1.0 and float(zbuf[0][~hit].max()) == -1.0
    # +X is left and +Y is up in NDC: the apex (world +y) is in the upper half of the image
    rows = hit.any(dim=1).nonzero().flatten()
    widths = hit.sum(dim=1)
    assert widths[rows[0]] < widths[rows[-1]]


def test_cuboid_front_view_is_symmetric():
    sc = Scene(40, 50, 8, 150, 375.0 / 2)
    C = R.camera_position_from_spherical_angles(5.0, 0.0, 0.0)
    p2f, _, _ = sc.ref.fragments(C, torch.zeros(1))
    hit = p2f[0] >= 0
    assert 0.02 < hit.float().mean() < 0.5
    cols = hit.any(dim=0).nonzero().flatten()
    assert abs((int(cols[0]) + int(cols[-1])) - (sc.W - 1)) <= 1      # centred on the principal point


def test_pose_error_known_values():
    g = dict(azimuth=0.3, elevation=0.2, theta=0.1)
    assert R.pose_error(g, g) < 1e-7
    assert abs(R.pose_error(g, dict(g, theta=0.1 + math.pi / 2)) - math.pi / 2) < 1e-6
    assert R.pose_error(g, None) == np.pi
    from nemo_amd.utils import pose_error
    assert abs(pose_error(g, dict(g, azimuth=0.5)) - R.pose_error(g, dict(g, azimuth=0.5))) < 1e-12


def test_oracle_gradient_against_finite_differences():
    sc = Scene(20, 25, 8, 80, 375.0 / 4)
    fm, _ = sc.feature_maps(1, first_seed=3)
    clu = R.clutter_score(fm, sc.clutter)
    campos, theta = random_poses(1, seed=2, dist=(4.9, 5.1))

    def loss_of(p):
        with torch.no_grad():
            return float(R.loss_fg_bg(torch.sum(sc.ref(p[:, :3], p[:, 3]) * fm, dim=1), clu))

    P = torch.cat([campos, theta[:, None]], dim=1).double()
    C = campos.clone().requires_grad_(True)
    t = theta.clone().requires_grad_(True)
    R.loss_fg_bg(torch.sum(sc.ref(C, t) * fm, dim=1), clu).backward()
    auto = torch.cat([C.grad[0], t.grad]).double()
    fd = torch.zeros(4, dtype=torch.float64)
    h = 2e-3
    for k in range(4):
        dp = torch.zeros_like(P)
        dp[0, k] = h
        fd[k] = (loss_of((P + dp).float()) - loss_of((P - dp).float())) / (2 * h)
    cos = float((auto * fd).sum() / (auto.norm() * fd.norm()))
    assert cos > 0.97, (auto, fd)


# ---- the frozen oracle outputs ----------------------------------------------------------------------
def test_oracle_reproduces_small_fixture():
    gold = np.load(os.path.join(GOLD, "oracle_small.npz"))
    sc = Scene(40, 50, 32, 150, 375.0 / 2)
    az, el, th, di, _, _ = R.get_param_samples(GRID_CFG)
    cs = torch.from_numpy(gold["grid_campos"])
    ths = torch.from_numpy(gold["grid_theta"])
    idx = torch.arange(0, 144, 9)                       # every ninth pose keeps the CPU suite short
    p2f, _, _ = R.fragments(sc.ref.verts, sc.ref.faces, cs[idx], ths[idx], sc.cam)
    assert np.array_equal(p2f.numpy().astype(np.int16), gold["grid_p2f"][idx.numpy()])
    C = R.camera_position_from_spherical_angles(di[0], el[1], az[3])
    assert torch.equal(C[0], cs[3 * 12 + 1 * 3 + 0])     # nesting azimuth -> elevation -> theta


# ---- product host logic -----------------------------------------------------------------------------
def test_meshelize_and_off_round_trip(tmp_path):
    from nemo_amd.utils import load_off, meshelize, save_off
    v, f = meshelize((-0.225, 0.225), (-0.5, 0.5), (-0.18, 0.18), 1000)
    assert v.shape == (1118, 3) and f.shape == (1904, 3)
    v_ref, f_ref = R.meshelize((-0.225, 0.225), (-0.5, 0.5), (-0.18, 0.18), 1000)
    assert np.array_equal(v, v_ref) and np.array_equal(f, f_ref)
    path = str(tmp_path / "01.off")
    save_off(path, v, f)
    v2, f2 = load_off(path)
    v3, f3 = R.parse_off(R.off_text(v, f))
    assert v2.dtype == np.float32 and f2.dtype == np.int32
    assert np.array_equal(v2, v3) and np.array_equal(f2, f3) and np.array_equal(f2, f)
    vt, ft = load_off(path, to_torch=True)
    assert torch.equal(vt, torch.from_numpy(v2)) and ft.dtype == torch.int32


def test_config_node_and_defaults(tmp_path):
    from nemo_amd.utils import ConfigNode, default_pose3d_config, load_config
    cfg = default_pose3d_config()
    assert cfg.inference.epochs == 300 and cfg.inference.optimizer.lr == 0.05
    assert cfg.get("batch_size_no_grad", 144) == 144 and cfg.inference.get("pre_render", True) is True
    (tmp_path / "model.yaml").write_text("down_sample_rate: 8\n")
    (tmp_path / "main.yaml").write_text("model: model.yaml\ninference:\n    epochs: 30\n")
    loaded = load_config(str(tmp_path / "main.yaml"), overrides={"inference.epochs": 60})
    assert loaded.model.down_sample_rate == 8 and loaded.inference.epochs == 60
    assert isinstance(loaded.inference, ConfigNode)


def test_camera_value_objects():
    import nemo_amd
    cams = nemo_amd.PerspectiveCameras(focal_length=375.0, principal_point=((50, 40),), image_size=((80, 100),), in_ndc=False)
    assert cams.image_size.tolist() == [[80.0, 100.0]] and cams._N == 1
    with pytest.raises(NotImplementedError):
        nemo_amd.PerspectiveCameras(focal_length=1.0, image_size=((8, 8),), in_ndc=True)
    rs = nemo_amd.RasterizationSettings(image_size=(80, 100), blur_radius=0.0, faces_per_pixel=1, bin_size=0)
    r = nemo_amd.MeshRasterizer(cameras=cams, raster_settings=rs)
    assert r.raster_settings.faces_per_pixel == 1
    with pytest.raises(NotImplementedError):
        nemo_amd.MeshInterpolateModule(torch.zeros(3, 3), torch.zeros(1, 3), torch.zeros(3, 4), rasterizer=object())


def test_product_path_has_no_cpu_fallback():
    import nemo_amd
    from nemo_amd import ops
    cams = nemo_amd.PerspectiveCameras(focal_length=10.0, principal_point=((5, 4),), image_size=((8, 10),), in_ndc=False)
    mod = nemo_amd.MeshInterpolateModule(torch.rand(4, 3), torch.tensor([[0, 1, 2]]), torch.rand(4, 8),
                                         rasterizer=nemo_amd.MeshRasterizer(cameras=cams))
    with pytest.raises(RuntimeError):
        mod(torch.tensor([[0.0, 0.0, 5.0]]), torch.zeros(1))           # not moved to a GPU
    with pytest.raises(RuntimeError):
        ops.clutter_score(torch.rand(1, 8, 4, 4), torch.rand(1, 8))     # CPU tensors are refused
    assert "oracle" not in "".join(open(os.path.join(ROOT, "nemo_amd", n)).read() for n in os.listdir(os.path.join(ROOT, "nemo_amd"))
                                   if n.endswith(".py")).replace("# oracle", "")


def test_solve_pose_rejects_unsupported_modes():
    import nemo_amd
    from nemo_amd.utils import default_pose3d_config
    cfg = default_pose3d_config()
    fm = torch.rand(1, 8, 4, 5)
    with pytest.raises(ValueError):
        nemo_amd.solve_pose(cfg, fm, None, torch.rand(1, 8), None, None, None, dof=5)
    cfg.inference.optimizer["class_name"] = "torch.optim.SGD"
    with pytest.raises(NotImplementedError):
        nemo_amd.solve_pose(cfg, fm, None, torch.rand(1, 8), None, None, None)


# ---- C ABI ---------------------------------------------------------------------------------------------
def test_c_abi_exports_every_declared_symbol():
    from nemo_amd import _lib
    header = open(os.path.join(ROOT, "include", "nemo_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(nemo_[a-z_]+)\s*\(", header)))
    assert declared == sorted(_lib.SYMBOLS), (declared, sorted(_lib.SYMBOLS))
    assert os.path.exists(_lib.LIB_PATH), "libnemo_hip.so is not built (run __graft_entry__.build())"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    lib.nemo_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.nemo_version()
    lib.nemo_grid_score_workspace.restype = ctypes.c_size_t
    lib.nemo_grid_score_workspace.argtypes = [ctypes.c_int32] * 3
    assert lib.nemo_grid_score_workspace(144, 64, 8000) == 32 * 64 * 144 * 4


# ---- multi-process gather (gloo, world size 2) ---------------------------------------------------------
_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from nemo_amd.distributed import shard_range, all_gather_rows
dist.init_process_group("gloo")
rank, ws = dist.get_rank(), dist.get_world_size()
B = 7
lo, hi = shard_range(B)
full = torch.arange(B * 7, dtype=torch.float32).reshape(B, 7)
got = all_gather_rows(full[lo:hi].clone(), n_total=B)
assert torch.equal(got, full), (rank, got)
assert [shard_range(B, r, ws) for r in range(ws)] == [(0, 4), (4, 7)]
dist.destroy_process_group()
print("ok", rank)
"""


def test_two_process_gather_gloo(tmp_path):
    import socket
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:     # a port nobody listens on, for the rendezvous
        sock.bind(("127.0.0.1", 0))
        port = str(sock.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", port, str(script), ROOT],
                         capture_output=True, text=True, timeout=240, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok") == 2


# ---------------------------------------------------------------------------------------------
# keypoint-initialised solver: host logic of the product against the oracle's restatement
# ---------------------------------------------------------------------------------------------
def test_peak_local_max_matches_scipy_restatement():
    from nemo_amd.solve_pose_keypoint import peak_local_max
    rng = np.random.RandomState(3)
    for shape in ((6, 6), (5, 9), (1, 7), (12, 12)):
        for md in (1, 2, 3):
            img = rng.rand(*shape).astype(np.float32)
            img[rng.randint(shape[0]), rng.randint(shape[1])] = 2.0        # one clear winner
            assert np.array_equal(peak_local_max(img, md), R.peak_local_max(img, md)), (shape, md)
    flat = np.ones((4, 4), dtype=np.float32)
    assert peak_local_max(flat, 1).shape == (0, 2) and R.peak_local_max(flat, 1).shape == (0, 2)
    # known answer: two separated bumps, the higher one first; a plateau neighbour closer than min_distance is dropped
    img = np.zeros((7, 7), dtype=np.float32)
    img[1, 1], img[5, 4], img[5, 5] = 3.0, 5.0, 5.0
    assert peak_local_max(img, 1).tolist() == [[5, 4], [5, 5], [1, 1]]
    assert peak_local_max(img, 2).tolist() == [[5, 4], [1, 1]]


def test_keypoint_projection_and_proposals_match_oracle():
    from nemo_amd.solve_pose_keypoint import _proposals, pre_compute_kp_coords
    from nemo_amd.utils import ConfigNode
    verts, _ = R.cuboid_mesh(number_vertices=120)
    az, el, th, di = np.linspace(0, 2 * np.pi, 5, endpoint=False), np.array([0.0, 0.4]), np.array([-0.2, 0.3]), np.array([4.0, 6.0])
    poses, kpc, kpv = pre_compute_kp_coords(verts, az, el, th, di, viewport=1500)
    poses_r, kpc_r, kpv_r = R.pre_compute_kp_coords(verts.astype(np.float64), az, el, th, di, viewport=1500)
    assert poses.shape == (5, 2, 2, 2, 4) and np.array_equal(poses, poses_r) and np.array_equal(kpv, kpv_r)
    assert np.allclose(kpc, kpc_r, rtol=0, atol=1e-3)
    # known answer: azimuth 0, elevation 0, theta 0: the camera sits on -y looking along +y, x to the right, z up
    p1, k1, _ = pre_compute_kp_coords(np.array([[0.1, 0.0, 0.0], [0.0, 0.0, 0.2]]), [0.0], [0.0], [0.0], [5.0], viewport=1000)
    assert np.allclose(k1[0], [[20.0, 0.0], [0.0, -40.0]], atol=1e-3)
    rng = np.random.RandomState(5)
    corr = rng.rand(poses.reshape(-1, 4).shape[0], 6, 5).astype(np.float32)
    px, py = np.linspace(0, 400, 6), np.linspace(0, 320, 5)
    for thr, search in ((0.97, True), (10.0, True), (0.5, False)):
        cfg = ConfigNode({"inference": {"search_translation": search, "min_distance": 1, "pre_rendering_thr": thr}})
        c = corr if search else corr[:, :1, :1]
        got, cmax = _proposals(cfg, c, poses, px if search else px[:1], py if search else py[:1])
        ref, cmax_r = R.keypoint_proposals(c, poses, px if search else px[:1], py if search else py[:1], search, 1, thr)
        assert np.array_equal(cmax, cmax_r) and len(got) == len(ref) >= 1
        for g, r in zip(got, ref):
            assert all(float(g[k]) == float(r[k]) for k in ("azimuth", "elevation", "theta", "distance", "px", "py"))


def test_table_layout_round_trip():
    from nemo_amd import ops

    class FakeMesh:
        V = 10
    dense = torch.arange(2 * 10 * 300, dtype=torch.float32).reshape(2, 10, 300)
    tab = ops.table_from_dense(FakeMesh, dense)
    assert tuple(tab.shape) == ops.table_shape(FakeMesh, 2, 300) == (2, 3, 3, 128, 4)
    assert torch.equal(ops.table_dense(tab, 10, 300), dense)
    assert float(tab.sum()) == float(dense.sum())            # padding is zero
    assert float(tab[1, 2, 1, 300 % 128 - 1, 1]) == float(dense[1, 5, 299])


def test_top_k_grid_poses_are_best_first_and_stable(monkeypatch):
    """Host logic of the multi-hypothesis initialisation: the k lowest grid losses per image, ties in grid
    order, column 0 = the arg-min pose (the device scoring is replaced by a fixed loss table)."""
    from nemo_amd import batch_solve_pose as bsp
    loss = torch.tensor([[0.5, 0.2, 0.9, 0.2, 0.1], [0.3, 0.3, 0.3, 0.8, 0.05]])
    monkeypatch.setattr(bsp.ops, "grid_score",
                        lambda pre, fm, clu, packed=None: (loss, loss.argmin(dim=1).int(), loss.min(dim=1)[0]))
    pos = torch.arange(15, dtype=torch.float32).reshape(5, 3)
    theta = torch.arange(5, dtype=torch.float32) * 0.1
    maps = torch.zeros(5, 1, 2, 2, 2)
    C1, t1, m1 = bsp.get_init_pos_rendered_dim0(maps, pos, theta, torch.zeros(2, 2, 2, 2))
    assert torch.equal(C1, pos[[4, 4]]) and torch.equal(t1, theta[[4, 4]]) and torch.equal(m1, torch.tensor([0.1, 0.05]))
    C3, t3, m3 = bsp.get_init_pos_rendered_dim0(maps, pos, theta, torch.zeros(2, 2, 2, 2), top_k=3)
    assert C3.shape == (2, 3, 3) and t3.shape == (2, 3)
    assert torch.equal(t3, theta[torch.tensor([[4, 1, 3], [4, 0, 1]])])
    assert torch.equal(C3[:, 0], C1) and torch.equal(m3[:, 0], m1)
    assert torch.equal(m3, torch.tensor([[0.1, 0.2, 0.2], [0.05, 0.3, 0.3]]))
    with pytest.raises(ValueError):
        bsp.get_init_pos_rendered_dim0(maps, pos, theta, torch.zeros(2, 2, 2, 2), top_k=6)


def test_resolve_occ_known_answers():
    """Two overlapping masks: the one whose (overlap + half the rest) score is larger keeps the overlap; equal
    claims go to the earlier proposal; disjoint masks are untouched.  These are the intended semantics, not the
    reference's output: solve_pose.py:350-402 ends by summing the kept masks into mask 0 in place (see resolve_occ's
    docstring), which this implementation deliberately does not reproduce."""
    from nemo_amd.solve_pose_keypoint import resolve_occ
    seg = np.zeros((3, 4, 6), dtype=np.uint8)
    seg[0, :, 0:3] = 1
    seg[1, :, 2:5] = 1          # overlaps proposal 0 in column 2
    seg[2, 0, 5] = 1            # disjoint
    score = np.ones((3, 4, 6), dtype=np.float32)
    score[1] *= 2.0             # proposal 1 claims more everywhere
    idx, kept = resolve_occ(seg, score)
    assert kept[0, :, 2].sum() == 0 and kept[1, :, 2].sum() == 4 and np.array_equal(kept[2], seg[2])
    assert np.array_equal(kept[0, :, :2], seg[0, :, :2])
    assert idx[0, 2] == 2 and idx[0, 0] == 1 and idx[0, 5] == 3 and idx[1, 5] == 0
    # equal claims: the earlier proposal wins the overlap
    idx2, kept2 = resolve_occ(seg[:2], np.ones((2, 4, 6), dtype=np.float32))
    assert kept2[0, :, 2].sum() == 4 and kept2[1, :, 2].sum() == 0
    # the inputs are not modified
    assert seg[0, :, 2].sum() == 4
