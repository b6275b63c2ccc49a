"""Thin tensor-level wrappers over the C ABI (one function per entry point of include/nemo_hip.h).

Each wrapper allocates its outputs with torch (device memory plumbing), passes raw pointers and the
current stream, and raises on a non-zero return code.  No arithmetic happens here.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import NemoAdam, NemoCamera, check, f32c, ptr, stream_ptr


class Camera:
    """Host-side value of the C struct nemo_camera."""

    def __init__(self, H, W, focal, px=None, py=None):
        self.H, self.W, self.focal = int(H), int(W), float(focal)
        self.px = float(W // 2) if px is None else float(px)
        self.py = float(H // 2) if py is None else float(py)

    def c(self):
        return NemoCamera(self.H, self.W, self.focal, self.px, self.py)


class MeshHandle:
    """Owner of a nemo_mesh_t (device copies of verts / faces / vertex memory)."""

    def __init__(self, verts, faces):
        lib = _lib.load()
        if not verts.is_cuda:
            raise RuntimeError("MeshHandle needs device tensors")
        self.device = verts.device
        v = f32c(verts)
        f = faces.detach().to(torch.int32).contiguous()
        if v.dim() != 2 or v.shape[1] != 3 or f.dim() != 2 or f.shape[1] != 3:
            raise ValueError("verts must be [V,3] and faces [F,3]")
        if f.numel() and (int(f.min()) < 0 or int(f.max()) >= v.shape[0]):
            raise ValueError("face index out of range")
        self.V, self.F, self.C = v.shape[0], f.shape[0], 0
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(lib.nemo_mesh_create(ptr(v), ptr(f, torch.int32), self.V, self.F, C.byref(h)), "nemo_mesh_create")
        self._h = h

    def set_memory(self, memory):
        lib = _lib.load()
        mem = f32c(memory.to(self.device))
        if mem.dim() != 2 or mem.shape[0] != self.V:
            raise ValueError(f"memory must be [V={self.V}, C], got {tuple(mem.shape)}")
        with torch.cuda.device(self.device):
            check(lib.nemo_mesh_set_memory(self._h, ptr(mem), mem.shape[1], stream_ptr()), "nemo_mesh_set_memory")
            torch.cuda.current_stream().synchronize()   # `mem` may be a temporary
        self.C = mem.shape[1]

    def set_parts(self, vert_counts, face_counts):
        """Names the meshes of a handle built from several meshes back to back (a multi-class bank): part l is the
        next vert_counts[l] vertices and face_counts[l] faces.  solve(..., hyp_to_part=...) then renders one part per
        hypothesis."""
        lib = _lib.load()
        vc, fc = [int(x) for x in vert_counts], [int(x) for x in face_counts]
        if len(vc) != len(fc) or not vc:
            raise ValueError("vert_counts and face_counts need one entry per mesh")
        arr = C.c_int32 * len(vc)
        with torch.cuda.device(self.device):
            check(lib.nemo_mesh_set_parts(self.h, arr(*vc), arr(*fc), len(vc)), "nemo_mesh_set_parts")
        self.part_verts, self.part_faces = vc, fc

    @property
    def h(self):
        if self._h is None:
            raise RuntimeError("mesh handle already destroyed")
        return self._h

    def close(self):
        if getattr(self, "_h", None) is not None:
            _lib.load().nemo_mesh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _image_index(hyp_to_image, n_img, device):
    """hyp_to_image as int32 on `device`.  A host tensor is range-checked here; a device tensor is not read
    back (no synchronisation in the solve path): the kernel gives a hypothesis whose index names no image a
    NaN result row instead."""
    if hyp_to_image is None:
        return None
    if not hyp_to_image.is_cuda and hyp_to_image.numel() and (int(hyp_to_image.min()) < 0 or int(hyp_to_image.max()) >= n_img):
        raise ValueError("hyp_to_image out of range")
    return hyp_to_image.to(device=device, dtype=torch.int32).contiguous()


def _rows(t, n, width):
    """t as contiguous f32 [n, -1]; with n == 0 (where -1 is ambiguous to torch) [0, width]."""
    t = f32c(t)
    return t.reshape(0, width) if n == 0 else t.reshape(n, -1).contiguous()


def _pose_args(campos, theta, principal):
    cp = f32c(campos).reshape(-1, 3)
    n = cp.shape[0]
    th = f32c(theta).reshape(-1)
    if th.numel() != n:
        raise ValueError(f"theta has {th.numel()} entries for {n} camera positions")
    pr = None
    if principal is not None:
        pr = f32c(principal).reshape(-1, 2)
        if pr.shape[0] == 1 and n > 1:
            pr = pr.expand(n, 2).contiguous()
        if pr.shape[0] != n:
            raise ValueError(f"principal has {pr.shape[0]} rows for {n} cameras")
    return cp, th, pr, n


def project(mesh, cam, campos, theta, principal=None):
    """-> (RT [n,12], proj [n,V,5]): R row-major + T, and view X,Y,Z + NDC x,y per vertex."""
    lib = _lib.load()
    cp, th, pr, n = _pose_args(campos, theta, principal)
    rt = torch.empty(n, 12, device=cp.device)
    proj = torch.empty(n, mesh.V, 5, device=cp.device)
    cc = cam.c()
    with torch.cuda.device(cp.device):
        check(lib.nemo_project(mesh.h, C.byref(cc), ptr(cp), ptr(th), ptr(pr), n, ptr(rt), ptr(proj), stream_ptr()),
              "nemo_project")
    return rt, proj


def project_keypoints(mesh, cam, campos, theta, principal=None, dist_thr=1e-3, restrict_to_boundary=True, R=None, T=None):
    """Training-side annotation: -> (kp [n,V,2] = (y, x) feature-map pixels, vis [n,V] bool).
    With R [n,3,3] (X_view = X @ R + T) and T [n,3] the caller's own extrinsics replace campos / theta."""
    lib = _lib.load()
    if (R is None) != (T is None):
        raise ValueError("R and T go together")
    if R is not None:
        rt = torch.cat([f32c(R).reshape(-1, 9), f32c(T).reshape(-1, 3)], dim=1).contiguous()
        n, dev, cp, th = rt.shape[0], rt.device, None, None
        pr = None
        if principal is not None:
            pr = f32c(principal).reshape(-1, 2)
            pr = pr.expand(n, 2).contiguous() if pr.shape[0] == 1 and n > 1 else pr
            if pr.shape[0] != n:
                raise ValueError(f"principal has {pr.shape[0]} rows for {n} cameras")
    else:
        cp, th, pr, n = _pose_args(campos, theta, principal)
        rt, dev = None, cp.device
    kp = torch.empty(n, mesh.V, 2, device=dev)
    vis = torch.empty(n, mesh.V, dtype=torch.uint8, device=dev)
    cc = cam.c()
    with torch.cuda.device(dev):
        check(lib.nemo_project_keypoints(mesh.h, C.byref(cc), ptr(cp), ptr(th), ptr(rt), ptr(pr), n, float(dist_thr),
                                         1 if restrict_to_boundary else 0, ptr(kp), ptr(vis, torch.uint8), stream_ptr()),
              "nemo_project_keypoints")
    return kp, vis.bool()


def render(mesh, cam, campos, theta, principal=None, maps=True, fragments=False, nearest=False):
    """-> dict(maps [n,C,H,W], pix_to_face [n,H,W] i32, bary [n,H,W,3], zbuf [n,H,W]) (requested keys).
    nearest: the maps take the feature of the nearest vertex of the covering face (mode='nearest') instead of the blend."""
    lib = _lib.load()
    cp, th, pr, n = _pose_args(campos, theta, principal)
    dev = cp.device
    out = {}
    if nearest:
        if fragments or not maps:
            raise ValueError("nearest=True renders maps only")
        out["maps"] = torch.empty(n, mesh.C, cam.H, cam.W, device=dev)
        cc = cam.c()
        with torch.cuda.device(dev):
            check(lib.nemo_render_nearest(mesh.h, C.byref(cc), ptr(cp), ptr(th), ptr(pr), n, ptr(out["maps"]), stream_ptr()),
                  "nemo_render_nearest")
        return out
    if maps:
        out["maps"] = torch.empty(n, mesh.C, cam.H, cam.W, device=dev)
    if fragments:
        out["pix_to_face"] = torch.empty(n, cam.H, cam.W, dtype=torch.int32, device=dev)
        out["bary"] = torch.empty(n, cam.H, cam.W, 3, device=dev)
        out["zbuf"] = torch.empty(n, cam.H, cam.W, device=dev)
    cc = cam.c()
    with torch.cuda.device(dev):
        check(lib.nemo_render(mesh.h, C.byref(cc), ptr(cp), ptr(th), ptr(pr), n, ptr(out.get("maps")),
                              ptr(out.get("pix_to_face"), torch.int32), ptr(out.get("bary")), ptr(out.get("zbuf")),
                              stream_ptr()), "nemo_render")
    return out


def render_backward(mesh, cam, campos, theta, principal, grad_maps, want_principal=False, nearest=False):
    """-> (grad_campos [n,3], grad_theta [n], grad_principal [n,2] or None).  nearest: the maps came from render(..., nearest=True)
    (mode='nearest'): the gradient passes through the softmax over the barycentrics (the reference's straight-through estimator)."""
    lib = _lib.load()
    cp, th, pr, n = _pose_args(campos, theta, principal)
    gm = f32c(grad_maps)
    if tuple(gm.shape) != (n, mesh.C, cam.H, cam.W):
        raise ValueError(f"grad_maps must be {(n, mesh.C, cam.H, cam.W)}, got {tuple(gm.shape)}")
    gC = torch.empty(n, 3, device=cp.device)
    gT = torch.empty(n, device=cp.device)
    gP = torch.empty(n, 2, device=cp.device) if want_principal else None
    cc = cam.c()
    with torch.cuda.device(cp.device):
        fn = lib.nemo_render_nearest_backward if nearest else lib.nemo_render_backward
        check(fn(mesh.h, C.byref(cc), ptr(cp), ptr(th), ptr(pr), n, ptr(gm), ptr(gC), ptr(gT), ptr(gP), stream_ptr()),
              "nemo_render_nearest_backward" if nearest else "nemo_render_backward")
    return gC, gT, gP


def raster_forms(mesh, cam, campos, theta, form, reps=20, keys=True):
    """Diagnostic: the optimisation loop's rasterise phase, form 0 (face per lane, the solver's) or 1 (tile-binned), timed
    inside the kernel.  -> (cycles per rasterisation [n], cycles of the binning part [n] (form 1), key buffers [n,H,W] int64 or None)."""
    lib = _lib.load()
    cp, th, _, n = _pose_args(campos, theta, None)
    cyc = torch.empty(2 * n, device=cp.device)
    kb = torch.empty(n, cam.H, cam.W, dtype=torch.int32, device=cp.device) if keys else None
    cc = cam.c()
    with torch.cuda.device(cp.device):
        check(lib.nemo_raster_forms(mesh.h, C.byref(cc), ptr(cp), ptr(th), n, int(form), int(reps), ptr(cyc), ptr(kb, torch.int32),
                                    stream_ptr()), "nemo_raster_forms")
    return cyc[:n], cyc[n:], (None if kb is None else kb.long() & 0xFFFFFFFF)


def clutter_score(feature_map, clutter):
    """feature_map [B,C,H,W], clutter [K,C] -> [B,H,W] (max over the K clutter vectors)."""
    lib = _lib.load()
    fm = f32c(feature_map)
    cl = f32c(clutter).reshape(-1, fm.shape[1])
    B, Cc, H, W = fm.shape
    out = torch.empty(B, H, W, device=fm.device)
    with torch.cuda.device(fm.device):
        check(lib.nemo_clutter_score(ptr(fm), ptr(cl), B, Cc, H * W, cl.shape[0], ptr(out), stream_ptr()),
              "nemo_clutter_score")
    return out


class PackedGrid:
    """A pre-rendered pose grid in the matrix-core scoring layout (opaque device buffer + its shape)."""

    def __init__(self, data, shape):
        self.data, self.shape = data, tuple(shape)


def grid_pack(pre_rendered):
    """pre_rendered [N,C,H,W] -> PackedGrid for grid_score(..., packed=...), or None when the channel count
    is not one the matrix-core form takes (16, 32, 64, 128).  Do it once, next to the pre-rendering."""
    lib = _lib.load()
    pre = f32c(pre_rendered)
    N, Cc, H, W = pre.shape
    nbytes = lib.nemo_grid_pack_bytes(N, Cc, H * W)
    if nbytes == 0:
        return None
    data = torch.empty(nbytes, dtype=torch.uint8, device=pre.device)
    with torch.cuda.device(pre.device):
        check(lib.nemo_grid_pack(ptr(pre), N, Cc, H * W, ptr(data, torch.uint8), stream_ptr()), "nemo_grid_pack")
    return PackedGrid(data, pre.shape)


def grid_score(pre_rendered, feature_map, clutter_score_map=None, packed=None):
    """pre_rendered [N,C,H,W], feature_map [B,C,H,W], clutter_score_map [B,H,W] or None
    -> (loss [B,N], argmin [B] int32, min_loss [B]).  With packed (grid_pack of the same grid) the scoring
    runs on the matrix cores and pre_rendered is only consulted for its shape (it may be None)."""
    lib = _lib.load()
    fm = f32c(feature_map)
    B = fm.shape[0]
    if packed is not None:
        N, Cc, H, W = packed.shape
        if tuple(fm.shape[1:]) != (Cc, H, W):
            raise ValueError(f"feature_map {tuple(fm.shape)} does not match the packed grid {packed.shape}")
        clu = None if clutter_score_map is None else f32c(clutter_score_map).reshape(B, H * W)
        loss = torch.empty(B, N, device=fm.device)
        amin = torch.empty(B, dtype=torch.int32, device=fm.device)
        mn = torch.empty(B, device=fm.device)
        nbytes = lib.nemo_grid_score_packed_workspace(N, B, Cc, H * W)
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=fm.device)
        with torch.cuda.device(fm.device):
            check(lib.nemo_grid_score_packed(ptr(packed.data, torch.uint8), ptr(fm), ptr(clu), N, B, Cc, H * W, ptr(loss),
                                             ptr(amin, torch.int32), ptr(mn), ptr(ws, torch.uint8), nbytes, stream_ptr()),
                  "nemo_grid_score_packed")
        return loss, amin, mn
    if getattr(pre_rendered, "_nemo_shape_only", False):
        raise ValueError("these pre-rendered maps were dropped after packing (a shape-only stand-in): pass packed=")
    pre = f32c(pre_rendered)
    N, Cc, H, W = pre.shape
    if tuple(fm.shape[1:]) != (Cc, H, W):
        raise ValueError(f"feature_map {tuple(fm.shape)} does not match pre-rendered maps {tuple(pre.shape)}")
    clu = None if clutter_score_map is None else f32c(clutter_score_map).reshape(B, H * W)
    loss = torch.empty(B, N, device=fm.device)
    amin = torch.empty(B, dtype=torch.int32, device=fm.device)
    mn = torch.empty(B, device=fm.device)
    nbytes = lib.nemo_grid_score_workspace(N, B, H * W)
    ws = torch.empty(max(nbytes, 4) // 4, device=fm.device)
    with torch.cuda.device(fm.device):
        check(lib.nemo_grid_score(ptr(pre), ptr(fm), ptr(clu), N, B, Cc, H * W, ptr(loss), ptr(amin, torch.int32),
                                  ptr(mn), ptr(ws), nbytes, stream_ptr()), "nemo_grid_score")
    return loss, amin, mn


def clutter_and_grid_score(feature_map, clutter, packed, workspace=None):
    """clutter_score followed by grid_score(..., packed=packed) from one read of the feature maps.
    feature_map [B,C,H,W], clutter [K,C], packed = grid_pack of the pose grid
    -> (clutter_score [B,H,W], loss [B,N], argmin [B] int32, min_loss [B]), each bit for bit what the two calls give.
    workspace: optional uint8 tensor of grid_score_workspace_bytes(packed, B) bytes kept by the caller."""
    lib = _lib.load()
    fm = f32c(feature_map)
    B = fm.shape[0]
    N, Cc, H, W = packed.shape
    if tuple(fm.shape[1:]) != (Cc, H, W):
        raise ValueError(f"feature_map {tuple(fm.shape)} does not match the packed grid {packed.shape}")
    cl = f32c(clutter).reshape(-1, Cc)
    clu = torch.empty(B, H, W, device=fm.device)
    loss = torch.empty(B, N, device=fm.device)
    amin = torch.empty(B, dtype=torch.int32, device=fm.device)
    mn = torch.empty(B, device=fm.device)
    nbytes = lib.nemo_grid_score_packed_workspace(N, B, Cc, H * W)
    ws = workspace if workspace is not None else torch.empty(max(nbytes, 1), dtype=torch.uint8, device=fm.device)
    if ws.dtype != torch.uint8 or ws.numel() < nbytes or ws.device != fm.device:
        raise ValueError(f"workspace must be a uint8 tensor of at least {nbytes} bytes on {fm.device}")
    with torch.cuda.device(fm.device):
        check(lib.nemo_clutter_grid_score_packed(ptr(packed.data, torch.uint8), ptr(fm), ptr(cl), cl.shape[0], N, B, Cc, H * W, ptr(clu),
                                                 ptr(loss), ptr(amin, torch.int32), ptr(mn), ptr(ws, torch.uint8), ws.numel(),
                                                 stream_ptr()), "nemo_clutter_grid_score_packed")
    return clu, loss, amin, mn


def grid_score_on_fragments(packed, fragments_workspace, B, clutter_score_map=None):
    """grid_score(..., packed=packed) for a batch whose operand fragments an earlier grid_score / clutter_and_grid_score
    call with `workspace=fragments_workspace` has left in that workspace (same B, channels and map size): the maps are not
    read again.  Bit for bit the result of grid_score.  -> (loss [B,N], argmin [B] int32, min_loss [B])."""
    lib = _lib.load()
    N, Cc, H, W = packed.shape
    need = int(lib.nemo_grid_fragments_bytes(B, Cc, H * W))
    if fragments_workspace.dtype != torch.uint8 or fragments_workspace.numel() < need or not fragments_workspace.is_contiguous():
        raise ValueError(f"fragments_workspace must be the uint8 workspace (at least {need} bytes) of a grid scoring of the same batch")
    dev = fragments_workspace.device
    clu = None if clutter_score_map is None else f32c(clutter_score_map).reshape(B, H * W)
    loss = torch.empty(B, N, device=dev)
    amin = torch.empty(B, dtype=torch.int32, device=dev)
    mn = torch.empty(B, device=dev)
    nbytes = int(lib.nemo_grid_score_fragments_workspace(N, B, H * W))
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(lib.nemo_grid_score_fragments(ptr(packed.data, torch.uint8), ptr(fragments_workspace, torch.uint8), ptr(clu), N, B, Cc, H * W,
                                            ptr(loss), ptr(amin, torch.int32), ptr(mn), ptr(ws, torch.uint8), nbytes, stream_ptr()),
              "nemo_grid_score_fragments")
    return loss, amin, mn


def grid_score_workspace_bytes(packed, B):
    """Bytes of the workspace grid_score / clutter_and_grid_score need for B images against a packed grid."""
    N, Cc, H, W = packed.shape
    return int(_lib.load().nemo_grid_score_packed_workspace(N, B, Cc, H * W))


def grid_score_static(pre_rendered, feature_map, clutter_score_map, out_loss, out_argmin, out_min):
    """grid_score through the entry point that allocates nothing: the caller brings every tensor (inputs float32
    contiguous on the device, out_loss [B,N] f32, out_argmin [B] i32, out_min [B] f32) and the library keeps the
    partial sums in its own fixed device array.  Calls that overlap on one GPU race on that array."""
    lib = _lib.load()
    N, Cc, H, W = pre_rendered.shape
    B = feature_map.shape[0]
    if tuple(feature_map.shape[1:]) != (Cc, H, W) or tuple(out_loss.shape) != (B, N) or out_argmin.numel() != B or out_min.numel() != B:
        raise ValueError("shapes of the grid, the feature maps and the outputs do not match")
    if clutter_score_map is not None and clutter_score_map.numel() != B * H * W:
        raise ValueError("clutter_score_map must be [B,H,W]")
    with torch.cuda.device(feature_map.device):
        check(lib.nemo_grid_score_static(ptr(pre_rendered), ptr(feature_map), ptr(clutter_score_map), N, B, Cc, H * W, ptr(out_loss),
                                         ptr(out_argmin, torch.int32), ptr(out_min), stream_ptr()), "nemo_grid_score_static")
    return out_loss, out_argmin, out_min


def grid_mfma_channels(c):
    """Channel counts the matrix-core grid scoring (and with it the re-centred scoring) takes."""
    return c in (16, 32, 64, 128)


def grid_score_recentred(packed, feature_map, clutter_score_map, dist_src, dist_tgt, p_src, p_tgt, padding_mode="border"):
    """Scores re-centred copies of a batch against a packed grid without materialising them.
    feature_map [B,C,h1,w1], clutter_score_map [B,h1,w1] or None; dist_src / dist_tgt [nv], p_src / p_tgt [nv,2]
    (x, y): copy bv is image bv % B re-centred like align_maps.  -> (loss [nv,N], argmin [nv] i32, min [nv])."""
    lib = _lib.load()
    fm = f32c(feature_map)
    B, Cc, h1, w1 = fm.shape
    N, Cg, h0, w0 = packed.shape
    if Cg != Cc:
        raise ValueError(f"feature_map has {Cc} channels, the packed grid {Cg}")
    if padding_mode not in ("zeros", "border"):
        raise NotImplementedError("padding_mode must be 'zeros' or 'border'")
    nv = int(dist_src.numel())
    args = [_rows(t.to(fm.device), nv, w) for t, w in ((dist_src, 1), (dist_tgt, 1), (p_src, 2), (p_tgt, 2))]
    if args[2].shape[1] != 2 or args[3].shape[1] != 2:
        raise ValueError("p_src / p_tgt must be [nv, 2]")
    clu = None if clutter_score_map is None else f32c(clutter_score_map).reshape(B, h1 * w1)
    loss = torch.empty(nv, N, device=fm.device)
    amin = torch.empty(nv, dtype=torch.int32, device=fm.device)
    mn = torch.empty(nv, device=fm.device)
    if not grid_mfma_channels(Cc):
        raise NotImplementedError(f"grid_score_recentred takes 16, 32, 64 or 128 channels, not {Cc}")
    if nv == 0:
        return loss, amin, mn
    nbytes = lib.nemo_grid_score_recentred_workspace(N, nv, Cc, h0, w0)
    if nbytes == 0:
        raise NotImplementedError(f"grid_score_recentred takes 16, 32, 64 or 128 channels, not {Cc}")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=fm.device)
    with torch.cuda.device(fm.device):
        check(lib.nemo_grid_score_recentred(ptr(packed.data, torch.uint8), ptr(fm), ptr(clu), ptr(args[0]), ptr(args[1]), ptr(args[2]),
                                            ptr(args[3]), N, B, nv, Cc, h1, w1, h0, w0, 1 if padding_mode == "border" else 0,
                                            ptr(loss), ptr(amin, torch.int32), ptr(mn), ptr(ws, torch.uint8), nbytes, stream_ptr()),
              "nemo_grid_score_recentred")
    return loss, amin, mn


TABLE_TILE = 128


def table_shape(mesh, B, HW):
    """Shape of the similarity table of B images: [B, ceil(HW/128), ceil(V/4), 128, 4]
    (pixel tile, vertex group, pixel in tile, vertex in group)."""
    return (B, (HW + TABLE_TILE - 1) // TABLE_TILE, (mesh.V + 3) // 4, TABLE_TILE, 4)


def table_dense(table, V, HW):
    """The table as [B, V, HW] (a copy; for inspection and tests)."""
    B, npt, vg, T, four = table.shape
    return table.permute(0, 2, 4, 1, 3).reshape(B, vg * four, npt * T)[:, :V, :HW].contiguous()


def table_from_dense(mesh, dense):
    """[B, V, HW] -> the table layout (the inverse of table_dense; padding entries zero)."""
    B, V, HW = dense.shape
    if V != mesh.V:
        raise ValueError("dense score map does not match the mesh")
    _, npt, vg, T, four = table_shape(mesh, B, HW)
    full = torch.zeros(B, vg * four, npt * T, device=dense.device, dtype=torch.float32)
    full[:, :V, :HW] = dense
    return full.view(B, vg, four, npt, T).permute(0, 3, 1, 4, 2).contiguous()


def similarity_table(mesh, feature_map, out=None):
    """feature_map [B,C,H,W] -> table [B, ceil(HW/128), ceil(V/4), 128, 4]
    (element (b,v,p) at [b, p // 128, v // 4, p % 128, v % 4]; padding entries are zero)."""
    lib = _lib.load()
    fm = f32c(feature_map)
    B, Cc, H, W = fm.shape
    if Cc != mesh.C:
        raise ValueError(f"feature map has {Cc} channels, vertex memory has {mesh.C}")
    if out is None:
        out = torch.empty(table_shape(mesh, B, H * W), device=fm.device)
    elif tuple(out.shape) != table_shape(mesh, B, H * W) or not out.is_contiguous() or out.dtype != torch.float32:
        raise ValueError("out must be a contiguous float32 tensor of shape table_shape(mesh, B, H * W)")
    with torch.cuda.device(fm.device):
        check(lib.nemo_similarity_table(mesh.h, ptr(fm), B, H * W, ptr(out), stream_ptr()), "nemo_similarity_table")
    return out


def vertex_similarity(mesh, table, clu, H, W, boxes=None):
    """table from similarity_table, clu [B,H,W], boxes [B,4] int32 (r0,r1,c0,c1) or None -> [B]."""
    lib = _lib.load()
    B = table.shape[0]
    if tuple(table.shape) != table_shape(mesh, B, H * W) or not table.is_contiguous():
        raise ValueError("table does not match the mesh and map size")
    out = torch.empty(B, device=table.device)
    bx = None if boxes is None else boxes.to(device=table.device, dtype=torch.int32).contiguous()
    with torch.cuda.device(table.device):
        check(lib.nemo_vertex_similarity(mesh.h, ptr(table), ptr(f32c(clu)), ptr(bx, torch.int32), B, H, W, ptr(out),
                                         stream_ptr()), "nemo_vertex_similarity")
    return out


def vertex_similarity_from_features(mesh, feature_map, clu, boxes=None):
    """vertex_similarity without the table: feature_map [B,C,H,W], clu [B,H,W], boxes [B,4] int32 (r0,r1,c0,c1) or None
    -> [B].  The vertex x pixel products are reduced to the per-pixel maximum inside the kernel; channel counts the
    matrix-core form does not take (not 16 / 32 / 64 / 128) go through similarity_table + vertex_similarity."""
    lib = _lib.load()
    fm = f32c(feature_map)
    B, Cc, H, W = fm.shape
    if Cc != mesh.C:
        raise ValueError("feature maps do not match the vertex memory")
    if Cc not in (16, 32, 64, 128):
        return vertex_similarity(mesh, similarity_table(mesh, fm), clu, H, W, boxes)
    cl = f32c(clu).reshape(B, H * W)
    out = torch.empty(B, device=fm.device)
    bx = None if boxes is None else boxes.to(device=fm.device, dtype=torch.int32).contiguous()
    nbytes = int(lib.nemo_vertex_similarity_workspace(B, H * W))
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=fm.device)
    with torch.cuda.device(fm.device):
        check(lib.nemo_vertex_similarity_features(mesh.h, ptr(fm), ptr(cl), ptr(bx, torch.int32), B, H, W, ptr(out), ptr(ws, torch.uint8),
                                                  nbytes, stream_ptr()), "nemo_vertex_similarity_features")
    return out


def align_maps(maps_source, distance_source, principal_source, target_shape, distance_target, principal_target,
               padding_mode="zeros"):
    """align_no_centered: maps_source [n,c,h1,w1] -> [n,c,h0,w0]; distances [n], principals [n,2] (x,y)."""
    lib = _lib.load()
    src = f32c(maps_source)
    n, c, h1, w1 = src.shape
    h0, w0 = int(target_shape[0]), int(target_shape[1])
    if padding_mode not in ("zeros", "border"):
        raise NotImplementedError("padding_mode must be 'zeros' or 'border'")
    args = [_rows(t.to(src.device), n, w) for t, w in ((distance_source, 1), (distance_target, 1), (principal_source, 2),
                                                       (principal_target, 2))]
    out = torch.empty(n, c, h0, w0, device=src.device)
    with torch.cuda.device(src.device):
        check(lib.nemo_align_maps(ptr(src), ptr(args[0]), ptr(args[1]), ptr(args[2]), ptr(args[3]), n, c, h1, w1, h0, w0,
                                  1 if padding_mode == "border" else 0, ptr(out), stream_ptr()), "nemo_align_maps")
    return out


def solve(mesh, cam, table, clu, init_campos, init_theta, init_principal=None, hyp_to_image=None, global_batch=None,
          dof=3, steps=300, lr=0.05, betas=(0.4, 0.6), eps=1e-8, lr_gamma=0.2, lr_every=None, history=False,
          score_boxes=None, hyp_to_part=None):
    """Runs the whole Adam loop on device.  -> (pose [n_hyp,8], loss_hist [n_hyp,steps] or None).
    pose columns: cx, cy, cz, theta, px, py, score (last forward), first-forward loss.
    hyp_to_part [n_hyp] (a handle with set_parts): the mesh every hypothesis renders; None = the whole handle."""
    lib = _lib.load()
    cp, th, pr, n_hyp = _pose_args(init_campos, init_theta, init_principal)
    n_img = table.shape[0]
    if hyp_to_image is None and n_hyp != n_img:
        raise ValueError("hyp_to_image is required when the number of hypotheses differs from the number of images")
    h2i = _image_index(hyp_to_image, n_img, cp.device)
    clu2 = _rows(clu, n_img, cam.H * cam.W)
    if tuple(table.shape) != table_shape(mesh, n_img, cam.H * cam.W) or not table.is_contiguous() or clu2.shape[1] != cam.H * cam.W:
        raise ValueError("table / clutter shapes do not match the mesh and camera")
    opt = NemoAdam(int(steps), int(lr_every if lr_every is not None else max(steps // 3, 0)), float(lr), float(betas[0]),
                   float(betas[1]), float(eps), float(lr_gamma))
    out = torch.empty(n_hyp, 8, device=cp.device)
    hist = torch.empty(n_hyp, steps, device=cp.device) if history else None
    cc = cam.c()
    gb = int(global_batch if global_batch is not None else n_img)
    boxes = None
    if score_boxes is not None:
        boxes = score_boxes.to(device=cp.device, dtype=torch.int32).reshape(n_hyp, 4).contiguous()
    with torch.cuda.device(cp.device):
        if hyp_to_part is not None:
            h2p = hyp_to_part.to(device=cp.device, dtype=torch.int32).contiguous()
            if h2p.numel() != n_hyp:
                raise ValueError("hyp_to_part needs one entry per hypothesis")
            check(lib.nemo_solve_parts(mesh.h, C.byref(cc), ptr(table), ptr(clu2), ptr(h2i, torch.int32), ptr(h2p, torch.int32),
                                       ptr(cp), ptr(th), ptr(pr), ptr(boxes, torch.int32), n_hyp, n_img, gb, int(dof), C.byref(opt),
                                       ptr(out), ptr(hist), stream_ptr()), "nemo_solve_parts")
        else:
            check(lib.nemo_solve(mesh.h, C.byref(cc), ptr(table), ptr(clu2), ptr(h2i, torch.int32), ptr(cp), ptr(th), ptr(pr),
                                 ptr(boxes, torch.int32), n_hyp, n_img, gb, int(dof), C.byref(opt), ptr(out), ptr(hist),
                                 stream_ptr()), "nemo_solve")
    return out, hist


def solve_from_features(mesh, cam, feature_map, clu, init_campos, init_theta, init_principal=None, hyp_to_image=None,
                        global_batch=None, dof=3, steps=300, lr=0.05, betas=(0.4, 0.6), eps=1e-8, lr_gamma=0.2, lr_every=None,
                        history=False, score_boxes=None, hyp_to_part=None, workspace=None, bulk_first_step=True):
    """solve() without a precomputed table: the table tiles the covered pixels need are computed when they first need
    them (bit for bit the result of similarity_table + solve).  feature_map [n_img,C,H,W]; workspace: an optional uint8
    tensor of solve_workspace_bytes(mesh, n_img, H * W) to reuse (it need not be initialised).  bulk_first_step: the
    tiles of the first step (and of a small ring of pixels around it) are listed by a launch that only looks and
    computed by one kernel in front of the loop, when the workspace has room for the lists (it has for up to two
    hypotheses per image); False: the loop computes every tile itself (same results).
    -> (pose [n_hyp,8], loss_hist [n_hyp,steps] or None)."""
    lib = _lib.load()
    fm = f32c(feature_map)
    n_img, Cc, H, W = fm.shape
    if Cc != mesh.C or (H, W) != (cam.H, cam.W):
        raise ValueError("feature maps do not match the vertex memory / camera")
    cp, th, pr, n_hyp = _pose_args(init_campos, init_theta, init_principal)
    if hyp_to_image is None and n_hyp != n_img:
        raise ValueError("hyp_to_image is required when the number of hypotheses differs from the number of images")
    h2i = _image_index(hyp_to_image, n_img, cp.device)
    clu2 = _rows(clu, n_img, H * W)
    if clu2.shape[1] != H * W:
        raise ValueError("clutter score map does not match the camera")
    nbytes = solve_workspace_bytes(mesh, n_img, H * W)
    if nbytes == 0:
        raise RuntimeError(f"nemo_solve_features takes 16, 32, 64 or 128 channels, not {mesh.C}: use similarity_table + solve")
    if workspace is None:
        workspace = torch.empty(nbytes, dtype=torch.uint8, device=fm.device)
    elif workspace.dtype != torch.uint8 or workspace.numel() < nbytes or not workspace.is_contiguous():
        raise ValueError(f"workspace must be a contiguous uint8 tensor of at least {nbytes} bytes")
    opt = NemoAdam(int(steps), int(lr_every if lr_every is not None else max(steps // 3, 0)), float(lr), float(betas[0]),
                   float(betas[1]), float(eps), float(lr_gamma))
    out = torch.empty(n_hyp, 8, device=cp.device)
    hist = torch.empty(n_hyp, steps, device=cp.device) if history else None
    cc = cam.c()
    gb = int(global_batch if global_batch is not None else n_img)
    boxes = None if score_boxes is None else score_boxes.to(device=cp.device, dtype=torch.int32).reshape(n_hyp, 4).contiguous()
    h2p = None
    if hyp_to_part is not None:
        h2p = hyp_to_part.to(device=cp.device, dtype=torch.int32).contiguous()
        if h2p.numel() != n_hyp:
            raise ValueError("hyp_to_part needs one entry per hypothesis")
    with torch.cuda.device(cp.device):
        ws_bytes = workspace.numel()
        if not bulk_first_step:         # the library takes the lists' room from what lies behind the tables
            n_tab = 1
            for d_ in table_shape(mesh, n_img, H * W):
                n_tab *= d_
            ws_bytes = (4 * n_tab + 255) // 256 * 256
        check(lib.nemo_solve_features(mesh.h, C.byref(cc), ptr(fm), ptr(workspace, torch.uint8), ws_bytes, ptr(clu2), ptr(h2i, torch.int32),
                                      ptr(h2p, torch.int32), ptr(cp), ptr(th), ptr(pr), ptr(boxes, torch.int32), n_hyp, n_img, gb,
                                      int(dof), C.byref(opt), ptr(out), ptr(hist), stream_ptr()), "nemo_solve_features")
    return out, hist


def solve_workspace_bytes(mesh, n_img, HW):
    """Bytes of workspace solve_from_features needs for n_img images (0: channel count not supported)."""
    return int(_lib.load().nemo_solve_features_workspace(mesh.h, int(n_img), int(HW)))


def workspace_tables(mesh, workspace, n_img, HW):
    """The table part of a solve_from_features workspace as a float32 view of table_shape (for inspection and tests)."""
    shape = table_shape(mesh, n_img, HW)
    n = 1
    for d in shape:
        n *= d
    return workspace[:4 * n].view(torch.float32).view(shape)


def workspace_dense(mesh, workspace, n_img, HW):
    """The tables of a solve_from_features workspace as [n_img, V, HW] (a copy; entries of tiles that were never
    computed hold whatever the workspace held)."""
    return table_dense(workspace_tables(mesh, workspace, n_img, HW), mesh.V, HW)


def on_demand_table_supported(mesh, cam=None, with_parts=False):
    """True when solve_from_features takes this mesh: channel count 16, 32, 64 or 128 and, given the camera (map size),
    tile presence maps that fit the workgroup's LDS (meshes of a few thousand vertices per part); otherwise
    similarity_table + solve do the same job."""
    if mesh.C not in (16, 32, 64, 128):
        return False
    if cam is None:
        return True
    return bool(_lib.load().nemo_solve_features_supported(mesh.h, int(cam.H), int(cam.W), 1 if with_parts else 0))


def keypoint_corr(mesh, table, kp_coords, kp_vis, px_samples, py_samples, down_sample_rate, H, W):
    """get_corr_pytorch on the table of ONE image.  kp_coords [N,V,2] f32, kp_vis [N,V] f32, px/py samples f64
    -> corr [N, n_px, n_py] f32."""
    lib = _lib.load()
    if tuple(table.shape) != table_shape(mesh, 1, H * W) or not table.is_contiguous():
        raise ValueError("keypoint_corr takes the similarity table of exactly one image")
    dev = table.device
    kp = f32c(kp_coords.to(dev))
    vis = f32c(kp_vis.to(dev))
    N = kp.shape[0]
    if tuple(kp.shape) != (N, mesh.V, 2) or tuple(vis.shape) != (N, mesh.V):
        raise ValueError("kp_coords must be [N,V,2] and kp_vis [N,V]")
    px = torch.as_tensor(px_samples, dtype=torch.float64).to(dev).contiguous()
    py = torch.as_tensor(py_samples, dtype=torch.float64).to(dev).contiguous()
    out = torch.empty(N, px.numel(), py.numel(), device=dev)
    with torch.cuda.device(dev):
        check(lib.nemo_keypoint_corr(mesh.h, ptr(table), ptr(kp), ptr(vis), ptr(px, torch.float64), ptr(py, torch.float64), N,
                                     px.numel(), py.numel(), float(down_sample_rate), H, W, ptr(out), stream_ptr()),
              "nemo_keypoint_corr")
    return out


def solve_flow(mesh, cam, table, clu, init_campos, init_theta, init_flow=None, hyp_to_image=None, global_batch=None,
               optimize_translation=True, translation_scale=0.1, steps=300, lr=0.05, betas=(0.4, 0.6), eps=1e-8,
               lr_gamma=1.0, lr_every=0, history=False):
    """The keypoint-initialised solver's Adam loop (translation = 2-D flow warp of the rendered map).
    -> (pose [n_hyp,8]: cx, cy, cz, theta, flow_x, flow_y, score, first loss; loss_hist or None)."""
    lib = _lib.load()
    cp, th, fl, n_hyp = _pose_args(init_campos, init_theta, init_flow)
    n_img = table.shape[0]
    if hyp_to_image is None and n_hyp != n_img:
        raise ValueError("hyp_to_image is required when the number of hypotheses differs from the number of images")
    h2i = _image_index(hyp_to_image, n_img, cp.device)
    clu2 = _rows(clu, n_img, cam.H * cam.W)
    if tuple(table.shape) != table_shape(mesh, n_img, cam.H * cam.W) or not table.is_contiguous() or clu2.shape[1] != cam.H * cam.W:
        raise ValueError("table / clutter shapes do not match the mesh and camera")
    opt = NemoAdam(int(steps), int(lr_every), float(lr), float(betas[0]), float(betas[1]), float(eps), float(lr_gamma))
    out = torch.empty(n_hyp, 8, device=cp.device)
    hist = torch.empty(n_hyp, steps, device=cp.device) if history else None
    cc = cam.c()
    gb = int(global_batch if global_batch is not None else n_img)
    with torch.cuda.device(cp.device):
        check(lib.nemo_solve_flow(mesh.h, C.byref(cc), ptr(table), ptr(clu2), ptr(h2i, torch.int32), ptr(cp), ptr(th), ptr(fl),
                                  n_hyp, n_img, gb, int(bool(optimize_translation)), float(translation_scale), C.byref(opt),
                                  ptr(out), ptr(hist), stream_ptr()), "nemo_solve_flow")
    return out, hist


def spherical_angles(campos):
    """campos [n,3] -> [n,3] = distance, elevation, azimuth."""
    lib = _lib.load()
    cp = f32c(campos).reshape(-1, 3)
    out = torch.empty_like(cp)
    with torch.cuda.device(cp.device):
        check(lib.nemo_spherical_angles(ptr(cp), cp.shape[0], ptr(out), stream_ptr()), "nemo_spherical_angles")
    return out
