"""MeshInterpolateModule - the neural-mesh renderer module, on HIP.

Drop-in for reference nemo/models/mesh_interpolate_module.py:32-38,92-217 (the mesh variant):
same constructor and forward signature, same output ([n, C, H, W] interpolated vertex features,
zeros on background).  The mesh and the vertex feature memory live in a nemo_mesh_t handle; forward
is one launch of the render kernel (pose -> R,T -> projection -> nearest-face rasterisation ->
barycentric interpolation) and is differentiable w.r.t. campos / theta / principal point through
nemo_render_backward, which is what autograd through PyTorch3D provides in the reference.

Several meshes at once (lists of vertices / faces / memory banks, reference :110-117, :133-143, :182-191):
camera i renders mesh i (or the one camera renders every mesh); each mesh has its own handle and launch.

Not supported (raise): the VoGE variant (needs the un-vendored VoGE package), blur_radius > 0,
faces_per_pixel > 1, off_set_mesh / deform_verts.
"""
import torch
import torch.nn as nn

from . import ops
from .cameras import MeshRasterizer
from .utils.mesh import pre_process_mesh_pascal


def MeshInterpolateModule(*args, **kwargs):
    rasterizer = kwargs.get("rasterizer", args[3] if len(args) > 3 else None)
    if not isinstance(rasterizer, MeshRasterizer):
        raise NotImplementedError("only the mesh rasteriser variant exists here (the VoGE renderer is not available)")
    vertices = kwargs.get("vertices", args[0] if args else None)
    if isinstance(vertices, (list, tuple)):
        return MeshInterpolateModuleMeshes(*args, **kwargs)
    return MeshInterpolateModuleMesh(*args, **kwargs)


class _RenderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, campos, theta, principal, module, nearest=False):
        cam = module.camera()
        out = ops.render(module.mesh, cam, campos, theta, principal, maps=True, nearest=nearest)["maps"]
        ctx.module, ctx.cam, ctx.nearest = module, cam, nearest
        ctx.save_for_backward(campos.detach(), theta.detach(), None if principal is None else principal.detach())
        ctx.shapes = (campos.shape, theta.shape, None if principal is None else principal.shape)
        return out

    @staticmethod
    def backward(ctx, grad_maps):
        campos, theta, principal = ctx.saved_tensors
        want_p = principal is not None and ctx.needs_input_grad[2]
        gC, gT, gP = ops.render_backward(ctx.module.mesh, ctx.cam, campos, theta, principal, grad_maps.contiguous(),
                                         want_principal=want_p, nearest=ctx.nearest)
        gP_out = None
        if want_p:
            gP_out = gP if ctx.shapes[2][0] == gP.shape[0] else gP.sum(dim=0, keepdim=True)
            gP_out = gP_out.reshape(ctx.shapes[2])
        return gC.reshape(ctx.shapes[0]), gT.reshape(ctx.shapes[1]), gP_out, None, None


class MeshInterpolateModuleMesh(nn.Module):
    def __init__(self, vertices, faces, memory_bank, rasterizer, post_process=None, off_set_mesh=False, **kwargs):
        super().__init__()
        if isinstance(vertices, (list, tuple)):
            raise ValueError("a list of meshes is taken by MeshInterpolateModule(...) / MeshInterpolateModuleMeshes")
        if off_set_mesh:
            raise NotImplementedError("off_set_mesh / deform_verts is not supported")
        self.n_mesh = 1
        self.faces = torch.as_tensor(faces)
        self.verts = pre_process_mesh_pascal(torch.as_tensor(vertices, dtype=torch.float32))
        self.memory = None
        self.rasterizer = rasterizer
        self.post_process = post_process
        self.off_set_mesh = off_set_mesh
        self.mesh = None
        self.update_memory(memory_bank, faces=None)

    # ---- state -----------------------------------------------------------------------------
    def update_memory(self, memory_bank, faces=None):
        if isinstance(memory_bank, (list, tuple)):
            raise NotImplementedError("one memory bank per module")
        if faces is not None:
            self.faces = torch.as_tensor(faces)
            if self.mesh is not None:
                dev = self.mesh.device
                self.mesh.close()
                self.mesh = ops.MeshHandle(self.verts.to(dev), self.faces.to(dev))
        self.memory = torch.as_tensor(memory_bank, dtype=torch.float32)
        if self.mesh is not None:
            self.mesh.set_memory(self.memory)

    @property
    def face_memory(self):
        """[F, 3, C] view of the memory in the reference's layout (materialised on demand only)."""
        return self.memory[self.faces.long().to(self.memory.device)]

    def to(self, *args, **kwargs):
        device = torch.device(kwargs["device"] if "device" in kwargs else args[0])
        self.rasterizer.cameras = self.rasterizer.cameras.to(device)
        if device.type == "cuda":
            if self.mesh is not None:
                self.mesh.close()
            self.mesh = ops.MeshHandle(self.verts.to(device), self.faces.to(device))
            self.memory = self.memory.to(device)
            self.mesh.set_memory(self.memory)
        elif self.mesh is not None:
            raise RuntimeError("MeshInterpolateModule has no CPU path")
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda") if device is None else device)

    def update_rasterizer(self, rasterizer):
        device = self.rasterizer.cameras.device
        self.rasterizer = rasterizer
        self.rasterizer.cameras = self.rasterizer.cameras.to(device)

    def camera(self):
        """nemo_camera from the (possibly mutated) rasterizer.cameras / raster_settings."""
        cams = self.rasterizer.cameras
        rs = self.rasterizer.raster_settings
        if getattr(rs, "faces_per_pixel", 1) != 1:
            raise NotImplementedError("faces_per_pixel must be 1")
        if getattr(rs, "blur_radius", 0.0) not in (0, 0.0):
            raise NotImplementedError("blur_radius must be 0")
        H, W = (int(v) for v in cams.image_size[0].tolist())
        pp = cams.principal_point
        px, py = (float(v) for v in pp.detach().reshape(-1, 2)[0].tolist())
        return ops.Camera(H, W, float(cams.focal_length.reshape(-1)[0]), px, py)

    def _principal_arg(self, n):
        """Per-camera principal points when the camera holds more than one (or a Parameter)."""
        pp = self.rasterizer.cameras.principal_point
        if not torch.is_tensor(pp):
            return None
        pp = pp.reshape(-1, 2)
        if pp.shape[0] == 1 and not pp.requires_grad:
            return None
        if pp.shape[0] not in (1, n):
            raise ValueError(f"cameras hold {pp.shape[0]} principal points for {n} poses")
        return pp

    # ---- forward ---------------------------------------------------------------------------
    def forward(self, campos, theta, blur_radius=0, deform_verts=None, mode="bilinear", **kwargs):
        campos = campos.reshape(-1, 3)
        theta = theta.reshape(-1)
        if theta.numel() == 1 and campos.shape[0] > 1:
            theta = theta.expand(campos.shape[0])
        out = self._render(campos, theta, self._principal_arg(campos.shape[0]), blur_radius, deform_verts, mode)
        if self.post_process is not None:
            out = self.post_process(out)
        return out

    def _render(self, campos, theta, principal, blur_radius=0, deform_verts=None, mode="bilinear"):
        if self.mesh is None:
            raise RuntimeError("call .to(device) / .cuda() first: MeshInterpolateModule renders on the GPU only")
        if blur_radius not in (0, 0.0):
            raise NotImplementedError("blur_radius > 0 is not supported")
        if mode not in ("bilinear", "nearest"):
            raise ValueError(f"unknown interpolation mode {mode!r}")
        if deform_verts is not None:
            raise NotImplementedError("deform_verts is not supported")
        needs_grad = torch.is_grad_enabled() and (campos.requires_grad or theta.requires_grad or (
            principal is not None and principal.requires_grad))
        nearest = mode == "nearest"         # nemo/utils/mesh.py:59-70; its backward is the straight-through softmax
        if needs_grad:
            return _RenderFn.apply(campos, theta, principal, self, nearest)
        return ops.render(self.mesh, self.camera(), campos, theta, principal, maps=True, nearest=nearest)["maps"]

    def rasterize(self, campos, theta):
        """Fragments of the same rasterisation: pix_to_face [n,H,W] int64 (packed index i*F + f as
        PyTorch3D reports after Meshes.extend, -1 background), bary_coords [n,H,W,3], zbuf [n,H,W]."""
        campos = campos.reshape(-1, 3)
        out = ops.render(self.mesh, self.camera(), campos, theta.reshape(-1), self._principal_arg(campos.shape[0]),
                         maps=False, fragments=True)
        p2f = out["pix_to_face"].long()
        offs = (torch.arange(campos.shape[0], device=p2f.device) * self.mesh.F)[:, None, None]
        return dict(pix_to_face=torch.where(p2f >= 0, p2f + offs, p2f), bary_coords=out["bary"], zbuf=out["zbuf"],
                    pix_to_face_local=out["pix_to_face"])


class MeshInterpolateModuleMeshes(nn.Module):
    """Several meshes in one module (reference mesh_interpolate_module.py:110-117, :182-191): forward with as many
    cameras as meshes renders mesh i from camera i, with one camera every mesh from that camera; the result is
    [n_mesh, C, H, W].  Every mesh is a single-mesh module of its own (own handle, own launch) sharing the
    rasteriser; `parts[i]` is what a solver takes for category i."""

    def __init__(self, vertices, faces, memory_bank, rasterizer, post_process=None, off_set_mesh=False, **kwargs):
        super().__init__()
        if not (isinstance(faces, (list, tuple)) and isinstance(memory_bank, (list, tuple))) or \
                not (len(vertices) == len(faces) == len(memory_bank)) or len(vertices) == 0:
            raise ValueError("vertices, faces and memory_bank must be lists of the same (non-zero) length")
        self.n_mesh = len(vertices)
        self.parts = nn.ModuleList([MeshInterpolateModuleMesh(v, f, m, rasterizer, post_process=None, off_set_mesh=off_set_mesh)
                                    for v, f, m in zip(vertices, faces, memory_bank)])
        self.rasterizer = rasterizer
        self.post_process = post_process
        self.off_set_mesh = off_set_mesh

    @property
    def faces(self):
        return [p.faces for p in self.parts]

    @property
    def face_memory(self):
        """[sum F, 3, C]: the meshes' face memories back to back, as the reference keeps them."""
        return torch.cat([p.face_memory for p in self.parts], dim=0)

    def update_memory(self, memory_bank, faces=None):
        if not isinstance(memory_bank, (list, tuple)) or len(memory_bank) != self.n_mesh:
            raise ValueError(f"memory_bank must be a list of {self.n_mesh} banks")
        for i, p in enumerate(self.parts):
            p.update_memory(memory_bank[i], None if faces is None else faces[i])

    def to(self, *args, **kwargs):
        for p in self.parts:
            p.to(*args, **kwargs)
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda") if device is None else device)

    def update_rasterizer(self, rasterizer):
        self.rasterizer = rasterizer
        for p in self.parts:
            p.update_rasterizer(rasterizer)

    def camera(self):
        return self.parts[0].camera()

    def forward(self, campos, theta, blur_radius=0, deform_verts=None, mode="bilinear", **kwargs):
        campos = campos.reshape(-1, 3)
        theta = theta.reshape(-1)
        n_cam = campos.shape[0]
        if n_cam not in (1, self.n_mesh):
            raise ValueError(f"{n_cam} cameras for {self.n_mesh} meshes")
        if theta.numel() == 1 and n_cam > 1:
            theta = theta.expand(n_cam)
        pp = self.parts[0]._principal_arg(n_cam)
        outs = []
        for i, p in enumerate(self.parts):
            j = i if n_cam > 1 else 0
            pi = None if pp is None else (pp[j:j + 1] if pp.shape[0] > 1 else pp)
            outs.append(p._render(campos[j:j + 1], theta[j:j + 1], pi, blur_radius, deform_verts, mode))
        out = torch.cat(outs, dim=0)
        if self.post_process is not None:
            out = self.post_process(out)
        return out
