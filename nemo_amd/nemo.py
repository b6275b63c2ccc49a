"""NeMo - the inference façade of the render-and-compare pose model
(reference nemo/models/nemo.py: _build_inference :211-325, fast_inference :327-360, evaluate :364-423).

Same entry points and sample-dict keys as the reference (`img`, `azimuth`, `elevation`, `theta`,
`distance`, `principal`, `bbox`, `this_name`); the pose solve goes through the HIP path
(batch_solve_pose.solve_pose).  The CNN backbone stays a PyTorch module supplied by the caller: anything
with `forward_test(img) -> [B, C, H/stride, W/stride]` unit-norm feature maps (reference
nemo/models/keypoint_representation_net.py:360-391).  Training (`train`, `_build_train`) and the VoGE /
keypoint-projector paths are outside this package.
"""
import numpy as np
import torch

from . import ops
from .batch_solve_pose import get_pre_render_samples, solve_pose as batch_solve_pose
from .solve_pose_keypoint import pre_compute_kp_coords, solve_pose as keypoint_solve_pose
from .cameras import MeshRasterizer, PerspectiveCameras, RasterizationSettings
from .feature_banks import FeatureBankNeMo
from .mesh_interpolate_module import MeshInterpolateModule
from .utils import get_param_samples, load_off, normalize_features, pose_error


class IdentityBackbone(torch.nn.Module):
    """Backbone stand-in for synthetic runs: the 'image' already is the feature map."""
    net_stride = 8

    def forward_test(self, img):
        return img


class NeMo:
    def __init__(self, cfg, cate, mode="test", backbone=None, memory_bank=None, num_noise=5, max_group=512,
                 down_sample_rate=8, mesh_path=None, training=None, inference=None, proj_mode="runtime", checkpoint=None,
                 transforms=(), device="cuda:0", image_size=None, mesh=None, **kwargs):
        if mode == "train":
            raise NotImplementedError("training is outside the MI355X solver path")
        self.cfg, self.cate, self.mode = cfg, cate, mode
        self.device = torch.device(device)
        self.num_noise, self.max_group, self.down_sample_rate = num_noise, max_group, down_sample_rate
        self.inference_params = inference if inference is not None else cfg.inference
        self.mesh_path = mesh_path.format(cate) if (mesh_path and "{:s}" in mesh_path) else mesh_path
        self.checkpoint = torch.load(checkpoint, map_location="cpu", weights_only=True) if isinstance(checkpoint, str) else checkpoint
        self.net = backbone if isinstance(backbone, torch.nn.Module) else IdentityBackbone()
        self.memory_bank_params = memory_bank or {}
        self.transforms = list(transforms)
        self._image_size = image_size            # (H, W) in image pixels
        self._mesh = mesh                        # (verts, faces) instead of mesh_path
        self._build_inference()

    # reference nemo.py:211-325
    def _build_inference(self):
        self.net = self.net.to(self.device)
        if self.checkpoint is not None and "state" in self.checkpoint and not isinstance(self.net, IdentityBackbone):
            self.net.load_state_dict(self.checkpoint["state"])
        if self._mesh is not None:
            xvert, xface = (torch.as_tensor(t) for t in self._mesh)
        else:
            xvert, xface = load_off(self.mesh_path, to_torch=True)
        self.num_verts = int(xvert.shape[0])
        memory = self.checkpoint["memory"].detach().float().cpu()
        input_size = int(memory.shape[1])
        self.memory_bank = FeatureBankNeMo(input_size=input_size, num_pos=self.num_verts, num_noise=0).load_memory(
            memory[: self.num_verts])
        self.feature_bank = memory[: self.num_verts]
        clutter = memory[self.num_verts:].to(self.device)
        self.clutter_bank = normalize_features(torch.mean(clutter, dim=0)).unsqueeze(0)
        self.kp_features = memory[: self.num_verts].to(self.device)

        image_h, image_w = self._image_size if self._image_size is not None else self.cfg.dataset.image_sizes[self.cate]
        map_shape = (image_h // self.down_sample_rate, image_w // self.down_sample_rate)
        cam_cfg = dict(self.inference_params.get("cameras", {})) if hasattr(self.inference_params, "get") else {}
        focal = cam_cfg.get("focal_length", 3000) / self.down_sample_rate
        cameras = PerspectiveCameras(focal_length=focal, principal_point=((map_shape[1] // 2, map_shape[0] // 2),),
                                     image_size=(map_shape,), in_ndc=False, device=self.device)
        raster_settings = RasterizationSettings(image_size=map_shape, blur_radius=0.0, faces_per_pixel=1, bin_size=0)
        self.inter_module = MeshInterpolateModule(xvert, xface, self.feature_bank,
                                                  rasterizer=MeshRasterizer(cameras=cameras, raster_settings=raster_settings),
                                                  post_process=None).to(self.device)
        az, el, th, di, px, py = get_param_samples(self.cfg)
        self.init_mode = self.cfg.inference.get("init_mode", "3d_batch")
        self._similarity = None
        if "batch" not in self.init_mode:
            # keypoint initialisation (reference nemo.py:313-320): projected vertices of every grid pose
            self.poses, self.kp_coords, self.kp_vis = pre_compute_kp_coords(
                xvert.numpy() if self._mesh is not None else self.mesh_path, azimuth_samples=az, elevation_samples=el,
                theta_samples=th, distance_samples=di, viewport=cam_cfg.get("focal_length", 3000))
            return
        dof = int(self.init_mode.split("d_")[0])
        self.feature_pre_rendered, self.cam_pos_pre_rendered, self.theta_pre_rendered = get_pre_render_samples(
            self.inter_module, azum_samples=az, elev_samples=el, theta_samples=th, distance_samples=di, device=self.device)
        if dof == 3:
            assert di.shape[0] == 1
            self.record_distance = float(di[0])
        if dof == 6:
            self.samples_principal = torch.stack(
                (torch.from_numpy(px).view(-1, 1).expand(-1, py.shape[0]) * map_shape[1],
                 torch.from_numpy(py).view(1, -1).expand(px.shape[0], -1) * map_shape[0])).view(2, -1).T.float().to(self.device)

    # reference nemo.py:327-360
    def fast_inference(self, sample):
        """Vertex-activation similarity of every image with this category's bank over its bbox crop."""
        bbox = sample["bbox"]
        img = sample["img"].to(self.device)
        with torch.no_grad():
            fm = self.net.forward_test(img)
        B, C, H, W = fm.shape
        boxes = torch.tensor([[int(float(bbox[i][0]) // 8), int(float(bbox[i][1]) // 8), int(float(bbox[i][2]) // 8),
                               int(float(bbox[i][3]) // 8)] for i in range(B)], dtype=torch.int32)
        clu = ops.clutter_score(fm, self.clutter_bank)
        # the vertex x pixel products are reduced to the per-pixel maximum inside the kernel: no [V, HW] table is written
        sim = ops.vertex_similarity_from_features(self.inter_module.mesh, fm, clu, boxes)
        return [s for s in sim]

    # reference nemo.py:364-423
    def evaluate(self, sample, debug=False):
        for t in self.transforms:
            sample = t(sample)
        img = sample["img"].to(self.device)
        with torch.no_grad():
            feature_map = self.net.forward_test(img)
        b = feature_map.shape[0]
        if "batch" not in self.init_mode:
            assert b == 1, "The batch size during validation should be 1"
            preds = keypoint_solve_pose(self.cfg, feature_map, self.inter_module, self.kp_features, self.clutter_bank, self.poses,
                                        self.kp_coords, self.kp_vis, debug=debug, device=self.device)
            return self._score_predictions([preds], sample)
        dof = int(self.init_mode.split("d_")[0])
        if dof == 6:
            principal = self.samples_principal
        elif ("principal" in sample) and self.cfg.inference.get("realign", False):
            principal = sample["principal"].float().to(self.device) / self.down_sample_rate
        else:
            principal = None
        ones = torch.ones(b, device=self.device)
        preds = batch_solve_pose(
            self.cfg, feature_map, self.inter_module, self.clutter_bank,
            cam_pos_pre_rendered=self.cam_pos_pre_rendered, theta_pre_rendered=self.theta_pre_rendered,
            feature_pre_rendered=self.feature_pre_rendered, device=self.device, principal=principal,
            distance_source=sample["distance"].to(self.device) if (dof == 3 and "distance" in sample) else ones,
            distance_target=self.record_distance * ones if dof == 3 else ones,
            pre_render=self.cfg.inference.get("pre_render", True), dof=dof)
        return self._score_predictions(preds, sample)

    @staticmethod
    def _score_predictions(preds, sample):
        classification_result = {}
        for i, pred in enumerate(preds):
            if "azimuth" in sample and "elevation" in sample and "theta" in sample:
                # with several refined proposals the keypoint solver returns the best one as a dict (solve_pose.py:329-338)
                best = pred["final"][0] if isinstance(pred["final"], list) else pred["final"]
                err = pose_error({k: sample[k][i] for k in ["azimuth", "elevation", "theta"]}, best)
                pred["pose_error"] = err
                if "this_name" in sample:
                    classification_result[sample["this_name"][i]] = (best["score"], err)
        return preds, classification_result


def classify_by_render_compare(models, sample):
    """3D-aware classification (reference nemo/models/nemo_cls.py:262-282,
    scripts/classification_accuracy.py:86-87): solve the pose against every category model and take the
    arg-min of the final loss.  `models`: dict category -> NeMo.  -> (labels [B], scores [B, n_cat])."""
    cats = list(models)
    scores = []
    for c in cats:
        preds, _ = models[c].evaluate(sample)
        scores.append([p["final"][0]["score"] for p in preds])
    scores = np.asarray(scores).T
    return [cats[j] for j in scores.argmin(axis=1)], scores
