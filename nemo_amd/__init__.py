"""nemo_amd - MI355X-native render-and-compare pose solver behind the reference NeMo Python API.

The compute path is libnemo_hip.so (hand-written HIP for gfx950, C ABI in include/nemo_hip.h);
this package is the host-side mirror of the reference interface for that path.
"""
from . import ops
from .batch_solve_pose import get_pre_render_samples, solve_pose
from .cameras import MeshRasterizer, PerspectiveCameras, RasterizationSettings
from . import cu_nemo
from .feature_banks import FeatureBankNeMo, MaskCreater
from .mesh_interpolate_module import MeshInterpolateModule
from .multi_category import MultiCategorySolver
from .nemo import NeMo, classify_by_render_compare
from . import solve_pose_keypoint
from .project_kp import PackedRaster

__all__ = ["ops", "get_pre_render_samples", "solve_pose", "MeshRasterizer", "PerspectiveCameras",
           "RasterizationSettings", "MeshInterpolateModule", "FeatureBankNeMo", "MaskCreater", "cu_nemo", "NeMo", "classify_by_render_compare", "MultiCategorySolver", "solve_pose_keypoint", "PackedRaster"]
