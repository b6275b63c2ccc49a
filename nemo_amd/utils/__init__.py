from .config import ConfigNode, default_pose3d_config, load_config
from .general import get_param_samples, normalize_features, tensor_linspace
from .mesh import (camera_position_from_spherical_angles, camera_position_to_spherical_angle, load_off, meshelize,
                   pre_process_mesh_pascal, save_off, vertex_memory_to_face_memory)
from .pose import cal_rotation_matrix, get_transformation_matrix, pose_error
