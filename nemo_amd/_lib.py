"""ctypes binding of libnemo_hip.so (the C ABI declared in include/nemo_hip.h).

The library is the product path: there is no CPU or PyTorch fallback.  If the shared object is
missing or a call fails, a RuntimeError is raised.  Tensors are handed over as raw device
pointers; work is enqueued on torch's current HIP stream.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NEMO_HIP_LIB") or os.path.join(_HERE, "libnemo_hip.so")      # NEMO_HIP_LIB: a diagnostic build

# every symbol include/nemo_hip.h declares
SYMBOLS = [
    "nemo_last_error", "nemo_version", "nemo_mesh_create", "nemo_mesh_set_memory", "nemo_mesh_destroy",
    "nemo_mesh_info", "nemo_mesh_set_parts", "nemo_solve_parts", "nemo_project", "nemo_render", "nemo_render_backward", "nemo_render_nearest_backward", "nemo_clutter_score", "nemo_align_maps",
    "nemo_grid_score_workspace", "nemo_grid_score", "nemo_grid_score_static", "nemo_grid_pack_bytes", "nemo_grid_pack",
    "nemo_grid_score_packed_workspace", "nemo_grid_score_packed", "nemo_clutter_grid_score_packed", "nemo_grid_fragments_bytes", "nemo_grid_score_fragments_workspace", "nemo_grid_score_fragments", "nemo_grid_score_recentred_workspace",
    "nemo_grid_score_recentred", "nemo_similarity_table_elems", "nemo_similarity_table", "nemo_vertex_similarity",
    "nemo_vertex_similarity_workspace", "nemo_vertex_similarity_features",
    "nemo_solve", "nemo_keypoint_corr", "nemo_solve_flow", "nemo_spherical_angles",
    "nemo_solve_features_workspace", "nemo_solve_features_supported", "nemo_solve_features", "nemo_render_nearest", "nemo_raster_forms",
    "nemo_project_keypoints", "nemo_gather_features", "nemo_gather_features_backward", "nemo_gather_idx", "nemo_mask_weight",
]


class NemoCamera(C.Structure):
    _fields_ = [("H", C.c_int32), ("W", C.c_int32), ("focal", C.c_float), ("px", C.c_float), ("py", C.c_float)]


class NemoAdam(C.Structure):
    _fields_ = [("steps", C.c_int32), ("lr_every", C.c_int32), ("lr", C.c_double), ("beta1", C.c_double),
                ("beta2", C.c_double), ("eps", C.c_double), ("lr_gamma", C.c_double)]


_lib = None


def load():
    """Load the shared library (once) and declare the signatures."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `make -C nemo_amd/csrc` (or __graft_entry__.build()). "
            "There is no fallback path.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, f32p = C.c_void_p, C.c_int32, C.c_void_p
    lib.nemo_last_error.restype = C.c_char_p
    lib.nemo_version.restype = C.c_char_p
    lib.nemo_mesh_create.argtypes = [vp, vp, i32, i32, C.POINTER(vp)]
    lib.nemo_mesh_set_memory.argtypes = [vp, vp, i32, vp]
    lib.nemo_mesh_destroy.argtypes = [vp]
    lib.nemo_mesh_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    lib.nemo_project.argtypes = [vp, C.POINTER(NemoCamera), f32p, f32p, f32p, i32, f32p, f32p, vp]
    lib.nemo_render.argtypes = [vp, C.POINTER(NemoCamera), f32p, f32p, f32p, i32, f32p, vp, f32p, f32p, vp]
    lib.nemo_render_backward.argtypes = [vp, C.POINTER(NemoCamera), f32p, f32p, f32p, i32, f32p, f32p, f32p, f32p, vp]
    lib.nemo_render_nearest_backward.argtypes = [vp, C.POINTER(NemoCamera), f32p, f32p, f32p, i32, f32p, f32p, f32p, f32p, vp]
    lib.nemo_clutter_score.argtypes = [f32p, f32p, i32, i32, i32, i32, f32p, vp]
    lib.nemo_grid_score_workspace.argtypes = [i32, i32, i32]
    lib.nemo_grid_score_workspace.restype = C.c_size_t
    lib.nemo_grid_score.argtypes = [f32p, f32p, f32p, i32, i32, i32, i32, f32p, vp, f32p, vp, C.c_size_t, vp]
    lib.nemo_grid_score_static.argtypes = [f32p, f32p, f32p, i32, i32, i32, i32, f32p, vp, f32p, vp]
    lib.nemo_grid_pack_bytes.argtypes = [i32, i32, i32]
    lib.nemo_grid_pack_bytes.restype = C.c_size_t
    lib.nemo_grid_pack.argtypes = [f32p, i32, i32, i32, vp, vp]
    lib.nemo_grid_score_packed_workspace.argtypes = [i32, i32, i32, i32]
    lib.nemo_grid_score_packed_workspace.restype = C.c_size_t
    lib.nemo_grid_score_packed.argtypes = [vp, f32p, f32p, i32, i32, i32, i32, f32p, vp, f32p, vp, C.c_size_t, vp]
    lib.nemo_grid_fragments_bytes.argtypes = [i32, i32, i32]
    lib.nemo_grid_fragments_bytes.restype = C.c_size_t
    lib.nemo_grid_score_fragments_workspace.argtypes = [i32, i32, i32]
    lib.nemo_grid_score_fragments_workspace.restype = C.c_size_t
    lib.nemo_grid_score_fragments.argtypes = [vp, vp, f32p, i32, i32, i32, i32, f32p, vp, f32p, vp, C.c_size_t, vp]
    lib.nemo_clutter_grid_score_packed.argtypes = [vp, f32p, f32p, i32, i32, i32, i32, i32, f32p, f32p, vp, f32p, vp, C.c_size_t, vp]
    lib.nemo_grid_score_recentred_workspace.argtypes = [i32, i32, i32, i32, i32]
    lib.nemo_grid_score_recentred_workspace.restype = C.c_size_t
    lib.nemo_grid_score_recentred.argtypes = [vp, f32p, f32p, f32p, f32p, f32p, f32p, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32p, vp,
                                              f32p, vp, C.c_size_t, vp]
    lib.nemo_similarity_table_elems.argtypes = [vp, i32]
    lib.nemo_similarity_table_elems.restype = C.c_size_t
    lib.nemo_similarity_table.argtypes = [vp, f32p, i32, i32, f32p, vp]
    lib.nemo_vertex_similarity.argtypes = [vp, f32p, f32p, vp, i32, i32, i32, f32p, vp]
    lib.nemo_vertex_similarity_workspace.argtypes = [i32, i32]
    lib.nemo_vertex_similarity_workspace.restype = C.c_size_t
    lib.nemo_vertex_similarity_features.argtypes = [vp, f32p, f32p, vp, i32, i32, i32, f32p, vp, C.c_size_t, vp]
    lib.nemo_solve.argtypes = [vp, C.POINTER(NemoCamera), f32p, f32p, vp, f32p, f32p, f32p, vp, i32, i32, i32, i32,
                               C.POINTER(NemoAdam), f32p, f32p, vp]
    lib.nemo_mesh_set_parts.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), i32]
    lib.nemo_solve_parts.argtypes = [vp, C.POINTER(NemoCamera), f32p, f32p, vp, vp, f32p, f32p, f32p, vp, i32, i32, i32, i32,
                                     C.POINTER(NemoAdam), f32p, f32p, vp]
    lib.nemo_keypoint_corr.argtypes = [vp, f32p, f32p, f32p, vp, vp, i32, i32, i32, C.c_float, i32, i32, f32p, vp]
    lib.nemo_solve_flow.argtypes = [vp, C.POINTER(NemoCamera), f32p, f32p, vp, f32p, f32p, f32p, i32, i32, i32, i32, C.c_float,
                                    C.POINTER(NemoAdam), f32p, f32p, vp]
    lib.nemo_align_maps.argtypes = [f32p, f32p, f32p, f32p, f32p, i32, i32, i32, i32, i32, i32, i32, f32p, vp]
    lib.nemo_spherical_angles.argtypes = [f32p, i32, f32p, vp]
    lib.nemo_project_keypoints.argtypes = [vp, C.POINTER(NemoCamera), f32p, f32p, f32p, f32p, i32, C.c_float, i32, f32p, vp, vp]
    lib.nemo_solve_features_supported.argtypes = [vp, i32, i32, i32]
    lib.nemo_render_nearest.argtypes = [vp, C.POINTER(NemoCamera), f32p, f32p, f32p, i32, f32p, vp]
    lib.nemo_raster_forms.argtypes = [vp, C.POINTER(NemoCamera), f32p, f32p, i32, i32, i32, f32p, vp, vp]
    lib.nemo_solve_features_workspace.argtypes = [vp, i32, i32]
    lib.nemo_solve_features_workspace.restype = C.c_size_t
    lib.nemo_solve_features.argtypes = [vp, C.POINTER(NemoCamera), f32p, vp, C.c_size_t, f32p, vp, vp, f32p, f32p, f32p, vp, i32, i32,
                                        i32, i32, C.POINTER(NemoAdam), f32p, f32p, vp]
    lib.nemo_gather_features.argtypes = [f32p, f32p, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32p, f32p, vp, vp]
    lib.nemo_gather_features_backward.argtypes = [f32p, f32p, f32p, f32p, vp, i32, i32, i32, f32p, f32p, vp]
    lib.nemo_gather_idx.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp]
    lib.nemo_mask_weight.argtypes = [vp, vp, vp, vp, f32p, C.c_float, C.c_float, i32, i32, i32, i32, f32p, vp]
    for name in SYMBOLS:
        fn = getattr(lib, name)
        if name not in ("nemo_last_error", "nemo_version", "nemo_grid_score_workspace", "nemo_similarity_table_elems",
                        "nemo_grid_pack_bytes", "nemo_grid_score_packed_workspace", "nemo_grid_score_recentred_workspace", "nemo_grid_fragments_bytes",
                        "nemo_grid_score_fragments_workspace", "nemo_vertex_similarity_workspace",
                        "nemo_solve_features_workspace"):
            fn.restype = C.c_int
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        raise RuntimeError(f"{what} failed ({code}): {load().nemo_last_error().decode()}")


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, dtype=torch.float32):
    """Device pointer of a contiguous CUDA tensor of the expected dtype (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("nemo_amd kernels need device tensors (got a CPU tensor); there is no CPU path")
    if t.dtype != dtype:
        raise RuntimeError(f"expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError("tensor must be contiguous")
    return C.c_void_p(t.data_ptr())


def f32c(t):
    """float32 contiguous device copy/view of `t` (detached).  CPU tensors are refused: no CPU path."""
    if not t.is_cuda:
        raise RuntimeError("nemo_amd kernels need device tensors (got a CPU tensor); there is no CPU path")
    return t.detach().to(torch.float32).contiguous()
