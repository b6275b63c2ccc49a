"""Drop-in for the reference's CuNeMo extension package (cu_layers/CuNeMo: gather_features.py, mask_weight.py):
`gather_features`, `get_mask`, `get_vert_shifts` with the reference's signatures, on the HIP entry points
nemo_gather_features / nemo_gather_features_backward / nemo_gather_idx / nemo_mask_weight.  torch only allocates
the tensors and builds the small per-image index tables, as the reference's wrappers do.
"""
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr


def _i32c(t):
    return t.to(torch.int32).contiguous()


def _mesh_n_tensor(mesh_n_list, device, dtype=torch.int32):
    if not torch.is_tensor(mesh_n_list):
        mesh_n_list = torch.tensor(list(mesh_n_list), dtype=dtype)
    return mesh_n_list.to(device=device, dtype=dtype)


def get_vert_shifts(indexs, mesh_n_list):
    """(vertex count, first slot) of every object of every image (mask_weight.py:41-47)."""
    mesh_n_list = _mesh_n_tensor(mesh_n_list, indexs.device, indexs.dtype)
    valid_mask = torch.logical_not(indexs < 0).type(indexs.dtype)
    verts_num = valid_mask * torch.gather(mesh_n_list[None].expand(indexs.shape[0], -1), dim=1, index=indexs.long().clamp(min=0))
    sample_shifts = torch.cumsum(verts_num, dim=1) - verts_num
    return verts_num, sample_shifts


class _GatherFeatures(torch.autograd.Function):
    """gather_features.py:37-77."""

    @staticmethod
    def forward(ctx, input_features, input_weights, input_valid, bank_shifts, sample_shifts, sample_indexs, KK):
        lib = _lib.load()
        feat = input_features.contiguous().float()
        w = input_weights.contiguous().float()
        if not feat.is_cuda:
            raise RuntimeError("gather_features runs on the GPU only: move the tensors to 'cuda'")
        N, K, C = feat.shape
        M, L, KK = sample_shifts.shape[1], bank_shifts.shape[0], int(KK)
        iv, bs, ss, si = _i32c(input_valid), _i32c(bank_shifts), _i32c(sample_shifts), _i32c(sample_indexs)
        out_f = torch.empty(KK, C, device=feat.device)
        out_w = torch.empty(KK, device=feat.device)
        fmap = torch.empty(N, K, dtype=torch.int32, device=feat.device)
        with torch.cuda.device(feat.device):
            check(lib.nemo_gather_features(ptr(feat), ptr(w), ptr(iv, torch.int32), ptr(bs, torch.int32), ptr(ss, torch.int32),
                                           ptr(si, torch.int32), N, M, K, C, L, KK, ptr(out_f), ptr(out_w), ptr(fmap, torch.int32),
                                           stream_ptr()), "nemo_gather_features")
        ctx.save_for_backward(feat, w, fmap)
        ctx.mark_non_differentiable(fmap)
        return out_f, out_w, fmap

    @staticmethod
    def backward(ctx, grad_features, grad_weights, grad_mapping):
        lib = _lib.load()
        feat, w, fmap = ctx.saved_tensors
        N, K, C = feat.shape
        gf, gw = grad_features.contiguous().float(), grad_weights.contiguous().float()      # autograd materialises unused ones as zeros
        gi_f = torch.empty_like(feat)
        gi_w = torch.empty_like(w)
        with torch.cuda.device(feat.device):
            check(lib.nemo_gather_features_backward(ptr(gf), ptr(gw), ptr(feat), ptr(w), ptr(fmap, torch.int32), N, K, C, ptr(gi_f),
                                                    ptr(gi_w), stream_ptr()), "nemo_gather_features_backward")
        return gi_f, gi_w, None, None, None, None, None


def gather_features(features, weights, sample_indexs, mesh_n_list):
    """gather_features.py:5-34: features [N,K,C], weights [N,K], sample_indexs [N,M] i32 (mesh of every object of
    every image, -1 = none), mesh_n_list [L] -> (bank-ordered weighted feature sums [sum(mesh_n_list), C],
    weight sums [sum(mesh_n_list)])."""
    dev = features.device
    mesh_n_list = _mesh_n_tensor(mesh_n_list, dev)
    assert dev == weights.device == sample_indexs.device
    assert features.is_cuda
    assert features.dim() == 3 and weights.dim() == 2 and sample_indexs.dim() == 2
    assert -1 <= int(sample_indexs.max()) < mesh_n_list.shape[0]
    assert sample_indexs.dtype == torch.int32
    total_verts = int(mesh_n_list.sum())
    verts_num, sample_shifts = get_vert_shifts(sample_indexs, mesh_n_list)
    input_valid = (sample_shifts[:, -1] + verts_num[:, -1]).type(torch.int32)
    bank_shifts = (torch.cumsum(mesh_n_list, dim=0) - mesh_n_list).type(torch.int32)
    gathered_features, gathered_weights, _ = _GatherFeatures.apply(features, weights, input_valid, bank_shifts,
                                                                  sample_shifts.type(torch.int32), sample_indexs, total_verts)
    return gathered_features, gathered_weights


def gather_idx(input_valid, sample_shifts, sample_indexs, sample_n_verts, cum_sum_n_verts_in_bank, K):
    """_C.gather_idx (mask_weight.py:85-109): -> object_idx, source_vert_idx, verts_start, num_verts, all [N,K] i32."""
    lib = _lib.load()
    if not sample_shifts.is_cuda:
        raise RuntimeError("gather_idx runs on the GPU only: move the tensors to 'cuda'")
    N, M = sample_shifts.shape
    dev = sample_shifts.device
    outs = [torch.zeros(N, int(K), dtype=torch.int32, device=dev) for _ in range(4)]
    args = [_i32c(t) for t in (input_valid, sample_shifts, sample_indexs, sample_n_verts, cum_sum_n_verts_in_bank)]
    with torch.cuda.device(dev):
        check(lib.nemo_gather_idx(*[ptr(a, torch.int32) for a in args], N, M, int(K), *[ptr(o, torch.int32) for o in outs],
                                  stream_ptr()), "nemo_gather_idx")
    return tuple(outs)


def mask_weight(object_idx, source_vert_idx, verts_start, verts_end, verts_dist_weight, weight_x, weight_c, clutter_start, total_v):
    """_C.mask_weight (mask_weight.py:50-83): -> [S, total_v] f32."""
    lib = _lib.load()
    vdw = verts_dist_weight.contiguous().float()
    if not vdw.is_cuda:
        raise RuntimeError("mask_weight runs on the GPU only: move the tensors to 'cuda'")
    S, V = object_idx.shape[0], vdw.shape[1]
    out = torch.empty(S, int(total_v), device=vdw.device)
    args = [_i32c(t) for t in (object_idx, source_vert_idx, verts_start, verts_end)]
    with torch.cuda.device(vdw.device):
        check(lib.nemo_mask_weight(*[ptr(a, torch.int32) for a in args], ptr(vdw), float(weight_x), float(weight_c), int(clutter_start),
                                   S, int(total_v), V, ptr(out), stream_ptr()), "nemo_mask_weight")
    return out


def get_mask(verts_dist_weight, sample_indexs, mesh_n_list, total_size, K_padded, weight_x, weight_c, mask_sel=None, n_noise=0):
    """mask_weight.py:5-38: the loss-weight matrix [S, total_size] of the selected slots against every bank row,
    and the bank row of every selected slot."""
    dev = verts_dist_weight.device
    mesh_n_list = _mesh_n_tensor(mesh_n_list, dev)
    assert dev == sample_indexs.device
    assert verts_dist_weight.is_cuda and verts_dist_weight.dim() == 3
    verts_num, sample_shifts = get_vert_shifts(indexs=sample_indexs, mesh_n_list=mesh_n_list)
    sample_shifts, sample_indexs, verts_num = sample_shifts.type(torch.int32), sample_indexs.type(torch.int32), verts_num.type(torch.int32)
    input_valid = sample_shifts[:, -1] + verts_num[:, -1]
    bank_starts = (torch.cumsum(mesh_n_list, dim=0) - mesh_n_list).type(torch.int32)
    object_idx, source_vert_idx, verts_start, num_verts = gather_idx(input_valid, sample_shifts, sample_indexs, verts_num, bank_starts,
                                                                    K_padded)
    if mask_sel is None:
        object_idx, source_vert_idx = object_idx.view(-1), source_vert_idx.view(-1)
        verts_start, num_verts = verts_start.view(-1), num_verts.view(-1)
    else:
        object_idx, source_vert_idx = object_idx[mask_sel], source_vert_idx[mask_sel]
        verts_start, num_verts = verts_start[mask_sel], num_verts[mask_sel]
    out_weight = mask_weight(object_idx, source_vert_idx, verts_start, verts_start + num_verts, verts_dist_weight, weight_x, weight_c,
                             total_size - n_noise, total_size)
    vert_index = torch.gather(bank_starts.long(), dim=0, index=object_idx.long()) + source_vert_idx
    return out_weight, vert_index.long()
