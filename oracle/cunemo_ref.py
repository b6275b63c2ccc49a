"""TEST INFRASTRUCTURE - CPU restatement (numpy, plain loops) of the reference's CuNeMo extension kernels.

Only tests/ may import this module; the product path never does.  Every function follows the reference
kernel it names, thread index by thread index; the reference accumulates with atomicAdd in no defined
order, this restatement accumulates in slot order in float64 and rounds once, so comparisons with it are
to float32 summation tolerance.

Parity unpinned: the reference ships no tests, fixtures or expected outputs for this extension (only the
input example of cu_layers/CuNeMo/gather_features.py:80-96, whose answer is derived by hand in the tests).
"""
import numpy as np


def slot_object(shifts, k):
    """The search loop shared by the kernels (gather_features.cu:48-60, mask_weight.cu:36-49):
    -> (position i of the last object whose first slot is <= k, or -1)."""
    found = -1
    for i, nxt in enumerate(shifts):
        if nxt == -1 or nxt > k:
            break
        found = i
    return found


def gather_features(feat, weight, input_valid, bank_shifts, sample_shifts, sample_indexs, KK):
    """GatherFeaturesKernel (gather_features.cu:24-68) -> (gathered_features [KK,C], gathered_weights [KK],
    forward_mapping [N,K])."""
    N, K, C = feat.shape
    out_f = np.zeros((KK, C), np.float64)
    out_w = np.zeros(KK, np.float64)
    fmap = np.full((N, K), -1, np.int32)
    for n in range(N):
        for k in range(K):
            if k >= input_valid[n]:
                continue
            i = slot_object(sample_shifts[n], k)
            mesh = sample_indexs[n, i] if i >= 0 else 0
            shift = sample_shifts[n, i] if i >= 0 else 0
            v = k - shift + bank_shifts[mesh]
            fmap[n, k] = v
            out_w[v] += weight[n, k]
            out_f[v] += feat[n, k].astype(np.float64) * np.float64(weight[n, k])
    return out_f.astype(np.float32), out_w.astype(np.float32), fmap


def gather_features_backward(g_feat, g_weight, feat, weight, fmap):
    """GatherFeaturesBackwardKernel (gather_features.cu:71-98)."""
    N, K, C = feat.shape
    gi_f = np.zeros((N, K, C), np.float32)
    gi_w = np.zeros((N, K), np.float32)
    for n in range(N):
        for k in range(K):
            v = fmap[n, k]
            if v < 0:
                continue
            gi_w[n, k] = np.float32(np.float64(g_weight[v]) + np.dot(g_feat[v].astype(np.float64), feat[n, k].astype(np.float64)))
            gi_f[n, k] = g_feat[v] * weight[n, k]
    return gi_f, gi_w


def gather_idx(input_valid, sample_shifts, sample_indexs, sample_n_verts, bank_starts, K):
    """GatherIdxKernel (mask_weight.cu:10-56) -> object_idx, source_vert_idx, verts_start, num_verts [N,K] i32."""
    N = sample_shifts.shape[0]
    outs = [np.zeros((N, K), np.int32) for _ in range(4)]
    for n in range(N):
        for k in range(K):
            if k >= input_valid[n]:
                continue
            i = slot_object(sample_shifts[n], k)
            mesh = sample_indexs[n, i] if i >= 0 else 0
            outs[0][n, k] = mesh
            outs[1][n, k] = k - (sample_shifts[n, i] if i >= 0 else 0)
            outs[2][n, k] = bank_starts[mesh]
            outs[3][n, k] = sample_n_verts[n, i] if i >= 0 else 0
    return tuple(outs)


def mask_weight(object_idx, source_vert_idx, verts_start, verts_end, verts_dist_weight, weight_x, weight_c, clutter_start, K):
    """MaskWeightKernel (mask_weight.cu:59-90) -> [S, K] f32."""
    S = object_idx.shape[0]
    out = np.zeros((S, K), np.float32)
    for s in range(S):
        for k in range(K):
            if verts_start[s] <= k < verts_end[s]:
                out[s, k] = verts_dist_weight[object_idx[s], source_vert_idx[s], k - verts_start[s]]
            elif k >= clutter_start:
                out[s, k] = weight_c
            else:
                out[s, k] = weight_x
    return out


def vert_shifts(sample_indexs, mesh_n_list):
    """get_vert_shifts (cu_layers/CuNeMo/mask_weight.py:41-47) in numpy."""
    mesh_n = np.asarray(mesh_n_list, np.int64)
    valid = (sample_indexs >= 0).astype(np.int64)
    verts_num = valid * mesh_n[np.clip(sample_indexs, 0, None)]
    shifts = np.cumsum(verts_num, axis=1) - verts_num
    return verts_num.astype(np.int32), shifts.astype(np.int32)
