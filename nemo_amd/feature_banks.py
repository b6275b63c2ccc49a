"""FeatureBankNeMo - the vertex / clutter feature memory (reference nemo/models/feature_banks.py:133-232) and the
loss-weight builder MaskCreater (:39-130).

At inference the bank is a container: `memory` = [memory_pos (V_total rows) ; memory_neg (clutter rows)], and a
multi-class bank is the per-mesh vertex blocks back to back (block i starts at the sum of the vertex counts before
it, the layout cu_layers/CuNeMo/gather_features.py:31 assumes).  `mesh_offsets` / `vertex_block` / `load_memory`
give the pose solver what it needs.  Constructor signature, attribute names and the return values of forward /
__call__ are the reference's (they are the drop-in boundary); the bodies are written for this package: the
clutter ring buffer is a slice assignment, the weight matrices come from one `torch.where`, and the multi-class
gather runs on the HIP replacement of the reference's CuNeMo extension (cu_nemo.gather_features / get_mask).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .cu_nemo import gather_features, get_mask

_DEFAULT_KAPPAS = {"pos": 0, "near": 1e5, "clutter": 5, "class": 1e5}


def _pair_weights(points, dist_thr, w_pos, w_near):
    """[..., K, 3 or 2] point sets -> [..., K, K] weights: w_pos on the diagonal, w_near for pairs no farther
    apart than dist_thr, 0 elsewhere."""
    gap = torch.linalg.vector_norm(points.unsqueeze(-2) - points.unsqueeze(-3), dim=-1)
    k = points.shape[-2]
    same = torch.eye(k, dtype=torch.bool, device=points.device).expand(gap.shape)
    zero = torch.zeros((), dtype=torch.float32, device=points.device)
    near = torch.where(gap <= dist_thr, zero + float(w_near), zero)
    return torch.where(same, zero + float(w_pos), near)


class MaskCreater:
    """Weights of the contrastive vertex loss: for every selected vertex slot one row over
    [every bank vertex | n_noise clutter rows].

    One mesh per image (sample_indexs is None): row = [pair weights of that vertex | kappa_clutter ...], from the mesh
    geometry given at construction (`verts_ori`, with dtype_template giving the batch size) or from per-image
    keypoint positions `kps`.  Several meshes per batch (sample_indexs given): the pair weights of the slot's own mesh
    sit at that mesh's bank columns, kappa_class on the other meshes' columns, kappa_clutter on the clutter columns
    (cu_nemo.get_mask).  Returns (weights, bank row of every selected slot)."""

    def __init__(self, dist_thr, kappas, n_noise, verts_ori=None, device="cpu"):
        self.kappas = dict(_DEFAULT_KAPPAS)
        self.kappas.update(kappas)
        self.dist_thr, self.n_noise, self.device = dist_thr, n_noise, device
        self.verts_dist_weight = None
        if verts_ori is None:
            return
        if not isinstance(verts_ori, list):
            raise TypeError("verts_ori is a list with one [K_i, 3] tensor per mesh")
        self.mesh_n_list = [int(v.shape[0]) for v in verts_ori]
        self.vert_sum_num = sum(self.mesh_n_list)
        widest = max(self.mesh_n_list)
        with torch.no_grad():
            padded = torch.zeros(len(verts_ori), widest, 3, dtype=verts_ori[0].dtype)
            for i, v in enumerate(verts_ori):
                padded[i, : v.shape[0]] = v.detach().cpu()
            # kappas as passed (not the defaults): the reference builds the stored matrix before merging them
            self.verts_dist_weight = _pair_weights(padded.float(), dist_thr, kappas["pos"], kappas["near"]).to(device)

    def _single_mesh_rows(self, pair_w, vis_mask, dtype_template):
        if pair_w.dim() == 2:
            pair_w = pair_w[None]
        n_set, k = pair_w.shape[0], pair_w.shape[1]
        clutter_cols = pair_w.new_full((n_set, k, self.n_noise), float(self.kappas["clutter"]))
        rows = torch.cat([pair_w, clutter_cols], dim=2)
        if n_set == 1:
            if dtype_template is None:
                raise ValueError("dtype_template gives the batch size when one mesh serves every image")
            n_set = dtype_template.shape[0]
            rows = rows.expand(n_set, -1, -1).contiguous()
        labels = torch.arange(k, device=self.device, dtype=torch.long).repeat(n_set, 1)
        if vis_mask is None:
            return rows.view(-1), labels.view(-1)
        return rows[vis_mask], labels[vis_mask]

    def __call__(self, sample_indexs=None, K_padded=None, vis_mask=None, kps=None, dtype_template=None):
        if kps is None:
            pair_w = self.verts_dist_weight
        else:
            pair_w = _pair_weights(kps.float(), self.dist_thr, self.kappas["pos"], self.kappas["near"]).to(self.device)
        if sample_indexs is None:
            return self._single_mesh_rows(pair_w, vis_mask, dtype_template)
        if kps is not None:
            raise NotImplementedError("several meshes per batch: only the mesh-geometry weights are supported")
        if K_padded is None:
            if dtype_template is None:
                raise ValueError("K_padded or dtype_template is required")
            K_padded = dtype_template.shape[1]
        idx = sample_indexs if sample_indexs.dim() == 2 else sample_indexs[:, None]
        return get_mask(pair_w, idx, self.mesh_n_list, self.vert_sum_num + self.n_noise, K_padded, self.kappas["class"],
                        self.kappas["clutter"], mask_sel=vis_mask, n_noise=self.n_noise)


class FeatureBankNeMo(nn.Module):
    def __init__(self, input_size, num_pos, num_noise=-1, max_groups=-1, momentum=0.5, **kwargs):
        super().__init__()
        bound = 1.0 / math.sqrt(input_size / 3)
        n_neg = num_noise * max_groups if max_groups > 0 else 0
        self.memory_pos = torch.empty(num_pos, input_size).uniform_(-bound, bound)
        self.memory_neg = torch.empty(n_neg, input_size).uniform_(-bound, bound)
        self.num_noise = num_noise
        self.max_lru = max_groups if max_groups > 0 else -1
        self.lru = 0                      # image slot of the clutter ring buffer that is written next
        self.momentum = momentum
        self.kwargs = kwargs              # mesh_n_list: vertex count of every mesh of a multi-class bank

    # ---- container ----
    @property
    def memory(self):
        return torch.cat((self.memory_pos, self.memory_neg), dim=0)

    def mesh_offsets(self):
        """First memory_pos row of every mesh (multi-class layout: blocks back to back)."""
        counts = self.kwargs.get("mesh_n_list")
        if counts is None:
            return [0]
        starts = [0]
        for n in list(counts)[:-1]:
            starts.append(starts[-1] + int(n))
        return starts

    def vertex_block(self, mesh_index=0):
        """memory_pos rows of one mesh, [V_mesh, C]."""
        counts = self.kwargs.get("mesh_n_list")
        if counts is None:
            return self.memory_pos
        first = self.mesh_offsets()[mesh_index]
        return self.memory_pos[first:first + int(counts[mesh_index])]

    def load_memory(self, memory):
        """Fill the bank from a checkpoint's `memory` tensor: rows [0, V_total) vertices, the rest clutter."""
        n_pos = self.memory_pos.shape[0]
        dev = self.memory_pos.device
        self.memory_pos = memory[:n_pos].detach().clone().to(dev)
        self.memory_neg = memory[n_pos:].detach().clone().to(dev)
        return self

    def to(self, device):
        self.memory_pos, self.memory_neg = self.memory_pos.to(device), self.memory_neg.to(device)
        return self

    def cuda(self, device=None):
        return self.to("cuda" if device is None else device)

    # ---- training-time update (reference :165-222) ----
    def _vertex_and_clutter_rows(self, x):
        if self.num_noise == 0:
            return x, None
        k = x.shape[1] - self.num_noise
        return x[:, :k], x[:, k:]

    def _bank_target(self, feats, visible, object_labels):
        """Per bank row the batch-averaged visible feature that the momentum update blends in."""
        if feats.dim() != 3:
            return feats                                    # the caller already reduced it to [V_total, C]
        n_img = feats.shape[0]
        if object_labels is None:
            return (feats * visible.to(feats.dtype).unsqueeze(-1)).sum(dim=0) / n_img
        summed, _ = gather_features(feats, weights=visible.to(torch.float32), sample_indexs=object_labels.to(torch.int32),
                                    mesh_n_list=self.kwargs.get("mesh_n_list"))
        return summed / n_img

    def _store_clutter(self, clutter_rows):
        """Ring buffer of max_lru images x num_noise rows: the batch's rows go to image slot `lru` onwards; rows
        that would run past the end of the buffer are dropped (no wrap within one batch)."""
        new = clutter_rows.reshape(-1, clutter_rows.shape[-1])
        cap = self.memory_neg.shape[0]
        start = 0 if new.shape[0] > cap else self.lru * self.num_noise
        fits = max(min(new.shape[0], cap - start), 0)
        buf = self.memory_neg.clone()
        buf[start:start + fits] = new[:fits]
        self.memory_neg = buf

    def forward(self, x, visible, x_to_bank=None, object_labels=None, vis_mask=None):
        """x [B, K (+ num_noise), C] vertex (+ clutter) features, visible [B, K].
        -> (similarity of the selected vertex features with every bank row, similarity of the clutter features with
        the vertex rows); the bank is then updated without gradient: momentum blend + re-normalisation of the vertex
        rows, clutter rows into the ring buffer."""
        vert_feats, clutter_feats = self._vertex_and_clutter_rows(x)
        noise_similarity = torch.zeros(1) if clutter_feats is None else clutter_feats @ self.memory_pos.T
        picked = vert_feats.view(-1) if vis_mask is None else vert_feats[vis_mask]
        similarity = picked @ self.memory.T
        with torch.no_grad():
            target = self._bank_target(vert_feats if x_to_bank is None else x_to_bank, visible, object_labels)
            blended = self.momentum * self.memory_pos + (1.0 - self.momentum) * target
            self.memory_pos = F.normalize(blended, p=2, dim=1)
            if clutter_feats is not None:
                self._store_clutter(clutter_feats)
            self.lru = (self.lru + x.shape[0]) % self.max_lru
        return similarity, noise_similarity
