"""FeatureBankNeMo - the vertex / clutter feature memory (reference nemo/models/feature_banks.py:133-232).

At inference the bank is a data container: `memory` = cat(memory_pos [V_total, C], memory_neg
[num_noise * max_groups, C]); a multi-class bank is the concatenation of the per-mesh vertex blocks
(offsets = exclusive cumulative sum of mesh_n_list).  The constructor signature and attribute names are
the reference's.  forward() is the training-time similarity + momentum update (:164-228): the matrix
products and the momentum blend are torch calls as in the reference, the multi-class gather of the vertex
features into bank order (`object_labels` given) runs on the HIP replacement of the reference's CuNeMo
extension (cu_nemo.gather_features).  MaskCreater (:39-130) builds the loss-weight matrices; its
multi-object branch uses cu_nemo.get_mask.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .cu_nemo import gather_features, get_mask


class MaskCreater:
    """Loss-weight matrices of the vertex / clutter contrastive loss (reference feature_banks.py:39-130):
    per selected vertex slot a row over [all bank vertices | n_noise clutter rows] holding kappa_pos on the
    vertex itself, kappa_near on vertices closer than dist_thr, kappa_class on other meshes' vertices and
    kappa_clutter on the clutter rows."""

    def __init__(self, dist_thr, kappas, n_noise, verts_ori=None, device="cpu"):
        if verts_ori is not None:
            assert isinstance(verts_ori, list)
            with torch.no_grad():
                vmax = max(v.shape[0] for v in verts_ori)
                self.vert_sum_num = sum(v.shape[0] for v in verts_ori)
                dists = []
                for vv in verts_ori:
                    padded = torch.cat([vv, torch.zeros((vmax - vv.shape[0], 3), dtype=vv.dtype, device=vv.device)], dim=0)
                    dists.append((padded[:, None] - padded[None]).pow(2).sum(-1).pow(.5))
                vert_dists = torch.stack(dists)
            self.verts_dist_weight = (vert_dists <= dist_thr).to(device).type(torch.float32) * kappas["near"] + \
                torch.eye(vmax).to(device).type(torch.float32).unsqueeze(dim=0) * (kappas["pos"] - kappas["near"])
            self.mesh_n_list = [v.shape[0] for v in verts_ori]
        else:
            self.verts_dist_weight = None
        self.kappas = {"pos": 0, "near": 1e5, "clutter": 5, "class": 1e5}
        self.kappas.update(kappas)
        self.dist_thr, self.n_noise, self.device = dist_thr, n_noise, device

    def __call__(self, sample_indexs=None, K_padded=None, vis_mask=None, kps=None, dtype_template=None):
        if kps is not None:
            vert_dists = (kps[:, None] - kps[:, :, None]).pow(2).sum(-1).pow(.5)
            verts_dist_weight = (vert_dists <= self.dist_thr).to(self.device).type(torch.float32) * self.kappas["near"] + \
                torch.eye(kps.shape[1]).to(self.device).type(torch.float32).unsqueeze(dim=0) * (self.kappas["pos"] - self.kappas["near"])
        else:
            verts_dist_weight = self.verts_dist_weight
        if sample_indexs is None:                       # one object per image: pose loss only
            if verts_dist_weight.dim() == 2:
                verts_dist_weight = verts_dist_weight[None]
            get = torch.cat([verts_dist_weight, torch.ones(verts_dist_weight.shape[0:2] + (self.n_noise,),
                                                           device=verts_dist_weight.device) * self.kappas["clutter"]], dim=2)
            if get.shape[0] == 1:
                assert dtype_template is not None
                b_ = dtype_template.shape[0]
                get = get.expand(b_, -1, -1).contiguous()
            else:
                b_ = get.shape[0]
            vert_index = torch.arange(verts_dist_weight.shape[1], device=self.device)[None].long().expand(b_, -1).contiguous()
            if vis_mask is not None:
                return get[vis_mask], vert_index[vis_mask]
            return get.view(-1), vert_index.view(-1)
        assert kps is None, "several objects per image: only the mesh-distance weights are supported"
        if sample_indexs.dim() == 1:
            sample_indexs = sample_indexs[:, None]
        if K_padded is None:
            assert dtype_template is not None
            K_padded = dtype_template.shape[1]
        total_size = self.vert_sum_num + self.n_noise
        return get_mask(verts_dist_weight, sample_indexs, self.mesh_n_list, total_size, K_padded, self.kappas["class"],
                        self.kappas["clutter"], mask_sel=vis_mask, n_noise=self.n_noise)


class FeatureBankNeMo(nn.Module):
    def __init__(self, input_size, num_pos, num_noise=-1, max_groups=-1, momentum=0.5, **kwargs):
        super().__init__()
        stdv = 1.0 / math.sqrt(input_size / 3)
        self.memory_pos = torch.rand(num_pos, input_size).mul_(2 * stdv).add_(-stdv)
        self.memory_neg = torch.rand(num_noise * max_groups if max_groups > 0 else 0, input_size).mul_(2 * stdv).add_(-stdv)
        self.memory_pos.requires_grad = False
        self.memory_neg.requires_grad = False
        self.num_noise = num_noise
        self.lru = 0
        self.max_lru = max_groups if max_groups > 0 else -1
        self.kwargs = kwargs
        self.momentum = momentum

    @property
    def memory(self):
        return torch.cat((self.memory_pos, self.memory_neg), dim=0)

    def mesh_offsets(self):
        """Start row of every mesh's vertex block in memory_pos (multi-class layout)."""
        n_list = self.kwargs.get("mesh_n_list")
        if n_list is None:
            return [0]
        out, acc = [], 0
        for n in n_list:
            out.append(acc)
            acc += int(n)
        return out

    def vertex_block(self, mesh_index=0):
        """memory rows of one mesh ([V_mesh, C])."""
        n_list = self.kwargs.get("mesh_n_list")
        if n_list is None:
            return self.memory_pos
        lo = self.mesh_offsets()[mesh_index]
        return self.memory_pos[lo:lo + int(n_list[mesh_index])]

    def load_memory(self, memory):
        """Fill the bank from a checkpoint's `memory` tensor (rows [0:V] vertices, the rest clutter)."""
        v = self.memory_pos.shape[0]
        self.memory_pos = memory[:v].detach().clone().to(self.memory_pos.device)
        self.memory_neg = memory[v:].detach().clone().to(self.memory_pos.device)
        return self

    def to(self, device):
        self.memory_pos = self.memory_pos.to(device)
        self.memory_neg = self.memory_neg.to(device)
        return self

    def cuda(self, device=None):
        return self.to("cuda" if device is None else device)

    def forward(self, x, visible, x_to_bank=None, object_labels=None, vis_mask=None):
        if self.num_noise == 0:
            t_, noise_similarity = x, torch.zeros(1)
        else:
            t_ = x[:, 0:(x.shape[1] - self.num_noise), :]
            noise_similarity = torch.matmul(x[:, -self.num_noise:, :], self.memory_pos.t())
        similarity = torch.matmul(t_.view(-1) if vis_mask is None else t_[vis_mask], self.memory.t())
        with torch.no_grad():
            if x_to_bank is None:
                x_to_bank = x[:, 0:(x.shape[1] - self.num_noise), :]
            b_ = x_to_bank.shape[0]
            if x_to_bank.dim() == 3:
                if object_labels is None:
                    x_to_bank = torch.mean(x_to_bank * visible.type(x.dtype)[..., None], dim=0)
                else:       # several meshes in the batch: visible vertex features summed into bank order
                    x_to_bank, _ = gather_features(x_to_bank, weights=visible.type(torch.float32),
                                                   sample_indexs=object_labels.type(torch.int32),
                                                   mesh_n_list=self.kwargs.get("mesh_n_list"))
                    x_to_bank = x_to_bank / b_
            self.memory_pos = F.normalize(self.memory_pos * self.momentum + x_to_bank * (1 - self.momentum), dim=1, p=2)
            if self.num_noise > 0:
                new = x[:, -self.num_noise:, :].contiguous().view(-1, x.shape[2])
                if x.shape[0] * self.num_noise > self.memory_neg.shape[0]:
                    self.memory_neg = new[0:self.memory_neg.shape[0]]
                else:
                    self.memory_neg = torch.cat([self.memory_neg[0:self.lru * self.num_noise], new,
                                                 self.memory_neg[(self.lru + x.shape[0]) * self.num_noise:]],
                                                dim=0)[:self.max_lru * self.num_noise]
            self.lru = (self.lru + x.shape[0]) % self.max_lru
        return similarity, noise_similarity
