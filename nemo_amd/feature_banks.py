"""FeatureBankNeMo - the vertex / clutter feature memory (reference nemo/models/feature_banks.py:133-232).

At inference the bank is a data container: `memory` = cat(memory_pos [V_total, C], memory_neg
[num_noise * max_groups, C]); a multi-class bank is the concatenation of the per-mesh vertex blocks
(offsets = exclusive cumulative sum of mesh_n_list).  The constructor signature and attribute names are
the reference's.  forward() (the training-time similarity + momentum update) is provided for the
single-class case in plain torch; the multi-class gather needs the reference's CuNeMo extension
and is outside the solver path.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


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
        if object_labels is not None:
            raise NotImplementedError("the multi-class bank update needs the CuNeMo gather (training side, out of scope)")
        if self.num_noise == 0:
            t_, noise_similarity = x, torch.zeros(1)
        else:
            t_ = x[:, 0:(x.shape[1] - self.num_noise), :]
            noise_similarity = torch.matmul(x[:, -self.num_noise:, :], self.memory_pos.t())
        similarity = torch.matmul(t_.view(-1) if vis_mask is None else t_[vis_mask], self.memory.t())
        with torch.no_grad():
            if x_to_bank is None:
                x_to_bank = x[:, 0:(x.shape[1] - self.num_noise), :]
            if x_to_bank.dim() == 3:
                x_to_bank = torch.mean(x_to_bank * visible.type(x.dtype)[..., None], dim=0)
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
