"""Sine position / time encodings behind the reference's ``models.position_encoding`` interface
(models/position_encoding.py:30-94,134-144).  The spatial encoding is a pure function of the pad mask and is
computed by one HIP kernel (td_pos_sine); the time table is a constant buffer (``te``, part of the state dict)."""
from __future__ import annotations

import math

import torch
from torch import nn

from .. import ops


class TimeEmbeddingSine(nn.Module):
    """te[p,0,2j] = sin(p * exp(-2j ln(1e4)/d)), te[p,0,2j+1] = cos(same)  (position_encoding.py:35-49)."""

    def __init__(self, max_len: int = 200, d_model: int = 512):
        super().__init__()
        self.max_len = max_len
        pos = torch.arange(max_len, dtype=torch.float32).unsqueeze(1)
        freq = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * (-math.log(10000.0) / d_model))
        te = torch.zeros(max_len, 1, d_model)
        te[:, 0, 0::2] = torch.sin(pos * freq)
        te[:, 0, 1::2] = torch.cos(pos * freq)
        self.register_buffer("te", te)

    def forward(self, ln: int) -> torch.Tensor:
        return self.te[:ln]


class TimeEmbeddingLearned(nn.Module):
    """--learn_time_embed (position_encoding.py:13-27): a learned table; an ablation outside the kernel scope (SURVEY.md
    8a'), served by the embedding weight itself - stock PyTorch autograd reaches it through the query-position rows."""

    def __init__(self, num_pos_feats: int = 200, d_model: int = 512):
        super().__init__()
        self.time_embed = nn.Embedding(num_pos_feats, d_model)
        nn.init.uniform_(self.time_embed.weight)

    def forward(self, ln: int) -> torch.Tensor:
        return self.time_embed.weight[:ln].unsqueeze(1)


class PositionEmbeddingLearned(nn.Module):
    """--position_embedding learned (position_encoding.py:96-131): row / column embedding tables, concatenated per pixel.
    Ablation: stock PyTorch ops (their autograd trains the tables); the result enters the HIP encoder like the sine one."""

    def __init__(self, num_pos_feats: int = 256):
        super().__init__()
        self.row_embed = nn.Embedding(50, num_pos_feats)
        self.col_embed = nn.Embedding(50, num_pos_feats)
        nn.init.uniform_(self.row_embed.weight)
        nn.init.uniform_(self.col_embed.weight)
        self.compute_dtype = torch.float32

    def forward(self, tensor_list) -> torch.Tensor:
        n, h, w = tensor_list.mask.shape
        dev = tensor_list.mask.device
        x_emb = self.col_embed(torch.arange(w, device=dev))
        y_emb = self.row_embed(torch.arange(h, device=dev))
        pos = torch.cat([x_emb.unsqueeze(0).repeat(h, 1, 1), y_emb.unsqueeze(1).repeat(1, w, 1)], dim=-1)  # (h, w, C)
        return pos.to(self.compute_dtype).unsqueeze(0).repeat(n, 1, 1, 1).permute(0, 3, 1, 2)  # (N, C, h, w), channels-last strides


class PositionEmbeddingSine(nn.Module):
    """forward(NestedTensor) -> (N, 2*num_pos_feats, h, w) like the reference; ``compute_dtype`` selects the
    element type (set by the owning backbone).  Internally the kernel writes token-major [N, h*w, C]; the returned
    tensor is the permuted view of it (channels-last strides, zero copy)."""

    def __init__(self, num_pos_feats: int = 64, temperature: float = 10000, normalize: bool = False, scale=None):
        super().__init__()
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        if not normalize or (scale is not None and abs(scale - 2 * math.pi) > 1e-12):
            raise NotImplementedError("only the reference's default (normalize=True, scale=2*pi) is implemented in HIP")
        self.num_pos_feats = num_pos_feats
        self.temperature = temperature
        self.normalize = normalize
        self.scale = 2 * math.pi
        self.compute_dtype = torch.float32

    def forward(self, tensor_list) -> torch.Tensor:
        mask = tensor_list.mask
        n, h, w = mask.shape
        pos = ops.pos_sine(mask, self.num_pos_feats, self.compute_dtype, float(self.temperature))  # [N, hw, C]
        return pos.view(n, h, w, -1).permute(0, 3, 1, 2)


def build_position_encoding(args):
    n_steps = args.hidden_dim // 2
    if args.position_embedding in ("v2", "sine"):
        return PositionEmbeddingSine(n_steps, normalize=True)
    if args.position_embedding in ("v3", "learned"):
        return PositionEmbeddingLearned(n_steps)
    raise ValueError(f"not supported {args.position_embedding}")
