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
    raise ValueError(f"not supported {args.position_embedding} (learned encodings are outside the HIP hot path)")
