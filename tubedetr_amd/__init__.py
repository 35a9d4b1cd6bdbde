"""tubedetr_amd - MI355X-native (gfx950) engine for TubeDETR's video-text encoder + space-time decoder hot path.

Drop-in for the reference's ``models`` package: ``tubedetr_amd.models.build_model(args)``.  See DESIGN.md."""
import os
from argparse import Namespace

import torch

__all__ = ["default_args", "models", "set_deterministic", "is_deterministic"]


_TORCH_DET_BEFORE = None  # torch's own (enabled, warn_only) state at the time this package switched it on


def set_deterministic(on: bool = True) -> None:
    """Run-to-run bit-reproducible steps (the exact-fp32 parity mode's regression anchor; TD_DETERMINISTIC=1 in the environment seeds the
    library's flag at its first launch).  The library's reductions that are normally split over workgroups and combined with fp32 atomics -
    weight gradients over the rows, LayerNorm's dgamma / dbeta, bias column sums - run as one sequential reduction per output element
    (``td_set_deterministic``: one atomic flag read at launch time), and torch's own index / embedding backward kernels are switched to
    their deterministic forms (``warn_only``: a torch op WITHOUT a deterministic implementation only warns - the bit-reproducibility claim
    is made for the paths the tests exercise).  Switching off restores the torch setting found when it was switched on.  A step already
    captured in a HIP graph keeps the grids it was captured with: re-capture after switching.  Slow: a 12 100-row weight gradient is then
    reduced by one workgroup per 128 x 128 tile."""
    global _TORCH_DET_BEFORE
    from . import _hip

    _hip.check(_hip.lib().td_set_deterministic(1 if on else 0), "td_set_deterministic")
    if on:
        if _TORCH_DET_BEFORE is None:
            _TORCH_DET_BEFORE = (torch.are_deterministic_algorithms_enabled(), torch.is_deterministic_algorithms_warn_only_enabled())
        torch.use_deterministic_algorithms(True, warn_only=True)
    elif _TORCH_DET_BEFORE is not None:
        torch.use_deterministic_algorithms(_TORCH_DET_BEFORE[0], warn_only=_TORCH_DET_BEFORE[1])
        _TORCH_DET_BEFORE = None


def is_deterministic() -> bool:
    from . import _hip

    return bool(_hip.lib().td_get_deterministic())


def default_args(**overrides) -> Namespace:
    """The hot-path subset of main.py's argparse defaults (main.py:32-337)."""
    a = dict(
        device="cuda", hidden_dim=256, dropout=0.1, nheads=8, dim_feedforward=2048, enc_layers=6, dec_layers=6,
        pass_pos_and_query=True, text_encoder_type="roberta-base", freeze_text_encoder=False, video_max_len_train=200,
        stride=5, no_tsa=False, guided_attn=True, fast=True, fast_mode="", learn_time_embed=False, rd_init_tsa=False,
        no_time_embed=False, position_embedding="sine", lr_backbone=1e-5, backbone="resnet101", dilation=False,
        freeze_backbone=False, num_queries=1, aux_loss=True, sted=True, sigma=1, bbox_loss_coef=5, giou_loss_coef=2,
        sted_loss_coef=10, guided_attn_loss_coef=1, compute_dtype=torch.float32,
    )
    a.update(overrides)
    return Namespace(**a)
