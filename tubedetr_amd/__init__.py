"""tubedetr_amd - MI355X-native (gfx950) engine for TubeDETR's video-text encoder + space-time decoder hot path.

Drop-in for the reference's ``models`` package: ``tubedetr_amd.models.build_model(args)``.  See DESIGN.md."""
import os
from argparse import Namespace

import torch

__all__ = ["default_args", "models", "set_deterministic"]


def set_deterministic(on: bool = True) -> None:
    """Run-to-run bit-reproducible steps (the exact-fp32 parity mode's regression anchor; also settable as TD_DETERMINISTIC=1 in the
    environment).  The library's reductions that are normally split over workgroups and combined with fp32 atomics - weight gradients
    over the rows, LayerNorm's dgamma / dbeta, bias column sums - run as one sequential reduction per output element (the C side reads
    the variable at every call), and torch's own index / embedding backward kernels are switched to their deterministic forms.  Slow:
    a 12 100-row weight gradient is then reduced by one workgroup per 128 x 128 tile."""
    os.environ["TD_DETERMINISTIC"] = "1" if on else "0"
    torch.use_deterministic_algorithms(bool(on), warn_only=True)


if os.environ.get("TD_DETERMINISTIC") == "1":
    set_deterministic(True)


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
