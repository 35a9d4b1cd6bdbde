"""tubedetr_amd - MI355X-native (gfx950) engine for TubeDETR's video-text encoder + space-time decoder hot path.

Drop-in for the reference's ``models`` package: ``tubedetr_amd.models.build_model(args)``.  See DESIGN.md."""
from argparse import Namespace

import torch

__all__ = ["default_args", "models"]


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
