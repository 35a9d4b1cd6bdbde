"""Evaluation-side post-processors behind the reference's ``models.postprocessors`` interface
(models/postprocessors.py:13-118): ``PostProcessSTVG`` (predicted start / end frame per video, with the ensembling of
several forward windows of one long video) and ``PostProcess`` (boxes to absolute xyxy).  Same call signatures and return
values as the reference's, so engine.evaluate (engine.py:292-340) drives them unchanged; the arithmetic runs on the
device: the constrained arg-max over (start, end) pairs is one HIP launch (td_sted_decode) instead of a (B, T, T) score
tensor and three torch reductions."""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
from torch import nn


def sted_decode(steds: torch.Tensor) -> torch.Tensor:
    """steds [n, T, 2] logits (-inf = impossible position) -> int64 [n, 2] (start index, end index), end > start."""
    from .. import _hip

    steds = steds.float().contiguous()
    n, T, _ = steds.shape
    out = torch.empty((n, 2), dtype=torch.int64, device=steds.device)
    _hip.check(_hip.lib().td_sted_decode(_hip.ptr(steds), _hip.ptr(out), n, T, _hip.stream_ptr()), "td_sted_decode")
    return out


class PostProcessSTVG(nn.Module):
    @torch.no_grad()
    def forward(self, outputs, frames_id=None, video_ids=None, time_mask=None):
        """outputs["pred_sted"]: [B, T, 2] logits; frames_id: B increasing lists of frame ids; video_ids: B ids (equal ids =
        consecutive windows of one video, ensembled); time_mask [B, T] False on padded positions.  Returns B' lists
        [start_frame, end_frame (exclusive)], one per distinct video, like the reference."""
        steds = outputs["pred_sted"]
        dev = steds.device
        video_ids = list(video_ids)
        if len(set(video_ids)) != len(video_ids):
            # consecutive windows of the same video: concatenate their (masked) logits along time (postprocessors.py:27-53)
            groups: List[List[int]] = [[0]]
            for i in range(1, len(video_ids)):
                if video_ids[i] == video_ids[i - 1]:
                    groups[-1].append(i)
                else:
                    groups.append([i])
            masked = steds.float().masked_fill(~time_mask[:, :, None], float("-inf"))
            T = steds.shape[1]
            max_dur = max(len(g) for g in groups) * T
            eff = torch.full((len(set(video_ids)), max_dur, 2), float("-inf"), device=dev)
            for i_v, g in enumerate(groups):
                eff[i_v, : len(g) * T] = masked[g].reshape(len(g) * T, 2)
            steds = eff
        pred = sted_decode(steds)  # [B', 2] indices
        max_length = steds.shape[1]
        fid = torch.tensor([list(row) + [0] * (max_length - len(row)) for row in frames_id], dtype=torch.long, device=dev)
        pred = torch.gather(fid, 1, pred).float()
        pred[:, 1] += 1  # the end frame is excluded in evaluation
        return pred.cpu().tolist()


class PostProcess(nn.Module):
    """Boxes cxcywh in [0, 1] -> absolute [x0, y0, x1, y1] (postprocessors.py:86-107)."""

    @torch.no_grad()
    def forward(self, outputs, target_sizes):
        b = outputs["pred_boxes"].float()
        cx, cy, w, h = b.unbind(-1)
        boxes = torch.stack((cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h), -1)
        img_h, img_w = target_sizes.unbind(1)
        boxes = boxes * torch.stack([img_w, img_h, img_w, img_h], dim=1)
        return [{"boxes": x} for x in boxes]


def build_postprocessors(args, dataset_name) -> Dict[str, nn.Module]:
    post: Dict[str, nn.Module] = {"bbox": PostProcess()}
    if dataset_name in ("vidstg", "hcstvg"):
        post[dataset_name] = PostProcessSTVG()
    return post
