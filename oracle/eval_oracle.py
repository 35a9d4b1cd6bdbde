"""TEST INFRASTRUCTURE: CPU restatement of the reference's evaluation-side post-processing (plain torch, fp32), pinned
against the reference's own outputs in tests/golden/eval_path.npz (oracle/gen_golden_eval.py).  Never imported by the
product package."""
from __future__ import annotations

from typing import List

import torch


def post_process_stvg(steds: torch.Tensor, frames_id: List[List[int]], video_ids: List, time_mask: torch.Tensor) -> List[List[float]]:
    """models/postprocessors.py:13-84."""
    video_ids = list(video_ids)
    if len(set(video_ids)) != len(video_ids):  # :24-53 consecutive windows of one video are concatenated along time
        lst = [steds[0].masked_fill(~time_mask[0][:, None], -float("inf"))]
        for i in range(1, len(video_ids)):
            cur = steds[i].masked_fill(~time_mask[i][:, None], -float("inf"))
            if video_ids[i] == video_ids[i - 1]:
                lst[-1] = torch.cat([lst[-1], cur], 0)
            else:
                lst.append(cur)
        mx = max(len(x) for x in lst)
        eff = torch.ones(len(set(video_ids)), mx, 2) * float("-inf")
        for i, x in enumerate(lst):
            eff[i, : len(x)] = x
        steds = eff
    T = steds.shape[1]
    mask = (torch.ones(T, T) * float("-inf")).tril(0).unsqueeze(0).expand(steds.shape[0], -1, -1)  # :55-61 end <= start impossible
    score = steds[:, :, 0].log_softmax(1).unsqueeze(2) + steds[:, :, 1].log_softmax(1).unsqueeze(1) + mask  # :62-67
    score, s_idx = score.max(dim=1)
    score, e_idx = score.max(dim=1)
    s_idx = torch.gather(s_idx, 1, e_idx.view(-1, 1)).squeeze(1)
    pred = torch.stack([s_idx, e_idx], 1)
    fid = torch.tensor([row + [0] * (T - len(row)) for row in frames_id]).long()  # :73-79
    pred = torch.gather(fid, 1, pred).float()
    pred[:, 1] += 1  # :80 the end frame is excluded in evaluation
    return pred.tolist()
