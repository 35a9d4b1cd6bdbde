"""Minimal training-step driver with the reference's calling protocol (engine.py:55-151): two model calls
(encode_and_save=True then False), keep-index gather of the annotated frames, time mask, criterion, weighted sum,
backward.  bench.py, smoke() and the parity tests drive the model through this exactly like engine.py would."""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from .util.misc import LRUCache, NestedTensor

_IDX = LRUCache()
_FUSED_CRITERION = [__import__("os").environ.get("TD_FUSED_CRITERION", "1") != "0"]  # 0: the stacked torch criterion (SetCriterion.forward)


class FixedTokenizer:
    """Tokenizer stand-in for synthetic clips: returns preset ids (there are no tokenizer files offline)."""

    def __init__(self, input_ids: torch.Tensor, attention_mask: torch.Tensor):
        self.ids, self.att = input_ids, attention_mask

    def batch_encode_plus(self, text, padding="longest", return_tensors="pt"):
        from transformers import BatchEncoding

        be = BatchEncoding({"input_ids": self.ids.clone(), "attention_mask": self.att.clone()})
        be._encodings = [None] * len(text)
        return be


def batch_to(batch: dict, device) -> dict:
    out = {}
    for k, v in batch.items():
        out[k] = v.to(device, non_blocking=True) if torch.is_tensor(v) else v
    return out


def forward_step(model, criterion, weight_dict: Dict[str, float], batch: dict):
    """batch: frames (n_slow,3,H,W), frames_mask, frames_fast / fast_mask (or None), durations, target_boxes (sum dur, 4),
    inter_idx; captions come from the model's tokenizer (set a FixedTokenizer for synthetic ids)."""
    durations: List[int] = batch["durations"]
    samples = NestedTensor(batch["frames"], batch["frames_mask"])
    samples_fast = NestedTensor(batch["frames_fast"], batch["fast_mask"]) if batch.get("frames_fast") is not None else None
    captions = batch.get("captions") or ["synthetic caption"] * len(durations)
    memory_cache = model(samples, durations, captions, encode_and_save=True, samples_fast=samples_fast)
    outputs = model(samples, durations, captions, encode_and_save=False, memory_cache=memory_cache)
    raw = dict(outputs)
    raw["aux_outputs"] = [dict(a) for a in outputs.get("aux_outputs", [])]

    t = max(durations)
    dev = outputs["pred_boxes"].device
    key = (tuple(durations), tuple(map(tuple, batch["inter_idx"])), str(dev))
    hit = _IDX.get(key)
    if hit is None:  # built once per (durations, inter_idx): H2D copies inside the step would synchronise the stream
        keep = []
        for i, (_d, inter) in enumerate(zip(durations, batch["inter_idx"])):
            keep.extend(range(i * t + inter[0], i * t + inter[1] + 1))
        time_mask = torch.zeros(len(durations), t, dtype=torch.bool)
        for i, d in enumerate(durations):
            time_mask[i, :d] = True
        hit = _IDX[key] = (torch.tensor(keep, dtype=torch.long, device=dev), time_mask.to(dev))
    keep, time_mask = hit
    targets = batch["target_boxes"]  # (n_annotated_frames, 4); the criterion also accepts the reference's list of dicts
    core = getattr(model, "module", model)  # DDP wrapper
    st = getattr(core, "_last_stacked", None)
    if (_FUSED_CRITERION[0] and st is not None and hasattr(criterion, "forward_fused") and "pred_sted" in outputs and st["pred_sted"] is not None
            and st["weights"] is not None and tuple(st["weights"].shape[1:]) == (len(durations), t, t) and torch.is_tensor(targets)):
        # one launch for the keep-gather + all 24 losses, one multiply + sum for the weighted total (engine.py:83-126)
        core._last_stacked = None
        assert len(targets) == keep.numel()
        loss_dict = criterion.forward_fused(st, keep, targets, batch["inter_idx"], time_mask, aux=bool(getattr(core, "aux_loss", True)))
        loss = (criterion.last_loss_matrix * criterion.weight_matrix(weight_dict, st["pred_boxes"].shape[0], dev)).sum()
        return loss, loss_dict, raw, memory_cache
    outputs["pred_boxes"] = outputs["pred_boxes"][keep]
    for a in outputs.get("aux_outputs", []):
        a["pred_boxes"] = a["pred_boxes"][keep]
    if "pred_sted" not in outputs:
        time_mask = None
    assert len(targets) == len(outputs["pred_boxes"])
    loss_dict = criterion(outputs, targets, batch["inter_idx"], time_mask)
    loss = sum(loss_dict[k] * weight_dict[k] for k in loss_dict if k in weight_dict)
    return loss, loss_dict, raw, memory_cache


def set_split_backward(model, on: bool) -> None:
    """Cut the step's autograd graph at the ResNet trunk's output (see backward_in_stages)."""
    core = getattr(model, "module", model)
    core.backbone[0].body.split_backward = bool(on)


def backward_in_stages(model, loss, after_first_stage=None, after_trunk_stage=None) -> None:
    """loss.backward() in two stages split at the trunk boundary (set_split_backward(model, True) before the forward):
    stage 1 = heads, decoder, encoder, text encoder, input_proj - 0.57 GB of the 0.74 GB of gradients are final when it
    ends; ``after_first_stage()`` runs there (the data-parallel reducer starts their all-reduce); stage 2 = the trunk's
    backward, which the collective overlaps.  Numerically identical to a single loss.backward().
    ``after_trunk_stage(k, weights)`` (optional): the trunk itself is issued stage by stage (layer4, layer3, layer2: k = 1, 2, 3 in the
    reducer's ``late_groups`` numbering), each stage's weight gradients in a launch of their own, and the callback runs behind each -
    the trunk's own 0.17 GB then leave in three pieces under the remaining backward instead of behind it."""
    loss.backward()
    if after_first_stage is not None:
        after_first_stage()
    core = getattr(model, "module", model)
    body = core.backbone[0].body
    if after_trunk_stage is None:
        body.backward_trunk()
    else:
        body.backward_trunk(after_stage=lambda stage, ws_: after_trunk_stage(4 - stage, ws_))


def trunk_stage_groups(model):
    """The trunk's trainable weights grouped by stage in backward order (layer4, layer3, layer2): ``late_groups`` of FlatGradAllReducer."""
    core = getattr(model, "module", model)
    body = core.backbone[0].body
    groups = {}
    for name, blk in body.blocks():
        for p in blk.parameters():
            if p.requires_grad:
                groups.setdefault(int(name[5]), []).append(p)
    return [groups[k] for k in sorted(groups, reverse=True)]


def train_step(model, criterion, weight_dict, batch, optimizer: Optional[torch.optim.Optimizer] = None, max_norm: float = 0.0):
    loss, loss_dict, _, _ = forward_step(model, criterion, weight_dict, batch)
    if optimizer is not None:
        optimizer.zero_grad(set_to_none=True)
    loss.backward()
    if optimizer is not None:
        if max_norm > 0:
            torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
        optimizer.step()
    return loss, loss_dict
