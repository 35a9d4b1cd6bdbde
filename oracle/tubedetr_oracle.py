"""CPU oracle for the TubeDETR video-text encoder + space-time decoder hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``tubedetr_amd/`` may import this module; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it, and only as the
checker / the reported CPU baseline, never as the product path.

This is a *functional restatement* (plain PyTorch fp32 on CPU, explicit attention math, no nn.Module
tree) of the algorithm implemented by the reference files

  models/backbone.py:20-105,220-233      FrozenBatchNorm2d, BackboneBase.forward, Joiner.forward
  torchvision==0.9.1 resnet101           (third party, absent from /root/reference; architecture restated)
  models/position_encoding.py:30-94      TimeEmbeddingSine, PositionEmbeddingSine
  models/transformer.py:195-491          Transformer.forward (encode / decode branches)
  models/transformer.py:502-751,768-773  encoder / decoder layers, FeatureResizer
  torch==1.8.1 nn.MultiheadAttention     (third party; formula restated in ``mha``)
  models/tubedetr.py:23-42,93-254        MLP heads, TubeDETR.forward
  models/tubedetr.py:257-460             SetCriterion (loss harness), util/box_ops.py:53-115

It operates on a flat ``state_dict`` whose keys/shapes are exactly the reference's
(``backbone.0.body.layer2.0.conv1.weight`` ...), so weights are exchangeable with the reference
model and with the product modules.

Parity status: PINNED.  ``oracle/gen_golden.py`` imports the reference itself in the build
container (stubs for the absent third-party packages), runs it on seeded inputs/weights and stores
outputs under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks this restatement against
those vectors.  The third-party arithmetic (torchvision resnet101, torch MHA, HF RoBERTa) is not
pinned by any reference-side test (the reference has none) - it is pinned only through those runs
on torch 2.10 / transformers 5.15.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# (name, planes, n_blocks, stride) of torchvision resnet101's four stages
RESNET101_STAGES = (("layer1", 64, 3, 1), ("layer2", 128, 4, 2), ("layer3", 256, 23, 2), ("layer4", 512, 3, 2))


@dataclass
class OracleConfig:
    """The subset of main.py's argparse flags (main.py:32-337) that changes the hot path."""

    hidden_dim: int = 256
    nheads: int = 8
    enc_layers: int = 6
    dec_layers: int = 6
    dim_feedforward: int = 2048
    num_queries: int = 1
    stride: int = 5
    video_max_len_train: int = 200
    fast: bool = True
    no_tsa: bool = False
    no_time_embed: bool = False
    sted: bool = True
    guided_attn: bool = True
    aux_loss: bool = True
    sigma: float = 1.0
    bbox_loss_coef: float = 5.0
    giou_loss_coef: float = 2.0
    sted_loss_coef: float = 10.0
    guided_attn_loss_coef: float = 1.0
    stages: Sequence = field(default_factory=lambda: RESNET101_STAGES)


# --------------------------------------------------------------------------------------------
# backbone  (models/backbone.py + torchvision resnet101)
# --------------------------------------------------------------------------------------------
def frozen_bn(x: Tensor, sd: Dict[str, Tensor], p: str) -> Tensor:
    """backbone.py:60-70  y = x*scale + (b - rm*scale), scale = w*rsqrt(rv + 1e-5)."""
    scale = sd[p + "weight"] * (sd[p + "running_var"] + 1e-5).rsqrt()
    shift = sd[p + "bias"] - sd[p + "running_mean"] * scale
    return x * scale[None, :, None, None] + shift[None, :, None, None]


def bottleneck(x: Tensor, sd, p: str, stride: int) -> Tensor:
    """torchvision Bottleneck v1.5 (stride on the 3x3), bias-free convs, FrozenBN after each."""
    idt = x
    y = F.relu(frozen_bn(F.conv2d(x, sd[p + "conv1.weight"]), sd, p + "bn1."))
    y = F.relu(frozen_bn(F.conv2d(y, sd[p + "conv2.weight"], stride=stride, padding=1), sd, p + "bn2."))
    y = frozen_bn(F.conv2d(y, sd[p + "conv3.weight"]), sd, p + "bn3.")
    if (p + "downsample.0.weight") in sd:
        idt = frozen_bn(F.conv2d(x, sd[p + "downsample.0.weight"], stride=stride), sd, p + "downsample.1.")
    return F.relu(y + idt)


def resnet_trunk(x: Tensor, sd, prefix: str = "backbone.0.body.", stages=RESNET101_STAGES, taps=None) -> Tensor:
    """conv1 7x7/2 -> FrozenBN -> ReLU -> maxpool 3x3/2 -> Bottleneck x [3,4,23,3]  => layer4 output."""
    y = F.relu(frozen_bn(F.conv2d(x, sd[prefix + "conv1.weight"], stride=2, padding=3), sd, prefix + "bn1."))
    y = F.max_pool2d(y, kernel_size=3, stride=2, padding=1)
    if taps is not None:
        taps["stem"] = y
    for name, _planes, nblocks, stride in stages:
        for i in range(nblocks):
            y = bottleneck(y, sd, f"{prefix}{name}.{i}.", stride if i == 0 else 1)
        if taps is not None:
            taps[name] = y
    return y


def nearest_index(out_size: int, in_size: int) -> Tensor:
    """Index table of F.interpolate(mode='nearest'): floor(dst * float32(in/out)), clamped.

    backbone.py:101-103 downsamples the pad mask this way; the product computes the same table."""
    scale = torch.tensor(in_size / out_size, dtype=torch.float32)
    idx = torch.floor(torch.arange(out_size, dtype=torch.float32) * scale).long()
    return idx.clamp_(max=in_size - 1)


def downsample_mask(mask: Tensor, h: int, w: int) -> Tensor:
    iy = nearest_index(h, mask.shape[-2])
    ix = nearest_index(w, mask.shape[-1])
    return mask[:, iy][:, :, ix]


def pos_sine(mask: Tensor, num_pos_feats: int = 128, temperature: float = 10000.0) -> Tensor:
    """position_encoding.py:71-94 with normalize=True, scale=2*pi.  mask (N,h,w) bool -> (N,2F,h,w)."""
    not_mask = (~mask).to(torch.float32)
    y_embed = not_mask.cumsum(1)
    x_embed = not_mask.cumsum(2)
    eps, scale = 1e-6, 2 * math.pi
    y_embed = y_embed / (y_embed[:, -1:, :] + eps) * scale
    x_embed = x_embed / (x_embed[:, :, -1:] + eps) * scale
    i = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / num_pos_feats)
    px = x_embed[..., None] / dim_t
    py = y_embed[..., None] / dim_t
    even = (torch.arange(num_pos_feats) % 2) == 0
    px = torch.where(even, px.sin(), px.cos())
    py = torch.where(even, py.sin(), py.cos())
    return torch.cat((py, px), dim=3).permute(0, 3, 1, 2)


def time_sine(max_len: int, d_model: int) -> Tensor:
    """position_encoding.py:35-44  te[p,0,2j]=sin(p*exp(-2j ln1e4/d)), te[p,0,2j+1]=cos(same)."""
    position = torch.arange(max_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
    te = torch.zeros(max_len, 1, d_model)
    te[:, 0, 0::2] = torch.sin(position * div_term)
    te[:, 0, 1::2] = torch.cos(position * div_term)
    return te


def backbone_joiner(frames: Tensor, mask: Tensor, sd, cfg: OracleConfig):
    """Joiner.forward (backbone.py:224-233): layer4 features, downsampled mask, sine pos-enc."""
    feat = resnet_trunk(frames, sd, stages=cfg.stages)
    m = downsample_mask(mask, feat.shape[-2], feat.shape[-1])
    pos = pos_sine(m, cfg.hidden_dim // 2).to(feat.dtype)
    return feat, m, pos


# --------------------------------------------------------------------------------------------
# transformer pieces (models/transformer.py, torch MHA formula)
# --------------------------------------------------------------------------------------------
def mha(q_in: Tensor, k_in: Tensor, v_in: Tensor, sd, p: str, nheads: int, key_padding_mask: Optional[Tensor]):
    """nn.MultiheadAttention forward in eval mode (dropout off), sequence-first tensors.

    q_in (Lq,B,E), k_in/v_in (Lk,B,E); packed in_proj (3E,E); q scaled by 1/sqrt(E/H); padded keys
    get -inf; returns (out (Lq,B,E), weights averaged over heads (B,Lq,Lk))."""
    Lq, B, E = q_in.shape
    Lk = k_in.shape[0]
    hd = E // nheads
    w, bias = sd[p + "in_proj_weight"], sd[p + "in_proj_bias"]
    q = q_in @ w[:E].t() + bias[:E]
    k = k_in @ w[E : 2 * E].t() + bias[E : 2 * E]
    v = v_in @ w[2 * E :].t() + bias[2 * E :]
    q = q.reshape(Lq, B * nheads, hd).transpose(0, 1) * (1.0 / math.sqrt(hd))
    k = k.reshape(Lk, B * nheads, hd).transpose(0, 1)
    v = v.reshape(Lk, B * nheads, hd).transpose(0, 1)
    scores = torch.bmm(q, k.transpose(1, 2))  # (B*H, Lq, Lk)
    if key_padding_mask is not None:
        scores = scores.view(B, nheads, Lq, Lk).masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
        scores = scores.view(B * nheads, Lq, Lk)
    probs = scores.softmax(dim=-1)
    ctx = torch.bmm(probs, v).transpose(0, 1).reshape(Lq, B, E)
    out = ctx @ sd[p + "out_proj.weight"].t() + sd[p + "out_proj.bias"]
    return out, probs.view(B, nheads, Lq, Lk).mean(dim=1)


def layer_norm(x: Tensor, sd, p: str, eps: float = 1e-5) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + "weight"], sd[p + "bias"], eps)


def ffn(x: Tensor, sd, p: str) -> Tensor:
    h = F.relu(x @ sd[p + "linear1.weight"].t() + sd[p + "linear1.bias"])
    return h @ sd[p + "linear2.weight"].t() + sd[p + "linear2.bias"]


def encoder_layer(src: Tensor, pos: Tensor, kpm: Tensor, sd, p: str, nheads: int) -> Tensor:
    """transformer.py:629-646 (post-norm; pos added to q,k only)."""
    qk = src + pos
    a, _ = mha(qk, qk, src, sd, p + "self_attn.", nheads, kpm)
    src = layer_norm(src + a, sd, p + "norm1.")
    return layer_norm(src + ffn(src, sd, p), sd, p + "norm2.")


def decoder_layer(tgt, memory, pos, query_pos, query_mask, memory_mask, sd, p: str, nheads: int, no_tsa: bool):
    """transformer.py:684-751: temporal self-attention, time-aligned cross-attention, FFN."""
    t, b, f = tgt.shape
    bs = memory.shape[1]
    qk = tgt + query_pos
    if no_tsa:  # transformer.py:701-711, sequence length 1 => softmax == 1
        r = lambda x: x.transpose(0, 1).reshape(bs * b, -1, f).transpose(0, 1)
        a, w = mha(r(qk), r(qk), r(tgt), sd, p + "self_attn.", nheads, None)
        a = a.reshape(b, t, f).transpose(0, 1)
    else:
        a, w = mha(qk, qk, tgt, sd, p + "self_attn.", nheads, query_mask)
    tgt = layer_norm(tgt + a, sd, p + "norm1.")
    # (t,b,f) -> (1, b*t, f) video-major frame axis
    tc = tgt.transpose(0, 1).reshape(bs, -1, f).transpose(0, 1)
    qc = query_pos.transpose(0, 1).reshape(bs, -1, f).transpose(0, 1)
    a, cw = mha(tc + qc, memory + pos, memory, sd, p + "cross_attn_image.", nheads, memory_mask)
    a = a.reshape(b, t, f).transpose(0, 1)
    tgt = layer_norm(tgt + a, sd, p + "norm3.")
    tgt = layer_norm(tgt + ffn(tgt, sd, p), sd, p + "norm4.")
    return tgt, w, cw


def mlp(x: Tensor, sd, p: str, nlayers: int) -> Tensor:
    """tubedetr.py:37-42 in eval mode (dropout off)."""
    for i in range(nlayers):
        x = x @ sd[f"{p}layers.{i}.weight"].t() + sd[f"{p}layers.{i}.bias"]
        if i < nlayers - 1:
            x = F.relu(x)
    return x


_ROBERTA_CACHE: dict = {}


def roberta_last_hidden(sd, input_ids: Tensor, attention_mask: Tensor, prefix: str = "transformer.text_encoder.") -> Tensor:
    """HF RobertaModel (third party, stays a library call on both sides): roberta-base geometry."""
    from transformers import RobertaConfig, RobertaModel

    if "m" not in _ROBERTA_CACHE:
        cfg = RobertaConfig(vocab_size=50265, max_position_embeddings=514, type_vocab_size=1, pad_token_id=1, layer_norm_eps=1e-5)
        _ROBERTA_CACHE["m"] = RobertaModel(cfg).eval()
    m = _ROBERTA_CACHE["m"]
    sub = {k[len(prefix) :]: v for k, v in sd.items() if k.startswith(prefix)}
    if any(v.requires_grad for v in sub.values()):
        out = torch.func.functional_call(m, sub, args=(), kwargs=dict(input_ids=input_ids, attention_mask=attention_mask), strict=False)
    else:
        m.load_state_dict(sub, strict=False)
        out = m(input_ids=input_ids, attention_mask=attention_mask)
    return out.last_hidden_state


# --------------------------------------------------------------------------------------------
# the two model calls (models/tubedetr.py:117-254 + models/transformer.py:195-491)
# --------------------------------------------------------------------------------------------
def encode(sd, cfg: OracleConfig, frames: Tensor, frames_mask: Tensor, durations: List[int], input_ids: Tensor,
           attention_mask: Tensor, frames_fast: Optional[Tensor] = None, fast_mask: Optional[Tensor] = None,
           taps: Optional[dict] = None) -> dict:
    """model(samples, durations, captions, encode_and_save=True, samples_fast=...) -> memory_cache."""
    assert cfg.stride > 0, "oracle restates the stride>0 (temporal sampling) path only"
    d, k = cfg.hidden_dim, cfg.stride
    b, t = len(durations), max(durations)
    n_clips = math.ceil(t / k)
    feat, mask, pos = backbone_joiner(frames, frames_mask, sd, cfg)
    wp, bp = sd["input_proj.weight"], sd["input_proj.bias"]
    src = F.conv2d(feat, wp, bp)
    _, f, h, w = src.shape
    hw = h * w
    assert src.shape[0] == b * n_clips
    if taps is not None:
        taps["feat"], taps["src"], taps["pos"] = feat, src, pos

    tpad_mask_t = torch.ones(b, t, h, w, dtype=torch.bool)
    fast_src = None
    if cfg.fast:
        with torch.no_grad():
            feat_f, mask_f, _ = backbone_joiner(frames_fast, fast_mask, sd, cfg)
        src_f = F.conv2d(feat_f, wp, bp)
        fast_src = torch.zeros(b, t, f, h, w)
        cum = 0
        for i, dur in enumerate(durations):
            fast_src[i, :dur] = src_f[cum : cum + dur]
            tpad_mask_t[i, :dur] = mask_f[cum : cum + dur]
            cum += dur
        fast_src = fast_src.view(b * t, f, h, w)
    else:  # tubedetr.py:172-178: frame masks are the owning slow clip's mask
        clip = 0
        for i, dur in enumerate(durations):
            cur = 0
            for c in range(math.ceil(dur / k)):
                cd = min(k, dur - c * k)
                tpad_mask_t[i, cur : cur + cd] = mask[clip : clip + 1].repeat(cd, 1, 1)
                cur += cd
                clip += 1
    mask = mask.clone()
    mask[:, 0, 0] = False
    tpad_mask_t = tpad_mask_t.view(b * t, h, w)
    tpad_mask_t[:, 0, 0] = False

    # ---- Transformer.forward(encode_and_save=True) ----
    src = src.flatten(2).permute(2, 0, 1)  # (hw, n, d)
    pos = pos.flatten(2).permute(2, 0, 1)
    mask = mask.flatten(1)
    query_embed = sd["query_embed.weight"][:1].unsqueeze(1).repeat(1, b * t, 1).view(t, b, d)
    if not cfg.no_time_embed:
        query_embed = query_embed + sd["transformer.time_embed.te"][:t].repeat(1, b, 1)
    query_mask = torch.ones(b, t, dtype=torch.bool)
    query_mask[:, 0] = False
    for i, dur in enumerate(durations):
        query_mask[i, :dur] = False

    text_hidden = roberta_last_hidden(sd, input_ids, attention_mask).transpose(0, 1)  # (L,B,768)
    text_attention_mask = attention_mask.ne(1)
    x = text_hidden @ sd["transformer.resizer.fc.weight"].t() + sd["transformer.resizer.fc.bias"]
    text_resized = layer_norm(x, sd, "transformer.resizer.layer_norm.", eps=1e-12)  # (L,B,d)
    L = text_resized.shape[0]
    rep = torch.arange(b).repeat_interleave(n_clips)
    text_mask_clip = text_attention_mask[rep]  # (b*n_clips, L)
    text_clip = text_resized[:, rep]  # (L, b*n_clips, d)

    src = torch.cat([src, text_clip], 0)
    mask = torch.cat([mask, text_mask_clip], 1)
    pos = torch.cat([pos, torch.zeros_like(text_clip)], 0)
    frame_mask = torch.cat([tpad_mask_t.flatten(1), text_attention_mask[torch.arange(b).repeat_interleave(t)]], 1)

    mem = src
    for l in range(cfg.enc_layers):
        mem = encoder_layer(mem, pos, mask, sd, f"transformer.encoder.layers.{l}.", cfg.nheads)
        if taps is not None:
            taps[f"enc{l}"] = mem

    # temporal replication (transformer.py:393-427): frame j of video i owned by clip i*n_clips + j//k
    owner = (torch.arange(b)[:, None] * n_clips + torch.arange(t)[None, :] // k).reshape(-1)
    img_memory = mem[:, owner]
    pos_embed = pos[:, owner]
    frame_mask[:, 0] = False
    if cfg.fast:  # transformer.py:373-375,387,441-445
        fast_mem = fast_src.flatten(2).permute(2, 0, 1) @ sd["transformer.fast_encoder.weight"].t() + sd["transformer.fast_encoder.bias"]
        vis = img_memory[:hw]
        agg = (vis + fast_mem) @ sd["transformer.fast_residual.weight"].t() + sd["transformer.fast_residual.bias"]
        img_memory = torch.cat([vis + agg, img_memory[hw:]], 0)
    return {
        "text_memory_resized": text_clip,
        "text_memory": img_memory[-L:],
        "text_attention_mask": text_mask_clip,
        "img_memory": img_memory,
        "mask": frame_mask,
        "pos_embed": pos_embed,
        "query_embed": query_embed,
        "query_mask": query_mask,
    }


def decode(sd, cfg: OracleConfig, cache: dict) -> dict:
    """model(..., encode_and_save=False, memory_cache=cache) -> outputs dict (tubedetr.py:204-254)."""
    query_pos = cache["query_embed"]
    tgt = torch.zeros_like(query_pos)
    t, b, d = tgt.shape
    hs, ws, cws = [], [], []
    for l in range(cfg.dec_layers):
        tgt, w, cw = decoder_layer(tgt, cache["img_memory"], cache["pos_embed"], query_pos, cache["query_mask"],
                                   cache["mask"], sd, f"transformer.decoder.layers.{l}.", cfg.nheads, cfg.no_tsa)
        hs.append(layer_norm(tgt, sd, "transformer.decoder.norm."))
        ws.append(w)
        cws.append(cw)
    hs = torch.stack(hs).transpose(1, 2)  # (layers, b, t, d)
    out = {"hs": hs}
    boxes = mlp(hs.flatten(1, 2), sd, "bbox_embed.", 3).sigmoid()
    out["pred_boxes"] = boxes[-1]
    if cfg.sted:
        sted = mlp(hs, sd, "sted_embed.", 2)
        out["pred_sted"] = sted[-1]
    if cfg.guided_attn:
        out["weights"], out["ca_weights"] = ws[-1], cws[-1]
    if cfg.aux_loss:
        out["aux_outputs"] = []
        for i in range(cfg.dec_layers - 1):
            a = {"pred_boxes": boxes[i]}
            if cfg.sted:
                a["pred_sted"] = sted[i]
            if cfg.guided_attn:
                a["weights"], a["ca_weights"] = ws[i], cws[i]
            out["aux_outputs"].append(a)
    return out


# --------------------------------------------------------------------------------------------
# loss harness (models/tubedetr.py:257-460, util/box_ops.py)
# --------------------------------------------------------------------------------------------
def cxcywh_to_xyxy(x: Tensor) -> Tensor:
    xc, yc, w, h = x.unbind(-1)
    return torch.stack([xc - 0.5 * w, yc - 0.5 * h, xc + 0.5 * w, yc + 0.5 * h], -1)


def giou_diag(a: Tensor, b: Tensor) -> Tensor:
    """diag(generalized_box_iou(a, b)) for matched xyxy boxes (box_ops.py:95-115)."""
    area = lambda z: (z[:, 2] - z[:, 0]) * (z[:, 3] - z[:, 1])
    lt, rb = torch.max(a[:, :2], b[:, :2]), torch.min(a[:, 2:], b[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, 0] * wh[:, 1]
    union = area(a) + area(b) - inter
    iou = inter / union
    elt, erb = torch.min(a[:, :2], b[:, :2]), torch.max(a[:, 2:], b[:, 2:])
    ewh = (erb - elt).clamp(min=0)
    earea = ewh[:, 0] * ewh[:, 1]
    return iou - (earea - union) / earea


def _losses_one(o: dict, target_boxes: Tensor, num_boxes: float, inter_idx, positive_map: Tensor, time_mask: Tensor, cfg) -> dict:
    out = {}
    pb = o["pred_boxes"]
    out["loss_bbox"] = (pb - target_boxes).abs().sum() / num_boxes
    out["loss_giou"] = (1 - giou_diag(cxcywh_to_xyxy(pb), cxcywh_to_xyxy(target_boxes))).sum() / num_boxes
    if cfg.sted:
        sted = o["pred_sted"].masked_fill(~time_mask[:, :, None], -1e32)
        T = sted.shape[1]
        eps = 1e-6
        total = 0
        for col, tgt in ((0, [x[0] for x in inter_idx]), (1, [x[1] for x in inter_idx])):
            tg = torch.tensor(tgt, dtype=torch.long)
            dist = (-((torch.arange(T)[None, :] - tg[:, None]) ** 2) / (2 * cfg.sigma ** 2)).exp()
            dist = F.normalize(dist + eps, p=1, dim=1)
            p = sted[:, :, col].softmax(1)
            total = total + p * ((p + eps) / dist).log() * time_mask
        out["loss_sted"] = total.mean()
    if cfg.guided_attn:
        pm = positive_map + (~time_mask)
        loss = -(1 - o["weights"] + 1e-6).log()
        loss = loss.masked_fill(pm[:, :, None], 0)
        nb_neg = (~pm).sum(1) + 1e-6
        out["loss_guided_attn"] = (loss.sum(2) / nb_neg[:, None]).sum(1).mean()
    return out


def criterion(outputs: dict, target_boxes: Tensor, inter_idx, time_mask: Tensor, cfg: OracleConfig, world_size: int = 1) -> dict:
    """SetCriterion.forward after engine.py's keep-gather (``outputs['pred_boxes']`` already gathered)."""
    num_boxes = max(float(len(target_boxes)) / world_size, 1.0)
    positive_map = torch.zeros(time_mask.shape, dtype=torch.bool)
    for kk, idx in enumerate(inter_idx):
        if idx[0] >= 0:
            positive_map[kk, idx[0] : idx[1] + 1] = True
    losses = _losses_one(outputs, target_boxes, num_boxes, inter_idx, positive_map, time_mask, cfg)
    for i, aux in enumerate(outputs.get("aux_outputs", [])):
        losses.update({f"{k}_{i}": v for k, v in _losses_one(aux, target_boxes, num_boxes, inter_idx, positive_map, time_mask, cfg).items()})
    return losses


def weight_dict(cfg: OracleConfig) -> dict:
    """tubedetr.py:482-494."""
    base = {"loss_bbox": cfg.bbox_loss_coef, "loss_giou": cfg.giou_loss_coef}
    if cfg.sted:
        base["loss_sted"] = cfg.sted_loss_coef
    if cfg.guided_attn:
        base["loss_guided_attn"] = cfg.guided_attn_loss_coef
    wd = dict(base)
    if cfg.aux_loss:
        for i in range(cfg.dec_layers - 1):
            wd.update({f"{k}_{i}": v for k, v in base.items()})
    return wd


def keep_indices(durations: List[int], inter_idx) -> Tensor:
    """engine.py:83-97."""
    t = max(durations)
    keep = []
    for i, (_dur, inter) in enumerate(zip(durations, inter_idx)):
        keep.extend(range(i * t + inter[0], i * t + inter[1] + 1))
    return torch.tensor(keep, dtype=torch.long)


def train_step(sd, cfg: OracleConfig, batch: dict, world_size: int = 1):
    """engine.py:67-126 forward part: two model calls, keep-gather, criterion, weighted sum."""
    cache = encode(sd, cfg, batch["frames"], batch["frames_mask"], batch["durations"], batch["input_ids"],
                   batch["attention_mask"], batch.get("frames_fast"), batch.get("fast_mask"))
    out = decode(sd, cfg, cache)
    keep = keep_indices(batch["durations"], batch["inter_idx"])
    g = dict(out)
    g["pred_boxes"] = out["pred_boxes"][keep]
    g["aux_outputs"] = [dict(a, pred_boxes=a["pred_boxes"][keep]) for a in out.get("aux_outputs", [])]
    b, t = len(batch["durations"]), max(batch["durations"])
    time_mask = torch.zeros(b, t, dtype=torch.bool)
    for i, dur in enumerate(batch["durations"]):
        time_mask[i, :dur] = True
    ld = criterion(g, batch["target_boxes"], batch["inter_idx"], time_mask, cfg, world_size)
    wd = weight_dict(cfg)
    loss = sum(ld[k] * wd[k] for k in ld if k in wd)
    return loss, ld, out, cache
