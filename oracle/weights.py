"""Deterministic state-dict spec, weight fill and synthetic clips shared by the oracle-side tools.

TEST INFRASTRUCTURE ONLY (see oracle/tubedetr_oracle.py header).

``state_spec`` lists every key/shape of the reference model's ``state_dict()`` (923 entries with
the default flags, SURVEY.md section 5 "Checkpoint / resume"); ``fill_state`` gives each entry a
value that depends only on (seed, key, shape), so the reference model (in gen_golden.py), this
oracle and the product modules can be given bit-identical weights without shipping 741 MB.
Every tensor the reference initialises to 0/1/identity (fast_residual, LayerNorm, FrozenBN
buffers, biases) is randomised on purpose: with the reference init the fast branch and every
FrozenBN are no-ops and parity would be vacuous (SURVEY.md section 8c caveat).
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict
from typing import Dict, List, Tuple

import torch

from .tubedetr_oracle import OracleConfig, time_sine


def _resnet_spec(spec: OrderedDict, cfg: OracleConfig, p: str = "backbone.0.body."):
    def bn(q, c):
        for s in ("weight", "bias", "running_mean", "running_var"):
            spec[q + s] = (c,)

    spec[p + "conv1.weight"] = (64, 3, 7, 7)
    bn(p + "bn1.", 64)
    inplanes = 64
    for name, planes, nblocks, _stride in cfg.stages:
        for i in range(nblocks):
            q = f"{p}{name}.{i}."
            spec[q + "conv1.weight"] = (planes, inplanes, 1, 1)
            bn(q + "bn1.", planes)
            spec[q + "conv2.weight"] = (planes, planes, 3, 3)
            bn(q + "bn2.", planes)
            spec[q + "conv3.weight"] = (planes * 4, planes, 1, 1)
            bn(q + "bn3.", planes * 4)
            if i == 0:
                spec[q + "downsample.0.weight"] = (planes * 4, inplanes, 1, 1)
                bn(q + "downsample.1.", planes * 4)
            inplanes = planes * 4
    return inplanes


def _mha_spec(spec, q, d):
    spec[q + "in_proj_weight"] = (3 * d, d)
    spec[q + "in_proj_bias"] = (3 * d,)
    spec[q + "out_proj.weight"] = (d, d)
    spec[q + "out_proj.bias"] = (d,)


def _ln_spec(spec, q, d):
    spec[q + "weight"] = (d,)
    spec[q + "bias"] = (d,)


def roberta_spec() -> "OrderedDict[str, Tuple[int, ...]]":
    from transformers import RobertaConfig, RobertaModel

    with torch.device("meta"):
        m = RobertaModel(RobertaConfig(vocab_size=50265, max_position_embeddings=514, type_vocab_size=1, pad_token_id=1, layer_norm_eps=1e-5))
    return OrderedDict((k, tuple(v.shape)) for k, v in m.state_dict().items())


def state_spec(cfg: OracleConfig, with_text_encoder: bool = True) -> "OrderedDict[str, Tuple[int, ...]]":
    """Keys in the reference's registration order: transformer, bbox_embed, query_embed, input_proj,
    backbone, sted_embed (tubedetr.py:73-91)."""
    d, ff = cfg.hidden_dim, cfg.dim_feedforward
    spec: OrderedDict = OrderedDict()
    for l in range(cfg.enc_layers):
        q = f"transformer.encoder.layers.{l}."
        _mha_spec(spec, q + "self_attn.", d)
        spec[q + "linear1.weight"], spec[q + "linear1.bias"] = (ff, d), (ff,)
        spec[q + "linear2.weight"], spec[q + "linear2.bias"] = (d, ff), (d,)
        _ln_spec(spec, q + "norm1.", d)
        _ln_spec(spec, q + "norm2.", d)
    for l in range(cfg.dec_layers):
        q = f"transformer.decoder.layers.{l}."
        _mha_spec(spec, q + "self_attn.", d)
        _mha_spec(spec, q + "cross_attn_image.", d)
        spec[q + "linear1.weight"], spec[q + "linear1.bias"] = (ff, d), (ff,)
        spec[q + "linear2.weight"], spec[q + "linear2.bias"] = (d, ff), (d,)
        for n in ("norm1.", "norm3.", "norm4."):
            _ln_spec(spec, q + n, d)
    _ln_spec(spec, "transformer.decoder.norm.", d)
    if not cfg.no_time_embed:
        spec["transformer.time_embed.te"] = (cfg.video_max_len_train, 1, d)
    if cfg.fast:
        for n in ("fast_encoder", "fast_residual"):
            spec[f"transformer.{n}.weight"], spec[f"transformer.{n}.bias"] = (d, d), (d,)
    if with_text_encoder:
        for k, s in roberta_spec().items():
            spec["transformer.text_encoder." + k] = s
    spec["transformer.resizer.fc.weight"], spec["transformer.resizer.fc.bias"] = (d, 768), (d,)
    _ln_spec(spec, "transformer.resizer.layer_norm.", d)
    for i, (a, b) in enumerate(((d, d), (d, d), (d, 4))):
        spec[f"bbox_embed.layers.{i}.weight"], spec[f"bbox_embed.layers.{i}.bias"] = (b, a), (b,)
    spec["query_embed.weight"] = (cfg.num_queries, d)
    spec["input_proj.weight"], spec["input_proj.bias"] = (d, cfg.stages[-1][1] * 4, 1, 1), (d,)
    _resnet_spec(spec, cfg)
    if cfg.sted:
        for i, (a, b) in enumerate(((d, d), (d, 2))):
            spec[f"sted_embed.layers.{i}.weight"], spec[f"sted_embed.layers.{i}.bias"] = (b, a), (b,)
    return spec


def _gen(seed: int, key: str) -> torch.Generator:
    g = torch.Generator()
    g.manual_seed((seed * 1000003 + zlib.crc32(key.encode())) % (2 ** 31))
    return g


def fill_tensor(key: str, shape, seed: int) -> torch.Tensor:
    """Value of one state-dict entry; a pure function of (seed, key, shape)."""
    g = _gen(seed, key)
    n = key.rsplit(".", 1)[-1]
    if key.endswith("time_embed.te"):
        return time_sine(shape[0], shape[2])
    if "position_ids" in key:
        return torch.arange(shape[-1]).expand(shape).clone()
    u = lambda lo, hi: torch.rand(shape, generator=g) * (hi - lo) + lo
    nrm = lambda std: torch.randn(shape, generator=g) * std
    if "backbone" in key:
        if len(shape) == 4:  # conv: kaiming-normal fan_out (torchvision), slightly damped
            return nrm(math.sqrt(2.0 / (shape[0] * shape[2] * shape[3])))
        if n == "weight":
            # residual-branch output BNs are damped so 33 blocks do not blow activations up
            return u(0.25, 0.5) if (".bn3." in key) else u(0.7, 1.3)
        if n == "running_var":
            return u(0.6, 1.4)
        return nrm(0.1)  # bias, running_mean
    if "text_encoder" in key:
        if "LayerNorm.weight" in key:
            return u(0.9, 1.1)
        if "LayerNorm.bias" in key or n == "bias":
            return nrm(0.02)
        return nrm(0.02)
    if "norm" in key:  # LayerNorms of the transformer / resizer
        return u(0.8, 1.2) if n == "weight" else nrm(0.1)
    if n == "bias" or key.endswith("in_proj_bias"):
        return nrm(0.05)
    if len(shape) >= 2:  # xavier-uniform (transformer.py:154-157)
        fan_out, fan_in = shape[0], int(torch.tensor(shape[1:]).prod())
        a = math.sqrt(6.0 / (fan_in + fan_out))
        return u(-a, a)
    return nrm(0.5)


def fill_state(spec: Dict[str, Tuple[int, ...]], seed: int, requires_grad: bool = False) -> "OrderedDict[str, torch.Tensor]":
    sd = OrderedDict()
    for k, s in spec.items():
        v = fill_tensor(k, s, seed)
        if requires_grad and v.is_floating_point() and is_trainable(k):
            v.requires_grad_(True)
        sd[k] = v
    return sd


def is_trainable(key: str) -> bool:
    """backbone.py:82-89: only layer2/3/4 convs train; FrozenBN entries and te are buffers."""
    if key.endswith("time_embed.te") or "position_ids" in key:
        return False
    if "backbone" in key:
        return len(key) > 0 and key.endswith("weight") and (".conv" in key or "downsample.0" in key) and any(s in key for s in ("layer2", "layer3", "layer4"))
    return True


def synthetic_batch(T: int, res: int, k: int, L: int, seed: int, b: int = 1, fast: bool = True, pad_w: int = 0,
                    durations=None, text_pad: int = 0) -> dict:
    """SURVEY.md section 8d synthetic clip: video ~ N(0,1) (3,T,res,res), slow = video[:, ::k], fast = all
    frames (datasets/vidstg.py:250-251), L token ids, every frame annotated.  ``pad_w`` marks the right
    ``pad_w`` pixel columns as padding (mask True, pixels 0) to exercise the mask / pos-enc path;
    ``durations`` (default [T]*b) may hold shorter videos; ``text_pad`` pads the last caption."""
    g = torch.Generator().manual_seed(seed)
    durations = list(durations) if durations is not None else [T] * b
    b = len(durations)
    fast_list, slow_list = [], []
    for dur in durations:
        v = torch.randn(dur, 3, res, res, generator=g)
        fast_list.append(v)
        slow_list.append(v[::k])
    frames_fast = torch.cat(fast_list, 0).contiguous()
    frames = torch.cat(slow_list, 0).contiguous()
    fast_mask = torch.zeros(frames_fast.shape[0], res, res, dtype=torch.bool)
    mask = torch.zeros(frames.shape[0], res, res, dtype=torch.bool)
    if pad_w:
        for x, m in ((frames_fast, fast_mask), (frames, mask)):
            x[..., res - pad_w :] = 0
            m[..., res - pad_w :] = True
    ids = torch.randint(3, 50000, (b, L), generator=g)
    ids[:, 0], ids[:, -1] = 0, 2
    att = torch.ones(b, L, dtype=torch.long)
    if text_pad:
        ids[-1, L - text_pad :] = 1
        ids[-1, L - text_pad - 1] = 2
        att[-1, L - text_pad :] = 0
    n_box = sum(durations)
    cxcy = torch.rand(n_box, 2, generator=g) * 0.6 + 0.2
    wh = torch.rand(n_box, 2, generator=g) * 0.3 + 0.1
    return {
        "frames": frames,
        "frames_mask": mask,
        "frames_fast": frames_fast if fast else None,
        "fast_mask": fast_mask if fast else None,
        "durations": durations,
        "input_ids": ids,
        "attention_mask": att,
        "target_boxes": torch.cat([cxcy, wh], 1),
        "inter_idx": [[0, d - 1] for d in durations],
    }
