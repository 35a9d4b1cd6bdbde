"""Generate tests/golden/*.npz by running THE REFERENCE ITSELF (/root/reference, read-only) on CPU.

TEST INFRASTRUCTURE ONLY; runs only in the build container (the reference does not exist on the
GPU box).  Usage:  python -m oracle.gen_golden            (from the repo root)

Recipe: import transformers first, then register stub modules for the third-party
packages that are absent here (torchvision, timm, hostlist, ...).  The torchvision stub's
``resnet101`` is a plain nn.Module definition of the torchvision architecture with torchvision's
child names; everything else - FrozenBatchNorm2d, BackboneBase, Joiner, PositionEmbeddingSine,
Transformer, TubeDETR, SetCriterion - is the reference's own code.  Weights come from
``oracle.weights.fill_state`` (pure function of seed/key/shape) and are loaded with
``load_state_dict(strict=True)``, which also pins the 923 state-dict keys/shapes.

Stored per case: inputs are NOT stored (regenerated from the seed by oracle.weights.synthetic_batch);
outputs (hs, boxes, sted, TSA / cross-attention weights, memory cache), the 24 losses, and
per-parameter gradient probes (norm + first 8 values) of the eval-mode (dropout-off) backward.
"""
from __future__ import annotations

import argparse
import os
import sys
import types

import numpy as np
import torch
from torch import nn

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

CASES = {
    # name: (synthetic_batch kwargs, OracleConfig overrides)
    "a_b1_T8_res96_k4": (dict(T=8, res=96, k=4, L=6, seed=11, pad_w=20), dict(stride=4)),
    "b_b2_T8-6_res64_k4": (dict(T=8, res=64, k=4, L=5, seed=12, durations=[8, 6], text_pad=2), dict(stride=4)),
    "c_nofast_T6_res64_k2": (dict(T=6, res=64, k=2, L=4, seed=13, fast=False, pad_w=9), dict(stride=2, fast=False)),
    "d_notsa_T5_res64_k5": (dict(T=5, res=64, k=5, L=4, seed=14), dict(stride=5, no_tsa=True)),
}
# Ablation variants (SURVEY.md 8a': accepted flags outside the kernel scope).  The CPU oracle does not restate them - these
# vectors pin the PRODUCT directly against the reference's own output: (synthetic_batch kwargs, OracleConfig overrides,
# reference-argument overrides that OracleConfig has no field for).
VARIANTS = {
    "v_gating_T6_res64_k2": (dict(T=6, res=64, k=2, L=4, seed=21), dict(stride=2), dict(fast_mode="gating")),
    "v_pool_T6_res64_k3": (dict(T=6, res=64, k=3, L=4, seed=22, pad_w=12), dict(stride=3), dict(fast_mode="pool")),
    "v_transformer_T4_res64_k2": (dict(T=4, res=64, k=2, L=4, seed=23), dict(stride=2), dict(fast_mode="transformer")),
    "v_noslow_T6_res64_k2": (dict(T=6, res=64, k=2, L=4, seed=24, text_pad=1), dict(stride=2), dict(fast_mode="noslow")),
    "v_stride0_T5-3_res64": (dict(T=5, res=64, k=1, L=4, seed=25, fast=False, durations=[5, 3]), dict(stride=0, fast=False), dict()),
    "v_learned_T6_res64_k2": (dict(T=6, res=64, k=2, L=4, seed=26), dict(stride=2), dict(learn_time_embed=True, position_embedding="learned")),
    # head / loss switches of main.py (--no_sted, --no_guided_attn, --no_aux_loss: the output dict and the loss dict lose keys) and --no_time_embed
    "v_boxesonly_T6_res64_k2": (dict(T=6, res=64, k=2, L=4, seed=27, pad_w=7), dict(stride=2, sted=False, guided_attn=False, aux_loss=False), dict()),
    # --freeze_backbone --freeze_text_encoder (the trunk's and RoBERTa's backward disappear from the graph), --sigma 2 and non-default loss coefficients
    "v_frozen_T6_res64_k2": (dict(T=6, res=64, k=2, L=5, seed=29, pad_w=5), dict(stride=2, sigma=2.0, bbox_loss_coef=3.0, giou_loss_coef=1.5, sted_loss_coef=7.0, guided_attn_loss_coef=0.5),
                             dict(freeze_backbone=True, freeze_text_encoder=True)),
    "v_notime_T6-5_res64_k2": (dict(T=6, res=64, k=2, L=4, seed=28, durations=[6, 5], text_pad=1), dict(stride=2, no_time_embed=True), dict()),
}
WEIGHT_SEED = 7


def install_stubs():
    from transformers import BatchEncoding, RobertaConfig, RobertaModel, RobertaTokenizerFast  # noqa: F401  (before the stubs)

    class Bottleneck(nn.Module):
        expansion = 4

        def __init__(self, inplanes, planes, stride, downsample, norm_layer):
            super().__init__()
            self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
            self.bn1 = norm_layer(planes)
            self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
            self.bn2 = norm_layer(planes)
            self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
            self.bn3 = norm_layer(planes * 4)
            self.relu = nn.ReLU(inplace=True)
            self.downsample = downsample

        def forward(self, x):
            idt = x if self.downsample is None else self.downsample(x)
            y = self.relu(self.bn1(self.conv1(x)))
            y = self.relu(self.bn2(self.conv2(y)))
            y = self.bn3(self.conv3(y))
            return self.relu(y + idt)

    class ResNet(nn.Module):
        def __init__(self, layers, norm_layer):
            super().__init__()
            self.inplanes = 64
            self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
            self.bn1 = norm_layer(64)
            self.relu = nn.ReLU(inplace=True)
            self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
            self.layer1 = self._make(64, layers[0], 1, norm_layer)
            self.layer2 = self._make(128, layers[1], 2, norm_layer)
            self.layer3 = self._make(256, layers[2], 2, norm_layer)
            self.layer4 = self._make(512, layers[3], 2, norm_layer)
            self.avgpool = nn.AdaptiveAvgPool2d(1)
            self.fc = nn.Linear(2048, 1000)

        def _make(self, planes, n, stride, norm_layer):
            ds = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False), norm_layer(planes * 4))
            blocks = [Bottleneck(self.inplanes, planes, stride, ds, norm_layer)]
            self.inplanes = planes * 4
            blocks += [Bottleneck(self.inplanes, planes, 1, None, norm_layer) for _ in range(1, n)]
            return nn.Sequential(*blocks)

    def resnet101(replace_stride_with_dilation=None, pretrained=False, norm_layer=nn.BatchNorm2d):
        assert not any(replace_stride_with_dilation or [False])
        return ResNet([3, 4, 23, 3], norm_layer)

    class IntermediateLayerGetter(nn.ModuleDict):
        def __init__(self, model, return_layers):
            layers, remaining = {}, dict(return_layers)
            for name, module in model.named_children():
                layers[name] = module
                remaining.pop(name, None)
                if not remaining:
                    break
            super().__init__(layers)
            self.return_layers = dict(return_layers)

        def forward(self, x):
            out = {}
            for name, module in self.items():
                x = module(x)
                if name in self.return_layers:
                    out[self.return_layers[name]] = x
            return out

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("hostlist")
    tv = mod("torchvision")
    tv.models = mod("torchvision.models", resnet101=resnet101)
    tv.models._utils = mod("torchvision.models._utils", IntermediateLayerGetter=IntermediateLayerGetter)
    tv.ops = mod("torchvision.ops")
    tv.ops.boxes = mod("torchvision.ops.boxes", box_area=lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]))
    timm = mod("timm")
    timm.models = mod("timm.models", create_model=None)
    sys.path.insert(0, REF)

    import models.transformer as mt

    class FixedTokenizer:
        """Stands in for RobertaTokenizerFast (no tokenizer files offline): returns preset ids."""

        def __init__(self):
            self.ids = self.att = None

        def batch_encode_plus(self, text, padding=None, return_tensors=None):
            be = BatchEncoding({"input_ids": self.ids.clone(), "attention_mask": self.att.clone()})
            be._encodings = [None] * len(text)
            return be

    tok = FixedTokenizer()
    mt.RobertaTokenizerFast.from_pretrained = staticmethod(lambda *a, **k: tok)
    mt.RobertaModel.from_pretrained = staticmethod(
        lambda *a, **k: RobertaModel(RobertaConfig(vocab_size=50265, max_position_embeddings=514, type_vocab_size=1, pad_token_id=1, layer_norm_eps=1e-5))
    )
    return tok


def ref_args(cfg, **extra):
    a = _ref_args(cfg)
    a.__dict__.update(extra)
    return a


def _ref_args(cfg):
    return types.SimpleNamespace(
        device="cpu", hidden_dim=cfg.hidden_dim, dropout=0.1, nheads=cfg.nheads, dim_feedforward=cfg.dim_feedforward,
        enc_layers=cfg.enc_layers, dec_layers=cfg.dec_layers, pass_pos_and_query=True, text_encoder_type="roberta-base",
        freeze_text_encoder=False, video_max_len_train=cfg.video_max_len_train, stride=cfg.stride, no_tsa=cfg.no_tsa,
        guided_attn=cfg.guided_attn, fast=cfg.fast, fast_mode="", learn_time_embed=False, rd_init_tsa=False,
        no_time_embed=cfg.no_time_embed, position_embedding="sine", lr_backbone=1e-5, backbone="resnet101", dilation=False,
        freeze_backbone=False, num_queries=cfg.num_queries, aux_loss=cfg.aux_loss, sted=cfg.sted, sigma=int(cfg.sigma),
        bbox_loss_coef=cfg.bbox_loss_coef, giou_loss_coef=cfg.giou_loss_coef, sted_loss_coef=cfg.sted_loss_coef,
        guided_attn_loss_coef=cfg.guided_attn_loss_coef,
    )


def run_case(name, tok):
    from oracle.tubedetr_oracle import OracleConfig, keep_indices
    from oracle.weights import fill_state, is_trainable, state_spec, synthetic_batch
    from models import build_model
    from util.misc import NestedTensor

    variant = name in VARIANTS
    bkw, ckw, extra = VARIANTS[name] if variant else (CASES[name] + ({},))
    cfg = OracleConfig(**ckw)
    batch = synthetic_batch(**bkw)
    torch.manual_seed(0)
    model, criterion, weight_dict = build_model(ref_args(cfg, **extra))
    ref_sd = model.state_dict()
    if variant:  # the variant's own parameters (fast_encoder.layers.0..., time_embed.time_embed.weight, ...) are filled by the same pure function
        spec = {k: tuple(v.shape) for k, v in ref_sd.items()}
    else:
        spec = state_spec(cfg)
        assert list(ref_sd.keys()) == list(spec.keys()), "state-dict key order/names differ from oracle.weights.state_spec"
        assert all(tuple(ref_sd[k].shape) == tuple(spec[k]) for k in spec)
    model.load_state_dict(fill_state(spec, WEIGHT_SEED), strict=True)
    trainable = {k for k, p in model.named_parameters() if p.requires_grad}
    unused = {k for k in trainable if "pooler" in k}
    if not variant:
        assert trainable == {k for k in spec if is_trainable(k)}, "freeze rule mismatch"
    model.eval()  # dropout off (parity mode); gradients still flow

    tok.ids, tok.att = batch["input_ids"], batch["attention_mask"]
    durations = batch["durations"]
    samples = NestedTensor(batch["frames"], batch["frames_mask"])
    samples_fast = NestedTensor(batch["frames_fast"], batch["fast_mask"]) if cfg.fast else None
    captions = ["x"] * len(durations)
    cache = model(samples, durations, captions, encode_and_save=True, samples_fast=samples_fast)
    out = model(samples, durations, captions, encode_and_save=False, memory_cache=cache)

    res = {}
    for k in ("img_memory", "mask", "pos_embed", "query_embed", "query_mask", "text_memory", "text_memory_resized", "text_attention_mask"):
        if cache[k] is not None:  # (stride 0: no time-query mask)
            res["cache." + k] = cache[k].detach().numpy()
    layers = out.get("aux_outputs", []) + [out]  # (no "aux_outputs" without --aux_loss; "pred_sted" / "weights" / "ca_weights" only with their flags)
    for key in ("pred_boxes", "pred_sted", "weights", "ca_weights"):
        if key in out:
            res["out." + key] = np.stack([o[key].detach().numpy() for o in layers])

    # engine.py:83-126 by hand
    keep = keep_indices(durations, batch["inter_idx"])
    out["pred_boxes"] = out["pred_boxes"][keep]
    for a in out.get("aux_outputs", []):
        a["pred_boxes"] = a["pred_boxes"][keep]
    b, t = len(durations), max(durations)
    time_mask = torch.zeros(b, t).bool()
    for i, d in enumerate(durations):
        time_mask[i, :d] = True
    targets = [{"boxes": bx[None]} for bx in batch["target_boxes"]]
    loss_dict = criterion(out, targets, batch["inter_idx"], time_mask)
    assert set(loss_dict) <= set(weight_dict)  # (== with --sted; without it the reference keeps loss_sted's coefficient in weight_dict, tubedetr.py:482-486)
    loss = sum(loss_dict[k] * weight_dict[k] for k in loss_dict)
    loss.backward()
    res["loss.names"] = np.array(sorted(loss_dict))
    res["loss.values"] = np.array([loss_dict[k].item() for k in sorted(loss_dict)], dtype=np.float64)
    res["loss.total"] = np.array(loss.item())
    names, norms, heads = [], [], []
    for k, p in model.named_parameters():
        if not p.requires_grad or k in unused:
            continue
        if variant and p.grad is None:  # parameters a variant leaves out of its graph (noslow: the whole space-text encoder)
            continue
        assert p.grad is not None, k
        names.append(k)
        norms.append(p.grad.double().norm().item())
        h = torch.zeros(8)
        fl = p.grad.flatten()[:8]
        h[: fl.numel()] = fl
        heads.append(h.numpy())
    res["grad.names"] = np.array(names)
    res["grad.norms"] = np.array(norms)
    res["grad.heads"] = np.stack(heads)
    res["meta.n_state_keys"] = np.array(len(spec))
    if variant:
        res["meta.state_keys"] = np.array(list(spec.keys()))
        res["meta.trainable"] = np.array(sorted(trainable))
        res["meta.no_grad"] = np.array(sorted(k for k, p in model.named_parameters() if p.requires_grad and p.grad is None))
    res["meta.n_params"] = np.array(sum(p.numel() for p in model.parameters()))
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **res)
    print(name, "loss", loss.item(), "keys", len(spec), "saved", sum(v.nbytes for v in res.values()) // 1024, "KiB")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("cases", nargs="*", default=list(CASES) + list(VARIANTS))
    a = ap.parse_args()
    sys.path.insert(0, os.path.dirname(HERE))
    tok = install_stubs()
    torch.set_num_threads(8)
    for c in a.cases:
        run_case(c, tok)
