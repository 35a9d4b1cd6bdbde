"""TubeDETR top module, heads and criterion behind the reference's ``models.tubedetr`` interface
(models/tubedetr.py:23-42 MLP, 45-254 TubeDETR, 257-460 SetCriterion, 463-506 build).

``model(samples, durations, captions, encode_and_save=True, samples_fast=...)`` returns the 9-key memory cache,
``model(..., encode_and_save=False, memory_cache=cache)`` returns pred_boxes / pred_sted / weights / ca_weights /
aux_outputs exactly like the reference, so engine.py's train_one_epoch / evaluate drive it unchanged.  All
heavy math runs in the gfx950 kernels (see backbone.py / transformer.py / functional.py); the per-video
padding bookkeeping the reference does with Python slice loops is a cached index gather here.
The criterion is small elementwise math on (T,4)/(T,2)/(T,T) tensors and stays PyTorch-ROCm ops (SURVEY.md
section 2 row 5)."""
from __future__ import annotations

import math
from typing import List

import torch
import torch.nn.functional as F
from torch import nn

from .. import functional as Fk
from ..util.misc import FrameSources, LRUCache, NestedTensor
from .backbone import build_backbone
from .transformer import build_transformer


class MLP(nn.Module):
    """Linear stack with ReLU between layers; optional dropout after EVERY layer incl. the last (tubedetr.py:37-42)."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers, dropout=0):
        super().__init__()
        self.num_layers = num_layers
        dims = [input_dim] + [hidden_dim] * (num_layers - 1) + [output_dim]
        self.layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))
        self.dropout = dropout

    def forward(self, rows: torch.Tensor) -> torch.Tensor:
        for i, layer in enumerate(self.layers):
            rows = Fk.linear(rows, layer.weight, layer.bias, relu=i < self.num_layers - 1,
                             dropout_p=float(self.dropout or 0.0), training=self.training)
        return rows


def _parts(x):
    return list(x.parts) if isinstance(x, FrameSources) else [(x, None)]


def _valid(x):
    return list(x.valid) if isinstance(x, FrameSources) else [None]


def _one_tensor(x):
    """The single un-indexed tensor behind ``x`` (a frame tensor, or a FrameSources wrapping exactly one)."""
    if isinstance(x, FrameSources):
        assert len(x.parts) == 1 and x.parts[0][1] is None, "dedupe needs the fast frames as one contiguous tensor"
        return x.parts[0][0]
    return x


class TubeDETR(nn.Module):
    def __init__(self, backbone, transformer, num_queries, aux_loss=False, video_max_len=200, stride=5, guided_attn=False,
                 fast=False, fast_mode="", sted=True, compute_dtype=torch.float32):
        super().__init__()
        self.num_queries = num_queries
        self.transformer = transformer
        hidden_dim = transformer.d_model
        self.bbox_embed = MLP(hidden_dim, hidden_dim, 4, 3)
        self.query_embed = nn.Embedding(num_queries, hidden_dim)
        self.input_proj = nn.Conv2d(backbone.num_channels, hidden_dim, kernel_size=1)  # parameter holder
        self.backbone = backbone
        self.aux_loss = aux_loss
        self.video_max_len = video_max_len
        self.stride = stride
        self.guided_attn = guided_attn
        self.fast = fast
        self.fast_mode = fast_mode
        self.sted = sted
        if sted:
            self.sted_embed = MLP(hidden_dim, hidden_dim, 2, 2, dropout=0.5)
        self._idx_cache = LRUCache()
        from .position_encoding import PositionEmbeddingSine

        from .backbone import Joiner

        # the sine fast path (encoding generated per clip by td_pos_sine, forward_split, want_pos=) needs THIS package's Joiner; any
        # other backbone object is driven through the reference's plain forward(tensor_list) -> (features, pos) protocol
        self._joiner = isinstance(backbone, Joiner)
        pe = backbone[1] if self._joiner else None
        self._sine_pos = isinstance(pe, PositionEmbeddingSine)
        if self._sine_pos:
            transformer.sine_pos = (pe.num_pos_feats, float(pe.temperature))
        # "The slow frames ARE the fast frames [::stride] of each video" (datasets/vidstg.py:250-251): the trunk then need not compute
        # those pixels twice (slow ones first, so backward still walks a contiguous prefix).  None (default) = PROVEN per call from the
        # inputs (``_slow_is_strided_fast``: the slow clip is an index list over the very buffer the fast frames are, and that list is
        # 0, k, 2k, ... of every video) - SURVEY 8a' "dead work the build may skip"; True = the caller vouches for inputs the model
        # cannot see through (two separate tensors with equal pixels); False = always run both passes as the reference does.
        self.slow_frames_are_strided_fast = None
        self.set_compute_dtype(compute_dtype)

    def set_compute_dtype(self, dt: torch.dtype):
        """torch.float32: exact-fp32 MFMA kernels (parity mode); torch.bfloat16: bf16 MFMA, fp32 accumulate."""
        assert dt in (torch.float32, torch.bfloat16)
        self.compute_dtype = dt
        self.backbone.set_compute_dtype(dt)
        self.transformer.compute_dtype = dt
        return self

    def _project(self, feat: torch.Tensor) -> torch.Tensor:
        """input_proj 1x1 conv on a channels-last (N,C,h,w) view -> (N,d,h,w) channels-last view."""
        n, c, h, w = feat.shape
        rows = feat.permute(0, 2, 3, 1).reshape(n * h * w, c)
        y = Fk.linear(rows, self.input_proj.weight.view(self.input_proj.out_channels, c), self.input_proj.bias)
        return y.view(n, h, w, -1).permute(0, 3, 1, 2)

    def _slow_is_strided_fast(self, samples, samples_fast, durations) -> bool:
        """Structural proof, no device traffic: samples.tensors is a FrameSources of ONE part (base, index) whose base IS samples_fast's
        frame tensor (same storage pointer, shape and dtype) and whose index list - known on the host, FrameSources.index_host - is
        exactly 0, k, 2k, ... within every video of ``durations``."""
        flag = self.slow_frames_are_strided_fast
        if flag is not None:
            return bool(flag)
        xs, xf = samples.tensors, samples_fast.tensors
        if not isinstance(xs, FrameSources) or len(xs.parts) != 1 or xs.index_host[0] is None:
            return False
        if isinstance(xf, FrameSources):
            if len(xf.parts) != 1 or xf.parts[0][1] is not None:
                return False
            fast = xf.parts[0][0]
        else:
            fast = xf
        base = xs.parts[0][0]
        if not (torch.is_tensor(fast) and fast.data_ptr() == base.data_ptr() and fast.shape == base.shape and fast.dtype == base.dtype and fast.stride() == base.stride()):
            return False
        want, off = [], 0
        for d in durations:
            want += list(range(off, off + d, self.stride))
            off += d
        return off == fast.shape[0] and tuple(want) == xs.index_host[0]

    def _dedupe_index(self, durations, device):
        """perm: fast-frame indices with each video's slow frames (0, k, 2k, ...) first, in the slow batch order;
        inv: position of every fast frame inside the permuted batch."""
        key = ("dedupe", tuple(durations), self.stride, str(device))
        if key not in self._idx_cache:
            slow, rest, base = [], [], 0
            for d in durations:
                slow += [base + j for j in range(0, d, self.stride)]
                rest += [base + j for j in range(d) if j % self.stride]
                base += d
            perm = torch.tensor(slow + rest, dtype=torch.long)
            inv = torch.empty_like(perm)
            inv[perm] = torch.arange(perm.numel())
            self._idx_cache[key] = (perm.to(device), inv.to(device))
        return self._idx_cache[key]

    def _frame_index(self, durations, device):
        key = (tuple(durations), str(device))
        if key not in self._idx_cache:
            t = max(durations)
            dest = torch.cat([torch.arange(d) + i * t for i, d in enumerate(durations)])
            self._idx_cache[key] = dest.to(device)
        return self._idx_cache[key]

    def forward(self, samples: NestedTensor, durations, captions, encode_and_save=True, memory_cache=None, samples_fast=None):
        if encode_and_save:
            assert memory_cache is None
            if not isinstance(samples, NestedTensor):
                samples = NestedTensor.from_tensor_list(samples)
            return self._encode(samples, durations, captions, samples_fast)
        assert memory_cache is not None
        return self._decode(memory_cache)

    def _encode_dense(self, samples, durations, captions):
        """--stride 0 (models/tubedetr.py:140-153): no temporal sampling - every frame is encoded with the text, videos shorter
        than the longest are padded in time with zero features / all-True masks.  An ablation outside the kernel scope
        (SURVEY.md 8a'): the trunk, the encoder and the decoder run as usual, the temporal padding is stock PyTorch indexing."""
        b, t = len(durations), max(durations)
        features, pos = self.backbone(samples)
        src, mask = features[-1].decompose()
        src, pos_embed = self._project(src), pos[-1]
        dev = src.device
        dest = self._frame_index(durations, dev)
        tpad_mask = mask.clone()
        if dest.numel() != b * t:
            def pad_time(x):  # (n, C, h, w) channels-last view -> (b*t, C, h, w), zero rows for the padded frames
                rows = torch.zeros((b * t,) + tuple(x.permute(0, 2, 3, 1).shape[1:]), dtype=x.dtype, device=dev)
                return rows.index_put((dest,), x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)

            src, pos_embed = pad_time(src), pad_time(pos_embed)
            tpad_mask = torch.ones((b * t,) + tuple(mask.shape[1:]), dtype=torch.bool, device=dev)
            tpad_mask[dest] = mask
        tpad_mask[:, 0, 0] = False  # avoid empty masks
        return self.transformer(src, tpad_mask, self.query_embed.weight, pos_embed, captions, encode_and_save=True,
                                durations=durations, tpad_mask_t=None, fast_src=None)

    def _encode(self, samples, durations, captions, samples_fast):
        if not self.stride:
            return self._encode_dense(samples, durations, captions)
        b, t, k = len(durations), max(durations), self.stride
        # sine encoding: the transformer forms the positional operand from the (original) pad mask itself, see Joiner.forward
        want_pos = not self._sine_pos
        merged = self._joiner and self.fast and samples_fast is not None and torch.is_grad_enabled() and samples_fast.tensors.shape[1:] == samples.tensors.shape[1:]
        dedupe = merged and sum(durations) == samples_fast.tensors.shape[0] and self._slow_is_strided_fast(samples, samples_fast, durations)
        if merged:
            # a pass that keeps activations for backward holds at most ResNetBody.max_frames frames (32-bit tensor addressing: 1 083
            # bf16 frames at res 352); a larger batch runs the slow frames (kept for backward) and the no-grad fast frames as two
            # passes (the no-grad one is one td_resnet_fwd call at any size: its oversized launches go out in frame groups)
            body = self.backbone[0].body
            limit = body.max_frames(samples.tensors.shape[-2], samples.tensors.shape[-1], self.compute_dtype)
            merged = (samples_fast.tensors.shape[0] if dedupe else samples.tensors.shape[0] + samples_fast.tensors.shape[0]) <= limit
        if merged and dedupe:
            # one trunk pass over the fast frames, permuted so that the slow (= every k-th) frames come first
            n_slow = samples.tensors.shape[0]
            perm, inv = self._dedupe_index(durations, samples_fast.tensors.device)
            assert perm.numel() == samples_fast.tensors.shape[0] and n_slow == sum(math.ceil(d / k) for d in durations)
            # (the permutation is an index list handed to the trunk's input kernel: the pixels are not copied)
            both = NestedTensor(FrameSources([(_one_tensor(samples_fast.tensors), perm)], _valid(samples_fast.tensors)), samples_fast.mask[perm])
            features, pos_all = self.backbone(both, n_slow, want_pos=want_pos)
            src_all, mask_all = features[-1].decompose()
            src, mask, pos = src_all[:n_slow], mask_all[:n_slow], [pos_all[-1][:n_slow] if want_pos else None]
            src_fast_feat, mask_fast = src_all.detach()[inv], mask_all[inv]
        elif merged:
            # slow (grad) and fast (no_grad, tubedetr.py:128-129) frames share the trunk weights: one launch sequence over
            # both, with the slow frames first; only they are saved-for / reached-by backward.
            n_slow = samples.tensors.shape[0]
            both = FrameSources(_parts(samples.tensors) + _parts(samples_fast.tensors), _valid(samples.tensors) + _valid(samples_fast.tensors))
            slow, pos_slow, fast = self.backbone.forward_split(both, n_slow, samples.mask, samples_fast.mask, want_pos=want_pos)
            (src, mask), pos = slow.decompose(), [pos_slow]
            src_fast_feat, mask_fast = fast.decompose()
        else:
            features, pos = self.backbone(samples, want_pos=want_pos) if self._joiner else self.backbone(samples)  # (a foreign backbone: the reference's protocol)
            src, mask = features[-1].decompose()
        dev = src.device
        dest = self._frame_index(durations, dev)
        identity = dest.numel() == b * t
        fast_src = None
        if self.fast:
            if not merged and dedupe:
                # the slow frames ARE fast frames (video[::k]): only the other fast frames go through the no-grad pass, the slow
                # frames' features are taken (detached) from the pass above
                n_slow = src.shape[0]
                perm, inv = self._dedupe_index(durations, samples_fast.tensors.device)
                assert n_slow == sum(math.ceil(d / k) for d in durations)
                if perm.numel() == n_slow:  # stride 1: every fast frame is a slow frame - there is no other frame to run
                    src_fast_feat, mask_fast = src.detach()[inv], mask[inv]
                else:
                    rest = NestedTensor(FrameSources([(_one_tensor(samples_fast.tensors), perm[n_slow:])], _valid(samples_fast.tensors)), samples_fast.mask[perm[n_slow:]])
                    with torch.no_grad():
                        features_rest, _ = self.backbone(rest, want_pos=False)
                    src_rest, mask_rest = features_rest[-1].decompose()
                    src_fast_feat, mask_fast = torch.cat([src.detach(), src_rest])[inv], torch.cat([mask, mask_rest])[inv]
            elif not merged:
                with torch.no_grad():  # the fast branch does not back-propagate into the backbone (tubedetr.py:128-129)
                    features_fast, _ = self.backbone(samples_fast, want_pos=False) if self._joiner else self.backbone(samples_fast)  # its encoding is never used
                src_fast_feat, mask_fast = features_fast[-1].decompose()
            src_fast = self._project(src_fast_feat)
        src = self._project(src)
        n, f, h, w = src.shape
        n_clips = math.ceil(t / k)
        assert n == b * n_clips, "all videos of a batch must have the same number of slow clips"
        tpad_mask_t = torch.ones(b * t, h, w, dtype=torch.bool, device=dev)
        if self.fast:
            if identity:
                fast_src, tpad_mask_t = src_fast, mask_fast.clone()
            else:
                fast_rows = torch.zeros((b * t, h, w, f), dtype=src_fast.dtype, device=dev)
                fast_rows = fast_rows.index_put((dest,), src_fast.permute(0, 2, 3, 1))
                fast_src = fast_rows.permute(0, 3, 1, 2)
                tpad_mask_t[dest] = mask_fast
        else:  # frames inherit the mask of their slow clip (tubedetr.py:172-178)
            clip_of = torch.cat([i * n_clips + torch.arange(d) // k for i, d in enumerate(durations)]).to(dev)
            tpad_mask_t[dest] = mask[clip_of]
        tpad_mask = mask.clone()
        tpad_mask[:, 0, 0] = False  # avoid empty masks
        tpad_mask_t[:, 0, 0] = False
        return self.transformer(src, tpad_mask, self.query_embed.weight, pos[-1], captions, encode_and_save=True,
                                durations=durations, tpad_mask_t=tpad_mask_t, fast_src=fast_src, pos_mask=None if want_pos else mask)

    def _decode(self, memory_cache):
        res = self.transformer(img_memory=memory_cache["img_memory"], mask=memory_cache["mask"], pos_embed=memory_cache["pos_embed"],
                               query_embed=memory_cache["query_embed"], query_mask=memory_cache["query_mask"], encode_and_save=False,
                               text_memory=memory_cache["text_memory"], text_mask=memory_cache["text_attention_mask"])
        if self.guided_attn:
            hs, weights, cross_weights = res
        else:
            hs = res
        nl, b, t, d = hs.shape
        rows = hs.reshape(nl * b * t, d)
        out = {}
        if self.sted:
            outputs_sted = Fk.cast(self.sted_embed(rows), torch.float32).view(nl, b, t, 2)
        outputs_coord = Fk.cast(self.bbox_embed(rows), torch.float32).view(nl, b * t, 4).sigmoid()
        # the per-layer entries below are views of these stacked tensors; the fused criterion (SetCriterion.forward_fused)
        # consumes them directly instead of re-stacking six dict entries
        self._last_stacked = {"pred_boxes": outputs_coord, "pred_sted": outputs_sted if self.sted else None,
                              "weights": weights if self.guided_attn else None, "b": b, "t": t}
        out["pred_boxes"] = outputs_coord[-1]
        if self.sted:
            out["pred_sted"] = outputs_sted[-1]
        if self.guided_attn:
            out["weights"] = weights[-1]
            out["ca_weights"] = cross_weights[-1]
        if self.aux_loss:
            out["aux_outputs"] = []
            for i in range(nl - 1):
                a = {"pred_boxes": outputs_coord[i]}
                if self.sted:
                    a["pred_sted"] = outputs_sted[i]
                if self.guided_attn:
                    a["weights"] = weights[i]
                    a["ca_weights"] = cross_weights[i]
                out["aux_outputs"].append(a)
        return out


# ----------------------------------------------------------------------------------------------------------
# criterion (PyTorch-ROCm elementwise ops; same loss names / formulas as tubedetr.py:270-372)
# ----------------------------------------------------------------------------------------------------------
def _xyxy(b):
    cx, cy, w, h = b.unbind(-1)
    return torch.stack((cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h), -1)


def _paired_giou(a, b):
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    # (the reference asserts non-degenerate boxes here, util/box_ops.py:105-106; a device-side assert would
    #  synchronise the stream every step, so it is left to the data pipeline)
    iwh = (torch.min(a[:, 2:], b[:, 2:]) - torch.max(a[:, :2], b[:, :2])).clamp(min=0)
    inter = iwh[:, 0] * iwh[:, 1]
    union = area_a + area_b - inter
    ewh = (torch.max(a[:, 2:], b[:, 2:]) - torch.min(a[:, :2], b[:, :2])).clamp(min=0)
    hull = ewh[:, 0] * ewh[:, 1]
    return inter / union - (hull - union) / hull


class CriterionFn(torch.autograd.Function):
    """All 24 losses in one HIP launch (csrc/criterion.hip); the launch also stores every loss's derivative, backward is
    one more tiny launch that scales them by the upstream gradient of the [layers, 4] loss matrix."""

    @staticmethod
    def forward(ctx, boxes, sted, weights, tgt, keep, tm_u8, pm_u8, inter_dev, num_boxes, sigma):
        from .. import _hip

        nl, bt, _ = boxes.shape
        b, T = tm_u8.shape
        dev = boxes.device
        boxes, tgt = boxes.contiguous(), tgt.contiguous().float()
        sted = sted.contiguous() if sted is not None else None
        weights = weights.contiguous() if weights is not None else None
        losses = torch.empty((nl, 4), dtype=torch.float32, device=dev)
        g_l1, g_giou = torch.empty_like(boxes), torch.empty_like(boxes)
        g_sted = torch.empty_like(sted) if sted is not None else None
        g_w = torch.empty_like(weights) if weights is not None else None
        nb_dev = num_boxes if torch.is_tensor(num_boxes) else None
        _hip.check(_hip.lib().td_criterion_fwd(_hip.ptr(boxes), _hip.ptr(tgt), _hip.ptr(keep), _hip.ptr(sted), _hip.ptr(weights), _hip.ptr(tm_u8),
                                               _hip.ptr(pm_u8), _hip.ptr(inter_dev), _hip.ptr(nb_dev), 0.0 if nb_dev is not None else float(num_boxes),
                                               float(sigma), nl, b, T, keep.numel(), _hip.ptr(losses), _hip.ptr(g_l1), _hip.ptr(g_giou),
                                               _hip.ptr(g_sted), _hip.ptr(g_w), _hip.stream_ptr()), "td_criterion_fwd")
        ctx.save_for_backward(g_l1, g_giou, g_sted, g_w)
        ctx.dims = (nl, b, T)
        return losses

    @staticmethod
    def backward(ctx, dl):
        from .. import _hip

        g_l1, g_giou, g_sted, g_w = ctx.saved_tensors
        nl, b, T = ctx.dims
        d_boxes = torch.empty_like(g_l1)
        d_sted = torch.empty_like(g_sted) if g_sted is not None else None
        d_w = torch.empty_like(g_w) if g_w is not None else None
        _hip.check(_hip.lib().td_criterion_bwd(_hip.ptr(dl.contiguous().float()), _hip.ptr(g_l1), _hip.ptr(g_giou), _hip.ptr(g_sted), _hip.ptr(g_w),
                                               _hip.ptr(d_boxes), _hip.ptr(d_sted), _hip.ptr(d_w), nl, b, T, _hip.stream_ptr()), "td_criterion_bwd")
        return d_boxes, d_sted, d_w, None, None, None, None, None, None, None


LOSS_COLUMNS = ("loss_bbox", "loss_giou", "loss_sted", "loss_guided_attn")


class SetCriterion(nn.Module):
    """Same loss names / formulas as the reference (tubedetr.py:270-372, 397-460).  The reference loops over the main
    output and the 5 auxiliary decoder layers in Python (24 x ~15 tiny kernels + as many autograd nodes); here the six
    layers are stacked and every loss is evaluated once on the stacked tensors - identical values, 6x fewer launches."""

    def __init__(self, losses, sigma=1):
        super().__init__()
        self.losses = losses
        self.sigma = sigma
        self._pm_cache = LRUCache()
        self._tgt_cache = LRUCache()
        self.external_num_boxes = None
        self.last_loss_matrix = None
        self._fused_keys: set = set()

    # ---- per-loss math on stacked layers: leading dim = decoder layer ----
    def _boxes(self, src, tgt, num_boxes):  # src (Lyr, n, 4), tgt (n, 4)
        nl = src.shape[0]
        giou = _paired_giou(_xyxy(src).reshape(-1, 4), _xyxy(tgt).repeat(nl, 1)).view(nl, -1)
        return {"loss_bbox": (src - tgt).abs().sum((1, 2)) / num_boxes, "loss_giou": (1 - giou).sum(1) / num_boxes}

    def _sted(self, sted, inter_idx, time_mask):  # sted (Lyr, b, T, 2)
        sted = sted.masked_fill(~time_mask[None, :, :, None], -1e32)
        T, dev, eps = sted.shape[2], sted.device, 1e-6
        key = (tuple(map(tuple, inter_idx)), T, str(dev))
        gauss = self._tgt_cache.get(key)
        if gauss is None:  # Gaussian start / end targets (b, T, 2), built once per annotation pattern
            grid = torch.arange(T)[None, :]
            gs = []
            for which in (0, 1):
                tgt = torch.tensor([x[which] for x in inter_idx], dtype=torch.long)
                g_ = (-((grid - tgt[:, None]) ** 2) / (2 * self.sigma ** 2)).exp()
                gs.append(F.normalize(g_ + eps, p=1, dim=1))
            gauss = self._tgt_cache[key] = torch.stack(gs, -1).to(dev)
        p = sted.softmax(2)
        kl = p * ((p + eps) / gauss[None]).log() * time_mask[None, :, :, None]
        return {"loss_sted": kl.sum(-1).mean((1, 2))}

    def _guided(self, w, positive_map, time_mask):  # w (Lyr, b, T, T)
        excl = positive_map + (~time_mask)
        loss = (-(1 - w + 1e-6).log()).masked_fill(excl[None, :, :, None], 0)
        nb_neg = (~excl).sum(1) + 1e-6
        return {"loss_guided_attn": (loss.sum(3) / nb_neg[None, :, None]).sum(2).mean(1)}

    # ---- reference-style single-layer entry points (kept for API parity) ----
    def loss_boxes(self, outputs, targets, num_boxes):
        tgt = torch.cat([t["boxes"] for t in targets], dim=0)
        return {k: v[0] for k, v in self._boxes(outputs["pred_boxes"][None], tgt, max(num_boxes, 1) if not torch.is_tensor(num_boxes) else num_boxes).items()}

    def loss_sted(self, outputs, num_boxes, inter_idx, positive_map, time_mask=None):
        return {k: v[0] for k, v in self._sted(outputs["pred_sted"][None], inter_idx, time_mask).items()}

    def loss_guided_attn(self, outputs, num_boxes, inter_idx, positive_map, time_mask=None):
        return {k: v[0] for k, v in self._guided(outputs["weights"][None], positive_map, time_mask).items()}

    def get_loss(self, loss, outputs, targets, num_boxes, inter_idx, positive_map, time_mask, **kw):
        if loss == "boxes":
            return self.loss_boxes(outputs, targets, num_boxes)
        if loss == "sted":
            return self.loss_sted(outputs, num_boxes, inter_idx, positive_map, time_mask)
        if loss == "guided_attn":
            return self.loss_guided_attn(outputs, num_boxes, inter_idx, positive_map, time_mask)
        raise AssertionError(f"do you really want to compute {loss} loss?")

    def _num_boxes(self, n_local, dev):
        if self.external_num_boxes is not None:
            # the caller keeps clamp(all_reduce(#boxes)/world, 1) in a device scalar (tubedetr_amd.distributed.
            # sync_num_boxes) so that the step itself holds no collective (HIP-graph replay on every rank)
            return self.external_num_boxes
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            nb = torch.as_tensor([n_local], dtype=torch.float, device=dev)
            torch.distributed.all_reduce(nb)
            # stays a device scalar: no .item() host sync in the step (the reference syncs here, tubedetr.py:413)
            return torch.clamp(nb / torch.distributed.get_world_size(), min=1)[0]
        return float(max(n_local, 1))

    def _positive_map(self, inter_idx, time_mask):
        key = (tuple(map(tuple, inter_idx)), tuple(time_mask.shape), str(time_mask.device))
        positive_map = self._pm_cache.get(key)
        if positive_map is None:
            pm = torch.zeros(time_mask.shape, dtype=torch.bool)
            for kk, idx in enumerate(inter_idx):
                if idx[0] >= 0:
                    pm[kk, idx[0] : idx[1] + 1] = True
            positive_map = self._pm_cache[key] = pm.to(time_mask.device)
        return positive_map

    def weight_matrix(self, weight_dict, nl: int, device) -> torch.Tensor:
        """[layers, 4] coefficients of ``weight_dict`` in the layout of ``forward_fused``'s loss matrix (0 where a loss has
        no weight), cached on the device: the weighted total is then one multiply + one sum."""
        key = ("wm", tuple(sorted(weight_dict.items())), nl, str(device), frozenset(self._fused_keys))  # (the fused key set follows forward_fused's aux flag)
        m = self._pm_cache.get(key)
        if m is None:
            rows = []
            for l in range(nl):
                sfx = "" if l == nl - 1 else f"_{l}"
                rows.append([float(weight_dict.get(c + sfx, 0.0)) if (c + sfx) in self._fused_keys else 0.0 for c in LOSS_COLUMNS])
            m = self._pm_cache[key] = torch.tensor(rows, dtype=torch.float32).to(device)
        return m

    def forward_fused(self, stacked, keep, tgt_boxes, inter_idx, time_mask, aux: bool = True):
        """Same 24 values as ``forward`` from the decoder's stacked outputs (boxes [layers, b*t, 4] of every frame + the
        keep indices, sted [layers, b, t, 2], weights [layers, b, t, t]) in ONE kernel launch.  Returns the reference's
        loss dict; ``self.last_loss_matrix`` ([layers, 4], differentiable) holds the same numbers for a fused weighted sum.
        ``aux=False`` (a model built without --aux_loss): only the last layer's keys, like the reference's loop over
        ``outputs["aux_outputs"]`` (tubedetr.py:434-458) that then never runs."""
        boxes = stacked["pred_boxes"]
        sted = stacked["pred_sted"] if "sted" in self.losses else None
        weights = stacked["weights"] if "guided_attn" in self.losses else None
        dev = boxes.device
        nl = boxes.shape[0]
        b = len(inter_idx)
        num_boxes = self._num_boxes(int(tgt_boxes.shape[0]), dev)
        positive_map = self._positive_map(inter_idx, time_mask)
        key = ("fz", tuple(map(tuple, inter_idx)), tuple(time_mask.shape), str(dev))
        hit = self._tgt_cache.get(key)
        if hit is None:
            hit = self._tgt_cache[key] = (time_mask.to(torch.uint8).contiguous(), positive_map.to(torch.uint8).contiguous(),
                                          torch.tensor([[int(x[0]), int(x[1])] for x in inter_idx], dtype=torch.int32).to(dev))
        tm_u8, pm_u8, inter_dev = hit
        L = CriterionFn.apply(boxes, sted, weights, tgt_boxes, keep, tm_u8, pm_u8, inter_dev, num_boxes, float(self.sigma))
        self.last_loss_matrix = L
        losses, self._fused_keys = {}, set()
        for l in (range(nl) if aux else (nl - 1,)):
            sfx = "" if l == nl - 1 else f"_{l}"
            for j, c in enumerate(LOSS_COLUMNS):
                if (c in ("loss_bbox", "loss_giou") and "boxes" in self.losses) or (c == "loss_sted" and sted is not None) or (c == "loss_guided_attn" and weights is not None):
                    losses[c + sfx] = L[l, j]
                    self._fused_keys.add(c + sfx)
        return losses

    def forward(self, outputs, targets, inter_idx=None, time_mask=None):
        dev = next(iter(outputs.values())).device
        if torch.is_tensor(targets):  # already concatenated (n, 4) target boxes
            tgt_boxes, n_local = targets, targets.shape[0]
        else:
            n_local = sum(len(t["boxes"]) for t in targets)
            tgt_boxes = torch.cat([t["boxes"] for t in targets], dim=0) if "boxes" in self.losses else None
        num_boxes = self._num_boxes(n_local, dev)
        positive_map = None
        if inter_idx is not None and time_mask is not None:
            positive_map = self._positive_map(inter_idx, time_mask)
        aux = outputs.get("aux_outputs", [])
        layers = list(aux) + [outputs]  # main output last
        stacked = {}
        if "boxes" in self.losses:
            stacked.update(self._boxes(torch.stack([o["pred_boxes"] for o in layers]), tgt_boxes, num_boxes))
        if "sted" in self.losses:
            stacked.update(self._sted(torch.stack([o["pred_sted"] for o in layers]), inter_idx, time_mask))
        if "guided_attn" in self.losses:
            stacked.update(self._guided(torch.stack([o["weights"] for o in layers]), positive_map, time_mask))
        losses = {}
        for k, v in stacked.items():
            parts = v.unbind(0)
            losses[k] = parts[-1]
            for i in range(len(aux)):
                losses[f"{k}_{i}"] = parts[i]
        return losses


def build(args):
    device = torch.device(args.device)
    backbone = build_backbone(args)
    transformer = build_transformer(args)
    model = TubeDETR(backbone, transformer, num_queries=args.num_queries, aux_loss=args.aux_loss, video_max_len=args.video_max_len_train,
                     stride=args.stride, guided_attn=args.guided_attn, fast=args.fast, fast_mode=args.fast_mode, sted=args.sted,
                     compute_dtype=getattr(args, "compute_dtype", torch.float32))
    weight_dict = {"loss_bbox": args.bbox_loss_coef, "loss_giou": args.giou_loss_coef, "loss_sted": args.sted_loss_coef}
    if args.guided_attn:
        weight_dict["loss_guided_attn"] = args.guided_attn_loss_coef
    if args.aux_loss:
        base = dict(weight_dict)
        for i in range(args.dec_layers - 1):
            weight_dict.update({f"{k}_{i}": v for k, v in base.items()})
    losses = ["boxes", "sted"] if args.sted else ["boxes"]
    if args.guided_attn:
        losses += ["guided_attn"]
    criterion = SetCriterion(losses=losses, sigma=args.sigma)
    criterion.to(device)
    return model, criterion, weight_dict
