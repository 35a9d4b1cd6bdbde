"""ResNet-101 + FrozenBatchNorm2d backbone behind the reference's ``models.backbone`` interface
(models/backbone.py:20-124,220-257; torchvision resnet101 body, third party).

Module tree and parameter names are the reference's (``backbone.0.body.layer3.5.conv2.weight`` ...), so
checkpoints and the optimizer's name-based parameter groups keep working.  The computation is NOT a module
graph: the whole trunk is one autograd Function (``ResNetTrunkFn``) that walks a static plan and launches the
implicit-GEMM HIP kernels directly on NHWC activations:

  * FrozenBatchNorm2d is folded into the prepared weights (scale) and the GEMM epilogue (bias); ReLU and the
    residual add live in the same epilogue.  No standalone BN / ReLU / add kernels exist.
  * backward: the ReLU mask of the producing layer is applied in the dgrad GEMM epilogue, the identity /
    downsample branch gradients are accumulated through the residual operand, and weight gradients are un-folded
    (x scale, back to [Co,Ci,R,S]) by one finalize kernel each.  Stem + layer1 are frozen (backbone.py:82-89): no
    gradient kernels run for them and nothing is saved under ``torch.no_grad()`` (the "fast" pass).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import List

import torch
from torch import nn
from torch.autograd import Function

from .. import ops
from ..functional import prepared
from ..util.misc import NestedTensor
from .position_encoding import build_position_encoding

RESNET_LAYERS = {"resnet50": (3, 4, 6, 3), "resnet101": (3, 4, 23, 3)}


class FrozenBatchNorm2d(nn.Module):
    """Buffers only (weight, bias, running_mean, running_var); eps = 1e-5 (backbone.py:20-70).  Never executed as
    a layer: ``fold()`` hands the buffers to the weight-prep kernel."""

    def __init__(self, n: int):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        state_dict.pop(prefix + "num_batches_tracked", None)  # checkpoints of nn.BatchNorm2d carry it
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    def fold(self):
        return (self.weight, self.bias, self.running_mean, self.running_var)

    def forward(self, x):  # pragma: no cover - kept for API completeness
        raise RuntimeError("FrozenBatchNorm2d is folded into the conv kernels; it is not called as a layer")


class ConvWeight(nn.Module):
    """Parameter holder with nn.Conv2d's ``weight`` name/shape/init (kaiming-normal fan_out like torchvision)."""

    def __init__(self, cin: int, cout: int, k: int, stride: int = 1, padding: int = 0):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")
        self.k, self.stride, self.padding, self.cin, self.cout = k, stride, padding, cin, cout


class Bottleneck(nn.Module):
    def __init__(self, inplanes: int, planes: int, stride: int, downsample: bool):
        super().__init__()
        self.conv1 = ConvWeight(inplanes, planes, 1)
        self.bn1 = FrozenBatchNorm2d(planes)
        self.conv2 = ConvWeight(planes, planes, 3, stride, 1)
        self.bn2 = FrozenBatchNorm2d(planes)
        self.conv3 = ConvWeight(planes, planes * 4, 1)
        self.bn3 = FrozenBatchNorm2d(planes * 4)
        self.stride = stride
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential(ConvWeight(inplanes, planes * 4, 1, stride), FrozenBatchNorm2d(planes * 4))


class ResNetBody(nn.Module):
    """conv1, bn1, layer1..layer4 with torchvision's child names (what IntermediateLayerGetter keeps)."""

    def __init__(self, layers=(3, 4, 23, 3)):
        super().__init__()
        self.conv1 = ConvWeight(3, 64, 7, 2, 3)
        self.bn1 = FrozenBatchNorm2d(64)
        inplanes = 64
        for i, (planes, n, stride) in enumerate(zip((64, 128, 256, 512), layers, (1, 2, 2, 2))):
            blocks = []
            for j in range(n):
                blocks.append(Bottleneck(inplanes, planes, stride if j == 0 else 1, j == 0))
                inplanes = planes * 4
            setattr(self, f"layer{i + 1}", nn.Sequential(*blocks))
        self.out_channels = inplanes

    def blocks(self):
        for li in range(1, 5):
            for j, blk in enumerate(getattr(self, f"layer{li}")):
                yield f"layer{li}.{j}", blk

    def trainable_weights(self) -> List[nn.Parameter]:
        return [p for p in self.parameters() if p.requires_grad]

    def forward(self, x: torch.Tensor, compute_dtype: torch.dtype) -> torch.Tensor:
        """x (N,3,H,W) fp32 NCHW -> layer4 features [N,h,w,2048] NHWC in ``compute_dtype``."""
        tw = self.trainable_weights() if torch.is_grad_enabled() else []
        return ResNetTrunkFn.apply(self, x, compute_dtype, *tw)


def _prep(conv: ConvWeight, bn: FrozenBatchNorm2d, dt, need_dgrad, cpad=None):
    return prepared(conv.weight, dt, bn=bn.fold(), need_dgrad=need_dgrad, cpad=cpad)


class ResNetTrunkFn(Function):
    @staticmethod
    def forward(ctx, body: ResNetBody, x: torch.Tensor, dt: torch.dtype, *trainable):
        assert x.dim() == 4 and x.shape[1] == 3, "frames must be (N,3,H,W)"
        vec = ops.vec_of(dt)
        save = len(trainable) > 0
        xn = ops.nchw_to_nhwc(x.detach().float().contiguous(), dt, vec)
        wf, _, b, _ = _prep(body.conv1, body.bn1, dt, False, cpad=vec)
        y = ops.conv_fwd(xn, wf, b, 7, 7, 2, 3, relu=True)
        y = ops.maxpool3x3s2(y)
        saved = []  # per trainable block: (x_in, h1, h2, out)
        for name, blk in body.blocks():
            tw_ = blk.conv1.weight.requires_grad  # one prepared-weight cache entry serves the grad and no_grad passes
            train = tw_ and save
            w1, _, b1, _ = _prep(blk.conv1, blk.bn1, dt, tw_)
            w2, _, b2, _ = _prep(blk.conv2, blk.bn2, dt, tw_)
            w3, _, b3, _ = _prep(blk.conv3, blk.bn3, dt, tw_)
            h1 = ops.conv_fwd(y, w1, b1, 1, 1, 1, 0, relu=True)
            h2 = ops.conv_fwd(h1, w2, b2, 3, 3, blk.stride, 1, relu=True)
            idt = y
            if blk.downsample is not None:
                wd_, _, bd, _ = _prep(blk.downsample[0], blk.downsample[1], dt, tw_)
                idt = ops.conv_fwd(y, wd_, bd, 1, 1, blk.stride, 0)
            out = ops.conv_fwd(h2, w3, b3, 1, 1, 1, 0, residual=idt, relu=True)
            if train:
                saved.append((y, h1, h2, out))
            y = out
        ctx.body, ctx.dt, ctx.saved = body, dt, saved
        return y

    @staticmethod
    def backward(ctx, dfeat):
        body, dt, saved = ctx.body, ctx.dt, ctx.saved
        blocks = [(n, b) for n, b in body.blocks() if b.conv1.weight.requires_grad]
        assert len(blocks) == len(saved)
        grads = {}

        def wgrad(g, xin, conv: ConvWeight, bn: FrozenBatchNorm2d):
            _, _, _, scale = _prep(conv, bn, dt, True)
            dwk = ops.conv_wgrad(g, xin, conv.k, conv.k, conv.stride, conv.padding)
            grads[id(conv.weight)] = ops.wgrad_finalize(dwk, scale, tuple(conv.weight.shape), conv.cin)

        g_out = ops.relu_bwd(dfeat.contiguous(), saved[-1][3])
        for bi in range(len(blocks) - 1, -1, -1):
            _, blk = blocks[bi]
            x_in, h1, h2, _out = saved[bi]
            _, w3d, _, _ = _prep(blk.conv3, blk.bn3, dt, True)
            _, w2d, _, _ = _prep(blk.conv2, blk.bn2, dt, True)
            wgrad(g_out, h2, blk.conv3, blk.bn3)
            g_h2 = ops.conv_dgrad(g_out, w3d, h2.shape[1:3], 1, 1, 1, 0, mask_src=h2)
            wgrad(g_h2, h1, blk.conv2, blk.bn2)
            g_h1 = ops.conv_dgrad(g_h2, w2d, h1.shape[1:3], 3, 3, blk.stride, 1, mask_src=h1)
            wgrad(g_h1, x_in, blk.conv1, blk.bn1)
            if blk.downsample is not None:
                wgrad(g_out, x_in, blk.downsample[0], blk.downsample[1])
            if bi == 0:
                break  # the first trainable block's input comes from frozen layers
            _, w1d, _, _ = _prep(blk.conv1, blk.bn1, dt, True)
            if blk.downsample is not None:
                _, wdd, _, _ = _prep(blk.downsample[0], blk.downsample[1], dt, True)
                dx = ops.conv_dgrad(g_h1, w1d, x_in.shape[1:3], 1, 1, 1, 0, mask_src=x_in)
                if blk.stride == 1:
                    dx = ops.conv_dgrad(g_out, wdd, x_in.shape[1:3], 1, 1, 1, 0, residual=dx, mask_src=x_in)
                else:
                    ops.conv1x1s_dgrad_scatter(g_out, wdd, dx, blk.stride, mask_src=x_in)
            else:
                dx = ops.conv_dgrad(g_h1, w1d, x_in.shape[1:3], 1, 1, 1, 0, residual=g_out, mask_src=x_in)
            g_out = dx
        ctx.saved = None
        tw = body.trainable_weights()
        return (None, None, None) + tuple(grads.get(id(p)) for p in tw)


class BackboneBase(nn.Module):
    """Freeze rule of backbone.py:82-89: only parameters whose name contains layer2/layer3/layer4 train."""

    def __init__(self, body: ResNetBody, train_backbone: bool, num_channels: int, return_interm_layers: bool):
        super().__init__()
        if return_interm_layers:
            raise NotImplementedError("intermediate layers are not on the TubeDETR hot path")
        for name, p in body.named_parameters():
            if not train_backbone or ("layer2" not in name and "layer3" not in name and "layer4" not in name):
                p.requires_grad_(False)
        self.body = body
        self.num_channels = num_channels
        self.compute_dtype = torch.float32

    def forward(self, tensor_list: NestedTensor):
        feat = self.body(tensor_list.tensors, self.compute_dtype)  # [N,h,w,C] NHWC
        n, h, w, _ = feat.shape
        m = tensor_list.mask
        # F.interpolate(mode="nearest") index rule: floor(dst * float32(in/out))  (backbone.py:101-103)
        iy = _nearest_index(h, m.shape[-2], m.device)
        ix = _nearest_index(w, m.shape[-1], m.device)
        mask = m[:, iy][:, :, ix]
        out = OrderedDict()
        out[0] = NestedTensor(feat.permute(0, 3, 1, 2), mask)  # (N,C,h,w) view, channels-last strides
        return out


_IDX_CACHE: dict = {}


def _nearest_index(out_size: int, in_size: int, device) -> torch.Tensor:
    key = (out_size, in_size, str(device))
    if key not in _IDX_CACHE:
        scale = torch.tensor(in_size / out_size, dtype=torch.float32)
        idx = torch.floor(torch.arange(out_size, dtype=torch.float32) * scale).long().clamp_(max=in_size - 1)
        _IDX_CACHE[key] = idx.to(device)
    return _IDX_CACHE[key]


class Backbone(BackboneBase):
    """ResNet backbone with frozen BatchNorm (random init: no pretrained files offline)."""

    def __init__(self, name: str, train_backbone: bool, return_interm_layers: bool, dilation: bool, layers=None):
        if dilation:
            raise NotImplementedError("--dilation is outside the HIP hot path")
        if name not in RESNET_LAYERS:
            raise NotImplementedError(f"backbone {name}: only the bottleneck ResNets are implemented")
        body = ResNetBody(layers or RESNET_LAYERS[name])
        super().__init__(body, train_backbone, body.out_channels, return_interm_layers)


class Joiner(nn.Sequential):
    def __init__(self, backbone, position_embedding):
        super().__init__(backbone, position_embedding)

    def set_compute_dtype(self, dt: torch.dtype):
        self[0].compute_dtype = dt
        self[1].compute_dtype = dt

    def forward(self, tensor_list: NestedTensor):
        xs = self[0](tensor_list)
        out, pos = [], []
        for _, x in xs.items():
            out.append(x)
            pos.append(self[1](x))
        return out, pos


def build_backbone(args):
    position_embedding = build_position_encoding(args)
    train_backbone = args.lr_backbone > 0
    backbone = Backbone(args.backbone, train_backbone, False, args.dilation, layers=getattr(args, "resnet_layers", None))
    model = Joiner(backbone, position_embedding)
    if getattr(args, "freeze_backbone", False):
        for p in model.parameters():
            p.requires_grad_(False)
    model.num_channels = backbone.num_channels
    return model
