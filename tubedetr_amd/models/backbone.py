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

import os

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
        self.layers = tuple(layers)
        self.split_backward = False
        self._split = None
        # normalisation applied on the device to uint8 input frames (datasets' T.Normalize with the ImageNet statistics)
        self.pixel_mean, self.pixel_std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)

    def blocks(self):
        for li in range(1, 5):
            for j, blk in enumerate(getattr(self, f"layer{li}")):
                yield f"layer{li}.{j}", blk

    def trainable_weights(self) -> List[nn.Parameter]:
        return [p for p in self.parameters() if p.requires_grad]

    def forward(self, x: torch.Tensor, compute_dtype: torch.dtype, n_grad=None) -> torch.Tensor:
        """x (N,3,H,W) fp32 NCHW -> layer4 features [N,h,w,2048] NHWC in ``compute_dtype``.  ``n_grad``: only the first
        n_grad frames are back-propagated (the rest are the reference's no_grad "fast" frames run in the same pass)."""
        feat, rest = self.forward_split(x, compute_dtype, n_grad)
        return feat if rest.shape[0] == 0 else torch.cat([feat, rest])  # (callers that want both parts use forward_split: no copy)

    @staticmethod
    def max_frames(H: int, W: int, dt: torch.dtype) -> int:
        """Frames one trunk pass can address: every activation goes through a 32-bit buffer descriptor (< 2^31 elements and < 4 GiB
        per tensor) and the largest one has 16 * H * W elements per frame (stem output / layer1 output)."""
        n = max(1, int(min(2**31 - 1, (2**32 - 4096) // dt.itemsize) // (16 * H * W)))
        return int(os.environ.get("TD_TRUNK_MAX_FRAMES", n))  # (tests lower it to exercise the chunking on small frames)

    def forward_split(self, x, compute_dtype: torch.dtype, n_grad=None):
        """-> (features of the first n_grad frames, differentiable; features of the remaining frames, not differentiable): two views
        of the pass's own workspace.  One tensor sliced afterwards would make autograd zero-fill and copy a full-size gradient
        (0.5 GB at 1 000 frames) for the slice's backward."""
        tw = self.trainable_weights() if torch.is_grad_enabled() else []
        feat, rest = ResNetTrunkFn.apply(self, x, compute_dtype, x.shape[0] if n_grad is None else int(n_grad), *tw)
        if self.split_backward and tw:
            # cut the autograd graph at the trunk output: loss.backward() then stops here (stage 1: decoder, encoder, text
            # encoder, input_proj) and backward_trunk() runs the trunk's backward as a separate stage - the gradient
            # exchange of everything else overlaps it (tubedetr_amd.harness.backward_in_stages)
            leaf = feat.detach().requires_grad_()
            self._split = (feat, leaf)
            return leaf, rest
        return feat, rest

    def backward_trunk(self, after_stage=None) -> bool:
        """Second backward stage of a split step; returns False when there is nothing to do.  ``after_stage(stage, weights)`` (optional): the
        pass is issued STAGE BY STAGE (layer4, layer3, layer2 - td_resnet_bwd's ``only_stage``), each stage's weight gradients leave in a
        batched launch of their own, are accumulated into ``.grad`` and handed to the callback - which may start their data-parallel exchange
        while the remaining stages are still computed (what DDP's bucketed reducer does for the reference, main.py:372-376)."""
        if self._split is None:
            return False
        feat, leaf = self._split
        self._split = None
        if leaf.grad is None:
            return False
        if after_stage is None:
            feat.backward(leaf.grad)
            return True
        self._split = (feat, leaf)
        for stage, ws_ in self.backward_trunk_iter():
            after_stage(stage, ws_)
        return True

    def backward_trunk_iter(self):
        """Generator form of ``backward_trunk(after_stage=...)``: every ``next()`` enqueues ONE stage (3 = layer4, 2, ..) and yields
        (stage, that stage's weights with ``.grad`` set) - a caller that replays the step from HIP graphs captures each stage in a graph of
        its own and launches the stage's exchange between the replays."""
        if self._split is None:
            return
        feat, leaf = self._split
        self._split = None
        if leaf.grad is None:
            return
        ctx = feat.grad_fn  # the autograd node of ResNetTrunkFn IS its ctx
        assert ctx is not None and hasattr(ctx, "dims"), "backward_trunk_iter: the trunk pass kept nothing for backward"
        first = None
        for name, blk in self.blocks():
            if blk.conv1.weight.requires_grad:
                first = int(name[5]) - 1
                break
        by_w = {id(p): p for p in self.trainable_weights()}
        for stage in range(3, (4 if first is None else first) - 1, -1):
            with torch.no_grad():
                got = ResNetTrunkFn._enqueue(ctx, leaf.grad, stage)
                ws_ = []
                for wid, g in got.items():
                    p = by_w[wid]
                    p.grad = g if p.grad is None else p.grad.add_(g)
                    ws_.append(p)
            yield stage, ws_


def _prep(conv: ConvWeight, bn: FrozenBatchNorm2d, dt, need_dgrad, cpad=None):
    return prepared(conv.weight, dt, bn=bn.fold(), need_dgrad=need_dgrad, cpad=cpad)


def _conv_list(body: "ResNetBody"):
    """(conv, bn) pairs in the executor's order: stem, then per block conv1, conv2, conv3[, downsample]."""
    out = [(body.conv1, body.bn1)]
    for _, blk in body.blocks():
        out += [(blk.conv1, blk.bn1), (blk.conv2, blk.bn2), (blk.conv3, blk.bn3)]
        if blk.downsample is not None:
            out.append((blk.downsample[0], blk.downsample[1]))
    return out


def _ptr_array(tensors):
    import ctypes as C

    return (C.c_void_p * len(tensors))(*[t.data_ptr() if t is not None else None for t in tensors])


class ResNetTrunkFn(Function):
    """Whole trunk = one autograd node = one native executor call per direction (csrc/resnet_exec.hip)."""

    @staticmethod
    def forward(ctx, body: "ResNetBody", x: torch.Tensor, dt: torch.dtype, n_grad: int, *trainable):
        import ctypes as C

        from .. import _hip

        from ..util.misc import FrameSources

        L = _hip.lib()
        code = _hip.dtype_code(dt)
        vec = ops.vec_of(dt)
        save = 1 if len(trainable) > 0 else 0
        if not isinstance(x, FrameSources):  # one tensor of frames: fp32 (normalised, the reference's format) or uint8 pixels
            x = x.detach()
            x = FrameSources([((x if x.dtype == torch.uint8 else x.float()).contiguous(), None)])
        assert len(x.shape) == 4 and x.shape[1] == 3, "frames must be (N,3,H,W)"
        N, _, H, W = x.shape
        mean = inv_std = None
        if x.dtype == torch.uint8:  # the datasets' T.Normalize, on the device
            mean = (C.c_float * 3)(*body.pixel_mean)
            inv_std = (C.c_float * 3)(*[1.0 / v for v in body.pixel_std])
        convs = _conv_list(body)
        preps = [prepared(c.weight, dt, bn=bn.fold(), need_dgrad=c.weight.requires_grad, cpad=vec if i == 0 else None)
                 for i, (c, bn) in enumerate(convs)]
        nb = (C.c_int * 4)(*body.layers)
        w_ptrs = [p[0] for p in preps]
        # pixel-pair stem (include/tubedetr_hip.h): bf16, even width, frozen stem (its weight gradient would need the 8-channel frames)
        pairs = dt == torch.bfloat16 and W % 2 == 0 and not convs[0][0].weight.requires_grad and os.environ.get("TD_STEM_PAIRS", "1") != "0"
        if pairs:
            w_pairs = torch.empty((preps[0][0].shape[0], 7 * 4 * 8), dtype=dt, device=x.device)
            _hip.check(L.td_stem_pair_weights(preps[0][0].data_ptr(), w_pairs.data_ptr(), w_pairs.shape[0], code, _hip.stream_ptr()), "td_stem_pair_weights")
            w_ptrs = [w_pairs] + w_ptrs[1:]
        keep_alive = [t.detach().contiguous() for t, _ in x.parts]
        first_stage = 4  # first stage with trainable weights (the reference trains layer2-4: 1); the frozen ones below it may run fused
        for name, blk in body.blocks():
            if blk.conv1.weight.requires_grad:
                first_stage = int(name[5]) - 1
                break

        def sources(a: int, b: int):
            """td_frame_source array describing frames [a, b) of the concatenated source list (pointer arithmetic only)."""
            out, base = [], 0
            for t, (_, idx), vhw in zip(keep_alive, x.parts, x.valid):
                n_part = idx.numel() if idx is not None else t.shape[0]
                lo, hi = max(a, base) - base, min(b, base + n_part) - base
                base += n_part
                if hi <= lo:
                    continue
                fs = _hip.FrameSource()
                fs.dtype, fs.n = (_hip.TD_U8 if t.dtype == torch.uint8 else _hip.TD_F32), hi - lo
                if idx is not None:  # the index list is cut, the pixels (and their per-source-frame extents) stay where they are
                    fs.data, fs.index = t.data_ptr(), idx.data_ptr() + 4 * lo
                    fs.valid_hw = vhw.data_ptr() if vhw is not None else None
                else:
                    fs.data, fs.index = t.data_ptr() + lo * t.stride(0) * t.element_size(), None
                    fs.valid_hw = vhw.data_ptr() + 8 * lo if vhw is not None else None
                out.append(fs)
            arr = (_hip.FrameSource * len(out))(*out)
            return arr, len(out)

        # every activation of a pass is addressed through a 32-bit buffer descriptor: < 2^31 elements and < 4 GiB per tensor.
        # The largest one has 16 * H * W elements per frame (stem output / layer1 output).  A no-grad pass over more frames than
        # that (eval / the fast frames of a large batch) is still ONE td_resnet_fwd call: the executor issues the launches whose
        # operands would leave the range over equal frame groups (resnet_exec.hip: FrameGroups).  Only beyond FOUR times that
        # many frames - where the 4-channel input tensor itself leaves the range - is the pass cut into chunks here.  A pass
        # that keeps its activations for backward is not split (its backward walks ONE workspace): the C library reports the limit.
        n_max = 4 * ResNetBody.max_frames(H, W, dt)
        n_chunks = 1 if save else -(-N // n_max)
        if n_chunks > 1:
            step_n = -(-N // n_chunks)
            full, hw = None, (C.c_int * 3)()
            assert n_grad == N or not save
            for a in range(0, N, step_n):
                b = min(N, a + step_n)
                srcs, n_srcs = sources(a, b)
                nbytes = L.td_resnet_fwd_ws_bytes(b - a, H, W, nb, code, 0)
                ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
                feat_p = C.c_void_p()
                _hip.check(L.td_resnet_fwd(srcs, n_srcs, mean, inv_std, b - a, H, W, nb, _ptr_array(w_ptrs), _ptr_array([p[2] for p in preps]),
                                           0, ws.data_ptr(), nbytes, C.byref(feat_p), hw, int(pairs), first_stage, code, _hip.stream_ptr()), "td_resnet_fwd")
                off = feat_p.value - ws.data_ptr()
                n_el = (b - a) * hw[0] * hw[1] * hw[2]
                if full is None:  # the chunks' features go straight into their rows of ONE result tensor (no clone + torch.cat: 1.6 GB less copied at 16 clips)
                    full = torch.empty((N, hw[0], hw[1], hw[2]), dtype=dt, device=x.device)
                full[a:b].copy_(ws[off : off + n_el * dt.itemsize].view(dt).view(b - a, hw[0], hw[1], hw[2]))
            return full[:n_grad], full[n_grad:]
        srcs, n_srcs = sources(0, N)
        nbytes = L.td_resnet_fwd_ws_bytes(N, H, W, nb, code, save)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        feat_p = C.c_void_p()
        hw = (C.c_int * 3)()
        _hip.check(L.td_resnet_fwd(srcs, n_srcs, mean, inv_std, N, H, W, nb, _ptr_array(w_ptrs), _ptr_array([p[2] for p in preps]),
                                   save, ws.data_ptr(), nbytes, C.byref(feat_p), hw, int(pairs), first_stage, code, _hip.stream_ptr()), "td_resnet_fwd")
        off = feat_p.value - ws.data_ptr()
        n_el = N * hw[0] * hw[1] * hw[2]
        feat = ws[off : off + n_el * dt.itemsize].view(dt).view(N, hw[0], hw[1], hw[2])
        if not save:
            feat = feat.clone()  # let the ring workspace go
            return feat[:n_grad], feat[n_grad:]
        # two views of the workspace (kept alive by them and by ctx until backward): the frames that are back-propagated and the rest
        ctx.body, ctx.dt, ctx.ws, ctx.preps, ctx.dims = body, dt, ws, preps, (n_grad, N, H, W)
        a, b = feat[:n_grad], feat[n_grad:]
        ctx.mark_non_differentiable(b)
        return a, b

    @staticmethod
    def _enqueue(ctx, dfeat, only_stage=-1):
        """Enqueue the trunk's backward (td_resnet_bwd) - the whole pass, or the launches of ONE stage (3 = layer4 .. first trainable stage;
        the caller walks them in that order).  Returns {id(weight): dW} of the convs whose gradient this call produced; the dW tensors of a
        pass are views of ONE zero-filled flat buffer kept on ctx."""
        import ctypes as C

        from .. import _hip, ops

        body, dt, ws, preps = ctx.body, ctx.dt, ctx.ws, ctx.preps
        N, N_fwd, H, W = ctx.dims
        L = _hip.lib()
        code = _hip.dtype_code(dt)
        convs = _conv_list(body)
        first_stage = 4
        for name, blk in body.blocks():
            if blk.conv1.weight.requires_grad:
                first_stage = int(name[5]) - 1
                break
        nb = (C.c_int * 4)(*body.layers)
        st = getattr(ctx, "bwd_state", None)
        if st is None:
            # every trainable conv's dW is a view of ONE zero-filled flat buffer: jobs that the batched launch splits along M accumulate
            # with atomics into zeros, and one fill replaces ~85 per-job fills per step (td_resnet_bwd: dW_prezeroed)
            sizes = [c.weight.numel() if c.weight.requires_grad else 0 for c, _ in convs]
            offs_ = [0]
            for n_ in sizes:
                offs_.append(offs_[-1] + (n_ + 63) // 64 * 64)
            flat = torch.zeros(offs_[-1], dtype=torch.float32, device=dfeat.device)
            dWs = [flat[o : o + n_].view_as(c.weight) if n_ else None for (c, _), n_, o in zip(convs, sizes, offs_)]
            nbytes = L.td_resnet_bwd_ws_bytes(N, H, W, nb, first_stage, code)
            bws = torch.empty(nbytes, dtype=torch.uint8, device=dfeat.device)
            tbytes = L.td_resnet_bwd_table_bytes(nb, first_stage)
            th, td_, done = ops.job_tables.take(tbytes, dfeat.device)  # weight-gradient job table: caller-owned staging
            st = ctx.bwd_state = dict(dWs=dWs, bws=bws, nbytes=nbytes, th=th, td=td_, tbytes=tbytes, done=done, dfeat=dfeat.contiguous(), first_stage=first_stage)
        _hip.check(L.td_resnet_bwd(st["dfeat"].data_ptr(), N, N_fwd, H, W, nb, first_stage, _ptr_array([p[1] for p in preps]),
                                   _ptr_array([p[3] for p in preps]), _ptr_array(st["dWs"]), ws.data_ptr(), st["bws"].data_ptr(), st["nbytes"],
                                   st["th"].data_ptr(), st["td"].data_ptr(), st["tbytes"], 1, code, int(only_stage), _hip.stream_ptr()), "td_resnet_bwd")
        # which convs belong to the stage(s) just issued: conv list order = stem, then per block conv1, conv2, conv3[, downsample]
        out, k = {}, 1
        for name, blk in body.blocks():
            n_c = 4 if blk.downsample is not None else 3
            stage = int(name[5]) - 1
            if stage >= first_stage and (only_stage == -1 or stage == only_stage):
                for (c, _), g in zip(convs[k : k + n_c], st["dWs"][k : k + n_c]):
                    if g is not None:
                        out[id(c.weight)] = g
            k += n_c
        if only_stage == -1 or only_stage == first_stage:  # the pass is complete
            st["done"]()
            ctx.ws = ctx.preps = None
        return out

    @staticmethod
    def backward(ctx, dfeat, _drest=None):
        by_id = ResNetTrunkFn._enqueue(ctx, dfeat, -1)
        return (None, None, None, None) + tuple(by_id.get(id(p)) for p in ctx.body.trainable_weights())


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

    @staticmethod
    def _mask_like(m: torch.Tensor, h: int, w: int) -> torch.Tensor:
        # F.interpolate(mode="nearest") index rule: floor(dst * float32(in/out))  (backbone.py:101-103)
        iy = _nearest_index(h, m.shape[-2], m.device)
        ix = _nearest_index(w, m.shape[-1], m.device)
        return m[:, iy][:, :, ix]

    def forward(self, tensor_list: NestedTensor, n_grad=None):
        feat = self.body(tensor_list.tensors, self.compute_dtype, n_grad)  # [N,h,w,C] NHWC
        n, h, w, _ = feat.shape
        out = OrderedDict()
        out[0] = NestedTensor(feat.permute(0, 3, 1, 2), self._mask_like(tensor_list.mask, h, w))  # (N,C,h,w) view, channels-last strides
        return out

    def forward_split(self, frames, n_grad: int, mask_grad: torch.Tensor, mask_rest: torch.Tensor):
        """The slow (back-propagated) and fast (no_grad, tubedetr.py:128-129) frames of a step in ONE trunk pass, returned apart:
        (NestedTensor of the first n_grad frames, NestedTensor of the rest).  The pad masks are down-sampled separately (the
        full-resolution masks of 1 000 frames are 124 MB: nothing concatenates them)."""
        a, b = self.body.forward_split(frames, self.compute_dtype, n_grad)
        h, w = a.shape[1], a.shape[2]
        return (NestedTensor(a.permute(0, 3, 1, 2), self._mask_like(mask_grad, h, w)),
                NestedTensor(b.permute(0, 3, 1, 2), self._mask_like(mask_rest, h, w)))


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

    def forward(self, tensor_list: NestedTensor, n_grad=None, want_pos: bool = True):
        """-> (features, pos) like the reference (backbone.py:225-233).  ``want_pos=False`` (TubeDETR's own call when the
        encoding is the sine one): pos entries are None - the transformer then produces the positional operand itself, straight
        from the pad mask and already extended by the zero rows of the text tokens (td_pos_sine), instead of receiving a
        tensor it would have to concatenate zeros to."""
        xs = self[0](tensor_list, n_grad)
        out, pos = [], []
        for _, x in xs.items():
            out.append(x)
            pos.append(self[1](x) if want_pos else None)
        return out, pos

    def forward_split(self, frames, n_grad: int, mask_grad, mask_rest, want_pos: bool = True):
        """-> (slow features, their pos or None, fast features): BackboneBase.forward_split + the positional encoding of the slow part."""
        slow, fast = self[0].forward_split(frames, n_grad, mask_grad, mask_rest)
        return slow, (self[1](slow) if want_pos else None), fast


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
