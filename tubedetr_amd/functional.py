"""torch.autograd.Function wrappers: forward AND backward of every op on the hot path run through the
hand-written gfx950 kernels (ops.py -> C ABI).  No torch compute kernels are used for the math here;
torch only owns the memory and the autograd graph.

All activation tensors are 2-D row-major [rows, channels] (batch-major tokens) or NHWC, in the compute
dtype of the model (torch.float32 = exact-fp32 parity mode, torch.bfloat16 = throughput mode).
Parameters stay fp32 (the reference's state_dict); per-step prepared copies (cast / transposed /
FrozenBN-folded) are cached on the parameter's version counter.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch.autograd import Function

from . import ops

Tensor = torch.Tensor

_EPOCH = [0]


def invalidate_prepared():
    """Drop every prepared (cast / transposed / BN-folded) weight copy, as an optimizer step would by bumping the
    parameters' version counters.  bench.py calls this once per step so that the per-step weight preparation is part
    of the measured forward pass even though the benchmark itself never updates the weights."""
    _EPOCH[0] += 1


class _PrepEntry:
    __slots__ = ("base_ref", "offset", "shape", "bn", "dtype", "need_dgrad", "cpad", "co_alloc", "res", "ver", "dims")


class _PrepRegistry:
    """All prepared-weight requests of one (device, dtype): their output buffers persist (overwritten in place, in
    stream order, whenever the weights change) so the device-side item table is static and ONE td_weight_prep_batch
    launch refreshes every layer of the model."""

    def __init__(self):
        self.entries = []
        self.table_dev = None
        self.total_blocks = 0

    def _signature(self, e):
        base = e.base_ref()
        if base is None:
            return None
        return (_EPOCH[0], base._version, base.data_ptr()) + (tuple((b._version, b.data_ptr()) for b in e.bn) if e.bn is not None else ())

    @staticmethod
    def _fill(it, e, blk):
        base = e.base_ref()
        Co, Ci, RS = e.dims
        wf, wd, b_out, sc = e.res
        it.W = base.data_ptr() + e.offset * 4
        if e.bn is not None:
            it.bn_w, it.bn_b, it.bn_rm, it.bn_rv = (t.data_ptr() for t in e.bn)
        it.w_fwd = wf.data_ptr()
        it.w_dgrad = wd.data_ptr() if wd is not None else None
        it.bias_out = b_out.data_ptr() if b_out is not None else None
        it.scale_out = sc.data_ptr() if sc is not None else None
        it.Co, it.Ci, it.RS, it.Cpad, it.Co_alloc, it.blk0 = Co, Ci, RS, e.cpad, e.co_alloc, blk
        return ((e.co_alloc + 15) // 16) * ((e.cpad + 31) // 32)

    def _build_table(self, device):
        from . import _hip

        self.entries = [e for e in self.entries if e.base_ref() is not None]
        items = (_hip.PrepItem * len(self.entries))()
        blk = 0
        for it, e in zip(items, self.entries):
            blk += self._fill(it, e, blk)
        self.total_blocks = blk
        raw = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8)
        self.table_dev = raw.to(device)  # (re)uploaded only when the set of layers changes

    def refresh_one(self, e, device, dtype):
        """A layer seen for the first time while the others are current (the very first forward registers the layers
        one by one): prepare just that one instead of re-running the whole batch per registration."""
        from . import _hip

        items = (_hip.PrepItem * 1)()
        nblk = self._fill(items[0], e, 0)
        tab = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(device)
        _hip.check(_hip.lib().td_weight_prep_batch(tab.data_ptr(), 1, nblk, _hip.dtype_code(dtype), _hip.stream_ptr()), "td_weight_prep_batch")
        e.ver = self._signature(e)
        self._keep = getattr(self, "_keep", [])
        self._keep.append(tab)  # the launch reads the table asynchronously
        if len(self._keep) > 1024:
            del self._keep[:512]

    def refresh_all(self, device, dtype):
        from . import _hip

        if self.table_dev is None or any(e.base_ref() is None for e in self.entries):
            self._build_table(device)
        if not self.entries:
            return
        _hip.check(_hip.lib().td_weight_prep_batch(self.table_dev.data_ptr(), len(self.entries), self.total_blocks,
                                                   _hip.dtype_code(dtype), _hip.stream_ptr()), "td_weight_prep_batch")
        for e in self.entries:
            e.ver = self._signature(e)


_REGISTRIES: dict = {}


def prepared(W: Tensor, dtype: torch.dtype, *, bn=None, need_dgrad: bool = True, cpad: Optional[int] = None, pad_out: int = 0):
    """(w_fwd, w_dgrad, bias_fold, scale) for parameter W (or a row-slice view of one).  The request is cached ON the
    parameter object (so a recycled device address can never alias a stale entry) and registered in a per-(device,
    dtype) registry: when any weight has changed, one batched launch refreshes every registered layer."""
    import weakref

    base = W._base if W._base is not None else W
    cache = base.__dict__.setdefault("_td_prepared", {})
    key = (W.storage_offset(), tuple(W.shape), dtype, need_dgrad, cpad, pad_out)
    e = cache.get(key)
    reg = _REGISTRIES.setdefault((str(W.device), dtype), _PrepRegistry())
    if e is None:
        e = _PrepEntry()
        if W.dim() == 2:
            Co, Ci = W.shape
            RS = 1
        else:
            Co, Ci, R, S_ = W.shape
            RS = R * S_
        assert W.dtype == torch.float32 and W.is_contiguous()
        # element offset of W inside `base` (NOT inside the storage: once FusedAdamWEMA has moved the parameters into its flat
        # buffer, base.data_ptr() itself sits at a storage offset - counting that twice read far outside the weight)
        e.base_ref, e.offset, e.shape, e.bn, e.dtype = weakref.ref(base), W.storage_offset() - base.storage_offset(), tuple(W.shape), bn, dtype
        e.need_dgrad, e.dims = need_dgrad, (Co, Ci, RS)
        e.cpad = cpad if cpad is not None else ops.pad_to(Ci, ops.vec_of(dtype))
        e.co_alloc = max(Co, pad_out)
        dev = W.device
        wf = torch.empty((e.co_alloc, RS * e.cpad), dtype=dtype, device=dev)
        wd = torch.empty((Ci, RS * e.co_alloc), dtype=dtype, device=dev) if need_dgrad else None
        b_out = torch.empty(e.co_alloc, dtype=torch.float32, device=dev) if bn is not None else None
        sc = torch.empty(e.co_alloc, dtype=torch.float32, device=dev) if bn is not None else None
        e.res, e.ver = (wf, wd, b_out, sc), None
        cache[key] = e
        others_current = all(o.ver is not None and o.ver == reg._signature(o) for o in reg.entries[-4:])
        reg.entries.append(e)
        reg.table_dev = None  # table must be rebuilt
        if others_current and not torch.cuda.is_current_stream_capturing():
            reg.refresh_one(e, W.device, dtype)
    if e.ver != reg._signature(e):
        reg.refresh_all(W.device, dtype)
    return e.res


_SEED_STATE = {"torch_seed": None, "rng": None}


def _seed() -> int:
    """Fresh 31-bit dropout seed.  A host-side generator re-keyed from torch's CPU seed (so torch.manual_seed controls
    it) - drawing from the torch generator itself costs ~10 us per dropout site."""
    import random

    ts = torch.initial_seed()
    if _SEED_STATE["torch_seed"] != ts:
        _SEED_STATE["torch_seed"] = ts
        _SEED_STATE["rng"] = random.Random(ts)
    return _SEED_STATE["rng"].getrandbits(31)


# ---- deferred weight gradients -------------------------------------------------------------------------------------
# The weight / bias gradients of the transformer's ~94 linear layers do not feed the backward chain.  Launched where
# autograd reaches them they are 94 latency-bound launches with split-M fp32 atomics into zero-filled buffers; instead
# each backward only RESERVES its gradient tensors, queues (g, x, dW*, db*) and the whole queue runs as ONE
# td_conv_wgrad_batch launch from an autograd final callback (end of loss.backward(), before it returns, on the
# caller's stream): thousands of output tiles fill the chip, nothing needs zero-initialising, no atomics.
# Deferral is skipped (immediate launch) whenever the reserved tensor could be read before the callback runs: a
# parameter that already has a .grad (AccumulateGrad would add in place) or carries gradient hooks (DDP's reducer).
import os as _os
import threading as _threading

_DEFER_ON = [_os.environ.get("TD_WGRAD_DEFER", "1") != "0"]
_DEFER_LOCK = _threading.Lock()
_DEFER_JOBS: list = []
_ARMED = [False]  # the final callback of the running backward pass has been queued


def set_wgrad_deferral(on: bool) -> None:
    _DEFER_ON[0] = bool(on)


_USES: dict = {}  # id(leaf parameter) -> number of deferral-capable forward uses since the last backward


def _leaf(p):
    if p is None or p.is_leaf:
        return p
    return p._base if (p._base is not None and p._base.is_leaf) else None


def _note_use(recording: bool, *params) -> None:
    """Forward-side bookkeeping (``recording`` = this node is part of an autograd graph): a parameter that enters more than
    one node of the same graph (input_proj runs on the slow and on the fast features) has its gradients SUMMED by the
    engine before the final callback runs, so its nodes must produce them immediately."""
    _drop_stale_deferred()
    if not recording:
        return
    for p in params:
        q = _leaf(p)
        if q is not None and q.requires_grad:
            _USES[id(q)] = _USES.get(id(q), 0) + 1


def _can_defer(*params) -> bool:
    if not _DEFER_ON[0]:
        return False
    for p in params:
        if p is None:
            continue
        q = _leaf(p)
        if q is None:
            return False
        if not q.requires_grad:
            continue
        if _USES.get(id(q), 0) != 1 or q.grad is not None or q._backward_hooks or getattr(q, "_post_accumulate_grad_hooks", None):
            return False
    return True


def _flush_deferred_wgrads():
    with _DEFER_LOCK:
        jobs = list(_DEFER_JOBS)
        _DEFER_JOBS.clear()
        _USES.clear()
        _ARMED[0] = False
    if jobs:
        ops.linear_wgrad_batch(jobs)


def _drop_stale_deferred():
    """Called from the forward functions: a queue that is non-empty here belongs to a backward pass that died before its
    final callback ran; its reserved outputs are gone, so the jobs must never be launched."""
    if _ARMED[0]:
        with _DEFER_LOCK:
            _DEFER_JOBS.clear()
            _USES.clear()
            _ARMED[0] = False


def _arm():
    """First thing in every backward of the linear-layer functions: make sure the end-of-backward callback is queued
    (it also resets the per-step use counts when nothing was deferred)."""
    with _DEFER_LOCK:
        if _ARMED[0]:
            return
        _ARMED[0] = True
    torch.autograd.Variable._execution_engine.queue_callback(_flush_deferred_wgrads)


def _wgrad(g: Tensor, x: Tensor, defer: bool, *, want_bias: bool, out: Optional[Tensor] = None, dbias: Optional[Tensor] = None,
           x2: Optional[Tensor] = None):
    """(dW [N,K] fp32, db [N] fp32 | None) of a linear layer y = x W^T + b from g = dL/dy.  ``out`` / ``dbias``: row
    slices of a packed gradient (MHA in_proj) to write into.  defer=True: outputs are reserved now and filled by the
    batched launch at the end of backward.  ``x2``: second operand stream of a two-source layer y = (x + x2) W^T (the
    positional operand, never added in memory): dW = g^T x + g^T x2, the second product as an accumulating job."""
    M, K = x.shape
    N = g.shape[1]
    dev = g.device
    if not defer:
        db = dbias if dbias is not None else (ops.zeros_f32(N, dev) if want_bias else None)
        dW = ops.linear_wgrad(g, x, out=out, dbias=db)
        if x2 is not None:
            ops.linear_wgrad(g, x2, out=dW)
        return dW, db
    dW = out if out is not None else torch.empty((N, K), dtype=torch.float32, device=dev)
    db = dbias if dbias is not None else (torch.empty(N, dtype=torch.float32, device=dev) if want_bias else None)
    with _DEFER_LOCK:
        # only raw pointers of the outputs are kept: a second reference would stop AccumulateGrad from adopting the tensor
        _DEFER_JOBS.append((g, x, dW.data_ptr(), db.data_ptr() if db is not None else None))
        if x2 is not None:
            _DEFER_JOBS.append((g, x2, dW.data_ptr(), None, True))
    return dW, db


class LinearFn(Function):
    """y = dropout(act(x @ W^T + b)); act in {none, relu}.  x [M,K], W fp32 [N,K], b fp32 [N]."""

    @staticmethod
    def forward(ctx, x, W, b, relu: bool, dropout_p: float, seed: int):
        _note_use(ctx.needs_input_grad[1], W, b)
        N = W.shape[0]
        vec = ops.vec_of(x.dtype)
        Np = ops.pad_to(N, vec)
        wf, wd, _, _ = prepared(W, x.dtype, pad_out=Np if Np != N else 0)
        bias = b.detach() if b is not None else None
        if bias is not None and Np != N:
            bias = torch.cat([bias, bias.new_zeros(Np - N)])
        y = ops.linear_fwd(x, wf, bias, relu=relu, dropout_p=dropout_p, seed=seed)
        ctx.save_for_backward(x, y if (relu or dropout_p > 0) else None, wd)
        ctx.cfg = (relu, dropout_p, seed, N, Np, b is not None)
        ctx.params = (W, b)
        return y[:, :N].contiguous() if Np != N else y

    @staticmethod
    def backward(ctx, dy):
        _arm()
        x, y, wd = ctx.saved_tensors
        relu, p, seed, N, Np, has_b = ctx.cfg
        dy = dy.contiguous()
        if Np != N:
            g = dy.new_zeros((dy.shape[0], Np))
            g[:, :N] = dy
            dy = g
        if relu:
            g = ops.relu_bwd(dy, y, 1.0 / (1.0 - p) if p > 0 else 1.0)  # y > 0 <=> kept and active
        elif p > 0:
            g = ops.dropout(dy, p, seed)
        else:
            g = dy
        dx = ops.linear_fwd(g, wd) if ctx.needs_input_grad[0] else None
        want_w, want_b = ctx.needs_input_grad[1], has_b and ctx.needs_input_grad[2]
        dW = db = None
        if want_w:  # the bias gradient (column sums of g) rides along in the weight-gradient launch
            dW_full, db_full = _wgrad(g, x, _can_defer(*ctx.params), want_bias=want_b)
            dW = dW_full[:N]
            db = db_full[:N] if want_b else None
        elif want_b:
            db = ops.colsum(g)[:N]
        return dx, dW, db, None, None, None


def linear(x, W, b=None, relu=False, dropout_p=0.0, training=False):
    p = dropout_p if training else 0.0
    return LinearFn.apply(x, W, b, relu, p, _seed() if p > 0 else 0)


class FFNFn(Function):
    """y = dropout2(relu_dropout(x W1^T + b1) W2^T + b2)   (models/transformer.py:643,748)."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, p: float, seed1: int, seed2: int):
        _note_use(ctx.needs_input_grad[1], W1, b1, W2, b2)
        w1f, w1d, _, _ = prepared(W1, x.dtype)
        w2f, w2d, _, _ = prepared(W2, x.dtype)
        h = ops.linear_fwd(x, w1f, b1.detach(), relu=True, dropout_p=p, seed=seed1)
        y = ops.linear_fwd(h, w2f, b2.detach(), dropout_p=p, seed=seed2)
        ctx.save_for_backward(x, h, w1d, w2d)
        ctx.cfg = (p, seed1, seed2)
        ctx.params = (W1, b1, W2, b2)
        return y

    @staticmethod
    def backward(ctx, dy):
        _arm()
        x, h, w1d, w2d = ctx.saved_tensors
        p, seed1, seed2 = ctx.cfg
        defer = _can_defer(*ctx.params)
        g2 = ops.dropout(dy.contiguous(), p, seed2) if p > 0 else dy.contiguous()
        dW2, db2 = _wgrad(g2, h, defer, want_bias=True)
        # dh = (g2 @ W2) * (h > 0) / (1-p): mask + scale fused in the GEMM epilogue
        dh = ops.linear_fwd(g2, w2d, mask_src=h, alpha=1.0 / (1.0 - p) if p > 0 else 1.0)
        dW1, db1 = _wgrad(dh, x, defer, want_bias=True)
        dx = ops.linear_fwd(dh, w1d) if ctx.needs_input_grad[0] else None
        return dx, dW1, db1, dW2, db2, None, None, None


def ffn(x, W1, b1, W2, b2, dropout_p=0.0, training=False):
    p = dropout_p if training else 0.0
    return FFNFn.apply(x, W1, b1, W2, b2, p, _seed() if p > 0 else 0, _seed() if p > 0 else 0)


class AddLayerNormFn(Function):
    """y = LayerNorm(x + r) (r optional)."""

    @staticmethod
    def forward(ctx, x, r, gamma, beta, eps: float):
        y, s, mean, rstd = ops.add_layernorm_fwd(x, r, gamma.detach(), beta.detach(), eps)
        ctx.save_for_backward(s, mean, rstd, gamma.detach())
        ctx.has_r = r is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        s, mean, rstd, gamma = ctx.saved_tensors
        ds, dg, db = ops.add_layernorm_bwd(dy.contiguous(), s, mean, rstd, gamma)
        return ds, (ds if ctx.has_r else None), dg, db, None


def add_layernorm(x, r, gamma, beta, eps=1e-5):
    return AddLayerNormFn.apply(x, r, gamma, beta, eps)


class AddFn(Function):
    """a + b on the elementwise kernel (pos-enc add in front of the Q/K projections)."""

    @staticmethod
    def forward(ctx, a, b):
        return ops.add(a, b)

    @staticmethod
    def backward(ctx, g):
        return g, g


class MHAFn(Function):
    """nn.MultiheadAttention forward/backward (packed in_proj, 1/sqrt(hd) scaling, key padding mask, prob dropout,
    out_proj, head-averaged weights) on batch-major rows.

    q_in [B*Lq, E], k_in [B*Lk, E] (None = same tensor as q_in), v_in [B*Lk, E]; returns (out [B*Lq,E],
    wavg [B,Lq,Lk] fp32 or None).  The optional ``out_dropout`` is the residual-branch dropout that follows the
    attention in the reference layers (dropout1 / dropout3), fused into the out_proj epilogue.

    ``q_pos`` [B*Lq, E] (or None): the positional operand of the query (and, for self-attention, key) projection -
    ``with_pos_embed(x, pos)`` of the reference layers (transformer.py:637-640, 698, 735).  It is never added to x in memory:
    the projection GEMM reads it as a second operand stream against the same weight (td_linear_ex), its weight gradient is a
    second, accumulating job; the input gradient of the projection is the gradient of x AND of q_pos."""

    @staticmethod
    def forward(ctx, q_in, q_pos, k_in, v_in, W_in, b_in, W_out, b_out, key_pad, B, Lq, Lk, H, need_w, p_attn, seed_attn, p_out, seed_out):
        _note_use(ctx.needs_input_grad[4], W_in, b_in, W_out, b_out)
        E = q_in.shape[1]
        dt = q_in.dtype
        same_qk = k_in is None  # self-attention with q = k = x + pos: one fused [rows, 2E] projection
        bi = b_in.detach()

        def proj(x_, w_, b_):
            return ops.linear_ex(x_, w_, b_, a2=q_pos, w_shared=True) if q_pos is not None else ops.linear_fwd(x_, w_, b_)

        if same_qk:
            wqk_f, wqk_d, _, _ = prepared(W_in[: 2 * E], dt)
            qk = proj(q_in, wqk_f, bi[: 2 * E]).view(B, Lq, 2 * E)
            q, k = qk[..., :E], qk[..., E:]
            wd_list = (wqk_d,)
        else:
            wq_f, wq_d, _, _ = prepared(W_in[:E], dt)
            wk_f, wk_d, _, _ = prepared(W_in[E : 2 * E], dt)
            q = proj(q_in, wq_f, bi[:E]).view(B, Lq, E)
            k = ops.linear_fwd(k_in, wk_f, bi[E : 2 * E]).view(B, Lk, E)
            wd_list = (wq_d, wk_d)
        wv_f, wv_d, _, _ = prepared(W_in[2 * E :], dt)
        v = ops.linear_fwd(v_in, wv_f, bi[2 * E :]).view(B, Lk, E)
        scale = 1.0 / math.sqrt(E // H)
        lean = not need_w and ops.mha_lean_ok(q, k, v, H)  # nobody reads the weights: nothing of size Lq x Lk is stored
        if lean:
            ctxv, probs, kp = ops.mha_lean_fwd(q, k, v, key_pad, H, scale, dropout_p=p_attn, seed=seed_attn)  # probs := row statistics
            wavg = None
        else:
            ctxv, probs, wavg = ops.mha_fwd(q, k, v, key_pad, H, scale, need_wavg=need_w, dropout_p=p_attn, seed=seed_attn)
            kp = None
        wo_f, wo_d, _, _ = prepared(W_out, dt)
        out = ops.linear_fwd(ctxv.view(B * Lq, E), wo_f, b_out.detach(), dropout_p=p_out, seed=seed_out)
        ctx.save_for_backward(q_in, q_pos, k_in, v_in, q, k, v, probs, ctxv, wv_d, wo_d, *wd_list)
        ctx.lean, ctx.kp = lean, kp
        ctx.cfg = (B, Lq, Lk, H, E, same_qk, scale, p_attn, seed_attn, p_out, seed_out)
        ctx.params = (W_in, b_in, W_out, b_out)
        return out, (wavg if need_w else None)

    @staticmethod
    def backward(ctx, dout, dwavg):
        _arm()
        q_in, q_pos, k_in, v_in, q, k, v, probs, ctxv, wv_d, wo_d, *wd_list = ctx.saved_tensors
        B, Lq, Lk, H, E, same_qk, scale, p_attn, seed_attn, p_out, seed_out = ctx.cfg
        dt = q_in.dtype
        dev = q_in.device
        defer = _can_defer(*ctx.params)
        g = ops.dropout(dout.contiguous(), p_out, seed_out) if p_out > 0 else dout.contiguous()
        if defer:  # every row block is written by exactly one deferred job: no zero fill
            dW_in = torch.empty((3 * E, E), dtype=torch.float32, device=dev)
            db_in = torch.empty(3 * E, dtype=torch.float32, device=dev)
        else:
            dW_in = ops.zeros_f32((3 * E, E), dev)
            db_in = ops.zeros_f32(3 * E, dev)
        dW_out, db_out = _wgrad(g, ctxv.view(B * Lq, E), defer, want_bias=True)
        dctx = ops.linear_fwd(g, wo_d).view(B, Lq, E)
        if same_qk:
            dqk = torch.empty((B, Lq, 2 * E), dtype=dt, device=dev)
            dq, dk = dqk[..., :E], dqk[..., E:]
        else:
            dq = torch.empty((B, Lq, E), dtype=dt, device=dev)
            dk = torch.empty((B, Lk, E), dtype=dt, device=dev)
        dv = torch.empty((B, Lk, E), dtype=dt, device=dev)
        dwa = dwavg.contiguous().float() if dwavg is not None else None
        if ctx.lean:
            ops.mha_lean_bwd(q, k, v, ctx.kp, ctxv, dctx, probs, H, scale, dq, dk, dv, dropout_p=p_attn, seed=seed_attn)
        else:
            ops.mha_bwd(q, k, v, dctx, probs, dwa, H, scale, dq, dk, dv, dropout_p=p_attn, seed=seed_attn)
        dv2 = dv.view(B * Lk, E)
        _wgrad(dv2, v_in, defer, want_bias=True, out=dW_in[2 * E :], dbias=db_in[2 * E :])
        d_v_in = ops.linear_fwd(dv2, wv_d) if ctx.needs_input_grad[3] else None
        d_q_in = d_k_in = None
        need_q = ctx.needs_input_grad[0] or (q_pos is not None and ctx.needs_input_grad[1])
        if same_qk:
            dqk2 = dqk.view(B * Lq, 2 * E)
            _wgrad(dqk2, q_in, defer, want_bias=True, out=dW_in[: 2 * E], dbias=db_in[: 2 * E], x2=q_pos)
            if need_q:
                d_q_in = ops.linear_fwd(dqk2, wd_list[0])
        else:
            dq2, dk2 = dq.view(B * Lq, E), dk.view(B * Lk, E)
            _wgrad(dq2, q_in, defer, want_bias=True, out=dW_in[:E], dbias=db_in[:E], x2=q_pos)
            _wgrad(dk2, k_in, defer, want_bias=True, out=dW_in[E : 2 * E], dbias=db_in[E : 2 * E])
            if need_q:
                d_q_in = ops.linear_fwd(dq2, wd_list[0])
            if ctx.needs_input_grad[2]:
                d_k_in = ops.linear_fwd(dk2, wd_list[1])
        # d(x + pos) flows to both operands of the projection
        return (d_q_in if ctx.needs_input_grad[0] else None, d_q_in if (q_pos is not None and ctx.needs_input_grad[1]) else None, d_k_in, d_v_in,
                dW_in, db_in, dW_out, db_out) + (None,) * 10


class _CrossKVShared:
    """Side channel between CrossKVFn and the six cross-attention nodes that consume its output: the key / value gradients of
    all layers land in ONE [rows, layers * E] buffer each (every layer's attention backward writes its own column block),
    so the engine has nothing to accumulate and the hoisted node turns them into d(memory) with two chained GEMMs."""

    def __init__(self, mem, pos, n_layers, E):
        self.mem, self.pos, self.n_layers, self.E = mem, pos, n_layers, E
        self.dK = self.dV = None
        self.written = set()
        self.wk_d = self.wv_d = None  # concatenated input-gradient weights [E, layers * E]

    def grads(self, like):
        if self.dK is None:
            self.dK = torch.empty((self.mem.shape[0], self.n_layers * self.E), dtype=like.dtype, device=like.device)
            self.dV = torch.empty_like(self.dK)
        return self.dK, self.dV


class CrossKVFn(Function):
    """The key and value projections of ALL decoder layers' time-aligned cross-attention in two GEMMs (the memory is the same
    for the six layers, transformer.py:734-740): K_all = (mem + pos) [W_k,0 | ... | W_k,5]^T + b with pos as the GEMM's second
    operand stream (td_linear_ex: ``memory + pos`` is never formed), V_all = mem [W_v,0 | ...]^T + b, [rows, layers * E].
    The parameters stay with their layers (each layer's node returns the gradient of its whole in_proj); this node only owns
    d(mem): two chained GEMMs over the column-concatenated gradients (K = layers * E), the second adding onto the first in
    its epilogue - instead of 2 x layers GEMMs whose results the autograd engine would sum with elementwise launches."""

    @staticmethod
    def forward(ctx, mem, pos, shared, *w_and_b):
        E, dt = shared.E, mem.dtype
        Ws, bs = w_and_b[0::2], w_and_b[1::2]
        kf, vf, kd, vd = [], [], [], []
        for W in Ws:
            f, d_, _, _ = prepared(W[E : 2 * E], dt)
            kf.append(f); kd.append(d_)
            f, d_, _, _ = prepared(W[2 * E :], dt)
            vf.append(f); vd.append(d_)
        bk = torch.cat([b.detach()[E : 2 * E] for b in bs])
        bv = torch.cat([b.detach()[2 * E :] for b in bs])
        wk = torch.cat(kf, dim=0)
        K_all = ops.linear_ex(mem, wk, bk, a2=pos, w_shared=True) if pos is not None else ops.linear_fwd(mem, wk, bk)
        V_all = ops.linear_fwd(mem, torch.cat(vf, dim=0), bv)
        shared.wk_d, shared.wv_d = torch.cat(kd, dim=1), torch.cat(vd, dim=1)
        ctx.shared = shared
        ctx.set_materialize_grads(False)
        return K_all, V_all

    @staticmethod
    def backward(ctx, gK, gV):
        sh = ctx.shared
        assert gK is None and gV is None, "the consumers of CrossKVFn hand their gradients over through the shared buffers"
        if sh.dK is None:
            return (None,) * (3 + 2 * sh.n_layers)
        for l in range(sh.n_layers):  # a layer outside the loss (never the case in training) contributes nothing
            if l not in sh.written:
                sh.dK[:, l * sh.E : (l + 1) * sh.E].zero_()
                sh.dV[:, l * sh.E : (l + 1) * sh.E].zero_()
        d_mem = d_pos = None
        need_pos = sh.pos is not None and ctx.needs_input_grad[1]  # learned position embeddings (the sine ones carry no gradient)
        if ctx.needs_input_grad[0] or need_pos:
            d_k = ops.linear_fwd(sh.dK, sh.wk_d)  # gradient of (mem + pos) through the key projections
            if need_pos:
                d_pos = d_k
            if ctx.needs_input_grad[0]:
                # += the value path, in the GEMM's epilogue (in place unless d_k is also the positional gradient)
                d_mem = ops.linear_fwd(sh.dV, sh.wv_d, residual=d_k, out=None if need_pos else d_k)
        return (d_mem, d_pos, None) + (None,) * (2 * sh.n_layers)


class MHAPreKVFn(Function):
    """MHAFn for a layer whose key / value projections were hoisted into CrossKVFn: q projection, attention core on column
    block ``layer`` of K_all / V_all (read where they lie: row stride layers * E), out projection.  Backward: the attention
    backward writes dK / dV straight into the shared buffers; the gradient of the layer's whole packed in_proj (q, k and v
    rows) is produced here, so the parameter has a single owner and its weight gradients stay deferrable."""

    @staticmethod
    def forward(ctx, q_in, q_pos, K_all, V_all, W_in, b_in, W_out, b_out, key_pad, shared, layer, B, Lq, Lk, H, need_w, p_attn, seed_attn, p_out, seed_out):
        _note_use(ctx.needs_input_grad[4], W_in, b_in, W_out, b_out)
        E = q_in.shape[1]
        dt = q_in.dtype
        nl = shared.n_layers
        wq_f, wq_d, _, _ = prepared(W_in[:E], dt)
        if q_pos is not None:  # tgt + query_pos as two operand streams of the projection (transformer.py:735)
            q = ops.linear_ex(q_in, wq_f, b_in.detach()[:E], a2=q_pos, w_shared=True).view(B, Lq, E)
        else:
            q = ops.linear_fwd(q_in, wq_f, b_in.detach()[:E]).view(B, Lq, E)
        k = K_all.view(B, Lk, nl * E)[..., layer * E : (layer + 1) * E]
        v = V_all.view(B, Lk, nl * E)[..., layer * E : (layer + 1) * E]
        scale = 1.0 / math.sqrt(E // H)
        ctxv, probs, wavg = ops.mha_fwd(q, k, v, key_pad, H, scale, need_wavg=need_w, dropout_p=p_attn, seed=seed_attn)
        wo_f, wo_d, _, _ = prepared(W_out, dt)
        out = ops.linear_fwd(ctxv.view(B * Lq, E), wo_f, b_out.detach(), dropout_p=p_out, seed=seed_out)
        ctx.save_for_backward(q_in, q_pos, q, K_all, V_all, probs, ctxv, wq_d, wo_d)
        ctx.cfg = (B, Lq, Lk, H, E, scale, p_attn, seed_attn, p_out, seed_out, layer)
        ctx.params = (W_in, b_in, W_out, b_out)
        ctx.shared = shared
        return out, (wavg if need_w else None)

    @staticmethod
    def backward(ctx, dout, dwavg):
        _arm()
        q_in, q_pos, q, K_all, V_all, probs, ctxv, wq_d, wo_d = ctx.saved_tensors
        B, Lq, Lk, H, E, scale, p_attn, seed_attn, p_out, seed_out, layer = ctx.cfg
        sh = ctx.shared
        nl = sh.n_layers
        dt, dev = q_in.dtype, q_in.device
        defer = _can_defer(*ctx.params)
        g = ops.dropout(dout.contiguous(), p_out, seed_out) if p_out > 0 else dout.contiguous()
        if defer:
            dW_in = torch.empty((3 * E, E), dtype=torch.float32, device=dev)
            db_in = torch.empty(3 * E, dtype=torch.float32, device=dev)
        else:
            dW_in = ops.zeros_f32((3 * E, E), dev)
            db_in = ops.zeros_f32(3 * E, dev)
        dW_out, db_out = _wgrad(g, ctxv.view(B * Lq, E), defer, want_bias=True)
        dctx = ops.linear_fwd(g, wo_d).view(B, Lq, E)
        dq = torch.empty((B, Lq, E), dtype=dt, device=dev)
        dK_all, dV_all = sh.grads(K_all)
        cols = slice(layer * E, (layer + 1) * E)
        k = K_all.view(B, Lk, nl * E)[..., cols]
        v = V_all.view(B, Lk, nl * E)[..., cols]
        dk = dK_all.view(B, Lk, nl * E)[..., cols]
        dv = dV_all.view(B, Lk, nl * E)[..., cols]
        dwa = dwavg.contiguous().float() if dwavg is not None else None
        ops.mha_bwd(q, k, v, dctx, probs, dwa, H, scale, dq, dk, dv, dropout_p=p_attn, seed=seed_attn)
        sh.written.add(layer)
        dq2 = dq.view(B * Lq, E)
        _wgrad(dq2, q_in, defer, want_bias=True, out=dW_in[:E], dbias=db_in[:E], x2=q_pos)
        _wgrad(dK_all[:, cols], sh.mem, defer, want_bias=True, out=dW_in[E : 2 * E], dbias=db_in[E : 2 * E], x2=sh.pos)
        _wgrad(dV_all[:, cols], sh.mem, defer, want_bias=True, out=dW_in[2 * E :], dbias=db_in[2 * E :])
        need_pos = q_pos is not None and ctx.needs_input_grad[1]
        d_q_in = ops.linear_fwd(dq2, wq_d) if (ctx.needs_input_grad[0] or need_pos) else None
        return (d_q_in if ctx.needs_input_grad[0] else None, d_q_in if need_pos else None, None, None, dW_in, db_in, dW_out, db_out) + (None,) * 12


class _CrossQ1Shared:
    """What the six time-aligned cross-attention nodes share: the memory rows they all read and the ONE fp32 buffer their
    backward kernels accumulate d(memory) in (first layer to run writes, the others add)."""

    def __init__(self, mem, pos, want_dmem=True):
        self.mem, self.pos = mem, pos
        self.dmem = None
        self.want_dmem = want_dmem  # False: the memory needs no gradient (nothing will collect the buffer)
        # bf16 mode (TD_CROSS_DMEM_DEFER=0: off): no fp32 [F*S, E] buffer and no read-modify-write per layer - every layer's backward
        # leaves sixteen coefficients per memory row in `coef`, CrossMemFn.backward forms the gradient of all layers in one pass
        self.n_layers = 0   # layer nodes that took this memory in the forward pass (their index = their coefficient block)
        self.coef = None    # [F*S, pad32(16 * n_layers)] bf16, zero-filled
        self.deferred = {}  # layer index -> (u, d_zext) of a layer whose backward has run


class CrossMemFn(Function):
    """Graph anchor of the decoder's shared memory for CrossQ1Fn: forward hands out an empty token the six layer nodes take as an
    input (so this node runs after all of them), backward returns the d(memory) they accumulated in the shared buffer."""

    @staticmethod
    def forward(ctx, mem, shared):
        ctx.shared = shared
        ctx.set_materialize_grads(False)
        return mem.new_empty(0)

    @staticmethod
    def backward(ctx, g):
        sh = ctx.shared
        assert g is None, "the consumers of CrossMemFn hand their gradient over through the shared buffer"
        if sh.coef is not None:
            assert sh.dmem is None
            E = sh.mem.shape[1]
            F_, H = next(iter(sh.deferred.values()))[0].shape[0], next(iter(sh.deferred.values()))[0].shape[1] // E
            for l in range(sh.n_layers):
                if l not in sh.deferred:  # (a layer outside the differentiated graph: its coefficient columns were never written)
                    sh.coef[:, 16 * l : 16 * l + 16].zero_()
            d = ops.cross_q1_dmem(sh.coef, [sh.deferred.get(l) for l in range(sh.n_layers)], F_, sh.mem.shape[0] // F_, H, E)
            sh.coef, sh.deferred = None, {}
            return d, None
        if sh.dmem is None:
            return None, None
        d = sh.dmem if sh.mem.dtype == torch.float32 else ops.cast(sh.dmem, sh.mem.dtype)
        sh.dmem = None
        return d, None


class CrossQ1Fn(Function):
    """nn.MultiheadAttention of the decoder's time-aligned cross-attention (transformer.py:725-745: one query per frame, keys =
    memory + pos, values = memory) with the key and value projections moved to the query side (csrc/cross_attn.hip):

        q      = (tgt + query_pos) W_q^T + b_q                       [F, E]         (two operand streams, as in MHAFn)
        u      = q Wk_blk^T,  u[f, h] = scale * W_k,h^T q[f, head h]   [F, H*E]       (block-structured W_k, td_head_blocks_expand)
        probs, weights, zext = frame core(u, memory, pos, key padding)               (td_cross_q1_fwd; zext [F, H*E + H])
        ctx    = zext Wv_blk^T = W_v,h z[f, h] + b_v,h sum_s pd                  [F, E]
        out    = dropout(ctx W_o^T + b_o)

    No key / value projection of the F*S memory rows exists in either direction.  The layer keeps ownership of its whole packed
    in_proj gradient: rows [0, E) from the deferred batch, rows [E, 3E) from the diagonal blocks of two dense [E, H*E (+H)]
    weight gradients (launched here: the extraction has to follow them); the key bias gets zeros (q . b_k shifts every score of
    a row by the same amount)."""

    @staticmethod
    def forward(ctx, q_in, q_pos, token, W_in, b_in, W_out, b_out, key_pad, shared, F, S, H, need_w, p_attn, seed_attn, p_out, seed_out):
        _note_use(ctx.needs_input_grad[3], W_in, b_in, W_out, b_out)
        E = q_in.shape[1]
        dt = q_in.dtype
        scale = 1.0 / math.sqrt(E // H)
        wq_f, wq_d, _, _ = prepared(W_in[:E], dt)
        bq = b_in.detach()[:E]
        q = ops.linear_ex(q_in, wq_f, bq, a2=q_pos, w_shared=True) if q_pos is not None else ops.linear_fwd(q_in, wq_f, bq)
        Wd, bd = W_in.detach(), b_in.detach()
        wk_n, wk_t = ops.head_blocks_expand(Wd[E : 2 * E], None, scale, H, dt)
        wv_n, wv_t = ops.head_blocks_expand(Wd[2 * E :], bd[2 * E :], 1.0, H, dt)
        u = ops.linear_fwd(q, wk_t)
        probs, wavg, zext = ops.cross_q1_fwd(u, shared.mem, shared.pos, key_pad, F, S, H, need_wavg=need_w, dropout_p=p_attn, seed=seed_attn)
        ctxv = ops.linear_fwd(zext, wv_n)
        wo_f, wo_d, _, _ = prepared(W_out, dt)
        out = ops.linear_fwd(ctxv, wo_f, b_out.detach(), dropout_p=p_out, seed=seed_out)
        ctx.save_for_backward(q_in, q_pos, q, u, probs, zext, ctxv, wq_d, wo_d, wk_n, wv_t)
        ctx.cfg = (F, S, H, E, scale, p_attn, seed_attn, p_out, seed_out)
        ctx.params = (W_in, b_in, W_out, b_out)
        ctx.shared = shared
        ctx.layer = shared.n_layers
        shared.n_layers += 1
        return out, (wavg if need_w else None)

    @staticmethod
    def backward(ctx, dout, dwavg):
        _arm()
        q_in, q_pos, q, u, probs, zext, ctxv, wq_d, wo_d, wk_n, wv_t = ctx.saved_tensors
        F, S, H, E, scale, p_attn, seed_attn, p_out, seed_out = ctx.cfg
        sh = ctx.shared
        dt, dev = q_in.dtype, q_in.device
        defer = _can_defer(*ctx.params)
        g = ops.dropout(dout.contiguous(), p_out, seed_out) if p_out > 0 else dout.contiguous()
        dW_in = torch.empty((3 * E, E), dtype=torch.float32, device=dev)
        db_in = torch.empty(3 * E, dtype=torch.float32, device=dev)
        dW_out, db_out = _wgrad(g, ctxv, defer, want_bias=True)
        dctx = ops.linear_fwd(g, wo_d)
        d_zext = ops.linear_fwd(dctx, wv_t)
        Gv = ops.linear_wgrad(dctx, zext)  # dense [E, H*E + H]; (not through the batched launch: its job-table ring is sized for a few calls per step)
        ops.head_blocks_extract(Gv, 1.0, dW_in[2 * E :], db_in[2 * E :], H)
        dwa = dwavg.contiguous().float() if dwavg is not None else None
        if sh.want_dmem and dt == torch.bfloat16 and _CROSS_DMEM_DEFER and sh.n_layers <= 8 and sh.dmem is None:
            if sh.coef is None:
                # every layer that runs overwrites all 16 of its columns for every row: only the padding columns need defined values
                # (the layers' product is an MFMA over the whole padded width: 0 * NaN from uninitialised memory would poison it);
                # columns of a layer whose backward never ran are zeroed in CrossMemFn.backward
                width = (16 * sh.n_layers + 31) // 32 * 32
                sh.coef = torch.empty((F * S, width), dtype=dt, device=dev)
                if width > 16 * sh.n_layers:
                    sh.coef[:, 16 * sh.n_layers :].zero_()
            d_u = ops.cross_q1_bwd_coef(u, sh.mem, sh.pos, probs, d_zext, dwa, sh.coef, 16 * ctx.layer, F, S, H, dropout_p=p_attn, seed=seed_attn)
            sh.deferred[ctx.layer] = (u, d_zext)
        else:
            first = sh.dmem is None
            if not sh.want_dmem:
                dmem = None  # the memory needs no gradient: the kernel neither forms nor stores it
            elif first:
                dmem = sh.dmem = torch.empty((F * S, E), dtype=torch.float32, device=dev)
            else:
                dmem = sh.dmem
            d_u = ops.cross_q1_bwd(u, sh.mem, sh.pos, probs, d_zext, dwa, dmem, sh.want_dmem and not first, F, S, H, dropout_p=p_attn, seed=seed_attn)
        dq = ops.linear_fwd(d_u, wk_n)
        Gk = ops.linear_wgrad(q, d_u)
        ops.head_blocks_extract(Gk, scale, dW_in[E : 2 * E], None, H)
        db_in[E : 2 * E].zero_()
        if defer:
            _wgrad(dq, q_in, True, want_bias=True, out=dW_in[:E], dbias=db_in[:E], x2=q_pos)
        else:  # the immediate kernels add into their output
            dW_in[:E].zero_()
            db_in[:E].zero_()
            _wgrad(dq, q_in, False, want_bias=True, out=dW_in[:E], dbias=db_in[:E], x2=q_pos)
        need_pos = q_pos is not None and ctx.needs_input_grad[1]
        d_q_in = ops.linear_fwd(dq, wq_d) if (ctx.needs_input_grad[0] or need_pos) else None
        return (d_q_in if ctx.needs_input_grad[0] else None, d_q_in if need_pos else None, None, dW_in, db_in, dW_out, db_out) + (None,) * 10


_CROSS_DMEM_DEFER = _os.environ.get("TD_CROSS_DMEM_DEFER", "1") != "0"


def cross_q1_memory(mem, pos):
    """-> (token, shared) for ``multihead_attention_q1``: the decoder memory rows [F*S, E] (and their positional rows or None) shared
    by the layers' time-aligned cross-attention."""
    shared = _CrossQ1Shared(mem.detach(), None if pos is None else pos.detach(), want_dmem=mem.requires_grad and torch.is_grad_enabled())
    return CrossMemFn.apply(mem, shared), shared


def multihead_attention_q1(q_in, anchor, W_in, b_in, W_out, b_out, key_pad, F, S, H, need_weights=True, attn_dropout=0.0, out_dropout=0.0,
                           training=False, q_pos=None):
    pa = attn_dropout if training else 0.0
    po = out_dropout if training else 0.0
    token, shared = anchor
    return CrossQ1Fn.apply(q_in, q_pos, token, W_in, b_in, W_out, b_out, key_pad, shared, F, S, H, need_weights,
                           pa, _seed() if pa > 0 else 0, po, _seed() if po > 0 else 0)


def cross_kv(mem, pos, attn_modules):
    """-> (K_all, V_all, shared) for ``multihead_attention_prekv``: keys from mem + pos (pos: rows like mem, or None), values from
    mem; attn_modules: the layers' cross-attention parameter holders."""
    E = attn_modules[0].embed_dim
    shared = _CrossKVShared(mem, pos, len(attn_modules), E)
    flat = []
    for m in attn_modules:
        flat += [m.in_proj_weight, m.in_proj_bias]
    K_all, V_all = CrossKVFn.apply(mem, pos, shared, *flat)
    return K_all, V_all, shared


def multihead_attention_prekv(q_in, kv, layer, W_in, b_in, W_out, b_out, key_pad, B, Lq, Lk, H, need_weights=False, attn_dropout=0.0,
                              out_dropout=0.0, training=False, q_pos=None):
    pa = attn_dropout if training else 0.0
    po = out_dropout if training else 0.0
    K_all, V_all, shared = kv
    return MHAPreKVFn.apply(q_in, q_pos, K_all, V_all, W_in, b_in, W_out, b_out, key_pad, shared, layer, B, Lq, Lk, H, need_weights,
                            pa, _seed() if pa > 0 else 0, po, _seed() if po > 0 else 0)


class ReplicaMaps:
    """Index vectors of the temporal replication (transformer.py:393-427: frame (i, j) of video i takes the memory of clip
    i * n_clips + j // k), built once per (durations, stride, hw, L) on the host and kept on the device.  Rows of the clip
    memory are c * S + s, rows of the frame memory f * S + s (S = hw visual + L text tokens).

      vis_src / vis_dst   [F * hw]   clip row / frame row of every visual token of every frame
      txt_src / txt_dst   [F * L]    the same for the text tokens
      all_src             [F * S]    clip row of every frame row (--no_fast: the whole memory is replicated)
      seg_*               CSR lists "frame rows summed into this clip row" for the backward: visual rows (compact output
                          [n * hw]), text rows (output rows clip_txt), all rows
      clip_vis / clip_txt [n * hw] / [n * L]  clip rows of the visual / text tokens"""

    def __init__(self, owner, n: int, hw: int, L: int, device):
        owner = owner.cpu().long()
        F, S = owner.numel(), hw + L
        i32 = lambda t_: t_.to(torch.int32).contiguous().to(device)
        p, l_ = torch.arange(hw), torch.arange(L)
        f = torch.arange(F)
        self.F, self.n, self.hw, self.L, self.S = F, n, hw, L, S
        self.vis_src = i32((owner[:, None] * S + p[None, :]).reshape(-1))
        self.vis_dst = i32((f[:, None] * S + p[None, :]).reshape(-1))
        self.txt_src = i32((owner[:, None] * S + hw + l_[None, :]).reshape(-1))
        self.txt_dst = i32((f[:, None] * S + hw + l_[None, :]).reshape(-1))
        self.all_src = i32((owner[:, None] * S + torch.arange(S)[None, :]).reshape(-1))
        c = torch.arange(n)
        self.iota_vis = i32(torch.arange(n * hw))  # identity (a residual read by compact row while the output row is mapped)
        self.clip_vis = i32((c[:, None] * S + p[None, :]).reshape(-1))
        self.clip_txt = i32((c[:, None] * S + hw + l_[None, :]).reshape(-1))
        # frames of every clip, in frame order (owner is non-decreasing by construction, but nothing here relies on it)
        order = torch.argsort(owner, stable=True)
        counts = torch.bincount(owner, minlength=n)
        starts = torch.cumsum(counts, 0) - counts

        def csr(cols, row_off):
            """output row (c, j) sums the input rows frame * S + row_off + j over the frames of clip c (input = frame-memory rows)"""
            m = cols.numel()
            per = counts[:, None].expand(n, m).reshape(-1)           # segment length of output row (c, j)
            ptr = torch.zeros(n * m + 1, dtype=torch.long)
            ptr[1:] = torch.cumsum(per, 0)
            idx = torch.empty(int(ptr[-1]), dtype=torch.long)
            for ci in range(n):  # n is the number of clips (tens to hundreds): host work done once per batch pattern
                fr = order[starts[ci] : starts[ci] + counts[ci]]
                blk = (fr[None, :] * S + row_off + cols[:, None]).reshape(-1)   # [m, count] -> row-major: segment of (c, j) contiguous
                idx[ptr[ci * m] : ptr[ci * m] + blk.numel()] = blk
            return i32(idx), i32(ptr)

        self.seg_vis = csr(p, 0)
        self.seg_txt = csr(l_, hw)
        self.seg_all = csr(torch.arange(S), 0)


class ReplicateRowsFn(Function):
    """frames[f * S + s] = mem[owner[f] * S + s]: the temporal replication as ONE gather pass of 16-byte accesses (td_rows_copy)
    instead of a Python loop of slice assignments; backward = segment sums over the frames of each clip (td_rows_segment_sum).
    The default (slow-fast) model never runs this for the visual tokens - SlowFastAggregateFn reads the clip rows through the
    index inside its GEMM; it serves --no_fast, where the replicated memory itself is what the decoder attends to."""

    @staticmethod
    def forward(ctx, mem, maps):
        out = torch.empty((maps.F * maps.S, mem.shape[1]), dtype=mem.dtype, device=mem.device)
        ops.rows_copy(mem, maps.all_src, out, None, maps.F * maps.S)
        ctx.maps, ctx.rows = maps, mem.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        maps = ctx.maps
        g = g.contiguous()
        d_mem = torch.empty((ctx.rows, g.shape[1]), dtype=g.dtype, device=g.device)
        ops.rows_segment_sum(g, maps.seg_all[0], maps.seg_all[1], d_mem)
        return d_mem, None


class SlowFastAggregateFn(Function):
    """Temporal replication + slow-fast aggregation (transformer.py:393-445) producing the frame memory [F * S, d] directly:

        frames[f, p]      = vis + fast_residual(vis + fast_mem[f, p]),   vis = mem[owner[f], p]      (visual tokens)
        frames[f, hw + l] = mem[owner[f], hw + l]                                                    (text tokens)

    ONE GEMM (td_linear_ex) reads the clip rows through the replication index as its first operand stream, the fast
    features as its second (same weight: (vis + fast) W^T), takes the residual through the same index and writes straight
    into the frame rows; one gather pass copies the text rows.  Neither mem[owner], nor vis + fast_mem, nor a concatenation
    exists in memory.  Backward: by linearity the replication's segment sum commutes with the aggregation -
    d_mem[vis] = G + G Wr with G = segment sum of the visual-row gradients (a GEMM over n * hw clip rows instead of F * hw frame
    rows); the mix operand of the weight gradient is re-formed (gather + add in one pass) instead of having been stored."""

    @staticmethod
    def forward(ctx, mem, fast_mem, W, b, maps):
        _note_use(ctx.needs_input_grad[2], W, b)
        dt = mem.dtype
        wf, wd, _, _ = prepared(W, dt)
        frames = torch.empty((maps.F * maps.S, mem.shape[1]), dtype=dt, device=mem.device)
        ops.linear_ex(mem, wf, b.detach(), a1_map=maps.vis_src, a2=fast_mem, w_shared=True, residual=mem, res_map=maps.vis_src, out=frames, out_map=maps.vis_dst)
        ops.rows_copy(mem, maps.txt_src, frames, maps.txt_dst, maps.F * maps.L)
        ctx.save_for_backward(mem, fast_mem, wd)
        ctx.maps, ctx.params = maps, (W, b)
        return frames

    @staticmethod
    def backward(ctx, g):
        _arm()
        mem, fast_mem, wd = ctx.saved_tensors
        maps = ctx.maps
        g = g.contiguous()
        d, dt, dev = g.shape[1], g.dtype, g.device
        n_vis = maps.F * maps.hw
        g_vis = torch.empty((n_vis, d), dtype=dt, device=dev)
        ops.rows_copy(g, maps.vis_dst, g_vis, None, n_vis)                       # visual-row gradients, compact
        d_fast = ops.linear_fwd(g_vis, wd) if ctx.needs_input_grad[1] else None   # d(vis + fast) = g Wr
        mix = torch.empty((n_vis, d), dtype=dt, device=dev)
        ops.rows_copy(mem, maps.vis_src, mix, None, n_vis, add=fast_mem)          # vis + fast_mem, re-formed for the weight gradient
        dW, db = _wgrad(g_vis, mix, _can_defer(*ctx.params), want_bias=True)
        d_mem = None
        if ctx.needs_input_grad[0]:
            d_mem = torch.empty_like(mem)
            ops.rows_segment_sum(g, maps.seg_txt[0], maps.seg_txt[1], d_mem, maps.clip_txt)
            G = torch.empty((maps.n * maps.hw, d), dtype=dt, device=dev)
            ops.rows_segment_sum(g, maps.seg_vis[0], maps.seg_vis[1], G)
            ops.linear_ex(G, wd, None, residual=G, res_map=maps.iota_vis, out=d_mem, out_map=maps.clip_vis)  # d_mem[vis] = G + G Wr
        return d_mem, d_fast, dW, db, None


class AttnCoreFn(Function):
    """softmax(scale q k^T + key padding) v on already projected q, k, v rows ([B*L, E] each, heads in column blocks of
    E // H = 32 or 64) - the attention core alone, for encoders that own their projections (RoBERTa: separate query / key /
    value Linear modules).  Probability dropout inside the kernel; no layout change, no copies."""

    @staticmethod
    def forward(ctx, q, k, v, key_pad, B, Lq, Lk, H, p_drop, seed):
        E = q.shape[1]
        scale = 1.0 / math.sqrt(E // H)
        q3, k3, v3 = q.view(B, Lq, E), k.view(B, Lk, E), v.view(B, Lk, E)
        out, probs, _ = ops.mha_fwd(q3, k3, v3, key_pad, H, scale, need_wavg=False, dropout_p=p_drop, seed=seed)
        ctx.save_for_backward(q3, k3, v3, probs)
        ctx.cfg = (B, Lq, Lk, H, E, scale, p_drop, seed)
        return out.view(B * Lq, E)

    @staticmethod
    def backward(ctx, dout):
        q3, k3, v3, probs = ctx.saved_tensors
        B, Lq, Lk, H, E, scale, p_drop, seed = ctx.cfg
        dq, dk, dv = torch.empty_like(q3), torch.empty_like(k3), torch.empty_like(v3)
        ops.mha_bwd(q3, k3, v3, dout.contiguous().view(B, Lq, E), probs, None, H, scale, dq, dk, dv, dropout_p=p_drop, seed=seed)
        return dq.view(B * Lq, E), dk.view(B * Lk, E), dv.view(B * Lk, E), None, None, None, None, None, None, None


def attention_core(q, k, v, key_pad, B, Lq, Lk, H, dropout_p=0.0, training=False):
    p = dropout_p if training else 0.0
    return AttnCoreFn.apply(q, k, v, key_pad, B, Lq, Lk, H, p, _seed() if p > 0 else 0)


def multihead_attention(q_in, k_in, v_in, W_in, b_in, W_out, b_out, key_pad, B, Lq, Lk, H, need_weights=False,
                        attn_dropout=0.0, out_dropout=0.0, training=False, q_pos=None):
    pa = attn_dropout if training else 0.0
    po = out_dropout if training else 0.0
    return MHAFn.apply(q_in, q_pos, k_in, v_in, W_in, b_in, W_out, b_out, key_pad, B, Lq, Lk, H, need_weights,
                       pa, _seed() if pa > 0 else 0, po, _seed() if po > 0 else 0)


class GeluFn(Function):
    """Exact (erf) GELU on the elementwise kernel; backward from the saved pre-activation."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return ops.gelu_fwd(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return ops.gelu_bwd(g.contiguous(), x)


def gelu(x):
    return GeluFn.apply(x)


class DropoutFn(Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        ctx.cfg = (p, seed)
        return ops.dropout(x.contiguous(), p, seed)

    @staticmethod
    def backward(ctx, g):
        p, seed = ctx.cfg
        return ops.dropout(g.contiguous(), p, seed), None, None


def dropout(x, p, training):
    if not training or p <= 0:
        return x
    return DropoutFn.apply(x, p, _seed())


class CastFn(Function):
    """dtype conversion on the cast kernel (fp32 <-> compute dtype at the module boundary)."""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        return ops.cast(x.contiguous(), dtype)

    @staticmethod
    def backward(ctx, g):
        return ops.cast(g.contiguous(), ctx.src), None


def cast(x, dtype):
    return x if x.dtype == dtype else CastFn.apply(x, dtype)
