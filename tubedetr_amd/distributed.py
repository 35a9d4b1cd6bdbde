"""Data-parallel gradient exchange for the hot path: one clip per GPU, one all-reduce of the gradients per step over
RCCL / xGMI (``torch.distributed`` backend "nccl" on ROCm; "gloo" in the CPU tests).

The reference wraps the model in ``DistributedDataParallel(find_unused_parameters=True)`` (main.py:372-376): per step
that walks the autograd graph twice (two forward calls), re-broadcasts 400+ frozen buffers and copies every gradient
into 25 MB buckets.  The exchange itself is only "average 185 M fp32 gradients", so here it is done directly:

  * gradients are gathered into ONE persistent flat buffer with a fused multi-tensor copy, optionally narrowed to bf16
    for the wire (halves the bytes each xGMI link carries; RCCL ring all-reduce is per-link bound on this fabric),
  * a single ``all_reduce`` (no bucketing: 288 GB of HBM makes one 0.74 GB collective cheaper than 30 small ones),
  * the averaged values are scattered back into the ``.grad`` tensors with one more fused copy - or, zero-copy, ``.grad``
    is re-pointed at the flat buffer (``attach``); the gather has no collective and can live inside the step's HIP graph.

Parameters that received no gradient (RoBERTa's pooler - the reason the reference needs find_unused_parameters) are
simply exchanged as zeros.  The forward / backward of the step itself contains no collective, so it can be replayed
from a HIP graph on every rank; ``SetCriterion``'s num_boxes all-reduce is hoisted out through ``sync_num_boxes``.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def _all_reduce(t: torch.Tensor, op, group=None) -> None:
    """dist.all_reduce; with the `gloo` backend (CPU tests, single-GPU debugging with several ranks on one device) a
    device buffer is moved through the host, since gloo builds without device support reject CUDA tensors."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        host = t.detach().contiguous().cpu()  # (gloo rejects non-contiguous tensors; .cpu() alone keeps the strides of a view)
        dist.all_reduce(host, op=op, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=op, group=group)


def broadcast_(t: torch.Tensor, src: int = 0, group=None) -> None:
    """dist.broadcast in place (what DistributedDataParallel's constructor does with rank 0's parameters and buffers, main.py:372-376);
    device tensors go through the host under `gloo`, like ``_all_reduce``."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        host = t.detach().contiguous().cpu()
        dist.broadcast(host, src, group=group)
        if dist.get_rank(group) != src:
            t.copy_(host)
    else:
        dist.broadcast(t, src, group=group)


def _reduce_scatter_all_gather(buf: torch.Tensor, shard: torch.Tensor, op, group=None, async_op: bool = False):
    """Average / sum the 1-D ``buf`` over the ranks as an explicit reduce-scatter + all-gather pair on the flat buffer (what a ring
    all-reduce is made of, as two collectives the backend may schedule over all xGMI links independently; it is also where a
    sharded optimizer step would go: between the two, each rank holds the reduced values of ITS slice).  The part of ``buf`` that
    does not divide by the world size (< world elements) goes through a plain all-reduce.  ``shard``: scratch of at least
    ceil(len / world) elements of buf's dtype.  Returns the work handles (async) or None."""
    world = dist.get_world_size(group)
    n = buf.numel()
    m = n - n % world
    works = []
    host = buf.is_cuda and dist.get_backend(group) == "gloo"  # (CPU tests with a device buffer: staged through the host, synchronous)
    if host:
        hb = buf.detach().contiguous().cpu()
        _reduce_scatter_all_gather(hb, torch.empty(max(1, m // world), dtype=hb.dtype), op, group)
        buf.copy_(hb)
        return None
    assert has_rs_ag(), "reduce_scatter_tensor / all_gather_into_tensor missing (FlatGradAllReducer falls back to all_reduce at construction)"
    if m:
        sh = shard[: m // world]
        w1 = dist.reduce_scatter_tensor(sh, buf[:m], op=op, group=group, async_op=async_op)
        # (collectives of one group run in issue order on the backend's stream: the gather reads the scattered shard)
        w2 = dist.all_gather_into_tensor(buf[:m], sh, group=group, async_op=async_op)
        works += [w1, w2]
    if n - m:
        works.append(dist.all_reduce(buf[m:], op=op, group=group, async_op=async_op))
    return works if async_op else None


def has_rs_ag() -> bool:
    return hasattr(dist, "reduce_scatter_tensor") and hasattr(dist, "all_gather_into_tensor")


class FlatGradAllReducer:
    def __init__(self, params: Iterable[torch.nn.Parameter], wire_dtype: torch.dtype = torch.float32, group=None, late=None,
                 collective: str = "all_reduce", late_groups=None):
        """``late``: optional set of parameters (or ids) whose gradient only becomes final in the LAST backward stage (the
        ResNet trunk when the step is split at the trunk boundary, harness.backward_in_stages): everything else is
        exchanged by ``launch(early=True)`` while that stage still runs - the overlap DDP's bucketed reducer gives the
        reference (main.py:372-376), with two collectives instead of ~30 buckets.
        ``late_groups`` (instead of ``late``): SEVERAL late stages, in the order their gradients become final (the trunk issued stage by
        stage, ResNetBody.backward_trunk(after_stage=...): layer4's parameters, layer3's, layer2's) - stage k of them is exchanged by
        ``launch(stage=k)``, 1-based, while the later ones are still computed; ``launch(early=False)`` still means "everything late"."""
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        assert late is None or late_groups is None, "late or late_groups, not both"
        groups = [list(late)] if late is not None else [list(g_) for g_ in (late_groups or [])]
        stage_by_id = {}
        for k_, g_ in enumerate(groups):
            for p in g_:
                stage_by_id[id(p) if torch.is_tensor(p) else p] = k_ + 1
        self.stage_of = [stage_by_id.get(id(p), 0) for p in self.params]  # 0 = final when the first backward stage ends
        self.n_late_stages = len(groups)
        self.is_late = [s_ > 0 for s_ in self.stage_of]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.wire_dtype = wire_dtype
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.wire = self.flat if wire_dtype == torch.float32 else torch.zeros(n, dtype=wire_dtype, device=dev)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off : off + p.numel()].view_as(p))
            off += p.numel()
        self.numel = n
        # maximal runs of consecutive parameters of the same stage: (first element, end element, stage)
        self.runs, off = [], 0
        for p, lt in zip(self.params, self.stage_of):
            if self.runs and self.runs[-1][2] == lt:
                self.runs[-1][1] = off + p.numel()
            else:
                self.runs.append([off, off + p.numel(), lt])
            off += p.numel()
        assert collective in ("all_reduce", "rs_ag")
        if collective == "rs_ag" and not has_rs_ag():
            # decided ONCE, here, so that whoever reports ``self.collective`` (bench.py's JSON line) reports what actually runs
            import warnings

            warnings.warn("torch.distributed has no reduce_scatter_tensor / all_gather_into_tensor: the gradient exchange uses all_reduce")
            collective = "all_reduce"
        self.collective = collective  # "rs_ag": reduce-scatter + all-gather on the flat buffer instead of one all-reduce (same result)
        self._shard = None  # rs_ag scratch: ceil(numel / world) elements, allocated at first use (never on the host-staged gloo path)
        self._pending: list = []
        self.always_communicate = False  # diagnostic: issue the collective even in a 1-rank group
        self._had_grad: Optional[List[bool]] = None     # which parameters had a local gradient at the last gather()
        self._global_used: Optional[List[bool]] = None  # ... on ANY rank (exchanged once, see _sync_usage)
        self._attached = False

    # ---- three phases; `reduce()` runs them back to back ----
    def gather(self, grads: Optional[List[Optional[torch.Tensor]]] = None) -> None:
        """Copy the gradients (default: the parameters' current ``.grad``) into the flat buffer with fused multi-tensor
        copies; missing ones become zeros.  Contains no collective: it can be captured in the step's HIP graph."""
        grads = [p.grad for p in self.params] if grads is None else grads
        have = [(v, g) for v, g in zip(self.views, grads) if g is not None and g.data_ptr() != v.data_ptr()]
        missing = [v for v, g in zip(self.views, grads) if g is None]
        if missing:
            torch._foreach_zero_(missing)
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        self._note_usage(range(len(self.params)), grads)

    def _note_usage(self, sel, grads) -> None:
        """Record which of the parameters ``sel`` had a local gradient.  A parameter no rank used when the usage map was
        exchanged and that now has a gradient here invalidates the map: ranks that still skip it would silently diverge,
        so it is re-exchanged (every rank must see the same change: data-dependent parameter usage is not supported by the
        reference's model either)."""
        if self._had_grad is None or len(self._had_grad) != len(self.params):
            self._had_grad = [True] * len(self.params)
        for i, g in zip(sel, grads):
            self._had_grad[i] = g is not None
            if g is not None and self._global_used is not None and not self._global_used[i]:
                self._global_used = None

    def _sync_usage(self) -> None:
        """DDP(find_unused_parameters=True) hands every rank the reduced gradient of a parameter ANY rank used and leaves
        globally unused ones (RoBERTa's pooler) at grad=None.  Same rule here: the usage bitmap is exchanged once (one
        small all-reduce + one host read, at the first step only) and kept; ``attach`` / ``scatter`` then serve every
        parameter that any rank used, so replicas cannot diverge when usage differs between ranks."""
        had = self._had_grad if self._had_grad is not None else [True] * len(self.params)
        if self.world > 1:
            flags = torch.tensor([1.0 if h else 0.0 for h in had], dtype=torch.float32, device=self.flat.device)
            _all_reduce(flags, dist.ReduceOp.SUM, self.group)
            self._global_used = [bool(v > 0) for v in flags.tolist()]
        else:
            self._global_used = list(had)
        if self._attached:
            self.attach()

    def all_reduce(self) -> None:
        """The one collective of a training step: average the flat buffer over the ranks (in place)."""
        if self._global_used is None:
            self._sync_usage()
        if not (self.world > 1 or self.always_communicate):
            return
        avg = dist.is_initialized() and dist.get_backend(self.group) == "nccl"  # RCCL averages in the reduction itself
        op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
        red = (lambda t_: _reduce_scatter_all_gather(t_, self._full_shard(), op, self.group)) if self.collective == "rs_ag" else \
            (lambda t_: _all_reduce(t_, op, self.group))
        if self.wire is not self.flat:
            self.wire.copy_(self.flat)
            red(self.wire)
            self.flat.copy_(self.wire)
        else:
            red(self.flat)
        if not avg:
            self.flat.div_(self.world)

    def _host_staged(self) -> bool:
        return self.flat.is_cuda and dist.is_initialized() and dist.get_backend(self.group) == "gloo"

    def _full_shard(self) -> Optional[torch.Tensor]:
        if self._host_staged():
            return None  # _reduce_scatter_all_gather stages the buffer through the host and brings its own scratch
        need = -(-self.numel // max(self.world, 1))
        if self._shard is None or self._shard.numel() < need:
            self._shard = torch.empty(need, dtype=self.wire.dtype, device=self.flat.device)
        return self._shard

    # ---- staged form: exchange what is final while the rest of backward still runs ----
    @staticmethod
    def _wanted(st: int, early, stage) -> bool:
        """Does a parameter / run of stage ``st`` belong to the selection?  stage = k: exactly late stage k; else early: stage 0 / every late stage."""
        if stage is not None:
            return st == stage
        return (st == 0) if early else (st > 0)

    def gather_stage(self, early: Optional[bool] = None, stage: Optional[int] = None) -> None:
        """Fused multi-tensor copy of one stage's gradients into the flat buffer (no collective: may be captured in the
        HIP graph of that stage)."""
        sel = [i for i, st in enumerate(self.stage_of) if self._wanted(st, early, stage)]
        grads = [self.params[i].grad for i in sel]
        have = [(self.views[i], g) for i, g in zip(sel, grads) if g is not None and g.data_ptr() != self.views[i].data_ptr()]
        missing = [self.views[i] for i, g in zip(sel, grads) if g is None]
        if missing:
            torch._foreach_zero_(missing)
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        self._note_usage(sel, grads)  # (same invalidation rule as gather(): the staged / graph path must not keep a stale usage map)

    def exchange_stage(self, early: Optional[bool] = None, stage: Optional[int] = None) -> None:
        """Start the all-reduce of one stage's runs WITHOUT waiting for it (async_op: the collective is ordered after the
        work enqueued on the current stream so far and runs on the backend's own stream); ``finish()`` joins."""
        if not (self.world > 1 or self.always_communicate):
            return
        avg = dist.get_backend(self.group) == "nccl"
        op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
        for a, b, st in self.runs:
            if not self._wanted(st, early, stage):
                continue
            buf = self.flat[a:b]
            if self.wire is not self.flat:
                self.wire[a:b].copy_(buf)
                buf = self.wire[a:b]
            if self.collective == "rs_ag":
                # (one scratch shard per run in flight: the runs of a stage are exchanged back to back on the backend's stream)
                shard = None if self._host_staged() else (torch.empty(max(1, -(-(b - a) // max(self.world, 1))), dtype=buf.dtype, device=buf.device) if self._pending else self._full_shard())
                # (async only where collectives of a group are stream-ordered - RCCL; gloo runs async operations on independent threads:
                #  the gather could read the shard before the scatter has written it)
                works = _reduce_scatter_all_gather(buf, shard, op, self.group, async_op=dist.get_backend(self.group) == "nccl")
                self._pending.append((works, a, b, avg))
                continue
            if buf.is_cuda and dist.get_backend(self.group) == "gloo":  # host-staged (tests): synchronous
                _all_reduce(buf, op, self.group)
                work = None
            else:
                work = dist.all_reduce(buf, op=op, group=self.group, async_op=True)
            self._pending.append((work, a, b, avg))

    def launch(self, early: Optional[bool] = None, stage: Optional[int] = None) -> None:
        """gather_stage + exchange_stage: exchange what is final (early: everything but the ``late`` parameters; stage = k: late
        stage k) while the rest of backward still runs."""
        self.gather_stage(early, stage)
        self.exchange_stage(early, stage)

    def finish(self, attach: Optional[bool] = True) -> None:
        """Wait (on the current stream) for the collectives started by ``launch`` and hand the averaged gradients over
        (attach=True: ``.grad`` becomes the view of the flat buffer; False: copied back into the existing ``.grad``; None:
        nothing - the views were attached earlier, e.g. when the step replays from HIP graphs)."""
        for work, a, b, avg in self._pending:
            for w_ in (work if isinstance(work, (list, tuple)) else [work]):
                if w_ is not None:
                    w_.wait()
            if self.wire is not self.flat:
                self.flat[a:b].copy_(self.wire[a:b])
            if not avg:
                self.flat[a:b].div_(self.world)
        self._pending = []
        if self._global_used is None:
            self._sync_usage()
        if attach is None:
            return
        if attach:
            self.attach()
        else:
            self.scatter()

    def attach(self) -> None:
        """Zero-copy hand-over: ``.grad`` of every parameter that had a gradient becomes its view of the flat buffer
        (what a fused optimizer wants anyway)."""
        used = self._global_used if self._global_used is not None else (self._had_grad or [True] * len(self.params))
        for p, v, u in zip(self.params, self.views, used):
            if u:
                p.grad = v
        self._attached = True

    def scatter(self) -> None:
        """Copy the averaged values back into the existing ``.grad`` tensors (keeps their storage)."""
        used = self._global_used if self._global_used is not None else [True] * len(self.params)
        have = []
        for v, p, u in zip(self.views, self.params, used):
            if p.grad is None and u:
                p.grad = v.clone()  # unused on this rank, used elsewhere: every replica applies the same update
            elif p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                have.append((p.grad, v))
        if have:
            torch._foreach_copy_([g for g, _ in have], [v for _, v in have])

    def reduce(self, attach: bool = False) -> None:
        """Average the current ``.grad`` of every parameter over the ranks, in place (``attach=True``: ``.grad`` is
        re-pointed at the flat buffer instead of being copied back)."""
        self.gather()
        self.all_reduce()
        if attach:
            self.attach()
        else:
            self.scatter()


def sync_num_boxes(n_local: int, out: torch.Tensor, group=None) -> torch.Tensor:
    """clamp(sum over ranks of the annotated-box count / world, min=1) into the device scalar ``out``
    (models/tubedetr.py:407-413), without a host synchronisation."""
    out.fill_(float(n_local))
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        _all_reduce(out, dist.ReduceOp.SUM, group)
        out.div_(dist.get_world_size(group))
    out.clamp_(min=1)
    return out
