"""Optimizer-side tail of a training step on flat buffers: gradient clipping, AdamW with the
reference's three name-based parameter groups and the EMA of the weights, as three HIP launches per step instead of
~3 elementwise torch launches for each of the 923 state-dict entries.

Replaces, with identical arithmetic (checked against torch in tests/test_optim_gpu.py):
    torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm); optimizer.step()     engine.py:147-151
    torch.optim.AdamW(param_dicts, lr, weight_decay) with the groups of                 main.py:381-413
    update_ema(model, model_ema, decay)                                                  util/optim.py:8-25
`adjust_learning_rate(optimizer, ...)` (util/optim.py:28-95) works unchanged on ``param_groups``.

Layout: every trainable parameter becomes a view of ONE fp32 buffer (``p.data`` is re-pointed, the autograd leaves and
the state_dict stay what they were); gradients are taken from the flat buffer of the data-parallel exchange
(``FlatGradAllReducer``, gathered there anyway) or gathered with one fused multi-tensor copy; Adam moments and the EMA
copy are flat buffers of the same shape.  An ``ema_model`` (the reference keeps ``deepcopy(model)``) is re-pointed at the
EMA buffer the same way, so evaluation / checkpointing code reading it needs no change.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch

from . import _hip
from .functional import invalidate_prepared


def reference_group(name: str) -> int:
    """main.py:381-405: 0 = everything else (args.lr), 1 = "backbone" in name (args.lr_backbone), 2 = "text_encoder" in
    name (args.text_encoder_lr).  A name holding both substrings lands in both reference groups; none does."""
    if "backbone" in name:
        return 1
    if "text_encoder" in name:
        return 2
    return 0


def adjust_learning_rate(optimizer, epoch: int, curr_step: int, num_training_steps: int, args) -> None:
    """The reference's schedule (util/optim.py:28-95) on any optimizer exposing the three ``param_groups`` (this module's
    FusedAdamWEMA or torch's): "step" (all rates / 10 after lr_drop epochs), "multistep" (halved at lr_drop, then every 50
    epochs), "linear_with_warmup" (text encoder: linear warm-up over fraction_warmup_steps, then linear decay to 0; the
    rest as "step"), "all_linear_with_warmup" (every rate follows the text encoder's)."""
    from bisect import bisect_right

    warm = round(args.fraction_warmup_steps * num_training_steps)

    def linear():
        if curr_step < warm:
            return float(curr_step) / float(max(1, warm))
        return max(0.0, float(num_training_steps - curr_step) / float(max(1, num_training_steps - warm)))

    if args.schedule == "step":
        gamma = 0.1 ** (epoch // args.lr_drop)
        text_gamma = gamma
    elif args.schedule == "multistep":
        gamma = 0.5 ** bisect_right(list(range(args.lr_drop, args.epochs, 50)), epoch)
        text_gamma = gamma
    elif args.schedule == "linear_with_warmup":
        gamma = 0.1 ** (epoch // args.lr_drop)
        text_gamma = linear()
    elif args.schedule == "all_linear_with_warmup":
        text_gamma = linear()
        gamma = text_gamma
    else:
        raise NotImplementedError(args.schedule)
    base = [args.lr, args.lr_backbone, args.text_encoder_lr]
    assert len(optimizer.param_groups) == len(base)
    for group, lr, gm in zip(optimizer.param_groups, base, [gamma, gamma, text_gamma]):
        group["lr"] = lr * gm


class FusedAdamWEMA:
    def __init__(self, model: torch.nn.Module, lr: float = 5e-5, lr_backbone: float = 1e-5, text_encoder_lr: float = 5e-5,
                 weight_decay: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, max_norm: float = 0.1,
                 ema_model: Optional[torch.nn.Module] = None, ema_decay: float = 0.9998, reducer=None):
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        self.names = [n for n, _ in named]
        self.params: List[torch.nn.Parameter] = [p for _, p in named]
        dev = self.params[0].device
        assert dev.type == "cuda", "the fused optimizer runs on the GPU only"
        n = sum(p.numel() for p in self.params)
        self.numel = n
        self.flat_p = torch.empty(n, dtype=torch.float32, device=dev)
        self.views, self.offsets, off = [], [], 0
        for p in self.params:
            v = self.flat_p[off : off + p.numel()].view_as(p)
            v.copy_(p.data)
            p.data = v  # the parameter now lives in the flat buffer (same leaf tensor, same state_dict entry)
            self.views.append(v)
            self.offsets.append(off)
            off += p.numel()
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.ema = None
        self.ema_decay = float(ema_decay)
        if ema_model is not None:
            # the EMA buffer starts from ema_model's OWN weights (a resumed run loads checkpoint["model_ema"] into it before the
            # optimizer is built, main.py:562-568), not from the model's
            self.ema = torch.empty_like(self.flat_p)
            ema_named = dict(ema_model.named_parameters())
            for nme, off_, p in zip(self.names, self.offsets, self.params):
                v = self.ema[off_ : off_ + p.numel()].view_as(p)
                v.copy_(ema_named[nme].data)
                ema_named[nme].data = v
        self.reducer = reducer
        if reducer is not None:
            assert [id(p) for p in reducer.params] == [id(p) for p in self.params], "reducer and optimizer must hold the same parameters in the same order"
            self.flat_g = reducer.flat
        else:
            self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad_views = [self.flat_g[o : o + p.numel()].view_as(p) for o, p in zip(self.offsets, self.params)]
        self.betas, self.eps, self.weight_decay, self.max_norm = betas, float(eps), float(weight_decay), float(max_norm)
        # the reference's three groups, in its order; adjust_learning_rate() writes param_groups[i]["lr"]
        self.group_of = [reference_group(nm) for nm in self.names]
        self.param_groups = [{"lr": lr, "params": [p for p, g in zip(self.params, self.group_of) if g == 0]},
                             {"lr": lr_backbone, "params": [p for p, g in zip(self.params, self.group_of) if g == 1]},
                             {"lr": text_encoder_lr, "params": [p for p, g in zip(self.params, self.group_of) if g == 2]}]
        # learning rates change every step under the warm-up schedules: the host runs ahead of the stream, so the pinned
        # staging buffer of upload i must not be rewritten before its async copy has been consumed - a small ring, each slot
        # guarded by the event recorded behind its copy
        self._lr_host = [torch.zeros(4, dtype=torch.float32).pin_memory() for _ in range(4)]
        self._lr_ev = [None] * 4
        self._lr_slot = 0
        self._lr_dev = torch.zeros(4, dtype=torch.float32, device=dev)
        self._lr_last = None
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.norm_clip = torch.zeros(2, dtype=torch.float32, device=dev)  # [total gradient norm, clip coefficient] of the last step
        self._ws = torch.empty(_hip.lib().td_grad_norm_ws_bytes(), dtype=torch.uint8, device=dev)
        self._active = None
        self._loaded_active = None  # load_state_dict: which parameters the checkpoint held state for (until the first step decides)
        self._segs = None

    # ---- segments: maximal runs of parameters with the same (group, has-gradient) ----
    def _build_segments(self, active: List[bool]):
        runs = []
        for off, p, g, a in zip(self.offsets, self.params, self.group_of, active):
            if runs and runs[-1][2] == g and runs[-1][3] == a:
                runs[-1][1] = off + p.numel()
            else:
                runs.append([off, off + p.numel(), g, a])
        if len(runs) > 32:
            raise RuntimeError(f"{len(runs)} optimizer segments: raise TD_OPTIM_MAX_SEGMENTS")
        arr = (_hip.OptimSegment * len(runs))()
        for s, (b, e, g, a) in zip(arr, runs):
            s.begin, s.end, s.group, s.active = b, e, g, int(a)
        self._segs, self._active = arr, list(active)

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def step(self):
        """clip_grad_norm_ + AdamW.step + update_ema.  Parameters whose ``.grad`` is None are left untouched, like torch
        does; which ones those are must not change between steps (it is a property of the model: RoBERTa's pooler)."""
        grads = [p.grad for p in self.params]
        active = [g is not None for g in grads]
        if active != self._active:
            self._build_segments(active)
        if self.reducer is None or any(g is not None and g.data_ptr() != v.data_ptr() for g, v in zip(grads, self.grad_views)):
            have = [(v, g) for v, g in zip(self.grad_views, grads) if g is not None and g.data_ptr() != v.data_ptr()]
            if have:  # one fused multi-tensor copy into the flat buffer (already there when the exchange attached its views)
                torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        lrs = tuple(float(g["lr"]) for g in self.param_groups)
        if lrs != self._lr_last:  # three scalars, uploaded only when adjust_learning_rate changed them
            i = self._lr_slot = (self._lr_slot + 1) % len(self._lr_host)
            if self._lr_ev[i] is not None:
                self._lr_ev[i].synchronize()  # the copy that last read this staging slot (4 uploads ago) has completed
            self._lr_host[i][:3] = torch.tensor(lrs)
            self._lr_dev.copy_(self._lr_host[i], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._lr_ev[i] = ev
            self._lr_last = lrs
        L = _hip.lib()
        st = _hip.stream_ptr()
        _hip.check(L.td_grad_norm_clip(self.flat_g.data_ptr(), self.numel, self._segs, len(self._segs), self.max_norm, self._ws.data_ptr(),
                                       self._ws.numel(), self.norm_clip.data_ptr(), self.step_dev.data_ptr(), st), "td_grad_norm_clip")
        _hip.check(L.td_adamw_ema_step(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                       self.ema.data_ptr() if self.ema is not None else None, self.numel, self._segs, len(self._segs),
                                       self._lr_dev.data_ptr(), self.norm_clip.data_ptr(), self.step_dev.data_ptr(), self.betas[0], self.betas[1],
                                       self.eps, self.weight_decay, self.ema_decay, st), "td_adamw_ema_step")
        invalidate_prepared()  # the kernels changed the weights behind torch's version counters: prepared bf16 copies are stale

    # ---- checkpointing: torch.optim.AdamW's own layout (what the reference stores in checkpoint["optimizer"], main.py:681) ----
    def _torch_order(self) -> List[int]:
        """torch numbers the parameters group by group (main.py:381-405: everything else, backbone, text encoder)."""
        return [i for g in (0, 1, 2) for i, gg in enumerate(self.group_of) if gg == g]

    def state_dict(self):
        """{"state": {index: {"step", "exp_avg", "exp_avg_sq"}}, "param_groups": [...]} exactly like
        ``torch.optim.AdamW(param_dicts).state_dict()``: a checkpoint written here resumes in the reference and vice versa.
        (The EMA weights are not optimizer state: they live in ``ema_model.state_dict()``, checkpoint["model_ema"].)"""
        order = self._torch_order()
        step = float(self.step_dev.item())
        # which parameters have state: those the step has updated; before the first step after a resume, those the loaded
        # checkpoint held state for (a parameter without a gradient - RoBERTa's pooler - has none in torch's checkpoint either)
        active = self._active if self._active is not None else (self._loaded_active if self._loaded_active is not None else [True] * len(self.params))
        state = {}
        for ti, pi in enumerate(order):
            if step == 0 or not active[pi]:
                continue  # torch creates a parameter's state at its first update
            o, p = self.offsets[pi], self.params[pi]
            state[ti] = {"step": torch.tensor(step, dtype=torch.float32),
                         "exp_avg": self.exp_avg[o : o + p.numel()].view_as(p).clone(),
                         "exp_avg_sq": self.exp_avg_sq[o : o + p.numel()].view_as(p).clone()}
        defaults = dict(torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=1.0, betas=self.betas, eps=self.eps,
                                          weight_decay=self.weight_decay).param_groups[0])  # this torch version's key set
        defaults.pop("params")
        groups, base = [], 0
        for g in range(3):
            cnt = sum(1 for x in self.group_of if x == g)
            groups.append(dict(defaults, lr=float(self.param_groups[g]["lr"]), params=list(range(base, base + cnt))))
            base += cnt
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        if "names" in sd and "exp_avg" in sd:  # flat format written by earlier versions of this class
            assert sd["names"] == self.names
            self.exp_avg.copy_(sd["exp_avg"])
            self.exp_avg_sq.copy_(sd["exp_avg_sq"])
            self.step_dev.copy_(sd["step"])
            if self.ema is not None and sd.get("ema") is not None:
                self.ema.copy_(sd["ema"])
            for g, lr in zip(self.param_groups, sd["lrs"]):
                g["lr"] = lr
            return
        order = self._torch_order()
        groups = sd["param_groups"]
        assert len(groups) == 3 and [len(g["params"]) for g in groups] == [sum(1 for x in self.group_of if x == g) for g in range(3)], \
            "optimizer state does not match the model's three parameter groups (main.py:381-405)"
        flat_ids = [i for g in groups for i in g["params"]]
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        step = 0.0
        loaded = [False] * len(self.params)
        for ti, pi in zip(flat_ids, order):
            st = sd["state"].get(ti)
            if st is None:
                continue
            loaded[pi] = True
            o, p = self.offsets[pi], self.params[pi]
            self.exp_avg[o : o + p.numel()].view_as(p).copy_(st["exp_avg"])
            self.exp_avg_sq[o : o + p.numel()].view_as(p).copy_(st["exp_avg_sq"])
            step = max(step, float(st["step"]))
        self.step_dev.fill_(int(step))  # (one step counter: torch's per-parameter steps are equal for every parameter that has state)
        self._loaded_active = loaded if any(loaded) else None
        self._active = self._segs = None  # the activity map of earlier steps is stale: the loaded one governs until the next step decides
        for g, saved in zip(self.param_groups, groups):
            g["lr"] = float(saved["lr"])
        self._lr_last = None
