"""Fused optimizer tail (tubedetr_amd/optim.py: grad-norm + clip, AdamW with the reference's three parameter groups, EMA;
3 HIP launches per step over flat buffers) against torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW (main.py:381-413,
engine.py:147-151) + update_ema (util/optim.py:8-25) on the real model's 185 M parameters, three steps, with a learning
rate change (adjust_learning_rate, util/optim.py:28-95) in between.  RoBERTa's pooler never gets a gradient: both sides
must leave it untouched."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref_update_ema(model, model_ema, decay):  # util/optim.py:8-25
    with torch.no_grad():
        msd = model.state_dict()
        for k, ema_v in model_ema.state_dict().items():
            ema_v.copy_(ema_v * decay + (1.0 - decay) * msd[k].detach())


def test_fused_clip_adamw_ema_matches_torch():
    import tubedetr_amd
    from tubedetr_amd.models import build_model
    from tubedetr_amd.optim import FusedAdamWEMA, reference_group

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model, _, _ = build_model(tubedetr_amd.default_args(stride=4))
    model.to(dev)
    ref = copy.deepcopy(model)
    ema_ref, ema_fused = copy.deepcopy(model), copy.deepcopy(model)
    lr, lr_b, lr_t, wd, max_norm, decay = 5e-5, 1e-5, 5e-5, 1e-4, 0.1, 0.9998
    groups = [{"params": [p for n, p in ref.named_parameters() if "backbone" not in n and "text_encoder" not in n and p.requires_grad]},
              {"params": [p for n, p in ref.named_parameters() if "backbone" in n and p.requires_grad], "lr": lr_b},
              {"params": [p for n, p in ref.named_parameters() if "text_encoder" in n and p.requires_grad], "lr": lr_t}]
    opt_ref = torch.optim.AdamW(groups, lr=lr, weight_decay=wd)
    opt = FusedAdamWEMA(model, lr=lr, lr_backbone=lr_b, text_encoder_lr=lr_t, weight_decay=wd, max_norm=max_norm, ema_model=ema_fused, ema_decay=decay)
    assert [reference_group(n) for n in opt.names].count(1) == len(groups[1]["params"])
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    p_ref = dict(ref.named_parameters())
    p_new = dict(model.named_parameters())
    g = torch.Generator(device=dev).manual_seed(5)
    p0 = {n: p_ref[n].detach().clone() for n in names}
    for step in range(3):
        for n in names:
            if "pooler" in n:  # never reached by the loss (the reason the reference needs find_unused_parameters)
                p_ref[n].grad = p_new[n].grad = None
                continue
            scale = 10.0 ** float(torch.randint(-4, 1, (1,)).item())  # gradients of very different magnitudes, clipping active
            gr = torch.randn(p_ref[n].shape, generator=g, device=dev) * scale
            p_ref[n].grad, p_new[n].grad = gr.clone(), gr.clone()
        if step == 2:  # adjust_learning_rate writes the three group rates
            for o in (opt_ref, opt):
                o.param_groups[0]["lr"], o.param_groups[1]["lr"], o.param_groups[2]["lr"] = lr * 0.1, lr_b * 0.1, lr_t * 0.5
        total = torch.nn.utils.clip_grad_norm_(ref.parameters(), max_norm)
        opt_ref.step()
        _ref_update_ema(ref, ema_ref, decay)
        opt.step()
        torch.cuda.synchronize()
        assert abs(opt.norm_clip[0].item() - total.item()) <= 1e-5 * total.item()
    worst = worst_ema = 0.0
    e_ref, e_new = dict(ema_ref.named_parameters()), dict(ema_fused.named_parameters())
    for n in names:
        ua, ub = p_new[n].detach() - p0[n], p_ref[n].detach() - p0[n]  # compare the UPDATES (~1e-4): a wrong group rate / step count shows
        if "pooler" not in n:
            worst = max(worst, ((ua - ub).abs().max() / ub.abs().max().clamp_min(1e-12)).item())
        worst_ema = max(worst_ema, ((e_new[n] - e_ref[n]).abs().max() / e_ref[n].abs().max().clamp_min(1e-3)).item())
    # (fp32 cancellation in (p - p0) bounds the comparison: LayerNorm weights ~1 carry 1.2e-7 of rounding against ~5e-5 updates)
    assert worst < 1e-2 and worst_ema < 1e-6, (worst, worst_ema)
    pool = [n for n in names if "pooler" in n]
    assert pool and all(torch.equal(p_new[n], p_ref[n]) for n in pool)
    assert opt.step_dev.item() == 3
    # the model still computes with the updated weights: parameters are views of the flat buffer
    assert p_new[names[0]].data_ptr() == opt.flat_p.data_ptr()


def test_optimizer_checkpoint_is_torch_adamw_layout_and_ema_resumes():
    """checkpoint["optimizer"] of the reference is a torch AdamW state_dict (main.py:681): a checkpoint written by the fused
    optimizer loads into torch.optim.AdamW built like main.py:381-413 and continues identically, and the other way round;
    the EMA buffer starts from ema_model's own (resumed) weights, not from the model's."""
    import tubedetr_amd
    from tubedetr_amd.models import build_model
    from tubedetr_amd.optim import FusedAdamWEMA

    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    model, _, _ = build_model(tubedetr_amd.default_args(stride=4, resnet_layers=(1, 1, 1, 1), enc_layers=1, dec_layers=1))
    model.to(dev)
    ref = copy.deepcopy(model)
    ema = copy.deepcopy(model)
    with torch.no_grad():
        for p in ema.parameters():
            p.add_(0.25)  # "loaded from checkpoint['model_ema']": differs from the model's weights
    ema_before = {n: p.detach().clone() for n, p in ema.named_parameters()}

    def torch_opt(m):
        groups = [{"params": [p for n, p in m.named_parameters() if "backbone" not in n and "text_encoder" not in n and p.requires_grad]},
                  {"params": [p for n, p in m.named_parameters() if "backbone" in n and p.requires_grad], "lr": 1e-5},
                  {"params": [p for n, p in m.named_parameters() if "text_encoder" in n and p.requires_grad], "lr": 5e-5}]
        return torch.optim.AdamW(groups, lr=5e-5, weight_decay=1e-4)

    opt = FusedAdamWEMA(model, max_norm=0.0, ema_model=ema, ema_decay=0.9998)
    for n, p in ema.named_parameters():
        if p.requires_grad:
            assert torch.equal(p, ema_before[n]), n  # not overwritten by the model's weights
    opt_ref = torch_opt(ref)
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    p_new, p_ref = dict(model.named_parameters()), dict(ref.named_parameters())
    g = torch.Generator(device=dev).manual_seed(2)

    def grads():
        for n in names:
            if "pooler" in n:
                p_ref[n].grad = p_new[n].grad = None
                continue
            gr = torch.randn(p_ref[n].shape, generator=g, device=dev) * 1e-2
            p_ref[n].grad, p_new[n].grad = gr.clone(), gr.clone()

    for _ in range(2):
        grads()
        opt.step()
        opt_ref.step()
    sd_fused, sd_torch = opt.state_dict(), opt_ref.state_dict()
    assert set(sd_fused) == {"state", "param_groups"}
    assert [g_["params"] for g_ in sd_fused["param_groups"]] == [g_["params"] for g_ in sd_torch["param_groups"]]
    assert set(sd_fused["param_groups"][0]) == set(sd_torch["param_groups"][0])
    assert set(sd_fused["state"]) == set(sd_torch["state"])  # the pooler's entries are absent on both sides
    for i, st in sd_torch["state"].items():
        assert float(sd_fused["state"][i]["step"]) == float(st["step"]) == 2.0
        for k_ in ("exp_avg", "exp_avg_sq"):
            a, b_ = sd_fused["state"][i][k_], st[k_]
            assert a.shape == b_.shape and (a - b_).abs().max().item() <= 1e-4 * max(1e-12, b_.abs().max().item()) + 1e-12, (i, k_)  # (fused multiply-adds round differently)
    # cross-load: fused checkpoint -> torch AdamW on a fresh copy, torch checkpoint -> fused on another; one more step each
    m2, m3 = copy.deepcopy(ref), copy.deepcopy(ref)
    o2 = torch_opt(m2)
    o2.load_state_dict(sd_fused)
    o3 = FusedAdamWEMA(m3, max_norm=0.0)
    o3.load_state_dict(sd_torch)
    assert o3.step_dev.item() == 2
    # a save straight after the resume (before any step) writes back what was loaded: no state for the pooler, no zero moments
    sd_again = o3.state_dict()
    assert set(sd_again["state"]) == set(sd_torch["state"])
    for i, st in sd_torch["state"].items():
        assert float(sd_again["state"][i]["step"]) == 2.0 and torch.equal(sd_again["state"][i]["exp_avg"].cpu(), st["exp_avg"].cpu())
    grads()
    p2, p3 = dict(m2.named_parameters()), dict(m3.named_parameters())
    for n in names:
        p2[n].grad = None if p_ref[n].grad is None else p_ref[n].grad.clone()
        p3[n].grad = None if p_ref[n].grad is None else p_ref[n].grad.clone()
    before = {n: p_ref[n].detach().clone() for n in names}
    opt_ref.step()
    o2.step()
    o3.step()
    torch.cuda.synchronize()
    for n in names:
        upd = (p_ref[n].detach() - before[n])
        scale = upd.abs().max().clamp_min(1e-12)
        assert ((p2[n].detach() - before[n] - upd).abs().max() / scale).item() < 1e-2, n
        assert ((p3[n].detach() - before[n] - upd).abs().max() / scale).item() < 1e-2, n
