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
