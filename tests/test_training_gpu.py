"""A short TRAINING TRAJECTORY, not one gradient: ten optimizer steps on one fixed tiny clip.

  * exact-fp32 mode + FusedAdamWEMA (clip_grad_norm 0.1, AdamW with the reference's three parameter groups, EMA) against the CPU
    oracle stepped by torch.optim.AdamW + torch.nn.utils.clip_grad_norm_ + the reference's update_ema formula (engine.py:146-151,
    main.py:381-413, util/optim.py:8-25): the loss of every step within 1e-3 (relative), the final weights and the final EMA
    weights against the oracle's;
  * bf16 mode (the kernels every throughput number uses, most of which exist in bf16 only) run the same ten steps: its loss
    stays within a stated band of the fp32 trajectory at every step and ends lower than it started - evidence that the bf16-only
    instances TRAIN, which a single good gradient does not show.
Dropout is off (eval mode) on both sides: the comparison needs identical arithmetic, not identical random streams; the learning
rates are 4x the reference defaults so that ten steps move the loss by several per cent."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

STEPS = 10
LR, LR_BACKBONE, LR_TEXT, WD, MAX_NORM, EMA_DECAY = 2e-4, 4e-5, 2e-4, 1e-4, 0.1, 0.9998


def _oracle_run(cfg, sd, batch):
    from oracle import tubedetr_oracle as O
    from oracle.weights import is_trainable
    from tubedetr_amd.optim import reference_group

    names = [k for k, v in sd.items() if v.requires_grad and is_trainable(k)]
    groups = [[], [], []]
    for k in names:
        groups[reference_group(k)].append(sd[k])
    opt = torch.optim.AdamW([{"params": groups[0]}, {"params": groups[1], "lr": LR_BACKBONE}, {"params": groups[2], "lr": LR_TEXT}], lr=LR, weight_decay=WD)
    ema = {k: sd[k].detach().clone() for k in names}
    losses = []
    for _ in range(STEPS):
        opt.zero_grad(set_to_none=True)
        loss, _, _, _ = O.train_step(sd, cfg, batch)
        loss.backward()
        torch.nn.utils.clip_grad_norm_([sd[k] for k in names], MAX_NORM)  # engine.py:147-148
        opt.step()
        with torch.no_grad():
            for k in names:
                ema[k].copy_(ema[k] * EMA_DECAY + (1.0 - EMA_DECAY) * sd[k].detach())  # util/optim.py:8-25
        losses.append(loss.item())
    return losses, {k: sd[k].detach().clone() for k in names}, ema


def _hip_run(cfg, sd0, batch, dtype):
    import tubedetr_amd
    from tubedetr_amd.harness import FixedTokenizer, batch_to, forward_step
    from tubedetr_amd.models import build_model
    from tubedetr_amd.optim import FusedAdamWEMA

    dev = torch.device("cuda:0")
    model, criterion, weight_dict = build_model(tubedetr_amd.default_args(stride=cfg.stride, compute_dtype=dtype))
    model.load_state_dict(sd0, strict=True)
    model.to(dev).eval()
    ema_model = copy.deepcopy(model)
    model.transformer.tokenizer = FixedTokenizer(batch["input_ids"], batch["attention_mask"])
    opt = FusedAdamWEMA(model, lr=LR, lr_backbone=LR_BACKBONE, text_encoder_lr=LR_TEXT, weight_decay=WD, max_norm=MAX_NORM, ema_model=ema_model, ema_decay=EMA_DECAY)
    b = batch_to(batch, dev)
    params = [p for p in model.parameters() if p.requires_grad]
    losses = []
    for _ in range(STEPS):
        for p in params:
            p.grad = None
        loss, _, _, _ = forward_step(model, criterion, weight_dict, b)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    torch.cuda.synchronize()
    return losses, {n: p.detach().float().cpu() for n, p in model.named_parameters() if p.requires_grad}, \
        {n: p.detach().float().cpu() for n, p in ema_model.named_parameters() if p.requires_grad}


def test_ten_step_training_trajectory_fp32_vs_oracle_and_bf16_band():
    from oracle.tubedetr_oracle import OracleConfig
    from oracle.weights import fill_state, state_spec, synthetic_batch

    cfg = OracleConfig(stride=2)
    batch = synthetic_batch(T=6, res=64, k=2, L=5, seed=3)
    sd0 = fill_state(state_spec(cfg), 11)
    sd = fill_state(state_spec(cfg), 11, requires_grad=True)  # the same values as leaves (trainable entries only)
    torch.set_num_threads(16)
    l_ref, w_ref, e_ref = _oracle_run(cfg, sd, batch)
    l32, w32, e32 = _hip_run(cfg, sd0, batch, torch.float32)
    l16, w16, e16 = _hip_run(cfg, sd0, batch, torch.bfloat16)
    print("oracle", [round(x, 5) for x in l_ref])
    print("fp32  ", [round(x, 5) for x in l32])
    print("bf16  ", [round(x, 5) for x in l16])
    assert l_ref[-1] < 0.99 * l_ref[0], "the trajectory must move: ten steps lower the oracle's loss by more than 1 %"
    for i, (a, b) in enumerate(zip(l32, l_ref)):
        assert abs(a - b) <= 1e-3 * abs(b), (i, a, b)
    # weights and EMA weights after ten steps: the UPDATE (w_10 - w_0) of every parameter, relative to the largest update of its tensor
    worst_w = worst_e = 0.0
    for k, wr in w_ref.items():
        if k not in w32:  # RoBERTa's pooler: no gradient, no state, not stepped on either side
            continue
        upd = (wr - sd0[k]).abs().max().item()
        if upd == 0.0:
            assert torch.equal(w32[k], sd0[k]), k
            continue
        worst_w = max(worst_w, ((w32[k] - wr).abs().max() / upd).item())
        eu = (e_ref[k] - sd0[k]).abs().max().item()
        worst_e = max(worst_e, ((e32[k] - e_ref[k]).abs().max() / max(eu, 1e-30)).item())
    print("worst relative error of a parameter's ten-step update", worst_w, "of its EMA update", worst_e)
    # Adam normalises every element's step to ~lr whatever the gradient's size: an element whose gradient is at fp32 rounding
    # level can flip its sign between two fp32 implementations, so the bound is on the tensor's largest update, with margin
    assert worst_w < 0.25 and worst_e < 0.25, (worst_w, worst_e)
    # bf16: same ten steps, the loss within 3 % of the fp32 trajectory at every step, and training (loss falls)
    for i, (a, b) in enumerate(zip(l16, l32)):
        assert abs(a - b) <= 3e-2 * abs(b), (i, a, b)
    assert l16[-1] < 0.99 * l16[0]
