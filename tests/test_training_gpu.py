"""A short TRAINING TRAJECTORY, not one gradient: ten optimizer steps on one fixed tiny clip.

  * exact-fp32 mode + FusedAdamWEMA (clip_grad_norm 0.1, AdamW with the reference's three parameter groups, EMA) against the CPU
    oracle stepped by torch.optim.AdamW + torch.nn.utils.clip_grad_norm_ + the reference's update_ema formula (engine.py:146-151,
    main.py:381-413, util/optim.py:8-25): the losses of the first three steps within 1e-3 (relative), all ten within 5 % (two fp32
    implementations under AdamW are a chaotic pair: see the comment at the assertion), the direction and length of every
    parameter's ten-step update and of its EMA copy against the oracle's;
  * bf16 mode (the kernels every throughput number uses, most of which exist in bf16 only) run the same ten steps: its loss
    stays within a stated band of the fp32 trajectory at every step and ends lower than it started - evidence that the bf16-only
    instances TRAIN, which a single good gradient does not show.
(This test also found a real defect: functional.prepared() counted a parameter's storage offset twice once FusedAdamWEMA had moved the
parameters into its flat buffer - the first forward after constructing the optimizer read far outside the weights.)
Dropout is off (eval mode) on both sides: the comparison needs identical arithmetic, not identical random streams.  Learning rates are
the reference's defaults: at 4x those the random-init model's loss falls 4x in ten steps and two fp32 implementations drift apart
by 2.5e-3 at step 3 (Adam turns gradient elements at rounding level into full-size steps of either sign) - measured, round 4."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

STEPS = 10
LR, LR_BACKBONE, LR_TEXT, WD, MAX_NORM, EMA_DECAY = 5e-5, 1e-5, 5e-5, 1e-4, 0.1, 0.9998  # the reference's defaults (main.py:38-44, 61-62)


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
    assert l_ref[-1] < 0.5 * l_ref[0], "the trajectory must move: ten steps halve the random-init model's loss"
    # Free-running fp32 vs the oracle.  The first three losses agree to 1e-3 (measured 3e-7, 0, 7e-5); from then on two fp32
    # implementations drift apart - AdamW's first steps are lr * sign(g) per element, so every element whose gradient is at
    # rounding level takes a full-size step of either sign - by 0.1 .. 2.1 % of the loss over three runs (the fp32 HIP path is not
    # run-to-run deterministic either: its weight gradients accumulate with fp32 atomics), bounded here at 5 %.
    for i, (a, b) in enumerate(zip(l32, l_ref)):
        assert abs(a - b) <= (1e-3 if i < 3 else 5e-2) * abs(b), (i, a, b)
    # the ten-step UPDATE of every parameter tensor (w_10 - w_0) and of its EMA copy: direction against the oracle's
    worst_w, worst_e, n_cmp = 1.0, 1.0, 0
    for k, wr in w_ref.items():
        if k not in w32:  # RoBERTa's pooler: no gradient, no state, not stepped on either side
            continue
        du_ref, du = (wr - sd0[k]).double().flatten(), (w32[k] - sd0[k]).double().flatten()
        # a tensor whose gradient is (numerically) zero only decays: its "update" is noise that Adam amplifies differently on the two
        # sides (decoder layer 0's self-attention in-projection: the queries are zero there) - compared are the tensors that take
        # real steps: update norm above 5 % of what full-size steps of every element would give
        if du_ref.norm() < 0.05 * STEPS * LR_BACKBONE * du_ref.numel() ** 0.5:
            assert du.norm() < 0.10 * STEPS * LR * du.numel() ** 0.5, k
            continue
        worst_w = min(worst_w, (du @ du_ref / (du.norm() * du_ref.norm())).item())
        # the EMA copy moves by (1 - decay) * steps = 2e-3 of the weight update: at fp32 rounding level of the stored values, so its
        # DIRECTION is noise - what is checked is the value: within fp32 rounding + that fraction of the two sides' weight difference
        e_err = (e32[k] - e_ref[k]).abs().max().item()
        e_tol = 1e-6 * e_ref[k].abs().max().item() + 1e-2 * (wr - sd0[k]).abs().max().item()
        worst_e = min(worst_e, 1.0 - e_err / max(e_tol, 1e-30))
        assert e_err <= e_tol, (k, e_err, e_tol)
        assert (e_ref[k] - sd0[k]).abs().max().item() > 0 and (e32[k] - sd0[k]).abs().max().item() > 0, k  # it did move, on both sides
        assert abs(du.norm() / du_ref.norm() - 1.0) < 0.1, (k, du.norm().item(), du_ref.norm().item())
        n_cmp += 1
    print("parameters compared", n_cmp, "worst cosine of a ten-step update", worst_w, "smallest EMA margin (1 - err / tol)", worst_e)
    assert n_cmp > 250 and worst_w > 0.9, (n_cmp, worst_w, worst_e)
    # bf16 (most of its kernels exist in bf16 only): the same ten steps stay within 10 % of the ORACLE's trajectory at every step and reach
    # the same loss level - they train.  Measured (round 5): 0.2 / 0.1 / 1.4 / 0.4 / 1.2 / 1.5 / 5.0 / 8.3 / 4.2 / 4.7 % - the bf16 trajectory lags by
    # a fraction of a step on the steep part of the curve.  The reference is the oracle, not the fp32 HIP run above: the bf16 trajectory is
    # bit-reproducible at this size and so is the oracle's, while the free-running fp32 HIP losses wander by +-0.3 % from run to run
    # (fp32 atomics; test_deterministic_mode_... below) - against THEM the same bf16 numbers measured 9.1 .. 9.4 %, a bound that a bad draw
    # of the fp32 run could have crossed.
    for i, (a, b) in enumerate(zip(l16, l_ref)):
        assert abs(a - b) <= 0.10 * abs(b), (i, a, b)
    assert l16[-1] < 0.5 * l16[0]


def test_deterministic_mode_makes_the_fp32_trajectory_bit_reproducible():
    """TD_DETERMINISTIC=1 (tubedetr_amd.set_deterministic): two runs of the same ten optimizer steps in the exact-fp32 mode give
    IDENTICAL losses, weights and EMA copies, bit for bit - the regression anchor a parity mode needs (without it the weight
    gradients, LayerNorm's dgamma / dbeta and the bias gradients are combined with fp32 atomics in arrival order and two runs
    differ at rounding level, which AdamW's sign-like first steps turn into 0.1 - 2 % of the loss by step ten)."""
    import tubedetr_amd
    from oracle.tubedetr_oracle import OracleConfig
    from oracle.weights import fill_state, state_spec, synthetic_batch

    cfg = OracleConfig(stride=2)
    batch = synthetic_batch(T=6, res=64, k=2, L=5, seed=3)
    sd0 = fill_state(state_spec(cfg), 11)
    tubedetr_amd.set_deterministic(True)
    try:
        la, wa, ea = _hip_run(cfg, sd0, batch, torch.float32)
        lb, wb, eb = _hip_run(cfg, sd0, batch, torch.float32)
    finally:
        tubedetr_amd.set_deterministic(False)
    assert la == lb, (la, lb)
    assert la[-1] < 0.5 * la[0]
    for k in wa:
        assert torch.equal(wa[k], wb[k]), k
        assert torch.equal(ea[k], eb[k]), k
    # ... and, with every reduction sequential, it follows the ORACLE's trajectory far more closely than the free-running mode does (whose
    # bound above is 1e-3 for three steps and 5 % after): measured 0 / 0 / 1e-7 / 4e-6 / 1.7e-4 for the first five losses, then 1.5e-3 / 3.9e-3 /
    # 1.16e-2 / 6.2e-3 / 5.7e-3 (round 5; two different fp32 implementations under AdamW still part ways - later, and less)
    sd = fill_state(state_spec(cfg), 11, requires_grad=True)
    torch.set_num_threads(16)
    l_ref, _, _ = _oracle_run(cfg, sd, batch)
    print("oracle       ", [round(x, 5) for x in l_ref])
    print("deterministic", [round(x, 5) for x in la])
    for i, (a, b) in enumerate(zip(la, l_ref)):
        assert abs(a - b) <= (2e-5 if i < 4 else (1e-3 if i < 5 else 2.5e-2)) * abs(b), (i, a, b)
