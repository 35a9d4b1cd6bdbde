"""Fused SetCriterion kernel (csrc/criterion.hip: keep-gather + 24 losses + their derivatives in one launch) against the
CPU oracle's criterion (models/tubedetr.py:270-372,397-460 restated) and torch autograd through it: loss values and the
gradients with respect to boxes, start/end logits and attention weights, for b = 1 and for b = 2 with different durations
(time mask) and partial annotation intervals."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", [dict(durations=[12], inter=[[0, 11]]), dict(durations=[9, 6], inter=[[2, 7], [0, 3]]),
                                  dict(durations=[100], inter=[[10, 80]])])
def test_fused_criterion_matches_oracle_and_autograd(case):
    from oracle.tubedetr_oracle import OracleConfig, criterion as oracle_criterion, weight_dict as oracle_weight_dict
    from tubedetr_amd.models.tubedetr import SetCriterion

    durations, inter = case["durations"], case["inter"]
    b, T, nl = len(durations), max(durations), 6
    g = torch.Generator().manual_seed(11)
    mk = lambda *s: torch.cat([torch.rand(*s, 2, generator=g) * 0.5 + 0.25, torch.rand(*s, 2, generator=g) * 0.3 + 0.1], -1)
    boxes = mk(nl, b * T).requires_grad_()                       # every frame, every layer (cxcywh)
    sted = (torch.randn(nl, b, T, 2, generator=g) * 2).requires_grad_()
    weights = torch.softmax(torch.randn(nl, b, T, T, generator=g) * 2, -1).requires_grad_()
    keep = torch.tensor([i * T + f for i, (s, e) in enumerate(inter) for f in range(s, e + 1)], dtype=torch.long)
    tgt = mk(keep.numel())
    tm = torch.zeros(b, T, dtype=torch.bool)
    for i, d in enumerate(durations):
        tm[i, :d] = True
    cfg = OracleConfig()
    wd = oracle_weight_dict(cfg)
    # reference: the oracle's per-layer criterion + engine.py's weighted sum, autograd for the gradients
    layers = [{"pred_boxes": boxes[l][keep], "pred_sted": sted[l], "weights": weights[l]} for l in range(nl)]
    out = dict(layers[-1])
    out["aux_outputs"] = layers[:-1]
    ref = oracle_criterion(out, tgt, inter, tm, cfg)
    total_ref = sum(ref[k] * wd[k] for k in ref)
    total_ref.backward()
    # fused
    dev = torch.device("cuda:0")
    crit = SetCriterion(["boxes", "sted", "guided_attn"], sigma=cfg.sigma)
    bx, st_, ws = (x.detach().to(dev).requires_grad_() for x in (boxes, sted, weights))
    got = crit.forward_fused({"pred_boxes": bx, "pred_sted": st_, "weights": ws}, keep.to(dev), tgt.to(dev), inter, tm.to(dev))
    assert set(got) == set(ref) and len(got) == 24
    for k in ref:
        assert abs(got[k].item() - ref[k].item()) <= 2e-5 * max(1.0, abs(ref[k].item())), (k, got[k].item(), ref[k].item())
    total = (crit.last_loss_matrix * crit.weight_matrix(wd, nl, dev)).sum()
    assert abs(total.item() - total_ref.item()) <= 2e-5 * abs(total_ref.item())
    total.backward()
    for name, a, r in (("boxes", bx.grad, boxes.grad), ("sted", st_.grad, sted.grad), ("weights", ws.grad, weights.grad)):
        err = (a.cpu() - r).abs().max().item()
        assert err <= 2e-4 * r.abs().max().item() + 1e-7, (name, err, r.abs().max().item())
