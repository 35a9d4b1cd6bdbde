"""CPU tests of the host-side mirror of the reference interface (no kernels are launched): state-dict layout, freeze
rule, NestedTensor, mask down-sampling index rule, criterion vs the oracle, refusal to run without a GPU."""
import pytest
import torch
import torch.nn.functional as F

import tubedetr_amd
from oracle.tubedetr_oracle import OracleConfig, criterion as oracle_criterion, weight_dict as oracle_weight_dict
from oracle.weights import is_trainable, state_spec
from tubedetr_amd.models import build_model
from tubedetr_amd.util.misc import NestedTensor


@pytest.fixture(scope="module")
def built():
    torch.manual_seed(0)
    return build_model(tubedetr_amd.default_args(device="cpu"))


def test_state_dict_matches_reference_layout(built):
    model, _, wd = built
    spec = state_spec(OracleConfig())  # pinned against the reference's own state_dict in oracle/gen_golden.py
    sd = model.state_dict()
    assert len(sd) == 923 and list(sd) == list(spec)
    assert all(tuple(sd[k].shape) == tuple(spec[k]) for k in spec)
    assert {k for k, p in model.named_parameters() if p.requires_grad} == {k for k in spec if is_trainable(k)}
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == 185234182
    assert wd == oracle_weight_dict(OracleConfig()) and len(wd) == 24
    # reference init facts: fast_residual zero-initialised, FrozenBN identity, time table is a buffer
    assert model.transformer.fast_residual.weight.abs().sum() == 0
    assert "transformer.time_embed.te" in dict(model.named_buffers())
    for n in ("backbone", "text_encoder"):
        assert any(n in k for k, _ in model.named_parameters())


def test_flag_variants_build_like_the_reference():
    m, _, _ = build_model(tubedetr_amd.default_args(device="cpu", fast=False, no_tsa=True))
    assert not hasattr(m.transformer, "fast_encoder") and len(m.state_dict()) == 919
    with pytest.raises(ValueError):
        build_model(tubedetr_amd.default_args(device="cpu", fast_mode="no-such-mode"))
    with pytest.raises(NotImplementedError):  # dilation / GroupNorm / timm backbones: other models, not flags of this one (SURVEY.md 8a')
        build_model(tubedetr_amd.default_args(device="cpu", dilation=True))


def test_forward_refuses_cpu(built):
    model, _, _ = built
    x = NestedTensor(torch.zeros(2, 3, 64, 64), torch.zeros(2, 64, 64, dtype=torch.bool))
    with pytest.raises((AssertionError, RuntimeError)):
        model(x, [4], ["a caption"], encode_and_save=True, samples_fast=NestedTensor(torch.zeros(4, 3, 64, 64), torch.zeros(4, 64, 64, dtype=torch.bool)))


def test_nested_tensor_from_clips():
    clips = [torch.randn(3, 4, 20, 30), torch.randn(3, 2, 24, 28)]
    nt = NestedTensor.from_tensor_list(clips)
    assert nt.tensors.shape == (6, 3, 24, 30) and nt.mask.shape == (6, 24, 30)
    assert torch.equal(nt.tensors[1, :, :20, :30], clips[0][:, 1]) and nt.tensors[0, :, 20:].abs().sum() == 0
    assert not nt.mask[0, :20, :30].any() and nt.mask[0, 20:].all() and nt.mask[4, :, 28:].all()
    imgs = NestedTensor.from_tensor_list([torch.randn(3, 10, 12), torch.randn(3, 8, 16)])
    assert imgs.tensors.shape == (2, 3, 10, 16) and imgs.mask[1, 8:].all() and not imgs.mask[1, :8, :16].any()


@pytest.mark.parametrize("size", [(352, 11), (224, 7), (356, 12), (353, 12), (100, 4), (587, 19)])
def test_mask_downsample_index_rule_matches_interpolate(size):
    from tubedetr_amd.models.backbone import _nearest_index

    n_in, n_out = size
    m = torch.rand(1, n_in, n_in) > 0.5
    ref = F.interpolate(m[None].float(), size=(n_out, n_out)).bool()[0]
    idx = _nearest_index(n_out, n_in, torch.device("cpu"))
    assert torch.equal(m[:, idx][:, :, idx], ref)


def test_criterion_matches_oracle(built):
    _, crit, _ = built
    g = torch.Generator().manual_seed(3)
    b, T = 2, 6
    durations = [6, 4]
    mk = lambda n: torch.cat([torch.rand(n, 2, generator=g) * 0.4 + 0.3, torch.rand(n, 2, generator=g) * 0.2 + 0.1], 1)
    boxes = mk(sum(durations))
    layer = lambda: {"pred_boxes": mk(sum(durations)), "pred_sted": torch.randn(b, T, 2, generator=g),
                     "weights": torch.softmax(torch.randn(b, T, T, generator=g), -1), "ca_weights": torch.rand(b * T, 1, 5, generator=g)}
    out = layer()
    out["aux_outputs"] = [layer() for _ in range(5)]
    tm = torch.zeros(b, T, dtype=torch.bool)
    for i, d in enumerate(durations):
        tm[i, :d] = True
    inter = [[0, 5], [1, 3]]
    got = crit(out, [{"boxes": x[None]} for x in boxes], inter, tm)
    ref = oracle_criterion(out, boxes, inter, tm, OracleConfig())
    assert set(got) == set(ref) and len(got) == 24
    for k in ref:
        assert torch.allclose(got[k], ref[k], rtol=1e-5, atol=1e-6), k


def test_text_encoder_is_not_silently_replaced(monkeypatch):
    """Without the explicit opt-in a missing roberta-base raises like the reference does (transformer.py:130-135);
    with it BOTH stand-ins are used together (never a pretrained model with the hash tokenizer)."""
    from tubedetr_amd.models import transformer as tr

    monkeypatch.delenv("TD_ALLOW_RANDOM_TEXT_ENCODER", raising=False)
    with pytest.raises(RuntimeError, match="TD_ALLOW_RANDOM_TEXT_ENCODER"):
        tr._load_text_encoder("roberta-base-that-does-not-exist")
    monkeypatch.setenv("TD_ALLOW_RANDOM_TEXT_ENCODER", "1")
    with pytest.warns(UserWarning, match="RANDOM-INIT"):
        tok, enc = tr._load_text_encoder("roberta-base-that-does-not-exist")
    assert isinstance(tok, tr.HashTokenizer) and enc.config.hidden_size == 768


def test_ablation_variants_keep_the_reference_parameter_names():
    """--fast_mode / --learn_time_embed / --position_embedding learned / --stride 0 are accepted (SURVEY.md 8a') and build the
    same state-dict keys, shapes and trainable set as the reference's modules (recorded by oracle/gen_golden.py VARIANTS)."""
    import os

    import numpy as np

    import tubedetr_amd
    from oracle.gen_golden import VARIANTS
    from tubedetr_amd.models import build_model

    gold_dir = os.path.join(os.path.dirname(__file__), "golden")
    for name, (_, ckw, extra) in VARIANTS.items():
        gold = np.load(os.path.join(gold_dir, name + ".npz"))
        model, _, _ = build_model(tubedetr_amd.default_args(device="cpu", **ckw, **extra))
        assert list(model.state_dict().keys()) == [str(k) for k in gold["meta.state_keys"]], name
        assert sorted(k for k, p in model.named_parameters() if p.requires_grad) == [str(k) for k in gold["meta.trainable"]], name
        assert sum(p.numel() for p in model.parameters()) == int(gold["meta.n_params"]), name
    # pass_pos_and_query=False: constructible like the reference; forward fails where the reference's does
    model, _, _ = build_model(tubedetr_amd.default_args(device="cpu", pass_pos_and_query=False))
    import pytest as _pytest

    with _pytest.raises(TypeError):
        model.transformer(encode_and_save=False)


def test_cross_attention_projections_commute_to_the_query_side():
    """The identity csrc/cross_attn.hip is built on, in float64 on the CPU against torch's own nn.MultiheadAttention: with ONE query
    per frame, softmax(q . (W_k (x + pos) + b_k)) (W_v x + b_v) = W_v,h (sum_s p[h,s] x_s) + b_v with scores u_h . (x_s + pos_s),
    u_h = W_k,h^T q_h / sqrt(d_h) - the key bias shifts every score of a row equally and drops out."""
    torch.manual_seed(3)
    F_, S, E, H = 5, 37, 256, 8
    hd = E // H
    mha = torch.nn.MultiheadAttention(E, H, dropout=0.0).double()
    with torch.no_grad():
        mha.in_proj_bias.normal_()
        mha.out_proj.bias.normal_()
    tgt, qpos = torch.randn(F_, E).double(), torch.randn(F_, E).double()
    mem, pos = torch.randn(F_, S, E).double(), torch.randn(F_, S, E).double()
    key_pad = torch.rand(F_, S) < 0.3
    key_pad[:, 0] = False
    # reference: (L = 1, N = F_, E) query, (S, F_, E) keys / values - transformer.py:727-740
    ref, ref_w = mha((tgt + qpos)[None], (mem + pos).transpose(0, 1), mem.transpose(0, 1), key_padding_mask=key_pad)
    W, b = mha.in_proj_weight.detach(), mha.in_proj_bias.detach()
    q = (tgt + qpos) @ W[:E].t() + b[:E]
    u = torch.einsum("fhj,hjc->fhc", q.view(F_, H, hd) / hd**0.5, W[E : 2 * E].view(H, hd, E))          # W_k,h^T q_h
    sc = torch.einsum("fhc,fsc->fhs", u, mem + pos).masked_fill(key_pad[:, None, :], float("-inf"))   # no key bias
    p = sc.softmax(-1)
    z = torch.einsum("fhs,fsc->fhc", p, mem)                                                           # weighted memory rows
    ctx = torch.einsum("fhc,hjc->fhj", z, W[2 * E :].view(H, hd, E)).reshape(F_, E) + b[2 * E :] * p.sum(-1).repeat_interleave(hd, 1)
    out = ctx @ mha.out_proj.weight.detach().t() + mha.out_proj.bias.detach()
    assert torch.allclose(out, ref[0].detach(), atol=1e-10)
    assert torch.allclose(p.mean(1), ref_w[:, 0].detach(), atol=1e-12)


def test_roofline_families_are_named_by_kernels_of_the_committed_profile():
    """Every kernel family bench.py reports (tools/kernel_families.py, shared with the PMC aggregation) is made of kernels that occur in the
    committed kernel-stats CSV of the round, and every name stem in a family's key is a substring of such a kernel name - the bench line
    cannot name a kernel that is no longer launched.  The PMC JSONs bench.py reads carry the exact names per family."""
    import csv
    import json
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    sys.path.insert(0, root)
    from kernel_families import FAMILY_OF_PROF_ID, family

    import bench

    tag = bench.PMC_TRAFFIC.split("_")[0]
    names = [r["Name"] for r in csv.DictReader(open(os.path.join(root, "profiles", f"{tag}_bench_cfg3x16_kernel_stats.csv")))]
    by_family = {}
    for n in names:
        by_family.setdefault(family(n), []).append(n)
    for fam_id, key in FAMILY_OF_PROF_ID.items():
        assert by_family.get(key), f"family {fam_id} ({key}): no kernel of the committed profile belongs to it"
        for stem in key.split(" + "):
            stem = stem.split("<")[0]
            assert any(stem in n for n in by_family[key]), (key, stem)
    traffic = json.load(open(os.path.join(root, "profiles", bench.PMC_TRAFFIC)))
    mfma = json.load(open(os.path.join(root, "profiles", bench.PMC_MFMA)))
    for fam_id, key in FAMILY_OF_PROF_ID.items():
        for src in (traffic, mfma):
            assert key in src and src[key]["kernels"], (key, "missing from the PMC aggregate")
            assert all(k in names for k in src[key]["kernels"]), key  # exact names of the kernel-stats CSV
