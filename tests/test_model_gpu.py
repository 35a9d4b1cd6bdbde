"""End-to-end parity on a real MI355X: the product model (tubedetr_amd.models, HIP kernels through the C ABI,
exact-fp32 mode) against (1) the golden vectors produced by the reference itself and (2) the CPU oracle run on the
same seeded inputs.  Bar (BASELINE.json north_star): box / start-end logits within 1e-3, attention argmax indices
bit-exact.  Also checks the 24 losses and the gradients of every trainable parameter."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
LOGIT_TOL = 1e-3


def _build(cfg, dtype=torch.float32):
    import tubedetr_amd
    from tubedetr_amd.models import build_model

    args = tubedetr_amd.default_args(stride=cfg.stride, fast=cfg.fast, no_tsa=cfg.no_tsa, compute_dtype=dtype)
    torch.manual_seed(0)
    return build_model(args)


def _load(model, cfg, seed):
    from oracle.weights import fill_state, state_spec

    sd = fill_state(state_spec(cfg), seed)
    model.load_state_dict(sd, strict=True)
    return sd


# Rows in which the HIP argmax may differ from the reference's because the reference itself holds its maximum more than once (to within 1e-6):
# a RECORDED constant per fixture, not an open hatch - anything not listed is 0, and a fixture that needs more than its constant fails.
TIED_ROWS_ALLOWED = {"v_notime_T6-5_res64_k2": 12}


def _same_argmax(got, ref, tie=1e-6):
    """Attention indices bit-exact - except in rows whose maximum the reference itself holds more than once (to within ``tie``): there
    the index is decided by the last bit of either implementation.  It happens in exactly one fixture: without time encodings
    (--no_time_embed) all time queries of a clip are identical in the first decoder layer, its temporal self-attention is uniform
    (1 / t in every column) and the reference's own argmax over such a row is 0, 2 or 4 depending on the row.
    Returns (every row equal or tied, number of rows that needed the tie rule)."""
    ga, ra = got.argmax(-1), ref.argmax(-1)
    ref_at_got = np.take_along_axis(ref, ga[..., None], -1)[..., 0]
    tied = (ga != ra) & (ref.max(-1) - ref_at_got <= tie)
    return bool(((ga == ra) | tied).all()), int(tied.sum())


def _compare_with_golden(model, criterion, weight_dict, batch, gold, fixture=None):
    from tubedetr_amd.harness import FixedTokenizer, batch_to, forward_step

    dev = torch.device("cuda:0")
    model.to(dev).eval()  # dropout off = the parity mode the fixtures were captured in
    model.transformer.tokenizer = FixedTokenizer(batch["input_ids"], batch["attention_mask"])
    loss, ld, out, cache = forward_step(model, criterion, weight_dict, batch_to(batch, dev))

    def cpu(x):
        return x.detach().float().cpu().numpy()

    for k in ("img_memory", "pos_embed", "query_embed", "text_memory", "text_memory_resized"):
        np.testing.assert_allclose(cpu(cache[k]), gold["cache." + k], rtol=0, atol=LOGIT_TOL, err_msg=k)
    for k in ("mask", "query_mask", "text_attention_mask"):
        if "cache." + k not in gold.files:
            assert cache[k] is None, k
            continue
        assert np.array_equal(cache[k].cpu().numpy().astype(bool), gold["cache." + k]), k
    layers = out.get("aux_outputs", []) + [out]
    assert len(layers) == gold["out.pred_boxes"].shape[0]  # (one entry without --aux_loss)
    for key in ("pred_boxes", "pred_sted", "weights", "ca_weights"):
        if "out." + key not in gold.files:  # --no_sted / --no_guided_attn: the reference's output dict has no such key (tubedetr.py:233-237)
            assert key not in out, key
            continue
        got = np.stack([cpu(o[key]) for o in layers])
        err = np.abs(got - gold["out." + key]).max()
        assert err < LOGIT_TOL, (key, err)
    n_tied = 0
    for key in ("weights", "ca_weights"):  # attention indices bit-exact
        if "out." + key not in gold.files:
            continue
        got = np.stack([cpu(o[key]) for o in layers])
        ok, n = _same_argmax(got, gold["out." + key])
        assert ok, key
        n_tied += n
    assert n_tied <= TIED_ROWS_ALLOWED.get(fixture, 0), (fixture, n_tied, "rows decided by the tie rule: more than the recorded constant")

    names = sorted(ld)
    assert names == list(gold["loss.names"])
    np.testing.assert_allclose([ld[k].item() for k in names], gold["loss.values"], rtol=1e-3, atol=1e-4)
    assert abs(loss.item() - float(gold["loss.total"])) <= 1e-3 * abs(float(gold["loss.total"]))  # (the weighted sum: weight_dict's coefficients)

    loss.backward()
    params = dict(model.named_parameters())
    for k, n, h in zip(gold["grad.names"], gold["grad.norms"], gold["grad.heads"]):
        g = params[str(k)].grad
        assert g is not None, k
        gn = g.double().norm().item()
        assert abs(gn - n) <= 5e-3 * n + 1e-4, (k, gn, n)
        hh = g.flatten()[:8].float().cpu().numpy()
        np.testing.assert_allclose(hh, h[: hh.size], rtol=2e-2, atol=2e-3 * max(n, 1e-2), err_msg=str(k))
    return params


@pytest.mark.parametrize("name", ["a_b1_T8_res96_k4", "b_b2_T8-6_res64_k4", "c_nofast_T6_res64_k2", "d_notsa_T5_res64_k5"])
def test_model_matches_reference_golden_fp32(name):
    from oracle.gen_golden import CASES, WEIGHT_SEED
    from oracle.tubedetr_oracle import OracleConfig
    from oracle.weights import synthetic_batch

    bkw, ckw = CASES[name]
    cfg = OracleConfig(**ckw)
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    model, criterion, weight_dict = _build(cfg)
    _load(model, cfg, WEIGHT_SEED)
    params = _compare_with_golden(model, criterion, weight_dict, synthetic_batch(**bkw), gold, fixture=name)
    unused = [k for k, p in params.items() if p.requires_grad and p.grad is None]
    assert all("pooler" in k for k in unused), unused


@pytest.mark.parametrize("name", ["v_gating_T6_res64_k2", "v_pool_T6_res64_k3", "v_transformer_T4_res64_k2", "v_noslow_T6_res64_k2", "v_stride0_T5-3_res64",
                                  "v_learned_T6_res64_k2", "v_boxesonly_T6_res64_k2", "v_notime_T6-5_res64_k2", "v_frozen_T6_res64_k2"])
def test_ablation_flags_match_reference_golden_fp32(name):
    """main.py's ablation flags (--fast_mode gating | pool | transformer | noslow, --stride 0, --learn_time_embed,
    --position_embedding learned, --no_sted + --no_guided_attn + --no_aux_loss, --no_time_embed, --freeze_backbone + --freeze_text_encoder + --sigma 2 + other loss
    coefficients; SURVEY.md 8a'): accepted, computed on this library's kernels + stock PyTorch ops for the
    variant's own arithmetic, and checked against the reference's own outputs, losses and gradients (the CPU oracle does
    not restate the variants: these vectors pin the product directly)."""
    import tubedetr_amd
    from oracle.gen_golden import VARIANTS, WEIGHT_SEED
    from oracle.weights import fill_state, synthetic_batch
    from tubedetr_amd.models import build_model

    bkw, ckw, extra = VARIANTS[name]
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    torch.manual_seed(0)
    model, criterion, weight_dict = build_model(tubedetr_amd.default_args(compute_dtype=torch.float32, **ckw, **extra))
    sd0 = model.state_dict()
    assert list(sd0.keys()) == [str(k) for k in gold["meta.state_keys"]]
    assert sorted(k for k, p in model.named_parameters() if p.requires_grad) == [str(k) for k in gold["meta.trainable"]]
    model.load_state_dict(fill_state({k: tuple(v.shape) for k, v in sd0.items()}, WEIGHT_SEED), strict=True)
    params = _compare_with_golden(model, criterion, weight_dict, synthetic_batch(**bkw), gold, fixture=name)
    no_grad = sorted(k for k, p in params.items() if p.requires_grad and p.grad is None)
    assert no_grad == [str(k) for k in gold["meta.no_grad"]]  # what the variant leaves out of its graph (noslow: the slow trunk + encoder)


def test_model_matches_oracle_fp32_padded_masks():
    """A case the golden set does not hold (different seed / padding / durations), checked against the CPU oracle."""
    from oracle.tubedetr_oracle import OracleConfig, train_step
    from oracle.weights import fill_state, state_spec, synthetic_batch
    from tubedetr_amd.harness import FixedTokenizer, batch_to, forward_step

    cfg = OracleConfig(stride=3)
    bkw = dict(T=7, res=96, k=3, L=7, seed=21, durations=[7, 7], pad_w=33, text_pad=3)
    batch = synthetic_batch(**bkw)
    sd = fill_state(state_spec(cfg), 5)
    with torch.no_grad():
        _, ld_ref, out_ref, cache_ref = train_step(sd, cfg, batch)
    model, criterion, weight_dict = _build(cfg)
    model.load_state_dict(sd, strict=True)
    dev = torch.device("cuda:0")
    model.to(dev).eval()
    model.transformer.tokenizer = FixedTokenizer(batch["input_ids"], batch["attention_mask"])
    with torch.no_grad():
        _, ld, out, cache = forward_step(model, criterion, weight_dict, batch_to(batch, dev))
    assert (cache["img_memory"].float().cpu() - cache_ref["img_memory"]).abs().max() < LOGIT_TOL
    for a, b in zip(out["aux_outputs"] + [out], out_ref["aux_outputs"] + [out_ref]):
        for key in ("pred_boxes", "pred_sted", "weights", "ca_weights"):
            assert (a[key].float().cpu() - b[key]).abs().max() < LOGIT_TOL, key
        assert torch.equal(a["weights"].argmax(-1).cpu(), b["weights"].argmax(-1))
        assert torch.equal(a["ca_weights"].argmax(-1).cpu(), b["ca_weights"].argmax(-1))
    for k in ld_ref:
        assert abs(ld[k].item() - ld_ref[k].item()) < 1e-3 * max(1.0, abs(ld_ref[k].item())), k


def test_model_bf16_close_to_fp32_and_trains():
    """Throughput mode (bf16 MFMA): outputs stay close to the fp32 path (tolerance 0.1 on logits / 0.05 on boxes: bf16
    has 8 mantissa bits and the path is 104 convs + 12 transformer layers deep), every trainable parameter except
    RoBERTa's unused pooler receives a finite gradient, train-mode dropout runs."""
    from oracle.tubedetr_oracle import OracleConfig
    from oracle.weights import fill_state, state_spec, synthetic_batch
    from tubedetr_amd.harness import FixedTokenizer, batch_to, forward_step

    cfg = OracleConfig(stride=4)
    batch = synthetic_batch(T=8, res=96, k=4, L=6, seed=31)
    sd = fill_state(state_spec(cfg), 9)
    dev = torch.device("cuda:0")
    outs = {}
    for dt in (torch.float32, torch.bfloat16):
        model, criterion, weight_dict = _build(cfg, dt)
        model.load_state_dict(sd, strict=True)
        model.to(dev).eval()
        model.transformer.tokenizer = FixedTokenizer(batch["input_ids"], batch["attention_mask"])
        with torch.no_grad():
            _, _, out, _ = forward_step(model, criterion, weight_dict, batch_to(batch, dev))
        outs[dt] = out
    assert (outs[torch.float32]["pred_boxes"] - outs[torch.bfloat16]["pred_boxes"]).abs().max() < 0.05
    assert (outs[torch.float32]["pred_sted"] - outs[torch.bfloat16]["pred_sted"]).abs().max() < 0.1
    model.train()
    torch.manual_seed(3)
    loss, ld, _, _ = forward_step(model, criterion, weight_dict, batch_to(batch, dev))
    assert torch.isfinite(loss)
    loss.backward()
    for k, p in model.named_parameters():
        if p.requires_grad and "pooler" not in k:
            assert p.grad is not None and torch.isfinite(p.grad).all(), k


def test_trunk_backward_stage_by_stage_is_bit_identical_in_deterministic_mode():
    """td_resnet_bwd(only_stage = 3, 2, 1) - the trunk's backward issued stage by stage, one batched weight-gradient launch per stage, for a
    gradient exchange that leaves in pieces - against the single pass: the same kernels on the same operands in the same order, so in
    deterministic mode (one work item per weight-gradient tile) every gradient is identical bit for bit, fp32 and bf16."""
    import tubedetr_amd
    from oracle.tubedetr_oracle import OracleConfig
    from oracle.weights import fill_state, state_spec, synthetic_batch
    from tubedetr_amd.harness import FixedTokenizer, backward_in_stages, batch_to, forward_step, set_split_backward

    cfg = OracleConfig(stride=2)
    batch = synthetic_batch(T=4, res=64, k=2, L=5, seed=43)
    sd = fill_state(state_spec(cfg), 17)
    dev = torch.device("cuda:0")
    tubedetr_amd.set_deterministic(True)
    try:
        for dt in (torch.float32, torch.bfloat16):
            res = {}
            for mode in ("single", "stages"):
                model, criterion, weight_dict = _build(cfg)
                model.load_state_dict(sd, strict=True)
                model.to(dev).eval()
                model.set_compute_dtype(dt)
                model.transformer.tokenizer = FixedTokenizer(batch["input_ids"], batch["attention_mask"])
                set_split_backward(model, mode == "stages")
                loss, _, _, _ = forward_step(model, criterion, weight_dict, batch_to(batch, dev))
                seen = []
                if mode == "stages":
                    backward_in_stages(model, loss, after_trunk_stage=lambda k_, ws_: seen.append((k_, len(ws_))))
                    assert [k_ for k_, _ in seen] == [1, 2, 3] and all(n_ > 0 for _, n_ in seen)
                else:
                    loss.backward()
                torch.cuda.synchronize()
                res[mode] = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
            assert res["single"].keys() == res["stages"].keys()
            bad = [n for n in res["single"] if not torch.equal(res["single"][n], res["stages"][n])]
            assert not bad, (dt, bad[:8])
    finally:
        tubedetr_amd.set_deterministic(False)


def test_dedupe_is_proven_from_the_inputs_not_flagged():
    """TubeDETR._slow_is_strided_fast: the slow clip as an index list over the very buffer the fast frames are, with the host copy of that
    list equal to 0, k, 2k, ... of every video (what data.ClipPipeline hands over) is a proof; anything else - another buffer, a shifted or
    permuted list, no host copy, a duration that does not match - is not, and then both passes run."""
    import tubedetr_amd
    from tubedetr_amd.models import build_model
    from tubedetr_amd.util.misc import FrameSources, NestedTensor

    dev = torch.device("cuda:0")
    model, _, _ = build_model(tubedetr_amd.default_args(stride=4, resnet_layers=(1, 1, 1, 1), enc_layers=1, dec_layers=1))
    assert model.slow_frames_are_strided_fast is None  # the default: prove it per call
    video = torch.zeros(14, 3, 32, 32, dtype=torch.uint8, device=dev)
    durations = [8, 6]
    good = (0, 4, 8, 12)

    def nt(x, n):
        return NestedTensor(x, torch.zeros(n, 32, 32, dtype=torch.bool, device=dev))

    def slow(base, idx, host):
        return nt(FrameSources([(base, torch.tensor(idx, dtype=torch.int32, device=dev))], None, [host]), len(idx))

    fast = nt(video, 14)
    assert model._slow_is_strided_fast(slow(video, good, good), fast, durations)
    assert model._slow_is_strided_fast(slow(video, good, good), nt(FrameSources([(video, None)]), 14), durations)
    assert not model._slow_is_strided_fast(slow(video.clone(), good, good), fast, durations)          # equal pixels, another buffer: not provable
    assert not model._slow_is_strided_fast(slow(video, good, None), fast, durations)                   # no host copy of the list
    assert not model._slow_is_strided_fast(slow(video, (0, 4, 9, 13), (0, 4, 9, 13)), fast, durations)  # not every 4th frame of the second video
    assert not model._slow_is_strided_fast(slow(video, (4, 0, 8, 12), (4, 0, 8, 12)), fast, durations)  # permuted
    assert not model._slow_is_strided_fast(slow(video, good, good), fast, [7, 7])                      # other durations: 0, 4, 7, 11 expected
    assert not model._slow_is_strided_fast(nt(video[::4], 4), fast, durations)                         # a plain tensor: nothing to reason about
    model.slow_frames_are_strided_fast = False
    assert not model._slow_is_strided_fast(slow(video, good, good), fast, durations)                   # switched off: the reference's two passes
    model.slow_frames_are_strided_fast = True
    assert model._slow_is_strided_fast(nt(video[::4], 4), fast, durations)                             # the caller vouches


def test_dedupe_slow_frames_is_exact(monkeypatch):
    """slow_frames_are_strided_fast=True (slow frames not recomputed: one trunk pass, or - beyond the frames one saved pass can
    address, forced here with TD_TRUNK_MAX_FRAMES - the slow pass plus a no-grad pass over the OTHER fast frames) gives the same
    outputs and gradients as the two-set pass when the slow frames really are fast[::k]."""
    from oracle.tubedetr_oracle import OracleConfig
    from oracle.weights import fill_state, state_spec, synthetic_batch
    from tubedetr_amd.harness import FixedTokenizer, batch_to, forward_step

    cfg = OracleConfig(stride=4)
    batch = synthetic_batch(T=8, res=64, k=4, L=5, seed=41, durations=[8, 6])
    sd = fill_state(state_spec(cfg), 13)
    dev = torch.device("cuda:0")
    res = {}
    for flag in (False, True, "split"):
        monkeypatch.delenv("TD_TRUNK_MAX_FRAMES", raising=False)
        if flag == "split":
            monkeypatch.setenv("TD_TRUNK_MAX_FRAMES", "6")  # 14 fast frames, 4 of them slow: no single pass
        model, criterion, weight_dict = _build(cfg)
        model.load_state_dict(sd, strict=True)
        model.to(dev).eval()
        model.slow_frames_are_strided_fast = bool(flag)
        model.transformer.tokenizer = FixedTokenizer(batch["input_ids"], batch["attention_mask"])
        loss, _, out, _ = forward_step(model, criterion, weight_dict, batch_to(batch, dev))
        loss.backward()
        res[flag] = (out["pred_boxes"].detach().clone(), out["pred_sted"].detach().clone(),
                     model.backbone[0].body.layer3[5].conv2.weight.grad.clone(), model.input_proj.weight.grad.clone())
    for other in (True, "split"):
        for a, b in zip(res[False], res[other]):
            assert (a - b).abs().max() <= 1e-5 * max(1.0, b.abs().max().item()), other


@pytest.mark.parametrize("text_stream", ["0", "1"])
def test_hip_graph_replay_matches_eager_step(text_stream):
    """bench.py measures the step replayed from a HIP graph: the replay (static inputs, device-side dropout step
    counter, batched weight-gradient job table re-uploaded by a captured copy node, in-place weight re-preparation) must
    produce the loss and the gradients of the same step launched eagerly.  text_stream "1" = bench.py's default at N = 1
    (RoBERTa forward and backward on their own stream: a forked branch of the graph), "0" = the single-stream graph the
    staged N > 1 mode is built on."""
    import tubedetr_amd
    from oracle.weights import synthetic_batch
    from tubedetr_amd import ops as ops_
    from tubedetr_amd.functional import invalidate_prepared
    from tubedetr_amd.harness import FixedTokenizer, batch_to, forward_step
    from tubedetr_amd.models import build_model

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model, criterion, weight_dict = build_model(tubedetr_amd.default_args(stride=2, compute_dtype=torch.bfloat16))
    model.to(dev).eval()  # dropout off: the eager and the captured run must then agree up to fp32-atomic ordering
    batch = batch_to(synthetic_batch(T=6, res=64, k=2, L=5, seed=4), dev)
    model.transformer.tokenizer = FixedTokenizer(batch["input_ids"], batch["attention_mask"])
    params = [p for p in model.parameters() if p.requires_grad]

    def run():
        invalidate_prepared()
        loss, _, _, _ = forward_step(model, criterion, weight_dict, batch)
        loss.backward()
        return loss

    os.environ["TD_TEXT_STREAM"] = text_stream
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                for p in params:
                    p.grad = None
                eager_loss = run()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        eager = {i: p.grad.detach().float().clone() for i, p in enumerate(params) if p.grad is not None}
        for p in params:
            p.grad = None
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            g_loss = run()
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
    finally:
        os.environ.pop("TD_TEXT_STREAM", None)
        ops_.set_dropout_counter(None)
    assert abs(g_loss.item() - eager_loss.item()) <= 2e-3 * abs(eager_loss.item())
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    checked, bad = 0, []
    for i, p in enumerate(params):
        if p.grad is None:
            assert i not in eager, names[i]
            continue
        a, b = p.grad.detach().float(), eager[i]
        scale = b.abs().max().clamp_min(1e-4)  # (RoBERTa's key biases have a mathematically zero gradient: ~1e-8 noise)
        err = ((a - b).abs().max() / scale).item()
        if not err < 2e-2:
            bad.append((names[i], err, scale.item()))
        checked += 1
    assert not bad, bad[:12]
    assert checked > 300


def test_no_grad_trunk_pass_is_chunked_identically(monkeypatch):
    """A no-grad trunk pass over more frames than one 32-bit buffer descriptor can address (the 800 fast frames of an fp32
    8-clip batch) is cut into chunks: same features as the single pass, for indexed / multi-part / uint8-with-extents sources."""
    import tubedetr_amd
    from tubedetr_amd.models import build_model
    from tubedetr_amd.util.misc import FrameSources

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model, _, _ = build_model(tubedetr_amd.default_args(stride=2, resnet_layers=(1, 1, 1, 1), enc_layers=1, dec_layers=1))
    model.to(dev).eval()
    body = model.backbone[0].body
    g = torch.Generator().manual_seed(0)
    a = torch.randn(5, 3, 32, 48, generator=g).to(dev)
    b = torch.randn(4, 3, 32, 48, generator=g).to(dev)
    idx = torch.tensor([4, 0, 2], dtype=torch.int32, device=dev)
    u8 = torch.randint(0, 256, (6, 3, 32, 48), generator=g, dtype=torch.uint8).to(dev)
    vhw = torch.tensor([[32, 48], [20, 48], [32, 30], [32, 48], [16, 16], [32, 48]], dtype=torch.int32, device=dev)
    cases = [FrameSources([(a, idx), (b, None)]), FrameSources([(u8, None)], [vhw]), FrameSources([(u8, torch.tensor([5, 1, 1, 4, 2], dtype=torch.int32, device=dev))], [vhw])]
    for dt in (torch.float32, torch.bfloat16):
        for fs in cases:
            with torch.no_grad():
                monkeypatch.delenv("TD_TRUNK_MAX_FRAMES", raising=False)
                whole = body(fs, dt)
                monkeypatch.setenv("TD_TRUNK_MAX_FRAMES", "2")
                parts = body(fs, dt)
            assert whole.shape == parts.shape == (fs.n_frames, 1, 2, 2048)
            assert torch.equal(whole, parts)
