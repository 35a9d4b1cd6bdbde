"""The CPU oracle (oracle/tubedetr_oracle.py) against vectors produced by the reference itself
(oracle/gen_golden.py, run in the build container).  This is what pins the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle.gen_golden import CASES, VARIANTS, WEIGHT_SEED
from oracle.tubedetr_oracle import OracleConfig, train_step
from oracle.weights import fill_state, state_spec, synthetic_batch

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def roberta_free_threads():
    torch.set_num_threads(min(8, os.cpu_count() or 1))


# the head / loss switches and --no_time_embed are restated by the oracle too (OracleConfig.sted / guided_attn / aux_loss / no_time_embed):
# their vectors live with the ablation variants (the output and loss dicts lose keys) and pin the oracle like the four base cases
# (v_frozen: --sigma 2 and non-default loss coefficients; its freeze flags only remove gradients the vectors then do not list)
ORACLE_VARIANTS = ["v_boxesonly_T6_res64_k2", "v_notime_T6-5_res64_k2", "v_frozen_T6_res64_k2"]


@pytest.mark.parametrize("name", list(CASES) + ORACLE_VARIANTS)
def test_oracle_matches_reference(name, roberta_free_threads):
    bkw, ckw = CASES[name] if name in CASES else VARIANTS[name][:2]
    cfg = OracleConfig(**ckw)
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    spec = state_spec(cfg)
    assert len(spec) == int(gold["meta.n_state_keys"])
    sd = fill_state(spec, WEIGHT_SEED, requires_grad=True)
    batch = synthetic_batch(**bkw)
    loss, ld, out, cache = train_step(sd, cfg, batch)

    for k in ("img_memory", "pos_embed", "query_embed", "text_memory", "text_memory_resized"):
        np.testing.assert_allclose(cache[k].detach().numpy(), gold["cache." + k], rtol=1e-4, atol=2e-5, err_msg=k)
    for k in ("mask", "query_mask", "text_attention_mask"):
        assert np.array_equal(cache[k].numpy(), gold["cache." + k]), k

    layers = out.get("aux_outputs", []) + [out]
    assert len(layers) == gold["out.pred_boxes"].shape[0]  # (one entry without --aux_loss)
    for key in ("pred_boxes", "pred_sted", "weights", "ca_weights"):
        if "out." + key not in gold.files:  # --no_sted / --no_guided_attn: the reference's output dict has no such key
            assert key not in out, key
            continue
        got = np.stack([o[key].detach().numpy() for o in layers])
        np.testing.assert_allclose(got, gold["out." + key], rtol=1e-4, atol=1e-5, err_msg=key)
    # "attention indices bit-exact": argmax over keys of TSA and cross-attention weights, all layers
    for key in ("weights", "ca_weights"):
        if "out." + key not in gold.files:
            continue
        got = np.stack([o[key].detach().numpy() for o in layers])
        assert np.array_equal(got.argmax(-1), gold["out." + key].argmax(-1)), key

    names = sorted(ld)
    assert names == list(gold["loss.names"])
    np.testing.assert_allclose([ld[k].item() for k in names], gold["loss.values"], rtol=2e-5, atol=1e-6)
    assert abs(loss.item() - float(gold["loss.total"])) < 1e-4 * abs(float(gold["loss.total"]))

    loss.backward()
    for k, n, h in zip(gold["grad.names"], gold["grad.norms"], gold["grad.heads"]):
        g = sd[str(k)].grad
        assert g is not None, k
        assert abs(g.double().norm().item() - n) <= 2e-3 * n + 1e-5, (k, g.norm().item(), n)
        hh = g.flatten()[:8].numpy()
        np.testing.assert_allclose(hh, h[: hh.size], rtol=5e-3, atol=1e-5 * max(n, 1.0), err_msg=str(k))
