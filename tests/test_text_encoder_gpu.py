"""RoBERTa on the HIP kernels (tubedetr_amd/models/text_encoder.py) against the HF RobertaModel it takes its parameters
from (the reference's text encoder, models/transformer.py:130-135,252-263): last_hidden_state and the gradients of every
parameter, fp32 mode (exact-fp32 kernels) tight, bf16 mode loose; with and without padded captions."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(layers=3):
    from transformers import RobertaConfig, RobertaModel

    torch.manual_seed(0)
    m = RobertaModel(RobertaConfig(vocab_size=50265, max_position_embeddings=514, type_vocab_size=1, pad_token_id=1, layer_norm_eps=1e-5,
                                   num_hidden_layers=layers))
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "LayerNorm.weight" in n:
                p.uniform_(0.8, 1.2)
            elif n.endswith("bias"):
                p.normal_(0, 0.05)
    return m.to("cuda:0").eval()


@pytest.mark.parametrize("padded", [False, True])
def test_hip_roberta_matches_hf_module(padded):
    from tubedetr_amd.models.text_encoder import run_roberta

    dev = torch.device("cuda:0")
    hf = _model()
    ref = copy.deepcopy(hf)
    g = torch.Generator().manual_seed(3)
    B, L = 2, 9
    ids = torch.randint(3, 50000, (B, L), generator=g)
    ids[:, 0] = 0
    att = torch.ones(B, L, dtype=torch.long)
    if padded:
        ids[1, 6:] = 1
        ids[1, 5] = 2
        att[1, 6:] = 0
    ids, att = ids.to(dev), att.to(dev)
    w = torch.randn(B, L, 768, generator=g).to(dev) * att[..., None]  # no loss on padded positions (their states are don't-care)
    out_ref = ref(input_ids=ids, attention_mask=att).last_hidden_state
    (out_ref * w).sum().backward()
    out = run_roberta(hf, ids, att, torch.float32, training=False, no_padding=not padded)
    (out * w).sum().backward()
    err = ((out - out_ref) * att[..., None]).abs().max().item()
    assert err < 2e-4 * out_ref.abs().max().item(), err
    gr = dict(ref.named_parameters())
    gmax = max(p.grad.abs().max().item() for p in ref.parameters() if p.grad is not None)  # key biases have a mathematically zero gradient:
    worst = 0.0                                                                           # errors are judged against the global gradient scale
    for n, p in hf.named_parameters():
        if "pooler" in n:
            assert p.grad is None
            continue
        a, b = p.grad, gr[n].grad
        assert a is not None and b is not None, n
        e = ((a - b).abs().max() / (b.abs().max() + 1e-3 * gmax)).item()
        worst = max(worst, e)
        assert e < 2e-3, (n, e)
    # bf16 throughput mode stays close
    out16 = run_roberta(hf, ids, att, torch.bfloat16, training=False, no_padding=not padded).float()
    assert ((out16 - out_ref) * att[..., None]).abs().max().item() < 0.06 * out_ref.abs().max().item()
