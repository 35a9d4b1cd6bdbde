"""RoBERTa forward / backward on the gfx950 kernels.

The reference calls HF ``RobertaModel`` (models/transformer.py:130-135, 252-263); run as a stock PyTorch module that is
~50 tiny hipBLASLt / elementwise launches per layer and direction on 30 tokens - a quarter of the GPU time of a step at
one clip per GPU.  ``run_roberta`` computes the same function from the SAME module's parameters (state-dict keys,
checkpoints and the optimizer's "text_encoder" parameter group are untouched) with this package's kernels:

  embeddings   word + position + token-type gathers (torch index ops: integer work), LayerNorm + dropout (HIP)
  per layer    q / k / v projections, attention output projection (+dropout), intermediate (+GELU), output (+dropout):
               td_conv_gemm with fused bias / dropout epilogues; residual + LayerNorm: td_add_layernorm; GELU: td_gelu;
               the 72 weight / bias gradients join the transformer's deferred batched launch (functional._wgrad)
  attention    12 heads x 64 on the in-house attention core (td_mha_fwd / td_mha_bwd, head-dim-64 instance): the projected
               rows are consumed where they lie (heads = column blocks), no transposes or copies

fp32 compute dtype = exact-fp32 kernels (parity mode), bf16 = MFMA throughput mode.  The pooler is never evaluated
(unused by the reference too: the reason it needs ``find_unused_parameters``).
"""
from __future__ import annotations

from typing import Optional

import torch
from .. import functional as Fk


def position_ids_from_input_ids(input_ids: torch.Tensor, padding_idx: int) -> torch.Tensor:
    """HF create_position_ids_from_input_ids: padded tokens keep padding_idx, the others count from padding_idx + 1."""
    mask = input_ids.ne(padding_idx).int()
    return (torch.cumsum(mask, dim=1).type_as(mask) * mask).long() + padding_idx


def run_roberta(hf, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor], compute_dtype: torch.dtype, training: bool,
                no_padding: bool = False) -> torch.Tensor:
    """last_hidden_state [B, L, hidden] in ``compute_dtype`` of the HF RobertaModel ``hf`` (its parameters are used in
    place).  attention_mask: [B, L] 1 = token, 0 = padding (None / no_padding=True: no padded positions)."""
    cfg = hf.config
    emb = hf.embeddings
    B, L = input_ids.shape
    H, nh = cfg.hidden_size, cfg.num_attention_heads
    hd = H // nh
    p_hid = float(cfg.hidden_dropout_prob)
    p_att = float(cfg.attention_probs_dropout_prob)
    if cfg.hidden_act != "gelu":
        raise NotImplementedError(f"hidden_act={cfg.hidden_act!r}: only RoBERTa's erf GELU is implemented")
    if no_padding:  # synthetic / unpadded captions: positions are padding_idx + 1 ..., known without looking at the ids
        pos_ids = torch.arange(emb.padding_idx + 1, emb.padding_idx + 1 + L, device=input_ids.device).unsqueeze(0).expand(B, L)
    else:
        pos_ids = position_ids_from_input_ids(input_ids, emb.padding_idx)
    x = emb.word_embeddings(input_ids) + emb.token_type_embeddings.weight[0] + emb.position_embeddings(pos_ids)  # fp32 [B, L, H]
    rows = Fk.cast(x.reshape(B * L, H), compute_dtype)
    rows = Fk.add_layernorm(rows, None, emb.LayerNorm.weight, emb.LayerNorm.bias, emb.LayerNorm.eps)
    rows = Fk.dropout(rows, p_hid, training)
    key_pad = None
    if attention_mask is not None and not no_padding:
        key_pad = attention_mask.ne(1)  # [B, L]: True = padded key, ignored
    for layer in hf.encoder.layer:
        sa, so = layer.attention.self, layer.attention.output
        q = Fk.linear(rows, sa.query.weight, sa.query.bias)
        k = Fk.linear(rows, sa.key.weight, sa.key.bias)
        v = Fk.linear(rows, sa.value.weight, sa.value.bias)
        ctx = Fk.attention_core(q, k, v, key_pad, B, L, L, nh, dropout_p=p_att, training=training)  # heads = column blocks of 64: no transposes
        a = Fk.linear(ctx, so.dense.weight, so.dense.bias, dropout_p=p_hid, training=training)
        rows = Fk.add_layernorm(a, rows, so.LayerNorm.weight, so.LayerNorm.bias, so.LayerNorm.eps)
        h = Fk.gelu(Fk.linear(rows, layer.intermediate.dense.weight, layer.intermediate.dense.bias))
        o = Fk.linear(h, layer.output.dense.weight, layer.output.dense.bias, dropout_p=p_hid, training=training)
        rows = Fk.add_layernorm(o, rows, layer.output.LayerNorm.weight, layer.output.LayerNorm.bias, layer.output.LayerNorm.eps)
    return rows.view(B, L, H)
