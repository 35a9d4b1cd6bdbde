"""Video-text encoder + space-time decoder behind the reference's ``models.transformer`` interface
(models/transformer.py:24-176 constructor / init, 178-491 forward, 494-773 layers, 780-801 build).

Same class names, constructor keywords, parameter names (``encoder.layers.0.self_attn.in_proj_weight`` ...)
and the same two-mode ``forward`` contract (encode -> 9-key memory_cache, decode -> hs / weights /
cross_weights).  What differs is the engine:

  * every matmul, attention, LayerNorm, residual add and dropout runs in hand-written gfx950 kernels through
    ``tubedetr_amd.functional`` (forward and backward); torch only builds index tensors and concatenates;
  * activations are batch-major rows [batch*tokens, 256] in the model's compute dtype; the reference's
    sequence-first tensors in ``memory_cache`` are zero-copy permuted views of them;
  * work the reference does and discards is skipped (SURVEY.md 8a': the encoder's attention-weight averaging,
    transformer.py:346), pos+memory for the cross-attention keys is formed once for all six layers, and the
    temporal replication / text replication are index gathers built once per (durations) instead of Python loops.

RoBERTa (HF ``RobertaModel``, third party) stays a PyTorch-ROCm module, as in SURVEY.md 8a E2.
"""
from __future__ import annotations

import copy
import math
import os
from typing import List, Optional

import torch
from torch import Tensor, nn

from .. import functional as Fk
from .. import ops as _ops
from ..util.misc import LRUCache
from .position_encoding import TimeEmbeddingLearned, TimeEmbeddingSine

FAST_MODES_IN_HIP = ("",)  # the default slow-fast aggregation; the ablation variants below run their aggregation on stock PyTorch ops
FAST_MODES = ("", "gating", "transformer", "pool", "noslow")  # main.py --fast_mode choices (transformer.py:113-122)


class MultiheadAttention(nn.Module):
    """Parameter container with nn.MultiheadAttention's names / shapes / init; computed by functional.MHAFn."""

    def __init__(self, embed_dim: int, num_heads: int, dropout: float = 0.0):
        super().__init__()
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)  # holder only (weight, bias)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.0)

    def run_prekv(self, q_in, kv, layer, key_pad, B, Lq, Lk, need_weights, out_dropout, training, q_pos=None):
        """Keys / values already projected for all layers (functional.cross_kv): this layer reads column block ``layer``."""
        return Fk.multihead_attention_prekv(q_in, kv, layer, self.in_proj_weight, self.in_proj_bias, self.out_proj.weight, self.out_proj.bias, key_pad,
                                            B, Lq, Lk, self.num_heads, need_weights, attn_dropout=self.dropout, out_dropout=out_dropout, training=training,
                                            q_pos=q_pos)

    def run_q1(self, q_in, anchor, key_pad, F, S, need_weights, out_dropout, training, q_pos=None):
        """One query per frame against that frame's S memory rows, key / value projections on the query side (functional.CrossQ1Fn)."""
        return Fk.multihead_attention_q1(q_in, anchor, self.in_proj_weight, self.in_proj_bias, self.out_proj.weight, self.out_proj.bias, key_pad,
                                         F, S, self.num_heads, need_weights, attn_dropout=self.dropout, out_dropout=out_dropout, training=training,
                                         q_pos=q_pos)

    def run(self, q_in, k_in, v_in, key_pad, B, Lq, Lk, need_weights, out_dropout, training, q_pos=None):
        """q_pos: the positional operand of the query (self-attention: query and key) projection - with_pos_embed() of the
        reference layers as a second operand stream of the projection GEMM, never added in memory (functional.MHAFn)."""
        return Fk.multihead_attention(q_in, k_in, v_in, self.in_proj_weight, self.in_proj_bias, self.out_proj.weight,
                                      self.out_proj.bias, key_pad, B, Lq, Lk, self.num_heads, need_weights,
                                      attn_dropout=self.dropout, out_dropout=out_dropout, training=training, q_pos=q_pos)


class TransformerEncoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu"):
        super().__init__()
        if activation != "relu":
            raise NotImplementedError("only the reference default activation (relu) is fused in the HIP FFN")
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.p = dropout

    def forward(self, src: Tensor, pos: Optional[Tensor], key_pad: Optional[Tensor], B: int, S: int) -> Tensor:
        """src, pos: rows [B*S, d] (batch-major); post-norm layer of transformer.py:629-646."""
        # q = k = src + pos (transformer.py:637-640): pos is the projection's second operand stream, not a materialised sum
        a, _ = self.self_attn.run(src, None, src, key_pad, B, S, S, False, self.p, self.training, q_pos=pos)
        src = Fk.add_layernorm(a, src, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        f = Fk.ffn(src, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias, self.p, self.training)
        return Fk.add_layernorm(f, src, self.norm2.weight, self.norm2.bias, self.norm2.eps)


class TransformerEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers, norm=None, return_weights=False):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(encoder_layer) for _ in range(num_layers)])
        self.num_layers = num_layers
        self.norm = norm
        self.return_weights = return_weights  # accepted for signature parity; the weights are dead in the reference

    def forward(self, src, key_pad, pos, B, S):
        out = src
        for layer in self.layers:
            out = layer(out, pos, key_pad, B, S)
        if self.norm is not None:
            out = Fk.add_layernorm(out, None, self.norm.weight, self.norm.bias, self.norm.eps)
        return out


class TransformerDecoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu", no_tsa=False):
        super().__init__()
        if activation != "relu":
            raise NotImplementedError("only relu")
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.cross_attn_image = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.norm4 = nn.LayerNorm(d_model)
        self.p = dropout
        self.no_tsa = no_tsa

    def forward(self, tgt, query_pos, mem, pos, query_mask, memory_mask, b: int, t: int, S: int, kv=None, index: int = 0):
        """tgt/query_pos rows [b*t, d] (video-major frames); mem / pos = memory and its positional rows [b*t*S, d]
        (transformer.py:684-751).  kv: the hoisted key / value projections of all layers (TransformerDecoder.forward).
        Every with_pos_embed() of the reference layer (tgt + query_pos for the self-attention's q = k and the cross-attention's
        query, memory + pos for its keys) is a second operand stream of the projection GEMM."""
        if self.no_tsa:  # every frame attends to itself only: sequence length 1 (transformer.py:701-711)
            a, w = self.self_attn.run(tgt, None, tgt, None, b * t, 1, 1, True, self.p, self.training, q_pos=query_pos)
        else:
            a, w = self.self_attn.run(tgt, None, tgt, query_mask, b, t, t, True, self.p, self.training, q_pos=query_pos)
        tgt = Fk.add_layernorm(a, tgt, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        if isinstance(kv, tuple) and len(kv) == 2:  # (token, shared) of functional.cross_q1_memory: no key / value projection of the memory
            a, cw = self.cross_attn_image.run_q1(tgt, kv, memory_mask, b * t, S, True, self.p, self.training, q_pos=query_pos)
        elif kv is not None:
            a, cw = self.cross_attn_image.run_prekv(tgt, kv, index, memory_mask, b * t, 1, S, True, self.p, self.training, q_pos=query_pos)
        else:  # (TD_KV_HOIST=0, A/B only: per-layer key / value projections over a materialised memory + pos)
            mem_k = Fk.AddFn.apply(mem, pos) if pos is not None else mem
            a, cw = self.cross_attn_image.run(tgt, mem_k, mem, memory_mask, b * t, 1, S, True, self.p, self.training, q_pos=query_pos)
        tgt = Fk.add_layernorm(a, tgt, self.norm3.weight, self.norm3.bias, self.norm3.eps)
        f = Fk.ffn(tgt, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias, self.p, self.training)
        tgt = Fk.add_layernorm(f, tgt, self.norm4.weight, self.norm4.bias, self.norm4.eps)
        return tgt, w, cw


class TransformerDecoder(nn.Module):
    def __init__(self, decoder_layer, num_layers, norm=None, return_intermediate=False, return_weights=False):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(decoder_layer) for _ in range(num_layers)])
        self.num_layers = num_layers
        self.norm = norm
        self.return_intermediate = return_intermediate
        self.return_weights = return_weights

    def forward(self, tgt, query_pos, mem, pos, query_mask, memory_mask, b, t, S):
        inter, ws, cws = [], [], []
        out = tgt
        # one key and one value projection GEMM for the six layers' shared memory (functional.CrossKVFn); keys = memory + pos
        # with pos as the GEMM's second operand stream
        # Default: the time-aligned cross-attention has ONE query per frame, so its key / value projections move to the query side
        # and the memory rows are never projected (functional.CrossQ1Fn).  TD_CROSS_Q1=0 (A/B), learned position embeddings (pos
        # needs a gradient) or another width / head count: the projected-memory path, hoisted over the layers (TD_KV_HOIST=0: per layer).
        att = self.layers[0].cross_attn_image
        if (os.environ.get("TD_CROSS_Q1", "1") != "0" and att.embed_dim == 256 and att.num_heads == 8 and S <= 320  # (S: the frame core's LDS budget)
                and not (pos is not None and pos.requires_grad)):
            kv = Fk.cross_q1_memory(mem, pos)
        else:
            kv = Fk.cross_kv(mem, pos, [l.cross_attn_image for l in self.layers]) if os.environ.get("TD_KV_HOIST", "1") != "0" else None
        for i, layer in enumerate(self.layers):
            out, w, cw = layer(out, query_pos, mem, pos, query_mask, memory_mask, b, t, S, kv=kv, index=i)
            if self.return_intermediate:
                inter.append(Fk.add_layernorm(out, None, self.norm.weight, self.norm.bias, self.norm.eps))
                ws.append(w)
                cws.append(cw)
        if not self.return_intermediate:
            out = Fk.add_layernorm(out, None, self.norm.weight, self.norm.bias, self.norm.eps) if self.norm is not None else out
            return (out, w, cw) if self.return_weights else out
        hs = torch.stack(inter)
        if not self.return_weights:
            return hs
        return hs, torch.stack(ws), torch.stack(cws)


class FeatureResizer(nn.Module):
    """fc (768->256) + LayerNorm(eps=1e-12) + dropout (transformer.py:754-773)."""

    def __init__(self, input_feat_size, output_feat_size, dropout, do_ln=True):
        super().__init__()
        self.do_ln = do_ln
        self.fc = nn.Linear(input_feat_size, output_feat_size, bias=True)
        self.layer_norm = nn.LayerNorm(output_feat_size, eps=1e-12)
        self.p = dropout

    def forward(self, rows: Tensor) -> Tensor:
        x = Fk.linear(rows, self.fc.weight, self.fc.bias)
        if self.do_ln:
            x = Fk.add_layernorm(x, None, self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps)
        return Fk.dropout(x, self.p, self.training)


class HashTokenizer:
    """Offline stand-in for RobertaTokenizerFast when no roberta-base files exist on the box: words are hashed
    into the RoBERTa id range, <s>=0, </s>=2, <pad>=1.  Exposes the one method the model calls."""

    def batch_encode_plus(self, text: List[str], padding="longest", return_tensors="pt"):
        import zlib

        from transformers import BatchEncoding

        rows = [[0] + [3 + zlib.crc32(w.encode()) % 49990 for w in s.split()] + [2] for s in text]
        L = max(len(r) for r in rows)
        ids = torch.tensor([r + [1] * (L - len(r)) for r in rows], dtype=torch.long)
        att = torch.tensor([[1] * len(r) + [0] * (L - len(r)) for r in rows], dtype=torch.long)
        be = BatchEncoding({"input_ids": ids, "attention_mask": att})
        be._encodings = [None] * len(text)
        return be


def _load_text_encoder(name: str):
    """RobertaTokenizerFast / RobertaModel ``from_pretrained`` like the reference (transformer.py:130-135): a missing or
    corrupt roberta-base raises.  Only with ``TD_ALLOW_RANDOM_TEXT_ENCODER=1`` (set by bench.py, smoke() and the tests:
    no weight or tokenizer files exist offline) the stand-ins are used - always BOTH of them, a pretrained model is never
    paired with the hash tokenizer - and a warning says so."""
    import os
    import warnings

    from transformers import RobertaConfig, RobertaModel, RobertaTokenizerFast

    try:
        tok = RobertaTokenizerFast.from_pretrained(name, local_files_only=True)
        enc = RobertaModel.from_pretrained(name, local_files_only=True)
        return tok, enc
    except Exception as exc:
        if os.environ.get("TD_ALLOW_RANDOM_TEXT_ENCODER", "0") != "1":
            raise RuntimeError(
                f"text encoder {name!r} could not be loaded ({type(exc).__name__}: {exc}); the reference needs the pretrained "
                "files too.  For synthetic benchmarks / tests without them set TD_ALLOW_RANDOM_TEXT_ENCODER=1 (random-init "
                "roberta-base geometry + hashing tokenizer)") from exc
        warnings.warn(f"TD_ALLOW_RANDOM_TEXT_ENCODER=1: {name!r} is unavailable, using a RANDOM-INIT RobertaModel and a hashing "
                      "tokenizer (synthetic runs only)")
        enc = RobertaModel(RobertaConfig(vocab_size=50265, max_position_embeddings=514, type_vocab_size=1, pad_token_id=1, layer_norm_eps=1e-5))
        return HashTokenizer(), enc


class Transformer(nn.Module):
    def __init__(self, d_model=512, nhead=8, num_encoder_layers=6, num_decoder_layers=6, dim_feedforward=2048, dropout=0.1,
                 activation="relu", return_intermediate_dec=False, pass_pos_and_query=True, text_encoder_type="roberta-base",
                 freeze_text_encoder=False, video_max_len=0, stride=0, no_tsa=False, return_weights=False, fast=False,
                 fast_mode="", learn_time_embed=False, rd_init_tsa=False, no_time_embed=False):
        super().__init__()
        if fast and fast_mode not in FAST_MODES:
            raise ValueError(f"fast_mode={fast_mode!r}: expected one of {FAST_MODES}")
        # Ablation flags (SURVEY.md 8a'): accepted like the reference's constructor does.  fast_mode variants, learned
        # time embeddings and stride=0 run the step's kernels as usual and only their own extra arithmetic on stock
        # PyTorch-ROCm ops (_aggregate_variant); pass_pos_and_query=False is stored and fails at forward(), where the
        # reference itself fails (see forward).
        self.pass_pos_and_query = pass_pos_and_query
        enc_layer = TransformerEncoderLayer(d_model, nhead, dim_feedforward, dropout, activation)
        self.encoder = TransformerEncoder(enc_layer, num_encoder_layers, None, return_weights=True)
        dec_layer = TransformerDecoderLayer(d_model, nhead, dim_feedforward, dropout, activation, no_tsa=no_tsa)
        self.decoder = TransformerDecoder(dec_layer, num_decoder_layers, nn.LayerNorm(d_model),
                                          return_intermediate=return_intermediate_dec, return_weights=return_weights)
        self._reset_parameters()
        self.return_weights = return_weights
        self.learn_time_embed = learn_time_embed
        self.use_time_embed = not no_time_embed
        if self.use_time_embed:
            self.time_embed = TimeEmbeddingLearned(video_max_len, d_model) if learn_time_embed else TimeEmbeddingSine(video_max_len, d_model)
        self.fast = fast
        self.fast_mode = fast_mode
        if fast:  # transformer.py:110-122
            if fast_mode == "gating":
                self.fast_encoder = nn.Linear(d_model, d_model)
            elif fast_mode == "transformer":
                self.fast_encoder = TransformerEncoder(TransformerEncoderLayer(d_model, nhead, dim_feedforward, dropout, activation), 1,
                                                       nn.LayerNorm(d_model), return_weights=True)
                self.fast_residual = nn.Linear(d_model, d_model)
            else:
                self.fast_encoder = nn.Linear(d_model, d_model)
                self.fast_residual = nn.Linear(d_model, d_model)
        self.rd_init_tsa = rd_init_tsa
        self._reset_temporal_parameters()
        self.tokenizer, self.text_encoder = _load_text_encoder(text_encoder_type)
        if freeze_text_encoder:
            for p in self.text_encoder.parameters():
                p.requires_grad_(False)
        self.expander_dropout = 0.1
        self.resizer = FeatureResizer(self.text_encoder.config.hidden_size, d_model, self.expander_dropout)
        self.d_model, self.nhead = d_model, nhead
        self.sine_pos = None  # (num_pos_feats, temperature) of the backbone's PositionEmbeddingSine, set by TubeDETR: forward(pos_embed=None, pos_mask=...)
        self.video_max_len, self.stride = video_max_len, stride
        self.compute_dtype = torch.float32
        self.hip_text_encoder = __import__("os").environ.get("TD_HIP_ROBERTA", "1") != "0"  # 0: HF RobertaModel as a torch module
        self._idx_cache = LRUCache()

    # ---- init (transformer.py:154-176) ----
    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def _reset_temporal_parameters(self):  # transformer.py:159-176
        for n, p in self.named_parameters():
            if "fast_encoder" in n and self.fast_mode == "transformer":
                nn.init.constant_(p, 1.0 if ("norm" in n and "weight" in n) else 0.0)
            if self.rd_init_tsa and "decoder" in n and "self_attn" in n and p.dim() > 1:
                nn.init.xavier_uniform_(p)
            if "fast_residual" in n:
                nn.init.constant_(p, 0)
            if self.fast_mode == "gating" and "fast_encoder" in n:
                nn.init.constant_(p, 0)

    # ---- helpers ----
    def _replica_maps(self, durations, n_clips, owner, n, hw, L, device):
        key = ("maps", tuple(durations), n_clips, self.stride, hw, L, str(device))
        hit = self._idx_cache.get(key)
        if hit is None:
            hit = self._idx_cache[key] = Fk.ReplicaMaps(owner, n, hw, L, device)
        return hit

    def _indices(self, durations, n_clips_per_video: int, device):
        """owner clip of every (video, frame) and the per-clip / per-frame video index; cached per durations."""
        key = (tuple(durations), n_clips_per_video, str(device))
        hit = self._idx_cache.get(key)
        if hit is None:
            b, t, k = len(durations), max(durations), (self.stride or 1)  # stride 0: every frame is its own clip
            vid = torch.arange(b)
            owner = (vid[:, None] * n_clips_per_video + torch.arange(t)[None, :] // k).reshape(-1)
            query_mask = torch.ones(b, t, dtype=torch.bool)
            query_mask[:, 0] = False  # avoid empty masks (transformer.py:236)
            for i, dur in enumerate(durations):
                query_mask[i, :dur] = False
            voc = vid.repeat_interleave(n_clips_per_video)
            hit = (owner.to(device), voc.to(device), vid.repeat_interleave(t).to(device), query_mask.to(device), voc.tolist())
            self._idx_cache[key] = hit
        return hit

    def _encode_text(self, text, device):
        if isinstance(text[0], str):
            tokenized = self.tokenizer.batch_encode_plus(text, padding="longest", return_tensors="pt")
            ids, att = tokenized["input_ids"], tokenized["attention_mask"]
            # decided on the host copy: HF's mask construction otherwise inspects the device mask (a stream sync)
            hint = getattr(tokenized, "_td_no_padding", None)  # tokenizers may state it for device-resident ids
            no_padding = bool(hint) if hint is not None else (bool(att.all()) if att.device.type == "cpu" else False)
            main = torch.cuda.current_stream(device)
            import os

            if os.environ.get("TD_TEXT_STREAM", "1") == "0":  # keep the text encoder on the main stream (single-stream HIP graph)
                side = main
            else:
                side = self._text_stream = getattr(self, "_text_stream", None) or torch.cuda.Stream(device)
            # RoBERTa (hundreds of tiny launches on 30 tokens) is independent of the visual backbone: it runs on its own
            # HIP stream, concurrently with the trunk's large GEMM kernels; autograd replays its backward there too.
            with torch.cuda.stream(side):
                if ids.device.type == "cpu":  # async H2D from pinned memory, on the side stream: no sync with the trunk
                    if self.hip_text_encoder and side is not main:
                        # the HIP RoBERTa reads this step's prepared (re-cast) weights, which the main stream refreshed in the
                        # backbone forward: order the side stream behind that launch (the HF module reads the fp32 parameters)
                        side.wait_stream(main)
                    tokenized["input_ids"] = ids.pin_memory().to(device, non_blocking=True)
                    tokenized["attention_mask"] = att.pin_memory().to(device, non_blocking=True)
                else:
                    side.wait_stream(main)
                    tokenized = tokenized.to(device)
                    if side is not main:
                        # The ids were allocated on the MAIN stream but RoBERTa reads them on the text stream - in the forward
                        # and again in the embedding backward (scatter-add by token id).  Without this the caching
                        # allocator hands their block back to the main stream the moment autograd releases the saved tensor,
                        # a main-stream kernel overwrites it while the embedding backward is still queued on the text
                        # stream, and the scatter-add runs with garbage row indices: the GPU memory fault of round 1
                        # ("memory-fault investigation").
                        tokenized["input_ids"].record_stream(side)
                        tokenized["attention_mask"].record_stream(side)
                if self.hip_text_encoder:  # RoBERTa on this package's kernels (models/text_encoder.py), result in the compute dtype
                    from .text_encoder import run_roberta

                    hidden_side = run_roberta(self.text_encoder, tokenized["input_ids"], tokenized["attention_mask"], self.compute_dtype,
                                              self.training, no_padding=no_padding)
                else:  # HF module as a stock PyTorch-ROCm graph (fp32)
                    enc = self.text_encoder(input_ids=tokenized["input_ids"], attention_mask=None if no_padding else tokenized["attention_mask"])
                    hidden_side = enc.last_hidden_state
            main.wait_stream(side)
            for t_ in (hidden_side, tokenized["input_ids"], tokenized["attention_mask"]):
                t_.record_stream(main)
            hidden = hidden_side  # (B, L, 768)
            Bt, L, _ = hidden.shape
            rows = Fk.cast(hidden.reshape(Bt * L, -1), self.compute_dtype)
            resized = self.resizer(rows).view(Bt, L, -1)  # batch-major [B, L, d]
            return tokenized["attention_mask"].ne(1), resized, tokenized
        mask, resized_seq_first, tokenized = text  # pre-encoded triple (transformer.py:264-266), (L,B,d)
        return mask, Fk.cast(resized_seq_first.transpose(0, 1).contiguous(), self.compute_dtype), tokenized

    def forward(self, src=None, mask=None, query_embed=None, pos_embed=None, text=None, encode_and_save=True, durations=None,
                tpad_mask_t=None, fast_src=None, img_memory=None, query_mask=None, text_memory=None, text_mask=None,
                memory_mask=None, pos_mask=None):
        if not self.pass_pos_and_query:
            # the reference sets pos_embed = None in this mode (transformer.py:242-248) and then concatenates it with the text
            # rows (:325, TypeError), and its decode branch adds to a `src` that is None (:463-469): the flag cannot complete a
            # step there either, so the same failure class is raised here instead of silently running something else
            raise TypeError("pass_pos_and_query=False: the reference's Transformer.forward fails in this mode (models/transformer.py:242-248 -> 325, 463-469)")
        if encode_and_save:
            return self._encode(src, mask, query_embed, pos_embed, text, durations, tpad_mask_t, fast_src, pos_mask)
        return self._decode(img_memory, mask, pos_embed, query_embed, query_mask)

    # ---- encode (transformer.py:195-460) ----
    def _encode(self, src, mask, query_embed, pos_embed, text, durations, tpad_mask_t, fast_src, pos_mask=None):
        n, d, h, w = src.shape  # (n_clips_total, d, h, w) channels-last view of NHWC rows
        dev, dt = src.device, self.compute_dtype
        hw = h * w
        b, t = len(durations), max(durations)
        n_clips = math.ceil(t / self.stride) if self.stride else t  # stride 0 (dense ablation): one clip per frame, no replication
        assert n == b * n_clips, "every video of the batch must yield the same number of slow clips"
        src_bm = src.permute(0, 2, 3, 1).reshape(n, hw, d)  # zero-copy when src is channels-last
        # all index / mask tensors of a (durations) pattern are built once and stay on the device: a host->device copy
        # inside the step is a stream synchronisation point
        owner, vid_of_clip, vid_of_frame, query_mask, clip_vid_list = self._indices(durations, n_clips, dev)

        # time queries (transformer.py:211-238): identical for every video, video-major rows [b*t, d]
        nq = query_embed.shape[0]
        if nq != 1:
            raise NotImplementedError("num_queries > 1 is outside the HIP hot path")
        q = query_embed[0].float()
        qpos_t = (q[None, :] + self.time_embed(t)[:, 0, :]) if self.use_time_embed else q[None, :].expand(t, -1)
        query_pos_bm = qpos_t[None].expand(b, t, d)  # fp32, autograd reaches query_embed.weight
        query_mask = query_mask.clone() if self.stride else None  # transformer.py:225-238: no time-query mask without temporal sampling

        text_attention_mask_orig, text_resized, tokenized = self._encode_text(text, dev)  # [B,L], [B,L,d]
        L = text_resized.shape[1]
        assert n_clips == n // text_resized.shape[0] == mask.shape[0] // text_attention_mask_orig.shape[0]
        text_clip = text_resized[vid_of_clip]  # [n, L, d]
        text_mask_clip = text_attention_mask_orig[vid_of_clip]
        self._repeat_tokenized(tokenized, vid_of_clip, clip_vid_list)

        S = hw + L
        x = torch.cat([src_bm.to(dt), text_clip], dim=1)  # [n, S, d]
        sine = pos_embed is None
        if sine:
            # PositionEmbeddingSine (position_encoding.py:71-94) straight from the pad mask, S rows per clip with the text
            # tokens' zero rows already in place (transformer.py:323-326) - one kernel, no tensor to concatenate zeros to
            assert self.sine_pos is not None and pos_mask is not None, "pos_embed=None needs the sine encoding's pad mask (pos_mask)"
            pos_full = _ops.pos_sine(pos_mask, self.sine_pos[0], dt, self.sine_pos[1], rows=S)
        else:  # a positional tensor handed in by the caller (learned encodings, the dense --stride 0 path, external callers)
            pos_bm = pos_embed.permute(0, 2, 3, 1).reshape(n, hw, d)
            pos_full = torch.cat([pos_bm.to(dt), torch.zeros(n, L, d, dtype=dt, device=dev)], dim=1)
        key_pad = torch.cat([mask.flatten(1), text_mask_clip], dim=1).to(torch.uint8)  # [n, S], 1 = ignore
        if self.fast and self.fast_mode == "noslow":  # no space-text attention for this baseline (transformer.py:330-340)
            mem = x
        else:
            mem = self.encoder(x.reshape(n * S, d), key_pad, pos_full.reshape(n * S, d), n, S).view(n, S, d)

        variant = self.fast and self.fast_mode not in FAST_MODES_IN_HIP
        if self.fast and fast_src is None:
            raise AttributeError("fast=True needs temporal sampling (stride > 0): the reference builds no fast_src without it (models/tubedetr.py:140-153)")
        if self.stride:
            # temporal replication (transformer.py:393-427): frame (i, j) <- clip i*n_clips + j//k, as index vectors
            maps = self._replica_maps(durations, n_clips, owner, n, hw, L, dev)
            frame_mask = torch.cat([tpad_mask_t.flatten(1), text_attention_mask_orig[vid_of_frame]], dim=1)  # [b*t, S]
            frame_mask[:, 0] = False  # "avoid empty masks" (transformer.py:424)
            if sine:  # the frames' positional rows come from the same kernel (a frame has its clip's pad mask), not from a gather of pos_full
                frames_pos = _ops.pos_sine(pos_mask[owner], self.sine_pos[0], dt, self.sine_pos[1], rows=S)
            else:
                frames_pos = Fk.ReplicateRowsFn.apply(pos_full.reshape(n * S, d), maps).view(b * t, S, d)
            mem2d = mem.reshape(n * S, d)
            if self.fast and not variant:  # transformer.py:373-375,387,441-445: replication + aggregation in one GEMM
                fs = fast_src.permute(0, 2, 3, 1).reshape(b * t * hw, d)
                fast_mem = Fk.linear(fs.to(dt), self.fast_encoder.weight, self.fast_encoder.bias)
                frames_mem = Fk.SlowFastAggregateFn.apply(mem2d, fast_mem, self.fast_residual.weight, self.fast_residual.bias, maps).view(b * t, S, d)
            else:
                frames_mem = Fk.ReplicateRowsFn.apply(mem2d, maps).view(b * t, S, d)
        else:
            frames_mem, frames_pos, frame_mask = mem, pos_full, key_pad.bool()
        if variant:
            frames_mem = self._aggregate_variant(frames_mem, fast_src, tpad_mask_t, text_resized, vid_of_frame, b, t, hw, d)
        return {
            "text_memory_resized": text_clip.transpose(0, 1),  # (L, n, d) seq-first views like the reference
            "text_memory": frames_mem[:, hw:].transpose(0, 1),
            "text_attention_mask": text_mask_clip,
            "tokenized": tokenized,
            "img_memory": frames_mem.transpose(0, 1),  # (S, b*t, d)
            "mask": frame_mask,
            "pos_embed": frames_pos.transpose(0, 1),
            "query_embed": query_pos_bm.transpose(0, 1),  # (t, b, d)
            "query_mask": query_mask,
        }

    def _aggregate_variant(self, frames_mem, fast_src, tpad_mask_t, text_resized, vid_of_frame, b, t, hw, d):
        """--fast_mode gating | transformer | pool | noslow (transformer.py:341-387, 429-445): ablation variants outside the
        kernel scope (SURVEY.md 8a').  The fast features enter as they do by default; the variant's own arithmetic - the
        masked spatial pooling, the gate, the aggregation - runs on stock PyTorch-ROCm ops with their autograd; the
        temporal transformer of the "transformer" variant is this package's encoder layer over the time axis."""
        import torch.nn.functional as F

        dt, mode = self.compute_dtype, self.fast_mode
        fs = fast_src.permute(0, 2, 3, 1).reshape(b * t, hw, d).to(dt)
        vis, txt = frames_mem[:, :hw], frames_mem[:, hw:]

        def lin(x_, m_):
            return F.linear(x_, m_.weight.to(x_.dtype), m_.bias.to(x_.dtype))

        if mode == "transformer":  # sequences over time, one per (video, pixel): (b*t, hw, d) -> rows [(b, hw), t]
            seq = fs.view(b, t, hw, d).permute(0, 2, 1, 3).reshape(b * hw * t, d).contiguous()
            te = Fk.cast(self.time_embed(t)[:, 0, :].float(), dt)
            pos = te[None].expand(b * hw, t, d).reshape(b * hw * t, d).contiguous()
            y = self.fast_encoder(seq, None, pos, b * hw, t)
            fast_mem = y.view(b, hw, t, d).permute(0, 2, 1, 3).reshape(b * t, hw, d)
        elif mode == "pool":  # masked mean over the frame's pixels, then the linear layer, broadcast back
            keep = (~tpad_mask_t.flatten(1))[:, :, None]
            cnt = keep.float().sum(dim=1).clamp(min=1)
            pooled = torch.div((fs * keep).sum(dim=1), cnt.to(fs.dtype))
            fast_mem = lin(pooled, self.fast_encoder)[:, None, :].expand(b * t, hw, d)
        else:
            fast_mem = lin(fs, self.fast_encoder)
        if mode == "noslow":
            vis, txt = fast_mem, text_resized[vid_of_frame]
        elif mode == "gating":
            vis = vis + vis * torch.sigmoid(fast_mem)
        else:
            vis = vis + lin(vis + fast_mem, self.fast_residual)
        return torch.cat([vis, txt], dim=1)

    @staticmethod
    def _repeat_tokenized(tokenized, vid_of_clip, clip_vid_list):
        """The reference repeats the BatchEncoding per clip in place (transformer.py:275-308)."""
        try:
            if getattr(tokenized, "_encodings", None) is not None:
                tokenized._encodings = [tokenized._encodings[i] for i in clip_vid_list]
            sel = vid_of_clip.to(tokenized["input_ids"].device)
            tokenized["input_ids"] = tokenized["input_ids"][sel]
            tokenized["attention_mask"] = tokenized["attention_mask"][sel]
        except Exception:
            pass

    # ---- decode (transformer.py:462-491) ----
    def _decode(self, img_memory, mask, pos_embed, query_embed, query_mask):
        dt = self.compute_dtype
        S, bt, d = img_memory.shape
        t, b, _ = query_embed.shape
        mem = img_memory.transpose(0, 1).reshape(bt * S, d)  # zero-copy for the cache produced by _encode
        pos = pos_embed.transpose(0, 1).reshape(bt * S, d)
        mem, pos = Fk.cast(mem, dt), Fk.cast(pos, dt)  # keys = memory + pos: pos is the key projection's second operand stream
        query_pos = Fk.cast(query_embed.transpose(0, 1).reshape(b * t, d).contiguous(), dt)
        tgt = torch.zeros_like(query_pos)
        res = self.decoder(tgt, query_pos, mem, pos, query_mask, mask, b, t, S)
        if self.return_weights:
            hs, weights, cross_weights = res
        else:
            hs = res
        hs = hs.view(hs.shape[0], b, t, d)  # the reference's hs.transpose(1, 2): (layers, b, t, d)
        if not self.return_weights:
            return hs
        return hs, weights, cross_weights


def build_transformer(args):
    return Transformer(
        d_model=args.hidden_dim, dropout=args.dropout, nhead=args.nheads, dim_feedforward=args.dim_feedforward,
        num_encoder_layers=args.enc_layers, num_decoder_layers=args.dec_layers, return_intermediate_dec=True,
        pass_pos_and_query=args.pass_pos_and_query, text_encoder_type=args.text_encoder_type,
        freeze_text_encoder=args.freeze_text_encoder, video_max_len=args.video_max_len_train, stride=args.stride,
        no_tsa=args.no_tsa, return_weights=args.guided_attn, fast=args.fast, fast_mode=args.fast_mode,
        learn_time_embed=args.learn_time_embed, rd_init_tsa=args.rd_init_tsa, no_time_embed=args.no_time_embed,
    )
